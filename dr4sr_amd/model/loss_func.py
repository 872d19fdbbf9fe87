"""Loss modules with the call signatures of the reference's model/loss_func.py, computed by HIP kernels.

Inside the training hot path the scores never exist as tensors: BaseModel.training_step hands the QUERY to the fused scorer
(dr4sr_score_bce_* / dr4sr_score_bpr_*: gather of the target / negative table rows, dots, loss, and in the fused step the whole
backward).  The modules below are the reference's second entry point — the loss called on score tensors a caller already has —
through dr4sr_loss_from_scores_fwd/_bwd (include/dr4sr_hip.h):

  BinaryCrossEntropyLoss.forward(pos, neg, reduce=True)   loss_func.py:9-38: the masked branch (pos.dim() == neg.dim() - 1) and the
                                                          plain-mean branch :33 (pos [B, L] with neg [B, K]: the negatives' term is
                                                          one scalar mean, no padding mask on it) — the latter composed of two calls
                                                          of the same kernel (round 3; unreachable from training_step)
  BPRLoss.forward(pos, neg)                               loss_func.py:44-49.  As in the reference it has NO `reduce` parameter, so the
                                                          reference's own training_step (basemodel.py:210 passes reduce=) raises
                                                          TypeError when loss_fn: 'bpr' is configured; BaseModel.training_step here
                                                          keeps that TypeError for reduce=False (there is no per-position BPR in the
                                                          reference to match) and computes the documented scalar for reduce=True.
"""
import torch
import torch.nn as nn

from .. import _lib


class _LossFromScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, kind, reduce):
        lib = _lib.load()
        if not pos.is_cuda:
            raise _lib.Dr4srError("dr4sr_amd loss modules run on the GPU only (no CPU path)")
        if neg.dim() != pos.dim() + 1:
            raise NotImplementedError("loss on scores: pos [...], neg [..., K] (loss_func.py:26); BinaryCrossEntropyLoss also takes the "
                                      "plain-mean layout pos [B, L], neg [B, K] (loss_func.py:33)")
        ctx.in_dtypes = (pos.dtype, neg.dtype)
        p, ng = pos.detach().contiguous().float(), neg.detach().contiguous().float()
        n, K = p.numel(), int(ng.shape[-1])
        lp = torch.empty(n, dtype=torch.float32, device=p.device)
        stats = torch.zeros(2, dtype=torch.float32, device=p.device)
        _lib.check(lib.dr4sr_loss_from_scores_fwd(_lib.ptr(p), _lib.ptr(ng), n, K, kind, _lib.ptr(lp), _lib.ptr(stats),
                                                  _lib.cur_stream()), "dr4sr_loss_from_scores_fwd")
        ctx.kind, ctx.reduce, ctx.K = kind, reduce, K
        ctx.save_for_backward(p, ng, stats)
        if reduce:
            return stats[1] / stats[0]
        return (lp / stats[0]).view(pos.shape)

    @staticmethod
    def backward(ctx, gout):
        p, ng, stats = ctx.saved_tensors
        lib = _lib.load()
        n = p.numel()
        dp, dn = torch.empty_like(p), torch.empty_like(ng)
        if ctx.reduce:
            g, scale = None, (gout.reshape(1) / stats[0]).contiguous().float()
        else:
            g, scale = gout.contiguous().view(-1).float(), (1.0 / stats[0]).reshape(1).contiguous()
        _lib.check(lib.dr4sr_loss_from_scores_bwd(_lib.ptr(p), _lib.ptr(ng), n, ctx.K, ctx.kind, _lib.ptr(g), _lib.ptr(scale),
                                                  _lib.ptr(dp), _lib.ptr(dn), _lib.cur_stream()), "dr4sr_loss_from_scores_bwd")
        return dp.to(ctx.in_dtypes[0]), dn.to(ctx.in_dtypes[1]), None, None     # gradients in the inputs' dtypes (half / bf16 scores)


class _BcePlainMean(torch.autograd.Function):
    """loss_func.py:33 — pos [B, L] (or any shape; -inf marks padding) with neg [B, K] of the SAME rank: the positives keep the masked,
    n_valid-normalised form (:14-21), the negatives' term is torch.mean over rows of sum_k softplus(neg_k) / K — a scalar added to every
    position when reduce=False.  Two calls of dr4sr_loss_from_scores_fwd: positives against a -inf negative (softplus = 0), negatives
    against a +1e30 positive (-logsigmoid = 0), so each call returns exactly one of the two terms."""

    @staticmethod
    def forward(ctx, pos, neg, reduce):
        lib = _lib.load()
        if not pos.is_cuda:
            raise _lib.Dr4srError("dr4sr_amd loss modules run on the GPU only (no CPU path)")
        p, ng = pos.detach().contiguous().float(), neg.detach().contiguous().float()
        n, rows, K = p.numel(), ng.numel() // int(ng.shape[-1]), int(ng.shape[-1])
        dev = p.device
        ninf = torch.full((n, 1), float("-inf"), dtype=torch.float32, device=dev)
        big = torch.full((rows,), 1e30, dtype=torch.float32, device=dev)
        lp, ln = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(rows, dtype=torch.float32, device=dev)
        sp, sn = torch.zeros(2, dtype=torch.float32, device=dev), torch.zeros(2, dtype=torch.float32, device=dev)
        _lib.check(lib.dr4sr_loss_from_scores_fwd(_lib.ptr(p), _lib.ptr(ninf), n, 1, 0, _lib.ptr(lp), _lib.ptr(sp), _lib.cur_stream()), "loss_from_scores_fwd")
        _lib.check(lib.dr4sr_loss_from_scores_fwd(_lib.ptr(big), _lib.ptr(ng), rows, K, 0, _lib.ptr(ln), _lib.ptr(sn), _lib.cur_stream()), "loss_from_scores_fwd")
        ctx.reduce, ctx.K, ctx.rows, ctx.in_dtypes = reduce, K, rows, (pos.dtype, neg.dtype)
        ctx.save_for_backward(p, ng, sp, ninf, big)
        neg_mean = sn[1] / rows
        if reduce:
            return sp[1] / sp[0] + neg_mean
        return (lp / sp[0]).view(pos.shape) + neg_mean

    @staticmethod
    def backward(ctx, gout):
        p, ng, sp, ninf, big = ctx.saved_tensors
        lib = _lib.load()
        n, rows = p.numel(), ctx.rows
        dp, dn = torch.empty_like(p), torch.empty_like(ng)
        scr_n, scr_p = torch.empty_like(ninf), torch.empty_like(big)
        gsum = gout.float().sum().reshape(1)               # the scalar negatives' term receives the SUM of the upstream gradient
        if ctx.reduce:
            g, scale = None, (gout.reshape(1).float() / sp[0]).contiguous()
        else:
            g, scale = gout.contiguous().view(-1).float(), (1.0 / sp[0]).reshape(1).contiguous()
        _lib.check(lib.dr4sr_loss_from_scores_bwd(_lib.ptr(p), _lib.ptr(ninf), n, 1, 0, _lib.ptr(g), _lib.ptr(scale), _lib.ptr(dp), _lib.ptr(scr_n),
                                                  _lib.cur_stream()), "loss_from_scores_bwd")
        scale_n = (gsum / rows).contiguous()
        _lib.check(lib.dr4sr_loss_from_scores_bwd(_lib.ptr(big), _lib.ptr(ng), rows, ctx.K, 0, None, _lib.ptr(scale_n), _lib.ptr(scr_p), _lib.ptr(dn),
                                                  _lib.cur_stream()), "loss_from_scores_bwd")
        return dp.to(ctx.in_dtypes[0]), dn.to(ctx.in_dtypes[1]), None


class BinaryCrossEntropyLoss(nn.Module):
    name = "bce"
    kind = 0

    def forward(self, pos_score, neg_score, reduce=True):
        if pos_score.dim() == neg_score.dim():              # loss_func.py:32-33: no padding mask on the negatives, plain mean
            return _BcePlainMean.apply(pos_score, neg_score, bool(reduce))
        return _LossFromScores.apply(pos_score, neg_score, 0, bool(reduce))


class BPRLoss(nn.Module):
    name = "bpr"
    kind = 1

    def forward(self, pos_score, neg_score):
        return _LossFromScores.apply(pos_score, neg_score, 1, True)
