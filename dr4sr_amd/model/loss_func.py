"""Loss modules with the call signatures of the reference's model/loss_func.py, computed by HIP kernels.

Inside the training hot path the scores never exist as tensors: BaseModel.training_step hands the QUERY to the fused scorer
(dr4sr_score_bce_* / dr4sr_score_bpr_*: gather of the target / negative table rows, dots, loss, and in the fused step the whole
backward).  The modules below are the reference's second entry point — the loss called on score tensors a caller already has —
through dr4sr_loss_from_scores_fwd/_bwd (include/dr4sr_hip.h):

  BinaryCrossEntropyLoss.forward(pos, neg, reduce=True)   loss_func.py:9-38, masked branch (pos.dim() == neg.dim() - 1; the plain-mean
                                                          branch :33 is unreachable from training_step and is not built)
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
            raise NotImplementedError("loss on scores: only the masked branch pos [...], neg [..., K] (loss_func.py:26) is built")
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
        return dp, dn, None, None


class BinaryCrossEntropyLoss(nn.Module):
    name = "bce"
    kind = 0

    def forward(self, pos_score, neg_score, reduce=True):
        return _LossFromScores.apply(pos_score, neg_score, 0, bool(reduce))


class BPRLoss(nn.Module):
    name = "bpr"
    kind = 1

    def forward(self, pos_score, neg_score):
        return _LossFromScores.apply(pos_score, neg_score, 1, True)
