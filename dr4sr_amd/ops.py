"""PyTorch custom-op registration of the dense C-ABI entry points: `torch.ops.dr4sr_hip.*`.

SURVEY.md §8(b) / BASELINE north_star describe the native layer as "PyTorch-ROCm custom ops".  The compute library itself stays a
torch-free C ABI (include/dr4sr_hip.h — that is what makes it bindable from anything); this module registers the stateless,
tensor-in / tensor-out subset of it with the torch dispatcher through `torch.library`, so that the ops are visible as
`torch.ops.dr4sr_hip.<name>`, carry schemas, fake (meta) kernels for shape inference and autograd formulas, and can be called from
a reference-style model written against plain torch ops:

    embed_gather_posadd(E, P, idx) -> x                       model/sasrec.py:43-46,:64   (bit-exact with torch)
    score_bce(query, E, target, neg) -> (loss_pos, stats)     model/basemodel.py:204-214 + loss_func.py:9-38  (autograd: d query, d E)
    score_bpr(query, E, target, neg) -> (loss_pos, stats)     ... + loss_func.py:40-48
    loss_from_scores(pos, neg, kind) -> (loss_pos, stats)     model/loss_func.py on score tensors (autograd: d pos, d neg)
    neg_sample(n, n_items, seed, step) -> ids                 model/basemodel.py:50-61
    full_score_topk(q, E, hist, item_blocked, k) -> (score, ids)   model/basemodel.py:354-365
    fused_adam_(params, grads, m, v, state, lr, b1, b2, eps, wd)   torch.optim.Adam on flat buffers (in place)

Every implementation only enqueues HIP kernels of libdr4sr_hip.so on the current stream; there is no CPU kernel (the ops raise on
CPU tensors).  The plan-based entry points (whole fused training steps, encoders with their workspaces) are driven by the engines
(dr4sr_amd/engine.py ...) and are deliberately not dispatcher ops: they own device state (workspace, RNG step, Adam step).
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch

from . import _lib

NS = "dr4sr_hip"


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Dr4srError("torch.ops.dr4sr_hip.* run on the GPU only (libdr4sr_hip.so has no CPU path)")


# ------------------------------------------------------------------------------------------------ embed_gather_posadd
@torch.library.custom_op(f"{NS}::embed_gather_posadd", mutates_args=())
def embed_gather_posadd(E: torch.Tensor, P: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _need_cuda(E, P, idx)
    lib = _lib.load()
    B, L = idx.shape
    N, D = E.shape
    E, P, idx = E.contiguous(), P.contiguous(), idx.contiguous()
    # nn.Embedding's IndexError (the kernel itself clamps).  The check is an extra launch + a blocking device-to-host read: skipped
    # inside a stream capture (a synchronisation there raises) and under DR4SR_NO_ID_CHECK — check the id tensor once up front then
    # (BaseModel._check_dataset_ids does so for every split's resident tensors)
    if not torch.cuda.is_current_stream_capturing() and not os.environ.get("DR4SR_NO_ID_CHECK"):
        _lib.check_ids(idx, N, "embed_gather_posadd")
    out = torch.empty(B, L, D, dtype=torch.float32, device=E.device)
    _lib.check(lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx), _lib.ptr(out), B, L, D, N, _lib.cur_stream()),
               "dr4sr_embed_gather_posadd")
    return out


@embed_gather_posadd.register_fake
def _(E, P, idx):
    return E.new_empty(idx.shape[0], idx.shape[1], E.shape[1])


# ------------------------------------------------------------------------------------------------ scorer + loss
def _score_fwd(kind: int, query, E, target, neg):
    _need_cuda(query, E, target, neg)
    lib = _lib.load()
    B = int(target.shape[0])
    L = int(target.shape[1]) if target.dim() == 2 else 1
    D = int(E.shape[1])
    q, Ec, tg, ng = query.contiguous(), E.contiguous(), target.contiguous(), neg.contiguous()
    lp = torch.empty(B * L, dtype=torch.float32, device=q.device)
    stats = torch.zeros(2, dtype=torch.float32, device=q.device)
    fn = lib.dr4sr_score_bpr_fwd if kind else lib.dr4sr_score_bce_fwd
    _lib.check(fn(_lib.ptr(q), _lib.ptr(Ec), _lib.ptr(tg), _lib.ptr(ng), None, None, _lib.ptr(lp), _lib.ptr(stats), B, L, D,
                  _lib.cur_stream()), "dr4sr_score_fwd")
    return lp.view(target.shape), stats


def _score_bwd(kind: int, query, E, target, neg, g_loss_pos):
    lib = _lib.load()
    B = int(target.shape[0])
    L = int(target.shape[1]) if target.dim() == 2 else 1
    D = int(E.shape[1])
    q, Ec, tg, ng = query.contiguous(), E.contiguous(), target.contiguous(), neg.contiguous()
    w = g_loss_pos.contiguous().view(-1).float()
    dq = torch.empty_like(q)
    dE = torch.zeros_like(Ec)
    fn = lib.dr4sr_score_bpr_bwd if kind else lib.dr4sr_score_bce_bwd
    _lib.check(fn(_lib.ptr(q), _lib.ptr(Ec), _lib.ptr(tg), _lib.ptr(ng), _lib.ptr(w), None, _lib.ptr(dq), _lib.ptr(dE), B, L, D,
                  _lib.cur_stream()), "dr4sr_score_bwd")
    return dq.view(query.shape), dE


def _make_scorer(name: str, kind: int):
    @torch.library.custom_op(f"{NS}::{name}", mutates_args=())
    def fwd(query: torch.Tensor, E: torch.Tensor, target: torch.Tensor, neg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return _score_fwd(kind, query, E, target, neg)

    @fwd.register_fake
    def _(query, E, target, neg):
        return query.new_empty(target.shape), query.new_empty(2)

    @torch.library.custom_op(f"{NS}::{name}_backward", mutates_args=())
    def bwd(query: torch.Tensor, E: torch.Tensor, target: torch.Tensor, neg: torch.Tensor,
            g_loss_pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return _score_bwd(kind, query, E, target, neg, g_loss_pos)

    @bwd.register_fake
    def _(query, E, target, neg, g_loss_pos):
        return torch.empty_like(query), torch.empty_like(E)

    def setup(ctx, inputs, output):
        ctx.save_for_backward(*inputs)

    def backward(ctx, g_lp, g_stats):
        query, E, target, neg = ctx.saved_tensors
        dq, dE = bwd(query, E, target, neg, g_lp)
        return dq, dE, None, None

    fwd.register_autograd(backward, setup_context=setup)
    return fwd, bwd


score_bce, score_bce_backward = _make_scorer("score_bce", 0)
score_bpr, score_bpr_backward = _make_scorer("score_bpr", 1)


@torch.library.custom_op(f"{NS}::loss_from_scores", mutates_args=())
def loss_from_scores(pos: torch.Tensor, neg: torch.Tensor, kind: int) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_cuda(pos, neg)
    lib = _lib.load()
    p, ng = pos.contiguous().float(), neg.contiguous().float()
    n, K = p.numel(), int(ng.shape[-1])
    lp = torch.empty(n, dtype=torch.float32, device=p.device)
    stats = torch.zeros(2, dtype=torch.float32, device=p.device)
    _lib.check(lib.dr4sr_loss_from_scores_fwd(_lib.ptr(p), _lib.ptr(ng), n, K, kind, _lib.ptr(lp), _lib.ptr(stats), _lib.cur_stream()),
               "dr4sr_loss_from_scores_fwd")
    return lp.view(pos.shape), stats


@loss_from_scores.register_fake
def _(pos, neg, kind):
    return pos.new_empty(pos.shape), pos.new_empty(2)


@torch.library.custom_op(f"{NS}::loss_from_scores_backward", mutates_args=())
def loss_from_scores_backward(pos: torch.Tensor, neg: torch.Tensor, kind: int, g_loss_pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    p, ng = pos.contiguous().float(), neg.contiguous().float()
    n, K = p.numel(), int(ng.shape[-1])
    dp, dn = torch.empty_like(p), torch.empty_like(ng)
    g = g_loss_pos.contiguous().view(-1).float()
    _lib.check(lib.dr4sr_loss_from_scores_bwd(_lib.ptr(p), _lib.ptr(ng), n, K, kind, _lib.ptr(g), None, _lib.ptr(dp), _lib.ptr(dn),
                                              _lib.cur_stream()), "dr4sr_loss_from_scores_bwd")
    return dp.view(pos.shape), dn.view(neg.shape)


@loss_from_scores_backward.register_fake
def _(pos, neg, kind, g_loss_pos):
    return torch.empty_like(pos), torch.empty_like(neg)


def _lfs_setup(ctx, inputs, output):
    pos, neg, kind = inputs
    ctx.kind = kind
    ctx.save_for_backward(pos, neg)


def _lfs_backward(ctx, g_lp, g_stats):
    pos, neg = ctx.saved_tensors
    dp, dn = loss_from_scores_backward(pos, neg, ctx.kind, g_lp)
    return dp, dn, None


loss_from_scores.register_autograd(_lfs_backward, setup_context=_lfs_setup)


# ------------------------------------------------------------------------------------------------ neg_sample / topk / adam
@torch.library.custom_op(f"{NS}::neg_sample", mutates_args=(), device_types="cuda")
def neg_sample(n: int, n_items: int, seed: int, step: int, like: torch.Tensor) -> torch.Tensor:
    """`like` only supplies the device (custom ops need a tensor input to dispatch on)"""
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.int64, device=like.device)
    _lib.check(lib.dr4sr_neg_sample(_lib.ptr(out), n, n_items, seed & 0xFFFFFFFFFFFFFFFF, step & 0xFFFFFFFF, _lib.cur_stream()), "dr4sr_neg_sample")
    return out


@neg_sample.register_fake
def _(n, n_items, seed, step, like):
    return like.new_empty(n, dtype=torch.int64)


@torch.library.custom_op(f"{NS}::full_score_topk", mutates_args=())
def full_score_topk(q: torch.Tensor, E: torch.Tensor, hist: Optional[torch.Tensor], item_blocked: Optional[torch.Tensor],
                    k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_cuda(q, E, hist, item_blocked)
    lib = _lib.load()
    B, D = q.shape
    N = int(E.shape[0])
    qc, Ec = q.contiguous(), E.contiguous()
    hc = hist.contiguous() if hist is not None else None
    bc = item_blocked.contiguous() if item_blocked is not None else None
    score = torch.empty(B, k, dtype=torch.float32, device=q.device)
    ids = torch.empty(B, k, dtype=torch.int64, device=q.device)
    nb = int(lib.dr4sr_full_score_topk_workspace_bytes(B, N))
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=q.device)
    _lib.check(lib.dr4sr_full_score_topk_masked_ws(_lib.ptr(qc), _lib.ptr(Ec), _lib.ptr(hc), _lib.ptr(bc), _lib.ptr(score), _lib.ptr(ids), B, D,
                                                   N, int(hc.shape[1]) if hc is not None else 0, k, _lib.ptr(ws), nb, _lib.cur_stream()),
               "dr4sr_full_score_topk_masked_ws")
    return score, ids


@full_score_topk.register_fake
def _(q, E, hist, item_blocked, k):
    return q.new_empty(q.shape[0], k), q.new_empty(q.shape[0], k, dtype=torch.int64)


@torch.library.custom_op(f"{NS}::fused_adam_", mutates_args=("params", "adam_m", "adam_v", "state"))
def fused_adam_(params: torch.Tensor, grads: torch.Tensor, adam_m: torch.Tensor, adam_v: torch.Tensor, state: torch.Tensor,
                lr: float, beta1: float, beta2: float, eps: float, weight_decay: float) -> None:
    """torch.optim.Adam on flat fp32 buffers; grads has n + 4 floats, grads[n] = the normaliser the gradient is divided by
    (include/dr4sr_hip.h: dr4sr_adam_flat); state = the engine's int32 device words (state[0] = step counter)"""
    _need_cuda(params, grads, adam_m, adam_v, state)
    lib = _lib.load()
    _lib.check(lib.dr4sr_adam_flat(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(adam_m), _lib.ptr(adam_v), params.numel(), _lib.ptr(state),
                                   lr, beta1, beta2, eps, weight_decay, _lib.cur_stream()), "dr4sr_adam_flat")


OPS = ("embed_gather_posadd", "score_bce", "score_bce_backward", "score_bpr", "score_bpr_backward", "loss_from_scores",
       "loss_from_scores_backward", "neg_sample", "full_score_topk", "fused_adam_")
