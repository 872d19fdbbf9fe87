"""ORACLE (test infrastructure, NOT product code) — CPU restatement of DR4SR+'s MetaModel
(/root/reference model/metamodel.py:123-194) and its implicit-differentiation optimiser
(utils/utils.py:134-252: Hypergrad, MetaOptimizer).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned against tests/golden/metamodel_sasrec.npz and metamodel_cl4srec.npz (made by RUNNING the reference: tools/make_golden.py
run_meta_case) by tests/test_meta_oracle.py.  The hyper-gradient here is the reference's exact one (torch
double-backward through the op-by-op SASRec restatement in sasrec_oracle.py); `hypergrad_fd` is the
finite-difference formulation the HIP product path uses, kept next to it so their distance can be tested on CPU.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch

from . import sasrec_oracle as so

META_NAMES = ["0.weight", "0.bias", "2.weight", "2.bias"]      # nn.Sequential(Linear, ReLU, Linear), metamodel.py:52-57


def selection(query, meta: Dict[str, torch.Tensor], gumbel, tau: float, tau_min: float, relu_gate=None):
    """metamodel.py:169-172 — F.gumbel_softmax(meta_module(query), tau=clip(tau, tau_min), hard=False)[..., 0] with the
    Gumbel noise given explicitly (gumbel[..., 2] = -log(Exp(1)) as torch draws it).
    relu_gate (0/1, optional): use this FROZEN activation pattern instead of (pre > 0) — autograd's second derivative of
    ReLU is 0 everywhere, so a finite difference must not see units flipping between the two evaluation points."""
    pre = query @ meta["0.weight"].T + meta["0.bias"]
    h = torch.relu(pre) if relu_gate is None else pre * relu_gate
    logits = h @ meta["2.weight"].T + meta["2.bias"]
    t = max(float(tau), float(tau_min))
    return ((logits + gumbel) / t).softmax(-1)[..., 0]


def mask_weight(w, user_id, target):
    """metamodel.py:180-185 — pattern rows (user_id == 0) get weight 1, PAD targets weight 0."""
    m = user_id == 0
    if w.dim() == 2:
        m = m.unsqueeze(-1)
    w = w.masked_fill(m, 1.0)
    return w.masked_fill(target == 0, 0.0)


def weighted_loss(loss_pos, query, meta, gumbel, tau, tau_min, user_id, target, relu_gate=None):
    """MetaModel.training_step (metamodel.py:174-194): (loss[b,l] * weight[b,l]).sum()"""
    w = mask_weight(selection(query, meta, gumbel, tau, tau_min, relu_gate), user_id, target)
    return (loss_pos * w).sum(), w


# ------------------------------------------------------------------------------------------------ SASRec sub-model
def sasrec_losses(cfg) -> Callable:
    """returns f(p, batch, reduce) -> (loss, query) for the SASRec sub-model (dropout 0)"""
    def f(p, batch, reduce):
        loss, q, _, _ = so.training_step(p, batch, cfg["H"], cfg["n_layer"], cfg["eps"], reduce=reduce)
        return loss, q
    return f


def gru4rec_losses(n_layer: int) -> Callable:
    """f(p, batch, reduce) for a GRU4Rec sub-model (model/gru4rec.py:12-34, 'origin' pooling, dropout 0); the scorer is
    BaseModel.training_step's (basemodel.py:204-214), shared by every sub-model"""
    from . import gru4rec_oracle as go

    def f(p, batch, reduce):
        q = go.gru4rec_encode(p, batch["in_item_id"], batch["seqlen"], n_layer, "origin")
        loss, _, _ = so.score_bce(q, p["item_embedding.weight"], batch["item_id"], batch["neg_item"], reduce)
        return loss, q
    return f


def fmlp_losses(n_layer: int, eps: float = 1e-12) -> Callable:
    """f(p, batch, reduce) for an FMLP sub-model (model/fmlp.py:18-39: one query per prefix row, scalar target, dropout 0)"""
    from . import fmlp_oracle as fo

    def f(p, batch, reduce):
        loss, q, _, _ = fo.training_step(p, batch, n_layer, eps, reduce=reduce)
        return loss, q
    return f


def cl4srec_losses(cfg) -> Callable:
    """f(p, batch, reduce) for a CL4SRec sub-model (model/cl4srec.py:49-73, dropout 0) on RECORDED views batch['_views'] =
    ((seq_i, len_i), (seq_j, len_j)); reduce=False returns the tuple (bce per position, cl_weight * InfoNCE rows)"""
    from . import cl4srec_oracle as co

    def f(p, batch, reduce):
        return co.training_loss_q(p, batch, batch["_views"], cfg, reduce=reduce)
    return f


def train_loss(f, p, meta, bt, gumbel, tau, tau_min, relu_gate=None):
    lp, q = f(p, bt, False)
    extra = 0.0
    if isinstance(lp, tuple):                                      # metamodel.py:186-192: CL4SRec's contrastive rows stay un-weighted
        lp, extra = lp[0], lp[1].sum()
    return weighted_loss(lp, q, meta, gumbel, tau, tau_min, bt["user_id"], bt["item_id"], relu_gate)[0] + extra


def relu_gate_of(f, p, meta, bt):
    with torch.no_grad():
        _, q = f(p, bt, False)
        return ((q @ meta["0.weight"].T + meta["0.bias"]) > 0).to(q.dtype)


def hypergrad_exact(f, p: Dict[str, torch.Tensor], meta: Dict[str, torch.Tensor], bt, bv, gumbel, tau, tau_min,
                    hpo_lr: float, truncate_iter: int = 3, dtype=torch.float32):
    """Hypergrad.grad (utils/utils.py:145-178) + _approx_inverse_hvp (:180-205), literally:
         v = p = dL_val/dW ; repeat K: v <- v - lr * H v ; p <- p + v ; return -(d/dphi)(dL_train/dW . p)"""
    P = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    M = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in meta.items()}
    names = list(P)
    plist = [P[k] for k in names]
    lval, _ = f(P, bv, True)
    gval = torch.autograd.grad(lval, plist, allow_unused=True)
    gval = [g if g is not None else torch.zeros_like(x) for g, x in zip(gval, plist)]
    ltr = train_loss(f, P, M, bt, gumbel.to(dtype), tau, tau_min)
    gtr = torch.autograd.grad(ltr, plist, create_graph=True, allow_unused=True)
    v = pacc = [g.clone() for g in gval]
    for _ in range(truncate_iter):
        hv = torch.autograd.grad(gtr, plist, grad_outputs=v, retain_graph=True, allow_unused=True)
        v = [cv - hpo_lr * h for cv, h in zip(v, hv)]
        pacc = [cp + cv for cp, cv in zip(pacc, v)]
    v3 = torch.autograd.grad(gtr, [M[k] for k in META_NAMES], grad_outputs=pacc, allow_unused=True)
    return {k: -g for k, g in zip(META_NAMES, v3)}, dict(zip(names, gval)), dict(zip(names, pacc))


def hypergrad_fd(f, p, meta, bt, bv, gumbel, tau, tau_min, hpo_lr: float, truncate_iter: int = 3, rel_step: float = 1e-2,
                 dtype=torch.float32, richardson: bool = False, forward_hvp: bool = False):
    """The same quantity from FIRST-ORDER gradients only (what the HIP path does, dr4sr_amd/model/metamodel.py):
         H v          ~ [G(W + e v) - G(W - e v)] / 2e,        G = dL_train/dW      (Neumann terms, scaled by hpo_lr)
         d/dphi(G.p)  ~ [dL_train/dphi(W + e p) - dL_train/dphi(W - e p)] / 2e
       with e = rel_step * |W| / |direction| and the meta-module's ReLU pattern frozen at W.
       richardson=True (the product's default): the mixed term as (4 D(e) - D(2e)) / 3 over probes at +-e and +-2e.
       forward_hvp=True (the product's default since round 4): the Neumann terms from ONE-sided differences against G(W)."""
    P = {k: v.detach().to(dtype).clone() for k, v in p.items()}
    M = {k: v.detach().to(dtype).clone() for k, v in meta.items()}
    names = list(P)
    gate = relu_gate_of(f, P, M, bt)

    def first_order(Wd, want_phi):
        Pl = {k: v.clone().requires_grad_(True) for k, v in Wd.items()}
        Ml = {k: v.clone().requires_grad_(True) for k, v in M.items()}
        ltr = train_loss(f, Pl, Ml, bt, gumbel.to(dtype), tau, tau_min, gate)
        ltr.backward()
        g = {k: (Pl[k].grad if Pl[k].grad is not None else torch.zeros_like(Pl[k])) for k in names}
        return (g, {k: Ml[k].grad for k in META_NAMES}) if want_phi else g

    def norm(d):
        return float(torch.sqrt(sum((x.double() ** 2).sum() for x in d.values())))

    def shifted(d, e):
        return {k: P[k] + e * d[k] for k in names}

    Pl = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lval, _ = f(Pl, bv, True)
    lval.backward()
    gval = {k: (Pl[k].grad if Pl[k].grad is not None else torch.zeros_like(Pl[k])) for k in names}
    wn = norm(P)
    v = {k: g.clone() for k, g in gval.items()}
    pacc = {k: g.clone() for k, g in gval.items()}
    g0 = first_order(P, False) if forward_hvp else None
    for _ in range(truncate_iter):
        e = rel_step * wn / max(norm(v), 1e-30)
        if forward_hvp:                                    # H v ~ [G(W + e v) - G(W)] / e: one probe per Neumann term (they enter scaled by hpo_lr)
            gp = first_order(shifted(v, e), False)
            v = {k: v[k] - hpo_lr * (gp[k] - g0[k]) / e for k in names}
        else:
            gp, gm = first_order(shifted(v, e), False), first_order(shifted(v, -e), False)
            v = {k: v[k] - hpo_lr * (gp[k] - gm[k]) / (2 * e) for k in names}
        pacc = {k: pacc[k] + v[k] for k in names}
    e = rel_step * wn / max(norm(pacc), 1e-30)
    def central(h):
        _, fp = first_order(shifted(pacc, h), True)
        _, fm = first_order(shifted(pacc, -h), True)
        return {k: (fp[k] - fm[k]) / (2 * h) for k in META_NAMES}
    d1 = central(e)
    if richardson:
        d2 = central(2 * e)
        d1 = {k: (4 * d1[k] - d2[k]) / 3 for k in META_NAMES}
    return {k: -d1[k] for k in META_NAMES}, gval, pacc


# ------------------------------------------------------------------------------------------------ MetaOptimizer
def clip_grad_norm_(grads: List[torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (utils/utils.py:243-244): g *= min(1, max_norm / (|g|_2 + 1e-6))"""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def sgd_momentum_step(params, grads, bufs, lr, momentum=0.9, weight_decay=0.0):
    """torch.optim.SGD(momentum=0.9, weight_decay=wd) (metamodel.py:68-69): g += wd*p ; buf = g (first) | mu*buf + g ; p -= lr*buf"""
    out_p, out_b = [], []
    for p, g, b in zip(params, grads, bufs):
        g = g + weight_decay * p
        b = g.clone() if b is None else momentum * b + g
        out_p.append(p - lr * b)
        out_b.append(b)
    return out_p, out_b


def meta_optimizer_step(name: str, params, grads, state, lr: float, meta_weight_decay: float, max_norm: float = 10.0):
    """MetaOptimizer.step's tail (utils/utils.py:242-250) for the optimizer MetaModel._get_meta_optimizers builds from `meta_optimizer`
    (/root/reference/model/metamodel.py:59-81): clip_grad_norm_(max_norm) over the meta module's gradients, then
      'adam' -> Adam(lr) | 'sgd' -> SGD(lr, weight_decay, momentum 0.9) | 'adagrad' -> Adagrad(lr) | 'rmsprop' -> RMSprop(lr)
      | any other name -> Adam(lr, weight_decay).
    The optimizers' arithmetic is optim_oracle's (pinned on torch.optim); state = {} on the first call.  Returns the new parameters."""
    from . import optim_oracle as OO
    n = name.lower()
    grads, _ = clip_grad_norm_(grads, max_norm)
    if n == "sgd":
        new, state["bufs"] = sgd_momentum_step(params, grads, state.get("bufs", [None] * len(params)), lr, 0.9, meta_weight_decay)
        return new
    kind = n if n in ("adam", "adagrad", "rmsprop") else "adam"
    wd = 0.0 if n in ("adam", "adagrad", "rmsprop") else meta_weight_decay
    sts = state.setdefault("per_param", [OO.init_state(p) for p in params])
    return [OO.step(kind, p, g, st, lr, wd) for p, g, st in zip(params, grads, sts)]
