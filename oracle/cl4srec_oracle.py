"""ORACLE (test infrastructure, NOT product code) — CPU restatement of CL4SRec's contrastive branch
(/root/reference model/cl4srec.py:49-73, module/data_augmentation.py:305-350, :577-619, module/functional.py:28-55).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against tests/golden/cl4srec_d64.npz (made by RUNNING the reference with its drawn views recorded: tools/make_golden.py
run_cl_case) by tests/test_cl_oracle.py.  Also holds reference-style augmentations in plain Python for distribution tests.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import sasrec_oracle as so


def mean_pool(x, seqlen):
    """seq_pooling_function(pooling_type='mean') (module/functional.py:28-55): rows >= seqlen zeroed, sum / seqlen"""
    L = x.shape[1]
    keep = (torch.arange(L).view(1, L) < seqlen.view(-1, 1)).unsqueeze(-1)
    return torch.where(keep, x, torch.zeros((), dtype=x.dtype)).sum(1) / seqlen.view(-1, 1).to(x.dtype)


def infonce(rep_i, rep_j, temperature=1.0, reduce=True):
    """InfoNCELoss(sim_method='inner_product', neg_type='batch_both') (data_augmentation.py:322-350)"""
    B = rep_i.shape[0]
    sim_ii = rep_i @ rep_i.T / temperature
    sim_ij = rep_i @ rep_j.T / temperature
    sim_ii = sim_ii.masked_fill(torch.eye(B, dtype=torch.bool), float("-inf"))
    logits = torch.cat([sim_ij, sim_ii], dim=-1)
    labels = torch.arange(B)
    if reduce:
        return F.cross_entropy(logits, labels)
    return F.cross_entropy(logits, labels, reduction="none") / B


def view_mean(p, view, view_len, H, n_layer, eps):
    x = so.sasrec_encode(p, view, view_len, H, n_layer, eps, None)
    return mean_pool(x, view_len)


def training_loss(p, batch, views, cfg, reduce=True):
    """CL4SRec.training_step (cl4srec.py:49-73) on given views = ((seq_i, len_i), (seq_j, len_j))"""
    H, nl, eps = cfg["H"], cfg["n_layer"], cfg["eps"]
    bce, q, _, _ = so.training_step(p, batch, H, nl, eps, reduce=reduce)
    (vi, li), (vj, lj) = views
    oi, oj = view_mean(p, vi, li, H, nl, eps), view_mean(p, vj, lj, H, nl, eps)
    keep = batch["seqlen"] != 1                                       # data_augmentation.py:613-615
    cl = infonce(oi[keep], oj[keep], cfg["temperature"], reduce)
    return bce + cfg["cl_weight"] * cl, bce, cl, oi, oj


def training_loss_q(p, batch, views, cfg, reduce=True):
    """CL4SRec.training_step(return_query=True) (cl4srec.py:49-73): reduce=True -> (bce + cl_weight * InfoNCE, query); reduce=False ->
    ((bce per position / n_valid, cl_weight * InfoNCE rows / kept rows), query) — the tuple MetaModel.training_step takes apart
    (metamodel.py:186-192)"""
    H, nl, eps = cfg["H"], cfg["n_layer"], cfg["eps"]
    bce, q, _, _ = so.training_step(p, batch, H, nl, eps, reduce=reduce)
    (vi, li), (vj, lj) = views
    oi, oj = view_mean(p, vi, li, H, nl, eps), view_mean(p, vj, lj, H, nl, eps)
    keep = batch["seqlen"] != 1
    cl = cfg["cl_weight"] * infonce(oi[keep], oj[keep], cfg["temperature"], reduce)
    return (bce + cl if reduce else (bce, cl)), q


# ---- reference-style augmentations (plain Python), for distribution tests of dr4sr_cl_augment
def crop_len(n, tau):
    return max(1, int(tau * n))


def mask_count(n, gamma):
    return int(gamma * n)


def reorder_len(n, beta):
    return int(beta * n)
