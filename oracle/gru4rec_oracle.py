"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's GRU4Rec target model.

model/gru4rec.py:12-34: Linear_{H->D}( GRU_{n_layer, no bias, batch_first}( dropout(E[idx]) ) ), then 'origin' / 'last' pooling
(module/layers.py:41-50, :69-73); GRULayer = torch.nn.GRU(bias=False) (module/layers.py:117-136).  torch's GRU cell, gates
ordered r|z|n in the 3H rows, h_0 = 0, all L steps computed (no packing):
    gi = x_t W_ih^T ; gh = h_{t-1} W_hh^T
    r = sigmoid(gi_r + gh_r) ; z = sigmoid(gi_z + gh_z) ; n = tanh(gi_n + r * gh_n) ; h_t = (1 - z) * n + z * h_{t-1}
Pinned against golden vectors produced by running the reference (tests/golden/gru4rec_d64.npz).
The optimizer is Adam with L2 weight_decay 1e-4 (configs/gru4rec.yaml:6-8) — oracle.sasrec_oracle.adam_step(wd=...).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SITE_EMB = 0           # gru4rec.py:18  Dropout(dropout_rate) on the item embeddings


def gru_layer(x, w_ih, w_hh):
    B, L, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.T
    h = torch.zeros(B, H, dtype=x.dtype)
    out = []
    for t in range(L):
        gh = h @ w_hh.T
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=1)


def gru4rec_encode(p, idx, seqlen, n_layer, pooling, mask=None, pdrop=0.0):
    x = F.embedding(idx, p["item_embedding.weight"], padding_idx=0)
    if mask is not None and pdrop > 0:
        x = x * mask.to(x.dtype) / (1.0 - pdrop)
    for l in range(n_layer):
        x = gru_layer(x, p[f"query_encoder.0.3.gru.weight_ih_l{l}"], p[f"query_encoder.0.3.gru.weight_hh_l{l}"])
    y = x @ p["query_encoder.1.weight"].T + p["query_encoder.1.bias"]
    B, L, D = y.shape
    if pooling == "origin":
        keep = (torch.arange(L).view(1, L) < seqlen.view(B, 1)).unsqueeze(-1)
        return torch.where(keep, y, torch.zeros((), dtype=y.dtype))
    if pooling == "last":
        return y[torch.arange(B), seqlen - 1]
    return y


def grads_of(p, batch, n_layer, mask=None, pdrop=0.0):
    from .sasrec_oracle import score_bce
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items() if k != "query_encoder.0.1.weight"}
    q = gru4rec_encode(leaf, batch["in_item_id"], batch["seqlen"], n_layer, "origin", mask, pdrop)
    loss, _, _ = score_bce(q, leaf["item_embedding.weight"], batch["item_id"], batch["neg_item"], True)
    loss.backward()
    return loss.detach(), q.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
