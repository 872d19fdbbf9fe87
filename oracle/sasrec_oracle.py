"""ORACLE (test infrastructure, NOT product code) — CPU restatement of DR4SR's SASRec training path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (dr4sr_amd) never routes through it and has no CPU fallback.

Everything here is plain torch on CPU (fp32 by default, fp64 on request), written from the math
spec of the reference, each function citing the reference file:line it follows.  The heavy
arithmetic of the reference lives in third-party torch modules (torch.nn.TransformerEncoder,
torch.optim.Adam, F.logsigmoid/softplus; torch is un-pinned in the reference's requirements.txt:1,
README.md:18 names 1.13.1+cu117; this container has 2.10.0) — so the restatement below spells those
modules' published formulas out op by op, and is PINNED against golden vectors produced by running
the reference itself in the build container (tools/make_golden.py -> tests/golden/*.npz,
checked by tests/test_oracle_golden.py).

Dropout: the reference draws 9 independent torch-RNG masks per step; they cannot be bit-matched.
Every function therefore takes the masks explicitly (`masks[site]`, float 0/1 *keep* masks);
`None` means p = 0.  The HIP library can materialise the exact masks it uses
(dr4sr_dropout_mask), which makes dropout-on parity testable.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

# dropout site ids shared with the HIP library (include/dr4sr_hip.h: DR4SR_SITE_*)
SITE_EMB = 0          # sasrec.py:66      self.dropout(seq_embs + position_embs)
SITE_ATTN = 1         # torch MHA        dropout on attention probabilities      (+4*layer)
SITE_PROJ = 2         # torch TEL        dropout1 after out_proj                 (+4*layer)
SITE_ACT = 3          # torch TEL        dropout after activation                (+4*layer)
SITE_FFN = 4          # torch TEL        dropout2 after linear2                  (+4*layer)


def site(kind: int, layer: int = 0) -> int:
    return kind if kind == SITE_EMB else kind + 4 * layer


def _drop(x, masks, s, p):
    if masks is None or p == 0.0 or s not in masks:
        return x
    return x * masks[s].to(x.dtype) / (1.0 - p)


def layer_prefix(i: int) -> str:
    return f"query_encoder.transformer_layer.layers.{i}."


# ------------------------------------------------------------------------------------------------
def embed_posadd(E, P, idx):
    """sasrec.py:43-46 + :64 — item_encoder(user_hist) + position_emb(arange(L)); exact IEEE add.
    item_encoder is nn.Embedding(padding_idx=0) (basemodel.py:42): lookups of id 0 pass no gradient."""
    L = idx.shape[1]
    return F.embedding(idx, E, padding_idx=0) + P[:L].unsqueeze(0)


def sasrec_layer(x, key_pad, p: Dict[str, torch.Tensor], pre: str, H: int, eps: float,
                 masks=None, layer=0, pdrop=0.0):
    """One post-norm torch.nn.TransformerEncoderLayer as configured at sasrec.py:21-30
    (batch_first, norm_first=False, gelu(erf), eps=layer_norm_eps) called with
    mask = triu(ones(L,L),1) (sasrec.py:58) and src_key_padding_mask = (idx==0) (sasrec.py:48)."""
    B, L, D = x.shape
    dh = D // H
    qkv = x @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, L, H, dh).transpose(1, 2)
    k = k.view(B, L, H, dh).transpose(1, 2)
    v = v.view(B, L, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool), 1)
    bias = causal.view(1, 1, L, L) | key_pad.view(B, 1, 1, L)
    s = s.masked_fill(bias, float("-inf"))
    a = torch.softmax(s, dim=-1)
    a = _drop(a, masks, site(SITE_ATTN, layer), pdrop)
    ctx = (a @ v).transpose(1, 2).reshape(B, L, D)
    o = ctx @ p[pre + "self_attn.out_proj.weight"].T + p[pre + "self_attn.out_proj.bias"]
    o = _drop(o, masks, site(SITE_PROJ, layer), pdrop)
    y = F.layer_norm(x + o, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    h = F.gelu(y @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"])
    h = _drop(h, masks, site(SITE_ACT, layer), pdrop)
    f = h @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
    f = _drop(f, masks, site(SITE_FFN, layer), pdrop)
    return F.layer_norm(y + f, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)


def sasrec_encode(p: Dict[str, torch.Tensor], idx, seqlen, H: int, n_layer: int, eps: float,
                  pooling: str, masks=None, pdrop=0.0, return_all=False):
    """SASRecQueryEncoder.forward (sasrec.py:39-75) followed by SeqPoolingLayer
    ('origin' layers.py:41-50 zeroes rows >= seqlen; 'last' layers.py:69-73 picks row seqlen-1)."""
    E = p["item_embedding.weight"]
    P = p["query_encoder.position_emb.weight"]
    x = embed_posadd(E, P, idx)
    x = _drop(x, masks, SITE_EMB, pdrop)
    acts = {"x0": x}
    key_pad = idx == 0
    for i in range(n_layer):
        x = sasrec_layer(x, key_pad, p, layer_prefix(i), H, eps, masks, i, pdrop)
        acts[f"layer{i}"] = x
    B, L, D = x.shape
    if pooling == "origin":
        keep = (torch.arange(L).view(1, L) < seqlen.view(B, 1)).unsqueeze(-1)
        q = torch.where(keep, x, torch.zeros((), dtype=x.dtype))
    elif pooling == "last":
        q = x[torch.arange(B), seqlen - 1]
    elif pooling is None:
        q = x
    else:
        raise ValueError(pooling)
    return (q, acts) if return_all else q


def score_bce(query, E, target, neg, reduce=True):
    """BaseModel.training_step scorer (basemodel.py:204-214) + BinaryCrossEntropyLoss
    (loss_func.py:9-38), masked branch (pos.dim()==neg.dim()-1), K negatives with weight 1/K.
    Returns (loss, pos_score, neg_score)."""
    pos = (query * E[target]).sum(-1)
    ng = (query.unsqueeze(-2) * E[neg]).sum(-1)
    pad = target == 0
    pos = pos.masked_fill(pad, float("-inf"))
    n = (~pad).sum()
    pos_l = F.logsigmoid(pos).masked_fill(pad, 0.0)
    neg_l = (F.softplus(ng) / ng.shape[-1]).sum(-1).masked_fill(pad, 0.0)
    if reduce:
        loss = -pos_l.sum() / n + neg_l.sum() / n
    else:
        loss = -pos_l / n + neg_l / n
    return loss, pos, ng


def bce_from_scores(pos, neg, reduce=True):
    """BinaryCrossEntropyLoss.forward (loss_func.py:9-38) on score tensors: pos [...] with -inf at padded positions, neg [..., K]
    weighted 1/K (_cal_weight :40-41) — the masked branch — or neg of pos's rank ([B, K]): the plain-mean branch (:32-33)."""
    pad = torch.isinf(pos)
    n = (~pad).sum()
    pos_l = F.logsigmoid(pos).masked_fill(pad, 0.0)
    if pos.dim() == neg.dim():                              # loss_func.py:32-33: plain mean of the negatives' term, no padding mask
        neg_m = (F.softplus(neg) / neg.shape[-1]).sum(-1).mean()
        return (-pos_l.sum() / n if reduce else -pos_l / n) + neg_m
    neg_l = (F.softplus(neg) / neg.shape[-1]).sum(-1).masked_fill(pad, 0.0)
    if reduce:
        return -pos_l.sum() / n + neg_l.sum() / n
    return -pos_l / n + neg_l / n


def bpr_from_scores(pos, neg):
    """BPRLoss.forward (loss_func.py:44-49): -sum_k logsigmoid(pos - neg_k) * softmax(ones)_k summed over unpadded positions / their
    count.  No `reduce` parameter exists (loss_func.py:44)."""
    pad = torch.isinf(pos)
    d = torch.where(pad.unsqueeze(-1), torch.zeros_like(neg), pos.masked_fill(pad, 0.0).unsqueeze(-1) - neg)
    l = F.logsigmoid(d).masked_fill(pad.unsqueeze(-1), 0.0)
    return -(l / neg.shape[-1]).sum(-1).sum() / (~pad).sum()


def score_bpr(query, E, target, neg):
    """BaseModel.training_step scorer (basemodel.py:204-208) + BPRLoss (loss_func.py:44-49).  Returns (loss, pos, neg scores)."""
    pos = (query * E[target]).sum(-1)
    ng = (query.unsqueeze(-2) * E[neg]).sum(-1)
    pos = pos.masked_fill(target == 0, float("-inf"))
    return bpr_from_scores(pos, ng), pos, ng


def adam_step(params, grads, m, v, t: int, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam single-tensor formula (basemodel.py:86, :199; L2 weight_decay folded into g)."""
    out = {}
    for k in params:
        g = grads[k]
        if wd != 0.0:
            g = g + wd * params[k]
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = v[k].sqrt() / math.sqrt(bc2) + eps
        out[k] = params[k] - (lr / bc1) * m[k] / denom
    return out


def full_score_topk(q_last, E, hist, k, domain_items=None):
    """BaseModel.topk (basemodel.py:354-365): q @ E^T, -inf on non-domain items (PAD col 0 is never
    in the domain list), -inf scattered over the user's history, torch.topk."""
    N = E.shape[0]
    score = q_last @ E.T
    mask = torch.ones(1, N, dtype=torch.bool)
    if domain_items is None:
        domain_items = torch.arange(1, N)
    mask[:, domain_items] = False
    score = score.masked_fill(mask, float("-inf"))
    score = torch.scatter(score, 1, hist, float("-inf"))
    return torch.topk(score, k)


def recall_at(hit, k):
    """evaluation/__init__.py:9-33 with one positive per row (target label == 1)."""
    return hit[:, :k].sum(-1).to(torch.float32)


def ndcg_at(hit, k):
    """evaluation/__init__.py:107-134 with one positive per row: ideal DCG = 1/log2(2) = 1."""
    k = min(k, hit.shape[1])
    denom = torch.log2(torch.arange(k, dtype=torch.float32) + 2.0).view(1, -1)
    return (hit[:, :k].to(torch.float32) / denom).sum(-1)


def neg_sample_reference_like(B, L, N, generator=None, two_d=True):
    """BaseModel._neg_sampling (basemodel.py:50-61): multinomial over a [B,N] ones matrix with
    column 0 zeroed, with replacement; L draws per row for 2-D targets, else 1."""
    w = torch.ones(B, N)
    w[:, 0] = 0
    neg = torch.multinomial(w, L if two_d else 1, replacement=True, generator=generator)
    return neg.unsqueeze(-1)


# ------------------------------------------------------------------------------------------------
def training_step(p, batch, H, n_layer, eps, masks=None, pdrop=0.0, reduce=True, loss_fn="bce"):
    """fwd of one reference training step on parameter dict `p` (leaf tensors may require grad)."""
    q = sasrec_encode(p, batch["in_item_id"], batch["seqlen"], H, n_layer, eps, "origin", masks, pdrop)
    if loss_fn == "bpr":
        loss, pos, ng = score_bpr(q, p["item_embedding.weight"], batch["item_id"], batch["neg_item"])
    else:
        loss, pos, ng = score_bce(q, p["item_embedding.weight"], batch["item_id"], batch["neg_item"], reduce)
    return loss, q, pos, ng


def grads_of(p, batch, H, n_layer, eps, masks=None, pdrop=0.0, dtype=torch.float32, loss_fn="bce"):
    """loss + d loss / d param for every parameter, via autograd on the restatement."""
    leaf = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in p.items()
            if k != "query_encoder.item_encoder.weight"}
    loss, q, pos, ng = training_step(leaf, batch, H, n_layer, eps, masks, pdrop, loss_fn=loss_fn)
    loss.backward()
    g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), q.detach(), g
