"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's FMLP target model.

Follows model/fmlp.py:18-39 (embedding + position, LayerNorm, dropout, encoder, x[:, -1]) and module/layers.py:740-807
(FilterLayer: rfft/irfft 'ortho' along the sequence with a learnable complex weight [1, L//2+1, D]; Intermediate:
dense 64->256, GELU, dense 256->64, dropout, LayerNorm(+residual); no dropout after the activation).  The spectral filter
is restated as the per-feature circular convolution it equals,
    m[r, d] = (1/L) sum_k c_k (Wre[k,d] cos(2 pi k r / L) - Wim[k,d] sin(2 pi k r / L)),  c_0 = c_{L/2} = 1, else 2
    y[b, l, d] = sum_r m[r, d] x[b, (l - r) mod L, d]
which is what the HIP kernels implement; test_fmlp_oracle pins it (and torch.fft itself) against golden vectors from the
reference.  Dropout is hard-coded 0.5 in the reference (configs/fmlp.yaml's dropout_rate is ignored); masks are explicit here.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SITE_EMB = 0           # fmlp.py:27    dropout(LayerNorm(emb + pos))
SITE_FILT = 1          # layers.py:756 out_dropout after the filter        (+2*layer)
SITE_FFN = 2           # layers.py:776 dropout after dense_2               (+2*layer)


def site(kind, layer=0):
    return kind if kind == SITE_EMB else kind + 2 * layer


def _drop(x, masks, s, p):
    if masks is None or p == 0.0 or s not in masks:
        return x
    return x * masks[s].to(x.dtype) / (1.0 - p)


def filter_kernel(cw, L):
    """complex_weight [1, L//2+1, D, 2] -> real circular-convolution kernel m [L, D]"""
    wre, wim = cw[0, :, :, 0], cw[0, :, :, 1]
    K = wre.shape[0]
    k = torch.arange(K, dtype=cw.dtype).view(K, 1)
    r = torch.arange(L, dtype=cw.dtype).view(1, L)
    ang = 2 * math.pi * k * r / L
    c = torch.full((K, 1), 2.0, dtype=cw.dtype)
    c[0] = 1.0
    if L % 2 == 0:
        c[L // 2] = 1.0
    return ((c * torch.cos(ang)).T @ wre - (c * torch.sin(ang)).T @ wim) / L          # [L, D]


def circ_conv(x, m):
    """y[b,l,d] = sum_r m[r,d] x[b,(l-r) mod L,d]"""
    B, L, D = x.shape
    idx = (torch.arange(L).view(L, 1) - torch.arange(L).view(1, L)) % L                # [l, r] -> (l-r) mod L
    return torch.einsum("blrd,rd->bld", x[:, idx, :], m)


def fmlp_encode(p, idx, n_layer, eps=1e-12, masks=None, pdrop=0.0, use_fft=False, return_all=False):
    E, P = p["item_embedding.weight"], p["position_embeddings.weight"]
    L, D = idx.shape[1], E.shape[1]
    x = F.embedding(idx, E, padding_idx=0) + P[:L].unsqueeze(0)
    x = F.layer_norm(x, (D,), p["LayerNorm.weight"], p["LayerNorm.bias"], eps)
    x = _drop(x, masks, SITE_EMB, pdrop)
    acts = {"x0": x}
    for i in range(n_layer):
        pre = f"item_encoder.layer.{i}."
        cw = p[pre + "filterlayer.complex_weight"]
        if use_fft:
            f = torch.fft.irfft(torch.fft.rfft(x, dim=1, norm="ortho") * torch.view_as_complex(cw.contiguous()), n=L, dim=1, norm="ortho")
        else:
            f = circ_conv(x, filter_kernel(cw, L))
        f = _drop(f, masks, site(SITE_FILT, i), pdrop)
        x = F.layer_norm(f + x, (D,), p[pre + "filterlayer.LayerNorm.weight"], p[pre + "filterlayer.LayerNorm.bias"], eps)
        acts[f"filter{i}"] = x
        h = F.gelu(x @ p[pre + "intermediate.dense_1.weight"].T + p[pre + "intermediate.dense_1.bias"])
        o = h @ p[pre + "intermediate.dense_2.weight"].T + p[pre + "intermediate.dense_2.bias"]
        o = _drop(o, masks, site(SITE_FFN, i), pdrop)
        x = F.layer_norm(o + x, (D,), p[pre + "intermediate.LayerNorm.weight"], p[pre + "intermediate.LayerNorm.bias"], eps)
        acts[f"layer{i}"] = x
    q = x[:, -1]
    return (q, acts) if return_all else q


def training_step(p, batch, n_layer, eps=1e-12, masks=None, pdrop=0.0, reduce=True, use_fft=False):
    from .sasrec_oracle import score_bce
    q = fmlp_encode(p, batch["in_item_id"], n_layer, eps, masks, pdrop, use_fft)
    loss, pos, ng = score_bce(q, p["item_embedding.weight"], batch["item_id"], batch["neg_item"], reduce)
    return loss, q, pos, ng


def grads_of(p, batch, n_layer, eps=1e-12, masks=None, pdrop=0.0, dtype=torch.float32):
    leaf = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    loss, q, _, _ = training_step(leaf, batch, n_layer, eps, masks, pdrop)
    loss.backward()
    return loss.detach(), q.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
