"""Loaders of the trained-checkpoint fixtures (tests/golden/sasrec_trained_toys.npz, metamodel_trained_toys.npz: made by
tools/make_golden.py run_trained_case / run_meta_case(real=True) by RUNNING the reference with its SHIPPED toys checkpoint on the REAL
toys rows).  Table-shaped arrays are stored as (row ids, rows) and ids as int32: this module puts them back."""
import os

import numpy as np
import torch

TABLE = "item_embedding.weight"


def _dense(g, key, base):
    out = np.array(base, copy=True)
    out[g[key + ".rows"]] = g[key + ".vals"]
    return out


def load_trained(golden_dir):
    """-> (g, params, batches): g = every array (table-shaped ones densified under their plain key), params = the checkpoint's tensors
    (the tied table once, under item_embedding.weight), batches = {'b0' | 'tail': batch dict with int64 ids}"""
    z = np.load(os.path.join(golden_dir, "sasrec_trained_toys.npz"))
    g = {k: z[k] for k in z.files}
    params = {k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    E = g["param." + TABLE]
    batches = {}
    for tag in ("b0", "tail"):
        batches[tag] = {k[len(tag) + 7:]: torch.from_numpy(v.astype(np.int64)) for k, v in g.items() if k.startswith(tag + ".batch.")}
        batches[tag]["neg_item"] = batches[tag]["neg_item"].reshape(batches[tag]["item_id"].shape + (1,))
        g[f"{tag}.grad.{TABLE}"] = _dense(g, f"{tag}.grad.{TABLE}", np.zeros_like(E))
        g[f"{tag}.adam2.{TABLE}"] = _dense(g, f"{tag}.adam2.{TABLE}", E)
    for k in ("in_item_id", "item_id", "seqlen", "user_hist", "topk_items"):
        g["eval." + k] = g["eval." + k].astype(np.int64)
    return g, params, batches


def load_meta_trained(golden_dir):
    """-> (g, meta_params, bt, bv); the sub-model's parameters are load_trained()'s"""
    z = np.load(os.path.join(golden_dir, "metamodel_trained_toys.npz"))
    g = {k: z[k] for k in z.files}
    pick = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}
    N, D = int(g["meta.num_items"]), 64
    for key in ("inner.grad." + TABLE, "outer.grad_val." + TABLE):
        g[key] = _dense(g, key, np.zeros((N, D), np.float32))
    return g, pick("meta_param."), pick("train."), pick("val.")


def curve_init(shapes, seed):
    """deterministic initial parameters from a numpy stream — the function tools/make_golden.py run_curve_case ran inside the reference
    (same code there), so that no parameter file travels: N(0, 0.02) for every matrix (PAD row zero), LayerNorm (1, 0), biases 0 =
    the distribution of the reference's utils/utils.py:70-81"""
    rng = np.random.default_rng(seed)
    out = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if "norm" in k and k.endswith("weight"):
            out[k] = np.ones(shp, np.float32)
        elif k.endswith("bias"):
            out[k] = np.zeros(shp, np.float32)
        else:
            out[k] = rng.normal(0.0, 0.02, size=shp).astype(np.float32)
    out[TABLE][0] = 0
    return out


def load_curve(golden_dir):
    """tests/golden/sasrec_toys_curve.npz -> (g, rows): the reference's per-epoch mean training losses on the REAL toys rows (two RNG
    seeds) and those rows as int64 tensors"""
    z = np.load(os.path.join(golden_dir, "sasrec_toys_curve.npz"))
    g = {k: z[k] for k in z.files}
    rows = {k[5:]: torch.from_numpy(g[k].astype(np.int64)) for k in ("rows.in_item_id", "rows.item_id", "rows.seqlen")}
    return g, rows
