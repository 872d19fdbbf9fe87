"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the optimizers the reference can be configured with
(/root/reference model/basemodel.py:79-98): torch.optim.Adam / SGD / Adagrad / RMSprop, each with torch's defaults as the reference
constructs them (lr and weight_decay only).  The arithmetic is torch's single-tensor formulas (a third-party dependency of the reference,
un-pinned in its requirements; the container has torch 2.10): pinned by tests/test_oracle_golden.py against torch.optim ITSELF run here.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch


def init_state(p):
    return {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}


def step(kind: str, p, g, st, lr: float, weight_decay: float = 0.0):
    """one optimizer step on a flat parameter tensor p with gradient g; returns the new p (st is updated in place)"""
    st["step"] += 1
    t = st["step"]
    g = g + weight_decay * p
    if kind == "adam":                                   # betas (0.9, 0.999), eps 1e-8
        b1, b2, eps = 0.9, 0.999, 1e-8
        st["m"] = st["m"] + (g - st["m"]) * (1 - b1)
        st["v"] = st["v"] * b2 + (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        return p - (lr / bc1) * st["m"] / (st["v"].sqrt() / (bc2 ** 0.5) + eps)
    if kind == "sgd":                                    # momentum 0, dampening 0, no nesterov
        return p - lr * g
    if kind == "adagrad":                                # lr_decay 0, initial_accumulator_value 0, eps 1e-10
        st["v"] = st["v"] + g * g
        return p - lr * g / (st["v"].sqrt() + 1e-10)
    if kind == "rmsprop":                                # alpha 0.99, eps 1e-8, momentum 0, not centered
        st["v"] = st["v"] * 0.99 + (1 - 0.99) * g * g
        return p - lr * g / (st["v"].sqrt() + 1e-8)
    raise KeyError(kind)
