"""FmlpEngine — flat-buffer training engine for the reference's FMLP (model/fmlp.py) on top of dr4sr_fmlp_* (C ABI)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib

_LAYER = [("filterlayer.complex_weight", lambda L, D, F: (1, L // 2 + 1, D, 2)),
          ("filterlayer.LayerNorm.weight", lambda L, D, F: (D,)), ("filterlayer.LayerNorm.bias", lambda L, D, F: (D,)),
          ("intermediate.dense_1.weight", lambda L, D, F: (F, D)), ("intermediate.dense_1.bias", lambda L, D, F: (F,)),
          ("intermediate.dense_2.weight", lambda L, D, F: (D, F)), ("intermediate.dense_2.bias", lambda L, D, F: (D,)),
          ("intermediate.LayerNorm.weight", lambda L, D, F: (D,)), ("intermediate.LayerNorm.bias", lambda L, D, F: (D,))]


def fmlp_param_names(n_layer):
    names = ["item_embedding.weight", "position_embeddings.weight", "LayerNorm.weight", "LayerNorm.bias"]
    for i in range(n_layer):
        names += [f"item_encoder.layer.{i}.{n}" for n, _ in _LAYER]
    return names


def fmlp_param_shapes(n_items, L, D, F, n_layer):
    shapes = [(n_items, D), (L, D), (D,), (D,)]
    for _ in range(n_layer):
        shapes += [fn(L, D, F) for _, fn in _LAYER]
    return shapes


class FmlpEngine:
    def __init__(self, n_items, L=50, D=64, F=256, n_layer=2, ln_eps=1e-12, p_drop=0.5, max_batch=256, device="cuda",
                 seed=2023, lr=1e-3, betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=0.0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.Dr4srError("FmlpEngine needs a GPU device; dr4sr_amd has no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_items, self.L, self.D, self.F, self.n_layer = n_items, L, D, F, n_layer
        self.ln_eps, self.p_drop, self.seed = float(ln_eps), float(p_drop), int(seed)
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, adam_eps, weight_decay
        self.optimizer = _lib.OPT_ADAM             # DR4SR_OPT_* (set_optimizer)
        self.max_batch = max_batch
        off = (C.c_int64 * (4 + 9 * n_layer))()
        self.n_params = int(self.lib.dr4sr_fmlp_param_layout(n_items, L, D, F, n_layer, off))
        self.offsets = list(off)
        dev = self.device
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_params + _lib.GRAD_TAIL, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=dev)
        self.names = fmlp_param_names(n_layer)
        self.shapes = fmlp_param_shapes(n_items, L, D, F, n_layer)
        self.views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.grad_views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for name, shp, o in zip(self.names, self.shapes, self.offsets):
            n = 1
            for s in shp:
                n *= s
            self.views[name] = self.params[o:o + n].view(shp)
            self.grad_views[name] = self.grads[o:o + n].view(shp)
        probe = self._plan(max_batch, None, None, None, None, False, with_ws=False)
        self.ws_bytes = int(self.lib.dr4sr_fmlp_workspace_bytes(C.byref(probe)))
        if self.ws_bytes <= 0:
            raise _lib.Dr4srError(f"FMLP shape L={self.L} D={self.D} F={self.F} layers={self.n_layer}: dr4sr_fmlp_workspace_bytes failed: "
                                  + _lib._ERR.get(self.ws_bytes, str(self.ws_bytes)) + " — built for even L <= 50, D = 64, F = 256")
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.neg_scratch = torch.zeros(max_batch, dtype=torch.int64, device=dev)

    def _plan(self, B, in_item_id, item_id, rows, neg_item, sample_neg, with_ws=True, perm_sel=None, loss_log=None):
        p = _lib.FmlpPlan()
        p.abi_version = _lib.ABI_VERSION
        p.B, p.L, p.D, p.F, p.n_layer, p.n_items = B, self.L, self.D, self.F, self.n_layer, self.n_items
        p.ln_eps, p.p_drop, p.seed = self.ln_eps, self.p_drop, self.seed
        p.params, p.grads = self.params.data_ptr(), self.grads.data_ptr()
        p.adam_m, p.adam_v, p.n_params = self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.n_params
        for name, t in (("in_item_id", in_item_id), ("item_id", item_id), ("rows", rows), ("neg_item", neg_item)):
            if t is not None:
                assert t.dtype == torch.int64 and t.is_contiguous() and t.device == self.device, name
                setattr(p, name, t.data_ptr())
        p.sample_neg = 1 if sample_neg else 0
        if with_ws:
            p.workspace, p.workspace_bytes = self.workspace.data_ptr(), self.ws_bytes
        p.state = self.state.data_ptr()
        p.lr, (p.beta1, p.beta2), p.adam_eps, p.weight_decay = self.lr, self.betas, self.adam_eps, self.weight_decay
        p.optimizer = self.optimizer
        if perm_sel is not None:           # (perm[n], stride, offset, counter[1] int32): rows[] is FILLED by the step's first kernel
            perm, stride, offset, counter = perm_sel
            assert rows is not None and perm.dtype == torch.int64 and counter.dtype == torch.int32
            p.perm, p.n_perm, p.perm_stride, p.perm_offset = perm.data_ptr(), int(perm.shape[0]), int(stride), int(offset)
            p.perm_counter = counter.data_ptr()
        if loss_log is not None:           # float32 device buffer: the step's mean loss lands at [batch index] (include/dr4sr_hip.h)
            assert loss_log.dtype == torch.float32
            p.loss_log = loss_log.data_ptr()
        self._keep = [in_item_id, item_id, rows, neg_item, perm_sel, loss_log]
        return p

    def make_plan(self, in_item_id, item_id, rows=None, neg_item=None, sample_neg=None, perm_sel=None, loss_log=None):
        B = int(rows.shape[0] if rows is not None else in_item_id.shape[0])
        if B > self.max_batch:
            raise _lib.Dr4srError(f"batch {B} > max_batch {self.max_batch}")
        if in_item_id.dim() != 2 or int(in_item_id.shape[1]) != self.L:
            raise _lib.Dr4srError(f"FMLP: in_item_id must be [rows, {self.L}], got {tuple(in_item_id.shape)}")
        if item_id is not None and item_id.dim() != 1:
            # model/fmlp.py:38 keeps ONE query per row (transformer_out[:, -1]); against [B, L] targets the reference's
            # (query * item_embedding(target)).sum(-1) (basemodel.py:182) cannot broadcast
            raise _lib.Dr4srError(f"FMLP scores one query per row: item_id must be [rows], got {tuple(item_id.shape)} "
                                  "(use the prefix-row data format: data.prefix_rows / configs/synthetic-toys-prefix.yaml)")
        if sample_neg is None:
            sample_neg = neg_item is None
        if neg_item is None:
            neg_item = self.neg_scratch
        return self._plan(B, in_item_id, item_id, rows, neg_item, sample_neg, perm_sel=perm_sel, loss_log=loss_log)

    def fwd_bwd(self, plan):
        _lib.check(self.lib.dr4sr_fmlp_fwd_bwd(C.byref(plan), _lib.cur_stream()), "dr4sr_fmlp_fwd_bwd")

    def train_step(self, plan):
        _lib.check(self.lib.dr4sr_fmlp_train_step(C.byref(plan), _lib.cur_stream()), "dr4sr_fmlp_train_step")

    def adam_step(self, plan=None):
        if plan is not None:               # the plan's optimizer launch: logs the step's loss when the plan carries a loss log
            _lib.check(self.lib.dr4sr_fmlp_adam_step(C.byref(plan), _lib.cur_stream()), "dr4sr_fmlp_adam_step")
            return
        _lib.check(self.lib.dr4sr_optimizer_flat(self.optimizer, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                                 _lib.ptr(self.adam_v), self.n_params, _lib.ptr(self.state), self.lr, self.betas[0],
                                                 self.betas[1], self.adam_eps, self.weight_decay, _lib.cur_stream()), "dr4sr_optimizer_flat")

    def encode(self, plan, training: bool, out: Optional[torch.Tensor] = None):
        if out is None:
            out = torch.empty(plan.B, self.D, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_fmlp_encode(C.byref(plan), int(training), _lib.ptr(out), _lib.cur_stream()), "dr4sr_fmlp_encode")
        return out

    def encode_bwd(self, plan, training: bool, d_out: torch.Tensor):
        _lib.check(self.lib.dr4sr_fmlp_encode_bwd(C.byref(plan), int(training), _lib.ptr(d_out.contiguous()), _lib.cur_stream()),
                   "dr4sr_fmlp_encode_bwd")

    def loss_and_count(self):
        tail = self.grads[self.n_params:self.n_params + 2].tolist()
        return (tail[1] / tail[0] if tail[0] > 0 else float("nan")), int(tail[0])

    def normalized_grads(self) -> Dict[str, torch.Tensor]:
        n = self.grads[self.n_params]
        return {k: v / n for k, v in self.grad_views.items()}

    def load_named(self, sd):
        for k, v in self.views.items():
            v.copy_(sd[k].to(self.device, torch.float32))

    def dropout_mask(self, n, site, step, p=None):
        n4 = (n + 3) // 4 * 4
        out = torch.empty(n4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_dropout_mask(_lib.ptr(out), n4, self.p_drop if p is None else p, self.seed, step, site,
                                               _lib.cur_stream()), "dr4sr_dropout_mask")
        return out[:n]
