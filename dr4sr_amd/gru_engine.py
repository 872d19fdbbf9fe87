"""GruEngine — flat-buffer training engine for the reference's GRU4Rec (model/gru4rec.py) on top of dr4sr_gru4rec_* (C ABI)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib


def gru_param_names(n_layer):
    names = ["item_embedding.weight"]
    for l in range(n_layer):
        names += [f"query_encoder.0.3.gru.weight_ih_l{l}", f"query_encoder.0.3.gru.weight_hh_l{l}"]
    return names + ["query_encoder.1.weight", "query_encoder.1.bias"]


def gru_param_shapes(n_items, D, H, n_layer):
    shapes = [(n_items, D)]
    for l in range(n_layer):
        shapes += [(3 * H, D if l == 0 else H), (3 * H, H)]
    return shapes + [(D, H), (D,)]


class GruEngine:
    def __init__(self, n_items, L=50, D=64, H=256, n_layer=2, p_drop=0.2, max_batch=256, device="cuda", seed=2023, lr=1e-3,
                 betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=1e-4):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.Dr4srError("GruEngine needs a GPU device; dr4sr_amd has no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_items, self.L, self.D, self.H, self.n_layer = n_items, L, D, H, n_layer
        self.p_drop, self.seed = float(p_drop), int(seed)
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, adam_eps, weight_decay
        self.optimizer = _lib.OPT_ADAM             # DR4SR_OPT_* (set_optimizer)
        self.max_batch = max_batch
        off = (C.c_int64 * (3 + 2 * n_layer))()
        self.n_params = int(self.lib.dr4sr_gru4rec_param_layout(n_items, D, H, n_layer, off))
        self.offsets = list(off)
        dev = self.device
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_params + _lib.GRAD_TAIL, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=dev)
        self.names = gru_param_names(n_layer)
        self.shapes = gru_param_shapes(n_items, D, H, n_layer)
        self.views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.grad_views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for name, shp, o in zip(self.names, self.shapes, self.offsets):
            n = 1
            for s in shp:
                n *= s
            self.views[name] = self.params[o:o + n].view(shp)
            self.grad_views[name] = self.grads[o:o + n].view(shp)
        probe = self._plan(max_batch, None, None, None, None, None, False, with_ws=False)
        self.ws_bytes = int(self.lib.dr4sr_gru4rec_workspace_bytes(C.byref(probe)))
        if self.ws_bytes <= 0:
            raise _lib.Dr4srError(f"GRU4Rec shape L={self.L} D={self.D} hidden={self.H} layers={self.n_layer}: dr4sr_gru4rec_workspace_bytes failed: "
                                  + _lib._ERR.get(self.ws_bytes, str(self.ws_bytes)) + " — built for L <= 64, D = 64, hidden 128 or 256")
        self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=dev)     # zero ONCE: the cooperative recurrence's
        #                                                                               granule tags / launch counter live in it
        self.neg_scratch = torch.zeros(max_batch * L, dtype=torch.int64, device=dev)

    def _plan(self, B, in_item_id, item_id, seqlen, rows, neg_item, sample_neg, with_ws=True, perm_sel=None, loss_log=None):
        p = _lib.GruPlan()
        p.abi_version = _lib.ABI_VERSION
        p.B, p.L, p.D, p.H, p.n_layer, p.n_items = B, self.L, self.D, self.H, self.n_layer, self.n_items
        p.p_drop, p.seed = self.p_drop, self.seed
        p.params, p.grads = self.params.data_ptr(), self.grads.data_ptr()
        p.adam_m, p.adam_v, p.n_params = self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.n_params
        for name, t in (("in_item_id", in_item_id), ("item_id", item_id), ("seqlen", seqlen), ("rows", rows), ("neg_item", neg_item)):
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
        self._keep = [in_item_id, item_id, seqlen, rows, neg_item, perm_sel, loss_log]
        return p

    def make_plan(self, in_item_id, item_id, seqlen, rows=None, neg_item=None, sample_neg=None, perm_sel=None, loss_log=None):
        B = int(rows.shape[0] if rows is not None else in_item_id.shape[0])
        if B > self.max_batch:
            raise _lib.Dr4srError(f"batch {B} > max_batch {self.max_batch}")
        if in_item_id.dim() != 2 or int(in_item_id.shape[1]) != self.L:
            raise _lib.Dr4srError(f"GRU4Rec: in_item_id must be [rows, {self.L}], got {tuple(in_item_id.shape)}")
        if item_id is not None and item_id.shape != in_item_id.shape:
            # pooling 'origin' keeps one query per position: (query * item_embedding(target)).sum(-1) (basemodel.py:182) needs [rows, L] targets
            raise _lib.Dr4srError(f"GRU4Rec scores one query per position: item_id must be {tuple(in_item_id.shape)}, got {tuple(item_id.shape)} "
                                  "(prefix-row data is the one-query-per-row models' format)")
        if seqlen is not None and (seqlen.dim() != 1 or seqlen.shape[0] != in_item_id.shape[0]):
            raise _lib.Dr4srError(f"GRU4Rec: seqlen must be [{int(in_item_id.shape[0])}], got {tuple(seqlen.shape)}")
        if sample_neg is None:
            sample_neg = neg_item is None
        if neg_item is None:
            neg_item = self.neg_scratch
        return self._plan(B, in_item_id, item_id, seqlen, rows, neg_item, sample_neg, perm_sel=perm_sel, loss_log=loss_log)

    def fwd_bwd(self, plan):
        _lib.check(self.lib.dr4sr_gru4rec_fwd_bwd(C.byref(plan), _lib.cur_stream()), "dr4sr_gru4rec_fwd_bwd")

    def train_step(self, plan):
        _lib.check(self.lib.dr4sr_gru4rec_train_step(C.byref(plan), _lib.cur_stream()), "dr4sr_gru4rec_train_step")

    def train_steps(self, plan, n: int):
        """n consecutive steps with one prep launch (the optimizer launches prepare the next step): needs a plan whose batch is selected
        on the device (perm_sel), else every step would train on the same rows"""
        _lib.check(self.lib.dr4sr_gru4rec_train_steps(C.byref(plan), int(n), _lib.cur_stream()), "dr4sr_gru4rec_train_steps")

    def adam_step(self, plan=None):
        if plan is not None:               # the plan's optimizer launch: logs the step's loss when the plan carries a loss log
            _lib.check(self.lib.dr4sr_gru4rec_adam_step(C.byref(plan), _lib.cur_stream()), "dr4sr_gru4rec_adam_step")
            return
        _lib.check(self.lib.dr4sr_optimizer_flat(self.optimizer, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                                 _lib.ptr(self.adam_v), self.n_params, _lib.ptr(self.state), self.lr, self.betas[0],
                                                 self.betas[1], self.adam_eps, self.weight_decay, _lib.cur_stream()), "dr4sr_optimizer_flat")

    def encode(self, plan, training: bool, pooling: int, out: Optional[torch.Tensor] = None):
        shape = (plan.B, self.D) if pooling == _lib.POOL_LAST else (plan.B, self.L, self.D)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_gru4rec_encode(C.byref(plan), int(training), pooling, _lib.ptr(out), _lib.cur_stream()),
                   "dr4sr_gru4rec_encode")
        return out

    def encode_bwd(self, plan, training: bool, pooling: int, d_out: torch.Tensor):
        _lib.check(self.lib.dr4sr_gru4rec_encode_bwd(C.byref(plan), int(training), pooling, _lib.ptr(d_out.contiguous()),
                                                     _lib.cur_stream()), "dr4sr_gru4rec_encode_bwd")

    def check_coop(self):
        """the cooperative recurrence never hangs: a wait that runs out sets a sticky error word (int32 word 2 of the workspace,
        include/dr4sr_hip.h) and the results of that launch are garbage — turn it into an exception (host sync: call per epoch)"""
        if int(self.workspace[:16].view(torch.int32)[2]) != 0:
            raise _lib.Dr4srError("cooperative GRU recurrence: an exchange wait timed out (workgroups not co-resident?); "
                                  "set DR4SR_GRU_NOCOOP=1 to use the single-workgroup recurrence")

    check_device_error = check_coop

    def uses_cooperative(self, B: int) -> bool:
        """True when a batch of B sequences runs the multi-CU cooperative recurrence on this device (csrc/gru_coop.hip)"""
        return bool(self.lib.dr4sr_gru4rec_uses_cooperative(int(B), int(self.H)))

    def uses_wavefront(self, B: int) -> bool:
        """True when a batch of B sequences runs both layers' forward recurrences in ONE launch, the second layer behind the first
        (csrc/gru_coop.hip k_gru_fwd_wave: two layers, hidden 256, batches the 16-slice cooperative form takes)"""
        return bool(self.lib.dr4sr_gru4rec_uses_wavefront(int(B), int(self.H), int(self.n_layer), int(self.L)))

    def loss_and_count(self):
        self.check_coop()
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
