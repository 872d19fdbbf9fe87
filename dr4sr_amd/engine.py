"""SasrecEngine — owns the flat fp32 buffers (params / grads / Adam moments), the workspace and the
device state words, and drives the C ABI (include/dr4sr_hip.h).  PyTorch supplies device memory and
the stream only; all arithmetic runs in libdr4sr_hip.so.

Replaces, on the device: one iteration of BaseModel.training_epoch
(/root/reference model/basemodel.py:193-199) for model = SASRec (model/sasrec.py:39-75).
"""
from __future__ import annotations

import logging
import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import _lib

_LAYER_TENSORS = [  # order of include/dr4sr_hip.h "Flat parameter layout"
    ("self_attn.in_proj_weight", lambda D, F: (3 * D, D)),
    ("self_attn.in_proj_bias", lambda D, F: (3 * D,)),
    ("self_attn.out_proj.weight", lambda D, F: (D, D)),
    ("self_attn.out_proj.bias", lambda D, F: (D,)),
    ("linear1.weight", lambda D, F: (F, D)),
    ("linear1.bias", lambda D, F: (F,)),
    ("linear2.weight", lambda D, F: (D, F)),
    ("linear2.bias", lambda D, F: (D,)),
    ("norm1.weight", lambda D, F: (D,)),
    ("norm1.bias", lambda D, F: (D,)),
    ("norm2.weight", lambda D, F: (D,)),
    ("norm2.bias", lambda D, F: (D,)),
]


def param_names(n_layer: int):
    names = ["item_embedding.weight", "query_encoder.position_emb.weight"]
    for i in range(n_layer):
        names += [f"query_encoder.transformer_layer.layers.{i}.{n}" for n, _ in _LAYER_TENSORS]
    return names


def param_shapes(n_items, L, D, F, n_layer):
    shapes = [(n_items, D), (L, D)]
    for _ in range(n_layer):
        shapes += [fn(D, F) for _, fn in _LAYER_TENSORS]
    return shapes


class SasrecEngine:
    def __init__(self, n_items: int, L: int, D: int, H: int, F: int, n_layer: int, ln_eps: float = 1e-12,
                 p_drop: float = 0.0, max_batch: int = 256, device="cuda", seed: int = 2023, lr: float = 1e-3,
                 betas=(0.9, 0.999), adam_eps: float = 1e-8, weight_decay: float = 0.0, n_slots: int = 1):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.Dr4srError("SasrecEngine needs a GPU device; dr4sr_amd has no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_items, self.L, self.D, self.H, self.F, self.n_layer = n_items, L, D, H, F, n_layer
        self.ln_eps, self.p_drop, self.seed = float(ln_eps), float(p_drop), int(seed)
        self.lr, self.betas, self.adam_eps, self.weight_decay = lr, betas, adam_eps, weight_decay
        self.optimizer = _lib.OPT_ADAM             # DR4SR_OPT_* (set_optimizer)
        self.max_batch = max_batch
        self.mean_len = None                     # mean valid length of the training split, set by the model (regime hint)
        self._mean_len = {}
        noff = 2 + 12 * n_layer
        off = (C.c_int64 * noff)()
        self.n_params = int(self.lib.dr4sr_sasrec_param_layout(n_items, L, D, F, n_layer, off))
        self.offsets = list(off)
        dev = self.device
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_params + _lib.GRAD_TAIL, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        self.state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=dev)
        self.names = param_names(n_layer)
        self.shapes = param_shapes(n_items, L, D, F, n_layer)
        self.views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.grad_views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for name, shp, o in zip(self.names, self.shapes, self.offsets):
            n = 1
            for s in shp:
                n *= s
            self.views[name] = self.params[o:o + n].view(shp)
            self.grad_views[name] = self.grads[o:o + n].view(shp)
        probe = self._plan(max_batch, None, None, None, None, None, False, with_ws=False)
        self.ws_bytes = int(self.lib.dr4sr_sasrec_workspace_bytes(C.byref(probe)))
        if self.ws_bytes <= 0:
            raise _lib.Dr4srError(f"SASRec encoder shape L={L} D={D} H={H} F={F} layers={n_layer}: dr4sr_sasrec_workspace_bytes failed: "
                                  + _lib._ERR.get(self.ws_bytes, str(self.ws_bytes))
                                  + " — built for L <= 64, (D, F) in {(64, 128), (64, 256), (128, 128)}, head_dim 32 or 64")
        # slot = (workspace, state words): several forward passes can be alive before their backward passes (CL4SRec encodes
        # three views per step); slot 0 is the default and the one the optimizer's step counter lives in
        self.workspaces = [torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev) for _ in range(n_slots)]
        self.states = [self.state] + [torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=dev) for _ in range(n_slots - 1)]
        self.workspace = self.workspaces[0]
        self.neg_scratch = torch.zeros(max_batch * L, dtype=torch.int64, device=dev)
        self._keep = []          # tensors referenced by the last plan

    # ------------------------------------------------------------------------------------------
    def _plan(self, B, in_item_id, item_id, seqlen, rows, neg_item, sample_neg, with_ws=True, perm_sel=None, slot=0, loss_log=None,
              expected_tokens=None):
        p = _lib.SasrecPlan()
        p.abi_version = _lib.ABI_VERSION
        p.B, p.L, p.D, p.H, p.F, p.n_layer, p.n_items = B, self.L, self.D, self.H, self.F, self.n_layer, self.n_items
        p.ln_eps, p.p_drop, p.seed = self.ln_eps, self.p_drop, self.seed
        p.params, p.grads = self.params.data_ptr(), self.grads.data_ptr()
        p.adam_m, p.adam_v = self.adam_m.data_ptr(), self.adam_v.data_ptr()
        p.n_params = self.n_params
        for name, t in (("in_item_id", in_item_id), ("item_id", item_id), ("seqlen", seqlen), ("rows", rows),
                        ("neg_item", neg_item)):
            if t is not None:
                assert t.dtype == torch.int64 and t.is_contiguous() and t.device == self.device, name
                setattr(p, name, t.data_ptr())
        p.sample_neg = 1 if sample_neg else 0
        if with_ws:
            p.workspace, p.workspace_bytes = self.workspaces[slot].data_ptr(), self.ws_bytes
            p.state = self.states[slot].data_ptr()
            p.seed = (self.seed + 0x9E3779B97F4A7C15 * slot) & 0xFFFFFFFFFFFFFFFF     # independent dropout streams per slot
        else:
            p.state = self.state.data_ptr()
        p.lr, (p.beta1, p.beta2), p.adam_eps, p.weight_decay = self.lr, self.betas, self.adam_eps, self.weight_decay
        p.optimizer = self.optimizer
        if perm_sel is not None:           # (perm[n], stride, offset, counter[1] int32): rows[] is FILLED by the step's first kernel
            perm, stride, offset, counter = perm_sel
            assert rows is not None and perm.dtype == torch.int64 and counter.dtype == torch.int32
            p.perm, p.n_perm, p.perm_stride, p.perm_offset = perm.data_ptr(), int(perm.shape[0]), int(stride), int(offset)
            p.perm_counter = counter.data_ptr()
        if loss_log is not None:           # float32 device buffer: the step's mean loss lands at [batch index] (see dr4sr_hip.h)
            assert loss_log.dtype == torch.float32
            p.loss_log = loss_log.data_ptr()
        p.expected_tokens = int(expected_tokens) if expected_tokens is not None else self._expected_tokens(B, seqlen, rows is not None)
        self._keep = [in_item_id, item_id, seqlen, rows, neg_item, perm_sel, loss_log]
        return p

    def _expected_tokens(self, B, seqlen, dataset_tensor: bool) -> int:
        """regime hint of the plan (include/dr4sr_hip.h: expected_tokens) = B * mean(min(seqlen, L)) when the caller gave none.
        Dataset tensors (batches are rows[] of them; they live as long as the dataset): one device reduction + host read per
        tensor, cached by (address, length, a checksum-free generation = the tensor's _version).  Per-batch tensors are transient —
        the allocator re-uses their memory for batches of other lengths, so they are NEVER cached: `self.mean_len` (the training
        split's mean, set by the model) when known, else measured (one synchronisation; not possible inside a graph capture, where
        the hint stays 0 = capacity rule and a warning says so once)."""
        if seqlen is None:
            return 0
        if dataset_tensor:
            key = (seqlen.data_ptr(), int(seqlen.shape[0]), int(seqlen._version))
            mean = self._mean_len.get(key)
            if mean is None:
                if torch.cuda.is_current_stream_capturing():
                    return 0
                mean = float(seqlen.clamp(0, self.L).float().mean()) if seqlen.numel() else 0.0
                if len(self._mean_len) > 64:
                    self._mean_len.clear()
                self._mean_len[key] = mean
        elif self.mean_len is not None:
            mean = self.mean_len
        elif torch.cuda.is_current_stream_capturing():
            if not getattr(self, "_warned_no_hint", False):
                self._warned_no_hint = True
                logging.getLogger("CDR").warning("dr4sr_amd: no expected_tokens hint for a per-batch plan inside a graph capture "
                                                 "(engine.mean_len unset): launch forms follow the capacity rule")
            return 0
        else:
            mean = float(seqlen.clamp(0, self.L).float().mean()) if seqlen.numel() else 0.0
        return max(1, int(B * mean))

    def make_plan(self, in_item_id, item_id, seqlen, rows=None, neg_item=None, sample_neg=None, perm_sel=None, slot=0, loss_log=None,
                  expected_tokens=None):
        """rows=None: the tensors ARE the batch ([B,L]/[B]); else they are dataset tensors indexed by rows[B].
        perm_sel: fused device-side batch selection (include/dr4sr_hip.h: dr4sr_sasrec_plan.perm).
        expected_tokens: the regime hint, when the caller knows the batch's valid-token count better than the engine's default
        (eval split, augmented views: CL4SRec's crops are much shorter than the training rows)."""
        B = int(rows.shape[0] if rows is not None else in_item_id.shape[0])
        if B > self.max_batch:
            raise _lib.Dr4srError(f"batch {B} > max_batch {self.max_batch}")
        if in_item_id.dim() != 2 or int(in_item_id.shape[1]) != self.L:
            raise _lib.Dr4srError(f"SASRec: in_item_id must be [rows, {self.L}], got {tuple(in_item_id.shape)}")
        if item_id is not None and item_id.shape != in_item_id.shape:
            # pooling 'origin' keeps one query per position: (query * item_embedding(target)).sum(-1) (basemodel.py:182) needs [rows, L] targets
            raise _lib.Dr4srError(f"SASRec scores one query per position: item_id must be {tuple(in_item_id.shape)}, got {tuple(item_id.shape)} "
                                  "(prefix-row data is the one-query-per-row models' format)")
        if seqlen is not None and (seqlen.dim() != 1 or seqlen.shape[0] != in_item_id.shape[0]):
            raise _lib.Dr4srError(f"SASRec: seqlen must be [{int(in_item_id.shape[0])}], got {tuple(seqlen.shape)}")
        if sample_neg is None:
            sample_neg = neg_item is None
        if neg_item is None:
            neg_item = self.neg_scratch
        return self._plan(B, in_item_id, item_id, seqlen, rows, neg_item, sample_neg, perm_sel=perm_sel, slot=slot, loss_log=loss_log,
                          expected_tokens=expected_tokens)

    # ------------------------------------------------------------------------------------------
    def fwd_bwd(self, plan):
        _lib.check(self.lib.dr4sr_sasrec_fwd_bwd(C.byref(plan), _lib.cur_stream()), "dr4sr_sasrec_fwd_bwd")

    def adam_step(self, plan):
        _lib.check(self.lib.dr4sr_adam_step(C.byref(plan), _lib.cur_stream()), "dr4sr_adam_step")

    def train_step(self, plan):
        _lib.check(self.lib.dr4sr_sasrec_train_step(C.byref(plan), _lib.cur_stream()), "dr4sr_sasrec_train_step")

    def fwd_bwd_prepared(self, plan):
        _lib.check(self.lib.dr4sr_sasrec_fwd_bwd_prepared(C.byref(plan), _lib.cur_stream()), "dr4sr_sasrec_fwd_bwd_prepared")

    def adam_step_prepare_next(self, plan):
        _lib.check(self.lib.dr4sr_adam_step_prepare_next(C.byref(plan), _lib.cur_stream()), "dr4sr_adam_step_prepare_next")

    def grad_buckets(self, plan):
        """[(lo, hi), ...] float ranges of self.grads that become final one after the other in a step of this plan: one range
        (everything, tail included) in the latency forms, (item + position table | encoder layers + tail) at scale
        (include/dr4sr_hip.h: dr4sr_sasrec_grad_buckets)"""
        b = (C.c_int64 * 3)()
        n = int(self.lib.dr4sr_sasrec_grad_buckets(C.byref(plan), b))
        if n < 1:
            _lib.check(n, "dr4sr_sasrec_grad_buckets")
        return [(int(b[0]), int(b[1]))] if n == 1 else [(int(b[0]), int(b[1])), (int(b[1]), int(b[2]))]

    def grad_buckets_for(self, rows: int, seqlen=None):
        """grad_buckets of a training plan of `rows` rows drawn from the dataset tensor `seqlen` (a probe plan: no buffers)"""
        keep = self._keep
        probe = self._plan(rows, None, None, None, None, None, False, with_ws=False,
                           expected_tokens=self._expected_tokens(rows, seqlen, True) if seqlen is not None else
                           (max(1, int(rows * self.mean_len)) if self.mean_len is not None else 0))
        self._keep = keep                                    # (the probe references nothing)
        return self.grad_buckets(probe)

    def fwd_bwd_phase(self, plan, prepared: bool, phase: int):
        """phase 1: the step up to the launch that completes grad_buckets()[0]; phase 2: the rest (dr4sr_sasrec_fwd_bwd_phase)"""
        _lib.check(self.lib.dr4sr_sasrec_fwd_bwd_phase(C.byref(plan), int(bool(prepared)), int(phase), _lib.cur_stream()),
                   "dr4sr_sasrec_fwd_bwd_phase")

    def train_steps(self, plan, n: int):
        """n consecutive steps (consecutive batches of plan.perm): one prep launch, the optimizer launches prepare the next step"""
        _lib.check(self.lib.dr4sr_sasrec_train_steps(C.byref(plan), int(n), _lib.cur_stream()), "dr4sr_sasrec_train_steps")

    def encode(self, plan, training: bool, pooling: int, out: Optional[torch.Tensor] = None):
        B = plan.B
        shape = (B, self.D) if pooling in (_lib.POOL_LAST, _lib.POOL_MEAN) else (B, self.L, self.D)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_sasrec_encode(C.byref(plan), int(training), pooling, _lib.ptr(out), _lib.cur_stream()),
                   "dr4sr_sasrec_encode")
        return out

    def encode_bwd(self, plan, training: bool, pooling: int, d_out: torch.Tensor):
        _lib.check(self.lib.dr4sr_sasrec_encode_bwd(C.byref(plan), int(training), pooling, _lib.ptr(d_out.contiguous()),
                                                    _lib.cur_stream()), "dr4sr_sasrec_encode_bwd")

    # ------------------------------------------------------------------------------------------
    def loss_and_count(self):
        """(mean loss, n_valid) of the last fwd_bwd — reads the gradient tail (one host sync)."""
        tail = self.grads[self.n_params:self.n_params + 2].tolist()
        n = tail[0]
        return (tail[1] / n if n > 0 else float("nan")), int(n)

    def normalized_grads(self) -> Dict[str, torch.Tensor]:
        n = self.grads[self.n_params]
        return {k: v / n for k, v in self.grad_views.items()}

    def load_named(self, sd: Dict[str, torch.Tensor]):
        for k, v in self.views.items():
            v.copy_(sd[k].to(self.device, torch.float32))

    def dropout_mask(self, n: int, site: int, step: int, p: Optional[float] = None) -> torch.Tensor:
        n4 = (n + 3) // 4 * 4
        out = torch.empty(n4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_dropout_mask(_lib.ptr(out), n4, self.p_drop if p is None else p, self.seed, step, site,
                                               _lib.cur_stream()), "dr4sr_dropout_mask")
        return out[:n]
