"""MetaModel (DR4SR+) with the class surface of the reference's model/metamodel.py (MetaModel :19-197) and the
implicit-differentiation optimiser of utils/utils.py:134-252 (Hypergrad, MetaOptimizer), over the HIP sub-models.

  inner step (metamodel.py:174-194)   loss = sum_p weight_p * loss_p,  weight = gumbel_softmax(meta_module(query))[..., 0],
                                      1 on pattern rows (user_id == 0), 0 on PAD targets.  The gradient reaches the
                                      sub-model through loss_p AND through weight_p(query) as in the reference.
  outer step (metamodel.py:123-166)   every `interval` steps after `warmup_epoch`: hyper-gradient of the plain loss on a
                                      second batch w.r.t. the meta module via a truncated Neumann series (3 Hessian-vector
                                      products) and one mixed second derivative; clip to 10; SGD(momentum 0.9, wd).

Second order WITHOUT double-backward kernels: the library's hand-written first-order backward G(W) = dL_train/dW is
differentiated by central differences along the needed directions only (dr4sr_fd_* in include/dr4sr_hip.h):
      H v            = [G(W + e v) - G(W - e v)] / 2e            (x hpo_lr = 1e-3 inside the Neumann recursion)
      d/dphi (G . p) = (4 D(e) - D(2e)) / 3,  D(h) = [dL_train/dphi(W + h p) - dL_train/dphi(W - h p)] / 2h   (Richardson, O(e^4))
with identical dropout masks, negatives and Gumbel noise in every evaluation and the meta module's ReLU pattern frozen at
W (autograd's ReLU'' = 0).  tests/test_meta_oracle.py shows this form within 2e-4 of the reference's double-backward result
(golden vectors from RUNNING the reference), tests/test_gpu_meta.py checks the HIP path against the same vectors, and
tests/test_oracle_trained.py / tests/test_gpu_trained.py do both at the reference's SHIPPED trained checkpoint on real toys rows.

All arithmetic runs in libdr4sr_hip.so; torch provides buffers, copies and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.nn as nn

from .. import _lib
from .. import parallel
from ..parallel import allreduce_flat, shard_bounds
from ..utils.graphs import capture
from ..utils.config import get_model_class, load_config
from .basemodel import BaseModel, normal_initialization
from .sasrec import _Linear


class _PhiStore:
    """flat fp32 storage of the meta module (layout of dr4sr_meta_param_count) + its gradient, exposing named views"""

    def __init__(self, lib, D: int, device):
        n = int(lib.dr4sr_meta_param_count(D))
        if n <= 0:
            raise _lib.Dr4srError(f"MetaModel: embed_dim {D} unsupported by the meta-module kernels (D = 64)")
        self.n = n
        self.params = torch.zeros(n, dtype=torch.float32, device=device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.views, self.grad_views = OrderedDict(), OrderedDict()
        off = 0
        for name, shp in (("0.weight", (D, D)), ("0.bias", (D,)), ("2.weight", (2, D)), ("2.bias", (2,))):
            k = 1
            for s in shp:
                k *= s
            self.views[name] = self.params[off:off + k].view(shp)
            self.grad_views[name] = self.grads[off:off + k].view(shp)
            off += k


class _Select(torch.autograd.Function):
    """MetaModel.selection + the weight masks of training_step through dr4sr_meta_select_fwd/_bwd (API path)"""

    @staticmethod
    def forward(ctx, meta, query, anchor, user_id, target):
        q = query.contiguous()
        w = meta._select_fwd(q, user_id, target, meta._gumbel, None, None)
        ctx.meta = meta
        ctx.save_for_backward(q, user_id, target)
        return w.view(target.shape)

    @staticmethod
    def backward(ctx, gw):
        q, user_id, target = ctx.saved_tensors
        dq = torch.zeros_like(q)
        ctx.meta._select_bwd(q, user_id, target, ctx.meta._gumbel, None, gw.contiguous().view(-1).float(), dq)
        return None, dq, None, None, None


class MetaOptimizer:
    """utils/utils.py:207-255 facade: clip_grad_norm_(10) + the meta optimizer MetaModel._get_meta_optimizers chose
    (metamodel.py:59-81): 'sgd' = SGD(lr, weight_decay, momentum 0.9) — configs/metamodel.yaml's — through dr4sr_meta_sgd_step; 'adam' =
    Adam(lr), 'adagrad' = Adagrad(lr), 'rmsprop' = RMSprop(lr), any other name = Adam(lr, weight_decay) through dr4sr_meta_opt_step;
    'sparse_adam' = torch.optim.SparseAdam, which raises on the dense hyper-gradient in the reference's first outer step — raised here
    when the optimizer is built."""

    def __init__(self, model, lr, hpo_lr, weight_decay, truncate_iter=3, max_grad_norm=10.0, name="sgd"):
        self.model, self.lr, self.hpo_lr = model, lr, hpo_lr
        self.truncate_iter, self.max_grad_norm, self.momentum = truncate_iter, max_grad_norm, 0.9
        n = str(name).lower()
        if n == "sparse_adam":
            raise RuntimeError("SparseAdam does not support dense gradients, please consider Adam instead")
        self.kind = {"sgd": _lib.OPT_SGD, "adam": _lib.OPT_ADAM, "adagrad": _lib.OPT_ADAGRAD, "rmsprop": _lib.OPT_RMSPROP}.get(n, _lib.OPT_ADAM)
        # only SGD and the else branch pass meta_weight_decay to torch (metamodel.py:68-69, :77)
        self.weight_decay = weight_decay if (n == "sgd" or n not in ("adam", "adagrad", "rmsprop")) else 0.0
        self.betas = (0.9, 0.99) if self.kind == _lib.OPT_RMSPROP else (0.9, 0.999)
        self.eps = 1e-10 if self.kind == _lib.OPT_ADAGRAD else 1e-8
        dev = model._phi.params.device
        self.momentum_buf = torch.zeros_like(model._phi.params)          # SGD: momentum buffer; Adam: exp_avg
        self.second_buf = torch.zeros_like(model._phi.params)            # exp_avg_sq / state_sum / square_avg
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.last_grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def zero_grad(self):
        self.model._phi.grads.zero_()

    def step_with(self, hyper_grad: torch.Tensor):
        phi = self.model._phi
        max_norm = self.max_grad_norm if self.max_grad_norm is not None else 0.0
        if self.kind == _lib.OPT_SGD:
            _lib.check(self.model.lib.dr4sr_meta_sgd_step(_lib.ptr(phi.params), _lib.ptr(hyper_grad), _lib.ptr(self.momentum_buf), phi.n,
                                                          self.lr, self.momentum, self.weight_decay, max_norm,
                                                          _lib.ptr(self.step_count), _lib.ptr(self.last_grad_norm), _lib.cur_stream()),
                       "dr4sr_meta_sgd_step")
        else:
            _lib.check(self.model.lib.dr4sr_meta_opt_step(self.kind, _lib.ptr(phi.params), _lib.ptr(hyper_grad), _lib.ptr(self.momentum_buf),
                                                          _lib.ptr(self.second_buf), phi.n, self.lr, self.betas[0], self.betas[1], self.eps,
                                                          self.weight_decay, max_norm, _lib.ptr(self.step_count),
                                                          _lib.ptr(self.last_grad_norm), _lib.cur_stream()), "dr4sr_meta_opt_step")


class MetaModel(BaseModel):
    def __init__(self, config: Dict, dataset_list) -> None:
        super().__init__(config, dataset_list)
        self.interval = config["train"]["interval"]
        self.step_counter = 0
        self.item_embedding = None                    # MetaModel is just a trainer without item embedding (metamodel.py:24)
        self.tau = nn.Parameter(torch.ones(1, device=self.device) * 10)
        self.counter = 0
        self._gumbel = None                           # explicit Gumbel noise [n,2] (tests); None = Philox in-kernel
        self._tau_host = None
        self._bufs: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------------------------------ setup
    def _init_model(self, train_data):
        self.sub_model: BaseModel = self._register_sub_model()
        self.sub_model._init_model(train_data)
        self.item_embedding = self.sub_model.item_embedding
        self.engine = self.sub_model.engine           # topk / evaluate run on the sub-model's engine
        self.lib = self.engine.lib
        self.device = self.sub_model.device
        self._phi = _PhiStore(self.lib, self.embed_dim, self.device)
        self.meta_module: nn.Module = self._register_meta_modules()
        self.meta_module.apply(normal_initialization)
        if self.world_size > 1:
            parallel.broadcast(self._phi.params, src=0)
        self.meta_optimizer = self._get_meta_optimizers()
        self.metaloader_iter = iter(self.current_epoch_metaloaders(nepoch=0))
        n = self.engine.n_params
        self._sel_ws = torch.empty(int(self.lib.dr4sr_meta_select_workspace_floats(
            int(self.config["train"]["batch_size"]) * self.max_seq_len)), dtype=torch.float32, device=self.device)
        self._stats = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._e = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._fd_scratch = torch.zeros(int(self.lib.dr4sr_fd_step_size_scratch_floats()), dtype=torch.float32, device=self.device)
        self._tail_p = torch.zeros(_lib.GRAD_TAIL, dtype=torch.float32, device=self.device)

    def _register_sub_model(self) -> BaseModel:
        sub_cfg = load_config({"dataset": self.config["data"]["dataset"], "model": self.config["model"]["sub_model"]})
        sub_cfg["train"]["device"] = self.config["train"]["device"]     # the reference hard-codes 0 (metamodel.py:47)
        for k in ("batch_size", "seed"):
            sub_cfg["train"][k] = self.config["train"][k]
        for sec, kv in (self.config["model"].get("sub_overrides") or {}).items():     # extension: e.g. {'model': {'dropout_rate': 0}}
            sub_cfg[sec].update(kv)
        self.logger.info(sub_cfg)
        name = sub_cfg["model"]["model"]
        if name not in ("SASRec", "GRU4Rec", "FMLP", "CL4SRec"):
            # tuple-loss sub-models (metamodel.py:186-192) other than CL4SRec would silently train a different objective: the weighted
            # steps here add CL4SRec's un-weighted contrastive term explicitly (_cl_extra)
            raise NotImplementedError(f"MetaModel sub_model {name!r}: the HIP path implements SASRec, GRU4Rec, FMLP and CL4SRec sub-models")
        return get_model_class(name)(sub_cfg, self.dataset_list)

    def _cl_sub(self) -> bool:
        from .cl4srec import CL4SRec
        return isinstance(self.sub_model, CL4SRec)

    def _cl_extra(self, batch, views=None):
        """metamodel.py:186-192, the tuple-loss branch: CL4SRec's contrastive rows enter the loss UN-weighted —
        + cl_weight * sum_rows InfoNCE_row / kept rows.  Accumulated into the flat gradient behind the weighted BCE step (whose
        {n_valid, weighted loss sum} tail scales it, CL4SRec._cl_term) and folded into the tail's loss."""
        if self._cl_sub():
            self.sub_model._cl_term(batch["in_" + self.fiid], batch["seqlen"], views=views, dp_counts=batch.get("_dp_counts"), fold_loss=True)

    def _register_meta_modules(self) -> nn.Module:
        D = self.embed_dim
        return nn.Sequential(_Linear(self._phi, "0.", D, D), nn.ReLU(), _Linear(self._phi, "2.", 2, D))

    def _get_meta_optimizers(self):
        tc = self.config["train"]
        return MetaOptimizer(self, float(tc["meta_learning_rate"]), float(tc["hpo_learning_rate"]), float(tc["meta_weight_decay"]),
                             name=tc["meta_optimizer"])

    def forward(self, batch):
        return self.sub_model.forward(batch)

    def current_epoch_metaloaders(self, nepoch):
        return self.dataset_list[0].get_loader()

    def _tau_eff(self) -> float:
        """clip(tau, min=tau_min) (metamodel.py:171).  tau is a Parameter that no optimizer ever steps (it is not in aux_params,
        metamodel.py:142), so its host copy is cached — no device->host read inside the step (or inside a graph capture)."""
        if self._tau_host is None:
            self._tau_host = max(float(self.tau.detach()), float(self.config["model"]["tau_min"]))
        return self._tau_host

    def load_state_dict(self, *args, **kwargs):
        self._tau_host = None
        return super().load_state_dict(*args, **kwargs)

    def _buf(self, name, n, dtype=torch.float32):
        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:n]

    # ------------------------------------------------------------------------------------------ selection kernels
    @staticmethod
    def _bl(target):
        return int(target.shape[0]), (int(target.shape[1]) if target.dim() == 2 else 1)

    def _select_fwd(self, q, user_id, target, gumbel, gate_in, gate_out):
        B, L = self._bl(target)
        w = torch.empty(B * L, dtype=torch.float32, device=q.device)
        _lib.check(self.lib.dr4sr_meta_select_fwd(_lib.ptr(q), _lib.ptr(self._phi.params), _lib.ptr(gumbel), self.engine.seed,
                                                  0, _lib.ptr(self._noise_step()), self._tau_eff(), _lib.ptr(user_id), _lib.ptr(target.contiguous()),
                                                  B, L, self.embed_dim, _lib.ptr(gate_in), _lib.ptr(gate_out), _lib.ptr(w),
                                                  _lib.cur_stream()), "dr4sr_meta_select_fwd")
        return w

    def _select_bwd(self, q, user_id, target, gumbel, gate_in, d_weight, d_query):
        B, L = self._bl(target)
        need = int(self.lib.dr4sr_meta_select_workspace_floats(B * L))
        if self._sel_ws.numel() < need:
            self._sel_ws = torch.empty(need, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dr4sr_meta_select_bwd(_lib.ptr(q), _lib.ptr(self._phi.params), _lib.ptr(gumbel), self.engine.seed,
                                                  0, _lib.ptr(self._noise_step()), self._tau_eff(), _lib.ptr(user_id), _lib.ptr(target.contiguous()),
                                                  B, L, self.embed_dim, _lib.ptr(gate_in), _lib.ptr(d_weight), None,
                                                  _lib.ptr(d_query), _lib.ptr(self._phi.grads), _lib.ptr(self._sel_ws),
                                                  _lib.cur_stream()), "dr4sr_meta_select_bwd")

    def _noise_step(self):
        """device word keying the Gumbel noise: the sub-model engine's RNG step (bumped by every training forward), so that a
        captured graph draws fresh noise per replay and the hyper-gradient probes (which restore that word) share one draw"""
        return self.engine.state[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1]

    def selection(self, query):
        """metamodel.py:169-172 (no masks): weight in (0,1) per position"""
        shp = query.shape[:-1]
        ones = torch.ones(shp, dtype=torch.int64, device=query.device)
        return _Select.apply(self, query, self._phi.params, None, ones).view(shp)

    # ------------------------------------------------------------------------------------------ API path (autograd)
    def training_step(self, batch, reduce=True, return_query=True, align=False):
        if self._cl_sub():
            # metamodel.py:186-192: rst = (loss_value[0] * weight).sum() + loss_value[1].sum(), and the contrastive rows of
            # reduce=False sum to cl_weight * the reduce=True loss (data_augmentation.py:349-353: cross_entropy(reduction='none') /
            # batch_size) — taken in that form, whose backward the InfoNCE kernel implements
            from .sasrec import SASRec
            sub = self.sub_model
            aug = sub.augmentation_model.augmentation
            if hasattr(aug, "begin_step"):
                aug.begin_step()
            loss_value, query = SASRec.training_step(sub, batch, reduce=False, return_query=True)
            cl = sub.config["model"]["cl_weight"] * sub.augmentation_model(batch, sub.query_encoder, reduce=True)["cl_loss"]
            if hasattr(aug, "end_step"):
                aug.end_step()
        else:
            loss_value, query = self.sub_model.training_step(batch, reduce=False, return_query=True, align=False)
            cl = 0.0
        weight = _Select.apply(self, query, self._phi.params, batch["user_id"].contiguous(), batch[self.fiid].contiguous())
        self.counter += 1
        return (loss_value * weight).sum() + cl

    # ------------------------------------------------------------------------------------------ fused weighted step
    def _weighted_fwd_bwd(self, batch, gate_in=None, gate_out=None):
        """un-normalised gradients of sum_p weight_p loss_p: d/dW into sub.engine.grads (tail = {n_valid, weighted loss sum}),
        d/dphi into self._phi.grads.  Every rank-local quantity is a plain SUM, so DP is one all-reduce of both buffers."""
        sub, eng, lib = self.sub_model, self.engine, self.lib
        tgt, neg, uid = batch[self.fiid].contiguous(), batch["neg_item"].contiguous(), batch["user_id"].contiguous()
        B, L = self._bl(tgt)
        eng.grads.zero_()
        self._phi.grads.zero_()
        self._stats.zero_()
        q = sub._encode_raw(batch, True)
        if q.numel() != B * L * eng.D:
            raise _lib.Dr4srError(f"MetaModel: the sub-model returns {q.numel() // eng.D} queries for {B * L} targets "
                                  f"(targets {tuple(tgt.shape)}); a one-query-per-row sub-model (FMLP) needs the prefix-row data format")
        lp = self._buf("lp", B * L)
        E = self.item_embedding.weight
        _lib.check(lib.dr4sr_score_bce_fwd(_lib.ptr(q), _lib.ptr(E), _lib.ptr(tgt), _lib.ptr(neg), None, None, _lib.ptr(lp),
                                           _lib.ptr(self._stats), B, L, eng.D, _lib.cur_stream()), "score_bce_fwd")
        w = self._select_fwd(q, uid, tgt, self._gumbel, gate_in, gate_out)
        dq = self._buf("dq", q.numel()).view(q.shape)
        dE = eng.grad_views["item_embedding.weight"]
        _lib.check(lib.dr4sr_score_bce_bwd(_lib.ptr(q), _lib.ptr(E), _lib.ptr(tgt), _lib.ptr(neg), _lib.ptr(w), None, _lib.ptr(dq),
                                           _lib.ptr(dE), B, L, eng.D, _lib.cur_stream()), "score_bce_bwd")
        self._select_bwd(q, uid, tgt, self._gumbel, gate_in, lp, dq)
        sub._encode_bwd_raw(batch, dq, True)
        tail = eng.grads[eng.n_params:eng.n_params + 2]
        tail[0:1].copy_(self._stats[0:1])
        tail[1:2].copy_(torch.dot(w, lp).view(1))      # reported loss only
        self._cl_extra(batch, views=batch.get("_views"))
        return w, lp

    def _phi_only(self, batch, gate_in):
        """d/dphi of sum_p weight_p loss_p at the current sub-model parameters, WITHOUT the encoder's backward: d weight_p / d phi
        needs the query rows and the per-position losses only.  The two mixed-derivative probes of the hyper-gradient
        (utils/utils.py:170-178) want exactly this; d/dW of those evaluations was computed and thrown away before.  Leaves
        d/dphi in self._phi.grads and n_valid in the gradient tail (everything else in engine.grads is zero)."""
        sub, eng, lib = self.sub_model, self.engine, self.lib
        tgt, neg, uid = batch[self.fiid].contiguous(), batch["neg_item"].contiguous(), batch["user_id"].contiguous()
        B, L = self._bl(tgt)
        eng.grads.zero_()
        self._phi.grads.zero_()
        self._stats.zero_()
        q = sub._encode_raw(batch, True)
        lp = self._buf("lp", B * L)
        _lib.check(lib.dr4sr_score_bce_fwd(_lib.ptr(q), _lib.ptr(self.item_embedding.weight), _lib.ptr(tgt), _lib.ptr(neg), None, None,
                                           _lib.ptr(lp), _lib.ptr(self._stats), B, L, eng.D, _lib.cur_stream()), "score_bce_fwd")
        self._select_bwd(q, uid, tgt, self._gumbel, gate_in, lp, None)
        eng.grads[eng.n_params:eng.n_params + 1].copy_(self._stats[0:1])

    def _fused_ok(self) -> bool:
        """SASRec sub-model with d = 64: the weighting runs inside the fused training step (dr4sr_sasrec_fwd_bwd_weighted)"""
        import os
        from .sasrec import SASRec
        # exactly SASRec or CL4SRec (whose extra contrastive term _cl_extra adds behind the weighted step): any other subclass with
        # extra loss terms must not take the plain weighted step
        from .cl4srec import CL4SRec
        return type(self.sub_model) in (SASRec, CL4SRec) and self.embed_dim == 64 and not os.environ.get("DR4SR_META_DENSE")

    def _fused_weighted(self, batch, gate_in=None, gate_out=None, weight_out=None, with_cl=True):
        """un-normalised d/dW of sum_p weight_p loss_p through the 12-launch fused step (no d/dphi: see include/dr4sr_hip.h)"""
        eng = self.engine
        plan = self.sub_model._batch_plan(batch)
        mw = _lib.MetaWeighting()
        mw.phi = self._phi.params.data_ptr()
        mw.gumbel = self._gumbel.data_ptr() if self._gumbel is not None else None
        uid = batch["user_id"].contiguous()
        mw.user_id = uid.data_ptr()
        for name, t in (("gate_in", gate_in), ("gate_out", gate_out), ("weight_out", weight_out)):
            if t is not None:
                setattr(mw, name, t.data_ptr())
        mw.tau = self._tau_eff()
        _lib.check(self.lib.dr4sr_sasrec_fwd_bwd_weighted(C.byref(plan), C.byref(mw), _lib.cur_stream()), "dr4sr_sasrec_fwd_bwd_weighted")
        self._keep_mw = (uid, gate_in, gate_out, weight_out)
        if with_cl:
            self._cl_extra(batch, views=batch.get("_views"))

    def _reduce_grads(self):
        if self.world_size > 1:
            allreduce_flat(self.engine.grads)
            allreduce_flat(self._phi.grads)

    def _local_batch(self, loader, perm, i):
        lo, hi = shard_bounds(i, loader.batch_size, loader.n, self.world_size, self.rank)
        rows = perm[lo:hi]
        batch = {k: v.index_select(0, rows) for k, v in loader.fields.items()}
        batch["index"] = rows
        if self.world_size > 1 and self._cl_sub():        # CL4SRec's in-batch negatives span the GLOBAL batch (CL4SRec._cl_term)
            bs = [shard_bounds(i, loader.batch_size, loader.n, self.world_size, k) for k in range(self.world_size)]
            batch["_dp_counts"] = [b - a for a, b in bs]
        return batch

    def _perm(self, loader):
        perm = loader.permutation()
        if self.world_size > 1:
            parallel.broadcast(perm, src=0)
        return perm

    def training_epoch(self, nepoch):
        loader = self.current_epoch_trainloaders(nepoch)
        sub, eng = self.sub_model, self.engine
        if not nepoch > self.config["train"]["warmup_epoch"]:
            out = sub.training_epoch(nepoch)              # metamodel.py:111-112: plain sub-model steps, no outer loop
            self.step_counter += len(loader)
            return out
        perm = self._perm(loader)
        nb = len(loader)
        if self._fused_ok() and not self._cl_sub() and self.world_size == 1 and bool(self.config["train"].get("hip_graph", True)):
            return [[{"loss_0": self._fused_meta_epoch(loader, perm, nepoch)}]]
        losses = torch.empty(nb, dtype=torch.float32, device=self.device)
        for i in range(nb):
            losses[i] = self._train_batch(self._local_batch(loader, perm, i), nepoch)
        return [[{"loss_0": losses}]]

    def _fused_meta_epoch(self, loader, perm, nepoch):
        """weighted epoch with NO per-step host work: batch selection, negatives, Gumbel noise, weighting, backward, Adam and the
        loss log all run on the device, `steps_per_graph` steps per captured graph; graphs never straddle an outer-loop boundary"""
        sub, eng = self.sub_model, self.engine
        B, n, nb = loader.batch_size, loader.n, len(loader)
        if getattr(self, "_perm_buf", None) is None or self._perm_buf.shape[0] != n:
            self._perm_buf = torch.empty(n, dtype=torch.int64, device=self.device)
            self._perm_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
        if getattr(self, "_loss_log", None) is None or self._loss_log.shape[0] != nb:
            self._loss_log = torch.empty(nb, dtype=torch.float32, device=self.device)
        self._perm_buf.copy_(perm)
        self._perm_counter.zero_()
        group, interval = int(self.config["train"].get("steps_per_graph", 16)), int(self.config["train"]["interval"])
        i = 0
        while i < nb:
            bl = min(B, n - i * B)
            full_left = (n // B) - i if bl == B else 1          # equally sized batches ahead (the ragged tail runs alone)
            k = max(1, min(group, interval - self.step_counter % interval, full_left))
            self._meta_step_graph(loader.fields, bl, k)()
            self.counter += k
            self.step_counter += k
            i += k
            if self.step_counter % interval == 0:
                self._outter_loop(nepoch)
        return self._loss_log.clone()

    def _meta_step_graph(self, fields, bl, k):
        # every device address the captured plan bakes in is part of the key (the loss log / permutation buffers are re-allocated
        # when the train split changes size)
        key = ("inner", fields["in_item_id"].data_ptr(), bl, k, self._loss_log.data_ptr(), self._perm_buf.data_ptr(),
               self._perm_counter.data_ptr())
        if key in self._graphs:
            return self._graphs[key]
        sub, eng, lib = self.sub_model, self.engine, self.lib
        plan = sub._train_plan(fields, sub._rows_buf[:bl], perm_sel=(self._perm_buf, self.config["train"]["batch_size"], 0, self._perm_counter),
                               loss_log=self._loss_log)
        mw = _lib.MetaWeighting()
        mw.phi, mw.user_id, mw.tau = self._phi.params.data_ptr(), fields["user_id"].data_ptr(), self._tau_eff()
        self._keep_inner = getattr(self, "_keep_inner", []) + [plan, mw]

        fuse_prep = bl <= 1024                              # the optimizer launch of step i prepares step i+1 (one prep launch per graph)

        def body():
            for j in range(k):
                fn = lib.dr4sr_sasrec_fwd_bwd_weighted_prepared if (fuse_prep and j > 0) else lib.dr4sr_sasrec_fwd_bwd_weighted
                _lib.check(fn(C.byref(plan), C.byref(mw), _lib.cur_stream()), "fwd_bwd_weighted")
                if fuse_prep and j < k - 1:
                    eng.adam_step_prepare_next(plan)
                else:
                    eng.adam_step(plan)
        undo = [eng.params, eng.adam_m, eng.adam_v, eng.state, self._perm_counter]
        snap = [t.clone() for t in undo]
        body()                                             # warm-up outside capture, side effects undone
        torch.cuda.synchronize()
        for dst, src in zip(undo, snap):
            dst.copy_(src)
        g = torch.cuda.CUDAGraph()
        with capture(g):
            body()
        self._graphs[key] = g.replay
        return g.replay

    def _train_batch(self, batch, nepoch):
        """one post-warm-up iteration of metamodel.py:101-120: weighted step, sub-model Adam, outer loop on the interval"""
        sub, eng = self.sub_model, self.engine
        bl = int(batch["user_id"].shape[0])
        cl_dp = self._cl_sub() and self.world_size > 1        # the contrastive term gathers the global batch inside the step: eager launches
        if bl > 0 and bool(self.config["train"].get("hip_graph", True)) and not cl_dp:
            st, run = self._weighted_graph(batch)
            for k in self._STATIC_KEYS:                    # the only per-step host work: four small copies + one graph replay
                st[k].copy_(batch[k])
            run()
        else:
            if bl > 0:
                batch["neg_item"] = self._neg_sampling(batch)
                if self._fused_ok():
                    self._fused_weighted(batch)
                else:
                    self._weighted_fwd_bwd(batch)
            else:                                          # a rank whose slice of the tail batch is empty contributes zeros
                eng.grads.zero_()
                self._phi.grads.zero_()
                self._cl_extra(batch)                      # (... and still takes part in the gather of the global batch)
            self._reduce_grads()
            eng.adam_step(sub._api_plan())
        self.counter += 1
        loss = eng.grads[eng.n_params + 1] / eng.grads[eng.n_params]     # metamodel.py:186-194: sum_p w_p loss_p (loss_p is / n_valid)
        self.step_counter += 1
        if self.step_counter % self.config["train"]["interval"] == 0:
            self._outter_loop(nepoch)
        return loss

    _STATIC_KEYS = ("user_id", "in_item_id", "item_id", "seqlen")

    def _weighted_graph(self, batch):
        """captured HIP graph(s) of the weighted step on static copies of the batch tensors: negative sampling keyed by the engine's
        device RNG step, weighted fwd/bwd, [all-reduce between two graphs], sub-model Adam"""
        bl = int(batch["user_id"].shape[0])
        key = (bl, tuple(batch["item_id"].shape[1:]))
        if key in self._graphs:
            return self._graphs[key]
        sub, eng, lib = self.sub_model, self.engine, self.lib
        st = {k: torch.empty_like(batch[k]) for k in self._STATIC_KEYS}
        tgt = st["item_id"]
        st["neg_item"] = torch.empty(*tgt.shape, 1, dtype=torch.int64, device=self.device)
        for k in self._STATIC_KEYS:
            st[k].copy_(batch[k])

        def fwd_bwd():
            _lib.check(lib.dr4sr_neg_sample_dev(_lib.ptr(st["neg_item"]), tgt.numel(), self.num_items, eng.seed ^ 0x5DEECE66D,
                                                _lib.ptr(self._noise_step()), _lib.cur_stream()), "neg_sample_dev")
            if self._fused_ok():
                self._fused_weighted(st)
            else:
                self._weighted_fwd_bwd(st)

        def adam():
            eng.adam_step(sub._api_plan())
        undo = [eng.params, eng.adam_m, eng.adam_v] + list(getattr(eng, "states", [eng.state]))
        if self._cl_sub():                                 # the views' draws are keyed by a device counter the graph advances
            aug = sub.augmentation_model.augmentation
            if getattr(aug, "step_dev", None) is None:
                sub._api_graph_begin()
            undo.append(aug.step_dev)
        snap = [t.clone() for t in undo]
        fwd_bwd()                                          # warm-up outside capture (allocates the persistent scratch buffers);
        adam()                                             # no collective here: ranks build their graphs at different steps
        torch.cuda.synchronize()
        for dst, src in zip(undo, snap):
            dst.copy_(src)
        if self.world_size == 1:
            g = torch.cuda.CUDAGraph()
            with capture(g):
                fwd_bwd()
                adam()
            run = g.replay
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with capture(ga):
                fwd_bwd()
            with capture(gb):
                adam()

            def run():
                ga.replay()
                self._reduce_grads()
                gb.replay()
        self._graphs[key] = (st, run)
        return self._graphs[key]

    def _neg_sampling(self, batch):
        return self.sub_model._neg_sampling(batch)

    # ------------------------------------------------------------------------------------------ outer loop
    def _outter_loop(self, nepoch):
        """metamodel.py:123-166: one meta ('validation') batch + one train batch, fresh shuffles, then MetaOptimizer.step"""
        ml = self.current_epoch_metaloaders(nepoch)
        bv = self._local_batch(ml, self._perm(ml), 0)
        bv["neg_item"] = self._neg_sampling(bv)
        tl = self.current_epoch_trainloaders(nepoch)
        bt = self._local_batch(tl, self._perm(tl), 0)
        bt["neg_item"] = self._neg_sampling(bt)
        self.hypergrad_step(bv, bt)

    def hypergrad(self, bv, bt) -> torch.Tensor:
        """Hypergrad.grad (utils/utils.py:145-178) -> flat d/dphi [n_phi]; see the module docstring for the formulation."""
        sub, eng, lib, st = self.sub_model, self.engine, self.lib, _lib.cur_stream
        n, nphi = eng.n_params, self._phi.n
        mo = self.meta_optimizer
        rel = float(self.config["train"].get("hypergrad_rel_step", 5e-4))
        theta0 = self._buf("theta0", n)
        theta0.copy_(eng.params)
        v, pacc, gp = self._buf("v", n), self._buf("pacc", n), self._buf("gp", n + _lib.GRAD_TAIL)
        # dL_val/dW: plain training_step on the meta batch (fresh dropout masks)                      utils.py:154-159
        eng.fwd_bwd(sub._batch_plan(bv))
        self._cl_extra(bv, views=bv.get("_views"))          # CL4SRec sub-model: L_val = BCE + cl_weight * InfoNCE on fresh views
        if self.world_size > 1:
            allreduce_flat(eng.grads)
        _lib.check(lib.dr4sr_scale_by(_lib.ptr(v), _lib.ptr(eng.grads), _lib.ptr(eng.grads[n:n + 1]), n, st()), "scale_by")
        pacc.copy_(v)
        # the train-batch graph: ONE set of dropout masks / Gumbel noise shared by every evaluation     utils.py:161-166
        rng0 = eng.state[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].clone()
        self.counter += 1
        gate = self._buf("gate", bt[self.fiid].numel(), torch.int64)
        q0 = sub._encode_raw(bt, True)
        self._select_fwd(q0, bt["user_id"].contiguous(), bt[self.fiid].contiguous(), self._gumbel, None, gate)
        rng_views = None
        if self._cl_sub():
            # ONE draw of the train batch's two views (and one set of dropout masks in the views' engine slots) for every evaluation
            # of L_train: the reference differentiates one graph of loss_train twice (utils/utils.py:161-178)
            bt = dict(bt)
            if bt.get("_views") is None and int(bt["seqlen"].shape[0]) > 0:
                aug = sub.augmentation_model.augmentation
                if hasattr(aug, "begin_step"):
                    aug.begin_step()
                ids_t, len_t = bt["in_" + self.fiid].contiguous(), bt["seqlen"].contiguous()
                bt["_views"] = aug.two_views(ids_t, len_t) if hasattr(aug, "two_views") else (aug(ids_t, len_t), aug(ids_t, len_t))
                if hasattr(aug, "end_step"):
                    aug.end_step()
            rng_views = [s[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].clone() for s in eng.states[1:]]

        fused = self._fused_ok()
        # Neumann probes.  DEFAULT (round 5): central differences, H v ~ [G(W + e v) - G(W - e v)] / 2e — O(e^2) truncation,
        # 6 evaluations of L_train; the reference's double backward is exact, and the central form stayed within 2.2e-5 of it in every
        # regime measured.  OPT-IN speed switch `train.hypergrad_forward_hvp: true` (round 4's default for BCE sub-models): one-sided
        # probes against a base gradient G(W) evaluated once — 3 evaluations instead of 6 (outer step 1.5 -> 1.32 ms) with O(e)
        # truncation: 1.2e-5 (SASRec, init), 1.3e-5 (shipped trained checkpoint), but 3.9e-4 where the curvature is large (CL4SRec's
        # InfoNCE term) — measured at two operating points per model only, hence not the default (INTEGRATION.md "Limitations").
        fwd_hvp = bool(self.config["train"].get("hypergrad_forward_hvp", False))
        g0 = self._buf("g0", n + _lib.GRAD_TAIL) if fwd_hvp else None
        gate_packed = None

        def restore_rng():
            eng.state[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].copy_(rng0)
            if rng_views is not None:
                for s_, r_ in zip(eng.states[1:], rng_views):
                    s_[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].copy_(r_)
        if fused:                                          # the same frozen pattern in the fused step's packed-token order
            gate_packed = self._buf("gate_packed", bt[self.fiid].numel(), torch.int64)
            restore_rng()
            self._fused_weighted(bt, gate_out=gate_packed, with_cl=fwd_hvp)      # (= G(W): the pattern recorded at W is the frozen one)
        elif fwd_hvp:
            restore_rng()
            self._weighted_fwd_bwd(bt, gate_in=gate)
        if fwd_hvp:
            self._reduce_grads()
            g0.copy_(eng.grads)
        def probe(direction, sign, need_phi=False):
            """d/dW (and, for the two mixed-derivative probes, the deterministic d/dphi) of L_train at W0 + sign * e * direction"""
            _lib.check(lib.dr4sr_fd_shift(_lib.ptr(eng.params), _lib.ptr(theta0), _lib.ptr(direction), _lib.ptr(self._e), sign, n,
                                          st()), "fd_shift")
            restore_rng()
            if need_phi:
                self._phi_only(bt, gate)
            elif fused:
                self._fused_weighted(bt, gate_in=gate_packed)
            else:
                self._weighted_fwd_bwd(bt, gate_in=gate)
            self._reduce_grads()

        for _ in range(mo.truncate_iter):                                                             # utils.py:180-205
            _lib.check(lib.dr4sr_fd_step_size_ws(_lib.ptr(theta0), _lib.ptr(v), n, rel, _lib.ptr(self._e), _lib.ptr(self._fd_scratch), st()),
                       "fd_step_size")
            probe(v, 1.0)
            if fwd_hvp:                                    # v -= hpo_lr (G+ - G0) / e  ==  fd_neumann's (G+ - G-) / 2e form with 2 hpo_lr
                _lib.check(lib.dr4sr_fd_neumann(_lib.ptr(v), _lib.ptr(pacc), _lib.ptr(eng.grads), _lib.ptr(g0), _lib.ptr(eng.grads[n:n + 1]),
                                                _lib.ptr(g0[n:n + 1]), _lib.ptr(self._e), 2.0 * mo.hpo_lr, n, st()), "fd_neumann")
                continue
            gp.copy_(eng.grads)
            probe(v, -1.0)
            _lib.check(lib.dr4sr_fd_neumann(_lib.ptr(v), _lib.ptr(pacc), _lib.ptr(gp), _lib.ptr(eng.grads), _lib.ptr(gp[n:n + 1]),
                                            _lib.ptr(eng.grads[n:n + 1]), _lib.ptr(self._e), mo.hpo_lr, n, st()), "fd_neumann")
        # mixed second derivative d/dphi (dL_train/dW . p)                                               utils.py:170-178
        _lib.check(lib.dr4sr_fd_step_size_ws(_lib.ptr(theta0), _lib.ptr(pacc), n, rel, _lib.ptr(self._e), _lib.ptr(self._fd_scratch), st()),
                   "fd_step_size")
        # Richardson-extrapolated central difference over the probes at +-e and +-2e (the probes are forward-only, _phi_only): this
        # term IS the hyper-gradient up to O(hpo_lr), and on TRAINED weights its plain two-point difference carries a truncation
        # error of 1e-3 of the reference's double-backward (tests/test_oracle_trained.py: same figure in fp32 and fp64, i.e. not
        # rounding) — the four-point form 1e-5.  train.hypergrad_richardson: false = the two-point form.
        hyper = self._buf("hyper", nphi)
        if bool(self.config["train"].get("hypergrad_richardson", True)):
            f4, nv4 = self._buf("f4", 4 * nphi).view(4, nphi), self._buf("nv4", 4)
            for j, sign in enumerate((1.0, -1.0, 2.0, -2.0)):
                probe(pacc, sign, need_phi=True)
                f4[j].copy_(self._phi.grads)
                nv4[j:j + 1].copy_(eng.grads[n:n + 1])
            _lib.check(lib.dr4sr_fd_diff4(_lib.ptr(hyper), _lib.ptr(f4[0]), _lib.ptr(f4[1]), _lib.ptr(f4[2]), _lib.ptr(f4[3]), _lib.ptr(nv4),
                                          _lib.ptr(self._e), -1.0, nphi, st()), "fd_diff4")
        else:
            fp = self._buf("fp", nphi)
            probe(pacc, 1.0, need_phi=True)
            fp.copy_(self._phi.grads)
            self._tail_p.copy_(eng.grads[n:n + _lib.GRAD_TAIL])
            probe(pacc, -1.0, need_phi=True)
            _lib.check(lib.dr4sr_fd_diff(_lib.ptr(hyper), _lib.ptr(fp), _lib.ptr(self._phi.grads), _lib.ptr(self._tail_p[0:1]),
                                         _lib.ptr(eng.grads[n:n + 1]), _lib.ptr(self._e), -1.0, nphi, st()), "fd_diff")
        eng.params.copy_(theta0)
        eng.state[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].copy_(rng0 + 1)
        if rng_views is not None:
            for s_, r_ in zip(eng.states[1:], rng_views):
                s_[_lib.STATE_RNGSTEP:_lib.STATE_RNGSTEP + 1].copy_(r_ + 1)
        return hyper

    def hypergrad_step(self, bv, bt):
        """MetaOptimizer.step (utils/utils.py:221-252): hyper-gradient -> p.grad, clip_grad_norm_(10), meta SGD step"""
        hyper = self.hypergrad(bv, bt)
        self._phi.grads.copy_(hyper)
        self.meta_optimizer.step_with(hyper)
        return hyper

    def evaluate(self) -> Dict:
        return super().evaluate()
