"""RecStudio-style trainer with the interface of the reference's model/basemodel.py (BaseModel :19-407), driving the
HIP engine instead of torch ops.  Same public surface: fit / evaluate / forward / training_step / _neg_sampling /
topk / _init_model / set_eval_domain / load_checkpoint, attrs item_embedding, optimizer, device, fiid, fuid, config,
logged_metrics, ckpt_path.

Two execution paths, numerically the same kernels:
  * API path  — forward() / training_step() return tensors wired into torch.autograd (custom Functions around the
                C ABI); loss.backward() fills p.grad (views of the flat gradient); optimizer.step() is the fused Adam.
  * fast path — training_epoch() replays one captured HIP graph per batch (batch selection, negative sampling,
                forward, scorer+BCE, backward, [RCCL all-reduce], Adam) with no host work in between.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import time
from collections import defaultdict
from typing import Dict, List

import torch
import torch.nn as nn

from .. import _lib, evaluation
from ..data.dataset import BaseDataset, SeparateDataset, SyntheticDataset
from .. import parallel
from ..parallel import allreduce_flat, shard_bounds
from ..utils.graphs import capture
from ..utils import callbacks
from .loss_func import BinaryCrossEntropyLoss, BPRLoss


def normal_initialization(module, initial_range=0.02):
    """utils/utils.py:70-81 of the reference: N(0, 0.02) for Embedding/Linear, PAD row zero, LayerNorm (1, 0)."""
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=initial_range)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=initial_range)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)


class FusedAdam:
    """optimizer facade over dr4sr_adam_step (torch.optim.Adam semantics, dense, flat buffers)."""

    def __init__(self, model):
        self.model = model
        self.param_groups = [{"lr": model.engine.lr, "weight_decay": model.engine.weight_decay}]

    def zero_grad(self, set_to_none: bool = False):
        self.model.engine.grads.zero_()

    def step(self):
        eng = self.model.engine
        eng.grads[eng.n_params:eng.n_params + 1].fill_(1.0)      # API path: the loss is already normalised (fill_: capturable)
        eng.adam_step(self.model._api_plan())

    def state_dict(self):
        eng = self.model.engine
        return {"step": int(eng.state[_lib.STATE_STEP]), "exp_avg": eng.adam_m.clone(), "exp_avg_sq": eng.adam_v.clone()}


class _ScoreBCE(torch.autograd.Function):
    """basemodel.py:204-214 + the configured loss (loss_func.py:9-38 BCE, :40-48 BPR) on the dense query through
    dr4sr_score_bce_fwd/bwd resp. dr4sr_score_bpr_fwd/bwd."""

    @staticmethod
    def forward(ctx, model, query, table, target, neg, reduce):
        lib, eng = model.engine.lib, model.engine
        bpr = isinstance(model.loss_fn, BPRLoss)
        ctx.fns = (lib.dr4sr_score_bpr_fwd, lib.dr4sr_score_bpr_bwd) if bpr else (lib.dr4sr_score_bce_fwd, lib.dr4sr_score_bce_bwd)
        B = target.shape[0]
        L = target.shape[1] if target.dim() == 2 else 1
        q = query.contiguous()
        tgt, ng = target.contiguous().view(-1), neg.contiguous().view(-1)
        lp = torch.empty(B * L, dtype=torch.float32, device=q.device)
        stats = torch.zeros(2, dtype=torch.float32, device=q.device)
        _lib.check(ctx.fns[0](_lib.ptr(q), _lib.ptr(table), _lib.ptr(tgt), _lib.ptr(ng), None, None,
                              _lib.ptr(lp), _lib.ptr(stats), B, L, eng.D, _lib.cur_stream()), "score_loss_fwd")
        ctx.model, ctx.reduce, ctx.shape = model, reduce, (B, L)
        ctx.save_for_backward(q, table, tgt, ng, stats)
        if reduce:
            return stats[1] / stats[0]
        return (lp / stats[0]).view(target.shape)

    @staticmethod
    def backward(ctx, gout):
        q, table, tgt, ng, stats = ctx.saved_tensors
        model = ctx.model
        lib, eng = model.engine.lib, model.engine
        B, L = ctx.shape
        dq = torch.empty_like(q)
        if ctx.reduce:
            w, scale = None, (gout.reshape(1) / stats[0]).contiguous()
        else:
            w, scale = gout.contiguous().view(-1).float(), (1.0 / stats[0]).reshape(1).contiguous()
        dE = eng.grad_views["item_embedding.weight"]
        _lib.check(ctx.fns[1](_lib.ptr(q), _lib.ptr(table), _lib.ptr(tgt), _lib.ptr(ng), _lib.ptr(w),
                              _lib.ptr(scale), _lib.ptr(dq), _lib.ptr(dE), B, L, eng.D, _lib.cur_stream()),
                   "score_loss_bwd")
        return None, dq, None, None, None, None


class BaseModel(nn.Module):
    _det_set_by_model = False          # train.deterministic turned DR4SR_DETERMINISTIC on (as opposed to the user's environment)

    def __init__(self, config: Dict, dataset_list: List[BaseDataset]) -> None:
        super().__init__()
        self.config = config
        self.ckpt_path = None
        self.logger = logging.getLogger("CDR")
        self.dataset_list = dataset_list
        self.device = config["train"]["device"]
        self.fuid, self.fiid = "user_id", "item_id"
        self.domain_name_list = dataset_list[0].domain_name_list
        self.domain_user_mapping = dataset_list[0].domain_user_mapping
        self.domain_item_mapping = dataset_list[0].domain_item_mapping
        self.training_time = 0
        self.inference_time = 0
        self.embed_dim = config["model"]["embed_dim"]
        self.max_seq_len = config["data"]["max_seq_len"]
        self.num_users = dataset_list[0].num_users
        self.num_items = dataset_list[0].num_items
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.engine = None                       # set by the subclass (owns item_embedding's storage)
        # train.deterministic (the reference asks for run-to-run determinism, utils/utils.py:13-20 — seeds + cudnn.deterministic): fixed
        # summation order in every reduction of the step (csrc/step.hip: DR4SR_DETERMINISTIC — the at-scale launch forms at every batch
        # size + ordered partial sums in the weight-gradient launch).  Process-wide, like the reference's flag; costs speed (bench.py:
        # deterministic_mode), not accuracy.
        #  The switch is read when an engine's workspace is carved, so it must be settled BEFORE this model's engine exists: a model
        # built with deterministic = false after one that turned the mode on gets the default mode back (only if a MODEL turned it on — a
        # DR4SR_DETERMINISTIC the user exported stays), and every change is logged.  Engines built earlier keep the mode they were sized for.
        # (a config WITHOUT the key — e.g. the sub-model a MetaModel builds from its own section — leaves the mode as it is.)
        want = bool(config["train"].get("deterministic", False))
        explicit_off = "deterministic" in config["train"] and not want
        user_env = os.environ.get("DR4SR_DETERMINISTIC", "0") not in ("", "0") and not BaseModel._det_set_by_model
        if explicit_off and not user_env and BaseModel._det_set_by_model:
            _lib.set_env("DR4SR_DETERMINISTIC", None)
            BaseModel._det_set_by_model = False
            logging.getLogger("CDR").info("train.deterministic: off for this model (an earlier model of this process had turned it on)")
        if want or user_env or BaseModel._det_set_by_model:
            if want and not user_env and not BaseModel._det_set_by_model:
                BaseModel._det_set_by_model = True
                logging.getLogger("CDR").info("train.deterministic: on (process-wide switch DR4SR_DETERMINISTIC, read when an engine is built)")
            _lib.set_env("DR4SR_DETERMINISTIC", "1")
            # bit-identical fits are tested for every shipped model: SASRec, CL4SRec, FMLP, GRU4Rec and MetaModel around each of them (round 6;
            # tools/det_fit_check.py, tests/test_gpu_deterministic.py) — the fused steps, the autograd-path backwards and the dense scorer all
            # sum in a fixed order in the mode.  A model class outside that list gets a warning, not a promise
            name = type(self).__name__
            inner = str(config["model"].get("sub_model", "")) if name == "MetaModel" else name
            if inner not in ("SASRec", "CL4SRec", "FMLP", "GRU4Rec"):
                logging.getLogger("CDR").warning("train.deterministic: bit-identical fits are tested for SASRec, CL4SRec, FMLP, GRU4Rec and MetaModel around them "
                                                 f"(tests/test_gpu_deterministic.py); {name}{'(' + inner + ')' if name == 'MetaModel' else ''} is not in that list")
        self._graphs = {}

    # ------------------------------------------------------------------------------------------ setup
    @staticmethod
    def _get_dataset_class(config):
        kind = config["data"]["dataset_class"]
        if kind == "general":
            return SeparateDataset
        if kind == "synthetic":
            return SyntheticDataset
        raise NotImplementedError(f"dataset_class '{kind}': only 'general' (SeparateDataset) is on the hot path")

    def _init_model(self, train_data):
        self.apply(normal_initialization)
        self._sync_replicas()
        self.optimizer = self._get_optimizers()
        self.loss_fn = self._get_loss_func()
        self._check_dataset_ids()

    def _check_dataset_ids(self):
        """The reference's nn.Embedding raises IndexError for an item id outside the table; the HIP gathers clamp (a launch cannot raise).
        The ids of the fused path are the splits' resident tensors, so they are checked ONCE here (dr4sr_check_ids) and the reference's
        error is raised before any step runs on a malformed dataset."""
        n = int(getattr(self.engine, "n_items", self.num_items)) if self.engine is not None else self.num_items
        for ds in self.dataset_list:
            fields = getattr(ds, "fields", None)
            try:
                f = fields() if callable(fields) else None
            except Exception:      # noqa: BLE001 — an eval split whose domain is not selected yet has no tensors to look at
                f = None
            if not f:
                continue
            for k in ("in_" + self.fiid, self.fiid):
                t = f.get(k)
                if t is not None and t.is_cuda:
                    _lib.check_ids(t, n, f"{type(ds).__name__}.{k}")

    def _sync_replicas(self):
        if self.world_size > 1:
            parallel.init_distributed(self.device)
            parallel.broadcast(self.engine.params, src=0)

    def _get_optimizers(self):
        # basemodel.py:79-98: adam / sgd / adagrad / rmsprop with torch's defaults, an unknown name = Adam without weight decay,
        # sparse_adam = the RuntimeError torch raises on the reference's dense gradients (_lib.optimizer_settings); all four run as
        # the one fused flat-buffer launch of csrc/step.hip (k_adam<OPT>), next-step prep and data-parallel tail included
        eng = self.engine
        eng.lr = float(self.config["train"]["learning_rate"])
        eng.optimizer, eng.betas, eng.adam_eps, eng.weight_decay = _lib.optimizer_settings(self.config["train"]["optimizer"],
                                                                                           self.config["train"]["weight_decay"])
        return FusedAdam(self)

    def _get_loss_func(self):
        name = self.config["model"]["loss_fn"]
        if name == "bce":
            return BinaryCrossEntropyLoss()
        if name == "bpr":
            return BPRLoss()
        raise NotImplementedError(f"loss_fn '{name}': 'bce' and 'bpr' (basemodel.py:100-104)")

    # ------------------------------------------------------------------------------------------ sampling / steps
    def _neg_sampling(self, batch):
        """uniform over 1..N-1 with replacement, never PAD; [B,L,1] for 2-D targets else [B,1] (basemodel.py:50-61)"""
        tgt = batch[self.fiid]
        n = tgt.numel()
        out = torch.empty(n, dtype=torch.int64, device=tgt.device)
        eng = self.engine
        step_dev = getattr(self, "_neg_step_dev", None)
        if step_dev is not None:                          # captured API step: the call counter lives on the device
            step_dev.add_(1)
            _lib.check(eng.lib.dr4sr_neg_sample_dev(_lib.ptr(out), n, self.num_items, eng.seed ^ 0x5DEECE66D, _lib.ptr(step_dev),
                                                    _lib.cur_stream()), "neg_sample_dev")
            return out.view(*tgt.shape, 1)
        self._neg_calls = getattr(self, "_neg_calls", 0) + 1
        _lib.check(eng.lib.dr4sr_neg_sample(_lib.ptr(out), n, self.num_items, eng.seed ^ 0x5DEECE66D, self._neg_calls,
                                            _lib.cur_stream()), "neg_sample")
        return out.view(*tgt.shape, 1)

    def forward(self, batch, need_pooling=True):
        raise NotImplementedError

    def _api_plan(self):
        raise NotImplementedError

    def training_step(self, batch, reduce=True, return_query=False):
        """basemodel.py:204-214.  loss_fn 'bpr': the reference calls self.loss_fn(pos, neg, reduce=reduce) and BPRLoss.forward has
        no such parameter (loss_func.py:44), i.e. it raises TypeError for every call; here the scalar BPR loss (what BPRLoss.forward
        computes, loss_func.py:45-49) is produced for reduce=True, and reduce=False — which has no BPR definition in the reference —
        keeps the reference's TypeError."""
        if isinstance(self.loss_fn, BPRLoss) and not reduce:
            raise TypeError("BPRLoss.forward() got an unexpected keyword argument 'reduce'")   # as the reference would
        query = self.forward(batch)
        loss = _ScoreBCE.apply(self, query, self.item_embedding.weight, batch[self.fiid], batch["neg_item"], reduce)
        return (loss, query) if return_query else loss

    # ------------------------------------------------------------------------------------------ fit
    def fit(self):
        self.callback = callbacks.EarlyStopping(self, "ndcg@20", self.config["data"]["dataset"],
                                                patience=self.config["train"]["early_stop_patience"])
        self.logger.info("save_dir:" + self.callback.save_dir)
        self._init_model(self.dataset_list[0])
        self.logger.info(self)
        self.fit_loop()

    def fit_loop(self):
        nepoch = 0
        try:
            self.train_start()
            for _ in range(self.config["train"]["epochs"]):
                self.logged_metrics = {"epoch": nepoch}
                tik = time.time()
                self.train()
                training_output_list = self.training_epoch(nepoch)
                torch.cuda.synchronize()
                self.training_time += time.time() - tik

                tik = time.time()
                self.eval()
                for domain in self.domain_name_list:
                    val_dataset = self.dataset_list[1]
                    val_dataset.set_eval_domain(domain)
                    self.set_eval_domain(domain)
                    outs = self.validation_epoch(nepoch, val_dataset.get_loader())
                    self.validation_epoch_end(outs, domain)
                summed = defaultdict(float)
                for k, v in self.logged_metrics.items():
                    for dn in self.domain_name_list:
                        if dn in k:
                            summed[k.removeprefix(dn + "_")] += v
                            break
                self.logged_metrics.update(summed)
                self.inference_time += time.time() - tik

                self.training_epoch_end(training_output_list)
                if self.callback(self, nepoch, self.logged_metrics):
                    break
                nepoch += 1
            self.training_end()
            self.callback.save_checkpoint(nepoch)
            self.ckpt_path = self.callback.get_checkpoint_path()
        except KeyboardInterrupt:
            self.callback.save_checkpoint(nepoch)
            self.ckpt_path = self.callback.get_checkpoint_path()

    def current_epoch_trainloaders(self, nepoch):
        return self.dataset_list[0].get_loader()

    def train_start(self):
        pass

    def training_end(self):
        pass

    def _train_plan(self, fields, rows):
        raise NotImplementedError

    def _fast_path_ok(self) -> bool:
        return (isinstance(getattr(self, "loss_fn", None), BinaryCrossEntropyLoss)
                and not os.environ.get("DR4SR_NO_FAST_PATH"))

    # ------------------------------------------------------------------------------------------ fast path (fused HIP graph per batch)
    _supports_perm_sel = False

    def _dp_in_graph(self) -> bool:
        """Data parallel: capture the RCCL all-reduce INSIDE the k-step graph?  Opt-in (`train.dp_graph_allreduce: true` or
        DR4SR_DP_GRAPH_ALLREDUCE=1): the default is the host-launched collective between two graphs, the form every multi-rank test
        covers — the in-graph form has only ever run with one RCCL rank (a 1-GPU box cannot host two), so it is not the default
        until a multi-GPU run has covered it.  DR4SR_DP_HOST_ALLREDUCE forces the host form."""
        if os.environ.get("DR4SR_DP_HOST_ALLREDUCE"):
            return False
        on = bool(self.config["train"].get("dp_graph_allreduce", False)) or bool(os.environ.get("DR4SR_DP_GRAPH_ALLREDUCE"))
        return on and parallel.can_capture()

    def _step_graph(self, fields, bl, perm_sel=None, group=1, loss_log=None, in_graph=False, dp_rows=None):
        """captured HIP graph(s) for a local batch of `bl` rows addressed through self._rows_buf[:bl]; with perm_sel =
        (perm, global batch, rank offset, counter) the rows are selected on the device by the step's first kernel, which makes
        consecutive steps host-free: `group` whole steps go into ONE graph and each writes its mean loss to loss_log[batch index].
        in_graph (data parallel only): the caller decided — from GLOBAL quantities, so every rank decides alike — that this step's
        collective is captured inside the graph.  dp_rows (data parallel only): the rows of a FULL per-rank slice of this global
        batch, the same number on every rank — the gradient's bucket count is decided from it, not from this rank's own slice, because
        ranks that disagreed on the number of collectives of a step would deadlock the communicator; None = one flat all-reduce."""
        key = (fields["in_item_id"].data_ptr(), bl, group, None if loss_log is None else loss_log.data_ptr(),
               None if perm_sel is None else (perm_sel[0].data_ptr(), perm_sel[1], perm_sel[2]), bool(in_graph), dp_rows)
        if key in self._graphs:
            return self._graphs[key]
        eng = self.engine
        plan = self._train_plan(fields, self._rows_buf[:bl]) if perm_sel is None else \
            self._train_plan(fields, self._rows_buf[:bl], perm_sel=perm_sel, loss_log=loss_log)
        use_graph = bool(self.config["train"].get("hip_graph", True))
        undo = [eng.params, eng.adam_m, eng.adam_v, eng.state] + ([perm_sel[3]] if perm_sel is not None else [])

        def warm_up(fn):
            """run the step once OUTSIDE capture (code-object load, LDS attributes) and undo its side effects"""
            snap = [t.clone() for t in undo]
            fn()
            torch.cuda.synchronize()
            for dst, src in zip(undo, snap):
                dst.copy_(src)

        if self.world_size == 1:
            def eager():
                if group > 1 and hasattr(eng, "train_steps"):
                    eng.train_steps(plan, group)             # one prep launch per graph (the optimizer launches prepare the next step)
                else:
                    for _ in range(group):
                        eng.train_step(plan)
            if use_graph:
                warm_up(eager)
                g = torch.cuda.CUDAGraph()
                with capture(g):
                    eager()
                run = g.replay
            else:
                run = eager
        else:
            # Data parallel: the sum-all-reduce of the flat gradient sits between backward and optimizer.
            #   default      two graphs (backward | optimizer) around a HOST-launched collective, `group` times per call;
            #   in_graph     (opt-in, RCCL) the collective captured inside ONE graph of `group` whole steps — no host work between
            #                backward, collective and optimizer, the optimizer launch of step j preparing step j+1 as on one GPU.
            # Every rank takes the same form for the same global batch: `in_graph` and `group` come from global quantities
            # (_fused_epoch), and a capture that fails on ANY rank sends ALL ranks to the host form (the success flag is reduced
            # with MIN before the form is chosen — mixing in-graph and host-launched collectives would deadlock the communicator).
            # Gradient buckets (parallel.dp_backward): at scale the table gradient is final one launch before the rest and its all-reduce
            # runs beside that launch; the latency forms have one bucket = the flat all-reduce.
            has_prep = perm_sel is not None and hasattr(eng, "fwd_bwd_prepared")      # the optimizer launch of step j prepares step j + 1 (any batch size)
            # Two buckets only INSIDE a captured graph (where the asynchronous table collective is a parallel branch for free): launched from
            # the host, the two-bucket step is four submissions per step instead of two and measured +91 us at 16 384 rows per rank with one
            # RCCL rank, against +30 us in the graph (profiles/round5_bench_default.json strong[].dp_1rank_rccl) — the host form stays flat.
            buckets = parallel.grad_buckets(eng, dp_rows if in_graph else None, fields.get("seqlen"), want=self.config["train"].get("dp_buckets"))
            flat = parallel.grad_buckets(eng, None)

            def body(reduce):
                for j in range(group):
                    parallel.dp_backward(eng, plan, has_prep and j > 0, buckets, reduce=reduce)
                    if has_prep and j < group - 1:
                        eng.adam_step_prepare_next(plan)
                    else:
                        eng.adam_step(plan)

            run = None
            if use_graph:
                # warm-up must NOT enter a collective: ranks create their graphs at different steps (a tail batch gives some ranks a
                # new slice size, others an old or empty one)
                warm_up(lambda: body(False))
                if in_graph:
                    ok, g = 1, None
                    try:
                        g = torch.cuda.CUDAGraph()
                        with capture(g):
                            body(True)
                    except Exception as e:                # noqa: BLE001 — any capture failure: keep training with the split form
                        self.logger.warning(f"in-graph all-reduce capture failed ({type(e).__name__}: {e}); using host-launched collectives")
                        ok, g = 0, None
                    if parallel.all_ok(bool(ok)):                   # control plane; every rank reaches this point for the same global batch
                        run = g.replay
                    elif ok:
                        self.logger.warning("in-graph all-reduce capture failed on another rank; using host-launched collectives")
                if run is None:
                    # Host-launched collective between graphs, one flat bucket.  The graph that holds the optimizer of step j also holds the
                    # backward of step j + 1, so a step costs ONE graph launch + one collective (round 4: two graphs + one collective per
                    # step; the extra launch was ~8 us of idle GPU per step at B = 256):  [fwd_bwd] AR ([adam+prep | fwd_bwd_prepared] AR)* [adam]
                    buckets = flat

                    def graph_of(fn):
                        g = torch.cuda.CUDAGraph()
                        with capture(g):
                            fn()
                        return g
                    prep = has_prep and group > 1
                    g_first = graph_of(lambda: eng.fwd_bwd(plan))
                    g_mid = None if group == 1 else \
                        graph_of(lambda: (eng.adam_step_prepare_next(plan), eng.fwd_bwd_prepared(plan))) if prep else \
                        graph_of(lambda: (eng.adam_step(plan), eng.fwd_bwd(plan)))
                    g_last = graph_of(lambda: eng.adam_step(plan))

                    def run():
                        for j in range(group):
                            (g_mid if j > 0 else g_first).replay()
                            allreduce_flat(eng.grads)             # RCCL sum: gradients + {n_valid, loss_sum, poison} tail
                        g_last.replay()
            else:
                buckets = flat

                def run():
                    body(True)
        self._graphs[key] = (run, plan)
        return self._graphs[key]

    def _fused_epoch(self, loader):
        eng, W, r = self.engine, self.world_size, self.rank
        B, n, nb = loader.batch_size, loader.n, len(loader)
        perm = loader.permutation()
        if W > 1:
            parallel.broadcast(perm, src=0)
        if getattr(self, "_loss_log", None) is None or self._loss_log.shape[0] != nb:
            self._loss_log = torch.empty(nb, dtype=torch.float32, device=self.device)     # persistent: graphs hold its address
        losses = self._loss_log
        tail = eng.grads[eng.n_params:eng.n_params + 2]
        fused_sel = self._supports_perm_sel
        if fused_sel:                                       # a1 on the device: one permutation upload per EPOCH, no per-step copy
            if getattr(self, "_perm_buf", None) is None or self._perm_buf.shape[0] != n:
                self._perm_buf = torch.empty(n, dtype=torch.int64, device=self.device)
                self._perm_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._perm_buf.copy_(perm)
            self._perm_counter.zero_()
        group = int(self.config["train"].get("steps_per_graph", 16)) if fused_sel else 1      # DP: k steps AND their k collectives per graph
        i = 0
        while i < nb:
            lo, hi = shard_bounds(i, B, n, W, r)
            bl = hi - lo
            if bl > 0 and fused_sel:
                if i == nb - 1 and W > 1:
                    self._perm_counter.fill_(i)             # a rank whose earlier tail slice was empty re-aligns its batch index
                sel = (self._perm_buf, B, lo - i * B, self._perm_counter)
                # k and the collective's form are functions of GLOBAL quantities only, so that every rank replays the same shape for
                # the same global batches: k = group while the next `group` global batches are all full (every rank's slice size is
                # then constant over them), else 1; a partial tail batch — where some ranks' slices are short or empty and take the
                # host all-reduce below — always uses the host-launched collective
                k = group if (i + group) * B <= n else 1
                full = (i + 1) * B <= n
                # (the in-graph form also needs every rank to HAVE rows — a rank with an empty slice of a full batch builds no graph and would
                #  miss the capture-success collective; then every rank takes the host form, as on partial tail batches)
                every_rank_has_rows = (B + W - 1) // W * (W - 1) < B
                run, _ = self._step_graph(loader.fields, bl, sel, group=k, loss_log=losses,      # k_adam logs the (all-reduced) mean loss
                                          in_graph=W > 1 and full and every_rank_has_rows and self._dp_in_graph(),
                                          dp_rows=(B + W - 1) // W if (W > 1 and full) else None)
                run()
                i += k
                continue
            if bl > 0:
                self._rows_buf[:bl].copy_(perm[lo:hi])
                run, _ = self._step_graph(loader.fields, bl)
                run()
            else:                                          # fewer rows than ranks: contribute zeros to the same collectives as the others
                parallel.dp_reduce_empty(eng, None)      # the ranks with rows take the host-launched flat form here (see every_rank_has_rows above)
                eng.adam_step(self._api_plan())
            losses[i] = tail[1] / tail[0]
            i += 1
        return [{"loss_0": losses.clone()}]

    def training_epoch(self, nepoch):
        loader = self.current_epoch_trainloaders(nepoch)
        if self._fast_path_ok():
            return [self._fused_epoch(loader)]
        outputs = []
        if self.world_size > 1:
            return [self._api_epoch_dp(loader)]
        if self._api_graph_ok():
            fields = getattr(loader, "fields", None)
            if fields is not None and hasattr(loader, "permutation"):
                # the loader's own batches are 7 gathers + 3 copies into the graph's static tensors per step; gather the fields the
                # step reads straight into the static tensors instead (same permutation, same batches)
                perm, bs = loader.permutation(), loader.batch_size
                return [[{"loss_0": self._api_step_graph(None, fields, perm[i:i + bs])} for i in range(0, loader.n, bs)]]
            return [[{"loss_0": self._api_step_graph(batch)} for batch in loader]]
        for batch in loader:                                        # API path (reference loop, basemodel.py:192-200)
            batch["neg_item"] = self._neg_sampling(batch)
            self.optimizer.zero_grad()
            loss = self.training_step(batch=batch)
            loss.backward()
            self.optimizer.step()
            outputs.append({"loss_0": loss.detach()})
        return [outputs]

    def _api_epoch_dp(self, loader):
        """The reference loop (basemodel.py:192-200) through the model API — DR4SR_NO_FAST_PATH, loss_fn 'bpr' — under data parallelism
        (round 6; it used to raise).  Every rank walks the global batches of ONE permutation (rank 0's, broadcast) and takes its
        shard_bounds slice; the global objective is the mean over ALL ranks' valid positions, so a rank scales its own mean loss by
        n_valid_local / n_valid_global before backward (the counts travel with the loss sums in one float64 all-reduce) and the flat gradient
        is SUM-reduced; the dense optimizer step then runs identically on every replica.  A rank whose slice of a short tail batch is empty
        contributes zeros to the same two collectives.  Eager (no graph): the path is host-bound by design."""
        eng, W, r = self.engine, self.world_size, self.rank
        fields = getattr(loader, "fields", None)
        if fields is None or not hasattr(loader, "permutation"):
            raise NotImplementedError("data parallelism needs the device-resident loader (fields + permutation)")
        B, n, nb = loader.batch_size, loader.n, len(loader)
        perm = loader.permutation()
        parallel.broadcast(perm, src=0)
        perm = perm.to(self.device)
        outputs = []
        cnt = torch.zeros(2, dtype=torch.float64, device=self.device)
        base = getattr(self, "_neg_calls", 0)
        for i in range(nb):
            lo, hi = parallel.shard_bounds(i, B, n, W, r)
            self.optimizer.zero_grad()
            cnt.zero_()
            loss = None
            if hi > lo:
                rows = perm[lo:hi]
                batch = {k: v.index_select(0, rows) for k, v in fields.items()}
                self._neg_calls = base + i * W + r             # one sampler stream per (global batch, rank)
                batch["neg_item"] = self._neg_sampling(batch)
                n_loc = (batch[self.fiid] != 0).sum().to(torch.float64)
                if int(n_loc) > 0:                               # (a slice without a valid target has no mean to take)
                    loss = self.training_step(batch=batch)       # mean over THIS rank's valid positions
                    cnt[0] = n_loc
                    cnt[1] = loss.detach().to(torch.float64) * n_loc
            parallel.allreduce_flat(cnt)
            if loss is not None:
                (loss * (n_loc / cnt[0]).to(torch.float32)).backward()
            parallel.allreduce_flat(eng.grads)
            self.optimizer.step()
            outputs.append({"loss_0": (cnt[1] / cnt[0]).to(torch.float32)})
        self._neg_calls = base + nb * W
        return outputs

    # ---- API path under a HIP graph: models whose step is a composition of C-ABI calls behind autograd (CL4SRec) are host-bound when
    # run eagerly (≈100 launches + autograd bookkeeping per step); the loop body of basemodel.py:192-200 is captured once per batch
    # size over static copies of the batch tensors and replayed.  Host-side call counters (negative sampler, augmentations) move to
    # device words for the duration.
    def _api_graph_ok(self) -> bool:
        return False

    def _api_graph_state(self):
        """tensors a warm-up run must not change"""
        eng = self.engine
        return [eng.params, eng.adam_m, eng.adam_v] + list(getattr(eng, "states", [eng.state]))

    def _api_step_body(self, batch):
        batch["neg_item"] = self._neg_sampling(batch)
        self.optimizer.zero_grad()
        loss = self.training_step(batch=batch)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _api_step_graph(self, batch, fields=None, rows=None):
        """one captured step on `batch`, or (batch None) on rows `rows` of the device-resident dataset tensors `fields`"""
        if batch is None:
            keep = self._api_graph_fields()
            bl = int(rows.shape[0])
            ent = getattr(self, "_api_graphs", {}).get(bl)
            if ent is None:                                # first step of this batch size: build the graph from a materialised batch
                return self._api_step_graph({k: v.index_select(0, rows) for k, v in fields.items() if keep is None or k in keep or k == self.fuid})
            g, static, out = ent
            for k, v in static.items():
                if k in fields:
                    torch.index_select(fields[k], 0, rows, out=v)
            g.replay()
            return out.clone()
        if not hasattr(self, "_api_graphs"):
            self._api_graphs = {}
            self._neg_step_dev = torch.full((1,), getattr(self, "_neg_calls", 0), dtype=torch.int32, device=self.device)
            self._api_graph_begin()
        bl = int(batch[self.fuid].shape[0])
        ent = self._api_graphs.get(bl)
        if ent is None:
            keep = self._api_graph_fields()
            static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v) and (keep is None or k in keep)}
            undo = self._api_graph_state() + self._api_graph_counters()
            snap = [t.clone() for t in undo]
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                        # warm-up outside capture (code objects, allocator pools)
                    self._api_step_body(dict(static))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for dst, src in zip(undo, snap):
                dst.copy_(src)
            g = torch.cuda.CUDAGraph()
            with capture(g):
                out = self._api_step_body(dict(static))
            ent = self._api_graphs[bl] = (g, static, out)
        g, static, out = ent
        for k, v in static.items():
            if k in batch:
                v.copy_(batch[k])
        g.replay()
        return out.clone()

    def _api_graph_begin(self):
        pass

    def _api_graph_fields(self):
        """batch fields the step reads (None = all): each one is a device copy per replay"""
        return None

    def _api_graph_counters(self):
        return [self._neg_step_dev]

    def training_epoch_end(self, output_list):
        output_list = output_list if isinstance(output_list, list) else [output_list]
        for outputs in output_list:
            if isinstance(outputs, list):
                metric = {"train_" + k: torch.hstack([e[k] for e in outputs]).mean() for k in outputs[0]}
            elif isinstance(outputs, torch.Tensor):
                metric = {"train_loss": outputs.item()}
            else:
                metric = {"train_" + k: v for k, v in outputs.items()}
            self.logged_metrics.update(metric)
        self.logger.info(self.logged_metrics)
        self.logger.info(f"training_time: {self.training_time}")
        self.logger.info(f"inference_time: {self.inference_time}")

    # ------------------------------------------------------------------------------------------ eval
    @torch.no_grad()
    def validation_epoch(self, nepoch, dataloader):
        return [self.validation_step(batch) for batch in dataloader]

    @torch.no_grad()
    def test_epoch(self, dataloader):
        return [self.test_step(batch) for batch in dataloader]

    def _epoch_end(self, outputs, metric_names):
        metric_list, bs = zip(*outputs)
        bs = torch.tensor(bs, dtype=torch.float32)
        out = {}
        for k in metric_list[0]:
            vals = torch.stack([m[k].float().cpu() for m in metric_list])
            out[k] = float((vals * bs).sum() / bs.sum())           # batch-size-weighted mean (basemodel.py:316-324)
        return out

    def validation_epoch_end(self, outputs, domain):
        names = evaluation.get_eval_metrics(self.config["eval"]["val_metrics"], self.config["eval"]["cutoff"], validation=True)
        out = {domain + "_" + k: v for k, v in self._epoch_end(outputs, names).items()}
        self.logged_metrics.update(out)
        return out

    def test_epoch_end(self, outputs, domain):
        names = evaluation.get_eval_metrics(self.config["eval"]["test_metrics"], self.config["eval"]["cutoff"], validation=False)
        out = {domain + "_" + k: v for k, v in self._epoch_end(outputs, names).items()}
        self.logged_metrics.update(out)
        return out

    def validation_step(self, batch):
        return self._test_step(batch, self.config["eval"]["val_metrics"], [self.config["eval"]["cutoff"][0]])

    def test_step(self, batch):
        return self._test_step(batch, self.config["eval"]["test_metrics"], self.config["eval"]["cutoff"])

    def _test_step(self, batch, metric, cutoffs):
        rank_m = evaluation.get_rank_metrics(metric)
        assert len(rank_m) > 0
        bs = batch["user_id"].size(0)
        _, topk_items = self.topk(batch, self.config["eval"]["topk"], batch["user_hist"])
        label = batch[self.fiid].view(-1, 1) == topk_items
        pos_rating = batch["label"].view(-1, 1)
        return {f"{name}@{c}": fn(label, pos_rating, c) for c in cutoffs for name, fn in rank_m}, bs

    def topk(self, batch, k, user_h=None):
        """full-item scores with PAD + history masked, top-k — basemodel.py:354-365 via dr4sr_full_score_topk"""
        blocked = self._domain_blocked(self.eval_domain)        # basemodel.py:358-360 domain_mask (None: every item 1..N-1 is in the domain)
        query = self.forward(batch).contiguous()
        B = query.shape[0]
        hist = user_h.contiguous() if user_h is not None else None
        score = torch.empty(B, k, dtype=torch.float32, device=query.device)
        ids = torch.empty(B, k, dtype=torch.int64, device=query.device)
        eng = self.engine
        need = int(eng.lib.dr4sr_full_score_topk_workspace_bytes(B, self.num_items))      # [B, N] scores: one MFMA GEMM + radix select
        ws = getattr(self, "_topk_ws", None)
        if ws is None or ws.numel() * 4 < need:
            ws = self._topk_ws = torch.empty(need // 4, dtype=torch.float32, device=query.device)
        _lib.check(eng.lib.dr4sr_full_score_topk_masked_ws(_lib.ptr(query), _lib.ptr(self.item_embedding.weight), _lib.ptr(hist),
                                                           _lib.ptr(blocked), _lib.ptr(score), _lib.ptr(ids), B, eng.D, self.num_items,
                                                           hist.shape[1] if hist is not None else 0, k, _lib.ptr(ws), ws.numel() * 4,
                                                           _lib.cur_stream()), "topk")
        return score, ids

    def _domain_blocked(self, domain):
        """uint8 [num_items], 1 = item outside `domain` (device tensor, cached per domain); None when the domain holds every item"""
        cache = self.__dict__.setdefault("_blocked_cache", {})
        if domain not in cache:
            items = torch.as_tensor(self.domain_item_mapping[domain], dtype=torch.int64)
            items = items[(items > 0) & (items < self.num_items)].unique()
            if int(items.numel()) == self.num_items - 1:
                cache[domain] = None
            else:
                m = torch.ones(self.num_items, dtype=torch.uint8)
                m[items] = 0
                cache[domain] = m.to(self.device)
        return cache[domain]

    def set_eval_domain(self, domain):
        self.eval_domain = domain

    def evaluate(self) -> Dict:
        test_data = self.dataset_list[-1]
        output = defaultdict(float)
        if self.world_size > 1:                            # rank 0 writes the checkpoint at the end of fit(): wait for the file
            parallel.barrier()
        self.load_checkpoint(os.path.join(self.config["eval"]["save_path"], self.ckpt_path))
        self.eval()
        for domain in self.domain_name_list:
            test_data.set_eval_domain(domain)
            self.set_eval_domain(domain)
            output.update(self.test_epoch_end(self.test_epoch(test_data.get_loader()), domain))
        summed = defaultdict(float)
        for k, v in output.items():
            for dn in self.domain_name_list:
                if dn in k:
                    summed[k.removeprefix(dn + "_")] += v
        output.update(summed)
        self.logger.info(dict(output))
        self.logger.info({"training_time": self.training_time, "inference_time": self.inference_time})
        return dict(output)

    def load_checkpoint(self, path: str) -> None:
        ckpt = torch.load(path, weights_only=False, map_location=self.device)
        self.config = ckpt["config"]
        self.load_state_dict(ckpt["parameters"])
