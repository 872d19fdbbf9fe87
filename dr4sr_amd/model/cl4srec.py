"""CL4SRec with the class surface of the reference's model/cl4srec.py (:15-73): SASRec + a contrastive term between two augmented
views of every sequence (module/data_augmentation.py:577-619).  The item table has one extra row (the mask item, id = num_items);
the three encoder passes of a step run in three engine slots so that each backward finds its own saved activations."""
from __future__ import annotations

import torch

from ..module import data_augmentation
from .sasrec import SASRec


class CL4SRec(SASRec):
    def _table_rows(self) -> int:
        return self.num_items + 1              # cl4srec.py:31-33: one more item for the mask augmentation

    def _n_slots(self) -> int:
        return 3                               # main pass + two views

    def _max_batch(self, config) -> int:
        return max(super()._max_batch(config), 2 * int(config["train"]["batch_size"]))     # both views as one batch of 2B sequences

    def _init_model(self, train_data):
        super()._init_model(train_data)
        self.augmentation_model = data_augmentation.CL4SRecAugmentation(self.config["model"], train_data,
                                                                       seed=int(self.config["train"]["seed"]) + 104729 * self.rank)

    def _fast_path_ok(self) -> bool:
        return False                           # the step is a composition of three encoder passes: API path (autograd over the C ABI)

    # the step is ≈100 small launches behind autograd: replayed as one HIP graph per batch size (BaseModel._api_step_graph)
    def _api_graph_ok(self) -> bool:
        return bool(self.config["train"].get("hip_graph", True)) and self.world_size == 1

    def _api_graph_begin(self):
        aug = self.augmentation_model.augmentation
        if hasattr(aug, "calls"):
            aug.step_dev = torch.full((1,), aug.calls, dtype=torch.int32, device=self.device)

    def _api_graph_fields(self):
        keep = {"in_" + self.fiid, self.fiid, "seqlen"}
        return keep if self._direct_step_ok() else keep | {self.fuid}

    def _api_graph_counters(self):
        sd = getattr(self.augmentation_model.augmentation, "step_dev", None)
        return super()._api_graph_counters() + ([sd] if sd is not None else [])

    # ---- the step body of the captured graph WITHOUT autograd: the same C-ABI calls in the same order, composed directly.
    # loss = BCE (mean over valid targets) + cl_weight * InfoNCE (mean over kept rows), cl4srec.py:49-73.  The main pass is the
    # fused SASRec step on the batch's negatives (un-normalised gradients + {n_valid, loss_sum} tail, csrc/step.hip); the two views
    # are encoded in slots 1 and 2, InfoNCE's backward is scaled by cl_weight * n_valid / n_rows ON THE DEVICE so that the optimizer's
    # division by n_valid leaves exactly cl_weight * d(mean InfoNCE), and the views' encoder backward passes accumulate into the
    # same flat gradient.  52 launches instead of 71 (no autograd glue kernels, 10-launch main pass, both views from one augmentation launch).
    def _direct_step_ok(self) -> bool:
        import os
        from .loss_func import BinaryCrossEntropyLoss
        return isinstance(self.loss_fn, BinaryCrossEntropyLoss) and not os.environ.get("DR4SR_CL_AUTOGRAD")

    def _api_step_body(self, batch):
        if not self._direct_step_ok():
            return super()._api_step_body(batch)
        from .. import _lib
        eng, lib = self.engine, self.engine.lib
        ids, tgt, lens = batch["in_" + self.fiid], batch[self.fiid], batch["seqlen"]
        batch["neg_item"] = self._neg_sampling(batch)
        n = eng.n_params
        # main pass: prep zeroes the flat gradient, fused forward + scorer + backward
        eng.fwd_bwd(eng.make_plan(ids, tgt, lens, neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False))
        stats = self._cl_term(ids, lens)
        sc = torch.empty(1, dtype=torch.float32, device=self.device)               # the reported loss
        _lib.check(lib.dr4sr_cl_scalars(_lib.ptr(eng.grads[n:n + 2]), _lib.ptr(stats), float(self.config["model"]["cl_weight"]), None,
                                        _lib.ptr(sc), _lib.cur_stream()), "dr4sr_cl_scalars")
        eng.adam_step(self._api_plan())
        return sc[0]

    def _cl_term(self, ids, lens, views=None, dp_counts=None, fold_loss=False, rows=None):
        """cl_weight * mean InfoNCE between two augmented views of the rows (cl4srec.py:52-55, data_augmentation.py:595-619), composed
        from C-ABI calls without autograd: its gradient is ACCUMULATED into engine.grads scaled by cl_weight * n_valid / rows, so that
        an optimizer (or a hyper-gradient probe) dividing the flat gradient by the tail's n_valid is left with exactly
        cl_weight * d(mean InfoNCE).  The main pass must have left {n_valid, loss_sum} in the gradient tail.  Returns InfoNCE's
        device stats {kept rows, sum of row losses}.

        views: ((seq_i, len_i), (seq_j, len_j)) to encode instead of drawing new ones (MetaModel's hyper-gradient probes evaluate one
        draw at several parameter points; tests).
        dp_counts: data parallel — rows held by each rank for this global batch (parallel.shard_bounds).  InfoNCE's negatives are the
        other rows of the BATCH ('batch_both'), so every rank all-gathers the pooled views (+ its n_valid, + the length-1 mask),
        evaluates the loss of the GLOBAL batch and back-propagates the rows it owns: the sum over ranks of the local gradients is
        the single-process gradient of the concatenated batch (tools/dp_cl_check.py; tests/test_host_cpu.py restates the scheme on
        the oracle).  A rank with no rows still takes part in the gather.
        fold_loss: also add the term's share cl_weight * mean InfoNCE * n_valid(local) to the tail's loss_sum (MetaModel / DP log the
        step's loss as tail[1] / tail[0]).
        rows (int64 [B]): ids / lens are DATASET tensors and the batch is their rows rows[0..B) — the fused epoch's form, where the
        main pass's first kernel selected the rows on the device and no batch tensor exists."""
        import os
        from .. import _lib, parallel
        eng, lib = self.engine, self.engine.lib
        am = self.augmentation_model
        aug = am.augmentation
        B, D, dev, n = int(rows.shape[0] if rows is not None else ids.shape[0]), eng.D, self.device, eng.n_params
        st = _lib.cur_stream
        clw, temp = float(self.config["model"]["cl_weight"]), float(am.InfoNCE_loss_fn.temperature)
        tail = eng.grads[n:n + 2]
        plans = ()
        step_in_prepare = False
        if B > 0:
            if views is None:
                if hasattr(aug, "begin_step"):
                    aug.begin_step()
                if rows is not None:
                    (aug_i, len_i), (aug_j, len_j) = aug.two_views(ids, lens, rows=rows)
                elif hasattr(aug, "two_views"):
                    (aug_i, len_i), (aug_j, len_j) = aug.two_views(ids, lens)
                else:
                    (aug_i, len_i), (aug_j, len_j) = aug(ids, lens), aug(ids, lens)
            else:
                (aug_i, len_i), (aug_j, len_j) = views
            # The views are independent sequences through one encoder: by default they run as ONE batch of 2B sequences (the halves of
            # two_views()' tensors are contiguous) — at these sizes a pass costs its launch chain, not its tokens, so one pass of 2B is
            # ~1.15x a pass of B instead of 2x.  DR4SR_CL_TWO_PASS keeps one pass per view in slots 1 and 2: the dropout streams of the
            # autograd body (the batched pass draws independent masks too, from one stream keyed by the row index in 2B).
            def halves(a, b):                          # b is the second half of the tensor a starts (two_views), not merely its neighbour in memory
                return (a.is_contiguous() and b.is_contiguous() and a.data_ptr() + a.numel() * 8 == b.data_ptr()
                        and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())
            batched = 2 * B <= eng.max_batch and halves(aug_i, aug_j) and halves(len_i, len_j) and not os.environ.get("DR4SR_CL_TWO_PASS")
            # regime hint of the views' plans: crops are shorter than the rows they come from (engine.make_plan: expected_tokens)
            fac = aug.expected_len_factor() if hasattr(aug, "expected_len_factor") else 1.0
            exp_v = max(1, int(B * eng.mean_len * fac)) if eng.mean_len is not None else None
            if batched:
                ids2 = torch.as_strided(aug_i, (2 * B, aug_i.shape[1]), aug_i.stride())
                len2 = torch.as_strided(len_i, (2 * B,), len_i.stride())
                plans = (eng.make_plan(ids2, None, len2, slot=1, expected_tokens=None if exp_v is None else 2 * exp_v),)
                q = eng.encode(plans[0], True, _lib.POOL_MEAN)
                q_i, q_j = q[:B], q[B:]
            else:
                plans = (eng.make_plan(aug_i.contiguous(), None, len_i.contiguous(), slot=1, expected_tokens=exp_v),
                         eng.make_plan(aug_j.contiguous(), None, len_j.contiguous(), slot=2, expected_tokens=exp_v))
                q_i, q_j = eng.encode(plans[0], True, _lib.POOL_MEAN), eng.encode(plans[1], True, _lib.POOL_MEAN)
            # rows-indirected (captured) step: the device call counter is advanced by the dr4sr_cl_prepare_rows_step launch below
            step_in_prepare = views is None and rows is not None and dp_counts is None and getattr(aug, "step_dev", None) is not None
            if views is None and hasattr(aug, "end_step") and not step_in_prepare:
                aug.end_step()
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        sc = torch.empty(1, dtype=torch.float32, device=dev)               # InfoNCE backward scale
        if B == 0 and dp_counts is None:                                   # an empty batch outside data parallelism: no rows, no term
            return stats.zero_()
        if dp_counts is None:
            valid = torch.empty(B, dtype=torch.uint8, device=dev)
            lse, loss_row = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
            dq = torch.empty(2, B, D, dtype=torch.float32, device=dev)
            if rows is not None and views is None and step_in_prepare:
                _lib.check(lib.dr4sr_cl_prepare_rows_step(_lib.ptr(lens), _lib.ptr(rows), B, _lib.ptr(valid), _lib.ptr(stats), _lib.ptr(dq),
                                                          dq.numel(), _lib.ptr(aug.step_dev), int(aug._in_step), st()), "dr4sr_cl_prepare_rows_step")
            elif rows is not None:
                _lib.check(lib.dr4sr_cl_prepare_rows(_lib.ptr(lens), _lib.ptr(rows), B, _lib.ptr(valid), _lib.ptr(stats), _lib.ptr(dq), dq.numel(),
                                                     st()), "dr4sr_cl_prepare_rows")
            else:
                _lib.check(lib.dr4sr_cl_prepare(_lib.ptr(lens.contiguous()), B, _lib.ptr(valid), _lib.ptr(stats), _lib.ptr(dq), dq.numel(), st()),
                           "dr4sr_cl_prepare")
            _lib.check(lib.dr4sr_infonce_fwd(_lib.ptr(q_i), _lib.ptr(q_j), _lib.ptr(valid), B, D, temp, _lib.ptr(lse), _lib.ptr(loss_row),
                                             _lib.ptr(stats), st()), "dr4sr_infonce_fwd")
            # (the step's two device scalars inside the backward launch: dr4sr_cl_scalars_dp + dr4sr_infonce_bwd were two)
            _lib.check(lib.dr4sr_infonce_bwd_scaled(_lib.ptr(q_i), _lib.ptr(q_j), _lib.ptr(valid), B, D, temp, _lib.ptr(lse), _lib.ptr(tail),
                                                    _lib.ptr(stats), clw, 1 if fold_loss else 0, _lib.ptr(dq[0]), _lib.ptr(dq[1]), st()),
                       "dr4sr_infonce_bwd_scaled")
            dq_i, dq_j = dq[0], dq[1]
        else:
            # ---- the global batch: [W][Bmax rows of (q_i | q_j | kept) ... | n_valid]
            assert rows is None, "data parallel: the step runs on materialised per-rank batches"
            W, r = len(dp_counts), self.rank
            assert dp_counts[r] == B, (dp_counts, r, B)
            Bmax, Bg, off = max(dp_counts), sum(dp_counts), sum(dp_counts[:r])
            RW = 2 * D + 1
            send = torch.zeros(Bmax * RW + 1, dtype=torch.float32, device=dev)
            if B > 0:
                rows = send[:Bmax * RW].view(Bmax, RW)
                rows[:B, :D].copy_(q_i)
                rows[:B, D:2 * D].copy_(q_j)
                rows[:B, 2 * D].copy_(lens != 1)                           # data_augmentation.py:613-615
            send[Bmax * RW:].copy_(tail[0:1])
            got = parallel.all_gather_flat(send)                            # [W, Bmax * RW + 1]
            parts = [got[k, :dp_counts[k] * RW].view(dp_counts[k], RW) for k in range(W) if dp_counts[k] > 0]
            allr = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
            xi, xj = allr[:, :D].contiguous(), allr[:, D:2 * D].contiguous()
            valid = allr[:, 2 * D].to(torch.uint8).contiguous()
            lse, loss_row = torch.empty(Bg, dtype=torch.float32, device=dev), torch.empty(Bg, dtype=torch.float32, device=dev)
            dq = torch.zeros(2, Bg, D, dtype=torch.float32, device=dev)
            stats.zero_()
            _lib.check(lib.dr4sr_infonce_fwd(_lib.ptr(xi), _lib.ptr(xj), _lib.ptr(valid), Bg, D, temp, _lib.ptr(lse), _lib.ptr(loss_row),
                                             _lib.ptr(stats), st()), "dr4sr_infonce_fwd")
            _lib.check(lib.dr4sr_cl_scalars_dp(_lib.ptr(got[0, Bmax * RW:]), W, Bmax * RW + 1, _lib.ptr(stats), clw, _lib.ptr(sc),
                                               _lib.ptr(tail) if fold_loss else None, st()), "dr4sr_cl_scalars_dp")
            _lib.check(lib.dr4sr_infonce_bwd(_lib.ptr(xi), _lib.ptr(xj), _lib.ptr(valid), Bg, D, temp, _lib.ptr(lse), _lib.ptr(sc),
                                             _lib.ptr(dq[0]), _lib.ptr(dq[1]), st()), "dr4sr_infonce_bwd")
            dq_i, dq_j = dq[0, off:off + B], dq[1, off:off + B]
        if B > 0:
            if len(plans) == 1:
                eng.encode_bwd(plans[0], True, _lib.POOL_MEAN, torch.cat([dq_i, dq_j], 0) if dp_counts is not None else dq.view(2 * B, D))
            else:
                eng.encode_bwd(plans[0], True, _lib.POOL_MEAN, dq_i.contiguous())
                eng.encode_bwd(plans[1], True, _lib.POOL_MEAN, dq_j.contiguous())
        return stats

    # ---- single GPU, fused epoch (round 4): NO per-step host work.  The step the graph replays selects its batch on the device — the
    # main pass's first kernel fills rows[] from the epoch permutation (dr4sr_sasrec_plan.perm) and draws the negatives in-kernel, the
    # views are drawn from the dataset tensors through rows[] (dr4sr_cl_augment2_rows_dev), the optimizer launch writes the step's
    # loss (BCE mean + cl_weight * InfoNCE mean: _cl_term folds the term into the tail) into loss_log[batch index] — so k whole steps
    # go into one graph.  The per-batch form (_api_step_graph) copied three gathered fields into static tensors, ran the negative
    # sampler as two launches and cloned the loss per step: ~35 us of a 0.33 ms step.  train.cl_fused_epoch: false restores it.
    def _fused_cl_ok(self) -> bool:
        import os
        return (self.world_size == 1 and self._direct_step_ok() and bool(self.config["train"].get("hip_graph", True))
                and bool(self.config["train"].get("cl_fused_epoch", True)) and not os.environ.get("DR4SR_CL_TWO_PASS")
                and hasattr(self.augmentation_model.augmentation, "two_views"))

    def _cl_rows_step(self, plan, fields, rows):
        """one step on rows[] of the dataset tensors: fused main pass (+ device-side batch selection / negatives when the plan carries
        them), the contrastive term through rows[], optimizer (logs the step's loss when the plan has a loss log)"""
        eng = self.engine
        eng.fwd_bwd(plan)
        self._cl_term(fields["in_" + self.fiid], fields["seqlen"], rows=rows, fold_loss=True)
        eng.adam_step(plan)

    def _fused_cl_graph(self, fields, bl, k):
        aug = self.augmentation_model.augmentation
        if getattr(aug, "step_dev", None) is None:
            self._api_graph_begin()
        # every address the captured graph bakes in is part of the key (_rows_buf / _neg_buf are re-allocated when the batch
        # size grows, and a cached graph would replay against freed buffers)
        key = ("cl_rows", fields["in_" + self.fiid].data_ptr(), bl, k, self._loss_log.data_ptr(), self._perm_buf.data_ptr(),
               self._perm_counter.data_ptr(), self._rows_buf.data_ptr(), self._neg_buf.data_ptr(), aug.step_dev.data_ptr())
        if key in self._graphs:
            return self._graphs[key]
        from ..utils.graphs import capture
        eng = self.engine
        rows = self._rows_buf[:bl]
        plan = eng.make_plan(fields["in_" + self.fiid], fields[self.fiid], fields["seqlen"], rows=rows, neg_item=self._neg_buf, sample_neg=True,
                             perm_sel=(self._perm_buf, int(self.config["train"]["batch_size"]), 0, self._perm_counter), loss_log=self._loss_log)

        def body():
            for _ in range(k):
                self._cl_rows_step(plan, fields, rows)
        undo = [eng.params, eng.adam_m, eng.adam_v, self._perm_counter, aug.step_dev] + list(eng.states)
        snap = [t.clone() for t in undo]
        body()                                             # warm-up outside capture, side effects undone
        torch.cuda.synchronize()
        for dst, src in zip(undo, snap):
            dst.copy_(src)
        g = torch.cuda.CUDAGraph()
        with capture(g):
            body()
        self._graphs[key] = (g.replay, plan)
        return self._graphs[key]

    def _fused_cl_epoch(self, loader):
        B, n, nb = loader.batch_size, loader.n, len(loader)
        dev = self.device
        if getattr(self, "_perm_buf", None) is None or self._perm_buf.shape[0] != n:
            self._perm_buf = torch.empty(n, dtype=torch.int64, device=dev)
            self._perm_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        if getattr(self, "_loss_log", None) is None or self._loss_log.shape[0] != nb:
            self._loss_log = torch.empty(nb, dtype=torch.float32, device=dev)
        if getattr(self, "_rows_buf", None) is None or self._rows_buf.shape[0] < B:
            self._rows_buf = torch.zeros(B, dtype=torch.int64, device=dev)
            self._neg_buf = torch.zeros(B * self.max_seq_len, dtype=torch.int64, device=dev)
        self._perm_buf.copy_(loader.permutation())
        self._perm_counter.zero_()
        group = int(self.config["train"].get("steps_per_graph", 16))
        i = 0
        while i < nb:
            bl = min(B, n - i * B)
            k = group if (i + group) * B <= n else 1
            run, _ = self._fused_cl_graph(loader.fields, bl, k)
            run()
            i += k
        return [{"loss_0": self._loss_log.clone()}]

    # ---- data parallel (round 4): the fused main pass + the views' passes on this rank's slice of every global batch, the contrastive
    # term over the gathered global batch (_cl_term), ONE sum-all-reduce of the flat gradient, dense Adam.  Eager launches (two
    # collectives per step sit between them; the single-GPU path replays one graph per batch size).
    def training_epoch(self, nepoch):
        if self.world_size <= 1:
            loader = self.current_epoch_trainloaders(nepoch)
            if self._fused_cl_ok() and getattr(loader, "fields", None) is not None and hasattr(loader, "permutation"):
                return [self._fused_cl_epoch(loader)]
            return super().training_epoch(nepoch)
        from .. import parallel
        from ..parallel import allreduce_flat, shard_bounds
        if not self._direct_step_ok():
            raise NotImplementedError("CL4SRec under data parallelism runs the directly composed step (BCE loss, DR4SR_CL_AUTOGRAD unset)")
        loader = self.current_epoch_trainloaders(nepoch)
        eng, W, r = self.engine, self.world_size, self.rank
        B, n, nb = loader.batch_size, loader.n, len(loader)
        perm = loader.permutation()
        parallel.broadcast(perm, src=0)
        losses = torch.empty(nb, dtype=torch.float32, device=self.device)
        tail = eng.grads[eng.n_params:eng.n_params + 2]
        keep = self._api_graph_fields()
        for i in range(nb):
            bounds = [shard_bounds(i, B, n, W, k) for k in range(W)]
            lo, hi = bounds[r]
            rows = perm[lo:hi]
            batch = {k: v.index_select(0, rows) for k, v in loader.fields.items() if k in keep or k == self.fuid}
            self._dp_step(batch, [b - a for a, b in bounds])
            losses[i] = tail[1] / tail[0]
        return [[{"loss_0": losses}]]

    def _dp_step(self, batch, counts):
        from ..parallel import allreduce_flat
        eng = self.engine
        ids, tgt, lens = batch["in_" + self.fiid], batch[self.fiid], batch["seqlen"]
        if int(ids.shape[0]) > 0:
            if "neg_item" not in batch:
                batch["neg_item"] = self._neg_sampling(batch)
            eng.fwd_bwd(eng.make_plan(ids, tgt, lens, neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False))
        else:
            eng.grads.zero_()
        self._cl_term(ids, lens, views=batch.get("_views"), dp_counts=counts, fold_loss=True)
        allreduce_flat(eng.grads)
        eng.adam_step(self._api_plan())

    def training_step(self, batch, reduce=True, return_query=False, align=False):
        aug = self.augmentation_model.augmentation
        if hasattr(aug, "begin_step"):
            aug.begin_step()
        rst = super().training_step(batch, reduce=reduce, return_query=return_query)
        cl_output = self.augmentation_model(batch, self.query_encoder, reduce=reduce)
        if hasattr(aug, "end_step"):
            aug.end_step()
        cl_loss = self.config["model"]["cl_weight"] * cl_output["cl_loss"]
        if not reduce:
            if return_query:
                loss_value, query = rst
                return (loss_value, cl_loss), query
            return rst, cl_loss
        if return_query:
            loss_value, query = rst
            return loss_value + cl_loss, query
        return rst + cl_loss
