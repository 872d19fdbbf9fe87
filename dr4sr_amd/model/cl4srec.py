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
        if self.world_size > 1:
            raise NotImplementedError("CL4SRec trains through the API path, which has no gradient all-reduce: single GPU only "
                                      "(data parallelism covers SASRec / GRU4Rec / FMLP / MetaModel)")
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
        am = self.augmentation_model
        aug = am.augmentation
        ids, tgt, lens = batch["in_" + self.fiid], batch[self.fiid], batch["seqlen"]
        batch["neg_item"] = self._neg_sampling(batch)
        if hasattr(aug, "begin_step"):
            aug.begin_step()
        n = eng.n_params
        # main pass: prep zeroes the flat gradient, fused forward + scorer + backward
        eng.fwd_bwd(eng.make_plan(ids, tgt, lens, neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False))
        # two views through the same encoder (mean pooling fused), one workspace slot each
        if hasattr(aug, "two_views"):
            (aug_i, len_i), (aug_j, len_j) = aug.two_views(ids, lens)
        else:
            (aug_i, len_i), (aug_j, len_j) = aug(ids, lens), aug(ids, lens)
        # The views are independent sequences through one encoder: by default they run as ONE batch of 2B sequences (the halves of
        # two_views()' tensors are contiguous) — at these sizes a pass costs its launch chain, not its tokens, so one pass of 2B is
        # ~1.15x a pass of B instead of 2x.  DR4SR_CL_TWO_PASS keeps one pass per view in slots 1 and 2: the dropout streams of the
        # autograd body (the batched pass draws independent masks too, from one stream keyed by the row index in 2B).
        import os
        B = int(ids.shape[0])
        batched = (hasattr(aug, "two_views") and 2 * B <= eng.max_batch and aug_i.data_ptr() + aug_i.numel() * 8 == aug_j.data_ptr()
                   and not os.environ.get("DR4SR_CL_TWO_PASS"))
        if batched:
            ids2 = torch.as_strided(aug_i, (2 * B, aug_i.shape[1]), aug_i.stride())
            len2 = torch.as_strided(len_i, (2 * B,), len_i.stride())
            plan_v = eng.make_plan(ids2, None, len2, slot=1)
            q = eng.encode(plan_v, True, _lib.POOL_MEAN)
            q_i, q_j = q[:B], q[B:]
        else:
            plan_i, plan_j = eng.make_plan(aug_i, None, len_i, slot=1), eng.make_plan(aug_j, None, len_j, slot=2)
            q_i, q_j = eng.encode(plan_i, True, _lib.POOL_MEAN), eng.encode(plan_j, True, _lib.POOL_MEAN)
        if hasattr(aug, "end_step"):
            aug.end_step()
        B, D = q_i.shape
        dev = self.device
        valid, stats = torch.empty(B, dtype=torch.uint8, device=dev), torch.empty(2, dtype=torch.float32, device=dev)
        lse, loss_row = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
        dq = torch.empty(2, B, D, dtype=torch.float32, device=dev)
        sc = torch.empty(2, dtype=torch.float32, device=dev)               # {InfoNCE backward scale, reported loss}
        st = _lib.cur_stream
        _lib.check(lib.dr4sr_cl_prepare(_lib.ptr(lens.contiguous()), B, _lib.ptr(valid), _lib.ptr(stats), _lib.ptr(dq), dq.numel(), st()),
                   "dr4sr_cl_prepare")
        temp = float(am.InfoNCE_loss_fn.temperature)
        _lib.check(lib.dr4sr_infonce_fwd(_lib.ptr(q_i), _lib.ptr(q_j), _lib.ptr(valid), B, D, temp, _lib.ptr(lse), _lib.ptr(loss_row),
                                         _lib.ptr(stats), st()), "dr4sr_infonce_fwd")
        clw = float(self.config["model"]["cl_weight"])
        tail = eng.grads[n:n + 2]
        _lib.check(lib.dr4sr_cl_scalars(_lib.ptr(tail), _lib.ptr(stats), clw, _lib.ptr(sc[0:1]), None, st()), "dr4sr_cl_scalars")
        _lib.check(lib.dr4sr_infonce_bwd(_lib.ptr(q_i), _lib.ptr(q_j), _lib.ptr(valid), B, D, temp, _lib.ptr(lse), _lib.ptr(sc[0:1]),
                                         _lib.ptr(dq[0]), _lib.ptr(dq[1]), st()), "dr4sr_infonce_bwd")
        if batched:
            eng.encode_bwd(plan_v, True, _lib.POOL_MEAN, dq.view(2 * B, D))
        else:
            eng.encode_bwd(plan_i, True, _lib.POOL_MEAN, dq[0])
            eng.encode_bwd(plan_j, True, _lib.POOL_MEAN, dq[1])
        _lib.check(lib.dr4sr_cl_scalars(_lib.ptr(tail), _lib.ptr(stats), clw, None, _lib.ptr(sc[1:2]), st()), "dr4sr_cl_scalars")
        loss = sc[1]
        eng.adam_step(self._api_plan())
        return loss

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
