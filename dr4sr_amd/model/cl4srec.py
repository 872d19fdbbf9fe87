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
        return {"in_" + self.fiid, self.fiid, "seqlen", self.fuid}

    def _api_graph_counters(self):
        sd = getattr(self.augmentation_model.augmentation, "step_dev", None)
        return super()._api_graph_counters() + ([sd] if sd is not None else [])

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
