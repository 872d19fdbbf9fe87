"""CL4SRec with the class surface of the reference's model/cl4srec.py (:15-73): SASRec + a contrastive term between two augmented
views of every sequence (module/data_augmentation.py:577-619).  The item table has one extra row (the mask item, id = num_items);
the three encoder passes of a step run in three engine slots so that each backward finds its own saved activations."""
from __future__ import annotations

from ..module import data_augmentation
from .sasrec import SASRec


class CL4SRec(SASRec):
    def _table_rows(self) -> int:
        return self.num_items + 1              # cl4srec.py:31-33: one more item for the mask augmentation

    def _n_slots(self) -> int:
        return 3                               # main pass + two views

    def _init_model(self, train_data):
        super()._init_model(train_data)
        self.augmentation_model = data_augmentation.CL4SRecAugmentation(self.config["model"], train_data,
                                                                       seed=int(self.config["train"]["seed"]) + 104729 * self.rank)

    def _fast_path_ok(self) -> bool:
        return False                           # the step is a composition of three encoder passes: API path (autograd over the C ABI)

    def training_step(self, batch, reduce=True, return_query=False, align=False):
        rst = super().training_step(batch, reduce=reduce, return_query=return_query)
        cl_output = self.augmentation_model(batch, self.query_encoder, reduce=reduce)
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
