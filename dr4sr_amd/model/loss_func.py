"""Loss objects with the call signature of the reference's model/loss_func.py.

BinaryCrossEntropyLoss is consumed by BaseModel.training_step through the fused HIP scorer (dr4sr_score_bce_*),
which implements the masked branch of loss_func.py:9-38; the module form below exists for API parity
(`model.loss_fn`) and simply dispatches to that kernel when handed scores computed elsewhere is NOT supported —
scores never leave the kernel in the hot path.  BPRLoss mirrors loss_func.py:40-48 including its missing
`reduce` kwarg (the reference's training_step cannot call it either; selecting it raises the same TypeError).
"""
import torch.nn as nn


class BinaryCrossEntropyLoss(nn.Module):
    name = "bce"

    def forward(self, pos_score, neg_score, reduce=True):
        raise RuntimeError("BinaryCrossEntropyLoss is fused into the HIP scorer (BaseModel.training_step); "
                           "dr4sr_amd has no eager PyTorch loss path")


class BPRLoss(nn.Module):
    name = "bpr"

    def forward(self, pos_score, neg_score):
        raise RuntimeError("BPRLoss: no HIP kernel yet (the reference's own training_step cannot call it: "
                           "model/basemodel.py:210 passes reduce=, model/loss_func.py:44 does not accept it)")
