"""GRU4Rec with the class surface of the reference's model/gru4rec.py (GRU4Rec :8-36) and module/layers.py
(VStackLayer :94-102, LambdaLayer :104-115, GRULayer :117-136), carrying parameters under the reference's state-dict names:
  item_embedding.weight == query_encoder.0.1.weight (tied), query_encoder.0.3.gru.weight_{ih,hh}_l{l}, query_encoder.1.{weight,bias}
Every Parameter is a view into GruEngine's flat buffer; arithmetic runs in libdr4sr_hip.so (dr4sr_gru4rec_*)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib
from ..gru_engine import GruEngine
from .basemodel import BaseModel
from .sasrec import _Embedding, _Linear, _bind


class LambdaLayer(nn.Module):
    def __init__(self, lambda_func):
        super().__init__()
        self.lambda_func = lambda_func


class _GRU(nn.Module):
    """parameter holder with torch.nn.GRU's attribute names (weight_ih_l{k}, weight_hh_l{k}; bias=False)"""

    def __init__(self, eng, prefix, n_layer):
        super().__init__()
        H = eng.H
        k = 1.0 / (H ** 0.5)
        for l in range(n_layer):
            for nm in (f"weight_ih_l{l}", f"weight_hh_l{l}"):
                p = _bind(self, nm, eng.views[prefix + nm], eng.grad_views[prefix + nm])
                p.data.uniform_(-k, k)                      # torch.nn.GRU.reset_parameters: U(-1/sqrt(H), 1/sqrt(H))


class GRULayer(nn.Module):
    def __init__(self, eng, prefix, n_layer):
        super().__init__()
        self.gru = _GRU(eng, prefix + "gru.", n_layer)
        self.return_hidden = False


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, anchor, idx, seqlen, training, pooling):
        eng = model.engine
        idx, seqlen = idx.contiguous(), seqlen.contiguous()
        out = eng.encode(eng.make_plan(idx, None, seqlen), training, pooling)
        ctx.model, ctx.args = model, (idx, seqlen, training, pooling)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, seqlen, training, pooling = ctx.args
        eng = ctx.model.engine
        eng.encode_bwd(eng.make_plan(idx, None, seqlen), training, pooling, gout.contiguous())
        return None, None, None, None, None, None


class GRU4Rec(BaseModel):
    def __init__(self, config, dataset_list) -> None:
        super().__init__(config, dataset_list)
        mc, tc = config["model"], config["train"]
        max_b = max(int(tc["batch_size"]), int(config["eval"]["batch_size"]))
        self.engine = GruEngine(self.num_items, self.max_seq_len, self.embed_dim, mc["hidden_size"], mc["layer_num"],
                                mc["dropout_rate"], max_b, self.device, seed=int(tc["seed"]) + 7919 * self.rank,
                                lr=float(tc["learning_rate"]), weight_decay=float(tc["weight_decay"]))
        self.device = self.engine.device
        eng = self.engine
        self.item_embedding = _Embedding(eng, "item_embedding.weight", self.num_items, self.embed_dim, padding_idx=0)
        inner = nn.Sequential(LambdaLayer(lambda x: x["in_" + self.fiid]), self.item_embedding, nn.Dropout(mc["dropout_rate"]),
                              GRULayer(eng, "query_encoder.0.3.", mc["layer_num"]))
        self.query_encoder = nn.Sequential(inner, _Linear(eng, "query_encoder.1.", self.embed_dim, mc["hidden_size"]))
        self._rows_buf = torch.zeros(int(tc["batch_size"]), dtype=torch.int64, device=self.device)
        self._neg_buf = torch.zeros(int(tc["batch_size"]) * self.max_seq_len, dtype=torch.int64, device=self.device)

    def training_epoch_end(self, output_list):
        self.engine.check_coop()                   # fail loudly if the multi-CU recurrence ever timed out during the epoch
        return super().training_epoch_end(output_list)

    def forward(self, batch, need_pooling=True):
        pooling = _lib.POOL_NONE if not need_pooling else (_lib.POOL_ORIGIN if self.training else _lib.POOL_LAST)
        return _Encode.apply(self, self.item_embedding.weight, batch["in_" + self.fiid], batch["seqlen"], bool(self.training), pooling)

    def training_step(self, batch, reduce=True, return_query=False, align=False):
        return super().training_step(batch, reduce, return_query)

    def _encode_raw(self, batch, training=True):
        eng = self.engine
        return eng.encode(eng.make_plan(batch["in_" + self.fiid], None, batch["seqlen"]), training, _lib.POOL_ORIGIN)

    def _encode_bwd_raw(self, batch, d_query, training=True):
        eng = self.engine
        eng.encode_bwd(eng.make_plan(batch["in_" + self.fiid], None, batch["seqlen"]), training, _lib.POOL_ORIGIN, d_query)

    def _batch_plan(self, batch):
        return self.engine.make_plan(batch["in_" + self.fiid], batch[self.fiid], batch["seqlen"],
                                     neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False)

    def _api_plan(self):
        return None

    _supports_perm_sel = True          # round 4: batch selection fused into the step's first kernel (k steps per graph, no per-step rows copy)

    def _train_plan(self, fields, rows, perm_sel=None, loss_log=None):
        return self.engine.make_plan(fields["in_item_id"], fields["item_id"], fields["seqlen"], rows=rows, neg_item=self._neg_buf,
                                     sample_neg=True, perm_sel=perm_sel, loss_log=loss_log)
