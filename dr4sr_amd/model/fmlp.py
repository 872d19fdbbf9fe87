"""FMLP with the class surface of the reference's model/fmlp.py (FMLP :8-51) and module/layers.py:740-807
(FilterLayer, Intermediate, Layer, FMLPEncoder), carrying parameters under the reference's state-dict names:
  item_embedding.weight, position_embeddings.weight, LayerNorm.{weight,bias},
  item_encoder.layer.{i}.filterlayer.{complex_weight, LayerNorm.*}, item_encoder.layer.{i}.intermediate.{dense_1,dense_2,LayerNorm}.*
Every Parameter is a view into FmlpEngine's flat buffer; arithmetic runs in libdr4sr_hip.so (dr4sr_fmlp_*).
As in the reference, sizes are fixed (L = 50, D = 64, hidden 256), dropout is 0.5 regardless of configs/fmlp.yaml, and
forward returns the encoder output at the LAST position in train and eval — the data must be per-prefix, left-padded rows
with scalar targets (dataset/dataset_transform.ipynb of the reference; README.md:78)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..fmlp_engine import FmlpEngine
from .basemodel import BaseModel
from .sasrec import _Embedding, _LayerNorm, _Linear, _bind


class FilterLayer(nn.Module):
    def __init__(self, eng, prefix):
        super().__init__()
        _bind(self, "complex_weight", eng.views[prefix + "complex_weight"], eng.grad_views[prefix + "complex_weight"])
        self.complex_weight.data.copy_(torch.randn(self.complex_weight.shape) * 0.02)       # layers.py:743
        self.out_dropout = nn.Dropout(0.5)
        self.LayerNorm = _LayerNorm(eng, prefix + "LayerNorm.", 64, 1e-12)


class Intermediate(nn.Module):
    def __init__(self, eng, prefix):
        super().__init__()
        self.dense_1 = _Linear(eng, prefix + "dense_1.", 256, 64)
        self.dense_2 = _Linear(eng, prefix + "dense_2.", 64, 256)
        self.LayerNorm = _LayerNorm(eng, prefix + "LayerNorm.", 64, 1e-12)
        self.dropout = nn.Dropout(0.5)


class Layer(nn.Module):
    def __init__(self, eng, prefix):
        super().__init__()
        self.filterlayer = FilterLayer(eng, prefix + "filterlayer.")
        self.intermediate = Intermediate(eng, prefix + "intermediate.")


class FMLPEncoder(nn.Module):
    def __init__(self, eng, num_hidden_layers=2):
        super().__init__()
        self.layer = nn.ModuleList([Layer(eng, f"item_encoder.layer.{i}.") for i in range(num_hidden_layers)])
        for lyr in self.layer[1:]:               # the reference deep-copies ONE layer: identical complex_weight everywhere
            lyr.filterlayer.complex_weight.data.copy_(self.layer[0].filterlayer.complex_weight.data)


class _Encode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, anchor, idx, training):
        eng = model.engine
        idx = idx.contiguous()
        out = eng.encode(eng.make_plan(idx, None), training)
        ctx.model, ctx.args = model, (idx, training)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, training = ctx.args
        eng = ctx.model.engine
        eng.encode_bwd(eng.make_plan(idx, None), training, gout.contiguous())
        return None, None, None, None


class FMLP(BaseModel):
    def __init__(self, config, dataset_list) -> None:
        super().__init__(config, dataset_list)
        tc = config["train"]
        if self.embed_dim != 64 or self.max_seq_len != 50:
            raise ValueError("FMLP hard-codes embed_dim 64 and max_seq_len 50 (reference model/fmlp.py:11-12)")
        max_b = max(int(tc["batch_size"]), int(config["eval"]["batch_size"]))
        self.engine = FmlpEngine(self.num_items, 50, 64, 256, config["model"]["layer_num"], 1e-12, 0.5, max_b, self.device,
                                 seed=int(tc["seed"]) + 7919 * self.rank, lr=float(tc["learning_rate"]),
                                 weight_decay=float(tc["weight_decay"]))
        self.device = self.engine.device
        eng = self.engine
        self.item_embedding = _Embedding(eng, "item_embedding.weight", self.num_items, 64, padding_idx=0)
        self.position_embeddings = _Embedding(eng, "position_embeddings.weight", 50, 64)
        self.LayerNorm = _LayerNorm(eng, "LayerNorm.", 64, 1e-12)
        self.dropout = nn.Dropout(0.5)
        self.item_encoder = FMLPEncoder(eng, num_hidden_layers=config["model"]["layer_num"])
        self._rows_buf = torch.zeros(int(tc["batch_size"]), dtype=torch.int64, device=self.device)
        self._neg_buf = torch.zeros(int(tc["batch_size"]), dtype=torch.int64, device=self.device)

    def forward(self, batch, need_pooling=True):
        return _Encode.apply(self, self.item_embedding.weight, batch["in_" + self.fiid], bool(self.training))

    def training_step(self, batch, reduce=True, return_query=False, align=False):
        return super().training_step(batch, reduce, return_query)

    def _encode_raw(self, batch, training=True):
        eng = self.engine
        return eng.encode(eng.make_plan(batch["in_" + self.fiid], None), training)

    def _encode_bwd_raw(self, batch, d_query, training=True):
        eng = self.engine
        eng.encode_bwd(eng.make_plan(batch["in_" + self.fiid], None), training, d_query)

    def _batch_plan(self, batch):
        return self.engine.make_plan(batch["in_" + self.fiid], batch[self.fiid],
                                     neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False)

    def _api_plan(self):
        return None

    _supports_perm_sel = True          # round 4: batch selection inside the step's first launch (k steps per graph, no per-step rows copy)

    def _train_plan(self, fields, rows, perm_sel=None, loss_log=None):
        return self.engine.make_plan(fields["in_item_id"], fields["item_id"], rows=rows, neg_item=self._neg_buf, sample_neg=True,
                                     perm_sel=perm_sel, loss_log=loss_log)
