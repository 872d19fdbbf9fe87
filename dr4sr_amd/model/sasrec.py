"""SASRec with the class surface of the reference's model/sasrec.py (SASRecQueryEncoder :10-75, SASRec :77-120).

The module tree exists to carry PARAMETERS under the reference's state-dict names (so checkpoints interchange):
  item_embedding.weight == query_encoder.item_encoder.weight (tied), query_encoder.position_emb.weight,
  query_encoder.transformer_layer.layers.{i}.{self_attn.in_proj_weight|in_proj_bias|out_proj.*, linear1.*, linear2.*,
  norm1.*, norm2.*}
Every Parameter is a VIEW into the engine's flat fp32 buffer (and .grad a view of the flat gradient); all arithmetic
runs in libdr4sr_hip.so.  There is no PyTorch compute path.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _lib
from ..engine import SasrecEngine
from ..parallel import allreduce_flat, shard_bounds
from .basemodel import BaseModel


def _bind(module: nn.Module, name: str, view: torch.Tensor, grad: torch.Tensor):
    p = nn.Parameter(view, requires_grad=True)
    p.grad = grad
    setattr(module, name, p)
    return p


class _Linear(nn.Linear):
    """parameter holder (never called); subclass of nn.Linear so normal_initialization treats it as the reference's"""

    def __init__(self, eng: SasrecEngine, prefix: str, out_f: int, in_f: int):
        nn.Module.__init__(self)
        self.in_features, self.out_features = in_f, out_f
        _bind(self, "weight", eng.views[prefix + "weight"], eng.grad_views[prefix + "weight"])
        _bind(self, "bias", eng.views[prefix + "bias"], eng.grad_views[prefix + "bias"])


class _LayerNorm(nn.LayerNorm):
    def __init__(self, eng: SasrecEngine, prefix: str, dim: int, eps: float):
        nn.Module.__init__(self)
        self.normalized_shape, self.eps, self.elementwise_affine = (dim,), eps, True
        _bind(self, "weight", eng.views[prefix + "weight"], eng.grad_views[prefix + "weight"])
        _bind(self, "bias", eng.views[prefix + "bias"], eng.grad_views[prefix + "bias"])


class _Embedding(nn.Embedding):
    def __init__(self, eng: SasrecEngine, name: str, num: int, dim: int, padding_idx=None):
        nn.Module.__init__(self)
        self.num_embeddings, self.embedding_dim, self.padding_idx = num, dim, padding_idx
        self.max_norm, self.norm_type, self.scale_grad_by_freq, self.sparse = None, 2.0, False, False
        _bind(self, "weight", eng.views[name], eng.grad_views[name])


class _SelfAttn(nn.Module):
    def __init__(self, eng, prefix, D):
        super().__init__()
        _bind(self, "in_proj_weight", eng.views[prefix + "in_proj_weight"], eng.grad_views[prefix + "in_proj_weight"])
        _bind(self, "in_proj_bias", eng.views[prefix + "in_proj_bias"], eng.grad_views[prefix + "in_proj_bias"])
        self.out_proj = _Linear(eng, prefix + "out_proj.", D, D)


class _EncoderLayer(nn.Module):
    def __init__(self, eng, prefix, D, F, eps):
        super().__init__()
        self.self_attn = _SelfAttn(eng, prefix + "self_attn.", D)
        self.linear1 = _Linear(eng, prefix + "linear1.", F, D)
        self.linear2 = _Linear(eng, prefix + "linear2.", D, F)
        self.norm1 = _LayerNorm(eng, prefix + "norm1.", D, eps)
        self.norm2 = _LayerNorm(eng, prefix + "norm2.", D, eps)


class _Encoder(nn.Module):
    def __init__(self, eng, prefix, D, F, eps, n_layer):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(eng, f"{prefix}layers.{i}.", D, F, eps) for i in range(n_layer)])


class _Encode(torch.autograd.Function):
    """SASRecQueryEncoder.forward + SeqPoolingLayer through dr4sr_sasrec_encode / _encode_bwd"""

    @staticmethod
    def forward(ctx, model, anchor, idx, seqlen, training, pooling, slot=0):
        eng = model.engine
        idx, seqlen = idx.contiguous(), seqlen.contiguous()
        plan = eng.make_plan(idx, None, seqlen, slot=slot)
        out = eng.encode(plan, training, pooling)
        ctx.model, ctx.args = model, (idx, seqlen, training, pooling, slot)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, seqlen, training, pooling, slot = ctx.args
        eng = ctx.model.engine
        eng.encode_bwd(eng.make_plan(idx, None, seqlen, slot=slot), training, pooling, gout.contiguous())
        return None, None, None, None, None, None, None


class SASRecQueryEncoder(nn.Module):
    def __init__(self, fiid, embed_dim, max_seq_len, n_head, hidden_size, dropout, activation, layer_norm_eps, n_layer,
                 item_encoder, engine: SasrecEngine, owner, bidirectional=False, training_pooling_type="origin",
                 eval_pooling_type="last") -> None:
        super().__init__()
        if bidirectional or activation != "gelu":
            raise NotImplementedError("HIP SASRec: causal attention with exact-erf GELU only (configs/sasrec.yaml)")
        self.fiid, self.item_encoder = fiid, item_encoder
        self.training_pooling_type, self.eval_pooling_type = training_pooling_type, eval_pooling_type
        pre = "query_encoder."
        self.position_emb = _Embedding(engine, pre + "position_emb.weight", max_seq_len, embed_dim)
        self.transformer_layer = _Encoder(engine, pre + "transformer_layer.", embed_dim, hidden_size, layer_norm_eps, n_layer)
        self.dropout = nn.Dropout(p=dropout)
        self._owner = [owner]                     # not a submodule (avoid a cycle in the module tree)
        # torch's MultiheadAttention init: xavier_uniform in_proj (identical in every deep-copied layer), zero biases
        w = torch.empty(3 * embed_dim, embed_dim)
        nn.init.xavier_uniform_(w)
        for lyr in self.transformer_layer.layers:
            lyr.self_attn.in_proj_weight.data.copy_(w)
            lyr.self_attn.in_proj_bias.data.zero_()

    _POOL = {"origin": _lib.POOL_ORIGIN, "last": _lib.POOL_LAST, "mean": _lib.POOL_MEAN}

    def forward(self, batch, need_pooling=True, slot=0, pooling=None):
        """slot: engine workspace of this pass (passes whose backward has not run yet must not share one); pooling: override
        ('mean' = module/functional.py:50-55 fused into the encoder call, used by the CL4SRec views)"""
        if batch.get("seq_emb", None) is not None or "input_weight" in batch:
            raise NotImplementedError("seq_emb / input_weight inputs are unused by the shipped configs and not on the HIP path")
        if pooling is not None:
            pooling = self._POOL[pooling]
        elif not need_pooling:
            pooling = _lib.POOL_NONE
        else:
            pooling = self._POOL[self.training_pooling_type if self.training else self.eval_pooling_type]
        model = self._owner[0]
        return _Encode.apply(model, model.item_embedding.weight, batch["in_" + self.fiid], batch["seqlen"],
                             bool(self.training), pooling, slot)


class SASRec(BaseModel):
    def __init__(self, config, dataset_list) -> None:
        super().__init__(config, dataset_list)
        mc, tc = config["model"], config["train"]
        max_b = self._max_batch(config)
        self.engine = SasrecEngine(self._table_rows(), self.max_seq_len, self.embed_dim, mc["head_num"], mc["hidden_size"],
                                   mc["layer_num"], mc["layer_norm_eps"], mc["dropout_rate"], max_b, self.device,
                                   seed=int(tc["seed"]) + 7919 * self.rank, lr=float(tc["learning_rate"]),
                                   weight_decay=float(tc["weight_decay"]), n_slots=self._n_slots())
        self.device = self.engine.device
        # regime hint for plans on per-batch tensors (engine._expected_tokens): mean valid length of the training split
        self.engine.mean_len = None
        fields = getattr(dataset_list[0], "fields", None)
        sl = fields().get("seqlen") if callable(fields) else None
        if sl is not None and sl.numel():
            self.engine.mean_len = float(sl.clamp(0, self.max_seq_len).float().mean())
        else:                                  # a dataset class without resident fields: per-batch plans measure their own tensors
            self.logger.info("SASRec: the training split exposes no resident seqlen tensor; per-batch plans measure their own "
                             "lengths (one synchronisation per new batch)")
        self.item_embedding = _Embedding(self.engine, "item_embedding.weight", self._table_rows(), self.embed_dim, padding_idx=0)
        self.query_encoder = SASRecQueryEncoder(self.fiid, self.embed_dim, self.max_seq_len, mc["head_num"], mc["hidden_size"],
                                                mc["dropout_rate"], mc["activation"], mc["layer_norm_eps"], mc["layer_num"],
                                                self.item_embedding, self.engine, self)
        self._rows_buf = torch.zeros(int(tc["batch_size"]), dtype=torch.int64, device=self.device)
        self._neg_buf = torch.zeros(int(tc["batch_size"]) * self.max_seq_len, dtype=torch.int64, device=self.device)
        self._dummy = torch.ones(1, dtype=torch.int64, device=self.device)

    def _table_rows(self) -> int:          # rows of the item table (CL4SRec adds the mask item)
        return self.num_items

    def _max_batch(self, config) -> int:   # rows the engine's workspaces are sized for
        return max(int(config["train"]["batch_size"]), int(config["eval"]["batch_size"]))

    def _n_slots(self) -> int:             # forward passes alive per step
        return 1

    def forward(self, batch, need_pooling=True):
        return self.query_encoder(batch, need_pooling)

    def training_step(self, batch, reduce=True, return_query=False, align=False):
        if align:
            raise NotImplementedError("alignment/uniformity objective (sasrec.py:110-119) is not used by any shipped config")
        return super().training_step(batch, reduce, return_query)

    # raw (non-autograd) hooks used by MetaModel's fused weighted step / hyper-gradient
    def _encode_raw(self, batch, training=True):
        eng = self.engine
        return eng.encode(eng.make_plan(batch["in_" + self.fiid], None, batch["seqlen"]), training, _lib.POOL_ORIGIN)

    def _encode_bwd_raw(self, batch, d_query, training=True):
        eng = self.engine
        eng.encode_bwd(eng.make_plan(batch["in_" + self.fiid], None, batch["seqlen"]), training, _lib.POOL_ORIGIN, d_query)

    def _batch_plan(self, batch):
        """plan of the fused fwd_bwd on a materialised batch with the batch's own negatives"""
        return self.engine.make_plan(batch["in_" + self.fiid], batch[self.fiid], batch["seqlen"],
                                     neg_item=batch["neg_item"].contiguous().view(-1), sample_neg=False)

    def _api_plan(self):
        """plan of a bare optimizer step (no batch): a 1-row dummy; its id tensor is built once (expand().contiguous() is a copy
        kernel — inside a captured step it would be replayed every time)"""
        if getattr(self, "_dummy_ids", None) is None:
            self._dummy_ids = self._dummy.view(1, 1).expand(1, self.max_seq_len).contiguous()
        return self.engine.make_plan(self._dummy_ids, None, self._dummy)

    _supports_perm_sel = True          # batch selection fused into the step's first kernel (no per-step rows copy)

    def _train_plan(self, fields, rows, perm_sel=None, loss_log=None):
        return self.engine.make_plan(fields["in_item_id"], fields["item_id"], fields["seqlen"], rows=rows,
                                     neg_item=self._neg_buf, sample_neg=True, perm_sel=perm_sel, loss_log=loss_log)
