"""Sequence augmentations and the contrastive loss of CL4SRec with the class surface of the reference's
module/data_augmentation.py (Item_Crop :20-41, Item_Mask :44-62, Item_Reorder :65-85, Item_Random :87-95, InfoNCELoss :305-350 for
sim_method='inner_product' / neg_type='batch_both', CL4SRecAugmentation :577-619) over libdr4sr_hip.so (dr4sr_cl_augment,
dr4sr_infonce_fwd/_bwd).

Differences that are not behavioural: the reference loops over the batch in Python drawing from torch / numpy / random; here one
kernel draws from Philox (same distributions, other streams).  Item_Crop returns sequences left-aligned in the ORIGINAL width L
(the reference pads to the longest crop): positions >= the new length are never computed by the packed encoder, so results agree.
"""
from __future__ import annotations

import torch

from .. import _lib

_MODES = {"item_crop": 0, "item_mask": 1, "item_reorder": 2, "item_random": 3}


class _DeviceAugmentation(torch.nn.Module):
    mode = 0

    def __init__(self, mask_id=0, tao=0.2, gamma=0.7, beta=0.2, seed=2023):
        super().__init__()
        self.mask_id, self.tao, self.gamma, self.beta, self.seed = int(mask_id), float(tao), float(gamma), float(beta), int(seed)
        self.calls = 0
        self.step_dev = None        # int32[1] device copy of `calls` while a captured training step is being built / replayed
        self._in_step = 0

    def expected_len_factor(self) -> float:
        """mean view length / mean input length (regime hint of the views' encoder plans, engine.make_plan(expected_tokens=...)): a crop
        keeps a tao share of a sequence, mask and reorder keep its length, 'random' draws the three uniformly"""
        return {0: self.tao, 1: 1.0, 2: 1.0, 3: (self.tao + 2.0) / 3.0}[self.mode]

    def begin_step(self):
        """captured steps: call at the top of the step body; the k-th forward() of the step draws stream step_dev + k, and
        end_step() advances the device counter (all of it recorded into the graph)"""
        self._in_step = 0

    def end_step(self):
        if self.step_dev is not None:
            self.step_dev.add_(self._in_step)

    def two_views(self, sequences, seq_lens, rows=None):
        """the two consecutive draws of a CL4SRec step in ONE launch (captured steps; eager: two forward() calls).  The two views are
        the halves of ONE [2B, L] / [2B] pair of tensors, so that a caller may also encode them as one batch of 2B sequences.
        rows (int64 [B], captured steps only): sequences / seq_lens are DATASET tensors and the batch is their rows rows[0..B) — the
        batch a fused step selected on the device is never materialised (dr4sr_cl_augment2_rows_dev; same draws as on the gathered rows)"""
        seq, sl = sequences.contiguous(), seq_lens.contiguous()
        L = int(seq.shape[1])
        B = int(rows.shape[0]) if rows is not None else int(seq.shape[0])
        out, out_len = seq.new_empty(2 * B, L), sl.new_empty(2 * B)
        oi, li, oj, lj = out[:B], out_len[:B], out[B:], out_len[B:]
        if rows is not None:
            assert self.step_dev is not None, "rows-indirected views are the captured step's form (device call counter)"
            lib = _lib.load()
            _lib.check(lib.dr4sr_cl_augment2_rows_dev(_lib.ptr(seq), _lib.ptr(sl), _lib.ptr(rows), _lib.ptr(oi), _lib.ptr(li), _lib.ptr(oj),
                                                      _lib.ptr(lj), B, L, self.mode, self.tao, self.gamma, self.beta, self.mask_id, self.seed,
                                                      _lib.ptr(self.step_dev), self._in_step + 1, _lib.cur_stream()), "dr4sr_cl_augment2_rows_dev")
            self._in_step += 2
            return (oi, li), (oj, lj)
        if self.step_dev is None:
            (a, la), (b, lb) = self.forward(sequences, seq_lens), self.forward(sequences, seq_lens)
            oi.copy_(a); li.copy_(la); oj.copy_(b); lj.copy_(lb)
            return (oi, li), (oj, lj)
        lib = _lib.load()
        _lib.check(lib.dr4sr_cl_augment2_dev(_lib.ptr(seq), _lib.ptr(sl), _lib.ptr(oi), _lib.ptr(li), _lib.ptr(oj), _lib.ptr(lj), B, L, self.mode,
                                             self.tao, self.gamma, self.beta, self.mask_id, self.seed, _lib.ptr(self.step_dev),
                                             self._in_step + 1, _lib.cur_stream()), "dr4sr_cl_augment2_dev")
        self._in_step += 2
        return (oi, li), (oj, lj)

    def forward(self, sequences, seq_lens):
        lib = _lib.load()
        seq, sl = sequences.contiguous(), seq_lens.contiguous()
        B, L = seq.shape
        out, out_len = torch.empty_like(seq), torch.empty_like(sl)
        if self.step_dev is not None:
            self._in_step += 1
            _lib.check(lib.dr4sr_cl_augment_dev(_lib.ptr(seq), _lib.ptr(sl), _lib.ptr(out), _lib.ptr(out_len), B, L, self.mode, self.tao,
                                                self.gamma, self.beta, self.mask_id, self.seed, _lib.ptr(self.step_dev), self._in_step,
                                                _lib.cur_stream()), "dr4sr_cl_augment_dev")
            return out, out_len
        self.calls += 1
        _lib.check(lib.dr4sr_cl_augment(_lib.ptr(seq), _lib.ptr(sl), _lib.ptr(out), _lib.ptr(out_len), B, L, self.mode, self.tao,
                                        self.gamma, self.beta, self.mask_id, self.seed, self.calls, _lib.cur_stream()), "dr4sr_cl_augment")
        return out, out_len


class Item_Crop(_DeviceAugmentation):
    mode = 0

    def __init__(self, tao=0.2, **kw):
        super().__init__(tao=tao, **kw)


class Item_Mask(_DeviceAugmentation):
    mode = 1

    def __init__(self, mask_id, gamma=0.7, **kw):
        super().__init__(mask_id=mask_id, gamma=gamma, **kw)


class Item_Reorder(_DeviceAugmentation):
    mode = 2

    def __init__(self, beta=0.2, **kw):
        super().__init__(beta=beta, **kw)


class Item_Random(_DeviceAugmentation):
    mode = 3

    def __init__(self, mask_id, tao=0.2, gamma=0.7, beta=0.2, **kw):
        super().__init__(mask_id=mask_id, tao=tao, gamma=gamma, beta=beta, **kw)


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rep_i, rep_j, valid, temperature, reduce):
        lib = _lib.load()
        xi, xj = rep_i.contiguous(), rep_j.contiguous()
        B, D = xi.shape
        lse = torch.empty(B, dtype=torch.float32, device=xi.device)
        loss_row = torch.empty(B, dtype=torch.float32, device=xi.device)
        stats = torch.zeros(2, dtype=torch.float32, device=xi.device)
        _lib.check(lib.dr4sr_infonce_fwd(_lib.ptr(xi), _lib.ptr(xj), _lib.ptr(valid), B, D, temperature, _lib.ptr(lse), _lib.ptr(loss_row),
                                         _lib.ptr(stats), _lib.cur_stream()), "dr4sr_infonce_fwd")
        ctx.save_for_backward(xi, xj, lse, stats)
        ctx.valid, ctx.temperature, ctx.reduce = valid, temperature, reduce
        if reduce:
            return stats[1] / stats[0]                                 # F.cross_entropy(logits, labels)
        rows = loss_row / stats[0]                                     # cross_entropy(reduction='none') / batch_size
        return rows if valid is None else rows[valid.bool()]

    @staticmethod
    def backward(ctx, gout):
        xi, xj, lse, stats = ctx.saved_tensors
        lib = _lib.load()
        B, D = xi.shape
        if not ctx.reduce:
            raise NotImplementedError("InfoNCE(reduce=False).backward with per-row upstream weights is not on the HIP path")
        scale = (gout.reshape(1) / stats[0]).contiguous()
        dxi, dxj = torch.zeros_like(xi), torch.zeros_like(xj)
        _lib.check(lib.dr4sr_infonce_bwd(_lib.ptr(xi), _lib.ptr(xj), _lib.ptr(ctx.valid), B, D, ctx.temperature, _lib.ptr(lse),
                                         _lib.ptr(scale), _lib.ptr(dxi), _lib.ptr(dxj), _lib.cur_stream()), "dr4sr_infonce_bwd")
        return dxi, dxj, None, None, None


class InfoNCELoss(torch.nn.Module):
    def __init__(self, temperature: float = 1.0, sim_method: str = "inner_product", neg_type: str = "batch_both") -> None:
        super().__init__()
        if sim_method != "inner_product" or neg_type != "batch_both":
            raise NotImplementedError("HIP InfoNCE: sim_method='inner_product', neg_type='batch_both' (what CL4SRecAugmentation uses)")
        self.temperature, self.sim_method, self.neg_type = float(temperature), sim_method, neg_type

    def forward(self, augmented_rep_i, augmented_rep_j, instance_labels=None, all_reps=None, reduce=True, valid=None):
        """valid (uint8/bool [B], extension): rows to keep — equivalent to indexing both inputs with it first, without the copy"""
        assert instance_labels is None and all_reps is None
        if valid is not None:
            valid = valid.to(torch.uint8).contiguous()
        return _InfoNCE.apply(augmented_rep_i, augmented_rep_j, valid, self.temperature, reduce)


class CL4SRecAugmentation(torch.nn.Module):
    def __init__(self, config, train_data, seed=2023) -> None:
        super().__init__()
        self.config = config
        self.fiid = train_data.fiid
        t = config["augment_type"]
        if t == "item_crop":
            self.augmentation = Item_Crop(config["tau"], seed=seed)
        elif t == "item_mask":
            self.augmentation = Item_Mask(mask_id=train_data.num_items, gamma=config["gamma"], seed=seed)
        elif t == "item_reorder":
            self.augmentation = Item_Reorder(beta=config["beta"], seed=seed)
        elif t == "item_random":
            self.augmentation = Item_Random(mask_id=train_data.num_items, tao=config["tau"], gamma=config["gamma"], beta=config["beta"],
                                            seed=seed)
        else:
            raise ValueError(f"augmentation type: '{t}' is invalided")
        self.InfoNCE_loss_fn = InfoNCELoss(temperature=config["temperature"], sim_method="inner_product", neg_type="batch_both")

    def forward(self, batch, query_encoder, reduce=True):
        seqs, lens = batch["in_" + self.fiid], batch["seqlen"]
        aug_i, len_i = self.augmentation(seqs, lens)
        aug_j, len_j = self.augmentation(seqs, lens)
        # need_pooling=False + seq_pooling_function('mean') of the reference, fused into the encoder call; one engine slot per view
        out_i = query_encoder({"in_" + self.fiid: aug_i, "seqlen": len_i}, need_pooling=False, slot=1, pooling="mean")
        out_j = query_encoder({"in_" + self.fiid: aug_j, "seqlen": len_j}, need_pooling=False, slot=2, pooling="mean")
        valid = batch["seqlen"] != 1                                   # data_augmentation.py:613-615
        return {"cl_loss": self.InfoNCE_loss_fn(out_i, out_j, reduce=reduce, valid=valid)}
