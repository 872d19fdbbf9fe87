"""Device-resident datasets with the interface of the reference's data/dataset.py (BaseDataset :10-119,
SeparateDataset :121-164) — the one all shipped configs use (`dataset_class: 'general'`).

On-disk format kept (so the reference's files load unchanged):
  dataset/<name>/<domain>/inter.csv            columns user_id,item_id,...,domain  -> num_users / num_items (+1 PAD)
  dataset/<name>/<domain>/train<train_file>.pth, val.pth, test.pth  = pickled python list of rows
     train row: [user_id, hist[L], target[L], seqlen, label[L], domain_id[L]]
     val/test : [user_id, hist[L], target,    seqlen, label,    domain_id[L], user_hist]
What changes is HOW batches are made: the reference runs DataLoader(self, bs, shuffle) with a per-sample
__getitem__ (7 scalar index ops) and default_collate (7 stacks) per batch; here a batch is ONE permutation slice
`rows` — the kernels index the resident tensors through it (a1 in SURVEY.md §8), and get_loader() yields the
same dict of tensors for API users via one index_select per field.
`dataset_class: 'synthetic'` builds Amazon-toys-shaped rows in memory (dr4sr_amd/data/synthetic.py).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List

import numpy as np
import torch

FIELDS_TRAIN = ("user_id", "in_item_id", "item_id", "seqlen", "label", "domain_id")


class BatchLoader:
    """Iterable over batches of a field dict; len() = number of batches (no drop_last, like the reference)."""

    def __init__(self, fields: Dict[str, torch.Tensor], batch_size: int, shuffle: bool, generator=None):
        self.fields, self.batch_size, self.shuffle, self.generator = fields, batch_size, shuffle, generator
        self.n = int(next(iter(fields.values())).shape[0])

    def __len__(self):
        return (self.n + self.batch_size - 1) // self.batch_size

    def permutation(self) -> torch.Tensor:
        dev = next(iter(self.fields.values())).device
        if self.shuffle:
            return torch.randperm(self.n, device=dev, generator=self.generator)
        return torch.arange(self.n, device=dev)

    def __iter__(self):
        perm = self.permutation()
        for i in range(0, self.n, self.batch_size):
            rows = perm[i:i + self.batch_size]
            batch = {k: v.index_select(0, rows) for k, v in self.fields.items()}
            batch["index"] = rows
            yield batch


class BaseDataset:
    def __init__(self, config: dict, phase: str = "train") -> None:
        self.name = config["data"]["dataset"]
        self.fuid, self.fiid = "user_id", "item_id"
        self.logger = logging.getLogger("CDR")
        self.config, self.phase = config, phase
        self.device = config["train"]["device"]
        self.domain_name_list = config["data"]["domain_name_list"]
        self.max_seq_len = config["data"]["max_seq_len"]
        self._data = None
        self.data = None
        self._load_datasets()
        self.domain_user_mapping = self.get_domain_user_mapping()
        self.domain_item_mapping = self.get_domain_item_mapping()
        self.eval_domain = self.domain_name_list[0]

    # ---- sizes -----------------------------------------------------------------------------------
    def __len__(self):
        if self.phase == "train":
            return len(self.data_index)
        return len(self.data[self.eval_domain][0])

    @property
    def num_users(self):
        return self._num_users

    @property
    def num_items(self):
        return self._num_items

    @property
    def num_domains(self):
        return len(self.domain_name_list)

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self.name}, phase={self.phase}, users={self._num_users}, items={self._num_items})"

    # ---- loading ---------------------------------------------------------------------------------
    def _domain_dir(self, domain):
        return os.path.join("dataset", self.name, domain)

    def _load_datasets(self):
        import pandas as pd
        frames = [pd.read_csv(os.path.join(self._domain_dir(d), "inter.csv")) for d in self.domain_name_list]
        self._inter_data = pd.concat(frames)
        self._num_users = int(self._inter_data["user_id"].nunique()) + 1      # +1 for padding
        self._num_items = int(self._inter_data["item_id"].nunique()) + 1

    def get_domain_user_mapping(self):
        return {d: self._inter_data[self._inter_data["domain"] == i]["user_id"].unique().tolist()
                for i, d in enumerate(self.domain_name_list)}

    def get_domain_item_mapping(self):
        return {d: self._inter_data[self._inter_data["domain"] == i]["item_id"].unique().tolist()
                for i, d in enumerate(self.domain_name_list)}

    def unpack(self, rows):
        """rows -> tuple of int64 device tensors in the reference's order (data/dataset.py:79-91).  `rows` is the reference's
        list of python rows or the column dict of data/packed.py (int32 arrays, possibly memory-mapped)."""
        from .packed import COLS, rows_to_arrays
        dev = self.device
        arrays = rows if isinstance(rows, dict) else rows_to_arrays(rows, self.max_seq_len)
        cols = [torch.from_numpy(np.asarray(arrays[k]).astype(np.int64)).to(dev) for k in COLS]
        if self.phase != "train":
            cols.append(cols[1])                                    # user_hist = the input sequence
        return tuple(cols)

    def _build(self):
        raise NotImplementedError

    def build(self):
        self._build()
        if self.phase == "train":
            self.data_index = torch.arange(len(self._data[0]))
        self.data = self._data

    def set_eval_domain(self, domain):
        self.eval_domain = domain

    def set_data_index(self, data_index):
        assert self.phase == "train"
        self.data_index = data_index
        self.data = [t[self.data_index.to(t.device)] for t in self._data]

    # ---- batches ---------------------------------------------------------------------------------
    def fields(self) -> Dict[str, torch.Tensor]:
        data = self.data if self.phase == "train" else self.data[self.eval_domain]
        f = dict(zip(FIELDS_TRAIN, data[:6]))
        if self.phase != "train":
            f["user_hist"] = data[6]
        return f

    def get_loader(self, batch_size=None, shuffle=True) -> BatchLoader:
        if self.phase == "train":
            bs = self.config["train"]["batch_size"] if batch_size is None else batch_size
            return BatchLoader(self.fields(), bs, shuffle)
        bs = self.config["eval"]["batch_size"] if batch_size is None else batch_size
        return BatchLoader(self.fields(), bs, False)

    def __getitem__(self, idx):
        batch = {k: v[idx] for k, v in self.fields().items()}
        batch["index"] = idx
        return batch


class SeparateDataset(BaseDataset):
    """rows of every domain concatenated (train) / kept per domain (val, test) — data/dataset.py:121-164"""

    def _load_datasets(self):
        super()._load_datasets()
        from .packed import load_split
        self._raw = []
        cache = bool(self.config["data"].get("packed_cache", True))
        for d in self.domain_name_list:
            fname = ("train" + self.config["data"]["train_file"] if self.phase == "train" else self.phase) + ".pth"
            self._raw.append(load_split(os.path.join(self._domain_dir(d), fname), self.max_seq_len, cache))

    def _build(self):
        if self.phase == "train":
            rows = {k: np.concatenate([dom[k] for dom in self._raw]) for k in self._raw[0]}
            self._data = self.unpack(rows)
        else:
            self._data = {d: self.unpack(rows) for d, rows in zip(self.domain_name_list, self._raw)}
        self._raw = None


class SyntheticDataset(BaseDataset):
    """Amazon-toys-shaped synthetic rows (no files).  data.n_rows / data.n_items / data.dense / data.seed / data.markov optional."""

    def _load_datasets(self):
        from .synthetic import TOYS_N_ITEMS, TOYS_N_ROWS
        d = self.config["data"]
        self._num_items = int(d.get("n_items", TOYS_N_ITEMS))
        self._n_rows = int(d.get("n_rows", TOYS_N_ROWS))
        self._num_users = self._n_rows + 1
        self._inter_data = None

    def get_domain_user_mapping(self):
        return {d: list(range(1, self._num_users)) for d in self.domain_name_list}

    def get_domain_item_mapping(self):
        return {d: list(range(1, self._num_items)) for d in self.domain_name_list}

    def _build(self):
        from .synthetic import make_rows
        d = self.config["data"]
        seed = int(d.get("seed", 2024))
        L = self.max_seq_len
        dev = self.device
        prefix = bool(d.get("prefix_rows", False))          # FMLP format: left-padded prefix, scalar target
        if self.phase == "train" and not prefix:
            r = make_rows(self._n_rows, self._num_items, L, seed, bool(d.get("dense", False)), float(d.get("markov", 0.0)))
            self._data = tuple(torch.from_numpy(r[k]).to(dev) for k in FIELDS_TRAIN)
            return
        if self.phase == "train":
            r = make_rows(self._n_rows, self._num_items, L, seed, bool(d.get("dense", False)), float(d.get("markov", 0.0)))
            sl = torch.from_numpy(r["seqlen"])
            hist = torch.from_numpy(r["in_item_id"])
            shift = (L - sl).view(-1, 1)                                            # roll the valid prefix to the right edge
            cols = (torch.arange(L).view(1, -1) - shift) % L
            lp = torch.where(torch.arange(L).view(1, -1) >= shift, hist.gather(1, cols), torch.zeros_like(hist))
            target = torch.from_numpy(r["item_id"]).gather(1, (sl - 1).clamp(min=0).view(-1, 1)).squeeze(1)
            self._data = (torch.from_numpy(r["user_id"]).to(dev), lp.to(dev), target.to(dev), sl.to(dev),
                          torch.ones_like(sl).to(dev), torch.zeros_like(hist).to(dev))
            return
        n_eval = int(d.get("n_eval_rows", min(self._n_rows, 4096)))
        r = make_rows(n_eval, self._num_items, L, seed + (1 if self.phase == "val" else 2), False, float(d.get("markov", 0.0)))
        hist = torch.from_numpy(r["in_item_id"]).to(dev)
        sl = torch.from_numpy(r["seqlen"]).to(dev)
        tgt_all = torch.from_numpy(r["item_id"]).to(dev)
        target = tgt_all.gather(1, (sl - 1).clamp(min=0).view(-1, 1)).squeeze(1)      # next item after the history
        if prefix:
            shift = (L - sl).view(-1, 1)
            ar = torch.arange(L, device=dev).view(1, -1)
            hist = torch.where(ar >= shift, hist.gather(1, (ar - shift) % L), torch.zeros_like(hist))
        cols = (torch.from_numpy(r["user_id"]).to(dev), hist, target, sl, torch.ones_like(sl),
                torch.zeros_like(hist), hist)
        self._data = {dn: cols for dn in self.domain_name_list}
