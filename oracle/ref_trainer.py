"""ORACLE (test / baseline infrastructure, NOT product code) — a CPU trainer that issues the SAME torch op
sequence as the reference's training loop, used as bench.py's `cpu_baseline` ("port") and as a second
pin of the restatement in oracle/sasrec_oracle.py.

Mirrors (file:line under /root/reference):
  data/dataset.py:105-108,:149-164  DataLoader(dataset, 256, shuffle=True) over per-sample dict __getitem__
  model/basemodel.py:50-61          _neg_sampling: ones[B,N] -> multinomial(replacement=True)
  model/sasrec.py:10-75             nn.Embedding + nn.TransformerEncoder(post-norm, gelu, 2 heads) + 'origin' pooling
  model/basemodel.py:204-214        tied-embedding scorer, pos[target==0] = -inf
  model/loss_func.py:9-38           BinaryCrossEntropyLoss
  model/basemodel.py:193-199        zero_grad / backward / Adam(lr=1e-3).step()
  utils/utils.py:11                 torch.autograd.set_detect_anomaly(True) is ON in the reference (flag here)
State-dict names equal the reference's, so golden parameters load with strict=True.
"""
from __future__ import annotations

import time

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset


class RowDataset(Dataset):
    def __init__(self, rows: dict):
        self.d = {k: torch.as_tensor(v) for k, v in rows.items()}
        self.n = self.d["seqlen"].shape[0]

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        b = {k: self.d[k][idx] for k in ("user_id", "in_item_id", "item_id", "seqlen", "label", "domain_id")}
        b["index"] = idx
        return b


class QueryEncoder(nn.Module):
    def __init__(self, item_encoder, D, L, H, Fh, p, eps, n_layer):
        super().__init__()
        self.item_encoder = item_encoder
        self.position_emb = nn.Embedding(L, D)
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=Fh, dropout=p, activation="gelu",
                                           layer_norm_eps=eps, batch_first=True, norm_first=False)
        self.transformer_layer = nn.TransformerEncoder(encoder_layer=layer, num_layers=n_layer)
        self.dropout = nn.Dropout(p=p)

    def forward(self, batch, training_pool=True):
        hist = batch["in_item_id"]
        L = hist.size(1)
        pos = torch.arange(L, dtype=torch.long).unsqueeze(0).expand_as(hist)
        x = self.item_encoder(hist) + self.position_emb(pos)
        causal = torch.triu(torch.ones((L, L), dtype=torch.bool), 1)
        out = self.transformer_layer(src=self.dropout(x), mask=causal, src_key_padding_mask=hist == 0)
        if training_pool:                                   # SeqPoolingLayer('origin')
            m = torch.arange(L).unsqueeze(0).unsqueeze(2).expand(out.size(0), -1, out.size(2))
            return out.masked_fill(m >= batch["seqlen"].view(-1, 1, 1), 0.0)
        idx = (batch["seqlen"] - 1).view(-1, 1, 1).expand(-1, -1, out.size(2))
        return out.gather(1, idx).squeeze(1)


class RefLikeSASRec(nn.Module):
    def __init__(self, n_items, D=64, L=50, H=2, Fh=128, p=0.5, eps=1e-12, n_layer=2):
        super().__init__()
        self.n_items, self.L = n_items, L
        self.item_embedding = nn.Embedding(n_items, D, padding_idx=0)
        self.query_encoder = QueryEncoder(self.item_embedding, D, L, H, Fh, p, eps, n_layer)
        self.apply(self._init)

    @staticmethod
    def _init(m):                                           # utils/utils.py:70-81 normal_initialization
        if isinstance(m, nn.Embedding):
            m.weight.data.normal_(0.0, 0.02)
            if m.padding_idx is not None:
                nn.init.constant_(m.weight.data[m.padding_idx], 0.0)
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)

    def neg_sampling(self, batch):
        w = torch.ones(batch["in_item_id"].shape[0], self.n_items)
        w[:, 0] = 0
        neg = torch.multinomial(w, self.L, replacement=True)
        return neg.reshape_as(batch["item_id"]).unsqueeze(-1)

    def training_step(self, batch):
        q = self.query_encoder(batch, True)
        W = self.item_embedding.weight
        pos = (q * W[batch["item_id"]]).sum(-1)
        neg = (q.unsqueeze(-2) * W[batch["neg_item"]]).sum(-1)
        pos[batch["item_id"] == 0] = -torch.inf
        pad = torch.isinf(pos)
        pos_l = F.logsigmoid(pos)
        pos_l.masked_fill_(pad, 0.0)
        pos_l = pos_l.sum() / (~pad).sum()
        neg_l = (F.softplus(neg) * (torch.ones_like(neg) / neg.size(-1))).sum(-1)
        neg_l.masked_fill_(pad, 0.0)
        neg_l = neg_l.sum() / (~pad).sum()
        return -pos_l + neg_l


def time_training(rows: dict, n_items: int, batch_size=256, warmup=3, max_steps=60, max_seconds=25.0, anomaly=True,
                  seed=2023, p=0.5, threads=None):
    """seq/s of the reference-equivalent CPU step (batch build + neg sampling + fwd + bwd + Adam).
    threads: torch intra-op threads for this run (None = leave as is); the reference sets nothing, i.e. torch's default = all
    cores, which on a 128-core host is slower than 8-32 threads for these microsecond-sized ops — bench.py sweeps and reports."""
    torch.manual_seed(seed)
    prev = torch.is_anomaly_enabled()
    prev_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(int(threads))
    torch.autograd.set_detect_anomaly(anomaly)
    try:
        model = RefLikeSASRec(n_items, p=p)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0)
        model.train()
        loader = DataLoader(RowDataset(rows), batch_size, shuffle=True)
        it = iter(loader)
        done, nseq, t0 = 0, 0, None
        while True:
            try:
                batch = next(it)
            except StopIteration:
                it = iter(loader)
                batch = next(it)
            if done == warmup:
                t0 = time.perf_counter()
            batch["neg_item"] = model.neg_sampling(batch)
            opt.zero_grad()
            loss = model.training_step(batch)
            loss.backward()
            opt.step()
            done += 1
            if done > warmup:
                nseq += batch["seqlen"].shape[0]
                el = time.perf_counter() - t0
                if done - warmup >= max_steps or el >= max_seconds:
                    return {"seq_per_s": nseq / el, "steps": done - warmup, "seconds": el, "loss": float(loss.detach()),
                            "threads": torch.get_num_threads(), "anomaly": anomaly}
    finally:
        torch.autograd.set_detect_anomaly(prev)
        torch.set_num_threads(prev_threads)
