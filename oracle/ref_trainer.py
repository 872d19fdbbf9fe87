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

Round 3 — the other two BASELINE workloads that bench.py times get the same kind of CPU leg:
  model/gru4rec.py:12-34, module/layers.py:117-136   RefLikeGRU4Rec: Embedding -> Dropout(0.2) -> torch.nn.GRU(bias=False, 2 x 256) -> Linear,
                                                     Adam(lr 1e-3, weight_decay 1e-4) (configs/gru4rec.yaml)
  model/metamodel.py:95-194, utils/utils.py:134-252  RefLikeMetaModel: weighted inner step every step (gumbel_softmax selection of a
                                                     64->64->2 MLP on the query rows, pattern rows -> 1, PAD -> 0) and, every
                                                     `interval` steps, the outer loop: Hypergrad.grad (Neumann series by double backward,
                                                     truncate_iter 3, hpo_lr 1e-3) -> clip_grad_norm_(10) -> SGD(momentum 0.9)

Round 5 — the two remaining workloads bench.py times:
  model/fmlp.py:8-39, module/layers.py:740-807       RefLikeFMLP: Embedding + position -> LayerNorm -> Dropout(0.5) -> 2 x (FilterLayer by
                                                     torch.fft.rfft / irfft 'ortho' with a complex weight, Intermediate 64 -> 256 -> 64) ->
                                                     the LAST position as the query; one target and one negative per row (basemodel.py:50-61)
  model/cl4srec.py:28-73, module/data_augmentation.py:20-95, :305-350, :577-619
                                                     RefLikeCL4SRec: SASRec with one more table row (the mask item) + two 'item_random' views
                                                     per step (ONE of crop / mask / reorder drawn per view for the whole batch, applied row by
                                                     row in Python, as the reference does) -> encoder -> mean pooling -> InfoNCE 'batch_both',
                                                     rows of length 1 dropped; loss = BCE + cl_weight x InfoNCE
"""
from __future__ import annotations

import random
import time

import numpy as np
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

    def encode(self, batch):                                # need_pooling=False (model/sasrec.py:69-70): every position's output
        hist = batch["in_item_id"]
        L = hist.size(1)
        pos = torch.arange(L, dtype=torch.long).unsqueeze(0).expand_as(hist)
        x = self.item_encoder(hist) + self.position_emb(pos)
        causal = torch.triu(torch.ones((L, L), dtype=torch.bool), 1)
        return self.transformer_layer(src=self.dropout(x), mask=causal, src_key_padding_mask=hist == 0)

    def forward(self, batch, training_pool=True):
        out = self.encode(batch)
        L = out.size(1)
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

    def training_step(self, batch, reduce=True, return_query=False):
        q = self.query_encoder(batch, True)
        loss = bce_scorer(q, self.item_embedding.weight, batch, reduce)
        return (loss, q) if return_query else loss


def bce_scorer(q, W, batch, reduce=True):
    """model/basemodel.py:204-214 (tied scorer, pos[target == 0] = -inf) + model/loss_func.py:9-38 (masked branch, both reduce forms)"""
    pos = (q * W[batch["item_id"]]).sum(-1)
    neg = (q.unsqueeze(-2) * W[batch["neg_item"]]).sum(-1)
    pos[batch["item_id"] == 0] = -torch.inf
    pad = torch.isinf(pos)
    pos_l = F.logsigmoid(pos)
    pos_l.masked_fill_(pad, 0.0)
    pos_l = pos_l.sum() / (~pad).sum() if reduce else pos_l / (~pad).sum()
    neg_l = (F.softplus(neg) * (torch.ones_like(neg) / neg.size(-1))).sum(-1)
    neg_l.masked_fill_(pad, 0.0)
    neg_l = neg_l.sum() / (~pad).sum() if reduce else neg_l / (~pad).sum()
    return -pos_l + neg_l


class RefLikeGRU4Rec(nn.Module):
    """model/gru4rec.py:12-34: item_embedding -> Dropout -> GRULayer (torch.nn.GRU, bias=False, batch_first: module/layers.py:117-136) ->
    Linear(hidden, D); 'origin' pooling in training.  normal_initialization touches the Embedding and the Linear only: the GRU keeps
    torch's U(-1/sqrt(H), 1/sqrt(H))."""

    def __init__(self, n_items, D=64, hidden=256, n_layer=2, p=0.2, L=50):
        super().__init__()
        self.n_items, self.L = n_items, L
        self.item_embedding = nn.Embedding(n_items, D, padding_idx=0)
        self.drop = nn.Dropout(p)
        self.gru = nn.GRU(input_size=D, hidden_size=hidden, num_layers=n_layer, bias=False, batch_first=True, bidirectional=False)
        self.out = nn.Linear(hidden, D)
        self.apply(RefLikeSASRec._init)

    neg_sampling = RefLikeSASRec.neg_sampling

    def training_step(self, batch, reduce=True, return_query=False):
        y = self.out(self.gru(self.drop(self.item_embedding(batch["in_item_id"])))[0])
        L = y.size(1)
        m = torch.arange(L).unsqueeze(0).unsqueeze(2).expand(y.size(0), -1, y.size(2))
        q = y.masked_fill(m >= batch["seqlen"].view(-1, 1, 1), 0.0)
        loss = bce_scorer(q, self.item_embedding.weight, batch, reduce)
        return (loss, q) if return_query else loss


class _FmlpFilter(nn.Module):
    """module/layers.py:745-763: learnable complex weight [1, L // 2 + 1, D] applied in the frequency domain along the sequence"""

    def __init__(self, L, D, p, eps):
        super().__init__()
        self.complex_weight = nn.Parameter(torch.randn(1, L // 2 + 1, D, 2) * 0.02)
        self.out_dropout = nn.Dropout(p)
        self.LayerNorm = nn.LayerNorm(D, eps=eps)

    def forward(self, x):
        spec = torch.fft.rfft(x, dim=1, norm="ortho") * torch.view_as_complex(self.complex_weight)
        y = torch.fft.irfft(spec, n=x.size(1), dim=1, norm="ortho")
        return self.LayerNorm(self.out_dropout(y) + x)


class _FmlpIntermediate(nn.Module):
    """module/layers.py:765-783: dense 64 -> 256, GELU, dense 256 -> 64, dropout, LayerNorm(+ residual)"""

    def __init__(self, D, p, eps):
        super().__init__()
        self.dense_1 = nn.Linear(D, 4 * D)
        self.dense_2 = nn.Linear(4 * D, D)
        self.LayerNorm = nn.LayerNorm(D, eps=eps)
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        return self.LayerNorm(self.dropout(self.dense_2(F.gelu(self.dense_1(x)))) + x)


class _FmlpLayer(nn.Module):
    def __init__(self, L, D, p, eps):
        super().__init__()
        self.filterlayer = _FmlpFilter(L, D, p, eps)
        self.intermediate = _FmlpIntermediate(D, p, eps)

    def forward(self, x):
        return self.intermediate(self.filterlayer(x))


class _FmlpEncoder(nn.Module):
    def __init__(self, n_layer, L, D, p, eps):
        super().__init__()
        self.layer = nn.ModuleList([_FmlpLayer(L, D, p, eps) for _ in range(n_layer)])


class RefLikeFMLP(nn.Module):
    """model/fmlp.py:8-39 (L = 50, D = 64, dropout 0.5 hard-coded there).  State-dict names equal the reference's."""

    def __init__(self, n_items, D=64, L=50, n_layer=2, p=0.5, eps=1e-12):
        super().__init__()
        self.n_items, self.L = n_items, L
        self.item_embedding = nn.Embedding(n_items, D, padding_idx=0)
        self.position_embeddings = nn.Embedding(L, D)
        self.LayerNorm = nn.LayerNorm(D, eps=eps)
        self.dropout = nn.Dropout(p)
        self.item_encoder = _FmlpEncoder(n_layer, L, D, p, eps)
        self.apply(RefLikeSASRec._init)

    def neg_sampling(self, batch):                          # basemodel.py:50-61, the branch of a one-dimensional target
        w = torch.ones(batch["in_item_id"].shape[0], self.n_items)
        w[:, 0] = 0
        return torch.multinomial(w, 1, replacement=True).reshape_as(batch["item_id"]).unsqueeze(-1)

    def training_step(self, batch, reduce=True, return_query=False):
        hist = batch["in_item_id"]
        pos = torch.arange(hist.size(1), dtype=torch.long).unsqueeze(0).expand_as(hist)
        x = self.dropout(self.LayerNorm(self.item_embedding(hist) + self.position_embeddings(pos)))
        for layer in self.item_encoder.layer:
            x = layer(x)
        q = x[:, -1]
        loss = bce_scorer(q, self.item_embedding.weight, batch, reduce)
        return (loss, q) if return_query else loss


def fmlp_rows(rows: dict) -> dict:
    """right-padded SASRec rows -> FMLP's rows: the history left-padded (its last item at position L - 1, where fmlp.py:38 reads the
    query) and ONE target per row, the item after the last one"""
    hist, tgt, sl = np.asarray(rows["in_item_id"]), np.asarray(rows["item_id"]), np.asarray(rows["seqlen"])
    n, L = hist.shape
    ar = np.arange(L)[None, :]
    shift = (L - sl)[:, None]
    left = np.where(ar >= shift, np.take_along_axis(hist, (ar - shift) % L, 1), 0)
    out = dict(rows)
    out["in_item_id"] = np.ascontiguousarray(left)
    out["item_id"] = np.ascontiguousarray(np.take_along_axis(tgt, np.clip(sl - 1, 0, None)[:, None], 1)[:, 0])
    return out


class RefLikeCL4SRec(RefLikeSASRec):
    """model/cl4srec.py:28-73 around RefLikeSASRec; the augmentations are module/data_augmentation.py:20-95 restated: ONE method per
    view for the whole batch, applied row by row on the host"""

    def __init__(self, n_items, cl_weight=0.1, temperature=1.0, tau=0.2, gamma=0.7, beta=0.2, **kw):
        super().__init__(n_items + 1, **kw)                 # one more row: the mask item (cl4srec.py:30-32)
        self.mask_id, self.real_items = n_items, n_items
        self.cl_weight, self.temperature, self.tau, self.gamma, self.beta = cl_weight, temperature, tau, gamma, beta

    def neg_sampling(self, batch):                          # the sampler's range is the table's row count (basemodel.py:52: num_items of the dataset)
        w = torch.ones(batch["in_item_id"].shape[0], self.real_items)
        w[:, 0] = 0
        return torch.multinomial(w, self.L, replacement=True).reshape_as(batch["item_id"]).unsqueeze(-1)

    def _crop(self, seqs, lens):
        out, new = [], torch.zeros_like(lens)
        for i in range(seqs.size(0)):
            n = int(lens[i])
            m = max(1, int(self.tau * n))
            s = int(torch.randint(0, n - m + 1, (1,)))
            out.append(seqs[i, s:s + m])
            new[i] = m
        return nn.utils.rnn.pad_sequence(out, batch_first=True), new

    def _mask(self, seqs, lens):
        out = seqs.clone()
        for i in range(seqs.size(0)):
            n = int(lens[i])
            idx = np.random.choice(n, size=int(self.gamma * n), replace=False).astype(np.int64)
            out[i][idx] = self.mask_id
        return out, lens

    def _reorder(self, seqs, lens):
        out = []
        for i in range(seqs.size(0)):
            n = int(lens[i])
            m = int(self.beta * n)
            s = random.randint(0, n - m)
            order = list(range(m))
            random.shuffle(order)
            out.append(torch.cat([seqs[i, :s], seqs[i, s:s + m][order], seqs[i, s + m:]]))
        return torch.stack(out, 0), lens

    fixed_views = None                                  # tests: ((seq_i, len_i), (seq_j, len_j)) recorded from the reference instead of a draw

    def _view(self, batch, which):
        if self.fixed_views is not None:
            v, n = self.fixed_views[which]
        else:
            aug = (self._crop, self._mask, self._reorder)[random.randint(0, 2)]
            v, n = aug(batch["in_item_id"], batch["seqlen"])
        x = self.query_encoder.encode({"in_item_id": v})
        keep = (torch.arange(x.size(1)).view(1, -1) < n.view(-1, 1)).unsqueeze(-1)            # mean pooling (module/functional.py:28-55)
        return x.masked_fill(~keep, 0.0).sum(1) / n.view(-1, 1).to(x.dtype)

    def training_step(self, batch, reduce=True, return_query=False):
        loss = super().training_step(batch, reduce)
        vi, vj = self._view(batch, 0), self._view(batch, 1)
        keep = batch["seqlen"] != 1
        vi, vj = vi[keep], vj[keep]
        n = vi.size(0)
        sim_ii = (vi @ vi.T / self.temperature).masked_fill(torch.eye(n, dtype=torch.bool), float("-inf"))
        logits = torch.cat([vi @ vj.T / self.temperature, sim_ii], dim=-1)                    # InfoNCELoss 'batch_both' (:322-350)
        return loss + self.cl_weight * F.cross_entropy(logits, torch.arange(n))


class RefLikeMetaModel(nn.Module):
    """model/metamodel.py: a trainer around a sub-model.  training_step = :174-194, the outer loop = :123-166 with utils/utils.py's
    Hypergrad (:134-205) and MetaOptimizer (:207-252) restated inline (same autograd.grad calls, same order)."""

    def __init__(self, sub: nn.Module, D=64, tau_min=1.0, meta_lr=1e-3, hpo_lr=1e-3, meta_wd=1e-3, truncate_iter=3, max_grad_norm=10.0):
        super().__init__()
        self.sub = sub
        self.meta_module = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 2))
        self.meta_module.apply(RefLikeSASRec._init)
        self.tau = nn.Parameter(torch.ones(1) * 10)
        self.tau_min, self.hpo_lr, self.truncate_iter, self.max_grad_norm = tau_min, hpo_lr, truncate_iter, max_grad_norm
        self.meta_opt = torch.optim.SGD(self.meta_module.parameters(), lr=meta_lr, momentum=0.9, weight_decay=meta_wd)

    gumbel = None                                       # tests: explicit Gumbel noise [..., 2] instead of torch's draw (golden vectors)

    def selection(self, query):
        logits = self.meta_module(query)
        tau = torch.clip(self.tau, min=self.tau_min)
        if self.gumbel is not None:                        # F.gumbel_softmax(hard=False) = softmax((logits + g) / tau)
            return ((logits + self.gumbel) / tau).softmax(-1)[..., 0].squeeze()
        return F.gumbel_softmax(logits, tau=tau, dim=-1, hard=False)[..., 0].squeeze()

    def training_step(self, batch):
        loss_value, query = self.sub.training_step(batch, reduce=False, return_query=True)
        weight = self.selection(query)
        mask = batch["user_id"] == 0
        if weight.dim() == 2:
            mask = mask.unsqueeze(-1)
        weight = weight.masked_fill(mask, 1)
        weight = weight.masked_fill(batch["item_id"] == 0, 0)
        return (loss_value * weight).sum()

    def outer_loop(self, batch_val, batch_train):
        # torch >= 2.x routes nn.MultiheadAttention to a fused CPU flash kernel that has no double backward (the reference's torch
        # 1.13 had only the math path): the two forward passes whose graph is differentiated twice run under the MATH backend, as
        # tools/make_golden.py does when it runs the reference's own outer loop
        from torch.nn.attention import SDPBackend, sdpa_kernel
        with sdpa_kernel(SDPBackend.MATH):
            val_loss = self.sub.training_step(batch_val)
            train_loss = self.training_step(batch_train)
        params, aux = list(self.sub.parameters()), list(self.meta_module.parameters())
        self.meta_opt.zero_grad()
        dval = torch.autograd.grad(val_loss, params, retain_graph=True, allow_unused=True)
        dtrain = torch.autograd.grad(train_loss, params, allow_unused=True, create_graph=True)
        keep = [i for i, (a, b) in enumerate(zip(dval, dtrain)) if a is not None and b is not None]
        dval, dtrain, params = [dval[i] for i in keep], [dtrain[i] for i in keep], [params[i] for i in keep]
        p = v = dval
        for _ in range(self.truncate_iter):
            g = torch.autograd.grad(dtrain, params, grad_outputs=v, retain_graph=True, allow_unused=True)
            g = [torch.zeros_like(x) if gi is None else gi * self.hpo_lr for gi, x in zip(g, params)]
            v = [cv - cg for cv, cg in zip(v, g)]
            p = [cp + cv for cp, cv in zip(p, v)]
        v3 = torch.autograd.grad(dtrain, aux, grad_outputs=p, allow_unused=True)
        for a, g in zip(aux, v3):
            a.grad = -g
        torch.nn.utils.clip_grad_norm_(aux, max_norm=self.max_grad_norm)
        self.meta_opt.step()


def time_training(rows: dict, n_items: int, batch_size=256, warmup=3, max_steps=60, max_seconds=25.0, anomaly=True,
                  seed=2023, p=0.5, threads=None, model_kind="sasrec", interval=30):
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
        meta = None
        if model_kind == "gru4rec":                        # configs/gru4rec.yaml: dropout 0.2, Adam weight_decay 1e-4
            model = RefLikeGRU4Rec(n_items, p=p)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        elif model_kind == "fmlp":                         # dropout 0.5 whatever the config says (fmlp.py:13, layers.py:744,:762)
            model, rows = RefLikeFMLP(n_items), fmlp_rows(rows)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0)
        elif model_kind == "cl4srec":                      # configs/cl4srec.yaml: item_random, tau 0.2, gamma 0.7, beta 0.2, cl_weight 0.1
            model = RefLikeCL4SRec(n_items, p=p)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0)
        else:
            model = RefLikeSASRec(n_items, p=p)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0)
            if model_kind == "metamodel":                  # past warm-up (metamodel.py:110-112): weighted steps + outer loop on the interval
                meta = RefLikeMetaModel(model)
        model.train()
        outer = 0
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
            loss = meta.training_step(batch) if meta is not None else model.training_step(batch)
            loss.backward()
            opt.step()
            done += 1
            if meta is not None and done % interval == 0:  # metamodel.py:117-120: fresh meta batch + fresh train batch
                bv, bt = next(iter(loader)), next(iter(loader))
                bv["neg_item"], bt["neg_item"] = model.neg_sampling(bv), model.neg_sampling(bt)
                meta.outer_loop(bv, bt)
                outer += 1
            if done > warmup:
                nseq += batch["seqlen"].shape[0]
                el = time.perf_counter() - t0
                if done - warmup >= max_steps or el >= max_seconds:
                    return {"seq_per_s": nseq / el, "steps": done - warmup, "seconds": el, "loss": float(loss.detach()),
                            "threads": torch.get_num_threads(), "anomaly": anomaly, "outer_steps": outer}
    finally:
        torch.autograd.set_detect_anomaly(prev)
        torch.set_num_threads(prev_threads)
