#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference (USTC-StarTeam/DR4SR).

This script only works in the build container where /root/reference exists.  It
imports the reference unmodified (with no-op stubs for the three absent third
party modules wandb / faiss / torchmetrics), builds a tiny synthetic dataset in
the reference's on-disk row format, runs fixed batches through the reference's
own `training_step` / `backward` / `optimizer.step` / `topk`, and dumps inputs and
outputs as small .npz fixtures under tests/golden/.

Only DATA is committed (tests/golden/*.npz); no reference source travels.

Reference entry points exercised (file:line under /root/reference):
  utils/utils.py:90-109    load_config            (3-way YAML merge)
  utils/utils.py:38-55     prepare_datasets / prepare_model
  model/basemodel.py:44-48 _init_model            (normal_initialization, Adam, BCE)
  model/basemodel.py:204   training_step          (encoder fwd + tied scorer + BCE)
  model/basemodel.py:354   topk                   (full-item scorer + history mask + top-k)
  model/sasrec.py:39-75    SASRecQueryEncoder.forward
  model/loss_func.py:9-38  BinaryCrossEntropyLoss.forward

Usage:  python tools/make_golden.py [--out tests/golden]
"""
import argparse
import os
import shutil
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
L = 50


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    wandb = types.ModuleType("wandb")
    wandb.init = lambda *a, **k: _Ctx()
    wandb.log = lambda *a, **k: None
    wandb.finish = lambda *a, **k: None
    wandb.sweep = lambda *a, **k: None
    wandb.agent = lambda *a, **k: None
    wandb.Image = lambda *a, **k: None
    wandb.config = {}
    sys.modules["wandb"] = wandb

    faiss = types.ModuleType("faiss")
    faiss.Kmeans = type("Kmeans", (), {})
    sys.modules["faiss"] = faiss

    tm = types.ModuleType("torchmetrics")
    tmf = types.ModuleType("torchmetrics.functional")
    for n in ("recall", "precision", "f1_score", "auroc", "accuracy",
              "mean_squared_error", "mean_absolute_error"):
        setattr(tmf, n, lambda *a, **k: None)
    tm.functional = tmf
    sys.modules["torchmetrics"] = tm
    sys.modules["torchmetrics.functional"] = tmf


# --------------------------------------------------------------------------- data
def _pad(seq, n=L):
    return list(seq) + [0] * (n - len(seq))


def build_dataset(root, n_items, seqlens, rng, dataset="amazon-toys", domain="toy"):
    """Write inter.csv/train_ori.pth/val.pth/test.pth in the reference row format.

    Row format (dataset/preprocess_amazon.ipynb cell 20; data/dataset.py:79-91):
      train: [user_id, hist[50], target[50], seqlen, label[50], domain_id[50]]
      val/test: [user_id, hist[50], target, seqlen, 1, domain_id[50], hist]
    """
    import torch
    d = os.path.join(root, "dataset", dataset, domain)
    os.makedirs(d, exist_ok=True)
    train, val, test = [], [], []
    for u, sl in enumerate(seqlens, start=1):
        full = rng.integers(1, n_items, size=sl + 3).tolist()      # sl inputs + 1 shift + val + test
        hist = full[:sl]
        tgt = full[1:sl + 1]
        train.append([u, _pad(hist), _pad(tgt), sl, [1] * sl + [0] * (L - sl), [0] * L])
        vh = full[:sl + 1][-L:]
        val.append([u, _pad(vh), full[sl + 1], len(vh), 1, [0] * L, _pad(vh)])
        th = full[:sl + 2][-L:]
        test.append([u, _pad(th), full[sl + 2], len(th), 1, [0] * L, _pad(th)])
    torch.save(train, os.path.join(d, "train_ori.pth"))
    torch.save(val, os.path.join(d, "val.pth"))
    torch.save(test, os.path.join(d, "test.pth"))
    # inter.csv must cover every user and every item id 1..n_items-1 (data/dataset.py:56-65)
    with open(os.path.join(d, "inter.csv"), "w") as f:
        f.write("user_id,item_id,rating,timestamp,domain\n")
        nu = len(seqlens)
        for i in range(1, n_items):
            f.write(f"{(i - 1) % nu + 1},{i},1.0,{i},0\n")
        for u in range(1, nu + 1):
            f.write(f"{u},{(u % (n_items - 1)) + 1},1.0,{u},0\n")
    return train, val, test


def build_dataset_fmlp(root, n_items, seqlens, rng, dataset="amazon-toys", domain="toy"):
    """Per-prefix, LEFT-padded rows with a scalar target (dataset/dataset_transform.ipynb cell 3; README.md:78):
    what FMLP needs because FMLP.forward returns x[:, -1] in train and eval (model/fmlp.py:38-39)."""
    import torch
    d = os.path.join(root, "dataset", dataset, domain)
    os.makedirs(d, exist_ok=True)
    lpad = lambda s: [0] * (L - len(s)) + list(s)
    train, val, test = [], [], []
    for u, sl in enumerate(seqlens, start=1):
        full = rng.integers(1, n_items, size=sl + 3).tolist()
        for k in range(1, sl + 1):                                  # one row per prefix
            if sl > 10 and k < sl - 2:                              # fixture size: long users keep their last 3 prefixes
                continue
            train.append([u, lpad(full[:k][-L:]), full[k], min(k, L), 1, [0] * L])
        vh = full[:sl + 1][-L:]
        val.append([u, lpad(vh), full[sl + 1], len(vh), 1, [0] * L, lpad(vh)])
        th = full[:sl + 2][-L:]
        test.append([u, lpad(th), full[sl + 2], len(th), 1, [0] * L, lpad(th)])
    torch.save(train, os.path.join(d, "train_ori.pth"))
    torch.save(val, os.path.join(d, "val.pth"))
    torch.save(test, os.path.join(d, "test.pth"))
    with open(os.path.join(d, "inter.csv"), "w") as f:
        f.write("user_id,item_id,rating,timestamp,domain\n")
        nu = len(seqlens)
        for i in range(1, n_items):
            f.write(f"{(i - 1) % nu + 1},{i},1.0,{i},0\n")
        for u in range(1, nu + 1):
            f.write(f"{u},{(u % (n_items - 1)) + 1},1.0,{u},0\n")
    return train, val, test


# --------------------------------------------------------------------------- runner
def run_case(out_dir, name, model_name, n_items, seqlens, embed_dim, seed, overrides=None):
    import torch
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        (build_dataset_fmlp if model_name == "FMLP" else build_dataset)(work, n_items, seqlens, rng)
        os.chdir(work)
        from utils import load_config, setup_environment, prepare_datasets, prepare_model
        config = load_config({"model": model_name, "dataset": "amazon-toys"})
        config["train"]["device"] = "cpu"
        config["data"]["train_file"] = "_ori"
        config["model"]["dropout_rate"] = 0.0
        config["model"]["embed_dim"] = embed_dim
        for sec, kv in (overrides or {}).items():
            config[sec].update(kv)
        setup_environment(config["train"])
        torch.manual_seed(seed)
        ds = prepare_datasets(config)
        model = prepare_model(config, ds)
        model._init_model(ds[0])
        if model_name == "FMLP":          # FMLP hard-codes nn.Dropout(0.5) (model/fmlp.py:13, module/layers.py:744,767)
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
        # The reference zero-inits biases; perturb every parameter slightly so that
        # bias / LayerNorm-affine paths are actually pinned by the fixture.
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "item_embedding" in n or "item_encoder" in n:
                    continue
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        out = {}
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k, v in sd0.items():
            out["param." + k] = v.numpy()

        # ---- training batch = all rows, in dataset order (no shuffle) ----------------
        loader = ds[0].get_loader(batch_size=len(ds[0]), shuffle=False)
        batch = next(iter(loader))
        model.train()
        torch.manual_seed(seed + 2)
        batch["neg_item"] = model._neg_sampling(batch)
        for k, v in batch.items():
            out["batch." + k] = v.numpy()

        cap = {}
        enc = getattr(model, "query_encoder", None)
        hooks = []
        if model_name == "FMLP":
            hooks.append(model.item_encoder.register_forward_pre_hook(
                lambda m, a, kw: cap.__setitem__("x0", a[0].detach().clone()), with_kwargs=True))
            for i, lyr in enumerate(model.item_encoder.layer):
                hooks.append(lyr.filterlayer.register_forward_hook(
                    lambda m, a, o, i=i: cap.__setitem__(f"filter{i}", o.detach().clone())))
                hooks.append(lyr.register_forward_hook(
                    lambda m, a, o, i=i: cap.__setitem__(f"layer{i}", o.detach().clone())))
        if model_name == "SASRec":
            hooks.append(enc.transformer_layer.register_forward_pre_hook(
                lambda m, a, kw: cap.__setitem__("x0", kw["src"].detach().clone()), with_kwargs=True))
            for i, lyr in enumerate(enc.transformer_layer.layers):
                hooks.append(lyr.register_forward_hook(
                    lambda m, a, o, i=i: cap.__setitem__(f"layer{i}", o.detach().clone())))

        model.optimizer.zero_grad()
        loss, query = model.training_step(batch, reduce=True, return_query=True)
        loss.backward()
        for h in hooks:
            h.remove()
        for k, v in cap.items():
            out["act." + k] = v.numpy()
        out["out.query"] = query.detach().numpy()
        out["out.loss"] = loss.detach().numpy()
        seen = set()
        for n, p in model.named_parameters():      # named_parameters de-duplicates the tied table
            out["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            seen.add(n)
        with torch.no_grad():
            loss_nr = model.training_step(batch, reduce=False)
        out["out.loss_noreduce"] = loss_nr.detach().numpy()

        # scores, restated from the two lines at basemodel.py:206-207 on the reference's query
        with torch.no_grad():
            W = model.item_embedding.weight
            out["out.pos_score"] = (query * W[batch["item_id"]]).sum(-1).numpy()
            out["out.neg_score"] = (query.unsqueeze(-2) * W[batch["neg_item"]]).sum(-1).numpy()

        # ---- one optimizer step --------------------------------------------------------
        model.optimizer.step()
        for n, p in model.named_parameters():
            out["adam1." + n] = p.detach().numpy().copy()
        # second step on the same batch (pins the moment/bias-correction recursion)
        if model_name in ("SASRec", "FMLP"):
            model.optimizer.zero_grad()
            loss2 = model.training_step(batch)
            loss2.backward()
            model.optimizer.step()
            out["out.loss_step2"] = loss2.detach().numpy()
            for n, p in model.named_parameters():
                out["adam2." + n] = p.detach().numpy().copy()

        # ---- eval: full-item scorer + top-k on the validation rows, with the INITIAL weights
        model.load_state_dict(sd0)
        model.eval()
        ds[1].set_eval_domain("toy")
        model.set_eval_domain("toy")
        vb = next(iter(ds[1].get_loader(batch_size=len(seqlens))))
        with torch.no_grad():
            k = 20
            score, items = model.topk(vb, k, vb["user_hist"])
            q_last = model.forward(vb)
        for kk, v in vb.items():
            out["eval." + kk] = v.numpy()
        out["eval.topk_score"] = score.numpy()
        out["eval.topk_items"] = items.numpy()
        out["eval.query_last"] = q_last.numpy()
        import evaluation
        label = vb["item_id"].view(-1, 1) == items
        out["eval.ndcg@20"] = evaluation.ndcg(label, vb["label"].view(-1, 1), 20, mean=False).numpy()
        out["eval.recall@20"] = evaluation.recall(label, vb["label"].view(-1, 1), 20, mean=False).numpy()
        out["eval.ndcg@10"] = evaluation.ndcg(label, vb["label"].view(-1, 1), 10, mean=False).numpy()
        out["eval.recall@10"] = evaluation.recall(label, vb["label"].view(-1, 1), 10, mean=False).numpy()

        out["meta.num_items"] = np.int64(model.num_items)
        out["meta.embed_dim"] = np.int64(embed_dim)
        mc = config["model"]
        out["meta.head_num"] = np.int64(mc.get("head_num", 0))
        out["meta.hidden_size"] = np.int64(mc.get("hidden_size", 0))
        out["meta.layer_num"] = np.int64(mc.get("layer_num", 0))
        out["meta.layer_norm_eps"] = np.float64(mc.get("layer_norm_eps", 1e-12))
        out["meta.lr"] = np.float64(config["train"]["learning_rate"])
        out["meta.weight_decay"] = np.float64(config["train"]["weight_decay"])
        out["meta.torch_version"] = np.array(torch.__version__)
        os.chdir(cwd)
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), loss={float(loss.detach()):.6f}")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def run_meta_case(out_dir, name, sub_model, n_items, seqlens, seed, real=False):
    """MetaModel (DR4SR+): weighted inner step + one outer hyper-gradient step, by RUNNING the reference
    (model/metamodel.py:123-194, utils/utils.py:134-252).  Dropout 0; the Gumbel noise of F.gumbel_softmax is pinned
    by re-seeding torch right before each weighted training_step and is stored (the script asserts that the stored
    noise reproduces the reference's weights).

    real=True: the sub-model carries the SHIPPED trained toys checkpoint (strict=True) and the two batches are the first 2 x 128 REAL
    toys rows (build_real_toys); the sub-model's parameters are then not stored (they are sasrec_trained_toys.npz's) and its table
    gradients travel as (row ids, rows)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        if real:
            n_items, _ = build_real_toys(work)
        else:
            (build_dataset_fmlp if sub_model == "FMLP" else build_dataset)(work, n_items, seqlens, rng)
        os.chdir(work)
        import utils as rutils
        import model.metamodel as mm
        orig_load = rutils.load_config

        def patched_load(cfg):                       # metamodel.py:41-50 hard-codes device 0 for the sub-model
            c = orig_load(cfg)
            c["train"]["device"] = "cpu"
            c["data"]["train_file"] = "_ori"
            c["model"]["dropout_rate"] = 0.0
            if sub_model == "GRU4Rec":
                c["model"]["hidden_size"] = 128
            return c
        mm.load_config = patched_load

        def register_sub_model(self):                 # metamodel.py:41-50 with the device line dropped
            sc = patched_load({"dataset": self.config["data"]["dataset"], "model": self.config["model"]["sub_model"]})
            return rutils.get_model_class(sc["model"])(sc, self.dataset_list)
        mm.MetaModel._register_sub_model = register_sub_model
        config = patched_load({"model": "MetaModel", "dataset": "amazon-toys"})
        config["model"]["sub_model"] = sub_model
        rutils.setup_environment(config["train"])
        torch.manual_seed(seed)
        ds = rutils.prepare_datasets(config)
        model = rutils.prepare_model(config, ds)
        model._init_model(ds[0])
        sub = model.sub_model
        if sub_model == "FMLP":
            for m in sub.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
        replay = None
        if sub_model == "CL4SRec":
            # the augmentations draw from torch / numpy / random streams: every evaluation of a batch must see the SAME two views (the
            # reference builds loss_train once and differentiates that one graph twice; this script evaluates it several times), so the
            # first draw per batch is recorded and replayed — and stored, like run_cl_case's views
            import random
            random.seed(seed + 5)
            np.random.seed(seed + 5)
            real_aug = sub.augmentation_model.augmentation

            class ReplayAug(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.store, self.mode, self.k = {}, "train", 0

                def forward(self, sequences, seq_lens):
                    lst = self.store.setdefault(self.mode, [])
                    i = self.k % 2
                    self.k += 1
                    if len(lst) <= i:
                        sq, ln = real_aug(sequences, seq_lens)
                        lst.append((sq.clone(), ln.clone()))
                    return lst[i][0].clone(), lst[i][1].clone()
            replay = ReplayAug()
            sub.augmentation_model.augmentation = replay
        g = torch.Generator().manual_seed(seed + 1)
        if real:
            ck = torch.load(os.path.join(TOYS_DIR, "pre-trained_embedding.ckpt"), weights_only=False, map_location="cpu")
            sub.load_state_dict(ck["parameters"], strict=True)
        with torch.no_grad():
            for n, p in ([] if real else list(sub.named_parameters())) + list(model.meta_module.named_parameters()):
                if "item_embedding" in n or "item_encoder.weight" in n:
                    continue
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        out = {}
        for k, v in ({} if real else sub.state_dict()).items():
            out["param." + k] = v.detach().numpy().copy()
        for k, v in model.meta_module.state_dict().items():
            out["meta_param." + k] = v.detach().numpy().copy()
        out["meta.tau"] = model.tau.detach().numpy().copy()

        rows = next(iter(ds[0].get_loader(batch_size=256 if real else len(ds[0]), shuffle=False)))
        nrow = rows["user_id"].shape[0]
        half = nrow // 2
        model.train()

        def take(sl):
            b = {k: v[sl].clone() for k, v in rows.items()}
            return b
        bt, bv = take(slice(0, half)), take(slice(half, nrow))     # train batch / meta ("validation") batch
        bt["user_id"][1] = 0                                        # a pattern row: weight forced to 1 (metamodel.py:180-183)
        torch.manual_seed(seed + 2)
        bt["neg_item"] = model._neg_sampling(bt)
        bv["neg_item"] = model._neg_sampling(bv)
        for k, v in bt.items():
            out["train." + k] = v.numpy()
        for k, v in bv.items():
            out["val." + k] = v.numpy()

        def gumbel_of(shape, s):
            torch.manual_seed(s)
            return -torch.empty(shape).exponential_().log()

        if replay is not None:                                      # draw both batches' views now: the first draw consumes torch's RNG
            with torch.no_grad():                                   # stream, which the Gumbel noise below is pinned on
                replay.mode = "val"
                sub.training_step(batch=bv, align=False)
                replay.mode = "train"
                sub.training_step(batch=bt, align=False)
        # ---- inner (weighted) step: metamodel.py:174-194 -----------------------------------------
        sub.optimizer.zero_grad()
        torch.manual_seed(seed + 3)
        loss = model.training_step(batch=bt, align=False)
        loss.backward()
        with torch.no_grad():
            lv, query = sub.training_step(bt, reduce=False, return_query=True, align=False)
            gshape = tuple(query.shape[:-1]) + (2,)
            gn = gumbel_of(gshape, seed + 3)
            logits = model.meta_module(query)
            tau = torch.clip(model.tau, min=config["model"]["tau_min"])
            w = ((logits + gn) / tau).softmax(-1)[..., 0]
            torch.manual_seed(seed + 3)
            w_ref = model.selection(query)
            assert torch.equal(w, w_ref), "stored Gumbel noise does not reproduce the reference's weights"
            wm = w.masked_fill((bt["user_id"] == 0).unsqueeze(-1) if w.dim() == 2 else (bt["user_id"] == 0), 1.0)
            wm = wm.masked_fill(bt["item_id"] == 0, 0.0)
            if isinstance(lv, tuple):                               # CL4SRec sub-model (metamodel.py:186-192): un-weighted contrastive rows
                out["inner.cl_rows"] = lv[1].numpy()
                assert torch.allclose((lv[0] * wm).sum() + lv[1].sum(), loss.detach(), rtol=1e-6, atol=1e-7)
                lv = lv[0]
            else:
                assert torch.allclose((lv * wm).sum(), loss.detach(), rtol=1e-6, atol=1e-7)
        out["inner.gumbel"] = gn.numpy()
        out["inner.query"] = query.numpy()
        out["inner.loss_pos"] = lv.numpy()
        out["inner.weight"] = wm.numpy()
        out["inner.loss"] = loss.detach().numpy()
        for n, p in sub.named_parameters():
            gnp = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            if real and n == "item_embedding.weight":
                _sparse_rows(out, "inner.grad." + n, gnp)
            else:
                out["inner.grad." + n] = gnp
        for n, p in model.meta_module.named_parameters():           # loss.backward() also reaches phi (never stepped by it)
            out["inner.meta_grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            p.grad = None

        # ---- outer step: metamodel.py:148-166, utils/utils.py:145-252 -------------------------------
        params = list(sub.parameters())
        aux = list(model.meta_module.parameters())
        # torch >= 2.x routes MultiheadAttention to a fused CPU flash kernel that has no double-backward; the math backend
        # (what torch 1.13, the reference's pinned version, always used) does.
        from torch.nn.attention import sdpa_kernel, SDPBackend
        ctx = sdpa_kernel(SDPBackend.MATH)
        ctx.__enter__()

        def losses():
            if replay is not None:
                replay.mode = "val"
            meta_loss = sub.training_step(batch=bv, align=False)
            if replay is not None:
                replay.mode = "train"
            torch.manual_seed(seed + 3)
            meta_train_loss = model.training_step(batch=bt, align=False)
            return meta_loss, meta_train_loss
        ml, mtl = losses()
        out["outer.val_loss"] = ml.detach().numpy()
        gval = torch.autograd.grad(ml, params, retain_graph=True, allow_unused=True)
        for (n, _), gg in zip(sub.named_parameters(), gval):
            if real and n == "item_embedding.weight":
                _sparse_rows(out, "outer.grad_val." + n, gg.numpy())
            else:
                out["outer.grad_val." + n] = gg.numpy().copy()
        hg = model.meta_optimizer.hypergrad.grad(loss_val=ml, loss_train=mtl, aux_params=aux, params=params)
        for (n, _), gg in zip(model.meta_module.named_parameters(), hg):
            out["outer.hypergrad." + n] = gg.detach().numpy().copy()
        ml, mtl = losses()
        for _ in range(2):                                          # two outer steps on the same pair: pins SGD momentum + wd + clip
            model.meta_optimizer.step(val_loss=ml, train_loss=mtl, aux_params=aux, parameters=params, return_grads=False)
            ml, mtl = losses()
            for n, p in model.meta_module.named_parameters():
                out[f"outer.step{_ + 1}." + n] = p.detach().numpy().copy()
        ctx.__exit__(None, None, None)
        tc = config["train"]
        for k in ("meta_learning_rate", "hpo_learning_rate", "meta_weight_decay"):
            out["meta." + k] = np.float64(tc[k])
        out["meta.meta_optimizer"] = np.array(tc["meta_optimizer"])
        out["meta.tau_min"] = np.float64(config["model"]["tau_min"])
        out["meta.sub_model"] = np.array(sub_model)
        out["meta.num_items"] = np.int64(model.num_items)
        smc = sub.config["model"]
        for k in ("head_num", "hidden_size", "layer_num"):
            out["meta." + k] = np.int64(smc.get(k, 0))
        out["meta.layer_norm_eps"] = np.float64(smc.get("layer_norm_eps", 1e-12))
        if replay is not None:
            for k in ("temperature", "cl_weight"):
                out["meta." + k] = np.float64(smc[k])
            for k in ("tau", "gamma", "beta"):                       # ('meta.tau' is the MetaModel's Gumbel temperature)
                out["meta.aug_" + k] = np.float64(smc[k])
            out["meta.augment_type"] = np.array(smc["augment_type"])
            for mode, lst in replay.store.items():
                assert len(lst) == 2, (mode, len(lst))
                for tag, (sq, ln) in zip("ij", lst):
                    full = torch.zeros(sq.shape[0], L, dtype=sq.dtype)
                    full[:, :sq.shape[1]] = sq                        # Item_Crop pads to the longest crop only
                    out[f"view.{mode}.{tag}"], out[f"view.{mode}.{tag}_len"] = full.numpy(), ln.numpy()
        os.chdir(cwd)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        hn = float(torch.sqrt(sum((h ** 2).sum() for h in hg)))
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), weighted loss={float(loss.detach()):.6f}, |hypergrad|={hn:.3e}")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def run_cl_case(out_dir, name, n_items, seqlens, seed, augment_type="item_random"):
    """CL4SRec (model/cl4srec.py:49-73): BCE + cl_weight * InfoNCE between two augmented views.  Dropout 0.  The augmentations are
    random (torch / numpy / random streams): the views the reference drew are RECORDED and stored, so the deterministic part
    (encoder on the views, mean pooling, length-1 filter, InfoNCE, total gradient) is pinned exactly; the augmentation
    distributions are tested separately."""
    import random
    import torch
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        build_dataset(work, n_items, seqlens, rng)
        os.chdir(work)
        from utils import load_config, setup_environment, prepare_datasets, prepare_model
        config = load_config({"model": "CL4SRec", "dataset": "amazon-toys"})
        config["train"]["device"] = "cpu"
        config["data"]["train_file"] = "_ori"
        config["model"]["dropout_rate"] = 0.0
        config["model"]["augment_type"] = augment_type
        setup_environment(config["train"])
        torch.manual_seed(seed)
        ds = prepare_datasets(config)
        model = prepare_model(config, ds)
        model._init_model(ds[0])
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "item_embedding" in n or "item_encoder" in n:
                    continue
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        out = {}
        for k, v in model.state_dict().items():
            out["param." + k] = v.detach().numpy().copy()
        batch = next(iter(ds[0].get_loader(batch_size=len(ds[0]), shuffle=False)))
        model.train()
        torch.manual_seed(seed + 2)
        random.seed(seed + 2)
        np.random.seed(seed + 2)
        batch["neg_item"] = model._neg_sampling(batch)
        for k, v in batch.items():
            out["batch." + k] = v.numpy()
        views = []
        real_aug = model.augmentation_model.augmentation

        class Recorder(torch.nn.Module):
            def forward(self, sequences, seq_lens):
                s, l = real_aug(sequences, seq_lens)
                full = torch.zeros_like(sequences)
                full[:, :s.shape[1]] = s                      # Item_Crop pads to the longest crop only
                views.append((full.clone(), l.clone()))
                return s, l
        model.augmentation_model.augmentation = Recorder()
        model.optimizer.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        (vi, li), (vj, lj) = views
        out["view.i"], out["view.i_len"], out["view.j"], out["view.j_len"] = vi.numpy(), li.numpy(), vj.numpy(), lj.numpy()
        out["out.loss"] = loss.detach().numpy()
        for n, p in model.named_parameters():
            out["grad." + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
        with torch.no_grad():                                     # the contrastive term alone, on the recorded views
            from module import functional as recfn
            enc = model.query_encoder
            oi = recfn.seq_pooling_function(enc({"in_item_id": vi, "seqlen": li}, need_pooling=False), li, pooling_type="mean")
            oj = recfn.seq_pooling_function(enc({"in_item_id": vj, "seqlen": lj}, need_pooling=False), lj, pooling_type="mean")
            keep = batch["seqlen"] != 1
            out["out.view_i_mean"], out["out.view_j_mean"] = oi.numpy(), oj.numpy()
            out["out.cl_loss"] = model.augmentation_model.InfoNCE_loss_fn(oi[keep], oj[keep]).numpy()
            out["out.cl_loss_rows"] = model.augmentation_model.InfoNCE_loss_fn(oi[keep], oj[keep], reduce=False).numpy()
            out["out.bce_loss"] = (loss.detach() - config["model"]["cl_weight"] * torch.from_numpy(out["out.cl_loss"])).numpy()
        model.optimizer.step()
        for n, p in model.named_parameters():
            out["adam1." + n] = p.detach().numpy().copy()
        mc = config["model"]
        out["meta.num_items"] = np.int64(model.num_items)
        for k in ("head_num", "hidden_size", "layer_num"):
            out["meta." + k] = np.int64(mc[k])
        for k in ("layer_norm_eps", "temperature", "cl_weight", "tau", "gamma", "beta"):
            out["meta." + k] = np.float64(mc[k])
        out["meta.augment_type"] = np.array(augment_type)
        out["meta.lr"] = np.float64(config["train"]["learning_rate"])
        os.chdir(cwd)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), loss={float(loss.detach()):.6f}, cl_loss={float(out['out.cl_loss']):.6f}")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


# --------------------------------------------------------------------------- shipped checkpoint + real rows
TOYS_DIR = os.path.join(REF, "dataset", "amazon-toys", "toy")


def build_real_toys(root, seed=2024):
    """The REAL amazon-toys training rows, rebuilt from the shipped seq2pat_data.pth (= every user's chronological items minus the last
    two, dataset/preprocess_amazon.ipynb cell 19) with the row recipe of cell 20.  The two held-out items are unknown: two seeded random
    items stand in for them; they only reach the val / test TARGETS (and the last inputs of the val/test histories) — the train rows do
    not contain them (history = us[:-3], target = us[1:-2]), so train_ori.pth is exact."""
    import torch
    seqs = torch.load(os.path.join(TOYS_DIR, "seq2pat_data.pth"), weights_only=False)
    n_items = max(max(s) for s in seqs) + 1
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "dataset", "amazon-toys", "toy")
    os.makedirs(d, exist_ok=True)
    train, val, test = [], [], []

    def top(seq):                                                  # truncate_or_pad of cell 20
        n = len(seq)
        return (seq[-L:], L) if n > L else (seq + [0] * (L - n), n)
    for u, s in enumerate(seqs, start=1):
        us = (list(s) + rng.integers(1, n_items, size=2).tolist())[-L:]
        h, sl = top(us[:-1])
        test.append([u, h, us[-1], sl, 1, [0] * L, h])
        h, sl = top(us[:-2])
        val.append([u, h, us[-2], sl, 1, [0] * L, h])
        h, sl = top(us[:-3])
        tgt, _ = top(us[-sl - 2:-2])
        train.append([u, h, tgt, sl, [1] * sl + [0] * (L - sl), [0] * L])
    torch.save(train, os.path.join(d, "train_ori.pth"))
    torch.save(val, os.path.join(d, "val.pth"))
    torch.save(test, os.path.join(d, "test.pth"))
    with open(os.path.join(d, "inter.csv"), "w") as f:             # every user and every item id (data/dataset.py:56-65)
        f.write("user_id,item_id,rating,timestamp,domain\n")
        nu = len(seqs)
        for i in range(1, n_items):
            f.write(f"{(i - 1) % nu + 1},{i},1.0,{i},0\n")
        for u in range(1, nu + 1):
            f.write(f"{u},{(u % (n_items - 1)) + 1},1.0,{u},0\n")
    return n_items, len(seqs)


def _sparse_rows(out, key, dense, ref=None):
    """table-shaped arrays travel as (row ids, rows): rows that differ from `ref` (or are non-zero)"""
    dense = np.asarray(dense)
    rows = np.nonzero(np.any(dense != (0 if ref is None else ref), axis=1))[0]
    out[key + ".rows"] = rows.astype(np.int64)
    out[key + ".vals"] = dense[rows].copy()


def run_trained_case(out_dir, name="sasrec_trained_toys", seed=21):
    """SASRec with the SHIPPED trained weights (dataset/amazon-toys/toy/pre-trained_embedding.ckpt, dict format utils/callbacks.py:70-76,
    loaded strict=True like model/basemodel.py:404-407) on the REAL toys rows: the first 256 rows and the real odd tail batch
    (19 412 mod 256 = 212 rows).  training_step / backward / 2 x Adam / topk, dropout 0."""
    import torch
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        n_items, n_users = build_real_toys(work)
        os.chdir(work)
        from utils import load_config, setup_environment, prepare_datasets, prepare_model
        config = load_config({"model": "SASRec", "dataset": "amazon-toys"})
        config["train"]["device"] = "cpu"
        config["data"]["train_file"] = "_ori"
        config["model"]["dropout_rate"] = 0.0
        setup_environment(config["train"])
        torch.manual_seed(seed)
        ds = prepare_datasets(config)
        model = prepare_model(config, ds)
        model._init_model(ds[0])
        ck = torch.load(os.path.join(TOYS_DIR, "pre-trained_embedding.ckpt"), weights_only=False, map_location="cpu")
        model.load_state_dict(ck["parameters"], strict=True)
        assert model.num_items == n_items == 11925 and len(ds[0]) == n_users == 19412
        out = {}
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k, v in sd0.items():
            if k == "query_encoder.item_encoder.weight":            # the tied table, stored once
                assert torch.equal(v, sd0["item_embedding.weight"])
                continue
            out["param." + k] = v.numpy()
        out["ckpt.epoch"] = np.int64(ck["epoch"])
        out["ckpt.metric_keys"] = np.array(sorted(ck["metric"]))
        out["ckpt.keys"] = np.array(sorted(ck))
        out["ckpt.val_ndcg20"] = np.float64(float(ck["metric"]["ndcg@20"]))

        full = next(iter(ds[0].get_loader(batch_size=len(ds[0]), shuffle=False)))
        bs = config["train"]["batch_size"]
        tail0 = (len(ds[0]) // bs) * bs
        model.train()
        for tag, sl in (("b0", slice(0, bs)), ("tail", slice(tail0, len(ds[0])))):
            model.load_state_dict(sd0)
            model.optimizer = model._get_optimizers()
            batch = {k: v[sl].clone() for k, v in full.items()}
            torch.manual_seed(seed + 2)
            batch["neg_item"] = model._neg_sampling(batch)
            for k in ("user_id", "in_item_id", "item_id", "seqlen", "neg_item"):
                out[f"{tag}.batch.{k}"] = batch[k].numpy().astype(np.int32)      # int32 on disk (ids < 2^31); tests widen
            model.optimizer.zero_grad()
            loss, query = model.training_step(batch, reduce=True, return_query=True)
            loss.backward()
            out[f"{tag}.query"] = query.detach().numpy()
            out[f"{tag}.loss"] = loss.detach().numpy()
            with torch.no_grad():
                out[f"{tag}.loss_noreduce"] = model.training_step(batch, reduce=False).numpy()
            for n, p in model.named_parameters():
                g = p.grad.numpy().copy()
                if n == "item_embedding.weight":
                    _sparse_rows(out, f"{tag}.grad.{n}", g)
                else:
                    out[f"{tag}.grad.{n}"] = g
            for step in (1, 2):
                if step == 2:
                    model.optimizer.zero_grad()
                    l2 = model.training_step(batch)
                    l2.backward()
                    out[f"{tag}.loss_step2"] = l2.detach().numpy()
                model.optimizer.step()
                for n, p in model.named_parameters():
                    v = p.detach().numpy().copy()
                    if n == "item_embedding.weight":
                        if step == 2:                                  # the table after both steps only (fixture size)
                            _sparse_rows(out, f"{tag}.adam{step}.{n}", v, ref=sd0[n].numpy())
                    else:
                        out[f"{tag}.adam{step}.{n}"] = v
            print(f"{name}/{tag}: rows {sl.start}..{sl.stop}, loss {float(loss.detach()):.6f}, valid {int((batch['item_id'] != 0).sum())}")

        # ---- eval (trained weights): full-item scorer + top-k on the first 256 validation rows -------------
        model.load_state_dict(sd0)
        model.eval()
        ds[1].set_eval_domain("toy")
        model.set_eval_domain("toy")
        vb = next(iter(ds[1].get_loader(batch_size=256)))
        with torch.no_grad():
            score, items = model.topk(vb, 100, vb["user_hist"])
            q_last = model.forward(vb)
        for kk in ("in_item_id", "item_id", "seqlen", "user_hist"):
            out["eval." + kk] = vb[kk].numpy().astype(np.int32)
        out["eval.topk_score"], out["eval.topk_items"], out["eval.query_last"] = score.numpy(), items.numpy().astype(np.int32), q_last.numpy()
        mc = config["model"]
        out["meta.num_items"], out["meta.embed_dim"] = np.int64(model.num_items), np.int64(mc["embed_dim"])
        for k in ("head_num", "hidden_size", "layer_num"):
            out["meta." + k] = np.int64(mc[k])
        out["meta.layer_norm_eps"] = np.float64(mc.get("layer_norm_eps", 1e-12))
        out["meta.lr"], out["meta.weight_decay"] = np.float64(config["train"]["learning_rate"]), np.float64(config["train"]["weight_decay"])
        out["meta.torch_version"] = np.array(torch.__version__)
        os.chdir(cwd)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from _golden_io import curve_init  # noqa: E402  (ONE definition: the GPU / CPU tests rebuild the same initial parameters)


def run_curve_case(out_dir, name="sasrec_toys_curve", epochs=4, seeds=(31, 32)):
    """END-TO-END statistical pin: the reference's own training loop (BaseModel.training_epoch, basemodel.py:176-200: DataLoader shuffle,
    multinomial negatives, dropout 0.5, Adam) on the REAL toys training rows from a deterministic init, `epochs` epochs, for two RNG
    seeds; the per-epoch mean training losses are stored together with the rows.  The HIP trainer draws other random streams (Philox),
    so its curve is compared statistically: it must lie as close to the reference's curves as they lie to each other."""
    import torch
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        n_items, n_users = build_real_toys(work)
        os.chdir(work)
        from utils import load_config, setup_environment, prepare_datasets, prepare_model
        curves, first_steps, out = [], [], {}
        for seed in seeds:
            config = load_config({"model": "SASRec", "dataset": "amazon-toys"})
            config["train"]["device"] = "cpu"
            config["data"]["train_file"] = "_ori"
            config["train"]["seed"] = seed
            setup_environment(config["train"])
            torch.manual_seed(seed)
            np.random.seed(seed)
            ds = prepare_datasets(config)
            model = prepare_model(config, ds)
            model._init_model(ds[0])
            shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if k != "query_encoder.item_encoder.weight"}
            init = curve_init(shapes, 77)
            with torch.no_grad():
                for k, v in model.named_parameters():
                    v.copy_(torch.from_numpy(init[k]))
            model.train()
            ep = []
            for e in range(epochs):
                outs = model.training_epoch(e)
                ls = [float(o["loss_0"]) for o in outs[0]]
                if e == 0:
                    first_steps.append(ls[:20])
                ep.append(float(np.mean(ls)))
                print(f"{name}: seed {seed} epoch {e}: mean loss {ep[-1]:.5f} ({len(ls)} steps)", flush=True)
            curves.append(ep)
            if "rows.in_item_id" not in out:
                rows = next(iter(ds[0].get_loader(batch_size=len(ds[0]), shuffle=False)))
                assert int(rows["in_item_id"].max()) < 32768
                out["rows.in_item_id"] = rows["in_item_id"].numpy().astype(np.int16)
                out["rows.item_id"] = rows["item_id"].numpy().astype(np.int16)
                out["rows.seqlen"] = rows["seqlen"].numpy().astype(np.int8)
                mc, tc = config["model"], config["train"]
                out["meta.num_items"] = np.int64(model.num_items)
                for k in ("embed_dim", "head_num", "hidden_size", "layer_num"):
                    out["meta." + k] = np.int64(mc[k])
                for k in ("dropout_rate", "layer_norm_eps"):
                    out["meta." + k] = np.float64(mc[k])
                out["meta.lr"], out["meta.weight_decay"], out["meta.batch_size"] = np.float64(tc["learning_rate"]), np.float64(tc["weight_decay"]), np.int64(tc["batch_size"])
                out["meta.init_seed"] = np.int64(77)
        out["curve.epoch_mean_loss"] = np.asarray(curves, np.float64)            # [seeds, epochs]
        out["curve.first_steps"] = np.asarray(first_steps, np.float64)           # [seeds, 20]
        os.chdir(cwd)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); curves {np.round(out['curve.epoch_mean_loss'], 4).tolist()}")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def neg_sampler_stats(out_dir):
    """Pin the *distribution* of basemodel.py:50-61 (uniform on 1..N-1, never PAD)."""
    import torch
    from model.basemodel import BaseModel

    class _M:                       # minimal duck for the unbound method
        fiid = "item_id"
        device = "cpu"
        num_items = 37
        max_seq_len = L
    torch.manual_seed(7)
    b = {"in_item_id": torch.zeros(4000, L, dtype=torch.long), "item_id": torch.zeros(4000, L, dtype=torch.long)}
    neg = BaseModel._neg_sampling(_M, b)
    cnt = torch.bincount(neg.flatten(), minlength=37).numpy()
    b1 = {"in_item_id": torch.zeros(9, L, dtype=torch.long), "item_id": torch.zeros(9, dtype=torch.long)}
    neg1 = BaseModel._neg_sampling(_M, b1)
    np.savez_compressed(os.path.join(out_dir, "neg_sampler_stats.npz"),
                        counts=cnt, shape2d=np.array(neg.shape), shape1d=np.array(neg1.shape),
                        dtype=np.array(str(neg.dtype)))
    print("neg sampler: shape", tuple(neg.shape), tuple(neg1.shape), "count[0] =", cnt[0],
          "min/max over 1..N-1:", cnt[1:].min(), cnt[1:].max())


def loss_module_vectors(out_dir):
    """model/loss_func.py:9-49 by RUNNING both loss classes on fixed score tensors (2-D and 1-D positives, K = 1 and K = 3 negatives,
    padded positions marked -inf exactly as basemodel.py:208 does), with autograd gradients.  Also records that the reference's
    training_step cannot call BPRLoss (basemodel.py:210 passes reduce=; loss_func.py:44 takes none): the TypeError text."""
    import torch
    import model.loss_func as lf
    g = torch.Generator().manual_seed(77)
    out = {}
    for tag, shape, K in (("a", (6, 50), 1), ("b", (9,), 1), ("c", (4, 7), 3)):
        pos = torch.randn(shape, generator=g) * 2.0
        neg = torch.randn(*shape, K, generator=g) * 2.0
        pad = torch.rand(shape, generator=g) < 0.4
        pad.view(-1)[0] = False
        pos = pos.masked_fill(pad, float("-inf"))
        out[f"{tag}.pos"], out[f"{tag}.neg"] = pos.numpy().copy(), neg.numpy().copy()
        for name, mod, kw in (("bce", lf.BinaryCrossEntropyLoss(), {"reduce": True}), ("bce_nr", lf.BinaryCrossEntropyLoss(), {"reduce": False}),
                              ("bpr", lf.BPRLoss(), {})):
            p = pos.clone().requires_grad_(True)
            n = neg.clone().requires_grad_(True)
            loss = mod(p, n, **kw)
            up = torch.ones_like(loss) if loss.dim() == 0 else torch.randn(loss.shape, generator=g)
            (loss * up).sum().backward()
            out[f"{tag}.{name}.loss"] = loss.detach().numpy().copy()
            out[f"{tag}.{name}.up"] = up.numpy().copy()
            out[f"{tag}.{name}.dpos"] = torch.nan_to_num(p.grad, nan=0.0).numpy().copy()     # -inf rows: grad is nan/0 in torch, 0 by definition
            out[f"{tag}.{name}.dneg"] = n.grad.numpy().copy()
    # loss_func.py:32-33, the plain-mean branch of BinaryCrossEntropyLoss: pos [B, L] with neg [B, K] of the SAME rank (no padding mask on
    # the negatives, their term is one scalar mean).  Drawn AFTER a / b / c so that those vectors keep their values.
    pos = torch.randn(5, 11, generator=g) * 2.0
    neg = torch.randn(5, 4, generator=g) * 2.0
    pad = torch.rand(5, 11, generator=g) < 0.4
    pad.view(-1)[0] = False
    pos = pos.masked_fill(pad, float("-inf"))
    out["d.pos"], out["d.neg"] = pos.numpy().copy(), neg.numpy().copy()
    for name, kw in (("bce", {"reduce": True}), ("bce_nr", {"reduce": False})):
        p = pos.clone().requires_grad_(True)
        n = neg.clone().requires_grad_(True)
        loss = lf.BinaryCrossEntropyLoss()(p, n, **kw)
        up = torch.ones_like(loss) if loss.dim() == 0 else torch.randn(loss.shape, generator=g)
        (loss * up).sum().backward()
        out[f"d.{name}.loss"] = loss.detach().numpy().copy()
        out[f"d.{name}.up"] = up.numpy().copy()
        out[f"d.{name}.dpos"] = torch.nan_to_num(p.grad, nan=0.0).numpy().copy()
        out[f"d.{name}.dneg"] = n.grad.numpy().copy()
    try:
        lf.BPRLoss()(torch.zeros(2), torch.zeros(2, 1), reduce=True)
        msg = ""
    except TypeError as e:
        msg = str(e)
    out["bpr.reduce_kwarg_error"] = np.array(msg)
    np.savez_compressed(os.path.join(out_dir, "loss_modules.npz"), **out)
    print("wrote loss_modules.npz; BPRLoss(reduce=) ->", msg)


def run_meta_optimizer_case(out_dir, name="metamodel_optimizers", seed=27):
    """MetaModel._get_meta_optimizers (metamodel.py:59-81) + MetaOptimizer.step's tail (utils/utils.py:242-250: p.grad = g,
    clip_grad_norm_(10), meta_optimizer.step()) by RUNNING the reference with `meta_optimizer` = adam / adagrad / rmsprop / an unknown
    name (its else branch: Adam WITH meta_weight_decay) / sgd: the reference builds its own torch optimizer over meta_module + tau, its
    own MetaOptimizer.step runs with Hypergrad.grad replaced by three fixed gradient sets (the second one large enough to be clipped);
    stored: the initial meta parameters, the gradients, the parameters after each step; 'sparse_adam': the error text of its first step."""
    import torch
    rng = np.random.default_rng(seed)
    work = tempfile.mkdtemp(prefix="dr4sr_golden_")
    cwd = os.getcwd()
    out = {}
    try:
        os.symlink(os.path.join(REF, "configs"), os.path.join(work, "configs"))
        build_dataset(work, 61, [3, 5, 8, 2], rng)
        os.chdir(work)
        import utils as rutils
        import model.metamodel as mm
        orig_load = rutils.load_config

        def patched_load(cfg):                       # metamodel.py:41-50 hard-codes device 0 for the sub-model
            c = orig_load(cfg)
            c["train"]["device"] = "cpu"
            c["data"]["train_file"] = "_ori"
            return c
        mm.load_config = patched_load

        def register_sub_model(self):                 # metamodel.py:41-50 with the device line dropped
            sc = patched_load({"dataset": self.config["data"]["dataset"], "model": self.config["model"]["sub_model"]})
            return rutils.get_model_class(sc["model"])(sc, self.dataset_list)
        mm.MetaModel._register_sub_model = register_sub_model
        gen = torch.Generator().manual_seed(seed)
        grads, phi0 = None, None
        for opt_name in ("adam", "adagrad", "rmsprop", "lamb", "sgd", "sparse_adam"):
            config = patched_load({"model": "MetaModel", "dataset": "amazon-toys"})
            config["model"]["sub_model"] = "SASRec"
            config["train"]["meta_optimizer"] = opt_name
            rutils.setup_environment(config["train"])
            torch.manual_seed(seed)
            ds = rutils.prepare_datasets(config)
            model = rutils.prepare_model(config, ds)
            model._init_model(ds[0])
            aux = list(model.meta_module.parameters())
            names = [n for n, _ in model.meta_module.named_parameters()]
            if grads is None:
                phi0 = [p.detach().clone() for p in aux]
                scales = (0.05, 3.0, 0.2)            # |g| of step 2 is far above max_grad_norm = 10: clipped
                grads = [[sc * torch.randn(p.shape, generator=gen) for p in aux] for sc in scales]
                for n, p in zip(names, phi0):
                    out["phi0." + n] = p.numpy().copy()
                for k, gs in enumerate(grads):
                    for n, gg in zip(names, gs):
                        out[f"grad{k + 1}." + n] = gg.numpy().copy()
                    out[f"grad{k + 1}.norm"] = np.float64(torch.sqrt(sum((gg.double() ** 2).sum() for gg in gs)))
                tc = config["train"]
                for k in ("meta_learning_rate", "hpo_learning_rate", "meta_weight_decay"):
                    out["meta." + k] = np.float64(tc[k])
            with torch.no_grad():
                for p, p0 in zip(aux, phi0):
                    p.copy_(p0)
            mo = model.meta_optimizer
            out[f"{opt_name}.torch_class"] = np.array(type(mo.meta_optimizer).__name__)
            for k, gs in enumerate(grads):
                mo.hypergrad.grad = lambda gs=gs, **kw: [gg.clone() for gg in gs]
                try:
                    mo.step(train_loss=None, val_loss=None, parameters=None, aux_params=aux)
                except Exception as e:          # noqa: BLE001 — sparse_adam: torch refuses dense gradients
                    out[f"{opt_name}.error"] = np.array(f"{type(e).__name__}: {e}")
                    break
                for n, p in zip(names, aux):
                    out[f"{opt_name}.step{k + 1}." + n] = p.detach().numpy().copy()
            assert float(model.tau.detach()) == 10.0                 # tau is in the optimizer's list but never has a gradient
        os.chdir(cwd)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); " + ", ".join(
            f"{k.split('.')[0]}={out[k]}" for k in out if k.endswith("torch_class")) + "; sparse_adam: " + str(out.get("sparse_adam.error")))
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    if not os.path.isdir(REF):
        sys.exit("reference not present; golden vectors can only be regenerated in the build container")
    _install_stubs()
    sys.path.insert(0, REF)
    seqlens = [1, 2, 3, 5, 8, 13, 21, 34, 47, 49, 50, 4, 2, 50]
    only = os.environ.get("GOLDEN_ONLY")
    if only == "loss":
        loss_module_vectors(out_dir)
        return
    if only == "cl":
        run_cl_case(out_dir, "cl4srec_d64", n_items=173, seqlens=seqlens, seed=16)
        return
    if only == "meta":
        run_meta_case(out_dir, "metamodel_sasrec", "SASRec", n_items=151, seqlens=seqlens, seed=15)
        return
    if only == "curve":
        run_curve_case(out_dir)
        return
    if only == "meta_opt":
        run_meta_optimizer_case(out_dir)
        return
    if only == "meta_cl":
        run_meta_case(out_dir, "metamodel_cl4srec", "CL4SRec", n_items=137, seqlens=seqlens, seed=23)
        return
    if only == "trained":
        run_trained_case(out_dir)
        return
    if only == "meta_trained":
        run_meta_case(out_dir, "metamodel_trained_toys", "SASRec", n_items=None, seqlens=None, seed=22, real=True)
        return
    if only == "fmlp":
        run_case(out_dir, "fmlp_d64", "FMLP", n_items=113, seqlens=[1, 3, 6, 50, 2], embed_dim=64, seed=14)
        return
    run_case(out_dir, "sasrec_d64", "SASRec", n_items=211, seqlens=seqlens, embed_dim=64, seed=11)
    run_case(out_dir, "sasrec_d128", "SASRec", n_items=97, seqlens=seqlens[:9], embed_dim=128, seed=12)
    # GRU4Rec: hidden 128 instead of configs/gru4rec.yaml's 256 only to keep the fixture small
    run_case(out_dir, "gru4rec_d64", "GRU4Rec", n_items=131, seqlens=seqlens[:10], embed_dim=64, seed=13,
             overrides={"model": {"hidden_size": 128}})
    run_case(out_dir, "fmlp_d64", "FMLP", n_items=113, seqlens=[1, 3, 6, 50, 2], embed_dim=64, seed=14)
    run_meta_case(out_dir, "metamodel_sasrec", "SASRec", n_items=151, seqlens=seqlens, seed=15)
    run_cl_case(out_dir, "cl4srec_d64", n_items=173, seqlens=seqlens, seed=16)
    neg_sampler_stats(out_dir)
    loss_module_vectors(out_dir)
    run_trained_case(out_dir)
    run_meta_case(out_dir, "metamodel_trained_toys", "SASRec", n_items=None, seqlens=None, seed=22, real=True)
    run_meta_case(out_dir, "metamodel_cl4srec", "CL4SRec", n_items=137, seqlens=seqlens, seed=23)
    run_curve_case(out_dir)
    run_meta_optimizer_case(out_dir)


if __name__ == "__main__":
    main()
