"""GPU parity of the CL4SRec path: augmentation kernel (legality + distribution), mean pooling, InfoNCE fwd/bwd, and the whole
training step of dr4sr_amd.model.cl4srec.CL4SRec vs golden vectors made by RUNNING the reference (views recorded)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cl4srec_oracle as CO  # noqa: E402
from tests.test_cl_oracle import load_cl  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_config(n_items, n_rows=300, batch=64, epochs=2, dropout=0.0, augment="item_random"):
    return {
        "data": {"dataset": "synthetic-toys", "domain_name_list": ["toy"], "max_seq_len": 50, "dataset_class": "synthetic",
                 "train_file": "", "n_items": n_items, "n_rows": n_rows, "n_eval_rows": 128, "seed": 5},
        "model": {"model": "CL4SRec", "embed_dim": 64, "loss_fn": "bce", "hidden_size": 128, "layer_num": 2, "head_num": 2,
                  "dropout_rate": dropout, "activation": "gelu", "layer_norm_eps": 1e-12, "augment_type": augment,
                  "temperature": 1.0, "cl_weight": 0.1, "tau": 0.2, "gamma": 0.7, "beta": 0.2},
        "train": {"batch_size": batch, "early_stop_mode": "max", "early_stop_patience": 20, "epochs": epochs, "device": "cuda",
                  "optimizer": "adam", "learning_rate": 0.001, "weight_decay": 0, "num_neg": 1, "seed": 2023, "hip_graph": True},
        "eval": {"batch_size": 128, "cutoff": [20, 10], "val_metrics": ["ndcg", "recall"], "test_metrics": ["ndcg", "recall"],
                 "topk": 100, "save_path": "./saved/"},
    }


def test_augment_kernel_legal_and_well_distributed():
    from dr4sr_amd.module.data_augmentation import Item_Crop, Item_Mask, Item_Random, Item_Reorder
    torch.manual_seed(0)
    B, L, N = 4096, 50, 1000
    sl = torch.randint(1, 51, (B,))
    seq = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        seq[b, :sl[b]] = torch.arange(1, int(sl[b]) + 1) + 100 * (b % 7)        # distinct items inside a sequence
    seqd, sld = seq.cuda(), sl.cuda()
    tau, gamma, beta = 0.2, 0.7, 0.2
    # crop
    out, ol = Item_Crop(tau)(seqd, sld)
    out, ol = out.cpu(), ol.cpu()
    starts = []
    for b in range(B):
        n, k = int(sl[b]), int(ol[b])
        assert k == CO.crop_len(n, tau) and (out[b, k:] == 0).all()
        s0 = int(out[b, 0] - seq[b, 0])
        assert 0 <= s0 <= n - k and torch.equal(out[b, :k], seq[b, s0:s0 + k])
        if n == 50:
            starts.append(s0)
    assert len(set(starts)) > 30                                                 # uniform over 41 start positions
    # mask
    out, ol = Item_Mask(N, gamma)(seqd, sld)
    out, ol = out.cpu(), ol.cpu()
    assert torch.equal(ol, sl)
    hit = torch.zeros(50)
    for b in range(B):
        n = int(sl[b])
        m = out[b, :n] == N
        assert int(m.sum()) == CO.mask_count(n, gamma) and torch.equal(out[b, :n][~m], seq[b, :n][~m]) and (out[b, n:] == 0).all()
        if n == 50:
            hit += m.float()
    assert hit.min() > 0.4 * hit.max()                                           # every position gets masked about equally often
    # reorder
    out, ol = Item_Reorder(beta)(seqd, sld)
    out, ol = out.cpu(), ol.cpu()
    moved = 0
    for b in range(B):
        n, k = int(sl[b]), CO.reorder_len(int(sl[b]), beta)
        assert sorted(out[b, :n].tolist()) == sorted(seq[b, :n].tolist()) and (out[b, n:] == 0).all()
        diff = (out[b, :n] != seq[b, :n]).nonzero().flatten()
        if len(diff):
            assert int(diff.max() - diff.min()) < k                              # all moves inside one window of length k
            moved += 1
    assert moved > B // 2
    # random: the method is drawn once per CALL (data_augmentation.py:95) and all three occur
    aug = Item_Random(N, tau, gamma, beta)
    kinds = set()
    for _ in range(24):
        o, l2 = aug(seqd[:64], sld[:64])
        o, l2 = o.cpu(), l2.cpu()
        big = sl[:64] >= 10
        if (l2[big] != sl[:64][big]).all():
            kinds.add("crop")
        elif (o == N).any():
            kinds.add("mask")
        else:
            kinds.add("reorder")
    assert kinds == {"crop", "mask", "reorder"}


def test_infonce_kernels_match_oracle():
    from dr4sr_amd.module.data_augmentation import InfoNCELoss
    torch.manual_seed(1)
    for B, D, T in ((37, 64, 1.0), (130, 128, 0.5)):
        xi, xj = torch.randn(B, D) * 0.5, torch.randn(B, D) * 0.5
        valid = torch.rand(B) > 0.15
        a, b = xi.clone().requires_grad_(True), xj.clone().requires_grad_(True)
        ref = CO.infonce(a[valid], b[valid], T)
        ref.backward()
        ad, bd = xi.cuda().requires_grad_(True), xj.cuda().requires_grad_(True)
        loss = InfoNCELoss(T)(ad, bd, valid=valid.cuda())
        loss.backward()
        assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
        assert rel(ad.grad.cpu().numpy(), a.grad.numpy()) < 2e-4 and rel(bd.grad.cpu().numpy(), b.grad.numpy()) < 2e-4
        rows = InfoNCELoss(T)(xi.cuda(), xj.cuda(), reduce=False, valid=valid.cuda())
        np.testing.assert_allclose(rows.cpu().numpy(), CO.infonce(xi[valid], xj[valid], T, reduce=False).numpy(), rtol=2e-4, atol=1e-7)


def test_cl4srec_training_step_matches_reference(golden_dir, monkeypatch):
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    g, p, batch, views, cfg = load_cl(golden_dir)
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    config = make_config(int(g["meta.num_items"]))
    seed_everything(config["train"]["seed"])
    ds = prepare_datasets(config)
    model = prepare_model(config, ds)
    model._init_model(ds[0])
    ref_sd = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    assert set(model.state_dict()) == set(ref_sd)
    model.load_state_dict(ref_sd, strict=True)
    dev = model.device
    bd = {k: v.to(dev) for k, v in batch.items()}
    # mean pooling of the views == the reference's need_pooling=False + seq_pooling_function('mean')
    model.train()
    (vi, li), (vj, lj) = views
    with torch.no_grad():
        oi = model.query_encoder({"in_item_id": vi.to(dev), "seqlen": li.to(dev)}, need_pooling=False, slot=1, pooling="mean")
    assert rel(oi.cpu().numpy(), g["out.view_i_mean"]) < 2e-4

    class Replay(torch.nn.Module):                                   # feed the views the reference drew
        def __init__(self):
            super().__init__()
            self.k = 0

        def forward(self, sequences, seq_lens):
            v = views[self.k % 2]
            self.k += 1
            return v[0].to(dev), v[1].to(dev)
    model.augmentation_model.augmentation = Replay()
    model.optimizer.zero_grad()
    loss = model.training_step(bd)
    loss.backward()
    assert abs(float(loss) - float(g["out.loss"])) < 3e-5
    for n, prm in model.named_parameters():
        ref = g["grad." + n]
        assert rel(prm.grad.cpu().numpy(), ref) < 3e-4, n
    model.optimizer.step()
    for n, prm in model.named_parameters():
        ref, gr = g["adam1." + n], g["grad." + n]
        d = np.abs(prm.detach().cpu().numpy() - ref)
        well = np.abs(gr) > 1e-5                                     # Adam is ill-conditioned where |g| ~ eps
        assert d[well].max(initial=0) < 2e-5 and d.max() < 2.1e-3, n


def test_cl4srec_fit_end_to_end(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd import quickstart
    out = quickstart.run(make_config(150, n_rows=300, batch=64, epochs=2, dropout=0.5))
    assert {"ndcg@20", "recall@20"} <= set(out) and all(np.isfinite(v) for v in out.values())


def test_cl4srec_graph_replayed_epoch_equals_eager_epoch(monkeypatch):
    """the HIP-graph replay of the API step (BaseModel._api_step_graph: device-side call counters for the negative sampler and the
    augmentations) trains exactly like the eager loop: same batches, same draws, same dropout masks"""
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    res = []
    for graph in (False, True):
        config = make_config(150, n_rows=200, batch=64, epochs=1, dropout=0.5)
        config["train"]["hip_graph"] = graph
        seed_everything(config["train"]["seed"])
        ds = prepare_datasets(config)
        model = prepare_model(config, ds)
        model._init_model(ds[0])
        model.train()
        assert model._api_graph_ok() == graph
        loader = ds[0].get_loader(shuffle=False)
        losses = []
        for _ in range(2):                                  # 2 epochs x (3 full batches + 1 partial): graphs are re-used across epochs
            for batch in loader:
                if graph:
                    losses.append(float(model._api_step_graph(batch)))
                else:
                    losses.append(float(model._api_step_body(batch)))
        res.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    (la, pa), (lb, pb) = res
    assert len(la) == 8 and np.allclose(la, lb, rtol=2e-4, atol=1e-5), (la, lb)
    for n in pa:
        assert float((pa[n] - pb[n]).abs().max()) < 5e-4, n     # 8 Adam steps of lr 1e-3: fp32 atomics order, nothing systematic


@pytest.mark.parametrize("dropout,two_pass", [(0.0, False), (0.0, True), (0.5, True)])
def test_cl4srec_direct_step_equals_autograd_step(monkeypatch, dropout, two_pass):
    """the step body the captured graph replays composes the C-ABI calls directly (fused main pass on the batch's negatives, InfoNCE
    backward scaled on the device, encoder backward passes of the two views accumulating into the flat gradient, Adam dividing by
    n_valid); DR4SR_CL_AUTOGRAD=1 runs the reference-shaped loop body (training_step -> loss.backward() -> optimizer.step()): same
    negatives, same views, same dropout masks, same losses and parameters.  By default the direct body encodes the two views as ONE
    batch of 2B sequences (exactly the same arithmetic; with dropout the masks come from one stream instead of two, so the
    bit-level comparison with dropout on uses DR4SR_CL_TWO_PASS = one pass per view)"""
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    if two_pass:
        monkeypatch.setenv("DR4SR_CL_TWO_PASS", "1")
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    res = []
    for autograd in (True, False):
        if autograd:
            monkeypatch.setenv("DR4SR_CL_AUTOGRAD", "1")
        else:
            monkeypatch.delenv("DR4SR_CL_AUTOGRAD")
        config = make_config(150, n_rows=200, batch=64, epochs=1, dropout=dropout)
        config["train"]["hip_graph"] = False
        seed_everything(config["train"]["seed"])
        ds = prepare_datasets(config)
        model = prepare_model(config, ds)
        model._init_model(ds[0])
        model.train()
        assert model._direct_step_ok() == (not autograd)
        losses = [float(model._api_step_body(batch)) for _ in range(2) for batch in ds[0].get_loader(shuffle=False)]
        res.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}))
    (la, pa), (lb, pb) = res
    assert len(la) == 8 and np.allclose(la, lb, rtol=2e-4, atol=1e-5), (la, lb)
    for n in pa:
        assert float((pa[n] - pb[n]).abs().max()) < 5e-4, n


@pytest.mark.parametrize("cls,kw", [("Item_Crop", dict(tao=0.2)), ("Item_Mask", dict(mask_id=1000, gamma=0.7)),
                                    ("Item_Reorder", dict(beta=0.2)), ("Item_Random", dict(mask_id=1000))])
def test_two_views_in_one_launch_equal_two_calls(cls, kw):
    """dr4sr_cl_augment2_dev (both views of a CL4SRec step from ONE launch, written into the halves of one [2B, L] tensor) draws what
    two consecutive dr4sr_cl_augment_dev calls draw (device call counter: the captured-step form)"""
    from dr4sr_amd.module import data_augmentation as DA
    torch.manual_seed(3)
    B, L = 300, 50
    sl = torch.randint(1, 51, (B,), device="cuda")
    seq = torch.randint(1, 900, (B, L), device="cuda") * (torch.arange(L, device="cuda")[None, :] < sl[:, None])
    a, b = getattr(DA, cls)(**kw), getattr(DA, cls)(**kw)
    for aug in (a, b):
        aug.step_dev = torch.full((1,), 7, dtype=torch.int32, device="cuda")
        aug.begin_step()
    (v1, l1), (v2, l2) = a.two_views(seq, sl)
    w1, m1 = b(seq, sl)
    w2, m2 = b(seq, sl)
    assert v1.data_ptr() + v1.numel() * 8 == v2.data_ptr()              # halves of one tensor: encodable as one batch of 2B sequences
    assert torch.equal(v1, w1) and torch.equal(l1, m1) and torch.equal(v2, w2) and torch.equal(l2, m2)
    assert not torch.equal(v1, v2) or cls == "Item_Reorder"             # two different draws (a reorder of tiny segments may coincide)
    a.end_step(); b.end_step()
    assert int(a.step_dev) == int(b.step_dev) == 9


@pytest.mark.parametrize("tail", [20, 37])
def test_cl4srec_data_parallel_equals_single_process(tail):
    """round 4: CL4SRec under DP.  InfoNCE's negatives are the other rows of the BATCH, so the ranks all-gather their
    pooled views (+ n_valid + the length-1 mask), evaluate the loss of the GLOBAL batch and back-propagate their own rows
    (CL4SRec._cl_term); 2 ranks sharing the GPU over the gloo transport == one process on the concatenated batches, same negatives,
    same views (tools/dp_cl_check.py).  tail = rows of the ragged last batch: 20 -> slices of 20 and 0 (an empty rank still takes
    part in the gather), 37 -> 32 and 5."""
    from _launch import report, torchrun
    out = torchrun(2, "tools/dp_cl_check.py", dict(DR4SR_DP_BACKEND="gloo", DP_CL_TAIL=str(tail)), timeout=400)
    lines = [l for l in out.stdout.splitlines() if l.startswith("DP_CL_CHECK")]
    err = out.stdout[out.stdout.find("DP_CL_ERROR"):][:3000] if "DP_CL_ERROR" in out.stdout else report(out)
    assert out.returncode == 0 and len(lines) == 2, err
    print(lines[0])
    assert "replica checksums equal: True" in lines[1]


def _cl_model(monkeypatch, dropout, n_rows=200, batch=64, **train):
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    config = make_config(150, n_rows=n_rows, batch=batch, epochs=1, dropout=dropout)
    config["train"].update(train)
    seed_everything(config["train"]["seed"])
    ds = prepare_datasets(config)
    model = prepare_model(config, ds)
    model._init_model(ds[0])
    model.train()
    return ds, model


@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_cl4srec_rows_step_equals_api_step_body(monkeypatch, dropout):
    """round 4: the step the fused CL4SRec epoch replays works on rows[] of the DATASET tensors (main pass through the plan's rows
    indirection, views drawn by dr4sr_cl_augment2_rows_dev, length-1 mask by dr4sr_cl_prepare_rows) — no batch tensor exists.  With the
    same negatives injected it must equal the per-batch step body (_api_step_body on the gathered batch): same views (the device call
    counter continues the host one), same dropout masks, same losses and parameters; DR4SR_CL_TWO_PASS keeps the bodies' dropout
    streams identical when dropout is on"""
    if dropout > 0:
        monkeypatch.setenv("DR4SR_CL_TWO_PASS", "1")
    res = []
    for mode in ("batch", "rows"):
        ds, model = _cl_model(monkeypatch, dropout)
        eng = model.engine
        f = ds[0].get_loader(shuffle=False).fields
        n, B, L = int(f["seqlen"].shape[0]), 64, model.max_seq_len
        gen = torch.Generator().manual_seed(11)
        losses = []
        if mode == "rows":
            model._api_graph_begin()                           # device call counter of the augmentations (continues the host count)
            rows_buf = torch.zeros(B, dtype=torch.int64, device=model.device)
        for i in range(0, n, B):
            rows = torch.arange(i, min(i + B, n), device=model.device)
            bl = int(rows.shape[0])
            negs = torch.randint(1, model.num_items, (bl, L, 1), generator=gen).to(model.device)
            if mode == "batch":
                model._neg_sampling = lambda b, negs=negs: negs
                batch = {k: f[k].index_select(0, rows) for k in ("in_item_id", "item_id", "seqlen", "user_id")}
                losses.append(float(model._api_step_body(batch)))
            else:
                rows_buf[:bl].copy_(rows)
                plan = eng.make_plan(f["in_item_id"], f["item_id"], f["seqlen"], rows=rows_buf[:bl], neg_item=negs.view(-1).contiguous(),
                                     sample_neg=False)
                model._cl_rows_step(plan, f, rows_buf[:bl])
                losses.append(float(eng.grads[eng.n_params + 1] / eng.grads[eng.n_params]))
        res.append((losses, eng.params.clone()))
    (la, pa), (lb, pb) = res
    assert len(la) == 4 and np.allclose(la, lb, rtol=1e-5, atol=1e-6), (la, lb)
    assert float((pa - pb).abs().max()) < 5e-4, float((pa - pb).abs().max())      # 4 Adam steps of lr 1e-3: fp32 atomics order where |g| ~ eps


def test_cl4srec_fused_epoch_trains_like_the_per_batch_graphs(monkeypatch):
    """fit()'s epoch with NO per-step host work (CL4SRec._fused_cl_epoch: batch selection, negatives, views, loss log on the device,
    k steps per graph, ragged last batch) against the per-batch graphs (train.cl_fused_epoch: false): other negative / shuffle
    streams, so the comparison is statistical — same loss level after the same number of epochs, and it decreases"""
    curves = {}
    for fused in (True, False):
        ds, model = _cl_model(monkeypatch, 0.5, n_rows=1000 + 37, batch=128, cl_fused_epoch=fused, steps_per_graph=3)
        assert model._fused_cl_ok() == fused
        means = []
        for ep in range(6):
            out = model.training_epoch(ep)
            l0 = torch.cat([o["loss_0"].reshape(-1).float() for o in out[0]])      # one entry per step (one dict per epoch or per step)
            assert l0.numel() == 9 and bool(torch.isfinite(l0).all())
            means.append(float(l0.float().mean()))
        curves[fused] = means
        if fused:
            assert any(isinstance(k, tuple) and k and k[0] == "cl_rows" for k in model._graphs)
            assert int(model.engine.state[0]) == 6 * 9
    a, b = curves[True], curves[False]
    assert a[-1] < a[0] and b[-1] < b[0], (a, b)
    assert abs(a[-1] - b[-1]) < 0.03 * b[-1] and abs(a[0] - b[0]) < 0.02 * b[0], (a, b)
