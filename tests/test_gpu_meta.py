"""GPU parity of the MetaModel (DR4SR+) path: meta-module selection kernels, weighted inner step, finite-difference
hyper-gradient and meta SGD step vs golden vectors made by RUNNING the reference (tests/golden/metamodel_sasrec.npz)
and vs oracle/metamodel_oracle.py.  Everything goes through the C ABI (dr4sr_meta_* / dr4sr_fd_*)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import metamodel_oracle as MO  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = MO.META_NAMES


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def make_config(n_items, sub="SASRec", dropout=0.0, n_rows=400, batch=64, epochs=2, warmup=-1, interval=2):
    return {
        "data": {"dataset": "synthetic-toys", "domain_name_list": ["toy"], "max_seq_len": 50, "dataset_class": "synthetic",
                 "train_file": "", "n_items": n_items, "n_rows": n_rows, "n_eval_rows": 128, "seed": 5},
        "model": {"model": "MetaModel", "sub_model": sub, "embed_dim": 64, "loss_fn": "bce", "hidden_size": 128, "layer_num": 2,
                  "head_num": 2, "dropout_rate": dropout, "activation": "gelu", "layer_norm_eps": 1e-12, "tau_min": 1,
                  "sub_overrides": {"model": {"dropout_rate": dropout}}},
        "train": {"batch_size": batch, "early_stop_mode": "max", "early_stop_patience": 20, "epochs": epochs, "device": "cuda",
                  "optimizer": "adam", "learning_rate": 0.001, "weight_decay": 0, "num_neg": 1, "seed": 2023, "hip_graph": True,
                  "interval": interval, "meta_optimizer": "sgd", "meta_learning_rate": 0.001, "hpo_learning_rate": 0.001,
                  "meta_weight_decay": 0.001, "descent_step": 30, "warmup_epoch": warmup, "hypergrad_rel_step": 5e-4},
        "eval": {"batch_size": 128, "cutoff": [20, 10], "val_metrics": ["ndcg", "recall"], "test_metrics": ["ndcg", "recall"],
                 "topk": 100, "save_path": "./saved/"},
    }


def build(cfg, monkeypatch):
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd.utils import prepare_datasets, prepare_model, seed_everything
    seed_everything(cfg["train"]["seed"])
    ds = prepare_datasets(cfg)
    model = prepare_model(cfg, ds)
    model._init_model(ds[0])
    return ds, model


def load_golden(golden_dir, model):
    z = np.load(os.path.join(golden_dir, "metamodel_sasrec.npz"))
    sub_sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    model.sub_model.load_state_dict(sub_sd, strict=True)
    model.meta_module.load_state_dict({k[11:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("meta_param.")}, strict=True)
    dev = model.device
    bt = {k[6:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("train.")}
    bv = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("val.")}
    model._gumbel = torch.from_numpy(z["inner.gumbel"]).to(dev).reshape(-1, 2).contiguous()
    return z, bt, bv


def test_select_kernels_match_oracle():
    from dr4sr_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    B, L, D = 9, 50, 64
    nphi = int(lib.dr4sr_meta_param_count(D))
    assert nphi == D * D + D + 2 * D + 2
    meta = {"0.weight": torch.randn(D, D) * 0.2, "0.bias": torch.randn(D) * 0.1, "2.weight": torch.randn(2, D) * 0.3,
            "2.bias": torch.randn(2) * 0.1}
    phi = torch.cat([meta[k].reshape(-1) for k in NAMES]).cuda()
    q = torch.randn(B, L, D)
    gum = -torch.empty(B, L, 2).exponential_().log()
    tgt = torch.randint(1, 100, (B, L))
    tgt[:, 37:] = 0
    tgt[2, 5:] = 0
    uid = torch.arange(1, B + 1)
    uid[4] = 0
    tau = 3.0
    up = torch.randn(B, L)
    # oracle
    qo = q.clone().requires_grad_(True)
    mo = {k: v.clone().requires_grad_(True) for k, v in meta.items()}
    w_ref = MO.mask_weight(MO.selection(qo, mo, gum, tau, 1.0), uid, tgt)
    (w_ref * up).sum().backward()
    # kernels
    qd, gd, td, ud, upd = q.cuda(), gum.cuda().reshape(-1, 2).contiguous(), tgt.cuda(), uid.cuda(), up.cuda().reshape(-1).contiguous()
    w = torch.empty(B * L, device="cuda")
    gate = torch.zeros(B * L, dtype=torch.int64, device="cuda")
    st = _lib.cur_stream()
    _lib.check(lib.dr4sr_meta_select_fwd(_lib.ptr(qd), _lib.ptr(phi), _lib.ptr(gd), 1, 0, None, tau, _lib.ptr(ud), _lib.ptr(td), B, L, D,
                                         None, _lib.ptr(gate), _lib.ptr(w), st), "fwd")
    np.testing.assert_allclose(w.cpu().numpy().reshape(B, L), w_ref.detach().numpy(), rtol=2e-5, atol=1e-6)
    dq = torch.zeros(B, L, D, device="cuda")
    dphi = torch.zeros(nphi, device="cuda")
    ws = torch.empty(int(lib.dr4sr_meta_select_workspace_floats(B * L)), device="cuda")
    for _ in range(2):                                   # accumulating semantics: two calls = twice the gradient
        _lib.check(lib.dr4sr_meta_select_bwd(_lib.ptr(qd), _lib.ptr(phi), _lib.ptr(gd), 1, 0, None, tau, _lib.ptr(ud), _lib.ptr(td), B, L, D,
                                             None, _lib.ptr(upd), None, _lib.ptr(dq), _lib.ptr(dphi), _lib.ptr(ws), st), "bwd")
    assert rel(dq.cpu().numpy() / 2, qo.grad.numpy()) < 1e-5
    ref_phi = torch.cat([mo[k].grad.reshape(-1) for k in NAMES]).numpy()
    assert rel(dphi.cpu().numpy() / 2, ref_phi) < 1e-5
    # the gate word equals (pre > 0) per unit, and a frozen gate reproduces pre * gate at a shifted query
    pre = (q @ meta["0.weight"].T + meta["0.bias"]) > 0
    bits = (gate.cpu().view(B, L, 1) >> torch.arange(64).view(1, 1, 64)) & 1
    valid = tgt != 0
    assert torch.equal(bits[valid].bool(), pre[valid])
    q2 = q + 0.05 * torch.randn_like(q)
    w2 = torch.empty(B * L, device="cuda")
    _lib.check(lib.dr4sr_meta_select_fwd(_lib.ptr(q2.cuda()), _lib.ptr(phi), _lib.ptr(gd), 1, 0, None, tau, _lib.ptr(ud), _lib.ptr(td), B, L,
                                         D, _lib.ptr(gate), None, _lib.ptr(w2), st), "fwd frozen")
    w2_ref = MO.mask_weight(MO.selection(q2, meta, gum, tau, 1.0, relu_gate=pre.float()), uid, tgt)
    np.testing.assert_allclose(w2.cpu().numpy().reshape(B, L), w2_ref.numpy(), rtol=2e-5, atol=1e-6)
    # in-kernel Gumbel noise: weights in (0,1), reproducible per (seed, step), different across steps
    wa, wb, wc = (torch.empty(B * L, device="cuda") for _ in range(3))
    for out, step in ((wa, 7), (wb, 7), (wc, 8)):
        _lib.check(lib.dr4sr_meta_select_fwd(_lib.ptr(qd), _lib.ptr(phi), None, 11, step, None, tau, None, _lib.ptr(td), B, L, D, None, None,
                                             _lib.ptr(out), st), "fwd philox")
    wd = torch.empty(B * L, device="cuda")                     # the step read from a device word (graph replays)
    sdev = torch.tensor([8], dtype=torch.int32, device="cuda")
    _lib.check(lib.dr4sr_meta_select_fwd(_lib.ptr(qd), _lib.ptr(phi), None, 11, 0, _lib.ptr(sdev), tau, None, _lib.ptr(td), B, L, D, None,
                                         None, _lib.ptr(wd), st), "fwd philox dev step")
    assert torch.equal(wd, wc)
    v = valid.reshape(-1).cuda()
    assert torch.equal(wa, wb) and not torch.equal(wa, wc)
    assert float(wa[v].min()) > 0 and float(wa[v].max()) < 1 and float(wa[~v].abs().max()) == 0


def test_inner_weighted_step_matches_reference(golden_dir, monkeypatch):
    z = np.load(os.path.join(golden_dir, "metamodel_sasrec.npz"))
    ds, model = build(make_config(int(z["meta.num_items"])), monkeypatch)
    z, bt, bv = load_golden(golden_dir, model)
    assert {"tau"} | {"meta_module." + k for k in NAMES} <= set(model.state_dict())
    model.train()
    sub, eng = model.sub_model, model.engine
    # API path: loss = model.training_step(batch); loss.backward()   (metamodel.py:105-113)
    sub.optimizer.zero_grad()
    model.meta_optimizer.zero_grad()
    loss = model.training_step(batch=bt, align=False)
    loss.backward()
    assert abs(float(loss) - float(z["inner.loss"])) < 3e-6 * max(1.0, abs(float(z["inner.loss"])))
    for n, p in sub.named_parameters():
        ref = z["inner.grad." + n]
        assert rel(p.grad.cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, n
    for n, p in model.meta_module.named_parameters():
        assert rel(p.grad.cpu().numpy(), z["inner.meta_grad." + n]) < 3e-4, n
    # fused path: same kernels without autograd, un-normalised sums + tail
    w, lp = model._weighted_fwd_bwd(bt)
    nv = float(eng.grads[eng.n_params])
    np.testing.assert_allclose(w.cpu().numpy().reshape(z["inner.weight"].shape), z["inner.weight"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose((lp / nv).cpu().numpy().reshape(z["inner.loss_pos"].shape), z["inner.loss_pos"], rtol=2e-4, atol=1e-7)
    assert abs(float(eng.grads[eng.n_params + 1]) / nv - float(z["inner.loss"])) < 3e-6
    for n, p in sub.named_parameters():
        ref = z["inner.grad." + n]
        assert rel((p.grad / nv).cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, n
    for n, p in model.meta_module.named_parameters():
        assert rel((p.grad / nv).cpu().numpy(), z["inner.meta_grad." + n]) < 3e-4, n


def test_fused_weighted_step_matches_reference_and_dense_path(golden_dir, monkeypatch):
    """dr4sr_sasrec_fwd_bwd_weighted (selection + weighting inside the 12-launch fused step) == the reference's inner step on the
    golden batch, and == the dense C-ABI composition under dropout with Philox Gumbel noise (same RNG step)"""
    from dr4sr_amd import _lib
    z = np.load(os.path.join(golden_dir, "metamodel_sasrec.npz"))
    ds, model = build(make_config(int(z["meta.num_items"])), monkeypatch)
    z, bt, bv = load_golden(golden_dir, model)
    model.train()
    sub, eng = model.sub_model, model.engine
    assert model._fused_ok()
    B, L = bt["item_id"].shape
    w = torch.zeros(B * L, device=model.device)
    model._fused_weighted(bt, weight_out=w)
    nv = float(eng.grads[eng.n_params])
    assert nv == float((bt["item_id"] != 0).sum())
    assert abs(float(eng.grads[eng.n_params + 1]) / nv - float(z["inner.loss"])) < 3e-6
    valid = (bt["item_id"] != 0).cpu().numpy()
    np.testing.assert_allclose(w.cpu().numpy()[:int(valid.sum())], z["inner.weight"][valid], rtol=1e-4, atol=1e-6)   # packed order
    for n, p in sub.named_parameters():
        ref = z["inner.grad." + n]
        assert rel((p.grad / nv).cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, n
    # dropout + in-kernel Gumbel noise: fused == dense composition when both start from the same RNG step
    cfg = make_config(int(z["meta.num_items"]), dropout=0.3)
    ds2, m2 = build(cfg, monkeypatch)
    m2.train()
    e2 = m2.engine
    bt2 = {k: v.clone() for k, v in bt.items()}
    rng = e2.state[3:4].clone()
    m2._fused_weighted(bt2)
    g_fused = e2.grads.clone()
    e2.state[3:4].copy_(rng)
    m2._weighted_fwd_bwd(bt2)
    g_dense = e2.grads.clone()
    n = e2.n_params
    assert float(g_fused[n]) == float(g_dense[n]) and abs(float(g_fused[n + 1]) - float(g_dense[n + 1])) < 1e-3
    assert rel(g_fused[:n].cpu().numpy(), g_dense[:n].cpu().numpy()) < 1e-4


@pytest.mark.parametrize("forward_hvp", [False, True])
def test_hypergradient_and_meta_sgd_match_reference(golden_dir, monkeypatch, forward_hvp):
    """forward_hvp: the opt-in one-sided Neumann probes (train.hypergrad_forward_hvp; round 5: the default is the central form)"""
    z = np.load(os.path.join(golden_dir, "metamodel_sasrec.npz"))
    cfg = make_config(int(z["meta.num_items"]))
    cfg["train"]["hypergrad_forward_hvp"] = forward_hvp
    ds, model = build(cfg, monkeypatch)
    z, bt, bv = load_golden(golden_dir, model)
    model.train()
    theta = model.engine.params.clone()
    hyper = model.hypergrad(bv, bt)
    assert torch.equal(theta, model.engine.params)                 # the probe shifts are undone exactly
    ref = np.concatenate([z["outer.hypergrad." + k].ravel() for k in NAMES])
    err = rel(hyper.cpu().numpy(), ref)
    print("hyper-gradient rel. error vs reference double-backward:", err)
    assert err < 1e-3, err                                          # north_star: 1e-3 rel fp32
    for s in (1, 2):                                                # MetaOptimizer.step x2: clip 10 + SGD momentum 0.9 + wd
        model.hypergrad_step(bv, bt)
        for k, p in model.meta_module.named_parameters():
            np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"outer.step{s}.{k}"], rtol=2e-5, atol=3e-7)


def load_golden_cl(golden_dir, model):
    z = np.load(os.path.join(golden_dir, "metamodel_cl4srec.npz"))
    sub_sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
    model.sub_model.load_state_dict(sub_sd, strict=True)
    model.meta_module.load_state_dict({k[11:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("meta_param.")}, strict=True)
    dev = model.device
    bt = {k[6:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("train.")}
    bv = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("val.")}
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    bt["_views"] = ((t("view.train.i"), t("view.train.i_len")), (t("view.train.j"), t("view.train.j_len")))
    bv["_views"] = ((t("view.val.i"), t("view.val.i_len")), (t("view.val.j"), t("view.val.j_len")))
    model._gumbel = torch.from_numpy(z["inner.gumbel"]).to(dev).reshape(-1, 2).contiguous()
    return z, bt, bv


def test_cl4srec_sub_model_matches_reference(golden_dir, monkeypatch):
    """round 4: MetaModel over a TUPLE-loss sub-model, metamodel.py:186-192 — weighted BCE + un-weighted
    cl_weight * InfoNCE.  Golden vectors from RUNNING the reference's MetaModel(CL4SRec) with its drawn views recorded
    (tools/make_golden.py run_meta_case): the weighted inner step in the fused, the dense and the autograd form, dL_val/dW, the
    finite-difference hyper-gradient against the reference's double backward, and two MetaOptimizer steps."""
    z = np.load(os.path.join(golden_dir, "metamodel_cl4srec.npz"))
    assert str(z["meta.sub_model"]) == "CL4SRec"
    ds, model = build(make_config(int(z["meta.num_items"]), sub="CL4SRec"), monkeypatch)
    z, bt, bv = load_golden_cl(golden_dir, model)
    model.train()
    sub, eng = model.sub_model, model.engine
    assert model._cl_sub() and model._fused_ok()
    n = eng.n_params
    # ---- fused weighted step + the contrastive term on the recorded views
    model._fused_weighted(bt)
    nv = float(eng.grads[n])
    assert nv == float((bt["item_id"] != 0).sum())
    assert abs(float(eng.grads[n + 1]) / nv - float(z["inner.loss"])) < 5e-6 * max(1.0, abs(float(z["inner.loss"])))
    for k, p in sub.named_parameters():
        ref = z["inner.grad." + k]
        assert rel((p.grad / nv).cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, k
    # ---- dense C-ABI composition: the same, plus d/dphi
    model._weighted_fwd_bwd(bt)
    assert abs(float(eng.grads[n + 1]) / nv - float(z["inner.loss"])) < 5e-6 * max(1.0, abs(float(z["inner.loss"])))
    for k, p in sub.named_parameters():
        ref = z["inner.grad." + k]
        assert rel((p.grad / nv).cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, k
    for k, p in model.meta_module.named_parameters():
        assert rel((p.grad / nv).cpu().numpy(), z["inner.meta_grad." + k]) < 3e-4, k
    # ---- API path (autograd over the C ABI), the reference's loop body: loss = model.training_step(batch); loss.backward()
    views = [bt["_views"][0], bt["_views"][1]]

    class Replay(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.k = 0

        def forward(self, sequences, seq_lens):
            v = views[self.k % 2]
            self.k += 1
            return v[0], v[1]
    real_aug = sub.augmentation_model.augmentation
    sub.augmentation_model.augmentation = Replay()
    sub.optimizer.zero_grad()
    model.meta_optimizer.zero_grad()
    loss = model.training_step(batch={k: v for k, v in bt.items() if k != "_views"}, align=False)
    loss.backward()
    sub.augmentation_model.augmentation = real_aug
    assert abs(float(loss) - float(z["inner.loss"])) < 5e-6 * max(1.0, abs(float(z["inner.loss"])))
    for k, p in sub.named_parameters():
        ref = z["inner.grad." + k]
        assert rel(p.grad.cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, k
    # ---- outer step: dL_val/dW (BCE + cl_weight * InfoNCE on the meta batch's views), hyper-gradient, MetaOptimizer
    eng.fwd_bwd(sub._batch_plan(bv))
    model._cl_extra(bv, views=bv["_views"])
    nvv = float(eng.grads[n])
    assert abs(float(eng.grads[n + 1]) / nvv - float(z["outer.val_loss"])) < 5e-6 * max(1.0, abs(float(z["outer.val_loss"])))
    for k, p in sub.named_parameters():
        ref = z["outer.grad_val." + k]
        assert rel((p.grad / nvv).cpu().numpy(), ref) < 3e-4 or np.abs(ref).max() < 1e-7, k
    theta = eng.params.clone()
    hyper = model.hypergrad(bv, bt)
    assert torch.equal(theta, eng.params)
    ref = np.concatenate([z["outer.hypergrad." + k].ravel() for k in NAMES])
    err = rel(hyper.cpu().numpy(), ref)
    print("MetaModel(CL4SRec) hyper-gradient rel. error vs reference double-backward:", err)
    assert err < 1e-3, err
    for s in (1, 2):
        model.hypergrad_step(bv, bt)
        for k, p in model.meta_module.named_parameters():
            np.testing.assert_allclose(p.detach().cpu().numpy(), z[f"outer.step{s}.{k}"], rtol=2e-5, atol=3e-7)


@pytest.mark.parametrize("sub", ["SASRec", "GRU4Rec", "FMLP", "CL4SRec"])
def test_metamodel_fit_end_to_end(tmp_path, monkeypatch, sub):
    """warm-up epoch (plain sub-model steps) then weighted epochs with an outer loop every 2 steps, then evaluate()"""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd import quickstart
    cfg = make_config(150, sub=sub, dropout=0.5 if sub != "GRU4Rec" else 0.2, n_rows=300, batch=64, epochs=3, warmup=0, interval=2)
    if sub == "FMLP":
        cfg["data"]["prefix_rows"] = True
    if sub == "GRU4Rec":
        cfg["model"]["sub_overrides"]["model"]["hidden_size"] = 128
    out = quickstart.run(cfg)
    assert {"ndcg@20", "recall@20"} <= set(out) and all(np.isfinite(v) for v in out.values())


def test_fused_weighted_step_at_scale_matches_dense_path(monkeypatch, at_scale):
    """B = 512 (T_max > 16 384): the weighted fused step runs with 32-row token tiles, the length-class attention launches (tiny VALU
    class included) and the owner-computed table gradient (scorer records carry weight * dpos / weight * dneg) — it must equal the
    dense C-ABI composition started from the same RNG step, with dropout and in-kernel Gumbel noise"""
    cfg = make_config(400, dropout=0.3, n_rows=1536, batch=512)
    ds, m = build(cfg, monkeypatch)
    m.train()
    e = m.engine
    loader = ds[0].get_loader()
    bt = m._local_batch(loader, m._perm(loader), 0)
    bt["neg_item"] = m._neg_sampling(bt)
    assert int(bt["item_id"].shape[0]) == 512
    rng = e.state[3:4].clone()
    m._fused_weighted(bt)
    g_fused = e.grads.clone()
    e.state[3:4].copy_(rng)
    m._weighted_fwd_bwd(bt)
    g_dense = e.grads.clone()
    n = e.n_params
    assert float(g_fused[n]) == float(g_dense[n]) and abs(float(g_fused[n + 1]) - float(g_dense[n + 1])) < 2e-3 * max(1.0, abs(float(g_dense[n + 1])))
    assert rel(g_fused[:n].cpu().numpy(), g_dense[:n].cpu().numpy()) < 1e-4
    # and the owner-computed table gradient of the weighted step is reproducible bit for bit
    e.state[3:4].copy_(rng)
    m._fused_weighted(bt)
    nE = e.n_items * e.D
    assert torch.equal(e.grads[:nE], g_fused[:nE])


@pytest.mark.parametrize("name", ["adam", "adagrad", "rmsprop", "lamb", "sgd"])
def test_meta_optimizer_choices_match_reference(golden_dir, name):
    """round 5, /root/reference/model/metamodel.py:59-81: every `meta_optimizer` choice through the product's MetaOptimizer
    (dr4sr_meta_opt_step / dr4sr_meta_sgd_step: clip_grad_norm_(10) + the optimizer step on the flat meta parameters) against the reference's
    own MetaOptimizer.step run with that choice (tests/golden/metamodel_optimizers.npz: three fixed hyper-gradients, the second clipped)"""
    from dr4sr_amd import _lib
    from dr4sr_amd.model.metamodel import MetaOptimizer, _PhiStore
    g = np.load(os.path.join(golden_dir, "metamodel_optimizers.npz"))
    dev = torch.device("cuda", 0)

    class Holder:                                                   # what MetaOptimizer reads of a MetaModel
        lib = _lib.load()
    Holder._phi = _PhiStore(Holder.lib, 64, dev)
    names = list(Holder._phi.views)
    for k in names:
        Holder._phi.views[k].copy_(torch.from_numpy(g["phi0." + k]))
    mo = MetaOptimizer(Holder, float(g["meta.meta_learning_rate"]), float(g["meta.hpo_learning_rate"]), float(g["meta.meta_weight_decay"]), name=name)
    for s in (1, 2, 3):
        hg = torch.cat([torch.from_numpy(g[f"grad{s}.{k}"]).reshape(-1) for k in names]).to(dev)
        mo.step_with(hg)
        torch.cuda.synchronize()
        assert abs(float(mo.last_grad_norm) - float(g[f"grad{s}.norm"])) < 1e-4 * float(g[f"grad{s}.norm"])
        for k in names:
            np.testing.assert_allclose(Holder._phi.views[k].cpu().numpy(), g[f"{name}.step{s}.{k}"], rtol=2e-5, atol=3e-7)
    assert int(mo.step_count) == 3


def test_meta_optimizer_sparse_adam_raises_like_the_reference(golden_dir):
    from dr4sr_amd.model.metamodel import MetaOptimizer
    g = np.load(os.path.join(golden_dir, "metamodel_optimizers.npz"))
    with pytest.raises(RuntimeError) as e:
        MetaOptimizer(None, 1e-3, 1e-3, 0.0, name="sparse_adam")
    assert str(e.value) in str(g["sparse_adam.error"])
