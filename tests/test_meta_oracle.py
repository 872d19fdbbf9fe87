"""Pin oracle/metamodel_oracle.py against the golden vectors produced by RUNNING the reference's MetaModel
(tools/make_golden.py run_meta_case -> tests/golden/metamodel_sasrec.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import metamodel_oracle as MO


def load_meta(golden_dir, name="metamodel_sasrec"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = {k: z[k] for k in z.files}
    pick = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}
    p = pick("param.")
    p.pop("query_encoder.item_encoder.weight", None)            # tied to item_embedding.weight
    cfg = {"H": int(g["meta.head_num"]), "n_layer": int(g["meta.layer_num"]), "eps": float(g["meta.layer_norm_eps"])}
    return g, p, pick("meta_param."), pick("train."), pick("val."), cfg


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_weighted_inner_step(golden_dir):
    g, p, meta, bt, bv, cfg = load_meta(golden_dir)
    f = MO.sasrec_losses(cfg)
    P = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    M = {k: v.clone().requires_grad_(True) for k, v in meta.items()}
    lp, q = f(P, bt, False)
    loss, w = MO.weighted_loss(lp, q, M, torch.from_numpy(g["inner.gumbel"]), float(g["meta.tau"][0]), float(g["meta.tau_min"]),
                               bt["user_id"], bt["item_id"])
    np.testing.assert_allclose(q.detach().numpy(), g["inner.query"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(lp.detach().numpy(), g["inner.loss_pos"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(w.detach().numpy(), g["inner.weight"], rtol=1e-5, atol=1e-7)
    assert (w.detach().numpy()[1][bt["item_id"][1].numpy() != 0] == 1.0).all()     # the user_id == 0 row
    np.testing.assert_allclose(float(loss), float(g["inner.loss"]), rtol=2e-6)
    loss.backward()
    for k, v in P.items():
        assert rel(v.grad.numpy() if v.grad is not None else np.zeros(v.shape), g["inner.grad." + k]) < 2e-4 \
            or np.abs(g["inner.grad." + k]).max() < 1e-7, k
    for k, v in M.items():
        assert rel(v.grad.numpy(), g["inner.meta_grad." + k]) < 2e-4, k


def test_hypergradient_exact_and_meta_sgd(golden_dir):
    g, p, meta, bt, bv, cfg = load_meta(golden_dir)
    f = MO.sasrec_losses(cfg)
    gum = torch.from_numpy(g["inner.gumbel"])
    tau, tmin, hlr = float(g["meta.tau"][0]), float(g["meta.tau_min"]), float(g["meta.hpo_learning_rate"])
    hg, gval, _ = MO.hypergrad_exact(f, p, meta, bt, bv, gum, tau, tmin, hlr)
    for k, v in gval.items():
        assert rel(v.numpy(), g["outer.grad_val." + k]) < 2e-4 or np.abs(g["outer.grad_val." + k]).max() < 1e-7, k
    for k in MO.META_NAMES:
        assert rel(hg[k].numpy(), g["outer.hypergrad." + k]) < 5e-4, (k, rel(hg[k].numpy(), g["outer.hypergrad." + k]))
    # two MetaOptimizer steps (clip 10, SGD momentum 0.9 + weight decay) — utils/utils.py:221-252, metamodel.py:68-69
    M = {k: v.clone() for k, v in meta.items()}
    bufs = [None] * 4
    for s in (1, 2):
        hg, _, _ = MO.hypergrad_exact(f, p, M, bt, bv, gum, tau, tmin, hlr)
        grads, _ = MO.clip_grad_norm_([hg[k] for k in MO.META_NAMES], 10.0)
        new, bufs = MO.sgd_momentum_step([M[k] for k in MO.META_NAMES], grads, bufs, float(g["meta.meta_learning_rate"]), 0.9,
                                         float(g["meta.meta_weight_decay"]))
        M = dict(zip(MO.META_NAMES, new))
        for k in MO.META_NAMES:
            np.testing.assert_allclose(M[k].numpy(), g[f"outer.step{s}.{k}"], rtol=1e-5, atol=2e-7)


@pytest.mark.parametrize("forward_hvp", [False, True])
@pytest.mark.parametrize("rel_step", [3e-4, 1e-3])
def test_first_order_formulation_matches_exact(golden_dir, rel_step, forward_hvp):
    """the finite-difference form used on the GPU reproduces the reference's double-backward hyper-gradient; forward_hvp (round 4, the
    product's default for BCE sub-models): the three Neumann terms from one-sided differences against G(W) — they enter scaled by hpo_lr"""
    g, p, meta, bt, bv, cfg = load_meta(golden_dir)
    f = MO.sasrec_losses(cfg)
    gum = torch.from_numpy(g["inner.gumbel"])
    tau, tmin, hlr = float(g["meta.tau"][0]), float(g["meta.tau_min"]), float(g["meta.hpo_learning_rate"])
    hg, _, pacc = MO.hypergrad_fd(f, p, meta, bt, bv, gum, tau, tmin, hlr, rel_step=rel_step, forward_hvp=forward_hvp)
    flat = lambda d: np.concatenate([np.asarray(d[k]).ravel() for k in MO.META_NAMES])
    ref = np.concatenate([g["outer.hypergrad." + k].ravel() for k in MO.META_NAMES])
    err = rel(flat({k: v.numpy() for k, v in hg.items()}), ref)
    print("rel_step", rel_step, "hypergrad rel err", err)
    assert err < 3e-4, err


# ------------------------------------------------------------------------------------------------ CL4SRec sub-model (round 4)
def load_meta_cl(golden_dir):
    g, p, meta, bt, bv, cfg = load_meta(golden_dir, "metamodel_cl4srec")
    cfg.update(temperature=float(g["meta.temperature"]), cl_weight=float(g["meta.cl_weight"]))
    t = lambda k: torch.from_numpy(g[k])
    bt["_views"] = ((t("view.train.i"), t("view.train.i_len")), (t("view.train.j"), t("view.train.j_len")))
    bv["_views"] = ((t("view.val.i"), t("view.val.i_len")), (t("view.val.j"), t("view.val.j_len")))
    return g, p, meta, bt, bv, cfg


def test_cl4srec_sub_model_weighted_step_and_hypergradient(golden_dir):
    """MetaModel over a tuple-loss sub-model (metamodel.py:186-192): weighted BCE + UN-weighted cl_weight * InfoNCE, the views the
    reference drew recorded; inner loss / gradients, dL_val/dW, the double-backward hyper-gradient, and its first-order form"""
    g, p, meta, bt, bv, cfg = load_meta_cl(golden_dir)
    f = MO.cl4srec_losses(cfg)
    gum = torch.from_numpy(g["inner.gumbel"])
    tau, tmin, hlr = float(g["meta.tau"][0]), float(g["meta.tau_min"]), float(g["meta.hpo_learning_rate"])
    P = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    M = {k: v.clone().requires_grad_(True) for k, v in meta.items()}
    (lp, cl_rows), q = f(P, bt, False)
    np.testing.assert_allclose(lp.detach().numpy(), g["inner.loss_pos"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(cl_rows.detach().numpy(), g["inner.cl_rows"], rtol=2e-5, atol=1e-7)
    loss = MO.train_loss(f, P, M, bt, gum, tau, tmin)
    np.testing.assert_allclose(float(loss), float(g["inner.loss"]), rtol=3e-6)
    loss.backward()
    for k, v in P.items():
        assert rel(v.grad.numpy() if v.grad is not None else np.zeros(v.shape), g["inner.grad." + k]) < 2e-4 \
            or np.abs(g["inner.grad." + k]).max() < 1e-7, k
    for k, v in M.items():
        assert rel(v.grad.numpy(), g["inner.meta_grad." + k]) < 2e-4, k
    hg, gval, _ = MO.hypergrad_exact(f, p, meta, bt, bv, gum, tau, tmin, hlr)
    for k, v in gval.items():
        assert rel(v.numpy(), g["outer.grad_val." + k]) < 2e-4 or np.abs(g["outer.grad_val." + k]).max() < 1e-7, k
    flat = lambda d: np.concatenate([np.asarray(d[k]).ravel() for k in MO.META_NAMES])
    ref = np.concatenate([g["outer.hypergrad." + k].ravel() for k in MO.META_NAMES])
    assert rel(flat({k: v.numpy() for k, v in hg.items()}), ref) < 5e-4
    hf, _, _ = MO.hypergrad_fd(f, p, meta, bt, bv, gum, tau, tmin, hlr, rel_step=5e-4, richardson=True)
    err = rel(flat({k: v.numpy() for k, v in hf.items()}), ref)
    print("CL4SRec sub-model: first-order hyper-gradient rel err", err)
    assert err < 1e-3, err


@pytest.mark.parametrize("name", ["adam", "adagrad", "rmsprop", "lamb", "sgd"])
def test_meta_optimizer_choices_match_reference(golden_dir, name):
    """metamodel.py:59-81: the oracle's restatement of every `meta_optimizer` choice against the reference's own MetaOptimizer.step run
    with that choice on three fixed hyper-gradients, the second one clipped (tests/golden/metamodel_optimizers.npz,
    tools/make_golden.py run_meta_optimizer_case); 'lamb' = an unknown name = the else branch, Adam WITH meta_weight_decay"""
    g = np.load(os.path.join(golden_dir, "metamodel_optimizers.npz"))
    assert str(g[name + ".torch_class"]) == {"adam": "Adam", "adagrad": "Adagrad", "rmsprop": "RMSprop", "lamb": "Adam", "sgd": "SGD"}[name]
    assert float(g["grad2.norm"]) > 10.0 > float(g["grad1.norm"])
    P = [torch.from_numpy(g["phi0." + k]) for k in MO.META_NAMES]
    st = {}
    for s in (1, 2, 3):
        grads = [torch.from_numpy(g[f"grad{s}.{k}"]) for k in MO.META_NAMES]
        P = MO.meta_optimizer_step(name, P, grads, st, float(g["meta.meta_learning_rate"]), float(g["meta.meta_weight_decay"]))
        for k, v in zip(MO.META_NAMES, P):
            np.testing.assert_allclose(v.numpy(), g[f"{name}.step{s}.{k}"], rtol=1e-5, atol=2e-7)
    assert "SparseAdam does not support dense gradients" in str(g["sparse_adam.error"])
