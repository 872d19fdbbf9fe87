"""Pin the oracle (oracle/sasrec_oracle.py) against golden vectors produced by RUNNING the reference
(tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import sasrec_oracle as O


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = {k: z[k] for k in z.files}
    params = {k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    batch = {k[len("batch."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    return g, params, batch


def meta(g):
    return int(g["meta.head_num"]), int(g["meta.layer_num"]), float(g["meta.layer_norm_eps"])


@pytest.mark.parametrize("name", ["sasrec_d64", "sasrec_d128"])
def test_forward_activations_and_loss(golden_dir, name):
    g, p, b = load(golden_dir, name)
    H, nl, eps = meta(g)
    q, acts = O.sasrec_encode(p, b["in_item_id"], b["seqlen"], H, nl, eps, "origin", return_all=True)
    # bit-exact embedding stage (pure gather + one IEEE add)
    assert np.array_equal(acts["x0"].numpy(), g["act.x0"])
    # the reference's per-layer outputs are only meaningful on rows < seqlen (pad rows are garbage but finite)
    L = b["in_item_id"].shape[1]
    valid = (torch.arange(L).view(1, L) < b["seqlen"].view(-1, 1)).numpy()
    for i in range(nl):
        np.testing.assert_allclose(acts[f"layer{i}"].numpy()[valid], g[f"act.layer{i}"][valid], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(q.numpy(), g["out.query"], rtol=2e-5, atol=2e-6)
    loss, pos, ng = O.score_bce(q, p["item_embedding.weight"], b["item_id"], b["neg_item"], True)
    tv = (b["item_id"] != 0).numpy()           # fixture holds basemodel.py:206 before the -inf fill of :208
    np.testing.assert_allclose(pos.numpy()[tv], g["out.pos_score"][tv], rtol=2e-5, atol=2e-6)
    assert np.isneginf(pos.numpy()[~tv]).all()
    np.testing.assert_allclose(ng.numpy(), g["out.neg_score"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(float(loss), float(g["out.loss"]), rtol=1e-6)
    loss_nr, _, _ = O.score_bce(q, p["item_embedding.weight"], b["item_id"], b["neg_item"], False)
    np.testing.assert_allclose(loss_nr.numpy(), g["out.loss_noreduce"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["sasrec_d64", "sasrec_d128"])
def test_gradients(golden_dir, name):
    g, p, b = load(golden_dir, name)
    H, nl, eps = meta(g)
    loss, q, grads = O.grads_of(p, b, H, nl, eps)
    for k, gv in grads.items():
        ref = g["grad." + k]
        scale = max(1e-8, float(np.abs(ref).max()))
        err = float(np.abs(gv.numpy() - ref).max()) / scale
        assert err < 2e-4, (k, err)


@pytest.mark.parametrize("name", ["sasrec_d64", "sasrec_d128"])
def test_adam_two_steps(golden_dir, name):
    g, p, b = load(golden_dir, name)
    H, nl, eps = meta(g)
    params = {k: v.clone() for k, v in p.items() if k != "query_encoder.item_encoder.weight"}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    for t, tag in ((1, "adam1."), (2, "adam2.")):
        full = dict(params)
        _, _, grads = O.grads_of(full, b, H, nl, eps)
        params = O.adam_step(params, grads, m, v, t, lr=float(g["meta.lr"]), wd=float(g["meta.weight_decay"]))
        for k in params:
            # Adam's first steps are ~lr*g/(|g|+eps): elements with |g| ~ eps=1e-8 are ill-conditioned
            # (fp32 rounding noise in g moves them by a visible fraction of lr) -> strict check only
            # where the reference gradient is well away from eps, loose bound elsewhere.
            well = np.abs(g["grad." + k]) > 1e-5
            np.testing.assert_allclose(params[k].numpy()[well], g[tag + k][well], rtol=0, atol=3e-6)
            np.testing.assert_allclose(params[k].numpy(), g[tag + k], rtol=0, atol=2e-4)


@pytest.mark.parametrize("name", ["sasrec_d64", "sasrec_d128"])
def test_eval_topk_and_metrics(golden_dir, name):
    g, p, _ = load(golden_dir, name)
    H, nl, eps = meta(g)
    idx = torch.from_numpy(g["eval.in_item_id"])
    sl = torch.from_numpy(g["eval.seqlen"])
    q = O.sasrec_encode(p, idx, sl, H, nl, eps, "last")
    np.testing.assert_allclose(q.numpy(), g["eval.query_last"], rtol=5e-5, atol=5e-6)
    hist = torch.from_numpy(g["eval.user_hist"])
    k = g["eval.topk_items"].shape[1]
    score, items = O.full_score_topk(q, p["item_embedding.weight"], hist, k)
    np.testing.assert_allclose(score.numpy(), g["eval.topk_score"], rtol=5e-5, atol=5e-6)
    assert (items.numpy() == g["eval.topk_items"]).mean() > 0.99     # ties/near-ties may swap
    hit = torch.from_numpy(g["eval.item_id"]).view(-1, 1) == torch.from_numpy(g["eval.topk_items"])
    for kk in (20, 10):
        np.testing.assert_allclose(O.ndcg_at(hit, kk).numpy(), g[f"eval.ndcg@{kk}"], rtol=1e-6)
        np.testing.assert_allclose(O.recall_at(hit, kk).numpy(), g[f"eval.recall@{kk}"], rtol=1e-6)


def test_neg_sampler_contract(golden_dir):
    z = np.load(os.path.join(golden_dir, "neg_sampler_stats.npz"))
    assert tuple(z["shape2d"]) == (4000, 50, 1) and tuple(z["shape1d"]) == (9, 1)
    assert z["counts"][0] == 0                      # PAD never sampled (basemodel.py:55)
    neg = O.neg_sample_reference_like(4000, 50, 37, torch.Generator().manual_seed(3))
    cnt = torch.bincount(neg.flatten(), minlength=37).numpy()
    assert cnt[0] == 0 and neg.shape == (4000, 50, 1) and neg.dtype == torch.int64
    exp = 4000 * 50 / 36
    for c in (cnt[1:], z["counts"][1:]):            # both uniform on 1..N-1 (chi-square, 35 dof)
        chi2 = float(((c - exp) ** 2 / exp).sum())
        assert chi2 < 80, chi2


@pytest.mark.parametrize("name", ["sasrec_d64", "sasrec_d128"])
def test_ref_like_trainer_matches_golden(golden_dir, name):
    """the cpu_baseline trainer (same torch modules as the reference) reproduces the reference's loss and grads"""
    from oracle.ref_trainer import RefLikeSASRec
    g, p, b = load(golden_dir, name)
    H, nl, eps = meta(g)
    m = RefLikeSASRec(int(g["meta.num_items"]), D=int(g["meta.embed_dim"]), H=H, Fh=int(g["meta.hidden_size"]), p=0.0,
                      eps=eps, n_layer=nl)
    m.load_state_dict(p, strict=True)
    m.train()
    loss = m.training_step(b)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["out.loss"]), rtol=1e-6)
    for n, prm in m.named_parameters():
        ref = g["grad." + n]
        assert float(np.abs(prm.grad.numpy() - ref).max()) <= 2e-5 * max(1e-8, float(np.abs(ref).max())) + 1e-9, n


def test_ref_like_gru4rec_trainer_matches_golden(golden_dir):
    """bench.py --model gru4rec's cpu_baseline leg (oracle/ref_trainer.RefLikeGRU4Rec: torch.nn.GRU, as module/layers.py:117-136)
    reproduces the reference's GRU4Rec loss and gradients"""
    from oracle.ref_trainer import RefLikeGRU4Rec
    z = np.load(os.path.join(golden_dir, "gru4rec_d64.npz"))
    g = {k: z[k] for k in z.files}
    m = RefLikeGRU4Rec(int(g["meta.num_items"]), D=int(g["meta.embed_dim"]), hidden=int(g["meta.hidden_size"]), n_layer=int(g["meta.layer_num"]), p=0.0)
    names = {"item_embedding.weight": "item_embedding.weight", "out.weight": "query_encoder.1.weight", "out.bias": "query_encoder.1.bias"}
    for l in range(int(g["meta.layer_num"])):
        names[f"gru.weight_ih_l{l}"] = f"query_encoder.0.3.gru.weight_ih_l{l}"
        names[f"gru.weight_hh_l{l}"] = f"query_encoder.0.3.gru.weight_hh_l{l}"
    m.load_state_dict({k: torch.from_numpy(g["param." + v]) for k, v in names.items()}, strict=True)
    m.train()
    b = {k[len("batch."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    loss = m.training_step(b)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["out.loss"]), rtol=1e-6)
    for n, prm in m.named_parameters():
        ref = g["grad." + names[n]]
        assert float(np.abs(prm.grad.numpy() - ref).max()) <= 2e-5 * max(1e-8, float(np.abs(ref).max())) + 1e-9, n


def test_ref_like_fmlp_trainer_matches_golden(golden_dir):
    """bench.py --model fmlp's cpu_baseline leg (oracle/ref_trainer.RefLikeFMLP: torch.fft filter + Intermediate, model/fmlp.py:8-39,
    module/layers.py:740-807) reproduces the reference's FMLP loss and gradients under the reference's own state-dict names"""
    from oracle.ref_trainer import RefLikeFMLP
    z = np.load(os.path.join(golden_dir, "fmlp_d64.npz"))
    g = {k: z[k] for k in z.files}
    m = RefLikeFMLP(int(g["meta.num_items"]), n_layer=int(g["meta.layer_num"]), p=0.0)
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}, strict=True)
    m.train()
    b = {k[len("batch."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    loss, q = m.training_step(b, return_query=True)
    loss.backward()
    np.testing.assert_allclose(q.detach().numpy(), g["out.query"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(float(loss.detach()), float(g["out.loss"]), rtol=2e-6)
    for n, prm in m.named_parameters():
        ref = g["grad." + n]
        assert float(np.abs(prm.grad.numpy() - ref).max()) <= 3e-4 * max(1e-8, float(np.abs(ref).max())) + 1e-9, n


def test_ref_like_cl4srec_trainer_matches_golden(golden_dir):
    """bench.py --model cl4srec's cpu_baseline leg (oracle/ref_trainer.RefLikeCL4SRec) on the views the reference drew
    (tests/golden/cl4srec_d64.npz): BCE + cl_weight x InfoNCE and its gradients; its own draws are legal augmentations"""
    from oracle.ref_trainer import RefLikeCL4SRec
    z = np.load(os.path.join(golden_dir, "cl4srec_d64.npz"))
    g = {k: z[k] for k in z.files}
    N = int(g["meta.num_items"])
    m = RefLikeCL4SRec(N, cl_weight=float(g["meta.cl_weight"]), temperature=float(g["meta.temperature"]), tau=float(g["meta.tau"]),
                       gamma=float(g["meta.gamma"]), beta=float(g["meta.beta"]), D=int(g["param.item_embedding.weight"].shape[1]), H=int(g["meta.head_num"]),
                       Fh=int(g["meta.hidden_size"]), p=0.0, eps=float(g["meta.layer_norm_eps"]), n_layer=int(g["meta.layer_num"]))
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}, strict=True)
    m.train()
    b = {k[len("batch."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch.")}
    m.fixed_views = ((torch.from_numpy(g["view.i"]), torch.from_numpy(g["view.i_len"])), (torch.from_numpy(g["view.j"]), torch.from_numpy(g["view.j_len"])))
    loss = m.training_step(b)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["out.loss"]), rtol=1e-5)
    for n, prm in m.named_parameters():
        ref = g["grad." + n]
        assert float(np.abs(prm.grad.numpy() - ref).max()) <= 2e-4 * max(1e-8, float(np.abs(ref).max())) + 1e-9, n
    # the trainer's own draws: crop keeps a window of max(1, int(tau n)) items, mask replaces int(gamma n) items by the mask item,
    # reorder permutes a window (module/data_augmentation.py:20-85)
    m.fixed_views = None
    hist, sl = b["in_item_id"], b["seqlen"]
    v, n = m._crop(hist, sl)
    for r in range(hist.shape[0]):
        n0, n1 = int(sl[r]), int(n[r])
        assert n1 == max(1, int(m.tau * n0)) and any(hist[r, s:s + n1].tolist() == v[r, :n1].tolist() for s in range(n0 - n1 + 1))
    v, n = m._mask(hist, sl)
    for r in range(hist.shape[0]):
        n0 = int(sl[r])
        assert int((v[r, :n0] == N).sum()) == int(m.gamma * n0) and bool(((v[r] == hist[r]) | (v[r] == N)).all())
    v, n = m._reorder(hist, sl)
    for r in range(hist.shape[0]):
        assert sorted(v[r].tolist()) == sorted(hist[r].tolist()) and int(n[r]) == int(sl[r])


def test_ref_like_metamodel_trainer_matches_golden(golden_dir):
    """bench.py --model metamodel's cpu_baseline leg (oracle/ref_trainer.RefLikeMetaModel) reproduces the reference's weighted inner
    loss and, after one outer loop (Hypergrad.grad -> clip -> SGD momentum), its meta-module parameters (metamodel_sasrec.npz)"""
    from oracle.ref_trainer import RefLikeMetaModel, RefLikeSASRec
    z = np.load(os.path.join(golden_dir, "metamodel_sasrec.npz"))
    g = {k: z[k] for k in z.files}
    pick = lambda pre: {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}
    p, bt, bv = pick("param."), pick("train."), pick("val.")
    N, D = p["item_embedding.weight"].shape
    sub = RefLikeSASRec(N, D=D, H=int(g["meta.head_num"]), Fh=int(g["meta.hidden_size"]), p=0.0, eps=float(g["meta.layer_norm_eps"]),
                        n_layer=int(g["meta.layer_num"]))
    sub.load_state_dict(p, strict=True)
    mm = RefLikeMetaModel(sub, D=D, tau_min=float(g["meta.tau_min"]), meta_lr=float(g["meta.meta_learning_rate"]),
                          hpo_lr=float(g["meta.hpo_learning_rate"]), meta_wd=float(g["meta.meta_weight_decay"]))
    mm.meta_module.load_state_dict(pick("meta_param."), strict=True)
    with torch.no_grad():
        mm.tau.copy_(torch.from_numpy(g["meta.tau"]))
    mm.gumbel = torch.from_numpy(g["inner.gumbel"])
    mm.train()
    np.testing.assert_allclose(float(mm.training_step(bt).detach()), float(g["inner.loss"]), rtol=2e-6)
    mm.outer_loop(bv, bt)
    for k, v in mm.meta_module.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["outer.step1." + k], rtol=1e-5, atol=2e-7)


def test_oracle_loss_modules_match_reference(golden_dir):
    """oracle bce_from_scores / bpr_from_scores vs the reference's BinaryCrossEntropyLoss / BPRLoss run on fixed scores
    (tests/golden/loss_modules.npz, made by tools/make_golden.py loss_module_vectors): loss and both gradients"""
    z = np.load(os.path.join(golden_dir, "loss_modules.npz"))
    assert "unexpected keyword argument 'reduce'" in str(z["bpr.reduce_kwarg_error"])       # a8: the reference cannot call BPR from training_step
    for tag in ("a", "b", "c", "d"):                     # d: the plain-mean branch (loss_func.py:32-33), BCE only
        pos0, neg0 = torch.from_numpy(z[f"{tag}.pos"]), torch.from_numpy(z[f"{tag}.neg"])
        for name in (("bce", "bce_nr") if tag == "d" else ("bce", "bce_nr", "bpr")):
            p, n = pos0.clone().requires_grad_(True), neg0.clone().requires_grad_(True)
            loss = O.bpr_from_scores(p, n) if name == "bpr" else O.bce_from_scores(p, n, reduce=(name == "bce"))
            np.testing.assert_allclose(loss.detach().numpy(), z[f"{tag}.{name}.loss"], rtol=2e-6, atol=1e-7)
            (loss * torch.from_numpy(z[f"{tag}.{name}.up"])).sum().backward()
            np.testing.assert_allclose(torch.nan_to_num(p.grad, nan=0.0).numpy(), z[f"{tag}.{name}.dpos"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(n.grad.numpy(), z[f"{tag}.{name}.dneg"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("kind", ["adam", "sgd", "adagrad", "rmsprop"])
def test_optimizer_oracle_matches_torch_optim(kind):
    """oracle/optim_oracle.py restates the optimizers of basemodel.py:79-98 (torch.optim with the defaults the reference constructs them
    with); pinned against torch.optim itself, 5 steps, with and without weight decay"""
    from oracle import optim_oracle as OO
    cls = {"adam": torch.optim.Adam, "sgd": torch.optim.SGD, "adagrad": torch.optim.Adagrad, "rmsprop": torch.optim.RMSprop}[kind]
    gen = torch.Generator().manual_seed(4)
    for wd in (0.0, 1e-2):
        p0 = torch.randn(257, generator=gen)
        ref = torch.nn.Parameter(p0.clone())
        opt = cls([ref], lr=1e-2, weight_decay=wd)
        p, st = p0.clone(), OO.init_state(p0)
        for _ in range(5):
            g = torch.randn(257, generator=gen) * torch.rand(257, generator=gen)
            ref.grad = g.clone()
            opt.step()
            p = OO.step(kind, p, g, st, 1e-2, wd)
            np.testing.assert_allclose(p.numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
