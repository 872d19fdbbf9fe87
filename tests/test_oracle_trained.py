"""Pin the oracle in the TRAINED-weight regime: oracle/sasrec_oracle.py and oracle/metamodel_oracle.py against vectors made by RUNNING the
reference with its shipped toys checkpoint (/root/reference/dataset/amazon-toys/toy/pre-trained_embedding.ckpt: table std 0.18 = 9 x
init, |in_proj| up to 1.23, LayerNorm gains up to 2.76) on the REAL toys rows — the first 256 rows and the real odd tail batch of 212
(tools/make_golden.py run_trained_case, run_meta_case(real=True)).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import metamodel_oracle as MO
from oracle import sasrec_oracle as O
from _golden_io import TABLE, load_meta_trained, load_trained


@pytest.fixture(scope="module")
def trained(golden_dir):
    return load_trained(golden_dir)


def _cfg(g):
    return int(g["meta.head_num"]), int(g["meta.layer_num"]), float(g["meta.layer_norm_eps"])


def test_fixture_is_the_shipped_checkpoint(trained):
    g, p, b = trained
    assert list(g["ckpt.keys"]) == ["config", "epoch", "metric", "model", "parameters"]        # utils/callbacks.py:70-76
    assert p[TABLE].shape == (11925, 64) and float(p[TABLE][0].abs().max()) == 0.0              # PAD row stayed exactly zero in training
    assert 0.17 < float(p[TABLE].std()) < 0.19                                                    # trained regime, not N(0, 0.02)
    assert b["b0"]["item_id"].shape == (256, 50) and b["tail"]["item_id"].shape == (212, 50)      # 19 412 mod 256
    for tag in ("b0", "tail"):                                                                    # the row recipe: targets = inputs shifted by one
        x, t, sl = b[tag]["in_item_id"], b[tag]["item_id"], b[tag]["seqlen"]
        for r in range(x.shape[0]):
            n = int(sl[r])
            assert torch.equal(x[r, 1:n], t[r, :n - 1]) and int(x[r, n:].abs().sum()) == 0 and int(t[r, n:].abs().sum()) == 0


@pytest.mark.parametrize("tag", ["b0", "tail"])
def test_forward_loss_and_gradients(trained, tag):
    g, p, batches = trained
    b = batches[tag]
    H, nl, eps = _cfg(g)
    loss, q, grads = O.grads_of(p, b, H, nl, eps)
    np.testing.assert_allclose(q.detach().numpy(), g[f"{tag}.query"], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(float(loss), float(g[f"{tag}.loss"]), rtol=1e-6)
    for k, gv in grads.items():
        ref = g[f"{tag}.grad.{k}"]
        err = float(np.abs(gv.numpy() - ref).max()) / max(1e-8, float(np.abs(ref).max()))
        assert err < 1e-4, (k, err)
    qq = O.sasrec_encode(p, b["in_item_id"], b["seqlen"], H, nl, eps, "origin")
    lnr, _, _ = O.score_bce(qq, p[TABLE], b["item_id"], b["neg_item"], False)
    np.testing.assert_allclose(lnr.numpy(), g[f"{tag}.loss_noreduce"], rtol=5e-5, atol=1e-8)


def test_adam_two_steps_tail_batch(trained):
    g, p, batches = trained
    b = batches["tail"]
    H, nl, eps = _cfg(g)
    params = {k: v.clone() for k, v in p.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    for t in (1, 2):
        loss, _, grads = O.grads_of(dict(params), b, H, nl, eps)
        params = O.adam_step(params, grads, m, v, t, lr=float(g["meta.lr"]), wd=float(g["meta.weight_decay"]))
    np.testing.assert_allclose(float(loss), float(g["tail.loss_step2"]), rtol=2e-6)
    for k in params:
        well = np.abs(g[f"tail.grad.{k}"]) > 1e-5
        d = np.abs(params[k].numpy() - g[f"tail.adam2.{k}"])
        assert d[well].max(initial=0) < 6e-6, (k, float(d[well].max(initial=0)))
        assert d.max() < 4e-4, k


def test_eval_topk(trained):
    g, p, _ = trained
    H, nl, eps = _cfg(g)
    q = O.sasrec_encode(p, torch.from_numpy(g["eval.in_item_id"]), torch.from_numpy(g["eval.seqlen"]), H, nl, eps, "last")
    np.testing.assert_allclose(q.numpy(), g["eval.query_last"], rtol=5e-5, atol=2e-5)
    score, items = O.full_score_topk(q, p[TABLE], torch.from_numpy(g["eval.user_hist"]), g["eval.topk_items"].shape[1])
    np.testing.assert_allclose(score.numpy(), g["eval.topk_score"], rtol=5e-5, atol=2e-5)
    assert (items.numpy() == g["eval.topk_items"]).mean() > 0.99


# ------------------------------------------------------------------------------------------------ MetaModel on the trained sub-model
@pytest.fixture(scope="module")
def meta_trained(golden_dir, trained):
    g, meta, bt, bv = load_meta_trained(golden_dir)
    cfg = {"H": int(g["meta.head_num"]), "n_layer": int(g["meta.layer_num"]), "eps": float(g["meta.layer_norm_eps"])}
    return g, trained[1], meta, bt, bv, cfg


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_meta_hypergradient_exact_on_trained_weights(meta_trained):
    g, p, meta, bt, bv, cfg = meta_trained
    f = MO.sasrec_losses(cfg)
    gum = torch.from_numpy(g["inner.gumbel"])
    tau, tmin, hlr = float(g["meta.tau"][0]), float(g["meta.tau_min"]), float(g["meta.hpo_learning_rate"])
    hg, gval, _ = MO.hypergrad_exact(f, p, meta, bt, bv, gum, tau, tmin, hlr)
    for k, v in gval.items():
        assert rel(v.numpy(), g["outer.grad_val." + k]) < 2e-4, k
    for k in MO.META_NAMES:
        assert rel(hg[k].numpy(), g["outer.hypergrad." + k]) < 5e-4, (k, rel(hg[k].numpy(), g["outer.hypergrad." + k]))


@pytest.mark.parametrize("richardson", [True, False])
def test_meta_first_order_form_on_trained_weights(meta_trained, richardson, rel_step=5e-4):
    """the finite-difference form the HIP path uses (mixed term Richardson-extrapolated), in fp32 on the CPU, against the reference's
    double-backward at trained weights; the plain two-point form misses the 1e-3 bar there by truncation (same error in fp64) — which
    is why the product does not use it"""
    g, p, meta, bt, bv, cfg = meta_trained
    f = MO.sasrec_losses(cfg)
    gum = torch.from_numpy(g["inner.gumbel"])
    tau, tmin, hlr = float(g["meta.tau"][0]), float(g["meta.tau_min"]), float(g["meta.hpo_learning_rate"])
    hg, _, _ = MO.hypergrad_fd(f, p, meta, bt, bv, gum, tau, tmin, hlr, rel_step=rel_step, richardson=richardson)
    flat = np.concatenate([hg[k].numpy().ravel() for k in MO.META_NAMES])
    ref = np.concatenate([g["outer.hypergrad." + k].ravel() for k in MO.META_NAMES])
    err = rel(flat, ref)
    print("rel_step", rel_step, "richardson", richardson, "hyper-gradient rel err at trained weights", err)
    if richardson:
        assert err < 1e-4, err
    else:
        assert 5e-4 < err < 2e-3, err


def test_ref_trainer_first_epoch_on_real_rows_matches_the_reference_curve(golden_dir):
    """END-TO-END statistical pin of oracle/ref_trainer.py (bench.py's cpu_baseline "port"): its training loop on the REAL toys rows
    from the deterministic init reaches the reference's own first-epoch mean loss (tools/make_golden.py run_curve_case: the reference's
    training_epoch, two seeds, agree to 0.02 % there).  Other random streams, same distribution."""
    from _golden_io import curve_init, load_curve
    from oracle import ref_trainer as RT
    g, rows = load_curve(golden_dir)
    torch.manual_seed(5)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    N, B = int(g["meta.num_items"]), int(g["meta.batch_size"])
    model = RT.RefLikeSASRec(N, D=int(g["meta.embed_dim"]), L=50, H=int(g["meta.head_num"]), Fh=int(g["meta.hidden_size"]),
                             p=float(g["meta.dropout_rate"]), eps=float(g["meta.layer_norm_eps"]), n_layer=int(g["meta.layer_num"]))
    sd = model.state_dict()
    init = curve_init({k: tuple(v.shape) for k, v in sd.items() if k != "query_encoder.item_encoder.weight"}, int(g["meta.init_seed"]))
    with torch.no_grad():
        for k, v in model.named_parameters():
            v.copy_(torch.from_numpy(init[k]))
    opt = torch.optim.Adam(model.parameters(), lr=float(g["meta.lr"]), weight_decay=float(g["meta.weight_decay"]))
    model.train()
    n = rows["seqlen"].shape[0]
    perm = torch.randperm(n)
    losses = []
    for i in range(0, n, B):
        idx = perm[i:i + B]
        batch = {k: v[idx] for k, v in rows.items()}
        batch["neg_item"] = model.neg_sampling(batch)
        opt.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    ref = g["curve.epoch_mean_loss"][:, 0]
    got = float(np.mean(losses))
    print("ref_trainer epoch-0 mean loss %.5f, reference %s" % (got, np.round(ref, 5).tolist()))
    assert len(losses) == 76 and abs(got - ref.mean()) < 4e-3 * ref.mean(), (got, ref)
