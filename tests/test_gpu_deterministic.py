"""Round 5: deterministic training — the reference asks for run-to-run determinism (/root/reference/utils/utils.py:13-20: seeds +
cudnn.deterministic = True).  DR4SR_DETERMINISTIC=1 / train.deterministic: every reduction of the step in a fixed order — the at-scale launch
forms at every batch size (owner-computed item-table gradient, per-sequence / list attention: no atomics) + the weight-gradient launch's partial
sums stored per token split and added in split order by k_wgrad_det_reduce (csrc/linear.hip).  Done = two 100-step runs bit-identical."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, D, steps, seed=2023, dropout=0.5, NL=2, dense=False):
    from test_gpu_parity import _random_params
    from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    L, N = 50, (20034 if D == 128 else TOYS_N_ITEMS)
    rows = make_rows(n_rows=4096, n_items=N, seed=17, dense=dense)
    data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
    perm = torch.from_numpy(np.random.default_rng(3).permutation(4096)).to(dev)
    eng = SasrecEngine(N, L, D, 2, 128, NL, 1e-12, dropout, B, dev, seed=seed, lr=1e-3)
    eng.load_named(_random_params(N, D, 128, NL, seed=6))
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    log = torch.zeros(steps, dtype=torch.float32, device=dev)
    plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=torch.zeros(B, dtype=torch.int64, device=dev),
                         neg_item=torch.zeros(B, L, dtype=torch.int64, device=dev), sample_neg=True, perm_sel=(perm, B, 0, counter), loss_log=log)
    eng.train_steps(plan, steps)
    torch.cuda.synchronize()
    return eng.params.clone(), log.clone(), eng.adam_v.clone()


@pytest.mark.parametrize("B,D", [(256, 64), (256, 128), (2048, 64)])
def test_deterministic_mode_two_runs_bit_identical(monkeypatch, B, D):
    """100 training steps (dropout 0.5, in-kernel negatives, device-side batch selection) twice from the same state: the flat parameter
    buffer, the Adam second moments and the per-step loss log are bit-identical; and the trajectory is the default mode's up to fp32
    summation order"""
    steps = 100 if B == 256 else 30
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    p1, l1, v1 = _run(B, D, steps)
    p2, l2, v2 = _run(B, D, steps)
    assert torch.equal(p1, p2) and torch.equal(v1, v2) and torch.equal(l1, l2)
    assert float(l1[-1]) < float(l1[0]) and bool(torch.isfinite(p1).all())
    monkeypatch.delenv("DR4SR_DETERMINISTIC")
    p0, l0, _ = _run(B, D, steps)
    # same training, other summation order: Adam turns ulps of a near-zero gradient into +- lr steps, so the parameters may drift by a few lr
    assert float((p0 - p1).abs().max()) < 3e-2 and float(((l0 - l1).abs() / l0.abs()).max()) < 1e-2


@pytest.mark.parametrize("B,NL", [(1, 2), (3, 2), (33, 1), (100, 3), (8192, 2)])
def test_deterministic_mode_odd_batches_and_layer_counts(monkeypatch, B, NL):
    """the same bit-identity at the edges: one row, batches that fill no token tile, one and three encoder layers, and a batch whose
    weight-gradient launch uses every token split (8 192 rows)"""
    steps = 10 if B == 8192 else 40
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    p1, l1, v1 = _run(B, 64, steps, NL=NL)
    p2, l2, v2 = _run(B, 64, steps, NL=NL)
    assert torch.equal(p1, p2) and torch.equal(v1, v2) and torch.equal(l1, l2)
    assert bool(torch.isfinite(p1).all()) and bool(torch.isfinite(l1).all())


@pytest.mark.parametrize("B,D", [(512, 64), (2048, 128)])
def test_deterministic_mode_long_sequences(monkeypatch, B, D):
    """all-50 rows (one workgroup per sequence attention, every token tile full) and the d = 128 tiles at scale"""
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    p1, l1, v1 = _run(B, D, 12, dense=True)
    p2, l2, v2 = _run(B, D, 12, dense=True)
    assert torch.equal(p1, p2) and torch.equal(v1, v2) and torch.equal(l1, l2) and bool(torch.isfinite(l1).all())


@pytest.mark.parametrize("D", [64, 128])
def test_deterministic_mode_gradients_match_oracle(monkeypatch, D):
    """the deterministic forms' loss and every gradient against the ORACLE's autograd at BASELINE's batch size (dropout 0, given negatives),
    and the bit-identical replay of the same batch (model/sasrec.py:39-75, model/basemodel.py:204-214, model/loss_func.py:9-38)"""
    from oracle import sasrec_oracle as O
    from test_gpu_parity import _random_params, _toys_batch, relerr
    from dr4sr_amd.engine import SasrecEngine
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    dev = torch.device("cuda", 0)
    B = 256
    b, N = _toys_batch(B, False, seed=5, n_items=20034 if D == 128 else None)
    params = _random_params(N, D, 128, 2, seed=1)
    eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, 0.0, B, dev)
    eng.load_named(params)
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    g1 = eng.grads.clone()
    loss, n = eng.loss_and_count()
    loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(loss_o)) < 2e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < 2e-4, k
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grads)


@pytest.mark.parametrize("B,D,dense", [(256, 64, False), (64, 64, True), (32, 128, True), (1, 64, True), (1280, 64, False)])
def test_deterministic_latency_form(monkeypatch, B, D, dense):
    """round 6: a plan of the latency regime keeps its latency launches in deterministic mode (dr4sr_sasrec_at_scale bit 6: 16-token
    tiles with the attention inside, bit 2) — the in-tile attention's shared dK | dV rows go through per-query-tile partial blocks that the next
    launch sums in tile order (csrc/linear.hip det_kv_rows), table gradient and weight-gradient splits through k_wgrad's ordered forms.
    Toys rows and all-50 rows (every key row collects from up to five query tiles), one row, the largest toys batch of the regime: two runs
    bit-identical, the first step's gradient equal to the default mode's up to the order of the sums, and the autograd-path backward
    (dr4sr_sasrec_encode_bwd) repeatable bit for bit"""
    from test_gpu_parity import _random_params
    from dr4sr_amd import _lib
    from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
    from dr4sr_amd.engine import SasrecEngine
    dev = torch.device("cuda", 0)
    N = 20034 if D == 128 else TOYS_N_ITEMS
    rows = make_rows(n_rows=B, n_items=N, seed=23, dense=dense)
    data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}

    def grads(det):
        (monkeypatch.setenv("DR4SR_DETERMINISTIC", "1") if det else monkeypatch.delenv("DR4SR_DETERMINISTIC", raising=False))
        eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, 0.5, B, dev, seed=9)
        eng.load_named(_random_params(N, D, 128, 2, seed=6))
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], neg_item=torch.zeros(B, 50, dtype=torch.int64, device=dev), sample_neg=True)
        bits = eng.lib.dr4sr_sasrec_at_scale(C.byref(plan))
        eng.fwd_bwd(plan)
        torch.cuda.synchronize()
        g = eng.grads.clone()
        q = eng.encode(plan, True, _lib.POOL_LAST)
        d_out = torch.randn(q.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        api = []
        for _ in range(2):
            eng.grads.zero_()
            eng.encode_bwd(plan, True, _lib.POOL_LAST, d_out)
            torch.cuda.synchronize()
            api.append(eng.grads.clone())
        for _ in range(20):
            eng.train_step(plan)
        torch.cuda.synchronize()
        return bits, g, api, eng.params.clone()

    bits, g1, api1, p1 = grads(True)
    assert bits & 64 and bits & 4 and not bits & 1, bits
    _, g2, api2, p2 = grads(True)
    assert torch.equal(g1, g2) and torch.equal(p1, p2) and torch.equal(api1[0], api1[1]) and torch.equal(api1[0], api2[0])
    bits0, g0, api0, _ = grads(False)
    assert not bits0 & 64 and bits0 & 4
    scale = float(g0[:-4].abs().max())
    assert float((g1 - g0)[:-4].abs().max()) < 2e-5 * scale and float((api1[0] - api0[0])[:-4].abs().max()) < 2e-5 * float(api0[0][:-4].abs().max())


def test_deterministic_config_key_sets_the_mode(monkeypatch, tmp_path):
    """train.deterministic: true in the YAML-level config = the switch (BaseModel.__init__), and fit() under it is reproducible end to end"""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DR4SR_CONFIG_DIR", os.path.join(ROOT, "configs"))
    from dr4sr_amd import _lib
    from dr4sr_amd.utils import load_config, prepare_datasets, prepare_model, seed_everything
    outs = []
    for _ in range(2):
        cfg = load_config({"model": "SASRec", "dataset": "synthetic-toys"})
        cfg["data"].update({"n_items": 300, "n_rows": 1000, "n_eval_rows": 128, "seed": 3})
        cfg["train"].update({"epochs": 2, "batch_size": 128, "deterministic": True, "device": "cuda:0"})
        seed_everything(cfg["train"]["seed"])
        ds = prepare_datasets(cfg)
        model = prepare_model(cfg, ds)
        assert os.environ.get("DR4SR_DETERMINISTIC") == "1"
        model._init_model(ds[0])
        model.train()
        for ep in range(2):
            model.training_epoch(ep)
        torch.cuda.synchronize()
        outs.append(model.engine.params.clone())
    _lib.set_env("DR4SR_DETERMINISTIC", None)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B", [4096, 256])
def test_deterministic_two_phase_step_is_bitwise_the_one_launch_step(monkeypatch, B):
    """the two-bucket data-parallel step (dr4sr_sasrec_fwd_bwd_phase 1 + 2: the last backward launch cut in two, each half followed by its
    ordered reduce) leaves bit for bit the gradient of dr4sr_sasrec_fwd_bwd in deterministic mode — every float of the flat buffer.
    B = 256: the deterministic latency form, whose table gradient is a set of k_wgrad jobs too (two buckets where the default mode has one)"""
    from test_gpu_dp import _engine, _grads_of
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    eng, plan, _ = _engine(B)
    assert len(eng.grad_buckets(plan)) == 2
    g_one = _grads_of(eng, lambda: eng.fwd_bwd(plan))
    g_two = _grads_of(eng, lambda: (eng.fwd_bwd_phase(plan, False, 1), eng.fwd_bwd_phase(plan, False, 2)))
    assert torch.equal(g_one, g_two)


def _fmlp_run(B, NL, L, N, steps, dropout=0.5, sample_neg=True):
    from dr4sr_amd.fmlp_engine import FmlpEngine, fmlp_param_names, fmlp_param_shapes
    gen = torch.Generator().manual_seed(77 + B)
    idx = torch.zeros(B, L, dtype=torch.long)
    for i in range(B):
        n = int(torch.randint(1, L + 1, (1,), generator=gen))
        idx[i, L - n:] = torch.randint(1, min(N, 400), (n,), generator=gen)          # few distinct ids: many rows meet in every table row
    tgt = torch.randint(1, N, (B,), generator=gen)
    neg = torch.randint(1, N, (B, 1), generator=gen)
    params = {}
    for nme, shp in zip(fmlp_param_names(NL), fmlp_param_shapes(N, L, 64, 256, NL)):
        params[nme] = (1.0 if nme.endswith("LayerNorm.weight") else 0.0) + 0.05 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = FmlpEngine(N, L, 64, 256, NL, 1e-12, dropout, B, "cuda", seed=11, lr=1e-3)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(idx.to(dev), tgt.to(dev), neg_item=neg.view(-1).contiguous().to(dev), sample_neg=sample_neg)
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    g0 = eng.grads.clone()
    for _ in range(steps):
        eng.train_step(plan)
    torch.cuda.synchronize()
    return g0, eng.params.clone(), params, {"in_item_id": idx, "item_id": tgt, "neg_item": neg}


@pytest.mark.parametrize("B,NL,L,N", [(256, 2, 50, 11925), (33, 3, 20, 150), (1024, 2, 50, 40000), (1, 1, 2, 150)])
def test_fmlp_deterministic_mode_two_runs_bit_identical(monkeypatch, B, NL, L, N):
    """round 6: FMLP under DR4SR_DETERMINISTIC — the item-table gradient owner-computed in token order (linear.hip
    launch_table_owner64 over the scorer's records + the embedding stage's rows), the dense_1 / dense_2 blocks and the Intermediate LayerNorm
    sums stored per token split and added in split order: the first step's whole flat gradient and the parameters after 40 steps (dropout
    0.5, in-kernel negatives) are bit-identical between two runs — and differ between two runs of the default mode's atomics only in the
    last bits (the mode changes the ORDER of the sums, nothing else)"""
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    g_a, p_a, _, _ = _fmlp_run(B, NL, L, N, 40)
    g_b, p_b, _, _ = _fmlp_run(B, NL, L, N, 40)
    assert torch.equal(g_a, g_b) and torch.equal(p_a, p_b) and bool(torch.isfinite(p_a).all())
    monkeypatch.delenv("DR4SR_DETERMINISTIC")
    g_c, p_c, _, _ = _fmlp_run(B, NL, L, N, 40)
    scale = float(g_c.abs().max())
    assert float((g_a - g_c).abs().max()) < 2e-5 * scale, (float((g_a - g_c).abs().max()), scale)


@pytest.mark.parametrize("B,NL,L", [(29, 2, 50), (100, 1, 8), (256, 2, 50)])
def test_fmlp_deterministic_mode_gradients_match_oracle(monkeypatch, B, NL, L):
    """the same gradients as the default mode's: loss and every parameter's gradient against the fp32 oracle (tests/test_gpu_fmlp.py's bar)"""
    from oracle import fmlp_oracle as FO
    from test_gpu_fmlp import relerr, REL
    from dr4sr_amd.fmlp_engine import FmlpEngine
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    N = 150
    _, _, params, batch = _fmlp_run(B, NL, L, N, 0, dropout=0.0, sample_neg=False)
    eng = FmlpEngine(N, L, 64, 256, NL, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(batch["in_item_id"].to(dev), batch["item_id"].to(dev), neg_item=batch["neg_item"].view(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    loss_o, _, grads_o = FO.grads_of(params, batch, NL)
    loss, n = eng.loss_and_count()
    assert n == B and abs(loss - float(loss_o)) < 3e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, grads_o[k]) < REL, k
    # the autograd path's backward (encode_bwd: no scorer records, embedding rows only) in the same mode
    out = eng.encode(plan, False)
    d_out = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    eng.grads.zero_()
    eng.encode_bwd(plan, False, d_out)
    torch.cuda.synchronize()
    g1 = eng.grads.clone()
    eng.grads.zero_()
    eng.encode_bwd(plan, False, d_out)
    torch.cuda.synchronize()
    assert torch.equal(g1, eng.grads)


def _gru_run(B, H, NL, N, steps, L=50):
    from dr4sr_amd.data.synthetic import make_rows
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    rows = make_rows(n_rows=B, n_items=N, seed=3)
    b = {k: torch.from_numpy(rows[k]) for k in ("in_item_id", "item_id", "seqlen")}
    gen = torch.Generator().manual_seed(1)
    params = {}
    for nme, shp in zip(gru_param_names(NL), gru_param_shapes(N, 64, H, NL)):
        params[nme] = 0.08 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = GruEngine(N, L, 64, H, NL, 0.2, B, "cuda", seed=5)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev))          # in-kernel negatives
    eng.fwd_bwd(plan)
    torch.cuda.synchronize()
    g0 = eng.grads.clone()
    for _ in range(steps):
        eng.train_step(plan)
    torch.cuda.synchronize()
    return g0, eng.params.clone()


@pytest.mark.parametrize("B,H,NL,N", [(256, 256, 2, 12102), (40, 128, 1, 300), (1024, 128, 2, 12102), (17, 256, 3, 300)])
def test_gru4rec_deterministic_mode_two_runs_bit_identical(monkeypatch, B, H, NL, N):
    """round 6: GRU4Rec under DR4SR_DETERMINISTIC — the separate glue launches at every size, the scorer's records + the masked
    d x rows owner-computed into the item table in token order (launch_table_owner64), every 64 x 64 weight-gradient job's block stored per
    token split and added in split order (k_wgrad64_det_reduce): the first step's flat gradient and the parameters after 30 steps (dropout
    0.2, in-kernel negatives; one- and two-layer recurrences, the layer wavefront and the per-layer launches) bit-identical between two
    runs, and equal to the default mode's up to the order of the sums"""
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    g_a, p_a = _gru_run(B, H, NL, N, 30)
    g_b, p_b = _gru_run(B, H, NL, N, 30)
    assert torch.equal(g_a, g_b) and torch.equal(p_a, p_b) and bool(torch.isfinite(p_a).all())
    monkeypatch.delenv("DR4SR_DETERMINISTIC")
    g_c, _ = _gru_run(B, H, NL, N, 0)
    scale = float(g_c.abs().max())
    assert float((g_a - g_c).abs().max()) < 2e-5 * scale, (float((g_a - g_c).abs().max()), scale)


@pytest.mark.parametrize("case", ["odd-40", "h128-3-layers", "h256-1-layer", "full-size-dropout"])
def test_gru4rec_deterministic_mode_gradients_match_oracle(monkeypatch, case):
    """the oracle parity cases of tests/test_gpu_gru.py with the mode on: loss and every gradient within the same bar"""
    import test_gpu_gru as G
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    if case == "full-size-dropout":
        G.test_gru4rec_full_size_vs_oracle(False)
    else:
        G._gru_case(*{"odd-40": (40, 256, 2, 50), "h128-3-layers": (29, 128, 3, 7), "h256-1-layer": (29, 256, 1, 50)}[case])


@pytest.mark.parametrize("D,bpr", [(64, False), (128, False), (64, True)])
def test_dense_scorer_backward_is_ordered_in_deterministic_mode(monkeypatch, D, bpr):
    """dr4sr_score_bce_bwd / dr4sr_score_bpr_bwd (the scorer of the dense C-ABI composition: MetaModel around GRU4Rec / FMLP, the autograd path) in
    deterministic mode: d_query as always, the table gradient by owner waves in position order (csrc/score.hip k_score_dense_owner) — two calls
    bit-identical, equal to the default mode's atomics up to summation order; few distinct ids so that many positions meet in every row"""
    from dr4sr_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    B, L, N = 200, 50, 300
    q = torch.randn(B, L, D, generator=g).to(dev)
    E = (0.3 * torch.randn(N, D, generator=g)).to(dev)
    tgt = torch.randint(0, N, (B, L), generator=g).to(dev)             # zeros = padded positions
    neg = torch.randint(1, N, (B, L), generator=g).to(dev)
    w = torch.rand(B * L, generator=g).to(dev)
    fn = lib.dr4sr_score_bpr_bwd if bpr else lib.dr4sr_score_bce_bwd

    def run():
        dq, dE = torch.empty_like(q), torch.zeros_like(E)
        _lib.check(fn(_lib.ptr(q), _lib.ptr(E), _lib.ptr(tgt), _lib.ptr(neg), _lib.ptr(w), None, _lib.ptr(dq), _lib.ptr(dE), B, L, D, _lib.cur_stream()), "score_bwd")
        torch.cuda.synchronize()
        return dq, dE
    dq0, dE0 = run()
    monkeypatch.setenv("DR4SR_DETERMINISTIC", "1")
    dq1, dE1 = run()
    dq2, dE2 = run()
    assert torch.equal(dE1, dE2) and torch.equal(dq1, dq2) and torch.equal(dq1, dq0)
    assert float((dE1 - dE0).abs().max()) < 1e-5 * float(dE0.abs().max()) and float(dE0.abs().max()) > 0


@pytest.mark.parametrize("name", ["SASRec", "MetaModel", "CL4SRec", "FMLP", "GRU4Rec", "MetaModel:GRU4Rec", "MetaModel:FMLP", "MetaModel:CL4SRec"])
def test_whole_fit_is_bit_identical_under_train_deterministic(name):
    """two complete fit() calls (3 epochs of B = 256 on 1 024 toys-sized rows, dropout 0.5, validation every epoch; MetaModel: warm-up epoch +
    an outer hyper-gradient step every 2 steps; CL4SRec: two drawn views + InfoNCE per step) end with bit-identical parameters (and meta
    module) — tools/det_fit_check.py, a fresh process per pair so that the mode is set before the library caches its switches"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DET_N_ITEMS="11925", DET_ROWS="1024", DET_BATCH="256")
    env.pop("DR4SR_DETERMINISTIC", None)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "det_fit_check.py"), name], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "identical: True" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
