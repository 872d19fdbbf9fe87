"""Round-2 parity tests for kernel paths that shipped without an oracle check ("parity gaps"):

  * GRU4Rec at B = 256 / 400 / 1024: the second bank of cooperative groups (grp >= 8, csrc/gru_coop.hip) and the single-workgroup
    recurrence k_gru_fwd/_bwd (csrc/gru.hip) that serves every batch with more than 24 groups; DR4SR_GRU_NOCOOP at a small batch.
  * SASRec throughput mode (B = 8192, persistent length-class attention lists, scatter-as-wgrad-job) against the ORACLE, not
    against another launch form of the same library.
  * every cross-check switch of DESIGN.md §5a that is read once per process (`static const ... getenv`): the oracle tests of
    tests/test_gpu_parity.py re-run in a subprocess with the switch set.
  * MetaModel hyper-gradient against oracle/metamodel_oracle.py's exact double-backward for GRU4Rec and FMLP sub-models, and for
    SASRec on TRAINED weights (200 Adam steps), where parameter norms are larger than at initialisation.

Reference arithmetic: /root/reference module/layers.py:117-136 (GRU), utils/utils.py:145-205 (Hypergrad)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gru4rec_oracle as GO  # noqa: E402
from oracle import metamodel_oracle as MO  # noqa: E402
from oracle import sasrec_oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    a = a.detach().cpu().double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


# ------------------------------------------------------------------------------------------------ GRU4Rec
def _gru_batch(B, N, L, seed, toys=True):
    from dr4sr_amd.data.synthetic import make_rows
    rows = make_rows(n_rows=B, n_items=N, seed=seed, dense=not toys)
    b = {k: torch.from_numpy(rows[k]) for k in ("in_item_id", "item_id", "seqlen")}
    gen = torch.Generator().manual_seed(seed + 1)
    b["neg_item"] = torch.randint(1, N, (B, L, 1), generator=gen)
    return b, gen


def _gru_vs_oracle(B, H, NL, seed=11, toys=True, rel=3e-4, N=2000):
    from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
    L = 50
    b, gen = _gru_batch(B, N, L, seed, toys)
    params = {}
    for nme, shp in zip(gru_param_names(NL), gru_param_shapes(N, 64, H, NL)):
        params[nme] = 0.08 * torch.randn(shp, generator=gen)
    params["item_embedding.weight"][0] = 0
    eng = GruEngine(N, L, 64, H, NL, 0.0, B, "cuda", seed=5)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    eng.check_device_error()
    op = dict(params)
    op["query_encoder.0.1.weight"] = params["item_embedding.weight"]
    loss_o, _, grads_o = GO.grads_of(op, b, NL)
    loss, n = eng.loss_and_count()
    assert n == int((b["item_id"] != 0).sum()) and abs(loss - float(loss_o)) < 3e-5, (loss, float(loss_o))
    worst = 0.0
    for k, gv in eng.normalized_grads().items():
        e = relerr(gv, grads_o[k])
        worst = max(worst, e)
        assert e < rel, (k, e)
    return eng, worst


@pytest.mark.parametrize("B", [130, 256, 384])
def test_gru4rec_cooperative_second_bank_vs_oracle(B):
    """9 / 16 / 24 groups of 16 sequences: groups 8.. map to the second bank of cooperative workgroups
    (grp = (jx / NS) * 8 + xcd, csrc/gru_coop.hip); 256 is the bench / BASELINE configs[2] batch"""
    from dr4sr_amd import _lib
    eng, worst = _gru_vs_oracle(B, 256, 2)
    assert eng.uses_cooperative(B), "this batch size is expected on the cooperative path"
    print("GRU coop B=%d worst grad relerr %.2e" % (B, worst))


@pytest.mark.parametrize("B,H,toys", [(400, 256, True), (1024, 256, True), (448, 128, False)])
def test_gru4rec_single_workgroup_recurrence_vs_oracle(monkeypatch, B, H, toys):
    """k_gru_fwd / k_gru_bwd (csrc/gru.hip), W_hh streamed from L2: the recurrence of every batch above 1536 sequences (and of any batch
    under DR4SR_GRU_NOCOOP, read per call — how these sizes reach it here)"""
    monkeypatch.setenv("DR4SR_GRU_NOCOOP", "1")
    eng, worst = _gru_vs_oracle(B, H, 2, toys=toys)
    assert not eng.uses_cooperative(B)
    print("GRU single-workgroup B=%d H=%d worst grad relerr %.2e" % (B, H, worst))


@pytest.mark.parametrize("B,H,toys", [(400, 256, True), (1024, 256, True), (448, 128, False), (1536, 256, True)])
def test_gru4rec_chunked_cooperative_recurrence_vs_oracle(B, H, toys):
    """more than 24 groups, at most 1536 sequences: consecutive cooperative launches over chunks of 256 sequences (csrc/gru_coop.hip
    coop_chunk) — full chunks, a ragged last chunk (400 = 256 + 144, 448 = 256 + 192), the largest chunk count (6)"""
    eng, worst = _gru_vs_oracle(B, H, 2, toys=toys)
    assert eng.uses_cooperative(B)
    print("GRU chunked cooperative B=%d H=%d worst grad relerr %.2e" % (B, H, worst))


def test_gru4rec_nocoop_switch_small_batch_vs_oracle(monkeypatch):
    """DR4SR_GRU_NOCOOP=1 (read per call): a batch that would run cooperatively takes the single-workgroup recurrence"""
    monkeypatch.setenv("DR4SR_GRU_NOCOOP", "1")
    eng, _ = _gru_vs_oracle(64, 256, 2)
    assert not eng.uses_cooperative(64)
    monkeypatch.delenv("DR4SR_GRU_NOCOOP")
    eng2, _ = _gru_vs_oracle(64, 256, 2)
    assert eng2.uses_cooperative(64)
    for k, v in eng.normalized_grads().items():
        assert relerr(v, eng2.normalized_grads()[k].cpu()) < 2e-5, k


# ------------------------------------------------------------------------------------------------ SASRec throughput mode
@pytest.mark.parametrize("dense", [False, True])
def test_sasrec_throughput_mode_B8192_vs_oracle(dense, at_scale):
    """B = 8192 (toys histogram: 44 k tokens; dense: 410 k tokens): BM = 32 token tiles, persistent short / long attention lists
    walked several entries per workgroup, embedding scatter as a k_wgrad job — loss and EVERY gradient against the dense
    oracle's autograd (128 host threads: seconds)"""
    from test_gpu_parity import _random_params, _toys_batch
    from dr4sr_amd.engine import SasrecEngine
    B = 8192 if not dense else 2048                      # dense oracle: [2048, 2, 50, 50] scores fwd+bwd stays in seconds
    b, N = _toys_batch(B, dense, seed=21)
    params = _random_params(N, 64, 128, 2, seed=4)
    eng = SasrecEngine(N, 50, 64, 2, 128, 2, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
    assert n == int((b["item_id"] != 0).sum())
    assert abs(loss - float(loss_o)) < 2e-5, (loss, float(loss_o))
    worst = 0.0
    for k, gv in eng.normalized_grads().items():
        e = relerr(gv, grads_o[k])
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print("SASRec B=%d dense=%s worst grad relerr %.2e" % (B, dense, worst))


@pytest.mark.parametrize("D,B,p", [(64, 2048, 0.0), (64, 2048, 0.5), (128, 1024, 0.3)])
def test_tiny_attention_class_equals_mfma_kernels(monkeypatch, at_scale, D, B, p):
    """sequences of 1..8 tokens run on the VALU kernels of csrc/attn_tiny_body.h in the split launches; DR4SR_ATTN_NOTINY sends the same
    list through the 16-row MFMA kernels: identical statistics / dropout element indexing, so losses and gradients agree to fp32
    summation order — also with dropout ON (the two classes regenerate the same Philox masks).  Batch with every length 1..8
    present, PAD items inside sequences (key-padding mask) and both head widths.  (Round 6: under DR4SR_ATTN_LISTS — the lists are the
    cross-check form of the wave-per-tile launches.)"""
    from test_gpu_parity import _random_params, _toys_batch
    monkeypatch.setenv("DR4SR_ATTN_LISTS", "1")
    from dr4sr_amd.engine import SasrecEngine
    b, N = _toys_batch(B, False, seed=33, n_items=3000)
    for r, n in enumerate([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 17]):       # both sides of every class boundary
        b["seqlen"][r] = n
        b["in_item_id"][r] = 0
        b["item_id"][r] = 0
        b["in_item_id"][r, :n] = torch.arange(1, n + 1)
        b["item_id"][r, :n] = torch.arange(2, n + 2)
    b["in_item_id"][4, 1] = 0                                         # a PAD id INSIDE a 5-token sequence: key 1 is masked for every query
    b["in_item_id"][7, 3] = 0
    params = _random_params(N, D, 128, 2, seed=6)
    eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, p, B, "cuda", seed=9)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    loss_t, n_t = eng.loss_and_count()
    g_tiny = {k: v.clone() for k, v in eng.normalized_grads().items()}
    if p == 0.0:
        loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
        assert abs(loss_t - float(loss_o)) < 2e-5
        for k, gv in g_tiny.items():
            assert relerr(gv, grads_o[k]) < 2e-4, k
    eng.state[3] -= 1                                                 # replay the same RNG step
    monkeypatch.setenv("DR4SR_ATTN_NOTINY", "1")
    eng.fwd_bwd(plan)
    loss_m, n_m = eng.loss_and_count()
    assert n_m == n_t and abs(loss_m - loss_t) < 1e-5
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, g_tiny[k].cpu()) < 2e-5, k
    # the forward runs the 1..8- and 9..16-token classes as ONE launch (k_attn_small_fwd); DR4SR_ATTN_NOMERGE: one launch per class
    monkeypatch.delenv("DR4SR_ATTN_NOTINY")
    monkeypatch.setenv("DR4SR_ATTN_NOMERGE", "1")
    eng.state[3] -= 1
    eng.fwd_bwd(plan)
    loss_s, n_s = eng.loss_and_count()
    assert n_s == n_t and abs(loss_s - loss_t) < 1e-6
    for k, gv in eng.normalized_grads().items():
        assert relerr(gv, g_tiny[k].cpu()) < 2e-5, k


_OWNER_CASES = [(64, 2048, 0.0, 3000), (64, 4096, 0.5, 11925), (128, 1024, 0.2, 20034), (64, 1024, 0.0, 70000), (64, 2048, 0.0, 30000)]


@pytest.mark.parametrize("D,B,p,n_items", _OWNER_CASES)
def test_owner_computed_table_gradient(monkeypatch, at_scale, D, B, p, n_items):
    """large batches: the item-table gradient is summed row by row by owner workgroups inside k_wgrad (csrc/linear.hip owner_job)
    instead of fp32 atomics from the scorer and the embedding scatter.  (1) it equals the atomic path (DR4SR_DE_ATOMIC) to fp32
    summation order — with dropout too; (2) it is a pure function of the batch: two replays give BIT-identical table gradients
    (the reference asks cudnn for determinism, utils/utils.py:19); (3) PAD row and untouched rows stay exactly zero; (4) vs the oracle"""
    from test_gpu_parity import _random_params, _toys_batch
    from dr4sr_amd.engine import SasrecEngine
    b, N = _toys_batch(B, False, seed=41, n_items=n_items)
    params = _random_params(N, D, 128, 2, seed=8)
    eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, p, B, "cuda", seed=3)
    eng.load_named(params)
    dev = eng.device
    plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    eng.fwd_bwd(plan)
    g1 = eng.grads.clone()
    eng.state[3] -= 1
    eng.fwd_bwd(plan)
    g2 = eng.grads.clone()
    nE = N * D
    assert torch.equal(g1[:nE], g2[:nE]), "owner-computed dE must be bit-reproducible"
    gE = g1[:nE].view(N, D).cpu()
    touched = torch.zeros(N, dtype=torch.bool)
    valid = b["item_id"] != 0
    touched[b["item_id"][valid]] = True
    touched[b["neg_item"].squeeze(-1)[valid]] = True
    touched[b["in_item_id"][b["in_item_id"] != 0]] = True
    assert float(gE[0].abs().max()) == 0.0 and float(gE[~touched].abs().max()) == 0.0
    if p == 0.0 and N <= 20034:
        loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
        for k, gv in eng.normalized_grads().items():
            assert relerr(gv, grads_o[k]) < 2e-4, k
    eng.state[3] -= 1
    monkeypatch.setenv("DR4SR_DE_ATOMIC", "1")
    eng.fwd_bwd(plan)
    ga = eng.grads.clone()
    assert relerr(g1[:nE], ga[:nE].cpu()) < 2e-5
    assert relerr(g1[nE:], ga[nE:].cpu()) < 2e-5
    # (5) the two feeds of the owner job — entries sorted by owner per token tile (k_post_mid's tile_sort, the default up to 1024
    # owners) and owners scanning every record (DR4SR_OWNER_SCAN; the only form above 1024 owners: the 70 000-item case) — agree
    monkeypatch.delenv("DR4SR_DE_ATOMIC")
    monkeypatch.setenv("DR4SR_OWNER_SCAN", "1")
    eng.state[3] -= 1
    eng.fwd_bwd(plan)
    gs = eng.grads.clone()
    eng.state[3] -= 1
    eng.fwd_bwd(plan)
    assert torch.equal(gs[:nE], eng.grads[:nE]), "scanning owners must be bit-reproducible too"
    assert relerr(g1[:nE], gs[:nE].cpu()) < 2e-5 and relerr(g1[nE:], gs[nE:].cpu()) < 2e-5


@pytest.mark.parametrize("n_items", [3000, 11925, 30000])
def test_owner_sorted_entries_with_64_row_tiles(monkeypatch, n_items):
    """DR4SR_BM=64: tile_sort with 192 entries per tile (three per lane of the sorting wave) at scale — the owner-computed table gradient
    tests of this file re-run with 64-row tiles"""
    monkeypatch.setenv("DR4SR_BM", "64")
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")
    for D, B, p, n in _OWNER_CASES:
        if n == n_items:
            test_owner_computed_table_gradient(monkeypatch, None, D, B, p, n)


def test_fuzz_large_batches_vs_oracle(monkeypatch):
    """tests/fuzz_scale.py: random at-scale batches (all-tiny / all-long / class-boundary / toys-like length mixes, catalogs of 2 ..
    11 925 items, both widths, PAD targets, PAD ids inside sequences) through the fused step vs the oracle"""
    import fuzz_scale
    monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")
    assert fuzz_scale.main(trials=10, seed=5) < 5e-4


# ------------------------------------------------------------------------------------------------ cross-check switches
# The oracle-backed tests of tests/test_gpu_parity.py / test_gpu_api.py that the switch matrix re-runs, by group name.  They run IN THIS
# PROCESS: the library re-reads its DR4SR_* switches after dr4sr_reload_env() (conftest hooks it to monkeypatch.setenv), so a switch case
# no longer costs a fresh interpreter (torch import + HIP context + module load, ~25 s each on the driver's box), and the oracle's
# gradients of a (parameters, batch) pair are computed once per session (conftest memoises oracle.sasrec_oracle.grads_of).
def _group_calls(group, dev, golden_dir, monkeypatch):
    import test_gpu_api as A
    import test_gpu_parity as P
    if group == "full_size":
        return [lambda a=a: P.test_full_size_batch_vs_oracle(dev, *a) for a in ((False, 64, None), (True, 64, None), (False, 128, 20034))]
    if group == "full_size_d64":
        return [lambda a=a: P.test_full_size_batch_vs_oracle(dev, *a) for a in ((False, 64, None), (True, 64, None))]
    if group == "fuzz":
        return [lambda: P.test_fuzz_odd_batches_vs_oracle(dev)]
    if group == "trajectory":
        return [lambda: P.test_training_trajectory_matches_oracle(dev)]
    if group == "dropout":
        return [lambda a=a: P.test_fwd_bwd_with_dropout_matches_oracle_with_same_masks(dev, golden_dir, *a) for a in (("sasrec_d64", 0.5), ("sasrec_d128", 0.2))]
    if group == "train_steps":
        return [lambda a=a: A.test_train_steps_equals_repeated_train_step(*a) for a in ((64, 640), (2048, 19412), (9000, 19412))]
    if group == "length_split":
        def run():
            monkeypatch.setenv("DR4SR_FORCE_SCALE", "1")
            P.test_large_batch_length_split_attention(dev, monkeypatch, None)
        return [run]
    raise KeyError(group)


_SWITCH_CASES = [
    # (environment, groups of oracle-backed tests that reach the switch)
    ({"DR4SR_NO_FUSE": "1"}, "full_size fuzz trajectory dropout"),
    ({"DR4SR_ATTN_VALU": "1"}, "full_size fuzz dropout"),
    ({"DR4SR_NO_PREP_FUSE": "1"}, "trajectory train_steps"),
    ({"DR4SR_QEB_SEPARATE": "1"}, "full_size fuzz"),
    ({"DR4SR_SCATTER_INLINE": "1"}, "length_split"),
    ({"DR4SR_ATTN_GRID_FIXED": "1"}, "length_split"),
    ({"DR4SR_ATTN_LISTS": "1", "DR4SR_FORCE_SCALE": "1"}, "full_size fuzz dropout"),      # round 6: the length-class lists instead of the wave-per-tile launches
    ({"DR4SR_BM": "32"}, "full_size fuzz"),                # the at-scale tile on small batches
    ({"DR4SR_BM": "64"}, "full_size_d64"),                 # tuning-only tile (d = 128 fits it up to L = 57 only: refused by the launch)
    # the middle regime (~5.5 k .. 14 k expected tokens): at-scale token-tile kernels with one attention workgroup per sequence
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_FORCE_ATTN_SPLIT": "0"}, "full_size fuzz dropout trajectory train_steps"),
    # ... and the converse (never chosen by the hint, must still be right): latency tiles with the length-class attention lists
    ({"DR4SR_FORCE_SCALE": "0", "DR4SR_FORCE_ATTN_SPLIT": "1"}, "full_size fuzz"),
    # two-phase next-step prep with FOUR optimizer workgroups: each owns 512 / 2 250 consecutive sequences, i.e. several 256-sequence
    # rounds with carried totals inside one workgroup (256 workgroups own 8 / 36) — what B > 65 536 does with the default grid
    ({"DR4SR_ADAM_BLOCKS": "4"}, "train_steps"),
    # round 3 — the at-scale forms of csrc/linear_wave.hip and their cross-checks, each against the oracle (dropout test included: the
    # wave tiles draw 8 decisions per Philox call, the 256-thread kernels 4 of the same 8)
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_NO_WAVE_TILES": "1"}, "full_size fuzz dropout"),
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WT_FWD_ONLY": "1"}, "full_size dropout"),
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WGRAD_F32": "1"}, "full_size trajectory"),
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WT_BF16X3": "1"}, "full_size dropout"),
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WT_FWD_WAVES": "16", "DR4SR_WT_BWD_WAVES": "8", "DR4SR_WT_MID_WAVES": "12", "DR4SR_WT_EMB_WAVES": "12"},
     "full_size fuzz"),
    ({"DR4SR_PREP2_INLINE": "1"}, "train_steps"),
    # round 4: the wave-tile backward recomputing the linear1 pre-activations instead of loading them (opt-in: measured slower)
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WT_RECOMPUTE_A": "1"}, "full_size fuzz dropout"),
    # round 4: attention inside the 16-token tile kernels is the latency regime's default (csrc/attn_tile.h); the one-workgroup-per-sequence
    # launches as the cross-check, and the tile form with every dK | dV row through atomics
    ({"DR4SR_ATTN_SEPARATE": "1"}, "full_size dropout"),       # (DR4SR_NO_FUSE / DR4SR_ATTN_VALU above run the separate launches too)
    ({"DR4SR_ATTN_TILE_ATOMICS": "1"}, "full_size dropout"),
    # round 4: the six whole fp32 weight-gradient jobs per layer instead of their 64 x 64 blocks (k_wgrad instead of k_wgrad_blk)
    ({"DR4SR_WGRAD_BLK": "0"}, "full_size fuzz trajectory"),
    # round 4: the layer-0 in_proj of the wave-tile embedding stage as a bf16x3 split (measured +0.6 %, opt-in)
    ({"DR4SR_FORCE_SCALE": "1", "DR4SR_WT_EMB_BF16X3": "1"}, "full_size dropout"),
    # round 4: tile = blockIdx.x instead of the XCD-aware order in the latency regime's tile kernels / in the weight-gradient launch
    ({"DR4SR_TILE_ORDER_PLAIN": "1"}, "full_size dropout trajectory"),
    ({"DR4SR_WGRAD_ORDER_PLAIN": "1"}, "full_size fuzz"),
    # the 4-wave per-sequence attention backward (head_dim 64 ran on it until round 3)
    ({"DR4SR_ATTN_BWD_4WAVE": "1", "DR4SR_ATTN_SEPARATE": "1"}, "full_size dropout"),
]


@pytest.mark.parametrize("env,groups", _SWITCH_CASES, ids=[",".join(e) for e, _ in _SWITCH_CASES])
def test_cross_check_switches_reproduce_the_oracle(env, groups, golden_dir, monkeypatch):
    """every cross-check / tuning switch of DESIGN.md 5a (csrc/linear.hip, linear_wave.hip, step.hip, attn_mfma.hip) re-runs the
    oracle-backed tests that reach it, in this process (the switches are cached per process and re-read on dr4sr_reload_env)"""
    dev = torch.device("cuda")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ran = 0
    for group in groups.split():
        for call in _group_calls(group, dev, golden_dir, monkeypatch):
            call()
            ran += 1
    assert ran > 0


# ------------------------------------------------------------------------------------------------ MetaModel hyper-gradient
def _meta_model(monkeypatch, sub, n_items=300, batch=48, dropout=0.0):
    from test_gpu_meta import build, make_config
    cfg = make_config(n_items, sub=sub, dropout=dropout, n_rows=400, batch=batch)
    if sub == "FMLP":
        cfg["data"]["prefix_rows"] = True
    if sub == "GRU4Rec":
        cfg["model"]["sub_overrides"]["model"]["hidden_size"] = 128
    ds, model = build(cfg, monkeypatch)
    model.train()
    return cfg, ds, model


def _cpu_batch(b):
    return {k: v.detach().cpu() for k, v in b.items()}


def _hyper_vs_oracle(model, f, tol, label):
    loader = model.dataset_list[0].get_loader()
    perm = model._perm(loader)
    bv, bt = model._local_batch(loader, perm, 0), model._local_batch(loader, perm, 1)
    bv["neg_item"], bt["neg_item"] = model._neg_sampling(bv), model._neg_sampling(bt)
    tgt = bt["item_id"]
    g = torch.Generator().manual_seed(17)
    gum = -torch.empty(*tgt.shape, 2).exponential_(generator=g).log()
    model._gumbel = gum.reshape(-1, 2).contiguous().to(model.device)
    sd = {k: v.detach().cpu().clone() for k, v in model.sub_model.state_dict().items()}
    p = {k: v for k, v in sd.items() if k not in ("query_encoder.item_encoder.weight", "query_encoder.0.1.weight")}
    meta = {k: v.detach().cpu().clone() for k, v in model.meta_module.state_dict().items()}
    tc = model.config["train"]
    ref, _, _ = MO.hypergrad_exact(f, p, meta, _cpu_batch(bt), _cpu_batch(bv), gum, float(model.tau.detach()),
                                   float(model.config["model"]["tau_min"]), float(tc["hpo_learning_rate"]))
    theta = model.engine.params.clone()
    hyper = model.hypergrad(bv, bt)
    assert torch.equal(theta, model.engine.params)
    refv = np.concatenate([ref[k].numpy().ravel() for k in MO.META_NAMES]).astype(np.float64)
    got = hyper.cpu().numpy().astype(np.float64)
    err = float(np.linalg.norm(got - refv) / max(np.linalg.norm(refv), 1e-30))
    print("%s: hyper-gradient rel. error vs exact double-backward %.2e (|ref| %.3e)" % (label, err, np.linalg.norm(refv)))
    assert err < tol, (label, err)
    return err


def test_hypergradient_gru4rec_submodel_vs_oracle(monkeypatch):
    cfg, ds, model = _meta_model(monkeypatch, "GRU4Rec")
    _hyper_vs_oracle(model, MO.gru4rec_losses(2), 1e-3, "MetaModel(GRU4Rec)")


def test_hypergradient_fmlp_submodel_vs_oracle(monkeypatch):
    cfg, ds, model = _meta_model(monkeypatch, "FMLP")
    model.engine.p_drop = 0.0          # FMLP hard-codes dropout 0.5 (fmlp.py:11-13); the exact oracle is the dropout-free function
    _hyper_vs_oracle(model, MO.fmlp_losses(2), 1e-3, "MetaModel(FMLP)")


def test_hypergradient_sasrec_trained_weights_vs_oracle(monkeypatch):
    """200 plain Adam steps first (weights leave the N(0, 0.02) initialisation: table rows grow ~3x, LN affine moves), then the
    finite-difference hyper-gradient must still sit within 1e-3 of the exact one"""
    cfg, ds, model = _meta_model(monkeypatch, "SASRec")
    sub = model.sub_model
    n0 = float(sub.engine.params.norm())
    for ep in range(30):
        sub.training_epoch(ep)
        if int(sub.engine.state[0]) >= 200:
            break
    assert int(sub.engine.state[0]) >= 200
    print("parameter norm %.3f -> %.3f after %d steps" % (n0, float(sub.engine.params.norm()), int(sub.engine.state[0])))
    f = MO.sasrec_losses({"H": 2, "n_layer": 2, "eps": 1e-12})
    _hyper_vs_oracle(model, f, 1e-3, "MetaModel(SASRec, trained)")
