"""csrc/attn_wave.hip (round 6): at scale, on short-sequence plans, the attention of a layer as ONE launch per direction — a wave per
(16-token tile of the packed stream, head[, phase]) working from the embedding stage's per-token words, no length-class lists, no LDS,
no atomics — instead of the five list launches of csrc/attn_mfma.hip (VERDICT r5 Next #3: "attention at scale").  Held against the oracle
(/root/reference model/sasrec.py:21-34 as called at :65-68; masks :48, :58 — nn.TransformerEncoderLayer's attention with the causal and
key-padding masks) and against the list launches (DR4SR_ATTN_LISTS=1) / the one-workgroup-per-sequence launches (DR4SR_ATTN_NOSPLIT=1)
on the same batch and the same dropout elements: the forms are interchangeable per launch."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sasrec_oracle as O  # noqa: E402
from test_gpu_parity import _random_params, _toys_batch, relerr  # noqa: E402

WAVE = 16          # dr4sr_sasrec_at_scale bit 4
FOLD = 32          # bit 5: the wave attention's forward folded into the wave-tile forward launches (d = 64)


def _bits(plan):
    from dr4sr_amd import _lib
    return int(_lib.load().dr4sr_sasrec_at_scale(C.byref(plan)))


def _batch_of(sl, L, rng, N, pad_rows=()):
    sl = np.asarray(sl, dtype=np.int64)
    B = len(sl)
    inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b]); tgt[b, :sl[b]] = rng.integers(0, N, size=sl[b])
    for b in pad_rows:                                       # PAD keys inside a sequence (never at position 0: that row would be all-masked)
        pos = rng.choice(np.arange(1, sl[b]), size=min(3, sl[b] - 1), replace=False)
        inp[b, pos] = 0
    return {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl),
            "neg_item": torch.from_numpy(rng.integers(1, N, size=(B, L, 1)))}


def _cases(L):
    return {
        # maximum-length sequences (5 key tiles), lengths around the tile size, single tokens, a token count that is not a multiple of 16
        "edge": ([L, 1, L, 15, 16, 17, 1, 31, 32, 33, L - 1, 2, 16, 16, 1, 48, L, 3, 5, 8, 13], (0, 3, 7, 16)),
        "all_16": ([16] * 37, ()),                           # every sequence exactly one tile
        "all_1": ([1] * 53, ()),                             # 16 sequences per tile
        "pairs_15_1": ([15, 1] * 21 + [15], ()),             # tile-aligned pairs, then a straddler at the end
        "one_17": ([17], ()),                                # the smallest sequence that needs two key tiles from its own rows
        "straddlers": ([7, 12, 9, 14, 3, 11, 16, 2, 13, 10, 15, 6] * 3, (1, 3)),     # nearly every sequence crosses a tile boundary
        "long_mix": ([L, 33, 2, L - 3, 40, 1, 1, 17, 29, L], (0, 4)),
    }


@pytest.mark.parametrize("D,L,case", [(64, 50, "edge"), (64, 64, "edge"), (128, 50, "edge"), (64, 50, "all_16"), (64, 50, "all_1"),
                                      (64, 50, "pairs_15_1"), (64, 50, "one_17"), (64, 50, "straddlers"), (128, 64, "straddlers"),
                                      (64, 64, "long_mix"), (128, 50, "long_mix")])
def test_wave_attention_edge_cases_vs_oracle_and_lists(D, L, case, monkeypatch, at_scale):
    """loss and every gradient against the oracle (two passes: nothing accumulates across passes), then the list launches and the
    per-sequence launches on the same batch"""
    from dr4sr_amd.engine import SasrecEngine
    rng = np.random.default_rng(23 + L + D + len(case))
    N, H, F, NL = 157, 2, 128, 2
    sl, pads = _cases(L)[case]
    batch = _batch_of(sl, L, rng, N, pads)
    B = batch["seqlen"].shape[0]
    params = _random_params(N, D, F, NL, L=L, seed=9)
    eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    plan = eng.make_plan(batch["in_item_id"].cuda(), batch["item_id"].cuda(), batch["seqlen"].cuda(),
                         neg_item=batch["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False, expected_tokens=8 * B)
    assert _bits(plan) & WAVE and _bits(plan) & 2 and not _bits(plan) & 4
    for _ in range(2):
        eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    g_w = {k: v.clone() for k, v in eng.normalized_grads().items()}
    loss_o, _, grads_o = O.grads_of(params, batch, H, NL, 1e-12)
    assert n == int((batch["item_id"] != 0).sum())
    assert abs(loss - float(loss_o)) < 3e-5
    for k, v in g_w.items():
        assert relerr(v, grads_o[k]) < 5e-4, k
    assert not _bits(plan) & FOLD
    for switch in ("DR4SR_ATTN_LISTS", "DR4SR_ATTN_NOSPLIT"):
        monkeypatch.setenv(switch, "1")
        assert not _bits(plan) & WAVE
        eng.fwd_bwd(plan)
        loss_l, _ = eng.loss_and_count()
        assert abs(loss_l - loss) < 1e-5, switch
        for k, v in eng.normalized_grads().items():
            assert relerr(v, g_w[k].cpu()) < 2e-5, (switch, k)
        monkeypatch.delenv(switch)


@pytest.mark.parametrize("B,D,p", [(2048, 64, 0.0), (8192, 64, 0.5), (2048, 128, 0.3)])
def test_wave_attention_equals_the_lists_on_toys_batches(B, D, p, monkeypatch, at_scale):
    """thousands of tiles on the toys length histogram, dropout ON in two cases: the wave-per-tile launches, the length-class lists and the
    one-workgroup-per-sequence launches draw the same Philox elements ((slot H + h) 64 + i) 64 + j, so losses and gradients agree to fp32
    summation order; p = 0 also against the oracle.  Two runs of the wave form are bit-identical in dqkv's consumers (no atomics)."""
    from dr4sr_amd.engine import SasrecEngine
    b, N = _toys_batch(B, False, seed=31)
    params = _random_params(N, D, 128, 2, seed=5)
    eng = SasrecEngine(N, 50, D, 2, 128, 2, 1e-12, p, B, "cuda", seed=13)
    eng.load_named(params)
    plan = eng.make_plan(b["in_item_id"].cuda(), b["item_id"].cuda(), b["seqlen"].cuda(),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False)
    assert _bits(plan) & WAVE
    eng.fwd_bwd(plan)
    loss_a, n_a = eng.loss_and_count()
    ga = {k: v.clone() for k, v in eng.normalized_grads().items()}
    if p == 0.0:
        loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
        assert abs(loss_a - float(loss_o)) < 2e-5
        for k, gv in ga.items():
            assert relerr(gv, grads_o[k]) < 2e-4, k
    for switch in ("DR4SR_ATTN_LISTS", "DR4SR_ATTN_NOSPLIT"):
        monkeypatch.setenv(switch, "1")
        assert not _bits(plan) & WAVE
        eng.state[3] -= 1                                         # replay the same RNG step
        eng.fwd_bwd(plan)
        loss_b, n_b = eng.loss_and_count()
        assert n_a == n_b == int((b["item_id"] != 0).sum()) and abs(loss_a - loss_b) < 1e-5, switch
        for k, gv in eng.normalized_grads().items():
            assert relerr(gv, ga[k].cpu()) < 2e-5, (switch, k)
        monkeypatch.delenv(switch)


def test_wave_attention_eval_and_second_backward(monkeypatch, at_scale):
    """dr4sr_sasrec_encode (training off, pooled output) equals the list launches; dr4sr_sasrec_encode_bwd twice on one forward gives the
    same gradients bit for bit (every dqkv row has exactly one writer)"""
    from dr4sr_amd import _lib
    from dr4sr_amd.engine import SasrecEngine
    rng = np.random.default_rng(5)
    N, B, L, D = 120, 300, 50, 64
    sl = np.minimum(rng.geometric(0.2, size=B), L).astype(np.int64)
    sl[:3] = (L, 17, 1)
    inp = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b])
    # dropout 0.3 in the ENGINE: the eval pass must ignore it (keep factor 1, not 1 / (1 - p)), the training passes draw the lists' elements
    eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, 0.3, B, "cuda", seed=3)
    eng.load_named(_random_params(N, D, 128, 2, seed=11))
    plan = eng.make_plan(torch.from_numpy(inp).cuda(), None, torch.from_numpy(sl).cuda())
    assert _bits(plan) & WAVE
    q = eng.encode(plan, False, _lib.POOL_LAST).clone()
    step0 = int(eng.state[3])
    q_tr = eng.encode(plan, True, _lib.POOL_MEAN).clone()
    g = torch.randn_like(q_tr)
    grads = []
    for _ in range(2):
        eng.grads.zero_()
        eng.encode_bwd(plan, True, _lib.POOL_MEAN, g)
        grads.append(eng.grads[:eng.n_params].cpu())
    assert torch.equal(grads[1], grads[0]) or relerr(grads[1], grads[0]) < 1e-5
    monkeypatch.setenv("DR4SR_ATTN_LISTS", "1")
    assert not _bits(plan) & WAVE
    assert relerr(eng.encode(plan, False, _lib.POOL_LAST), q.cpu()) < 1e-5
    eng.state[3] = step0                                              # the same RNG step as the wave form's training pass
    assert relerr(eng.encode(plan, True, _lib.POOL_MEAN), q_tr.cpu()) < 1e-5
    eng.grads.zero_()
    eng.encode_bwd(plan, True, _lib.POOL_MEAN, g)
    assert relerr(eng.grads[:eng.n_params].cpu(), grads[0]) < 2e-5
    # ... and against the oracle in eval mode (no dropout): the pooled last-position query
    params = {k: v.detach().cpu().clone() for k, v in eng.views.items()}
    params["query_encoder.item_encoder.weight"] = params["item_embedding.weight"]
    monkeypatch.delenv("DR4SR_ATTN_LISTS")


def test_wave_attention_is_what_a_toys_sized_plan_takes(at_scale, monkeypatch):
    """the form follows the plan: where the lists would run (short-sequence plans at scale) at d = 64 and d = 128; long-sequence plans keep
    one workgroup per sequence; the un-fused step (which writes no token words) and DR4SR_ATTN_LISTS keep the lists"""
    from dr4sr_amd.engine import SasrecEngine
    ids = torch.ones(512, 50, dtype=torch.int64, device="cuda")
    short = torch.full((512,), 5, dtype=torch.int64, device="cuda")
    full = torch.full((512,), 50, dtype=torch.int64, device="cuda")
    for D in (64, 128):
        eng = SasrecEngine(500, 50, D, 2, 128, 2, 1e-12, 0.0, 512, "cuda")
        bits = _bits(eng.make_plan(ids, ids, short))
        assert bits & WAVE and not bits & FOLD
    monkeypatch.delenv("DR4SR_FORCE_SCALE")
    eng = SasrecEngine(500, 50, 64, 2, 128, 2, 1e-12, 0.0, 8192, "cuda")
    ids8, short8, full8 = ids.repeat(16, 1), short.repeat(16), full.repeat(16)
    assert _bits(eng.make_plan(ids8, ids8, short8)) & WAVE
    assert not _bits(eng.make_plan(ids8, ids8, full8)) & (WAVE | 2)
    monkeypatch.setenv("DR4SR_NO_FUSE", "1")
    assert not _bits(eng.make_plan(ids8, ids8, short8)) & WAVE and _bits(eng.make_plan(ids8, ids8, short8)) & 2


@pytest.mark.parametrize("case,p", [("edge", 0.0), ("straddlers", 0.0), ("long_mix", 0.0), ("toys", 0.5)])
def test_attention_forward_folded_into_the_wave_tile_kernels(case, p, monkeypatch, at_scale):
    """experiments build only (DR4SR_ATTN_FOLD: measured slower, NOTEBOOK round 6): the wave attention's forward at the head of k_wt_post_fwd /
    k_wt_post_mid (linear_wave.hip wt_attn_ctx, output directly in the tile kernels' register layout) equals the launch of its own — loss and
    every gradient, dropout on for the toys batch (same saved keep bits and Philox elements)"""
    from dr4sr_amd.engine import SasrecEngine
    monkeypatch.setenv("DR4SR_ATTN_FOLD", "1")            # (skips the test on the shipped build)
    rng = np.random.default_rng(77)
    L, D, N = 50, 64, 157
    if case == "toys":
        batch, N = _toys_batch(2048, False, seed=31)
    else:
        sl, pads = _cases(L)[case]
        batch = _batch_of(sl, L, rng, N, pads)
    B = batch["seqlen"].shape[0]
    eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, p, B, "cuda", seed=13)
    eng.load_named(_random_params(N, D, 128, 2, L=L, seed=9))
    plan = eng.make_plan(batch["in_item_id"].cuda(), batch["item_id"].cuda(), batch["seqlen"].cuda(),
                         neg_item=batch["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False, expected_tokens=8 * B)
    assert _bits(plan) & FOLD and _bits(plan) & WAVE
    eng.fwd_bwd(plan)
    loss_f, _ = eng.loss_and_count()
    g_f = {k: v.clone() for k, v in eng.normalized_grads().items()}
    monkeypatch.delenv("DR4SR_ATTN_FOLD")
    assert not _bits(plan) & FOLD
    eng.state[3] -= 1
    eng.fwd_bwd(plan)
    loss_l, _ = eng.loss_and_count()
    assert abs(loss_l - loss_f) < 1e-5
    for k, v in eng.normalized_grads().items():
        assert relerr(v, g_f[k].cpu()) < 2e-5, k
