"""csrc/attn_tile_sa.hip (round 4, opt-in: DR4SR_ATTN_WINDOW=1): at scale, on short-sequence plans at d = 64, the window attention of
csrc/attn_tile.h as ONE launch per layer and direction instead of the two / three length-class list launches of csrc/attn_mfma.hip
(built for VERDICT r3 #3, measured slower than the lists — NOTEBOOK round 4 — and kept as a tested alternative form).  Held against the
oracle (model/sasrec.py:21-34, :48, :58 — nn.TransformerEncoderLayer's attention with the causal and key-padding masks) and against the
list launches (the default) / the one-workgroup-per-sequence launches (DR4SR_ATTN_NOSPLIT=1) on the same batch and the same dropout
elements."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sasrec_oracle as O  # noqa: E402
from test_gpu_parity import _random_params, _toys_batch, relerr  # noqa: E402


@pytest.fixture(autouse=True)
def _window_form(monkeypatch):
    monkeypatch.setenv("DR4SR_ATTN_WINDOW", "1")


def _bits(plan):
    from dr4sr_amd import _lib
    return int(_lib.load().dr4sr_sasrec_at_scale(C.byref(plan)))


def _edge_batch(L, rng, N):
    sl = np.array([L, 1, L, 15, 16, 17, 1, 31, 32, 33, L - 1, 2, 16, 16, 1, 48, L, 3, 5, 8, 13], dtype=np.int64)
    B = len(sl)
    inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b]); tgt[b, :sl[b]] = rng.integers(0, N, size=sl[b])
    for b in (0, 3, 7, 16):                                  # PAD keys inside a sequence (never at position 0: that row would be all-masked)
        pos = rng.choice(np.arange(1, sl[b]), size=min(3, sl[b] - 1), replace=False)
        inp[b, pos] = 0
    return {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl),
            "neg_item": torch.from_numpy(rng.integers(1, N, size=(B, L, 1)))}


@pytest.mark.parametrize("L,env", [(64, {}), (50, {}), (64, {"DR4SR_ATTN_TILE_FULL": "1"}), (50, {"DR4SR_ATTN_TILE_ATOMICS": "1"}),
                                   (64, {"DR4SR_ATTN_SA_WPC": "1"}), (50, {"DR4SR_NO_WAVE_TILES": "1"})])
def test_window_attention_launches_edge_cases(L, env, monkeypatch, at_scale):
    """maximum-length sequences (every tile of theirs fetches the far rows of its window on demand — the plan is DECLARED short through
    expected_tokens, which is what selects this form), lengths around the tile size, single tokens, PAD ids inside sequences, a token
    count that is not a multiple of 16; two passes (the second finds the first one's dK | dV in the workspace); the whole window staged
    at once (DR4SR_ATTN_TILE_FULL), every dK | dV row through atomics (DR4SR_ATTN_TILE_ATOMICS), one workgroup per CU (several tiles per
    workgroup: the persistent loop), and under the 256-thread token-tile kernels (DR4SR_NO_WAVE_TILES) — loss and every gradient against
    the oracle, then the list launches on the same batch."""
    from dr4sr_amd.engine import SasrecEngine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(23 + L)
    N, D, H, F, NL = 157, 64, 2, 128, 2
    batch = _edge_batch(L, rng, N)
    B = batch["seqlen"].shape[0]
    assert int(batch["seqlen"].sum()) % 16 != 0
    params = _random_params(N, D, F, NL, L=L, seed=9)
    eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    plan = eng.make_plan(batch["in_item_id"].cuda(), batch["item_id"].cuda(), batch["seqlen"].cuda(),
                         neg_item=batch["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False, expected_tokens=8 * B)
    assert _bits(plan) & 8 and _bits(plan) & 2 and not _bits(plan) & 4
    for _ in range(2):
        eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    g_sa = {k: v.clone() for k, v in eng.normalized_grads().items()}
    loss_o, _, grads_o = O.grads_of(params, batch, H, NL, 1e-12)
    assert n == int((batch["item_id"] != 0).sum())
    assert abs(loss - float(loss_o)) < 3e-5
    for k, v in g_sa.items():
        assert relerr(v, grads_o[k]) < 5e-4, k
    monkeypatch.delenv("DR4SR_ATTN_WINDOW")
    assert not _bits(plan) & 8 and _bits(plan) & 2
    eng.fwd_bwd(plan)
    loss_l, _ = eng.loss_and_count()
    assert abs(loss_l - loss) < 1e-5
    for k, v in eng.normalized_grads().items():
        assert relerr(v, g_sa[k].cpu()) < 2e-5, k


@pytest.mark.parametrize("B,p", [(2048, 0.0), (8192, 0.5)])
def test_window_attention_launches_equal_the_lists_on_toys_batches(B, p, monkeypatch, at_scale):
    """thousands of 16-token tiles (every workgroup of the persistent grid walks several) on the toys length histogram, with dropout ON in
    the larger case: the window launches, the length-class lists and the one-workgroup-per-sequence launches draw the same Philox elements,
    so losses and gradients agree to fp32 summation order; p = 0 also against the oracle"""
    from dr4sr_amd.engine import SasrecEngine
    b, N = _toys_batch(B, False, seed=31)
    params = _random_params(N, 64, 128, 2, seed=5)
    eng = SasrecEngine(N, 50, 64, 2, 128, 2, 1e-12, p, B, "cuda", seed=13)
    eng.load_named(params)
    plan = eng.make_plan(b["in_item_id"].cuda(), b["item_id"].cuda(), b["seqlen"].cuda(),
                         neg_item=b["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False)
    assert _bits(plan) & 8
    eng.fwd_bwd(plan)
    loss_a, n_a = eng.loss_and_count()
    ga = {k: v.clone() for k, v in eng.normalized_grads().items()}
    if p == 0.0:
        loss_o, _, grads_o = O.grads_of(params, b, 2, 2, 1e-12)
        assert abs(loss_a - float(loss_o)) < 2e-5
        for k, gv in ga.items():
            assert relerr(gv, grads_o[k]) < 2e-4, k
    for switch in ("DR4SR_ATTN_WINDOW", "DR4SR_ATTN_NOSPLIT"):     # the lists (window form off) / one workgroup per sequence
        if switch == "DR4SR_ATTN_WINDOW":
            monkeypatch.delenv(switch)
        else:
            monkeypatch.setenv(switch, "1")
        assert not _bits(plan) & 8
        eng.state[3] -= 1                                         # replay the same RNG step
        eng.fwd_bwd(plan)
        loss_b, n_b = eng.loss_and_count()
        assert n_a == n_b == int((b["item_id"] != 0).sum()) and abs(loss_a - loss_b) < 1e-5, switch
        for k, gv in eng.normalized_grads().items():
            assert relerr(gv, ga[k].cpu()) < 2e-5, (switch, k)


def test_window_attention_launches_eval_and_second_backward(monkeypatch, at_scale):
    """dr4sr_sasrec_encode (training off, pooled output) equals the list launches; dr4sr_sasrec_encode_bwd twice on one forward gives the
    same gradients (the top layer's dK | dV rows are zeroed by k_pack, the lower layers' by the backward launch above them)"""
    from dr4sr_amd import _lib
    from dr4sr_amd.engine import SasrecEngine
    rng = np.random.default_rng(5)
    N, B, L, D = 120, 300, 50, 64
    sl = np.minimum(rng.geometric(0.2, size=B), L).astype(np.int64)
    sl[:3] = (L, 17, 1)
    inp = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b])
    eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, 0.0, B, "cuda")
    eng.load_named(_random_params(N, D, 128, 2, seed=11))
    plan = eng.make_plan(torch.from_numpy(inp).cuda(), None, torch.from_numpy(sl).cuda())
    assert _bits(plan) & 8
    q = eng.encode(plan, False, _lib.POOL_LAST).clone()
    q_tr = eng.encode(plan, True, _lib.POOL_MEAN).clone()
    g = torch.randn_like(q_tr)
    grads = []
    for _ in range(2):
        eng.grads.zero_()
        eng.encode_bwd(plan, True, _lib.POOL_MEAN, g)
        grads.append(eng.grads[:eng.n_params].cpu())
    assert relerr(grads[1], grads[0]) < 1e-5
    monkeypatch.delenv("DR4SR_ATTN_WINDOW")
    assert not _bits(plan) & 8
    assert relerr(eng.encode(plan, False, _lib.POOL_LAST), q.cpu()) < 1e-5
    eng.encode(plan, True, _lib.POOL_MEAN)
    eng.grads.zero_()
    eng.encode_bwd(plan, True, _lib.POOL_MEAN, g)
    assert relerr(eng.grads[:eng.n_params].cpu(), grads[0]) < 2e-5


def test_window_attention_launches_only_on_short_sequence_plans(monkeypatch, at_scale):
    """the form follows the plan: long-sequence batches (expected mean length > 16) and d = 128 keep the lists / per-sequence launches,
    and without the switch no plan takes it"""
    from dr4sr_amd.engine import SasrecEngine
    ids = torch.ones(512, 50, dtype=torch.int64, device="cuda")
    eng = SasrecEngine(500, 50, 64, 2, 128, 2, 1e-12, 0.0, 512, "cuda")
    short = torch.full((512,), 5, dtype=torch.int64, device="cuda")
    full = torch.full((512,), 50, dtype=torch.int64, device="cuda")
    assert _bits(eng.make_plan(ids, ids, short)) & 8
    assert not _bits(eng.make_plan(ids, ids, full)) & 8
    eng128 = SasrecEngine(500, 50, 128, 2, 128, 2, 1e-12, 0.0, 512, "cuda")
    assert not _bits(eng128.make_plan(ids, ids, short)) & 8
    monkeypatch.delenv("DR4SR_ATTN_WINDOW")
    assert _bits(eng.make_plan(ids, ids, short)) == 3 | 16           # lists regime, taken by the wave-per-tile form (round 6)
