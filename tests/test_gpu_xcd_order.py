"""XCD-aware block -> tile order of the latency regime's token-tile kernels and of the weight-gradient launch behind them (round 4,
csrc/kernels.h xcd_tile / xcd_per, csrc/linear.hip wgrad_body): XCD x owns a CONTIGUOUS range of token tiles (whole 64-token tiles), so
a tile's attention window — the rows of the tiles in front of it — was written on the same XCD.  A placement choice, never a correctness
one: every token count (tile counts of every residue mod 8, ranges that are not multiples of four tiles, a last XCD with no tile at all,
one tile in total) must give the gradients of the plain order (DR4SR_TILE_ORDER_PLAIN / DR4SR_WGRAD_ORDER_PLAIN) and of the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sasrec_oracle as O  # noqa: E402
from test_gpu_parity import _random_params, relerr  # noqa: E402


def _batch(lengths, N, L, rng):
    sl = np.asarray(lengths, dtype=np.int64)
    B = len(sl)
    inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b]); tgt[b, :sl[b]] = rng.integers(1, N, size=sl[b])
    return {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl),
            "neg_item": torch.from_numpy(rng.integers(1, N, size=(B, L, 1)))}


# token counts -> 16-token tiles: 1, 8, 9 (two XCD ranges of 4 tiles + 1), 33 (ranges of 8: the last four XCDs idle), 93 (the toys batch),
# 100 (ranges of 16: XCD 6 holds 4 tiles, XCD 7 none), 129
@pytest.mark.parametrize("D,tokens", [(64, 7), (64, 128), (64, 131), (64, 520), (64, 1480), (64, 1599), (128, 2050), (128, 131)])
def test_xcd_tile_order_equals_plain_order_and_oracle(D, tokens, monkeypatch):
    from dr4sr_amd.engine import SasrecEngine
    rng = np.random.default_rng(tokens + D)
    N, L, H, F, NL = 211, 50, 2, 128, 2
    lengths, left = [], tokens
    while left > 0:
        n = int(min(left, rng.integers(1, 24)))
        lengths.append(n); left -= n
    batch = _batch(lengths, N, L, rng)
    B = len(lengths)
    params = _random_params(N, D, F, NL, L=L, seed=3)
    eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, 0.0, B, "cuda")
    eng.load_named(params)
    plan = eng.make_plan(batch["in_item_id"].cuda(), batch["item_id"].cuda(), batch["seqlen"].cuda(),
                         neg_item=batch["neg_item"].squeeze(-1).contiguous().cuda(), sample_neg=False)
    eng.fwd_bwd(plan)
    loss, n = eng.loss_and_count()
    g = {k: v.clone() for k, v in eng.normalized_grads().items()}
    loss_o, _, grads_o = O.grads_of(params, batch, H, NL, 1e-12)
    assert n == tokens and abs(loss - float(loss_o)) < 3e-5
    for k, v in g.items():
        assert relerr(v, grads_o[k]) < 5e-4, k
    for switch in ("DR4SR_WGRAD_ORDER_PLAIN", "DR4SR_TILE_ORDER_PLAIN"):
        monkeypatch.setenv(switch, "1")
        eng.fwd_bwd(plan)
        loss_p, _ = eng.loss_and_count()
        assert abs(loss_p - loss) < 1e-6, switch
        for k, v in eng.normalized_grads().items():
            assert relerr(v, g[k].cpu()) < 2e-5, (switch, k)
