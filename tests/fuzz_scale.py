"""random LARGE batches (B = 330 .. 3000) through the fused step vs the oracle: random batch size, catalog size (down to 2 items: every
token hits the same table rows), length mix (incl. all-tiny, all-long, lengths at every class boundary), PAD targets, PAD ids inside
sequences.  With DR4SR_FORCE_SCALE=1 (how tests/test_gpu_r2_paths.py runs it) every trial takes the at-scale forms — 32-row tiles,
length-class attention launches with the VALU class for 1..8 tokens, owner-computed table gradient; without it the regime follows the
batch's expected tokens, so the same trials also drive the latency forms at batch sizes far above the reference's 256."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sasrec_oracle as O
from test_gpu_parity import _random_params, relerr
from dr4sr_amd.engine import SasrecEngine
dev = "cuda"


def main(trials=8, seed=0):
    """the trials of this file as a function: tests call it in-process (the launch-form switches are re-readable through
    dr4sr_reload_env), `python tests/fuzz_scale.py` runs it stand-alone (TRIALS, SEED)"""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for trial in range(trials):
        B = int(rng.choice([330, 400, 777, 1500, 3000]))
        N = int(rng.choice([2, 40, 3000, 11925]))
        D = int(rng.choice([64, 128]))
        NL = int(rng.choice([1, 2, 2, 3]))                               # 1: only the fused last layer; 3: a middle layer with both boundary fusions
        L = 50
        mix = int(rng.integers(0, 4))
        if mix == 0: sl = rng.integers(1, 9, size=B)                       # all tiny
        elif mix == 1: sl = rng.integers(17, L + 1, size=B)               # all long
        elif mix == 2: sl = rng.choice([1, 2, 8, 9, 16, 17, 49, 50], size=B)   # class boundaries
        else: sl = np.minimum(L, rng.geometric(0.18, size=B))             # toys-like
        inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            inp[b, :sl[b]] = rng.integers(1, N, size=sl[b])
            tgt[b, :sl[b]] = rng.integers(0, N, size=sl[b])            # some targets are PAD
        for b in rng.integers(0, B, size=8):                            # PAD ids INSIDE a few sequences (key-padding mask), never at position 0
            if sl[b] > 2: inp[b, int(rng.integers(1, sl[b]))] = 0
        neg = rng.integers(1, N, size=(B, L, 1))
        b_ = {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl.astype(np.int64)),
              "neg_item": torch.from_numpy(neg)}
        params = _random_params(N, D, 128, NL, L=L, seed=trial)
        eng = SasrecEngine(N, L, D, 2, 128, NL, 1e-12, 0.0, B, dev)
        eng.load_named(params)
        plan = eng.make_plan(b_["in_item_id"].to(dev), b_["item_id"].to(dev), b_["seqlen"].to(dev),
                             neg_item=b_["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
        eng.fwd_bwd(plan)
        loss, n = eng.loss_and_count()
        nv = int((b_["item_id"] != 0).sum())
        loss_o, _, grads_o = O.grads_of(params, b_, 2, NL, 1e-12)
        assert n == nv, (n, nv)
        e = abs(loss - float(loss_o))
        g = max(relerr(v, grads_o[k]) for k, v in eng.normalized_grads().items())
        worst = max(worst, g)
        print("trial %2d B=%4d N=%5d D=%3d mix=%d T=%6d n_valid=%6d  |dloss| %.1e  max grad relerr %.1e" % (trial, B, N, D, mix, int(sl.sum()), nv, e, g), "layers", NL)
        assert e < 3e-5 and g < 5e-4
    print("FUZZ-SCALE ok, worst grad relerr %.2e" % worst)
    return worst


if __name__ == "__main__":
    main(int(os.environ.get("TRIALS", "8")), int(os.environ.get("SEED", "0")))
