"""random small batches (odd sizes, random lengths incl. 1 and L, random PAD targets) through the fused step vs the oracle"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sasrec_oracle as O
from test_gpu_parity import _random_params, relerr
from dr4sr_amd.engine import SasrecEngine
dev = "cuda"


def main(trials=12, seed=0):
    """the trials of this file as a function: tests call it in-process (the launch-form switches are re-readable through
    dr4sr_reload_env), `python tests/fuzz_parity.py` runs it stand-alone (TRIALS, SEED)"""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for trial in range(trials):
        B = int(rng.choice([1, 2, 3, 5, 17, 31, 64, 100]))
        N = int(rng.choice([2, 3, 50, 300]))
        D = int(rng.choice([64, 128]))
        L = int(rng.choice([8, 50, 64]))
        sl = rng.integers(1, L + 1, size=B)
        sl[rng.integers(0, B)] = 1
        inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            inp[b, :sl[b]] = rng.integers(1, N, size=sl[b])
            tgt[b, :sl[b]] = rng.integers(0, N, size=sl[b])            # some targets are PAD
        neg = rng.integers(1, N, size=(B, L, 1))
        b_ = {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl.astype(np.int64)),
              "neg_item": torch.from_numpy(neg)}
        params = _random_params(N, D, 128, 2, L=L, seed=trial)
        eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, 0.0, B, dev)
        eng.load_named(params)
        plan = eng.make_plan(b_["in_item_id"].to(dev), b_["item_id"].to(dev), b_["seqlen"].to(dev),
                             neg_item=b_["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
        eng.fwd_bwd(plan)
        loss, n = eng.loss_and_count()
        nv = int((b_["item_id"] != 0).sum())
        if nv == 0:
            assert n == 0
            continue
        loss_o, _, grads_o = O.grads_of(params, b_, 2, 2, 1e-12)
        assert n == nv, (n, nv)
        e = abs(loss - float(loss_o))
        g = max(relerr(v, grads_o[k]) for k, v in eng.normalized_grads().items())
        worst = max(worst, g)
        print("trial %2d L=%2d B=%3d N=%3d D=%3d T=%4d n_valid=%4d  |dloss| %.1e  max grad relerr %.1e" % (trial, L, B, N, D, int(sl.sum()), nv, e, g))
        assert e < 3e-5 and g < 5e-4
    print("FUZZ ok, worst grad relerr %.2e" % worst)
    return worst


if __name__ == "__main__":
    main(int(os.environ.get("TRIALS", "12")), int(os.environ.get("SEED", "0")))
