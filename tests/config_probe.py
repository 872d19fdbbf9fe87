"""every encoder shape check_plan (csrc/step.hip) accepts — heads, FFN width, depth, sequence length — through the fused step vs the oracle;
shapes it rejects must raise Dr4srError, never compute"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sasrec_oracle as O
from test_gpu_parity import _random_params, relerr
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd import _lib
dev = "cuda"
rng = np.random.default_rng(0)
CASES = [(64, 1, 128, 2, 50), (64, 2, 256, 2, 50), (64, 1, 256, 1, 50), (128, 4, 128, 2, 50), (128, 2, 128, 2, 50), (64, 2, 128, 1, 50),
         (64, 2, 128, 3, 50), (64, 2, 128, 4, 20), (128, 4, 128, 3, 64), (64, 2, 128, 2, 1), (64, 2, 128, 2, 3),
         (64, 4, 128, 2, 50), (32, 2, 128, 2, 50), (64, 2, 64, 2, 50), (128, 2, 256, 2, 50), (64, 2, 128, 2, 100), (128, 1, 128, 2, 50)]
for (D, H, F, NL, L) in CASES:
    B, N = 37, 211
    sl = rng.integers(1, L + 1, size=B); sl[0] = 1; sl[1] = L
    inp = np.zeros((B, L), dtype=np.int64); tgt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        inp[b, :sl[b]] = rng.integers(1, N, size=sl[b]); tgt[b, :sl[b]] = rng.integers(0, N, size=sl[b])
    neg = rng.integers(1, N, size=(B, L, 1))
    b_ = {"in_item_id": torch.from_numpy(inp), "item_id": torch.from_numpy(tgt), "seqlen": torch.from_numpy(sl.astype(np.int64)),
          "neg_item": torch.from_numpy(neg)}
    tag = "D=%3d H=%d F=%3d NL=%d L=%3d" % (D, H, F, NL, L)
    try:
        eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, 0.0, B, dev)
    except _lib.Dr4srError as e:
        print(tag, " rejected:", str(e)[:90]); continue
    params = _random_params(N, D, F, NL, L=L, seed=7)
    eng.load_named(params)
    plan = eng.make_plan(b_["in_item_id"].to(dev), b_["item_id"].to(dev), b_["seqlen"].to(dev),
                         neg_item=b_["neg_item"].squeeze(-1).contiguous().to(dev), sample_neg=False)
    try:
        eng.fwd_bwd(plan)
    except _lib.Dr4srError as e:
        print(tag, " FAILED AT LAUNCH:", str(e)[:90]); continue
    loss, n = eng.loss_and_count()
    loss_o, _, grads_o = O.grads_of(params, b_, H, NL, 1e-12)
    e = abs(loss - float(loss_o))
    g = max(relerr(v, grads_o[k]) for k, v in eng.normalized_grads().items())
    print(tag, " |dloss| %.1e  max grad relerr %.1e  %s" % (e, g, "ok" if (e < 3e-5 and g < 5e-4) else "MISMATCH"))
