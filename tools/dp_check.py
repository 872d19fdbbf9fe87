"""Data-parallel check on ONE GPU: W ranks (gloo, all on cuda:0) each take a contiguous slice of every global batch, all-reduce the
flat un-normalised gradient (+ {n_valid, loss} tail) and run the same dense Adam step; rank 0 also trains a single-rank engine on the
full global batches.  Prints the max parameter difference after the steps (fp32 atomics order only).
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.parallel import allreduce_flat, shard_bounds
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
U, B, L, N, steps = 512, 128, 50, TOYS_N_ITEMS, 4
rows = make_rows(n_rows=U, n_items=N, seed=21)
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
negs = torch.randint(1, N, (U, L), generator=torch.Generator().manual_seed(4)).to(dev)
perm = torch.from_numpy(np.random.default_rng(9).permutation(U)).to(dev)


def make(bmax):
    eng = SasrecEngine(N, L, 64, 2, 128, 2, 1e-12, 0.0, bmax, dev, seed=5, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    for k, v in eng.views.items():
        v.copy_(torch.ones(v.shape) if "norm" in k and k.endswith("weight") else 0.05 * torch.randn(v.shape, generator=g))
    eng.views["item_embedding.weight"][0] = 0
    return eng


def train(eng, w, r):
    per = (B + w - 1) // w
    rb = torch.zeros(per, dtype=torch.int64, device=dev)
    nb = torch.zeros(per, L, dtype=torch.int64, device=dev)
    for i in range(steps):
        lo, hi = shard_bounds(i, B, U, w, r)
        bl = hi - lo
        rb[:bl].copy_(perm[lo:hi])
        nb[:bl].copy_(negs[perm[lo:hi]])
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rb[:bl], neg_item=nb[:bl].contiguous().view(-1),
                             sample_neg=False)
        eng.fwd_bwd(plan)
        if w > 1:
            g = eng.grads.cpu()                       # gloo: reduce on the host
            dist.all_reduce(g)
            eng.grads.copy_(g)
        eng.adam_step(plan)
    torch.cuda.synchronize()
    return eng.params.clone()


p_dp = train(make(B), world, rank)
if rank == 0:
    p_one = train(make(B), 1, 0)
    d = float((p_dp - p_one).abs().max())
    print("DP_CHECK world=%d max|dp - single| = %.3e  (max|param| %.3f)" % (world, d, float(p_one.abs().max())))
    assert d < 2e-4, d
# replicas identical?
chk = torch.tensor([float(p_dp.double().sum())], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0:
    print("DP_CHECK replica checksums equal:", all(float(x) == float(lst[0]) for x in lst))
dist.destroy_process_group()
