"""Data-parallel check on ONE GPU: W ranks (gloo transport of dr4sr_amd/parallel.py, all on cuda:0) each take a contiguous slice of
every global batch, sum-all-reduce the flat un-normalised gradient (+ {n_valid, loss} tail) through the PRODUCT's reduction path
(parallel.grad_buckets / dp_backward: one flat bucket in the latency launch forms, table bucket | encoder bucket at scale) and run the
same dense Adam step; rank 0 also trains a single-rank engine on the full global batches.  Prints the max parameter difference after
the steps (fp32 summation order only).  The row count leaves a partial tail batch whose slices are uneven and, at 8 ranks, EMPTY
(an empty rank contributes zeros and still enters the collective).
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_check.py
  DP_D=128 DP_B=256 DP_U=980 python -m torch.distributed.run --nproc-per-node 8 ... tools/dp_check.py      (32 rows per rank; tail 212)
  DP_B=4096 DP_U=9000 DP_STEPS=3 ...                                                                        (at scale: two buckets)"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DR4SR_DP_BACKEND", "gloo")
from dr4sr_amd import parallel
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.parallel import shard_bounds
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
D = int(os.environ.get("DP_D", "64"))
U, B, L = int(os.environ.get("DP_U", "512")), int(os.environ.get("DP_B", "128")), 50
N = 20034 if D == 128 else TOYS_N_ITEMS
steps = int(os.environ.get("DP_STEPS", str((U + B - 1) // B)))          # default: one epoch, the partial tail batch included
rows = make_rows(n_rows=U, n_items=N, seed=21)
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
negs = torch.randint(1, N, (U, L), generator=torch.Generator().manual_seed(4)).to(dev)
perm = torch.from_numpy(np.random.default_rng(9).permutation(U)).to(dev)


def make(bmax):
    eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, 0.0, bmax, dev, seed=5, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    for k, v in eng.views.items():
        v.copy_(torch.ones(v.shape) if "norm" in k and k.endswith("weight") else 0.05 * torch.randn(v.shape, generator=g))
    eng.views["item_embedding.weight"][0] = 0
    return eng


info = {}


def train(eng, w, r):
    per = (B + w - 1) // w
    rb = torch.zeros(per, dtype=torch.int64, device=dev)
    nb = torch.zeros(per, L, dtype=torch.int64, device=dev)
    for i in range(steps):
        lo, hi = shard_bounds(i, B, U, w, r)
        bl = hi - lo
        full = (i + 1) * B <= U
        # the bucket count is a function of GLOBAL quantities (a full slice's rows), never of this rank's own slice
        buckets = parallel.grad_buckets(eng, per if (w > 1 and full) else None, data["seqlen"])
        if w > 1:
            info.setdefault("buckets", set()).add(len(buckets))
            info["empty"] = info.get("empty", 0) + (bl == 0)
        if bl == 0:                                           # fewer rows than ranks: contribute zeros to the same collectives
            parallel.dp_reduce_empty(eng, buckets)
            eng.adam_step(plan_any)
            continue
        rb[:bl].copy_(perm[lo:hi])
        nb[:bl].copy_(negs[perm[lo:hi]])
        plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rb[:bl], neg_item=nb[:bl].contiguous().view(-1),
                             sample_neg=False)
        plan_any = plan
        if w > 1:
            parallel.dp_backward(eng, plan, False, buckets)
        else:
            eng.fwd_bwd(plan)
        eng.adam_step(plan)
    torch.cuda.synchronize()
    return eng.params.clone()


p_dp = train(make((B + world - 1) // world), world, rank)
if rank == 0:
    p_one = train(make(B), 1, 0)
    d = float((p_dp - p_one).abs().max())
    print("DP_CHECK world=%d d=%d B=%d U=%d steps=%d buckets=%s max|dp - single| = %.3e  (max|param| %.3f)" %
          (world, D, B, U, steps, sorted(info.get("buckets", [1])), d, float(p_one.abs().max())))
    assert d < 2e-4, d
# replicas identical?  (and did any rank see an empty slice)
chk = torch.tensor([float(p_dp.double().sum()), float(info.get("empty", 0))], dtype=torch.float64)
lst = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(lst, chk)
if rank == 0:
    print("DP_CHECK replica checksums equal:", all(float(x[0]) == float(lst[0][0]) for x in lst), "; empty slices seen:", int(sum(float(x[1]) for x in lst)))
dist.destroy_process_group()
