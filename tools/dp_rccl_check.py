"""Data-parallel check with REAL RCCL ranks, one device per rank (a multi-GPU node, or the logical devices of one MI355X in DPX / QPX / CPX
compute-partition mode — tools/partition_rccl_check.sh): W ranks each take a contiguous slice of every global batch, sum-all-reduce the
flat un-normalised gradient (+ {n_valid, loss} tail) over RCCL and run the same dense Adam step,
  form "host":     the collective launched by the host between backward and optimizer (BaseModel's default),
  form "in_graph": k whole steps (fwd_bwd -> all-reduce -> adam) captured in ONE HIP graph and replayed (train.dp_graph_allreduce);
rank 0 also trains a single-rank engine on the full global batches: max parameter difference (fp32 summation order only), replicas
bit-identical.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_rccl_check.py"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(240, exit=True)
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from dr4sr_amd import parallel
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.parallel import allreduce_flat, shard_bounds
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
from dr4sr_amd.utils.graphs import capture

rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
assert torch.cuda.device_count() >= world, "one device per rank: %d devices for %d ranks" % (torch.cuda.device_count(), world)
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
parallel.init_distributed(dev)
assert parallel.can_capture() and dist.get_world_size() == world      # the library's own RCCL communicator; the gloo group is the control plane
U, B, L, N, K, REPLAYS = 4096, 256 * world, 50, TOYS_N_ITEMS, 4, 3
steps = K * REPLAYS
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


def train(eng, w, r, form):
    per = (B + w - 1) // w
    rb = torch.zeros(steps, per, dtype=torch.int64, device=dev)
    nb = torch.zeros(steps, per, L, dtype=torch.int64, device=dev)
    plans = []
    for i in range(steps):
        lo, hi = shard_bounds(i, B, U, w, r)
        assert hi - lo == per
        rb[i].copy_(perm[lo:hi])
        nb[i].copy_(negs[perm[lo:hi]])
        plans.append(eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rb[i], neg_item=nb[i].view(-1), sample_neg=False))
    if form == "in_graph":
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            for rep in range(REPLAYS):                   # one graph per group of K steps (each step has its own batch tensors)
                g = torch.cuda.CUDAGraph()
                with capture(g, stream=stream):
                    for j in range(K):
                        p = plans[rep * K + j]
                        eng.fwd_bwd(p)
                        allreduce_flat(eng.grads)
                        eng.adam_step(p)
                g.replay()
            stream.synchronize()
    else:
        for p in plans:
            eng.fwd_bwd(p)
            if w > 1:
                allreduce_flat(eng.grads)
            eng.adam_step(p)
    torch.cuda.synchronize()
    return eng.params.clone()


for form in ("host", "in_graph"):
    p_dp = train(make(B // world), world, rank, form)
    chk = torch.tensor([float(p_dp.double().sum())], dtype=torch.float64)      # (control plane: host tensors)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    if rank == 0:
        p_one = train(make(B), 1, 0, "host")
        d = float((p_dp - p_one).abs().max())
        same = all(float(x) == float(lst[0]) for x in lst)
        print("DP_RCCL world=%d form=%s: %d steps, max|dp - single| = %.3e (max|param| %.3f), replicas identical: %s"
              % (world, form, steps, d, float(p_one.abs().max()), same), flush=True)
        assert d < 2e-4 and same, (d, same)
    dist.barrier()
if rank == 0:
    print("DP_RCCL_OK", flush=True)
parallel.shutdown()
