"""RCCL all-reduce captured INSIDE a k-step HIP graph, replayed many times — the only RCCL-in-graph evidence a
1-GPU box can give: one rank (world_size 1, the library's own RCCL communicator — dr4sr_comm_*, include/dr4sr_hip.h ABI 8), k whole data-parallel steps per graph in the form
BaseModel._step_graph / bench.py capture (fwd_bwd[_prepared] -> all-reduce of the flat gradient + tail -> adam_step[_prepare_next]),
REPLAYS x k steps, per-step loss log and final parameters against the un-captured single-GPU form (dr4sr_sasrec_train_step in a
loop, no collective, no graph).  Same seeds, dropout and in-kernel negatives on both sides.
  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29549 tools/dp_graph_check.py"""
import os, sys, faulthandler
faulthandler.dump_traceback_later(300, exit=True)
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["DR4SR_BENCH_FORCE_DP"] = "1"                     # parallel.init_distributed: build the group although world_size == 1
from dr4sr_amd import parallel
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
from dr4sr_amd.utils.graphs import capture

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_distributed(dev)
assert parallel.can_capture() and parallel.backend_name() == "rccl"
# DP_GRAPH_B=8192: the at-scale launch forms -> TWO gradient buckets (parallel.dp_backward: the table bucket's all-reduce is a parallel
# branch of the captured graph beside the last weight-gradient launch), optimizer launches with the two-phase prep
U, B, L, N, K, REPLAYS = 19412, int(os.environ.get("DP_GRAPH_B", "256")), 50, TOYS_N_ITEMS, int(os.environ.get("DP_GRAPH_K", "4")), int(os.environ.get("DP_GRAPH_REPLAYS", "30"))
rows = make_rows(n_rows=U, n_items=N, seed=21)
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
perm = torch.from_numpy(np.random.default_rng(9).permutation(U)).to(dev)
steps = K * REPLAYS


def make():
    eng = SasrecEngine(N, L, 64, 2, 128, 2, 1e-12, 0.5, B, dev, seed=5, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    for k, v in eng.views.items():
        v.copy_(torch.ones(v.shape) if "norm" in k and k.endswith("weight") else 0.05 * torch.randn(v.shape, generator=g))
    eng.views["item_embedding.weight"][0] = 0
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    log = torch.zeros(steps, dtype=torch.float32, device=dev)
    plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=torch.zeros(B, dtype=torch.int64, device=dev),
                         neg_item=torch.zeros(B, L, dtype=torch.int64, device=dev), sample_neg=True, perm_sel=(perm, B, 0, counter),
                         loss_log=log)
    return eng, plan, counter, log


# ---- un-captured reference: single-GPU steps, no collective
eng0, plan0, c0, log0 = make()
for _ in range(steps):
    eng0.train_step(plan0)
torch.cuda.synchronize()
# ---- captured: k DP steps + their k RCCL all-reduces in ONE graph
eng1, plan1, c1, log1 = make()
buckets = parallel.grad_buckets(eng1, B, data["seqlen"])          # (two buckets are opt-in: DR4SR_DP_BUCKETS=2)
if os.environ.get("DP_GRAPH_EXPECT_BUCKETS"):
    assert len(buckets) == int(os.environ["DP_GRAPH_EXPECT_BUCKETS"]), buckets
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for j in range(2):                                        # warm-up outside the capture (code objects, LDS attributes), then undo it
        snap = [t.clone() for t in (eng1.params, eng1.adam_m, eng1.adam_v, eng1.state, c1)]
        parallel.dp_backward(eng1, plan1, False, buckets)
        eng1.adam_step(plan1)
        stream.synchronize()
        for dst, src in zip((eng1.params, eng1.adam_m, eng1.adam_v, eng1.state, c1), snap):
            dst.copy_(src)
    g = torch.cuda.CUDAGraph()
    with capture(g, stream=stream):
        for j in range(K):
            parallel.dp_backward(eng1, plan1, j > 0, buckets)
            (eng1.adam_step_prepare_next if j < K - 1 else eng1.adam_step)(plan1)
    for _ in range(REPLAYS):
        g.replay()
    stream.synchronize()
torch.cuda.synchronize()
assert int(c0) == steps and int(c1) == steps and int(eng1.state[0]) == steps, (int(c0), int(c1), int(eng1.state[0]))
d_first = float((log0[:10] - log1[:10]).abs().max())
d_all = float(((log0 - log1).abs() / log0.abs()).max())
dp = float((eng0.params - eng1.params).abs().max())
print("DP_GRAPH rccl in-graph all-reduce (B = %d, %d bucket(s)): %d steps = %d replays x %d steps/graph; loss log max abs diff first 10 steps %.2e, max rel diff all %.2e, "
      "final params max abs diff %.2e (|param| max %.3f), loss %.4f -> %.4f" % (B, len(buckets), steps, REPLAYS, K, d_first, d_all, dp, float(eng0.params.abs().max()),
                                                                                 float(log1[0]), float(log1[-1])))
assert d_first < 2e-5 and d_all < 5e-3 and dp < 5e-3 and float(log1[-1]) < float(log1[0])
print("DP_GRAPH_OK", flush=True)
parallel.shutdown()
