"""Where the data-parallel form's fixed cost goes, with the ONE RCCL rank a 1-GPU box has: per-GPU batches B, k-step graphs of
  A  dr4sr_sasrec_train_steps (the single-GPU form)
  B  flat body, no collective            C/D  two-phase body (split 1 / 2), no collective
  E  flat body + all-reduce (current stream)
  F/G two-phase + the two ASYNC all-reduces (split 1 / 2: parallel.dp_backward)      H  two-phase + both all-reduces on the current stream
  python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29557 tools/dp_cost_probe.py"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["DR4SR_BENCH_FORCE_DP"] = "1"
from dr4sr_amd import parallel, _lib
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
from dr4sr_amd.utils.graphs import capture

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_distributed(dev)
L, N, K = 50, TOYS_N_ITEMS, int(os.environ.get("PROBE_K", "20"))
rows = make_rows(n_items=N, seed=2024)
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
U = data["seqlen"].shape[0]
perm = torch.from_numpy(np.random.default_rng(7).permutation(U)).to(dev)
lib = _lib.load()


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    lib.dr4sr_reload_env()


def run(B, form, split=None):
    setenv(DR4SR_DP_SPLIT_LAYER=split)
    eng = SasrecEngine(N, L, 64, 2, 128, 2, 1e-12, 0.5, B, dev, seed=2023, lr=1e-3)
    g = torch.Generator().manual_seed(1)
    for k, v in eng.views.items():
        v.copy_(torch.ones(v.shape) if "norm" in k and k.endswith("weight") else 0.02 * torch.randn(v.shape, generator=g))
    eng.views["item_embedding.weight"][0] = 0
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=torch.zeros(B, dtype=torch.int64, device=dev),
                         neg_item=torch.zeros(B, L, dtype=torch.int64, device=dev), sample_neg=True, perm_sel=(perm, B, 0, counter))
    two = [(0, eng.offsets[2]), (eng.offsets[2], eng.n_params + 4)]
    flat = [(0, eng.n_params + 4)]

    def body(n):
        if form == "A":
            eng.train_steps(plan, n)
            return
        for j in range(n):
            prepared = j > 0
            if form == "B":
                parallel.dp_backward(eng, plan, prepared, flat, reduce=False)
            elif form in ("C", "D"):
                parallel.dp_backward(eng, plan, prepared, two, reduce=False)
            elif form == "E":
                parallel.dp_backward(eng, plan, prepared, flat)
            elif form in ("F", "G"):
                parallel.dp_backward(eng, plan, prepared, two)
            elif form == "H":
                eng.fwd_bwd_phase(plan, prepared, 1)
                parallel.allreduce_flat(eng.grads[two[0][0]:two[0][1]])
                eng.fwd_bwd_phase(plan, prepared, 2)
                parallel.allreduce_flat(eng.grads[two[1][0]:two[1][1]])
            (eng.adam_step_prepare_next if j < n - 1 else eng.adam_step)(plan)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        body(2)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g, stream=stream):
            body(K)
        for _ in range(3):
            g.replay()
        stream.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (5 * K))
    return 1e3 * sorted(ts)[len(ts) // 2]


for B in [int(x) for x in os.environ.get("PROBE_B", "4096,16384").split(",")]:
    base = run(B, "A")
    print("DP_COST B=%d  A train_steps %.4f ms" % (B, base), flush=True)
    for form, split, what in (("B", None, "flat, no collective"), ("C", "1", "two-phase split 1, no collective"), ("D", "2", "two-phase split 2 (table-only first launch), no collective"),
                              ("E", None, "flat + all-reduce"), ("F", "1", "two-phase split 1 + async all-reduces"), ("G", "2", "two-phase split 2 + async all-reduces"),
                              ("H", "1", "two-phase split 1 + all-reduces on the current stream")):
        ms = run(B, form, split)
        print("DP_COST B=%d  %s %-62s %.4f ms  (+%.1f us)" % (B, form, what, ms, 1e3 * (ms - base)), flush=True)
parallel.shutdown()
