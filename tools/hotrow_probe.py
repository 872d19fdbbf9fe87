import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
import bench
dev = torch.device("cuda", 0)
B, L, D, N = 256, 50, 64, TOYS_N_ITEMS
rows = make_rows(n_items=N, seed=2024)
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
eng = SasrecEngine(N, L, D, 2, 128, 2, 1e-12, 0.5, B, dev, seed=2023)
bench.init_params_like_reference(eng, 2023)
rb = torch.arange(B, device=dev)
neg = torch.zeros(B, L, dtype=torch.int64, device=dev)
def run(tag, ids):
    plan = eng.make_plan(ids, data["item_id"][:B], data["seqlen"][:B], rows=rb, neg_item=neg, sample_neg=True)
    for _ in range(5): eng.fwd_bwd(plan)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): eng.fwd_bwd(plan)
    torch.cuda.synchronize(); print(tag, "%.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
ids = data["in_item_id"][:B].clone()
run("plain", ids)
m = ids.clone(); sel = (torch.rand_like(m, dtype=torch.float32) < 0.7) & (m > 0); m[sel] = N
run("mask->N", m)
m2 = ids.clone(); m2[sel] = 5
run("mask->5", m2)
m3 = ids.clone(); m3[sel] = 0
run("mask->0", m3)
