"""phase ticks (s_memtime, 100 MHz) of block 0 of k_post_fwd / k_post_bwd with the attention inside (csrc/attn_tile.h); DR4SR_STAMPS"""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dr4sr_amd import _lib
from dr4sr_amd.engine import SasrecEngine
from dr4sr_amd.data.synthetic import make_rows, TOYS_N_ITEMS
import bench
lib = _lib.load()
dev = torch.device("cuda", 0)
B, L, D, H, F, NL, N = 256, 50, int(os.environ.get("EMBED_DIM", "64")), 2, 128, 2, TOYS_N_ITEMS
rows = make_rows(n_items=N, seed=2024, dense=bool(int(os.environ.get("DENSE", "0"))))
data = {k: torch.from_numpy(rows[k]).to(dev) for k in ("in_item_id", "item_id", "seqlen")}
eng = SasrecEngine(N, L, D, H, F, NL, 1e-12, 0.5, B, dev, seed=2023)
bench.init_params_like_reference(eng, 2023)
rb = torch.arange(B, device=dev)
neg = torch.zeros(B, L, dtype=torch.int64, device=dev)
plan = eng.make_plan(data["in_item_id"], data["item_id"], data["seqlen"], rows=rb, neg_item=neg, sample_neg=True)
for _ in range(3):
    eng.train_step(plan)
torch.cuda.synchronize()
Tmax = B * L
r = lambda nfl: (nfl * 4 + 255) // 256 * 256
off = r(B + 1) + r((Tmax + 15) // 16 + 1) + r(4 + 7 * B) + r(4 * B + 4 * 1024) + r(Tmax * H) + 2 * (NL + 1) * r(Tmax * D)      # csrc/step.hip carve_workspace: ... -> dctx
os.environ["DR4SR_STAMPS"] = "1"
lib.dr4sr_reload_env()
for kind, layer in (("post_fwd", 0), ("post_bwd", 0)):
    kid = _lib.KERNEL_IDS[kind]
    for _ in range(3):
        _lib.check(lib.dr4sr_sasrec_launch_kernel(C.byref(plan), kid, layer, _lib.cur_stream()), kind)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        _lib.check(lib.dr4sr_sasrec_launch_kernel(C.byref(plan), kid, layer, _lib.cur_stream()), kind)
    b.record(); b.synchronize()
    st = eng.workspace[off:off + 32 * 8].view(torch.int64).cpu().numpy()
    if kind == "post_fwd":
        idx = [0, 27, 28, 8, 9, 10, 11, 12, 1, 2, 3, 4, 5, 6, 15]
    else:
        idx = [16, 17, 18, 20, 21, 22, 23]
    print(kind, layer, "us/launch %.2f" % (a.elapsed_time(b) * 1e3 / 20), "stamps", idx, "ticks since first", [int(st[i] - st[idx[0]]) for i in idx],
          "phase B (thread 128): start, end", int(st[24] - st[idx[0]]), int(st[25] - st[idx[0]]))
# ... and the same stamps as the REAL step leaves them (k_post_bwd of layer 0 is the last writer of 16..25)
for _ in range(3):
    eng.train_step(plan)
torch.cuda.synchronize()
st = eng.workspace[off:off + 32 * 8].view(torch.int64).cpu().numpy()
idx = [16, 17, 18, 20, 21, 22, 26]
print("real step, k_post_bwd layer 0: stamps", idx, [int(st[i] - st[16]) for i in idx], "phase B start, end", int(st[24] - st[16]), int(st[25] - st[16]))
