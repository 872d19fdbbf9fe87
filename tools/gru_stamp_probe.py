"""Per-phase shader-clock stamps of one middle time step of the two-layer GRU wavefront kernels (csrc/gru_coop.hip, WAVE_STAMP).
   python tools/gru_stamp_probe.py [fwd|bwd]   (the stamps of the LAST stamped launch survive; `fwd` runs the forward hook alone after
   three steps; `bwd` needs DR4SR_GRU_WAVE_BWD=1: only the opt-in one-launch backward, k_gru_bwd_pair, carries stamps)"""
import os, sys
os.environ["DR4SR_GRU_WAVE_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from dr4sr_amd import _lib
from dr4sr_amd.gru_engine import GruEngine, gru_param_names, gru_param_shapes
from dr4sr_amd.data.synthetic import make_rows

which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
B, N, L, H, NL = 256, 12102, 50, 256, 2
rows = make_rows(n_rows=B, n_items=N, seed=3, dense=False)
b = {k: torch.from_numpy(rows[k]) for k in ("in_item_id", "item_id", "seqlen")}
gen = torch.Generator().manual_seed(4)
neg = torch.randint(1, N, (B, L), generator=gen)
params = {n: 0.08 * torch.randn(s, generator=gen) for n, s in zip(gru_param_names(NL), gru_param_shapes(N, 64, H, NL))}
eng = GruEngine(N, L, 64, H, NL, 0.2, B, "cuda", seed=5)
eng.load_named(params)
dev = eng.device
plan = eng.make_plan(b["in_item_id"].to(dev), b["item_id"].to(dev), b["seqlen"].to(dev), neg_item=neg.to(dev), sample_neg=False)
for _ in range(3):
    eng.fwd_bwd(plan)
if which == "fwd":
    _lib.check(eng.lib.dr4sr_gru4rec_launch_kernel(C.byref(plan), 3, 0, _lib.cur_stream()), "hook")
torch.cuda.synchronize()
w = eng.workspace[:256].view(torch.int32).cpu().tolist()
for role, name in ((0, "leader"), (1, "follower")):
    st = [x & 0xffffffff for x in w[8 + 16 * role: 8 + 16 * role + 8]]
    d = [(st[i + 1] - st[i]) & 0xffffffff for i in range(7)]
    print(which, name, "stamps deltas (cycles @100MHz? shader clk):", d)

if which == "fwd":
    ids = w[40:64]
    print("block -> xcc id:", [x & 0xf for x in ids])
    print("block -> hw_id (cu bits 8..11, sh 12, se 13..15):", [((x >> 4) >> 8) & 0xff for x in ids])
