"""K1 microbench only: dr4sr_embed_gather_posadd on 16.7M Zipf-distributed tokens (the bench's roofline_gather workload)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dr4sr_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
N, L, D = 11925, 50, 64
ntok = 16777200
Bg = ntok // L
g = torch.Generator().manual_seed(0)
w = 1.0 / torch.arange(1, N, dtype=torch.float64) ** 0.8
idx = (torch.multinomial(w, Bg * L, replacement=True, generator=g) + 1).view(Bg, L).to(dev)
E = torch.randn(N, D, device=dev); P = torch.randn(L, D, device=dev)
out = torch.empty(Bg, L, D, device=dev)
for _ in range(3):
    lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx), _lib.ptr(out), Bg, L, D, N, _lib.cur_stream())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx), _lib.ptr(out), Bg, L, D, N, _lib.cur_stream())
b.record(); b.synchronize()
us = a.elapsed_time(b) * 100
ref = E[idx[:4]] + P
print("us/launch %.1f  algorithmic %.0f GB/s (%.1f%% of 8 TB/s)  exact %s" % (us, ntok * 520 / us / 1e3, ntok * 520 / us / 1e3 / 80, bool((out[:4] == ref).all())))
