"""Run the K1 gather microbench alone (for rocprofv3 --pmc passes): python tools/gather_pmc.py [tokens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dr4sr_amd import _lib
from dr4sr_amd.data.synthetic import TOYS_N_ITEMS, make_rows
lib = _lib.load()
ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024
L, D, N = 50, 64, TOYS_N_ITEMS
dev = torch.device("cuda")
rows = make_rows()
idx = torch.from_numpy(rows["in_item_id"]).to(dev)
B = ntok // L
idx = idx.repeat((B + idx.shape[0] - 1) // idx.shape[0], 1)[:B].contiguous()
idx = torch.where(idx == 0, torch.randint(1, N, idx.shape, device=dev), idx)
E = torch.randn(N, D, device=dev); P = torch.randn(L, D, device=dev); out = torch.empty(B, L, D, device=dev)
for _ in range(5):
    lib.dr4sr_embed_gather_posadd(_lib.ptr(E), _lib.ptr(P), _lib.ptr(idx), _lib.ptr(out), B, L, D, N, None)
torch.cuda.synchronize()
print("tokens", B * L)
