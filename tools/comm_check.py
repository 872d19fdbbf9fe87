"""The library's own RCCL transport (include/dr4sr_hip.h ABI 8, csrc/comm.hip) driven straight through ctypes with the ONE rank a 1-GPU box
can host — no torch.distributed anywhere in this process:
  * dr4sr_comm_unique_id / dr4sr_comm_init_rank / dr4sr_comm_destroy, rank / world queries, argument and RCCL error codes;
  * dr4sr_allreduce_f32 / _f64, dr4sr_allgather_bytes, dr4sr_broadcast_bytes on torch's current stream: at one rank every collective is
    the identity, bit for bit;
  * the asynchronous form (dr4sr_allreduce_f32_async + dr4sr_comm_join) captured INSIDE a HIP graph as a parallel branch between kernels
    that write and read the reduced buffer, replayed COMM_REPLAYS times: ordering holds on every replay (the value chain below breaks if the
    collective's branch is not joined), no thread of ours exists, the process exits cleanly afterwards.
  python tools/comm_check.py"""
import ctypes as C
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from dr4sr_amd import _lib                      # noqa: E402
from dr4sr_amd.utils.graphs import capture      # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
threads_before = threading.active_count()

ident = C.create_string_buffer(_lib.COMM_ID_BYTES)
assert lib.dr4sr_comm_unique_id(ident) == 0
comm = C.c_void_p()
assert lib.dr4sr_comm_init_rank(ident, 1, 1, 0, C.byref(comm)) == -1 and not comm.value          # rank >= world
assert lib.dr4sr_comm_init_rank(ident, 0, 1, 0, C.byref(comm)) == 0 and comm.value
assert lib.dr4sr_comm_rank(comm) == 0 and lib.dr4sr_comm_world(comm) == 1 and lib.dr4sr_comm_async_error(comm) == 0
st = _lib.cur_stream

g = torch.Generator().manual_seed(3)
x = torch.randn(833348, generator=g).to(dev)                     # a toys-sized flat gradient + tail
ref = x.clone()
assert lib.dr4sr_allreduce_f32(comm, _lib.ptr(x), x.numel(), st()) == 0
assert lib.dr4sr_allreduce_f32(comm, _lib.ptr(x), 0, st()) == 0 and lib.dr4sr_allreduce_f32(comm, None, 4, st()) == -1
d = torch.randn(5, generator=g, dtype=torch.float64).to(dev)
dref = d.clone()
for op in (0, 1, 2):
    assert lib.dr4sr_allreduce_f64(comm, _lib.ptr(d), d.numel(), op, st()) == 0
assert lib.dr4sr_allreduce_f64(comm, _lib.ptr(d), d.numel(), 3, st()) == -1
gat = torch.empty(1, 1000, dtype=torch.int64, device=dev)
src = torch.arange(1000, dtype=torch.int64, device=dev)
assert lib.dr4sr_allgather_bytes(comm, _lib.ptr(src), _lib.ptr(gat), src.numel() * 8, st()) == 0
assert lib.dr4sr_broadcast_bytes(comm, _lib.ptr(src), src.numel() * 8, 0, st()) == 0
assert lib.dr4sr_broadcast_bytes(comm, _lib.ptr(src), 8, 1, st()) == -1                           # root outside the communicator
torch.cuda.synchronize()
assert torch.equal(x, ref) and torch.equal(d, dref) and torch.equal(gat[0], src) and torch.equal(src, torch.arange(1000, device=dev))

# ---- asynchronous collective as a parallel branch of a captured graph
REPLAYS, K = int(os.environ.get("COMM_REPLAYS", "200")), 4
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)
acc = torch.zeros(1 << 20, device=dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    graph = torch.cuda.CUDAGraph()
    with capture(graph, stream=stream):
        for _ in range(K):
            a.add_(1.0)                                                                  # "phase 1": bucket 0 final
            assert lib.dr4sr_allreduce_f32_async(comm, _lib.ptr(a), a.numel(), st()) == 0
            b.add_(2.0)                                                                  # "phase 2" beside the collective
            assert lib.dr4sr_allreduce_f32_async(comm, _lib.ptr(b), b.numel(), st()) == 0
            assert lib.dr4sr_comm_join(comm, st()) == 0
            acc.add_(a).add_(b)                                                          # "optimizer": reads both buckets
            assert lib.dr4sr_allreduce_f32(comm, _lib.ptr(acc), acc.numel(), st()) == 0   # the blocking form inside the same capture
    for _ in range(REPLAYS):
        graph.replay()
    stream.synchronize()
n = REPLAYS * K
want = 3.0 * n * (n + 1) / 2                                      # sum_{i=1..n} (i + 2 i)
assert float(a[0]) == n and float(b[-1]) == 2 * n and float(acc[0]) == want and float(acc[-1]) == want, (float(a[0]), float(acc[0]), want)
assert bool((acc == want).all())
assert lib.dr4sr_comm_async_error(comm) == 0
threads_after = threading.active_count()
del graph
assert lib.dr4sr_comm_destroy(comm) == 0 and lib.dr4sr_comm_destroy(None) == 0
print("COMM_CHECK one-rank RCCL communicator through the C ABI: collectives identity, %d replays of a %d-step graph with 2 async + 1 blocking "
      "all-reduce per step ordered correctly (acc = %.0f); python threads %d -> %d; torch.distributed initialised: %s"
      % (REPLAYS, K, want, threads_before, threads_after, torch.distributed.is_initialized()), flush=True)
assert not torch.distributed.is_initialized()
print("COMM_CHECK_OK", flush=True)
