"""Single-node data parallelism over users (one process per GPU, torch.distributed 'nccl' = RCCL over xGMI).

The reference has no distributed path (/root/reference utils/callbacks.py:130 is a TODO).  The scheme:
  * every rank holds a full replica (encoder + item table, 3.3 MB fp32 for toys d=64) and the full dataset tensors;
  * one global permutation per epoch (rank 0's, broadcast); the i-th GLOBAL batch of B rows is split into
    contiguous slices of ceil(B/W) rows, rank r takes slice r (`shard_bounds`);
  * each rank accumulates UN-normalised gradients of its slice plus the tail {n_valid, loss_sum, poison}
    (include/dr4sr_hip.h "Flat parameter layout"), then ONE sum-all-reduce of the whole flat buffer.  It is deliberately not
    bucketed: every gradient producer of the fused step (weight-gradient GEMMs, embedding scatter, LayerNorm / loss partial
    reductions) is a job of the step's LAST launch (k_wgrad, csrc/linear.hip), so no bucket is complete before the backward
    has ended and a second collective would only add its latency;
  * dr4sr_adam_step divides by the all-reduced n_valid, i.e. the reference's global-batch normalisation
    (loss_func.py:18-19, :29-30), and every replica takes the bit-identical dense Adam step.

Transport.  `nccl` (RCCL) reduces device buffers in place and can be captured into a HIP graph (`allreduce_flat` inside
`torch.cuda.graph`).  `gloo` (DR4SR_DP_BACKEND=gloo) exists for functional checks of the N-rank code path on ONE GPU — RCCL refuses
two ranks on one device — and stages the buffer through the host; it cannot be captured (`can_capture()` is False) and is never a
performance path.
"""
from __future__ import annotations

import os
from typing import Tuple


def shard_bounds(i: int, B: int, n: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) positions (in the epoch permutation) of rank `rank`'s slice of global batch i of size B over n rows."""
    g0 = i * B
    gl = max(0, min(B, n - g0))
    per = (B + world - 1) // world
    return g0 + min(rank * per, gl), g0 + min((rank + 1) * per, gl)


def world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def backend_name() -> str:
    return os.environ.get("DR4SR_DP_BACKEND", os.environ.get("DR4SR_BENCH_BACKEND", "nccl")).lower()


def init_distributed(device=None):
    """Create the default process group once (no-op for a single process or when the launcher's script already did).
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run."""
    import torch
    import torch.distributed as dist
    if world_size() <= 1 and not os.environ.get("DR4SR_BENCH_FORCE_DP"):
        return False
    if dist.is_initialized():
        return True
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend_name() == "nccl":
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group("nccl", device_id=torch.device(device))
    else:
        dist.init_process_group(backend_name())
    return True


def can_capture() -> bool:
    """True when collectives on device buffers may be recorded into a HIP graph (RCCL)"""
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "nccl"


def _staged(t) -> bool:
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() != "nccl"


def allreduce_flat(grads, group=None):
    """sum-all-reduce of (a slice of) the flat gradient buffer INCLUDING its {n_valid, loss_sum, poison} tail"""
    import torch.distributed as dist
    if _staged(grads):
        h = grads.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        grads.copy_(h)
    else:
        dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    return grads


def all_gather_flat(t, group=None):
    """every rank's equally sized flat buffer, as one [world, n] tensor on t's device (CL4SRec: the pooled views + n_valid of each
    rank — InfoNCE's in-batch negatives are the GLOBAL batch)"""
    import torch
    import torch.distributed as dist
    W = dist.get_world_size(group)
    if _staged(t):
        h = t.cpu()
        out = torch.empty(W, h.numel(), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h.view(1, -1), group=group) if dist.get_backend() == "nccl" else \
            dist.all_gather(list(out.unbind(0)), h.view(-1), group=group)
        return out.to(t.device)
    out = torch.empty(W, t.numel(), dtype=t.dtype, device=t.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, t.view(1, -1), group=group)
    else:
        dist.all_gather(list(out.unbind(0)), t.view(-1), group=group)
    return out


def broadcast(t, src: int = 0, group=None):
    import torch.distributed as dist
    if _staged(t):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def barrier():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
