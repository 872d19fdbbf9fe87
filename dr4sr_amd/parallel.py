"""Single-node data parallelism over users (one process per GPU, torch.distributed 'nccl' = RCCL over xGMI).

The reference has no distributed path (/root/reference utils/callbacks.py:130 is a TODO).  The scheme:
  * every rank holds a full replica (encoder + item table, 3.3 MB fp32 for toys d=64) and the full dataset tensors;
  * one global permutation per epoch (rank 0's, broadcast); the i-th GLOBAL batch of B rows is split into
    contiguous slices of ceil(B/W) rows, rank r takes slice r (`shard_bounds`);
  * each rank accumulates UN-normalised gradients of its slice plus the tail {n_valid, loss_sum, poison}
    (include/dr4sr_hip.h "Flat parameter layout"), then sum-all-reduces the flat buffer:
      - latency launch forms (small per-rank batches): ONE all-reduce of the whole buffer — every gradient producer of the fused step
        is a job of the step's last launch, nothing is final earlier;
      - at-scale launch forms: TWO buckets (`dp_backward`, SURVEY.md §8(e) "optionally 2 buckets ... overlapped with backward").  The
        engine cuts the last backward launch in two (dr4sr_sasrec_fwd_bwd_phase): after the first the item + position table
        gradient — 92 % of the bytes — is final and its all-reduce is issued asynchronously, so it runs BESIDE the second launch
        (the remaining weight-gradient GEMMs); the 280 KB encoder bucket + tail follows, and only that one is exposed;
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


def allreduce_begin(t, group=None):
    """start the sum-all-reduce of `t` (a contiguous slice of the flat gradient); returns a handle for `allreduce_end`.
    RCCL: asynchronous — the collective is ordered behind everything already enqueued on the current stream and runs on the process
    group's own stream; the current stream does not wait for it before `allreduce_end`, so kernels enqueued in between run beside it
    (inside a graph capture the two become parallel branches of the graph).  gloo (functional runs on one GPU): reduced on the spot
    through the host, handle None."""
    import torch.distributed as dist
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_end(handle):
    """the current stream waits for the collective started by `allreduce_begin` (no host wait with RCCL)"""
    if handle is not None:
        handle.wait()


def grad_buckets(eng, rows, seqlen=None):
    """[(lo, hi), ...] float ranges of eng.grads all-reduced one after the other in a data-parallel step whose FULL per-rank slice holds
    `rows` rows (the tail sits in the last range).  The answer must be the same on every rank — ranks that disagreed on the number of
    collectives of a step would deadlock — so it is a function of `rows` (ceil(global batch / world), equal everywhere) and of the
    replicated dataset's mean length, never of a rank's own (possibly short or empty) slice: a rank whose own slice runs the latency
    launch forms under a two-bucket decision simply has both buckets final after phase 1 (dp_backward).  One range: rows=None (partial
    tail batches), engines without the two-phase step (GRU4Rec, FMLP), DR4SR_DP_FLAT (cross-check)."""
    flat = [(0, int(eng.grads.numel()))]
    if rows is None or os.environ.get("DR4SR_DP_FLAT") or not hasattr(eng, "grad_buckets_for"):
        return flat
    return eng.grad_buckets_for(int(rows), seqlen)


def dp_backward(eng, plan, prepared: bool, buckets=None, reduce: bool = True):
    """backward of one data-parallel step + the sum-all-reduce of its gradient; `buckets` from grad_buckets (None = flat).  One bucket:
    fwd_bwd[_prepared] then the flat all-reduce.  Two buckets: phase 1 -> all-reduce(table bucket) started -> phase 2 (runs beside it) -> all-reduce(encoder bucket + tail)
    -> the current stream joins both.  reduce=False: the kernels only (graph warm-ups must not enter a collective)."""
    if buckets is None or len(buckets) == 1:
        (eng.fwd_bwd_prepared if prepared else eng.fwd_bwd)(plan)
        if reduce:
            allreduce_flat(eng.grads)
        return
    eng.fwd_bwd_phase(plan, prepared, 1)
    h0 = allreduce_begin(eng.grads[buckets[0][0]:buckets[0][1]]) if reduce else None
    eng.fwd_bwd_phase(plan, prepared, 2)
    h1 = allreduce_begin(eng.grads[buckets[1][0]:buckets[1][1]]) if reduce else None
    allreduce_end(h0)
    allreduce_end(h1)


def dp_reduce_empty(eng, buckets=None):
    """a rank whose slice of this global batch is EMPTY: contribute zeros to the same collectives, in the same order, as the ranks
    that have rows (`buckets` = the global decision of grad_buckets, like theirs)"""
    eng.grads.zero_()
    if buckets is None or len(buckets) == 1:
        allreduce_flat(eng.grads)
        return
    hs = [allreduce_begin(eng.grads[lo:hi]) for lo, hi in buckets]
    for h in hs:
        allreduce_end(h)


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
