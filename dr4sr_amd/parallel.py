"""Single-node data parallelism over users (one process per GPU; collectives = RCCL over xGMI through libdr4sr_hip.so's own C ABI).

The reference has no distributed path (/root/reference utils/callbacks.py:130 is a TODO).  The scheme:
  * every rank holds a full replica (encoder + item table, 3.3 MB fp32 for toys d=64) and the full dataset tensors;
  * one global permutation per epoch (rank 0's, broadcast); the i-th GLOBAL batch of B rows is split into
    contiguous slices of ceil(B/W) rows, rank r takes slice r (`shard_bounds`);
  * each rank accumulates UN-normalised gradients of its slice plus the tail {n_valid, loss_sum, poison}
    (include/dr4sr_hip.h "Flat parameter layout"), then sum-all-reduces the flat buffer:
      - latency launch forms (small per-rank batches): ONE all-reduce of the whole buffer — every gradient producer of the fused step
        is a job of the step's last launch, nothing is final earlier;
      - at-scale launch forms: optionally TWO buckets (`dp_backward`, SURVEY.md §8(e) "optionally 2 buckets ... overlapped with
        backward").  The engine cuts the last backward launch in two (dr4sr_sasrec_fwd_bwd_phase): after the first the item + position
        table gradient — 92 % of the bytes — is final and its all-reduce is started on the communicator's side stream, so it runs BESIDE
        the second launch (the remaining weight-gradient GEMMs); the 280 KB encoder bucket + tail follows, and only that one is exposed;
  * dr4sr_adam_step divides by the all-reduced n_valid, i.e. the reference's global-batch normalisation
    (loss_func.py:18-19, :29-30), and every replica takes the bit-identical dense Adam step.

Transport (round 6).  Two planes:
  * DATA plane, `rccl` (default): the library's own communicator (include/dr4sr_hip.h ABI 8, csrc/comm.hip: ncclAllReduce / ncclAllGather /
    ncclBroadcast enqueued on the CALLER's stream).  A collective is ordered like a kernel launch — no Work object, no watchdog or heartbeat
    thread, capturable into the step's HIP graph by construction (`can_capture()`).  torch.distributed's ProcessGroupNCCL is never
    created: an exception escaping one of its background threads aborted a graph-replay loop on the round-5 driver box.
  * CONTROL plane: a CPU `gloo` group (rendezvous through torch.distributed.run's store): carries the 128-byte RCCL unique id, host
    decisions that every rank must take alike (`all_ok`), timings (`host_allreduce`) and barriers.  Never on the step's critical path.
`gloo` as the DATA plane (DR4SR_DP_BACKEND=gloo) exists for functional checks of the N-rank code path on ONE GPU — RCCL refuses two
ranks on one device — and stages device buffers through the host; it cannot be captured and is never a performance path.
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
    """the DATA plane: 'rccl' (native communicator of libdr4sr_hip.so; 'nccl' is accepted as its old name) or 'gloo' (staged, functional)"""
    b = os.environ.get("DR4SR_DP_BACKEND", os.environ.get("DR4SR_BENCH_BACKEND", "rccl")).lower()
    return "rccl" if b == "nccl" else b


_COMM = None            # ctypes handle of the native communicator (dr4sr_comm*), None = staged gloo data plane
_COMM_DEVICE = None
FALLBACK_REASON = None  # set when init_distributed(allow_fallback=True) could not create the RCCL communicator on every rank


def _native():
    return _COMM is not None


def _check(rc: int, what: str):
    if rc != 0:
        from . import _lib
        lib = _lib.load()
        msg = lib.dr4sr_comm_error_string(rc)
        raise _lib.Dr4srError("%s failed: rc %d (%s)" % (what, rc, msg.decode() if msg else "?"))


def init_distributed(device=None, allow_fallback=False, init_timeout=None):
    """Create the control group (gloo) and, for the rccl data plane, the native communicator — once (no-op for a single process unless
    DR4SR_BENCH_FORCE_DP asks for the 1-rank form).  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from torch.distributed.run.
    Collective: every rank of the job must call it.  A communicator that cannot be created on EVERY rank raises (training must not silently
    run over a slow transport); allow_fallback=True (bench.py: a flagged line beats no line) drops all ranks to the host-staged gloo data
    plane instead and records why in FALLBACK_REASON.  init_timeout (seconds, bench.py only): the communicator's bootstrap runs in a helper
    thread and a rank on which it has not returned in time counts as failed — a wedged bootstrap then costs the timeout and a flagged line
    instead of the whole run (the ranks agree over the control plane; None = wait for it, as training does)."""
    global _COMM, _COMM_DEVICE, FALLBACK_REASON
    import torch
    import torch.distributed as dist
    if world_size() <= 1 and not os.environ.get("DR4SR_BENCH_FORCE_DP"):
        return False
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    if backend_name() == "rccl" and _COMM is None:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ident = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if dist.get_rank() == 0:
            _check(lib.dr4sr_comm_unique_id(ident), "dr4sr_comm_unique_id")
        box = [bytes(ident.raw)]
        dist.broadcast_object_list(box, src=0)                       # the id travels over the control plane
        ident = C.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
        torch.cuda.synchronize(device)
        handle, err = C.c_void_p(), None
        rank_, world_ = dist.get_rank(), dist.get_world_size()

        def boot():
            _check(lib.dr4sr_comm_init_rank(ident, rank_, world_, int(idx), C.byref(handle)), "dr4sr_comm_init_rank")
        try:
            if init_timeout is None:
                boot()
            else:                                            # (ctypes releases the GIL for the call: the wait below really runs)
                import threading
                box_err = []

                def guarded():
                    try:
                        boot()
                    except Exception as e:      # noqa: BLE001
                        box_err.append(e)
                th = threading.Thread(target=guarded, name="dr4sr-comm-bootstrap", daemon=True)
                th.start()
                th.join(float(init_timeout))
                if th.is_alive():
                    raise TimeoutError("dr4sr_comm_init_rank did not return within %.0f s" % float(init_timeout))
                if box_err:
                    raise box_err[0]
        except Exception as e:      # noqa: BLE001 — decided below, on every rank alike
            err = e
        if all_ok(err is None):
            _COMM, _COMM_DEVICE = handle, device
            import atexit
            atexit.register(shutdown)
        else:
            if err is None:
                lib.dr4sr_comm_destroy(handle)
            elif isinstance(err, TimeoutError):
                handle = C.c_void_p()                        # (the helper thread may still write the old one: never touched again)
            why = "the RCCL communicator could not be created on every rank (this rank: %s)" % (err if err is not None else "ok")
            if not allow_fallback:
                raise _lib.Dr4srError(why)
            FALLBACK_REASON = why + "; host-staged gloo data plane instead (functional, not a performance path)"
    return True


def shutdown():
    """destroy the native communicator (if any) and the control group; safe to call twice"""
    global _COMM, _COMM_DEVICE
    import torch.distributed as dist
    if _COMM is not None:
        from . import _lib
        import torch
        try:
            torch.cuda.synchronize(_COMM_DEVICE)
        except Exception:      # noqa: BLE001
            pass
        h, _COMM, _COMM_DEVICE = _COMM, None, None
        _lib.load().dr4sr_comm_destroy(h)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def can_capture() -> bool:
    """True when collectives on device buffers may be recorded into a HIP graph (the native RCCL data plane)"""
    return _native()


def _staged(t) -> bool:
    return t.is_cuda and not _native()


def _ptr_stream(t):
    from . import _lib
    assert t.is_cuda and t.is_contiguous(), "collectives take contiguous device tensors"
    return _lib.ptr(t), _lib.cur_stream()


def _lib_allreduce(t, fn_async=False):
    import torch
    from . import _lib
    lib = _lib.load()
    p, st = _ptr_stream(t)
    if t.dtype == torch.float32:
        fn = lib.dr4sr_allreduce_f32_async if fn_async else lib.dr4sr_allreduce_f32
        _check(fn(_COMM, p, t.numel(), st), "dr4sr_allreduce_f32")
    elif t.dtype == torch.float64 and not fn_async:
        _check(lib.dr4sr_allreduce_f64(_COMM, p, t.numel(), 0, st), "dr4sr_allreduce_f64")
    else:
        raise TypeError("allreduce of %s is not part of the transport (float32; float64 blocking)" % t.dtype)


def allreduce_flat(grads, group=None):
    """sum-all-reduce of (a slice of) the flat gradient buffer INCLUDING its {n_valid, loss_sum, poison} tail, ordered on the current stream"""
    import torch.distributed as dist
    if grads.is_cuda and _native():
        _lib_allreduce(grads)
    elif _staged(grads):
        h = grads.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        grads.copy_(h)
    else:
        dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    return grads


def allreduce_begin(t, group=None):
    """start the sum-all-reduce of `t` (a contiguous slice of the flat gradient); returns a handle for `allreduce_end`.
    RCCL: asynchronous (dr4sr_allreduce_f32_async) — the collective is ordered behind everything already enqueued on the current stream
    and runs on the communicator's side stream; the current stream does not wait for it before `allreduce_end`, so kernels enqueued in
    between run beside it (inside a graph capture the two become parallel branches of the graph).  gloo (functional runs on one GPU, CPU
    tests): reduced on the spot, handle None."""
    import torch.distributed as dist
    if t.is_cuda and _native():
        _lib_allreduce(t, fn_async=True)
        return "rccl"
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return None


def allreduce_end(handle):
    """the current stream waits for the collectives started by `allreduce_begin` (dr4sr_comm_join: no host wait)"""
    if handle is not None:
        from . import _lib
        _check(_lib.load().dr4sr_comm_join(_COMM, _lib.cur_stream()), "dr4sr_comm_join")


def all_ok(ok: bool) -> bool:
    """control plane: True iff `ok` on EVERY rank (a decision all ranks must take alike, e.g. "my graph capture succeeded")"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return bool(ok)
    f = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return float(f) >= 1.0


def host_allreduce(values, op: str = "max"):
    """control plane: element-wise max / min / sum over ranks of a list of host floats (timings); returns a list"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
    return [float(x) for x in t]


def host_allgather(value: float):
    """control plane: every rank's host float, in rank order"""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(value)], dtype=torch.float64)
    if not dist.is_initialized():
        return [float(mine)]
    outs = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, mine)
    return [float(x) for x in outs]


def grad_buckets(eng, rows, seqlen=None, want=None):
    """[(lo, hi), ...] float ranges of eng.grads all-reduced one after the other in a data-parallel step whose FULL per-rank slice holds
    `rows` rows (the tail sits in the last range).  The answer must be the same on every rank — ranks that disagreed on the number of
    collectives of a step would deadlock — so it is a function of `rows` (ceil(global batch / world), equal everywhere) and of the
    replicated dataset's mean length, never of a rank's own (possibly short or empty) slice: a rank whose own slice runs the latency
    launch forms under a two-bucket decision simply has both buckets final after phase 1 (dp_backward).
    ONE flat range by default (round 6): with one RCCL rank the bucketed in-graph step measured 21-36 us SLOWER than the flat one at every
    at-scale size (the launch cut; BENCH_r05 strong[].dp_1rank_rccl) and what it hides — the table bucket's wire time — has never been
    measured on a multi-GPU node.  Two buckets are opt-in: want=2 (`train.dp_buckets: 2`) or DR4SR_DP_BUCKETS=2; bench.py measures both
    in-graph forms on a real node and reports the faster.  Always one range: rows=None (partial tail batches), engines without the
    two-phase step (GRU4Rec, FMLP)."""
    flat = [(0, int(eng.grads.numel()))]
    if want is None:
        want = int(os.environ.get("DR4SR_DP_BUCKETS", "1") or 1)
    if rows is None or int(want) < 2 or not hasattr(eng, "grad_buckets_for"):
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
    if t.is_cuda and _native():
        from . import _lib
        t = t.contiguous()
        out = torch.empty(W, t.numel(), dtype=t.dtype, device=t.device)
        _check(_lib.load().dr4sr_allgather_bytes(_COMM, _lib.ptr(t), _lib.ptr(out), t.numel() * t.element_size(), _lib.cur_stream()),
               "dr4sr_allgather_bytes")
        return out
    h = t.cpu() if t.is_cuda else t
    out = torch.empty(W, h.numel(), dtype=h.dtype)
    dist.all_gather(list(out.unbind(0)), h.reshape(-1), group=group)
    return out.to(t.device)


def broadcast(t, src: int = 0, group=None):
    import torch.distributed as dist
    if t.is_cuda and _native():
        from . import _lib
        assert t.is_contiguous()
        _check(_lib.load().dr4sr_broadcast_bytes(_COMM, _lib.ptr(t), t.numel() * t.element_size(), int(src), _lib.cur_stream()),
               "dr4sr_broadcast_bytes")
    elif _staged(t):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def barrier():
    """control-plane barrier (host).  Device work is NOT drained by it: callers that time regions add torch.cuda.synchronize()."""
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
