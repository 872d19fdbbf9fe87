"""HIP-graph capture helper.

`capture(g, ...)` = `torch.cuda.graph(g, capture_error_mode="thread_local", ...)` with the Python garbage collector held off for the
duration: a cyclic-GC pass that happens to run INSIDE a capture and finalises an older `torch.cuda.CUDAGraph` / device tensor
issues hipGraphExecDestroy / hipFree on the capturing thread, which is illegal during capture and aborts the process (seen in
the test suite: "Fatal Python error: Aborted ... Garbage-collecting" inside BaseModel._api_step_graph after earlier tests had left
graph-holding models behind).  torch.cuda.graph collects BEFORE the capture for the same reason; it does not stop a collection
that an allocation inside the captured region triggers.
"""
from __future__ import annotations

import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(g: "torch.cuda.CUDAGraph", **kw):
    kw.setdefault("capture_error_mode", "thread_local")      # other threads (RCCL watchdog) may touch the device meanwhile
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g, **kw):
            yield g
    finally:
        if was:
            gc.enable()
