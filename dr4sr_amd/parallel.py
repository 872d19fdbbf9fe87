"""Single-node data parallelism over users (one process per GPU, torch.distributed 'nccl' = RCCL over xGMI).

The reference has no distributed path.  The scheme:
  * every rank holds a full replica (encoder + item table, 3.3 MB fp32 for toys d=64) and the full dataset tensors;
  * one global permutation per epoch (rank 0's, broadcast); the i-th GLOBAL batch of B rows is split into
    contiguous slices of ceil(B/W) rows, rank r takes slice r (`shard_bounds`);
  * each rank accumulates UN-normalised gradients of its slice plus the tail {n_valid, loss_sum}
    (include/dr4sr_hip.h "Flat parameter layout"), then ONE sum-all-reduce of the whole flat buffer;
  * dr4sr_adam_step divides by the all-reduced n_valid, i.e. the reference's global-batch normalisation
    (loss_func.py:18-19, :29-30), and every replica takes the bit-identical dense Adam step.
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(i: int, B: int, n: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) positions (in the epoch permutation) of rank `rank`'s slice of global batch i of size B over n rows."""
    g0 = i * B
    gl = max(0, min(B, n - g0))
    per = (B + world - 1) // world
    return g0 + min(rank * per, gl), g0 + min((rank + 1) * per, gl)


def allreduce_flat(grads, group=None):
    """sum-all-reduce of the flat gradient buffer INCLUDING its {n_valid, loss_sum} tail"""
    import torch.distributed as dist
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    return grads
