"""Multi-GPU helpers: one process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm).

The path shards two ways, both present in the reference:

* vmap batch -- samples are independent (``jax_backend.py:933-937``); values are gathered, gradients
  of shared arguments are summed (``jax_backend.py:945-947``);
* contraction slices -- ids ``0..S-1`` laid out row-major into ``[G, ceil(S/G)]`` padded with ``-1``
  (reference ``tensorcircuit/experimental.py:881-890``), padded ids contribute nothing, results
  are summed (``experimental.py:1145-1152``).

The only collective is one small all-reduce of ``[value || flattened gradients]`` per step
(a few KB: latency-bound over xGMI, bandwidth irrelevant).
"""

import math
from typing import List, Sequence, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of a batch of ``total`` samples owned by ``rank`` (ceil split, may be empty)."""
    per = math.ceil(total / world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def slice_table(num_slices: int, world: int) -> np.ndarray:
    """``[world, ceil(S/world)]`` int32 table of slice ids padded with -1
    (reference experimental.py:881-890)."""
    per = math.ceil(num_slices / world) if num_slices else 0
    tab = -np.ones((world, max(per, 1)), dtype=np.int32)
    for s in range(num_slices):
        tab[s // per, s % per] = s
    return tab


def allreduce_sum_packed(tensors: Sequence, group=None):
    """Sum a list of tensors over all ranks with ONE collective: pack -> all_reduce -> unpack.
    Works with any initialised ``torch.distributed`` backend (RCCL on GPUs, gloo in the CPU tests);
    a single-process run returns the inputs unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, off = [], 0
    for t in tensors:
        k = t.numel()
        out.append(flat[off: off + k].reshape(t.shape).to(t.dtype))
        off += k
    return out
