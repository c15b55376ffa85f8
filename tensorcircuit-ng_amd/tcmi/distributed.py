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


class AbiCommunicator:
    """The C ABI's own collective (``tcmi_comm_* / tcmi_allreduce_sum``: RCCL opened by libtcmi.so at first use) for hosts
    without a ``torch.distributed`` process group -- the reference's JAX / numpy processes bound through ctypes.  The
    package itself keeps using ``allreduce_sum_packed`` above; this class is the Python face of the ABI entries and what
    their GPU test drives.

        uid = AbiCommunicator.unique_id()            # on ONE rank; hand the 128 bytes to the others out of band
        comm = AbiCommunicator(uid, rank, world)     # every rank, after selecting its device (collective)
        comm.allreduce_sum_(tensor)                  # in place, on the current stream
    """

    def __init__(self, unique_id: bytes, rank: int, world: int, librccl: str = None):
        import ctypes

        from . import _lib

        self._lib = _lib.lib()
        if librccl is None:
            librccl = self._torch_rccl()
        if librccl:
            _lib.check(self._lib.tcmi_comm_load(librccl.encode()), "tcmi_comm_load")
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of AbiCommunicator.unique_id()")
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self._lib.tcmi_comm_init(buf, int(rank), int(world), ctypes.byref(h)), "tcmi_comm_init")
        self.handle, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def _torch_rccl():
        """The librccl torch ships (one RCCL instance per process when torch is the array container)."""
        import os

        try:
            import torch

            p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            return p if os.path.exists(p) else None
        except ImportError:
            return None

    @staticmethod
    def unique_id(librccl: str = None) -> bytes:
        import ctypes

        from . import _lib

        L = _lib.lib()
        librccl = librccl or AbiCommunicator._torch_rccl()
        if librccl:
            _lib.check(L.tcmi_comm_load(librccl.encode()), "tcmi_comm_load")
        buf = ctypes.create_string_buffer(128)
        _lib.check(L.tcmi_comm_unique_id(buf), "tcmi_comm_unique_id")
        return buf.raw

    def allreduce_sum_(self, t):
        import torch

        from . import _lib

        code = {torch.float32: _lib.TCMI_F32, torch.float64: _lib.TCMI_F64, torch.complex64: _lib.TCMI_C64,
                torch.complex128: _lib.TCMI_C128}.get(t.dtype)
        if code is None or not t.is_cuda or not t.is_contiguous():
            raise ValueError("allreduce_sum_ takes a contiguous float32/64 or complex64/128 device tensor")
        _lib.check(self._lib.tcmi_allreduce_sum(self.handle, t.data_ptr(), t.numel(), code,
                                                torch.cuda.current_stream(t.device).cuda_stream), "tcmi_allreduce_sum")
        return t

    def close(self):
        from . import _lib

        if self.handle is not None:
            _lib.check(self._lib.tcmi_comm_destroy(self.handle), "tcmi_comm_destroy")
            self.handle = None
