"""Pauli-sum Hamiltonians without the matrix.

Reference: ``tensorcircuit/quantum.py:2222-2358`` (``PauliStringSum2MVP``: matrix-vector product of a
Pauli-string sum without building the matrix) and ``PauliStringSum2COO`` (the sparse form the reference
recommends for big Hamiltonians, ``docs/source/faq.rst:101``).  On the hip backend the "sparse
Hamiltonian" is the list of strings itself: ``H|psi>`` is one launch of ``tcmi_apply_pauli_sum`` (gathers
by X-mask, signs by Z-mask) and ``<psi|H|psi>`` goes through the fused measurement passes, so the 2^n x
2^n COO matrix is never materialised."""

from typing import Any, Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from . import cons

Tensor = Any


class PauliSum:
    """``sum_i weights[i] * P(structures[i])`` with structure entries 0, 1, 2, 3 = I, X, Y, Z (qubit 0 first)."""

    is_pauli_sum = True

    def __init__(self, structures: Sequence[Sequence[int]], weights: Optional[Sequence[float]] = None):
        self.structures = [[int(v) for v in s] for s in np.asarray(structures).tolist()]
        if len(self.structures) == 0:
            raise ValueError("empty Pauli sum")
        self.n = len(self.structures[0])
        if weights is None:
            weights = [1.0] * len(self.structures)
        self.weights = [float(np.real(w)) for w in np.asarray(cons.backend.numpy(weights) if not isinstance(
            weights, (list, tuple, np.ndarray)) else weights).reshape(-1)]
        if len(self.weights) != len(self.structures):
            raise ValueError("weights and structures differ in length")
        self._table = None

    def __len__(self) -> int:
        return len(self.structures)

    def _device_table(self, device, n_exec):
        import torch
        from .executor import _dev

        if self._table is None or self._table[0] != (device, n_exec):
            rows = []
            for k, s in enumerate(self.structures):
                xm = zm = ny = 0
                for q, p in enumerate(s):
                    bit = 1 << (n_exec - 1 - (q + n_exec - self.n))
                    if p in (1, 2):
                        xm |= bit
                    if p in (2, 3):
                        zm |= bit
                    ny += p == 2
                rows.append((xm, zm, ny, k))
            rows.sort(key=lambda r: r[0])
            arr = np.array([[r[0], r[1], r[2]] for r in rows], dtype=np.int64).astype(np.uint32).view(np.int32)
            w = np.array([self.weights[r[3]] for r in rows], dtype=np.float64)
            self._table = ((device, n_exec), _dev(arr.reshape(-1, 3), device), _dev(w.reshape(1, -1), device))
        return self._table[1], self._table[2]

    def matvec(self, psi: Tensor) -> Tensor:
        """``H |psi>`` for a state (or batch of states) of 2^n amplitudes."""
        import torch

        psi = cons.backend.convert_to_tensor(psi)
        shape = psi.shape
        st = psi.reshape(-1, 2**self.n).contiguous()
        if not st.is_cuda:
            raise _lib.TcmiError("Backend 'hip': the Pauli-sum matrix-vector product runs on the GPU only")
        code = _lib.TCMI_C64 if st.dtype == torch.complex64 else _lib.TCMI_C128
        terms, w = self._device_table(st.device, self.n)
        B = st.shape[0]
        wb = w.expand(B, -1).contiguous()
        out = torch.empty_like(st)
        stream = torch.cuda.current_stream(st.device).cuda_stream
        _lib.check(_lib.lib().tcmi_apply_pauli_sum(st.data_ptr(), out.data_ptr(), st.shape[1], B, self.n,
                                                   terms.data_ptr(), len(self), wb.data_ptr(), wb.stride(0), code,
                                                   stream), "tcmi_apply_pauli_sum")
        return out.reshape(shape)


def PauliStringSum2COO(ls: Sequence[Sequence[int]], weight: Optional[Sequence[float]] = None,
                       numpy: bool = False) -> PauliSum:
    """reference quantum.py ``PauliStringSum2COO``: returns the matrix-free ``PauliSum`` (what the hip backend
    treats as its sparse operator: ``backend.is_sparse`` is True for it)."""
    return PauliSum(ls, weight)


PauliStringSum2COO_numpy = PauliStringSum2COO
PauliStringSum2COO_tf = PauliStringSum2COO


def PauliStringSum2MVP(structures: Sequence[Sequence[int]], weights: Sequence[float]) -> Callable[[Tensor], Tensor]:
    """reference quantum.py:2222-2358: ``mvp(psi) = sum_i w_i P_i psi``."""
    if not len(structures):
        return lambda psi: cons.backend.zeros_like(psi)
    h = PauliSum(structures, weights)
    return h.matvec
