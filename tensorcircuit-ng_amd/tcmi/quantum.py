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


def sample2all(sample: Tensor, n: int, format: str = "count_vector", jittable: bool = False, dim: Optional[int] = None) -> Any:
    """reference quantum.py ``sample2all``: convert ``sample_int`` ([shots]) or ``sample_bin`` ([shots, n])
    results to ``sample_int | sample_bin | count_vector | count_tuple | count_dict_bin | count_dict_int``."""
    import torch
    from collections import Counter

    s = cons.backend.convert_to_tensor(sample)
    if s.dim() == 2:
        sample_bin = s.to(torch.int64)
        if n > 62:
            if format == "sample_bin":
                return sample_bin
            if format == "count_dict_bin":
                return dict(Counter("".join(str(int(v)) for v in shot) for shot in sample_bin.cpu().tolist()))
            raise ValueError(f"n={n} is too large for measurement representaion: {format}")
        w = (2 ** torch.arange(n - 1, -1, -1, device=s.device, dtype=torch.int64))
        sample_int = (sample_bin * w).sum(-1)
    elif s.dim() == 1:
        sample_int = s.to(torch.int64)
        sample_bin = (sample_int.unsqueeze(-1) >> torch.arange(n - 1, -1, -1, device=s.device, dtype=torch.int64)) & 1
    else:
        raise ValueError("unrecognized tensor shape for sample")
    if format == "sample_int":
        return sample_int
    if format == "sample_bin":
        return sample_bin
    vals, counts = torch.unique(sample_int, return_counts=True)
    if format == "count_tuple":
        return vals, counts
    if format == "count_vector":
        out = torch.zeros(2 ** n, dtype=torch.int64, device=s.device)
        out[vals] = counts
        return out
    if format == "count_dict_bin":
        return {format_bin(int(v), n): int(c) for v, c in zip(vals.cpu().tolist(), counts.cpu().tolist())}
    if format == "count_dict_int":
        return {int(v): int(c) for v, c in zip(vals.cpu().tolist(), counts.cpu().tolist())}
    raise ValueError(f"unsupported format {format}")


def format_bin(v: int, n: int) -> str:
    return format(v, "0%db" % n)


def reduced_density_matrix(state: Tensor, cut: Any, p: Optional[Tensor] = None) -> Tensor:
    """reference quantum.py ``reduced_density_matrix`` for a pure state: trace out the qubits in ``cut`` (an
    index list, or an int = the first ``cut`` qubits); the partial trace is one ``tcmi_cgemm``."""
    import torch
    from . import linalg as LA

    psi = cons.backend.convert_to_tensor(state).reshape(-1)
    n = int(round(np.log2(psi.numel())))
    tr = list(range(cut)) if isinstance(cut, int) else [int(c) % n for c in cut]
    keep = [i for i in range(n) if i not in tr]
    m = psi.reshape([2] * n).permute(keep + tr).reshape(2 ** len(keep), -1).contiguous()
    if p is not None:
        m = m * cons.backend.cast(cons.backend.convert_to_tensor(p), cons.dtypestr).sqrt().reshape(1, -1)
    return LA.matmul(m, m.conj().t().resolve_conj())


class QuVector:
    """The part of the reference's ``QuVector`` (tensorcircuit/quantum.py) the circuit constructors use: a state given as
    an MPS -- site tensors ``[bond-left, physical, bond-right]``, the layout of ``Circuit(n, tensors=...)``
    (reference basecircuit.py:72-102) and of ``MPSCircuit.get_tensors`` -- or as a dense vector (``Circuit.quvector()``,
    reference basecircuit.py ``quvector``).  ``eval`` is the dense state: the chain is contracted left to right, one
    ``tcmi_cgemm`` per site (``backend.matmul``: differentiable through its own backward rule)."""

    def __init__(self, tensors: Optional[Sequence[Any]] = None, dense: Optional[Any] = None):
        if (tensors is None) == (dense is None):
            raise ValueError("QuVector takes either MPS tensors or a dense state")
        self.tensors = list(tensors) if tensors is not None else None
        self.dense = dense

    @classmethod
    def from_tensors(cls, tensors: Sequence[Any]) -> "QuVector":
        return cls(tensors=tensors)

    @property
    def n(self) -> int:
        if self.tensors is not None:
            return len(self.tensors)
        return int(self.dense.numel() if hasattr(self.dense, "numel") else np.size(self.dense)).bit_length() - 1

    def eval(self) -> Tensor:
        K = cons.backend
        if self.dense is not None:
            return K.reshape(K.cast(K.convert_to_tensor(self.dense), cons.dtypestr), [-1])
        ts = [K.cast(K.convert_to_tensor(t), cons.dtypestr) for t in self.tensors]
        for j, t in enumerate(ts):
            if len(t.shape) != 3 or int(t.shape[1]) != 2:
                raise ValueError(f"MPS tensor {j} has shape {tuple(t.shape)}; expected (bond-left, 2, bond-right)")
        if int(ts[0].shape[0]) != 1 or int(ts[-1].shape[2]) != 1:
            raise ValueError("an MPS state needs boundary bonds of dimension 1")
        acc = K.reshape(ts[0], [2, int(ts[0].shape[2])])               # [2^j, D_j]
        for t in ts[1:]:
            dl, dr = int(t.shape[0]), int(t.shape[2])
            if dl != int(acc.shape[1]):
                raise ValueError("MPS bond dimensions do not match")
            acc = K.reshape(K.matmul(acc, K.reshape(t, [dl, 2 * dr])), [-1, dr])
        return K.reshape(acc, [-1])

    eval_matrix = eval


class QuOperator:
    """The part of the reference's ``QuOperator`` (tensorcircuit/quantum.py) the hot path's callers use: an operator on n
    qubits given either as ONE local tensor on some sites (``from_local_tensor``, the MPO of
    templates.measurements.mpo_expectation's known-answer test, reference tests/test_templates.py:191-211) or as a dense
    matrix (``Circuit.get_quoperator``).  There is no tensor-network algebra here: ``eval_matrix`` is the dense matrix."""

    def __init__(self, n: int, local: Optional[Any] = None, loc: Optional[Sequence[int]] = None, dense: Optional[Any] = None):
        self.n, self.local, self.loc, self.dense = int(n), local, (list(loc) if loc is not None else None), dense

    @classmethod
    def from_local_tensor(cls, tensor: Any, space: Sequence[int], loc: Sequence[int], out_axes=None, in_axes=None):
        """reference quantum.py ``QuOperator.from_local_tensor``: ``tensor`` acts on the sites ``loc`` of a register with
        local dimensions ``space`` (qubits only here); identity elsewhere."""
        if any(int(d) != 2 for d in space):
            raise NotImplementedError("Backend 'hip' has not implemented QuOperator on local dimensions other than 2")
        k = len(loc)
        t = cons.backend.reshape(cons.backend.cast(cons.backend.convert_to_tensor(tensor), cons.dtypestr), [2**k, 2**k])
        return cls(len(space), local=t, loc=loc)

    @classmethod
    def from_matrix(cls, matrix: Any, n: int):
        return cls(n, dense=matrix)

    def eval_matrix(self) -> Tensor:
        K = cons.backend
        if self.dense is not None:
            return self.dense
        if self.n > 12:
            raise NotImplementedError("Backend 'hip' has not implemented dense QuOperator matrices beyond 12 qubits")
        k = len(self.loc)
        rest = [q for q in range(self.n) if q not in self.loc]
        full = K.reshape(K.kron(self.local, K.eye(2 ** len(rest), dtype=cons.dtypestr)), [2] * (2 * self.n))
        order = list(self.loc) + rest                       # axis j of `full` (both halves) is qubit order[j]
        perm = [order.index(q) for q in range(self.n)]
        return K.reshape(K.transpose(full, perm + [self.n + x for x in perm]), [2**self.n, 2**self.n])

    eval = eval_matrix
