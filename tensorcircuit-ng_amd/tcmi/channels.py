"""Kraus channels (reference ``tensorcircuit/channels.py:55-345, 542-580``): lists of ``Gate`` objects with
concrete (host) parameters, and ``kraus_to_super_gate``."""

from typing import List, Sequence

import numpy as np

from . import cons
from .gates import Gate


def _g(m):
    return Gate(np.asarray(m, dtype=cons.npdtype))


_I = np.eye(2)
_X = np.array([[0, 1], [1, 0]])
_Y = np.array([[0, -1j], [1j, 0]])
_Z = np.array([[1, 0], [0, -1]])


def depolarizingchannel(px: float, py: float, pz: float) -> List[Gate]:
    """reference channels.py:55-106."""
    return [_g(np.sqrt(1 - px - py - pz) * _I), _g(np.sqrt(px) * _X), _g(np.sqrt(py) * _Y), _g(np.sqrt(pz) * _Z)]


def isotropicdepolarizingchannel(p: float, num_qubits: int = 1) -> List[Gate]:
    """reference channels.py:103-136: (1 - p) rho + p / (4^n - 1) sum_j P_j rho P_j."""
    return generaldepolarizingchannel(float(p) / (4 ** num_qubits - 1), num_qubits)


def generaldepolarizingchannel(p, num_qubits: int = 1) -> List[Gate]:
    """reference channels.py:139-230: sqrt(prob_j) P_j over the n-qubit Pauli strings in lexicographic (I, X, Y, Z)
    order, first factor = first qubit; ``p`` a float (every non-identity string) or the 4^n - 1 probabilities."""
    m = 4 ** num_qubits - 1
    if np.ndim(p) == 0:
        probs = [1 - m * float(p)] + m * [float(p)]
    else:
        if len(p) != m:
            raise ValueError(f"Invalid probability input {p}")
        probs = [1 - float(sum(p))] + [float(x) for x in p]
    if not np.all(np.array(probs) >= 0):
        raise ValueError(f"Invalid probability input {p}")
    paulis = [_I, _X, _Y, _Z]
    out = []
    for j, pr in enumerate(probs):
        mat = np.ones((1, 1), dtype=np.complex128)
        for q in range(num_qubits):
            mat = np.kron(mat, paulis[(j // 4 ** (num_qubits - 1 - q)) % 4])
        out.append(Gate((np.sqrt(pr) * mat).astype(cons.npdtype).reshape([2] * (2 * num_qubits))))
    return out


def amplitudedampingchannel(gamma: float, p: float) -> List[Gate]:
    """reference channels.py:233-283."""
    g00, g01 = np.array([[1, 0], [0, 0]]), np.array([[0, 1], [0, 0]])
    g10, g11 = np.array([[0, 0], [1, 0]]), np.array([[0, 0], [0, 1]])
    return [_g(np.sqrt(p) * (g00 + np.sqrt(1 - gamma) * g11)), _g(np.sqrt(p) * np.sqrt(gamma) * g01),
            _g(np.sqrt(1 - p) * (np.sqrt(1 - gamma) * g00 + g11)), _g(np.sqrt(1 - p) * np.sqrt(gamma) * g10)]


def resetchannel() -> List[Gate]:
    """reference channels.py:286-310."""
    return [_g([[1, 0], [0, 0]]), _g([[0, 1], [0, 0]])]


def phasedampingchannel(gamma: float) -> List[Gate]:
    """reference channels.py:313-345."""
    g00, g11 = np.array([[1, 0], [0, 0]]), np.array([[0, 0], [0, 1]])
    return [_g(g00 + np.sqrt(1 - gamma) * g11), _g(np.sqrt(gamma) * g11)]


def kraus_to_super_gate(kraus_list: Sequence[Gate]) -> np.ndarray:
    """reference channels.py:542-580: sum_k K_k (x) conj(K_k) as a matrix on (ket, bra) indices."""
    out = None
    for k in kraus_list:
        m = np.asarray(k.tensor if isinstance(k, Gate) else k, dtype=np.complex128)
        d = int(round(np.sqrt(m.size)))
        m = m.reshape(d, d)
        t = np.kron(m, m.conj())
        out = t if out is None else out + t
    return out


def kraus_identity_check(kraus: Sequence[Gate]) -> None:
    """reference channels.py ``kraus_identity_check``: sum_k K_k^dagger K_k = 1."""
    acc = None
    for k in kraus:
        m = np.asarray(k.tensor, dtype=np.complex128)
        d = int(round(np.sqrt(m.size)))
        m = m.reshape(d, d)
        acc = m.conj().T @ m if acc is None else acc + m.conj().T @ m
    np.testing.assert_allclose(acc, np.eye(acc.shape[0]), atol=1e-5)


single_qubit_kraus_identity_check = kraus_identity_check
channels = ["depolarizing", "amplitudedamping", "reset", "phasedamping", "generaldepolarizing", "isotropicdepolarizing"]
