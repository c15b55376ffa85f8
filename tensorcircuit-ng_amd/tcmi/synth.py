"""Exact plan-time synthesis of dense k-qubit gates (k >= 3) into the operations the tile-VM executes
natively: dense gates on <= 2 qubits and diagonal gates on any number of qubits.

The reference contracts a rank-2k gate tensor like any other node (``basecircuit.py:183-371``); the HIP
executor applies gates on register bits, so a dense 8x8 (toffoli, fredkin, ``any`` on three qubits ...)
is rewritten once, on the host, by the quantum Shannon decomposition (cosine-sine decomposition +
demultiplexing): for U on qubits (a, rest)

    U = blockdiag(A1, A2) . [[C, -S], [S, C]] . blockdiag(B1, B2)
    blockdiag(X1, X2) = (1 (x) V) . diag(D, D^*) . (1 (x) W),   X1 X2^H = V D^2 V^H,  W = D V^H X2
    [[C, -S], [S, C]] = (G (x) 1) . diag(e^{-i theta_j}, e^{+i theta_j}) . (G^H (x) 1),   G = S.H

Every identity is exact (no dropped global phase), so amplitudes keep parity with the dense gate.
"""

from typing import List, Sequence, Tuple

import numpy as np
import scipy.linalg

_H = np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2)
_S = np.array([[1, 0], [0, 1j]], dtype=np.complex128)
_G = _S @ _H

Op = Tuple[np.ndarray, Tuple[int, ...]]


def _demultiplex(x1: np.ndarray, x2: np.ndarray):
    """blockdiag(x1, x2) = (1 (x) v) diag(d, conj d) (1 (x) w)."""
    t, z = scipy.linalg.schur(x1 @ x2.conj().T, output="complex")
    d2 = np.diag(t)
    d2 = d2 / np.abs(d2)
    d = np.sqrt(d2)
    w = (d[:, None] * z.conj().T) @ x2
    return z, d, w


def decompose_dense(u: np.ndarray, qubits: Sequence[int]) -> List[Op]:
    """Ops ``(matrix, qubits)`` in application order whose product is ``u`` on ``qubits`` (first qubit =
    most significant index bit): dense matrices on <= 2 qubits, diagonal matrices on any number."""
    qubits = tuple(int(q) for q in qubits)
    k = len(qubits)
    u = np.asarray(u, dtype=np.complex128).reshape(2**k, 2**k)
    if k <= 2:
        return [(u, qubits)]
    if np.abs(u - np.diag(np.diag(u))).max() < 1e-14:
        return [(u, qubits)]
    h = 2 ** (k - 1)
    (a1, a2), theta, (b1, b2) = scipy.linalg.cossin(u, p=h, q=h, separate=True)
    rest = qubits[1:]
    ops: List[Op] = []

    def multiplexed(x1, x2):
        v, d, w = _demultiplex(x1, x2)
        out = decompose_dense(w, rest)
        out.append((np.diag(np.concatenate([d, d.conj()])), qubits))
        out += decompose_dense(v, rest)
        return out

    ops += multiplexed(b1, b2)
    ops.append((_G.conj().T, qubits[:1]))
    ops.append((np.diag(np.concatenate([np.exp(-1j * theta), np.exp(1j * theta)])), qubits))
    ops.append((_G, qubits[:1]))
    ops += multiplexed(a1, a2)
    return ops


_CNOT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.complex128)


def lower_diagonal(d: np.ndarray, qubits: Sequence[int]) -> List[Op]:
    """A unit-modulus diagonal on k >= 3 qubits as diagonals on <= 2 qubits plus CNOT ladders.  The phases
    are expanded in Z-products (Walsh-Hadamard): phi(x) = sum_S c_S prod_{i in S} z_i.  Terms of order <= 2
    become two-qubit diagonals; a term of order m >= 3 uses z_t -> z_{s2} .. z_{s(m-1)} z_t under CNOTs
    into the last qubit, which turns it into the two-body term z_{s1} z_t (exact, no dropped phase)."""
    qubits = tuple(int(q) for q in qubits)
    k = len(qubits)
    dd = np.asarray(d, dtype=np.complex128)
    dd = np.diag(dd) if dd.ndim == 2 else dd
    if k <= 2:
        return [(np.diag(dd), qubits)]
    phi = np.angle(dd)
    z = np.array([[1 - 2 * ((x >> (k - 1 - i)) & 1) for i in range(k)] for x in range(2**k)], dtype=np.float64)
    coef = {}
    for mask in range(2**k):
        sub = [i for i in range(k) if (mask >> i) & 1]
        prod = np.prod(z[:, sub], axis=1) if sub else np.ones(2**k)
        c = float(np.dot(phi, prod)) / 2**k
        if abs(c) > 1e-15:
            coef[tuple(sub)] = c
    ops: List[Op] = []
    zz = np.array([1, -1, -1, 1], dtype=np.float64)
    z0 = np.array([1, 1, -1, -1], dtype=np.float64)
    z1 = np.array([1, -1, 1, -1], dtype=np.float64)
    # orders 0..2: one two-qubit diagonal per pair; constants and singles ride on pairs (0, j)
    for i in range(k):
        for j in range(i + 1, k):
            ph = coef.get((i, j), 0.0) * zz
            if i == 0 and j == 1:
                ph = ph + coef.get((), 0.0) + coef.get((0,), 0.0) * z0 + coef.get((1,), 0.0) * z1
            elif i == 0:
                ph = ph + coef.get((j,), 0.0) * z1
            if np.abs(ph).max() > 0:
                ops.append((np.diag(np.exp(1j * ph)), (qubits[i], qubits[j])))
    for sub, c in coef.items():
        if len(sub) < 3:
            continue
        t = sub[-1]
        ladder = [(_CNOT, (qubits[sidx], qubits[t])) for sidx in sub[1:-1]]
        ops += ladder
        ops.append((np.diag(np.exp(1j * c * zz)), (qubits[sub[0]], qubits[t])))
        ops += ladder[::-1]
    return ops


def lower(ops: Sequence[Op]) -> List[Op]:
    out: List[Op] = []
    for m, qs in ops:
        if len(qs) > 2:
            out += lower_diagonal(m, qs)
        else:
            out.append((m, qs))
    return out


def expand(ops: Sequence[Op], qubits: Sequence[int]) -> np.ndarray:
    """Dense product of ``ops`` on the ordered ``qubits`` (test helper / self-check)."""
    qubits = list(qubits)
    k = len(qubits)
    full = np.eye(2**k, dtype=np.complex128)
    for m, qs in ops:
        j = len(qs)
        pos = [qubits.index(q) for q in qs]
        t = np.asarray(m, dtype=np.complex128).reshape([2] * (2 * j))
        f = full.reshape([2] * k + [2**k])
        f = np.tensordot(t, f, axes=(list(range(j, 2 * j)), pos))
        f = np.moveaxis(f, list(range(j)), pos)
        full = f.reshape(2**k, 2**k)
    return full
