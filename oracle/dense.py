"""Independent dense state-vector simulator (test infrastructure, see oracle/__init__.py).

A different algorithm from ``oracle.tn`` (gate-by-gate on a 2^n vector, no network,
no path): used to cross-check the TN restatement and to produce golden vectors.
Conventions: qubit 0 = most significant bit (tests/test_circuit.py:47-53); a k-qubit
matrix ``U[out, in]`` acts on the listed qubits in the listed order
(tensorcircuit/basecircuit.py:288-290).
"""

import numpy as np

from . import gates as G


def zero_state(n, dtype=np.complex128):
    psi = np.zeros(2**n, dtype=dtype)
    psi[0] = 1.0
    return psi


def apply_gate(psi, n, matrix, qubits):
    """Return U_(qubits) |psi>."""
    k = len(qubits)
    qubits = [q if q >= 0 else n + q for q in qubits]
    u = np.asarray(matrix, dtype=psi.dtype).reshape(2**k, 2**k)
    t = psi.reshape([2] * n)
    t = np.moveaxis(t, qubits, range(k))
    shp = t.shape
    t = u @ t.reshape(2**k, -1)
    t = np.moveaxis(t.reshape(shp), range(k), qubits)
    return np.ascontiguousarray(t).reshape(-1)


def apply_gate_inplace(psi, n, matrix, qubits):
    """The same product as ``apply_gate`` for one-qubit gates and for diagonal gates, updating ``psi`` in place through
    strided views (no transposed copies of the state): what makes the n = 28 fixtures of tests/golden/make_golden_full.py
    affordable (a 4 GiB state; ``apply_gate`` moves it three times per gate).  Anything else falls back to ``apply_gate``.
    Returns the updated vector."""
    k = len(qubits)
    qubits = [q if q >= 0 else n + q for q in qubits]
    u = np.asarray(matrix, dtype=psi.dtype).reshape(2**k, 2**k)
    if k == 1:
        q = qubits[0]
        v = psi.reshape(2**q, 2, 2 ** (n - 1 - q))
        x0 = v[:, 0, :].copy()
        v[:, 0, :] *= u[0, 0]
        v[:, 0, :] += u[0, 1] * v[:, 1, :]
        v[:, 1, :] *= u[1, 1]
        v[:, 1, :] += u[1, 0] * x0
        return psi
    if np.abs(u - np.diag(np.diag(u))).max() == 0.0 and len(set(qubits)) == k:
        t = psi.reshape([2] * n)
        for idx in range(2**k):
            sel = [slice(None)] * n
            for j, q in enumerate(qubits):
                sel[q] = (idx >> (k - 1 - j)) & 1
            t[tuple(sel)] *= u[idx, idx]
        return psi
    return apply_gate(psi, n, matrix, qubits)


def run(n, ops, dtype=np.complex128, inputs=None, inplace=False):
    """``ops`` = iterable of ``(matrix, qubits)``.  ``inplace``: use ``apply_gate_inplace`` (large n)."""
    psi = zero_state(n, dtype) if inputs is None else np.asarray(inputs, dtype=dtype).copy()
    for m, qs in ops:
        psi = apply_gate_inplace(psi, n, m, list(qs)) if inplace else apply_gate(psi, n, m, list(qs))
    return psi


def expectation(psi, n, *ops):
    """<psi| prod ops |psi> (complex scalar; tensorcircuit/basecircuit.py:419-447)."""
    phi = psi
    for m, qs in ops:
        if isinstance(qs, int):
            qs = [qs]
        phi = apply_gate(phi, n, m, list(qs))
    return np.vdot(psi, phi)


def pauli_string_expectation(psi, n, ps):
    """ps[i] in {0,1,2,3} = I,X,Y,Z on qubit i."""
    ops = [(G.PAULI[p], [i]) for i, p in enumerate(ps) if p]
    return expectation(psi, n, *ops)


def amplitude(psi, n, bits):
    if isinstance(bits, str):
        bits = [int(c) for c in bits]
    idx = 0
    for b in bits:
        idx = (idx << 1) | int(b)
    return psi[idx]
