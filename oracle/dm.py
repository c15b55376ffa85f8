"""Dense density-matrix oracle (TEST INFRASTRUCTURE ONLY): rho -> U rho U^dagger and rho -> sum_k K rho K^dagger
on a 2^n x 2^n numpy matrix, the semantics of reference ``tensorcircuit/densitymatrix.py:222-300``."""

import numpy as np


def _embed(m, qubits, n):
    """Full 2^n x 2^n matrix of ``m`` acting on ``qubits`` (qubit 0 = most significant)."""
    k = len(qubits)
    t = np.asarray(m, dtype=np.complex128).reshape([2] * (2 * k))
    full = np.eye(2 ** n, dtype=np.complex128).reshape([2] * (2 * n))
    full = np.tensordot(t, full, axes=(list(range(k, 2 * k)), list(qubits)))
    full = np.moveaxis(full, list(range(k)), list(qubits))
    return full.reshape(2 ** n, 2 ** n)


def run(n, ops):
    """ops: ("u", matrix, qubits) | ("k", [kraus matrices], qubits)."""
    rho = np.zeros((2 ** n, 2 ** n), dtype=np.complex128)
    rho[0, 0] = 1
    for kind, m, qs in ops:
        if kind == "u":
            u = _embed(m, qs, n)
            rho = u @ rho @ u.conj().T
        else:
            rho = sum(_embed(k, qs, n) @ rho @ _embed(k, qs, n).conj().T for k in m)
    return rho


def expectation(rho, n, *ops):
    m = np.eye(2 ** n, dtype=np.complex128)
    for o, qs in ops:
        m = m @ _embed(o, qs, n)
    return np.trace(rho @ m)
