"""Kraus operators of the one-qubit noise channels (TEST INFRASTRUCTURE, see oracle/__init__.py), restated from
the formulas of the reference so that the density-matrix tests do not feed the product's own operators to the
oracle: reference tensorcircuit/channels.py:55-100 (depolarizing), :233-283 (generalised amplitude damping),
:286-310 (reset), :313-341 (phase damping)."""

import numpy as np

_I = np.eye(2, dtype=np.complex128)
_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)


def depolarizing(px, py, pz):
    """channels.py:95-100: sqrt(1 - px - py - pz) I, sqrt(px) X, sqrt(py) Y, sqrt(pz) Z."""
    return [np.sqrt(1 - px - py - pz) * _I, np.sqrt(px) * _X, np.sqrt(py) * _Y, np.sqrt(pz) * _Z]


def amplitudedamping(gamma, p):
    """channels.py:233-283: sqrt(p) diag(1, sqrt(1 - gamma)), sqrt(p) sqrt(gamma) |0><1|,
    sqrt(1 - p) diag(sqrt(1 - gamma), 1), sqrt(1 - p) sqrt(gamma) |1><0|."""
    sg, s1 = np.sqrt(gamma), np.sqrt(1 - gamma)
    return [
        np.sqrt(p) * np.array([[1, 0], [0, s1]], dtype=np.complex128),
        np.sqrt(p) * np.array([[0, sg], [0, 0]], dtype=np.complex128),
        np.sqrt(1 - p) * np.array([[s1, 0], [0, 1]], dtype=np.complex128),
        np.sqrt(1 - p) * np.array([[0, 0], [sg, 0]], dtype=np.complex128),
    ]


def phasedamping(gamma):
    """channels.py:337-341: diag(1, sqrt(1 - gamma)), diag(0, sqrt(gamma))."""
    return [np.array([[1, 0], [0, np.sqrt(1 - gamma)]], dtype=np.complex128),
            np.array([[0, 0], [0, np.sqrt(gamma)]], dtype=np.complex128)]


def reset():
    """channels.py:308-310: |0><0|, |0><1|."""
    return [np.array([[1, 0], [0, 0]], dtype=np.complex128), np.array([[0, 1], [0, 0]], dtype=np.complex128)]


def generaldepolarizing(p, num_qubits=1):
    """reference channels.py:139-230: Kraus operators sqrt(prob_j) P_j, P_j the n-qubit Pauli strings in the order of
    itertools.product(I, X, Y, Z) with the first factor on the first (most significant) qubit; ``p`` scalar: every
    non-identity string has probability p, identity 1 - (4^n - 1) p."""
    import itertools

    m = 4 ** num_qubits - 1
    probs = [1 - m * p] + [p] * m if np.ndim(p) == 0 else [1 - sum(p)] + list(p)
    ks = []
    for pr, fac in zip(probs, itertools.product([_I, _X, _Y, _Z], repeat=num_qubits)):
        mat = np.array([[1.0 + 0j]])
        for f in fac:
            mat = np.kron(mat, f)
        ks.append(np.sqrt(pr) * mat)
    return ks


def isotropicdepolarizing(p, num_qubits=1):
    """reference channels.py:103-136."""
    return generaldepolarizing(p / (4 ** num_qubits - 1), num_qubits)
