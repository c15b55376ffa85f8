"""Expectation values on the hip backend: every operator product is expanded in Pauli strings and
evaluated by the fused measurement passes (``executor.CompiledMeasure``).

Reference semantics: ``Circuit.expectation(*ops)`` = complex scalar <psi| prod_k O_k |psi> with the
state contracted once and cached when ``reuse=True`` (``tensorcircuit/basecircuit.py:375-447``,
``tensorcircuit/circuit.py:833-913``).
"""

import itertools
from typing import List, Sequence, Tuple

import numpy as np

from . import cons
from . import gates as G

_PAULIS = [G._i_matrix, G._x_matrix, G._y_matrix, G._z_matrix]


def pauli_decompose(matrix: np.ndarray, tol: float = 1e-14):
    """k-qubit matrix -> list of (pauli codes tuple, coefficient) with M = sum c * P."""
    m = np.asarray(matrix, dtype=np.complex128)
    d = m.shape[0]
    k = int(round(np.log2(d)))
    out = []
    for codes in itertools.product(range(4), repeat=k):
        p = np.eye(1, dtype=np.complex128)
        for c in codes:
            p = np.kron(p, _PAULIS[c])
        coef = np.trace(p.conj().T @ m) / d
        if abs(coef) > tol:
            out.append((codes, coef))
    return out


def ops_to_pauli_sum(n: int, ops: Sequence[Tuple[np.ndarray, Tuple[int, ...]]]):
    """Product of operators on disjoint qubits -> (list of length-n pauli strings, coefficients)."""
    parts = []
    for m, index in ops:
        parts.append([(codes, coef, index) for codes, coef in pauli_decompose(m)])
    strings, coefs = [], []
    for combo in itertools.product(*parts) if parts else [()]:
        ps = [0] * n
        c = 1.0 + 0.0j
        for codes, coef, index in combo:
            c *= coef
            for q, code in zip(index, codes):
                ps[q] = code
        strings.append(tuple(ps))
        coefs.append(c)
    return strings, np.array(coefs, dtype=np.complex128)


def _circuit_full_state(circuit):
    """[B, 2^n_exec] executor buffer of the circuit's state, cached on the circuit (reuse=True,
    reference basecircuit.py:375-391; any later gate resets ``state_tensor``)."""
    from .functional import circuit_state_full

    st = getattr(circuit, "state_tensor", None)
    if st is None:
        st = circuit_state_full(circuit)
        circuit.state_tensor = st
    return st


def pauli_sum_values(circuit, strings):
    """<psi|P_t|psi> for every string, as a complex128 tensor [nterms] (or [B, nterms] under vmap)."""
    from .functional import circuit_pauli_values

    return circuit_pauli_values(circuit, tuple(tuple(int(p) for p in s) for s in strings))


def expectation_of_ops(circuit, ops):
    n = circuit._nqubits
    strings, coefs = ops_to_pauli_sum(n, ops)
    vals = pauli_sum_values(circuit, strings)
    import torch

    w = torch.as_tensor(coefs, device=vals.device)
    out = (vals * w).sum(-1)
    return out.to(getattr(torch, cons.dtypestr))
