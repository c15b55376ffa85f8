"""Canonical workloads of SURVEY.md section 8 (test infrastructure, see oracle/__init__.py).

Each builder drives any object with the reference ``Circuit`` gate-method surface
(``h / rx / cx / exp1``), so the same function builds the oracle circuit and the product
circuit -- exactly how the reference's harnesses are written.
"""

import numpy as np

from . import gates as G


def hea_b(c, n, nlayers, params, zz=None):
    """``templates/blocks.py:146-185`` ``example_block`` (= benchmarks/scripts/vqe_tc.py:116-127):
    H on all; per layer exp1(ZZ, theta=p[2j, i]) ladder then rx(p[2j+1, i]).  params [2d, n]."""
    if zz is None:
        zz = G.ZZ
    for i in range(n):
        c.h(i)
    for j in range(nlayers):
        for i in range(n - 1):
            c.exp1(i, i + 1, unitary=zz, theta=params[2 * j, i])
        for i in range(n):
            c.rx(i, theta=params[2 * j + 1, i])
    return c


def hea_a(c, n, nlayers, params):
    """``benchmarks/scripts_v2/benchmark_core.py:6-14``: H; per layer rx then cx ladder.
    params [d, n]."""
    for i in range(n):
        c.h(i)
    for j in range(nlayers):
        for i in range(n):
            c.rx(i, theta=params[j, i])
        for i in range(n - 1):
            c.cx(i, i + 1)
    return c


def hea_b_ops(n, nlayers, params):
    """The same circuit as a flat ``(matrix, qubits)`` list for ``oracle.dense.run``."""
    ops = [(G.H, [i]) for i in range(n)]
    for j in range(nlayers):
        for i in range(n - 1):
            ops.append((G.exp1(G.ZZ, params[2 * j, i]), [i, i + 1]))
        for i in range(n):
            ops.append((G.rx(params[2 * j + 1, i]), [i]))
    return ops


def hea_a_ops(n, nlayers, params):
    ops = [(G.H, [i]) for i in range(n)]
    for j in range(nlayers):
        for i in range(n):
            ops.append((G.rx(params[j, i]), [i]))
        for i in range(n - 1):
            ops.append((G.CNOT, [i, i + 1]))
    return ops


def tfim_terms(n, j=1.0, h=-1.0):
    """``benchmarks/scripts/vqe_tc.py:75-81``: E = sum_i h<X_i> + sum_{i<n-1} j<Z_i Z_{i+1}>.
    Returned as (weight, pauli-string list with 0 I, 1 X, 2 Y, 3 Z)."""
    terms = []
    for i in range(n):
        ps = [0] * n
        ps[i] = 1
        terms.append((h, ps))
    for i in range(n - 1):
        ps = [0] * n
        ps[i] = 3
        ps[i + 1] = 3
        terms.append((j, ps))
    return terms


def tfim_energy(c, n, j=1.0, h=-1.0):
    """The reference's python loop over 2n-1 ``c.expectation`` calls (vqe_tc.py:75-81)."""
    e = 0.0
    for i in range(n):
        e += h * c.expectation((G.X, [i]))
    for i in range(n - 1):
        e += j * c.expectation((G.Z, [i]), (G.Z, [i + 1]))
    return e


def tfim_energy_dense(psi, n, j=1.0, h=-1.0):
    """Same energy from a flat state with bit tricks (independent of the TN path)."""
    p = np.abs(psi.astype(np.complex128)) ** 2
    idx = np.arange(psi.size)
    e = 0.0
    for i in range(n - 1):
        b0 = (idx >> (n - 1 - i)) & 1
        b1 = (idx >> (n - 2 - i)) & 1
        e += j * np.sum(p * (1 - 2 * (b0 ^ b1)))
    psi = psi.astype(np.complex128)
    for i in range(n):
        flipped = psi.reshape(-1)[idx ^ (1 << (n - 1 - i))]
        e += h * np.real(np.vdot(psi, flipped))
    return float(e)


def config_params(config):
    """Seeded parameter sets of SURVEY.md section 8(d)."""
    if config == 1:
        return 10, 4, np.ones([8, 10])
    if config == "1r":
        return 10, 4, np.random.default_rng(0).normal(0, 1, [8, 10])
    if config == 2:
        return 24, 8, np.random.default_rng(24).uniform(0, 2 * np.pi, [16, 24]).astype(np.float32)
    if config == 3:
        return 28, 12, np.random.default_rng(28).normal(0, 0.1, [32, 24, 28]).astype(np.float32)
    raise ValueError(config)
