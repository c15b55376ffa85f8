"""Reverse-mode value_and_grad of the TFIM energy of an HEA-B circuit on a dense state vector (test infrastructure, see
oracle/__init__.py; used by tests and by bench.py's ``cpu_baseline`` legs only).

What it stands for: the reference's VQE step ``K.jit(K.vvag(energy))`` (benchmarks/scripts/vqe_tc.py:107-141,
backends/abstract_backend.py:2262-2293) on the CPU backend.  The reference differentiates through its tensordot chain
with the framework's tape; this file is the *adjoint state-vector method* in plain numpy -- one forward run, then one sweep
back through the gates carrying ``psi`` (un-computed) and ``lambda = H psi`` (pulled back), with
``dE/dtheta_k = 2 c Im <lambda_k| G_k |psi_k>`` for a gate ``exp(-i c theta G)``.  It does the least arithmetic a CPU
implementation of the same step can do (three state updates and one inner product per gate; the tape-based reference does
more), so as a CPU baseline it flatters the CPU.  Checked against central differences of ``oracle.dense`` in
tests/test_oracle_kat.py.

Conventions (oracle/gates.py, oracle/workloads.py:13-26 = templates/blocks.py:146-185): qubit 0 = most significant bit;
``exp1(ZZ, theta) = exp(-i theta ZZ)`` (gates.py:920-953), ``rx(theta) = exp(-i theta X / 2)`` (gates.py:692-707);
params ``[2 d, n]``: row ``2 j`` = the ZZ ladder of layer j (n - 1 angles used), row ``2 j + 1`` = its rx angles.
"""

import numpy as np


def _rx_inplace(psi, n, q, theta):
    """psi <- rx(theta) on qubit q (theta < 0: the inverse)."""
    v = psi.reshape(2**q, 2, 2 ** (n - 1 - q))
    c, s = np.cos(theta / 2.0), -1j * np.sin(theta / 2.0)
    a0 = v[:, 0, :].copy()
    v[:, 0, :] *= c
    v[:, 0, :] += s * v[:, 1, :]
    v[:, 1, :] *= c
    v[:, 1, :] += s * a0


def _zz_inplace(psi, n, q, theta):
    """psi <- exp(-i theta Z_q Z_{q+1}) psi."""
    v = psi.reshape(2**q, 2, 2, 2 ** (n - 2 - q))
    same, diff = np.exp(-1j * theta), np.exp(1j * theta)
    v[:, 0, 0, :] *= same
    v[:, 1, 1, :] *= same
    v[:, 0, 1, :] *= diff
    v[:, 1, 0, :] *= diff


def _im_x(lam, psi, n, q):
    """Im <lam| X_q |psi>."""
    l_ = lam.reshape(2**q, 2, 2 ** (n - 1 - q))
    p_ = psi.reshape(2**q, 2, 2 ** (n - 1 - q))
    return (np.vdot(l_[:, 0, :], p_[:, 1, :]) + np.vdot(l_[:, 1, :], p_[:, 0, :])).imag


def _im_zz(lam, psi, n, q):
    """Im <lam| Z_q Z_{q+1} |psi>."""
    l_ = lam.reshape(2**q, 2, 2, 2 ** (n - 2 - q))
    p_ = psi.reshape(2**q, 2, 2, 2 ** (n - 2 - q))
    return (np.vdot(l_[:, 0, 0, :], p_[:, 0, 0, :]) + np.vdot(l_[:, 1, 1, :], p_[:, 1, 1, :])
            - np.vdot(l_[:, 0, 1, :], p_[:, 0, 1, :]) - np.vdot(l_[:, 1, 0, :], p_[:, 1, 0, :])).imag


def tfim_apply(psi, n, j=1.0, h=-1.0):
    """(sum_i h X_i + sum_i j Z_i Z_{i+1}) |psi> (benchmarks/scripts/vqe_tc.py:75-81)."""
    out = np.zeros_like(psi)
    for q in range(n):
        v = psi.reshape(2**q, 2, 2 ** (n - 1 - q))
        o = out.reshape(2**q, 2, 2 ** (n - 1 - q))
        o[:, 0, :] += h * v[:, 1, :]
        o[:, 1, :] += h * v[:, 0, :]
    for q in range(n - 1):
        v = psi.reshape(2**q, 2, 2, 2 ** (n - 2 - q))
        o = out.reshape(2**q, 2, 2, 2 ** (n - 2 - q))
        o[:, 0, 0, :] += j * v[:, 0, 0, :]
        o[:, 1, 1, :] += j * v[:, 1, 1, :]
        o[:, 0, 1, :] -= j * v[:, 0, 1, :]
        o[:, 1, 0, :] -= j * v[:, 1, 0, :]
    return out


def hea_b_state(n, nlayers, params, dtype=np.complex128):
    psi = np.full(2**n, 1.0 / np.sqrt(2.0**n), dtype=dtype)        # H on every qubit of |0...0>
    for j in range(nlayers):
        for i in range(n - 1):
            _zz_inplace(psi, n, i, float(params[2 * j, i]))
        for i in range(n):
            _rx_inplace(psi, n, i, float(params[2 * j + 1, i]))
    return psi


def hea_b_tfim_value_and_grad(n, nlayers, params, j=1.0, h=-1.0, dtype=np.complex128):
    """(E, dE/dparams) with params [2 nlayers, n]; the unused last ZZ angle of a row has gradient 0."""
    params = np.asarray(params, dtype=np.float64)
    psi = hea_b_state(n, nlayers, params, dtype)
    lam = tfim_apply(psi, n, j, h)
    e = float(np.vdot(psi, lam).real)
    g = np.zeros_like(params)
    for jl in reversed(range(nlayers)):
        for i in reversed(range(n)):
            th = float(params[2 * jl + 1, i])
            g[2 * jl + 1, i] = _im_x(lam, psi, n, i)                 # 2 * (1/2) * Im <lam| X |psi>
            _rx_inplace(psi, n, i, -th)
            _rx_inplace(lam, n, i, -th)
        for i in reversed(range(n - 1)):
            th = float(params[2 * jl, i])
            g[2 * jl, i] = 2.0 * _im_zz(lam, psi, n, i)
            _zz_inplace(psi, n, i, -th)
            _zz_inplace(lam, n, i, -th)
    return e, g
