"""Pins oracle/mps.py against the reference's own MPS known-answer test
(reference tests/test_mpscircuit.py:19-60,109-131,380: N=8, D=6 truncation fidelities) and against
the dense oracle (untruncated MPS == state vector)."""

import numpy as np
import scipy.linalg

from oracle import dense, mps as omps
from oracle import gates as G

N, D = 8, 6


def reproducible_unitary(n, param):
    # reference tests/test_mpscircuit.py:27-35
    e = 2 ** n
    a = np.arange(e * e).reshape(e, e).astype(np.complex128)
    a = a + np.sin(a) * param * 1j
    a = a - a.conj().T
    return scipy.linalg.expm(a).reshape((2,) * (2 * n))


def gate_list(params=(1.0, 1.0, 1.0)):
    # reference tests/test_mpscircuit.py:38-60 (simulate)
    o1, o2, o3 = (reproducible_unitary(k + 1, params[k]) for k in range(3))
    ops = [(G.H, (0,))]
    for i in range(0, N - 1, 2):
        ops.append((o2, (i, i + 1)))
        ops.append((o1, (i,)))
    ops.append((o3, (int(N * 0.1), int(N * 0.5), int(N * 0.9))))
    ops.append((o2, (1, N - 2)))
    ops.append((G.CZ, (2, 3)))
    return ops


def dense_state(ops):
    psi = np.zeros(2 ** N, dtype=np.complex128)
    psi[0] = 1
    for g, idx in ops:
        psi = dense.apply_gate(psi, N, np.asarray(g).reshape(2 ** len(idx), -1), list(idx))
    return psi


def run_mps(ops, split=None):
    m = omps.MPSCircuit(N, split=split)
    for k, (g, idx) in enumerate(ops):
        if k == len(ops) - 1:
            assert abs(m._mps.check_canonical()) < 1e-12
        m.apply(g, *idx)
    return m


def test_truncation_kat():
    ops = gate_list()
    w_c = dense_state(ops)
    m = run_mps(ops, omps.split_rules(max_singular_values=D))
    np.testing.assert_allclose(m.get_norm(), np.linalg.norm(m.wavefunction()), atol=1e-12)
    m.normalize()
    real_fid = abs(np.vdot(m.wavefunction(), w_c)) ** 2
    np.testing.assert_allclose(real_fid, 0.902663090851, atol=1e-5)
    np.testing.assert_allclose(m._fidelity, 0.910305380327, atol=1e-5)
    assert abs(m._mps.check_canonical()) < 1e-12
    assert max(m.get_bond_dimensions()) <= D


def test_exact_matches_dense_and_outputs():
    ops = gate_list()
    w_c = dense_state(ops)
    m = run_mps(ops)
    np.testing.assert_allclose(m.wavefunction(), w_c, atol=1e-10)
    assert abs(m._mps.check_canonical()) < 1e-12
    s = "01" * (N // 2)
    np.testing.assert_allclose(m.amplitude(s), w_c[int(s, 2)], atol=1e-12)
    # reference tests/test_mpscircuit.py:152-181
    t = (np.sin(np.arange(16)) + np.cos(np.arange(16)) * 1j).reshape(2, 2, 2, 2)
    e_mps = m.expectation((G.Z, [3]), (t, [2, 6]), (G.TOFFOLI.reshape((2,) * 6), [7, 1, 5]))
    phi = w_c.copy()
    for g, idx in ((G.Z, [3]), (t.reshape(4, 4), [2, 6]), (G.TOFFOLI, [7, 1, 5])):
        phi = dense.apply_gate(phi, N, np.asarray(g).reshape(2 ** len(idx), -1), idx)
    np.testing.assert_allclose(e_mps, np.vdot(w_c, phi), atol=1e-7)
    e_ps = m.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4])
    ref = dense.pauli_string_expectation(w_c, N, [1, 2, 1, 2, 3, 2, 3, 0])
    np.testing.assert_allclose(e_ps, ref, atol=1e-7)


def test_from_wavefunction_kat():
    # reference tests/test_mpscircuit.py:184-214 (external wavefunction, relative error 0.276089)
    w = np.abs(np.sin(np.arange(2 ** N) % np.exp(1))).astype(np.complex128)
    w /= np.linalg.norm(w)
    exact = omps.MPSCircuit(N, wavefunction=w)
    np.testing.assert_allclose(exact.wavefunction(), w, atol=1e-7)
    tr = omps.MPSCircuit(N, wavefunction=w, split=omps.split_rules(max_singular_values=D))
    real_fid = abs(np.vdot(tr.wavefunction(), w)) ** 2
    sv = np.linalg.svd(w.reshape(2 ** (N // 2), 2 ** (N // 2)))[1]
    upper = np.sum(sv[0:D] ** 2)
    np.testing.assert_allclose(np.log((1 - real_fid) / (1 - upper)), 0.276089, atol=1e-4)
