"""Host logic of tcmi.MPSCircuit on CPU.  The device primitives of tcmi/linalg.py are replaced by
torch-CPU stand-ins (test infrastructure only) so that the centre-position bookkeeping, the SWAP / MPO
routes and the truncation plumbing can be compared with the oracle and the reference's KAT
(tests/test_mpscircuit.py:380) without a GPU."""

import numpy as np
import pytest
import torch

import tcmi as tc
from tcmi import linalg as LA
from oracle import mps as omps

from test_oracle_mps import D, N, dense_state, gate_list


def _svd_trunc(mat, max_singular_values=None, max_truncation_err=None, relative=False, absorb=0):
    u, s, vh, rest = omps.svd_trunc(mat.numpy(), max_singular_values, max_truncation_err, relative)
    if absorb == 1:
        u = u * s.reshape(1, -1)
    elif absorb == 2:
        vh = s.reshape(-1, 1) * vh
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return f(u), f(s), f(vh), f(rest)


@pytest.fixture
def cpu_linalg(monkeypatch):
    monkeypatch.setattr(LA, "matmul", lambda a, b: a @ b)
    monkeypatch.setattr(LA, "site_gate", lambda g, t: torch.einsum("ab,lbr->lar", g, t))
    monkeypatch.setattr(LA, "gate_mix", lambda t, g, L, R: torch.einsum(
        "xyab,labr->lxyr", g.reshape(2, 2, 2, 2), t.reshape(L, 2, 2, R)).reshape(-1))
    monkeypatch.setattr(LA, "svd_trunc", _svd_trunc)
    monkeypatch.setattr(LA, "qr", lambda m: tuple(torch.linalg.qr(m)))
    tc.set_dtype("complex128")
    yield
    tc.set_dtype("complex64")


def _run(ops, split=None):
    m = tc.MPSCircuit(N, split=split)
    for g, idx in ops:
        m.apply(tc.gates.Gate(np.asarray(g)), *idx)
    return m


def test_truncation_kat_host(cpu_linalg):
    ops = gate_list()
    w_c = dense_state(ops)
    m = _run(ops, tc.cons.split_rules(max_singular_values=D))
    np.testing.assert_allclose(float(m.get_norm()), float(torch.linalg.vector_norm(m.wavefunction())), atol=1e-12)
    m.normalize()
    real_fid = abs(np.vdot(m.wavefunction().numpy(), w_c)) ** 2
    np.testing.assert_allclose(real_fid, 0.902663090851, atol=1e-5)
    np.testing.assert_allclose(float(m._fidelity), 0.910305380327, atol=1e-5)
    assert float(m._mps.check_canonical()) < 1e-12
    assert max(m.get_bond_dimensions()) <= D


def test_exact_route_matches_oracle(cpu_linalg):
    ops = gate_list()
    w_c = dense_state(ops)
    m = _run(ops)
    o = omps.MPSCircuit(N)
    for g, idx in ops:
        o.apply(g, *idx)
    np.testing.assert_allclose(m.wavefunction().numpy(), w_c, atol=1e-10)
    assert m.get_bond_dimensions() == o.get_bond_dimensions()
    assert m.get_center_position() == o.get_center_position()
    s = "01" * (N // 2)
    np.testing.assert_allclose(complex(m.amplitude(s)), w_c[int(s, 2)], atol=1e-12)
    e = m.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4])
    np.testing.assert_allclose(complex(e), o.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4]), atol=1e-10)
    t = (np.sin(np.arange(16)) + np.cos(np.arange(16)) * 1j).reshape(2, 2, 2, 2)
    e2 = m.expectation((tc.gates.z(), [3]), (tc.gates.Gate(t), [2, 6]), (tc.gates.toffoli(), [7, 1, 5]))
    from oracle import gates as OG
    want = o.expectation((OG.Z, [3]), (t, [2, 6]), (OG.TOFFOLI.reshape((2,) * 6), [7, 1, 5]))
    np.testing.assert_allclose(complex(e2), want, atol=1e-10)


def test_named_gates_and_errors(cpu_linalg):
    m = tc.MPSCircuit(4)
    m.H(0)
    m.cnot(0, 3)
    m.rx(2, theta=0.3)
    m.rzz(1, 2, theta=0.7)
    c = omps.MPSCircuit(4)
    c.H(0)
    c.cnot(0, 3)
    c.rx(2, theta=0.3)
    c.rzz(1, 2, theta=0.7)
    np.testing.assert_allclose(m.wavefunction().numpy(), c.wavefunction(), atol=1e-12)
    assert m.is_valid()
    with pytest.raises(ValueError, match="duplicate qubits"):
        m.cnot(1, 1)
    with pytest.raises(ValueError, match="must be adjacent"):
        m.apply_adjacent_double_gate(tc.gates.cnot(), 0, 2)
    with pytest.raises(NotImplementedError):
        m.apply_general_gate(tc.gates.x(), 0, mpo=True)


def test_from_wavefunction_and_proj(cpu_linalg):
    w = np.abs(np.sin(np.arange(2 ** N) % np.exp(1))).astype(np.complex128)
    w /= np.linalg.norm(w)
    exact = tc.MPSCircuit(N, wavefunction=w)
    np.testing.assert_allclose(exact.wavefunction().numpy(), w, atol=1e-7)
    tr = tc.MPSCircuit(N, wavefunction=w, split=tc.cons.split_rules(max_singular_values=D))
    m = _run(gate_list(), tc.cons.split_rules(max_singular_values=D))
    proj = complex(m.proj_with_mps(tr))
    np.testing.assert_allclose(proj, np.vdot(tr.wavefunction().numpy(), m.wavefunction().numpy()), atol=1e-12)
