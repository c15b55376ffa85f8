"""DMCircuit (reference tensorcircuit/densitymatrix.py) on the doubled state-vector plan:
density matrix, expectations and a gradient against the dense numpy oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import channels as OC, dm as odm, gates as G  # noqa: E402


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(request.param)
    yield tc
    tc.set_dtype("complex64")


def _build(tc, n, theta, ops=None):
    c = tc.DMCircuit(n)
    rec = (lambda *a: ops.append(a)) if ops is not None else (lambda *a: None)
    for i in range(n):
        c.h(i); rec("u", G.H, [i])
    c.cnot(0, 1); rec("u", G.CNOT, [0, 1])
    c.rx(2, theta=theta); rec("u", G.rx(float(theta)), [2])
    c.depolarizing(1, px=0.1, py=0.05, pz=0.02); rec("k", OC.depolarizing(0.1, 0.05, 0.02), [1])
    c.rzz(1, 2, theta=0.4); rec("u", G.rzz(0.4), [1, 2])
    c.amplitudedamping(0, gamma=0.3, p=0.9); rec("k", OC.amplitudedamping(0.3, 0.9), [0])
    c.toffoli(0, 2, 3); rec("u", G.TOFFOLI, [0, 2, 3])
    c.phasedamping(3, gamma=0.25); rec("k", OC.phasedamping(0.25), [3])
    c.ry(3, theta=0.7); rec("u", G.ry(0.7), [3])
    c.reset(4) if n > 4 else None
    if n > 4:
        rec("k", OC.reset(), [4])
    return c


def test_density_matrix_and_expectations(tcd):
    tc = tcd
    n, theta = 5, 0.37
    ops = []
    c = _build(tc, n, theta, ops)
    rho = odm.run(n, ops)
    got = tc.backend.numpy(c.densitymatrix())
    tol = 2e-6 if tc.dtypestr == "complex64" else 1e-12
    np.testing.assert_allclose(got, rho, atol=tol)
    c.check_density_matrix(c.densitymatrix())
    e1 = complex(c.expectation((tc.gates.z(), [0])))
    np.testing.assert_allclose(e1, odm.expectation(rho, n, (G.Z, [0])), atol=tol * 10)
    e2 = complex(c.expectation_ps(x=[1], z=[2]))
    np.testing.assert_allclose(e2, odm.expectation(rho, n, (G.X, [1]), (G.Z, [2])), atol=tol * 10)
    for ks in (tc.channels.depolarizingchannel(0.1, 0.2, 0.3), tc.channels.amplitudedampingchannel(0.4, 0.7),
               tc.channels.phasedampingchannel(0.6), tc.channels.resetchannel()):
        tc.channels.kraus_identity_check(ks)


def test_gradient_through_noisy_circuit(tcd):
    tc = tcd
    n = 4

    def f(t):
        return tc.backend.real(_build(tc, n, t).expectation((tc.gates.z(), [2])))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(np.array(0.37, dtype=rdt)))

    def ref(t):
        o = []
        _build(tc, n, t, o)
        return np.real(odm.expectation(odm.run(n, o), n, (G.Z, [2])))

    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-9
    np.testing.assert_allclose(float(v), ref(0.37), atol=tol)
    eps = 1e-5
    np.testing.assert_allclose(float(g), (ref(0.37 + eps) - ref(0.37 - eps)) / (2 * eps), atol=2e-4 if tc.dtypestr == "complex64" else 1e-6)


def test_gradient_of_a_parameter_before_a_channel(tcd):
    """A parameter in FRONT of non-unitary super-gates with a non-zero true gradient: the adjoint sweep cannot
    un-compute psi through a channel (U^dagger is not the inverse), the executor restarts from checkpoints
    (executor.CompiledCircuit._vjp_segmented).  Finite differences of the dense density-matrix oracle."""
    tc = tcd
    n = 3

    def build(t0, t1, ops=None):
        c = tc.DMCircuit(n)
        rec = (lambda *a: ops.append(a)) if ops is not None else (lambda *a: None)
        c.ry(0, theta=t0); rec("u", G.ry(float(t0)), [0])
        c.h(1); rec("u", G.H, [1])
        c.cnot(0, 1); rec("u", G.CNOT, [0, 1])
        c.amplitudedamping(0, gamma=0.35, p=0.8); rec("k", OC.amplitudedamping(0.35, 0.8), [0])
        c.rx(0, theta=t1); rec("u", G.rx(float(t1)), [0])
        c.depolarizing(1, px=0.05, py=0.1, pz=0.15); rec("k", OC.depolarizing(0.05, 0.1, 0.15), [1])
        c.cnot(1, 2); rec("u", G.CNOT, [1, 2])
        return c

    def f(p):
        c = build(p[0], p[1])
        return tc.backend.real(c.expectation((tc.gates.z(), [0])) + 0.5 * c.expectation((tc.gates.x(), [2])))

    def ref(p):
        o = []
        build(p[0], p[1], o)
        rho = odm.run(n, o)
        return np.real(odm.expectation(rho, n, (G.Z, [0])) + 0.5 * odm.expectation(rho, n, (G.X, [2])))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    p0 = np.array([0.7, 1.3])
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0.astype(rdt)))
    np.testing.assert_allclose(float(v), ref(p0), atol=2e-5 if tc.dtypestr == "complex64" else 1e-9)
    eps = 1e-5
    fd = np.array([(ref(p0 + eps * e) - ref(p0 - eps * e)) / (2 * eps) for e in np.eye(2)])
    assert np.abs(fd).min() > 0.05   # both true gradients are far from zero
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=3e-4 if tc.dtypestr == "complex64" else 1e-6)


def _random_two_qubit_channel(seed, nk=3):
    """nk Kraus operators of a random two-qubit CPTP map: blocks of a Haar isometry [4 nk, 4] (Stinespring)."""
    rng = np.random.default_rng(seed)
    z = rng.normal(size=(4 * nk, 4)) + 1j * rng.normal(size=(4 * nk, 4))
    q, _ = np.linalg.qr(z)
    return [q[4 * k:4 * k + 4, :] for k in range(nk)]


def test_two_qubit_kraus_channels(tcd):
    """Row f4: two-qubit channels (reference densitymatrix.py:222-244 with a two-index ``general_kraus``; channel
    constructors channels.py:103-230) against the dense oracle with oracle-built Kraus operators: a two-qubit
    depolarizing channel on non-adjacent qubits, a random CPTP map, gates before and after, one-qubit noise between."""
    tc = tcd
    n = 4
    ops = []
    c = tc.DMCircuit(n)
    for i in range(n):
        c.h(i); ops.append(("u", G.H, [i]))
    c.rx(1, theta=0.3); ops.append(("u", G.rx(0.3), [1]))
    c.cnot(0, 2); ops.append(("u", G.CNOT, [0, 2]))
    ks = tc.channels.generaldepolarizingchannel(0.02, 2)
    tc.channels.kraus_identity_check(ks)
    oks = OC.generaldepolarizing(0.02, 2)
    for a, b in zip(ks, oks):
        np.testing.assert_allclose(np.asarray(a.tensor).reshape(4, 4), b, atol=1e-7)
    c.general_kraus(ks, [3, 1]); ops.append(("k", oks, [3, 1]))
    c.ry(3, theta=0.9); ops.append(("u", G.ry(0.9), [3]))
    c.depolarizing(0, px=0.05, py=0.0, pz=0.1); ops.append(("k", OC.depolarizing(0.05, 0.0, 0.1), [0]))
    rk = _random_two_qubit_channel(11)
    c.general_kraus([tc.gates.Gate(k.reshape(2, 2, 2, 2)) for k in rk], [0, 1]); ops.append(("k", rk, [0, 1]))
    c.cz(1, 2); ops.append(("u", G.CZ, [1, 2]))
    iso = tc.channels.isotropicdepolarizingchannel(0.3, 2)
    c.apply_general_kraus(iso, [2, 3]); ops.append(("k", OC.isotropicdepolarizing(0.3, 2), [2, 3]))
    rho = odm.run(n, ops)
    tol = 2e-6 if tc.dtypestr == "complex64" else 1e-12
    np.testing.assert_allclose(tc.backend.numpy(c.densitymatrix()), rho, atol=tol)
    c.check_density_matrix(c.densitymatrix())
    np.testing.assert_allclose(complex(c.expectation_ps(z=[1], x=[3])), odm.expectation(rho, n, (G.Z, [1]), (G.X, [3])),
                               atol=10 * tol)
    with pytest.raises(NotImplementedError):
        c.general_kraus([np.eye(8)], [0, 1, 2])


def test_gradient_through_two_qubit_channel(tcd):
    tc = tcd
    n = 3
    oks = OC.generaldepolarizing(0.03, 2)

    def build(t, ops=None):
        c = tc.DMCircuit(n)
        rec = (lambda *a: ops.append(a)) if ops is not None else (lambda *a: None)
        c.h(0); rec("u", G.H, [0])
        c.rx(1, theta=t); rec("u", G.rx(float(t)), [1])
        c.cnot(1, 2); rec("u", G.CNOT, [1, 2])
        c.general_kraus(tc.channels.generaldepolarizingchannel(0.03, 2), [0, 1]); rec("k", oks, [0, 1])
        c.ry(0, theta=2 * t); rec("u", G.ry(2 * float(t)), [0])
        return c

    def f(t):
        return tc.backend.real(build(t).expectation((tc.gates.z(), [0]), (tc.gates.z(), [2])))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(np.array(0.41, dtype=rdt)))

    def ref(t):
        o = []
        build(t, o)
        return np.real(odm.expectation(odm.run(n, o), n, (G.Z, [0]), (G.Z, [2])))

    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-9
    np.testing.assert_allclose(float(v), ref(0.41), atol=tol)
    eps = 1e-5
    np.testing.assert_allclose(float(g), (ref(0.41 + eps) - ref(0.41 - eps)) / (2 * eps),
                               atol=2e-4 if tc.dtypestr == "complex64" else 1e-6)
