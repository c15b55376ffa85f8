"""Second-order derivatives and the vjp / jvp / jacfwd / jacrev / hessian entries of the backend (reference
abstract_backend.py:2295-2492; reference tests: tests/test_backends.py jvp / vjp / jac / hessian cases) on circuits, against
central differences of ``oracle.dense`` in float64."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ops(n, theta):
    """H layer, rzz ladder, rx, ry on every qubit, a crossing cz: theta [3, n]."""
    from oracle import gates as G

    ops = [(G.H, [i]) for i in range(n)]
    for i in range(n - 1):
        ops.append((G.rzz(theta[0, i]), [i, i + 1]))
    for i in range(n):
        ops.append((G.rx(theta[1, i]), [i]))
        ops.append((G.ry(theta[2, i]), [i]))
    ops.append((G.CZ, [0, n - 1]))
    return ops


def _circuit(tc, n, theta):
    c = tc.Circuit(n)
    for i in range(n):
        c.h(i)
    for i in range(n - 1):
        c.rzz(i, i + 1, theta=theta[0, i])
    for i in range(n):
        c.rx(i, theta=theta[1, i])
        c.ry(i, theta=theta[2, i])
    c.cz(0, n - 1)
    return c


def _energy_oracle(n, theta):
    from oracle import dense, gates as G

    psi = dense.run(n, _ops(n, theta))
    e = 0.0
    for i in range(n - 1):
        e += np.real(np.vdot(psi, dense.apply_gate(psi, n, np.kron(G.Z, G.Z), [i, i + 1])))
    for i in range(n):
        e -= 0.7 * np.real(np.vdot(psi, dense.apply_gate(psi, n, G.X, [i])))
    return e


def _fd_hessian(fun, x, h=1e-4):
    x = np.asarray(x, dtype=np.float64)
    sz = x.size
    H = np.zeros((sz, sz))
    for i in range(sz):
        for j in range(i, sz):
            def at(si, sj):
                y = x.reshape(-1).copy()
                y[i] += si * h
                y[j] += sj * h
                return fun(y.reshape(x.shape))
            H[i, j] = H[j, i] = (at(1, 1) - at(1, -1) - at(-1, 1) + at(-1, -1)) / (4 * h * h)
    return H


@pytest.mark.parametrize("dt,tol", [("complex128", 2e-6), ("complex64", 3e-3)])
def test_hessian_of_a_tfim_energy(dt, tol):
    import tcmi as tc

    tc.set_backend("hip"); tc.set_dtype(dt)
    n = 4
    theta = np.random.default_rng(0).uniform(0, 2 * np.pi, [3, n])

    def energy(t):
        c = _circuit(tc, n, t)
        e = 0.0
        for i in range(n - 1):
            e = e + c.expectation_ps(z=[i, i + 1])
        for i in range(n):
            e = e - 0.7 * c.expectation_ps(x=[i])
        return tc.backend.real(e)

    x = tc.backend.convert_to_tensor(theta.astype(np.float64 if dt == "complex128" else np.float32))
    H = tc.backend.numpy(tc.backend.hessian(energy)(x)).reshape(3 * n, 3 * n)
    want = _fd_hessian(lambda t: _energy_oracle(n, t), theta)
    assert np.abs(H - H.T).max() < tol
    np.testing.assert_allclose(H, want, atol=tol)
    # the unused angle theta[0, n - 1] has a zero row
    k = 0 * n + (n - 1)
    assert np.abs(H[k]).max() == 0.0
    tc.set_dtype("complex64")


def test_jvp_vjp_and_jacobians_of_a_state_function():
    import torch
    import tcmi as tc
    from oracle import dense

    tc.set_backend("hip"); tc.set_dtype("complex128")
    try:
        n = 4
        rng = np.random.default_rng(1)
        theta = rng.uniform(0, 2 * np.pi, [3, n])
        v = rng.normal(size=[3, n])

        def state(t):
            return _circuit(tc, n, t).state()

        x = tc.backend.convert_to_tensor(theta)
        psi, tang = tc.backend.jvp(state, x, tc.backend.convert_to_tensor(v))
        h = 1e-5
        fd = (dense.run(n, _ops(n, theta + h * v)) - dense.run(n, _ops(n, theta - h * v))) / (2 * h)
        np.testing.assert_allclose(tc.backend.numpy(psi), dense.run(n, _ops(n, theta)), atol=1e-10)
        np.testing.assert_allclose(tc.backend.numpy(tang), fd, atol=1e-8)
        # vjp: Re <w | J v> = <vjp(w), v> for real parameters (torch's convention for complex cotangents)
        w = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
        _, back = tc.backend.vjp(state, x, tc.backend.convert_to_tensor(w))
        lhs = np.real(np.vdot(w, fd))
        assert abs(lhs - float((back * torch.as_tensor(v, device=back.device)).sum())) < 1e-7
        # Jacobians: both modes, shape output + input, equal to the jvp columns
        def probs(t):
            c = _circuit(tc, n, t)
            return tc.backend.stack([tc.backend.real(c.expectation_ps(z=[1])),
                                     tc.backend.real(c.expectation_ps(z=[1]) + c.expectation_ps(x=[0, 2]))])

        jr = tc.backend.numpy(tc.backend.jacrev(probs)(x))
        jf = tc.backend.numpy(tc.backend.jacfwd(probs)(x))
        assert jr.shape == (2, 3, n) and jf.shape == (2, 3, n)
        np.testing.assert_allclose(jr, jf, atol=1e-9)
    finally:
        tc.set_dtype("complex64")


def test_hessian_through_a_nonlinear_function_of_expectations():
    """The cotangent of the expectation values depends on the parameters too ((<Z_1> - 0.3)^2 <X_0 X_2>): the g-slot rule of
    the measurement primitive's backward."""
    import tcmi as tc
    from oracle import dense, gates as G

    tc.set_backend("hip"); tc.set_dtype("complex128")
    try:
        n = 4
        theta = np.random.default_rng(2).uniform(0, 2 * np.pi, [3, n])

        def loss(t):
            c = _circuit(tc, n, t)
            z = tc.backend.real(c.expectation_ps(z=[1]))
            xx = tc.backend.real(c.expectation_ps(x=[0, 2]))
            return (z - 0.3) ** 2 * xx

        def loss_oracle(t):
            psi = dense.run(n, _ops(n, t))
            z = np.real(np.vdot(psi, dense.apply_gate(psi, n, G.Z, [1])))
            xx = np.real(np.vdot(psi, dense.apply_gate(psi, n, np.kron(G.X, G.X), [0, 2])))
            return (z - 0.3) ** 2 * xx

        H = tc.backend.numpy(tc.backend.hessian(loss)(tc.backend.convert_to_tensor(theta))).reshape(3 * n, 3 * n)
        np.testing.assert_allclose(H, _fd_hessian(loss_oracle, theta), atol=2e-6)
    finally:
        tc.set_dtype("complex64")
