"""Reverse mode on the GPU (adjoint sweep + Pauli-sum cotangent kernels) through the reference's
backend API: value_and_grad / grad / vmap / vectorized_value_and_grad
(reference abstract_backend.py:2262-2293, 2520-2591)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, workloads as W  # noqa: E402
from tcmi import _knobs as KN  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hea_golden.npz"))


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(request.param)
    yield tc
    tc.set_dtype("complex64")


def _tfim(tc, c, n, j=1.0, h=-1.0):
    e = 0.0
    for i in range(n):
        e += h * c.expectation((tc.gates.x(), [i]))
    for i in range(n - 1):
        e += j * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
    return tc.backend.real(e)


def test_operator_measurement_kat(tcd):
    """reference tests/test_templates.py:191-211: value 0.84147, gradient 0.54032 (atol 1e-4)."""
    tc = tcd

    def f(theta):
        c = tc.Circuit(2)
        c.ry(0, theta=theta)
        c.H(1)
        return tc.backend.real(c.expectation((tc.gates.x(), [0])))

    v, g = tc.backend.jit(tc.backend.value_and_grad(f))(tc.backend.ones([], dtype=tc.rdtypestr))
    np.testing.assert_allclose(tc.backend.numpy(v), 0.84147, atol=1e-4)
    np.testing.assert_allclose(tc.backend.numpy(g), 0.54032, atol=1e-4)


@pytest.mark.parametrize("n,d", [(6, 2), (10, 4)])
def test_tfim_value_and_grad_golden(tcd, n, d):
    """HEA-B + TFIM energy (benchmarks/scripts/vqe_tc.py:107-141) against the committed
    central-difference gradients of the oracle."""
    tc = tcd
    params = GOLD[f"tfim_{n}_{d}_params"]

    def f(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return _tfim(tc, c, n)

    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr))
    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-7
    np.testing.assert_allclose(tc.backend.numpy(v), GOLD[f"tfim_{n}_{d}_energy"], atol=tol)
    assert tuple(g.shape) == params.shape
    np.testing.assert_allclose(tc.backend.numpy(g), GOLD[f"tfim_{n}_{d}_grad"], atol=tol)


def test_two_argument_grad_and_aux(tcd):
    """argnums=(0, 1), has_aux=True exactly as benchmarks/scripts/vqe_tc.py:107-141."""
    tc = tcd
    n, nlayer = 5, 2
    rng = np.random.default_rng(0)
    px, pzz = rng.normal(size=nlayer * n), rng.normal(size=nlayer * n)

    def energy_raw(paramx, paramzz):
        c = tc.Circuit(n)
        for i in range(n):
            c.H(i)
        for j in range(nlayer):
            for i in range(n - 1):
                c.exp1(i, i + 1, theta=paramzz[j * n + i], unitary=tc.gates._zz_matrix)
            for i in range(n):
                c.rx(i, theta=paramx[j * n + i])
        return _tfim(tc, c, n), 1.0

    vag = tc.backend.jit(tc.backend.value_and_grad(energy_raw, argnums=(0, 1), has_aux=True))
    (e, fd), (gx, gzz) = vag(tc.backend.convert_to_tensor(px, dtype=tc.rdtypestr),
                             tc.backend.convert_to_tensor(pzz, dtype=tc.rdtypestr))
    assert fd == 1.0

    def ref(px_, pzz_):
        ops = [(G.H, [i]) for i in range(n)]
        for j in range(nlayer):
            ops += [(G.exp1(G.ZZ, pzz_[j * n + i]), [i, i + 1]) for i in range(n - 1)]
            ops += [(G.rx(px_[j * n + i]), [i]) for i in range(n)]
        return W.tfim_energy_dense(dense.run(n, ops), n)

    eps = 1e-6
    fgx = np.array([(ref(px + eps * np.eye(px.size)[i], pzz) - ref(px - eps * np.eye(px.size)[i], pzz)) / (2 * eps) for i in range(px.size)])
    fgz = np.array([(ref(px, pzz + eps * np.eye(px.size)[i]) - ref(px, pzz - eps * np.eye(px.size)[i])) / (2 * eps) for i in range(px.size)])
    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-7
    np.testing.assert_allclose(tc.backend.numpy(e), ref(px, pzz), atol=tol)
    np.testing.assert_allclose(tc.backend.numpy(gx), fgx, atol=tol)
    np.testing.assert_allclose(tc.backend.numpy(gzz), fgz, atol=tol)
    assert abs(fgz[n - 1]) < 1e-12 and abs(tc.backend.numpy(gzz)[n - 1]) < 1e-12  # unused parameter


def test_mixed_gates_gradient(tcd):
    """Every differentiable gate family + constant gates in between, multi-pass plan (n = 14)."""
    tc = tcd
    n = 14
    rng = np.random.default_rng(5)
    p0 = rng.uniform(0, 2 * np.pi, 12)

    def build(c, p, api):
        for i in range(n):
            c.h(i)
        c.rx(0, theta=p[0]); c.ry(13, theta=p[1]); c.rz(5, theta=p[2]); c.cnot(0, 13)
        c.rzz(2, 9, theta=p[3]); c.rxx(12, 1, theta=p[4]); c.ryy(3, 4, theta=p[5])
        c.phase(7, theta=p[6]); c.crx(8, 6, theta=p[7]); c.swap(1, 10)
        c.cphase(11, 2, theta=p[8]); c.iswap(6, 7, theta=p[9]); c.u(4, theta=p[10], phi=p[11], lbd=0.4)
        c.exp1(0, 5, unitary=api["xz"], theta=p[0] * 0.5 + 0.1); c.cz(3, 12)

    def f(p):
        c = tc.Circuit(n)
        build(c, p, {"xz": np.kron(G.X, G.Z)})
        return tc.backend.real(
            c.expectation_ps(x=[0, 5]) + 0.7 * c.expectation_ps(z=[13], y=[2]) - 0.3 * c.expectation_ps(z=[4, 9, 12])
        )

    def ref(p):
        ops = [(G.H, [i]) for i in range(n)]
        ops += [(G.rx(p[0]), [0]), (G.ry(p[1]), [13]), (G.rz(p[2]), [5]), (G.CNOT, [0, 13]),
                (G.rzz(p[3]), [2, 9]), (G.rxx(p[4]), [12, 1]), (G.ryy(p[5]), [3, 4]), (G.phase(p[6]), [7]),
                (G.controlled(G.rx(p[7])), [8, 6]), (G.SWAP, [1, 10]), (G.controlled(G.phase(p[8])), [11, 2]),
                (G.iswap(p[9]), [6, 7]), (G.u(p[10], p[11], 0.4), [4]),
                (G.exp1(np.kron(G.X, G.Z), p[0] * 0.5 + 0.1), [0, 5]), (G.CZ, [3, 12])]
        psi = dense.run(n, ops)
        pe = lambda ps: dense.pauli_string_expectation(psi, n, ps).real
        s1 = [0] * n; s1[0] = 1; s1[5] = 1
        s2 = [0] * n; s2[13] = 3; s2[2] = 2
        s3 = [0] * n; s3[4] = 3; s3[9] = 3; s3[12] = 3
        return pe(s1) + 0.7 * pe(s2) - 0.3 * pe(s3)

    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0, dtype=tc.rdtypestr))
    eps = 1e-6
    fd = np.array([(ref(p0 + eps * np.eye(12)[i]) - ref(p0 - eps * np.eye(12)[i])) / (2 * eps) for i in range(12)])
    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-7
    np.testing.assert_allclose(tc.backend.numpy(v), ref(p0), atol=tol)
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=tol)


def test_vvag_semantics(tcd):
    """reference tests/test_backends.py:890-939: vvag == per-sample value_and_grad; gradients of the
    non-vectorised argument are summed over the batch (g11 / batch == g01)."""
    tc = tcd
    n, d, B = 6, 2, 4
    rng = np.random.default_rng(11)
    w = tc.backend.convert_to_tensor(rng.normal(size=[2 * d, n]), dtype=tc.rdtypestr)
    xs = tc.backend.convert_to_tensor(rng.normal(size=[B, n]), dtype=tc.rdtypestr)

    def f(x, weights):
        c = tc.Circuit(n)
        for i in range(n):
            c.rx(i, theta=x[i])
        W.hea_b(c, n, d, weights, zz=tc.gates._zz_matrix)
        return _tfim(tc, c, n)

    vs, (gx, gw) = tc.backend.vvag(f, argnums=(0, 1), vectorized_argnums=0)(xs, w)
    assert tuple(vs.shape) == (B,) and tuple(gx.shape) == (B, n) and tuple(gw.shape) == (2 * d, n)
    vag = tc.backend.value_and_grad(f, argnums=(0, 1))
    gw_sum = 0
    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-8
    for b in range(B):
        v1, (g1x, g1w) = vag(xs[b], w)
        np.testing.assert_allclose(tc.backend.numpy(vs[b]), tc.backend.numpy(v1), atol=tol)
        np.testing.assert_allclose(tc.backend.numpy(gx[b]), tc.backend.numpy(g1x), atol=tol)
        gw_sum = gw_sum + tc.backend.numpy(g1w)
    np.testing.assert_allclose(tc.backend.numpy(gw), gw_sum, atol=tol * B)
    # identical samples: summed shared gradient / batch == single gradient
    same = tc.backend.stack([xs[0]] * B)
    _, (_, gw_same) = tc.backend.vvag(f, argnums=(0, 1), vectorized_argnums=0)(same, w)
    _, (_, g01) = vag(xs[0], w)
    np.testing.assert_allclose(tc.backend.numpy(gw_same) / B, tc.backend.numpy(g01), atol=tol)


def test_grad_at_scale_vs_single_precision_pair():
    """n = 20, d = 3: complex64 gradient against the complex128 gradient of the same kernels
    (self-consistency at a size the dense oracle's finite differences cannot reach in seconds)."""
    import tcmi as tc

    tc.set_backend("hip")
    n, d = 20, 3
    params = np.random.default_rng(20).normal(0, 0.3, [2 * d, n])
    outs = {}
    for dt in ("complex64", "complex128"):
        tc.set_dtype(dt)

        def f(p):
            c = tc.Circuit(n)
            W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            return _tfim(tc, c, n)

        v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr))
        outs[dt] = (float(tc.backend.numpy(v)), tc.backend.numpy(g).astype(np.float64))
    tc.set_dtype("complex64")
    assert abs(outs["complex64"][0] - outs["complex128"][0]) < 1e-4
    assert np.abs(outs["complex64"][1] - outs["complex128"][1]).max() < 5e-4
    # energy against the dense oracle
    want = W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, params)), n)
    assert abs(outs["complex128"][0] - want) < 1e-9


def _np(tc, t):
    return tc.backend.numpy(t)


def test_pauli_sum_hamiltonian_matrix_free(tcd):
    """PauliStringSum2COO / PauliStringSum2MVP / operator_expectation (reference
    quantum.py:2222-2358, templates/measurements.py:156-191) without materialising H: TFIM at n=12 against
    the oracle energy, H|psi> against a dense Kronecker build at n=6, and the gradient through it."""
    tc = tcd
    n, d = 12, 3
    structures, weights = [], []
    for i in range(n):
        s = [0] * n; s[i] = 1
        structures.append(s); weights.append(-1.0)
    for i in range(n - 1):
        s = [0] * n; s[i] = 3; s[i + 1] = 3
        structures.append(s); weights.append(1.0)
    h = tc.quantum.PauliStringSum2COO(structures, weights)
    assert tc.backend.is_sparse(h)
    rng = np.random.default_rng(5)
    params = rng.normal(0, 0.3, [2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return tc.templates.measurements.operator_expectation(c, h)

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    v, g = tc.backend.value_and_grad(energy)(tc.backend.convert_to_tensor(params.astype(rdt)))
    def ref(p):
        return W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, p)), n)

    want = ref(params)
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    np.testing.assert_allclose(_np(tc, v), want, atol=tol * 10)
    eps = 1e-5
    pp, pm = params.copy(), params.copy()
    pp[1, 3] += eps; pm[1, 3] -= eps
    fd = (ref(pp) - ref(pm)) / (2 * eps)
    np.testing.assert_allclose(_np(tc, g)[1, 3], fd, atol=2e-4 if tc.dtypestr == "complex64" else 1e-7)
    # matrix-vector product vs a dense Kronecker build
    m = 6
    st = [[1, 0, 2, 3, 0, 1], [3, 3, 0, 0, 2, 2], [0, 0, 0, 0, 0, 3]]
    wt = [0.5, -1.25, 2.0]
    mvp = tc.quantum.PauliStringSum2MVP(st, wt)
    psi = rng.normal(size=2**m) + 1j * rng.normal(size=2**m)
    dense_h = np.zeros((2**m, 2**m), dtype=np.complex128)
    for s, w in zip(st, wt):
        t = np.array([[1.0]])
        for p in s:
            t = np.kron(t, G.PAULI[p])
        dense_h += w * t
    got = _np(tc, mvp(tc.backend.cast(tc.backend.convert_to_tensor(psi), tc.dtypestr)))
    np.testing.assert_allclose(got, dense_h @ psi, atol=tol * 50)
    ket = tc.backend.cast(tc.backend.convert_to_tensor(psi.reshape(-1, 1)), tc.dtypestr)
    got2 = _np(tc, tc.backend.sparse_dense_matmul(tc.quantum.PauliStringSum2COO(st, wt), ket))
    np.testing.assert_allclose(got2[:, 0], dense_h @ psi, atol=tol * 50)


def test_torch_interface_and_quantumnet():
    """SURVEY 8f rank 3 (reference interfaces/torch.py:17-125, torchnn.py:16-138): a quantum function inside
    torch autograd and as an nn.Module trained by a torch optimiser (batched inputs are vmapped)."""
    import torch
    import tcmi as tc
    from tcmi import torchnn

    tc.set_dtype("complex64")

    def f(params):
        c = tc.Circuit(1)
        c.rx(0, theta=params[0])
        c.ry(0, theta=params[1])
        return tc.backend.real(c.expectation([tc.gates.z(), [0]]))

    ft = tc.interfaces.torch_interface(f, jit=True)
    a = torch.ones([2], requires_grad=True, device="cuda")
    b = ft(a)
    (b ** 2).backward()
    # <Z> = cos(a0) cos(a1); d/da0 (cos^2 cos^2) = -2 cos a0 sin a0 cos^2 a1
    want = -2 * np.cos(1.0) * np.sin(1.0) * np.cos(1.0) ** 2
    np.testing.assert_allclose(a.grad.cpu().numpy(), [want, want], atol=2e-5)

    n, nlayers, batch = 6, 2, 4

    def qpred(x, weights):
        c = tc.Circuit(n)
        for i in range(n):
            c.rx(i, theta=x[i])
        for j in range(nlayers):
            for i in range(n - 1):
                c.cnot(i, i + 1)
            for i in range(n):
                c.rx(i, theta=weights[2 * j, i])
                c.ry(i, theta=weights[2 * j + 1, i])
        return tc.backend.real(tc.backend.stack([c.expectation_ps(z=[i]) for i in range(n)]))

    torch.manual_seed(0)
    ql = torchnn.QuantumNet(qpred, weights_shape=[2 * nlayers, n]).cuda()
    x = torch.rand([batch, n], device="cuda")
    y = ql(x)
    assert tuple(y.shape) == (batch, n)
    opt = torch.optim.Adam(ql.parameters(), lr=0.1)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = ((ql(x) - 1.0) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0] * 0.7


@pytest.mark.parametrize("n,d", [(10, 3), (17, 2)])
def test_jit_traced_value_and_grad(tcd, n, d):
    """backend.jit(value_and_grad(f)) / jit(vvag(f)) (reference idiom benchmarks/scripts/vqe_tc.py:136-141): after
    the first calls the host side is a traced pipeline (tcmi/jit.py); values and gradients equal the plain
    path, untraceable functions silently keep the plain path."""
    import torch
    tc = tcd
    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    rng = np.random.default_rng(n)

    def energy(p, q):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for j in range(d):
            for i in range(n - 1):
                c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
            for i in range(n):
                c.rx(i, theta=p[2 * j + 1, i])
        for i in range(n):
            c.ry(i, theta=q[i])
        c.rz(0, theta=0.3)                                   # a constant angle
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation_ps(x=[i])
        for i in range(n - 1):
            e += 0.5 * c.expectation_ps(z=[i, i + 1])
        return tc.backend.real(e) + 0.25

    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-9
    plain = tc.backend.value_and_grad(energy, argnums=(0, 1))
    fast = tc.backend.jit(tc.backend.value_and_grad(energy, argnums=(0, 1)))
    for trial in range(4):
        p = tc.backend.convert_to_tensor(rng.normal(size=[2 * d, n]).astype(rdt))
        q = tc.backend.convert_to_tensor(rng.normal(size=[n]).astype(rdt))
        v0, (g0p, g0q) = plain(p, q)
        v1, (g1p, g1q) = fast(p, q)
        np.testing.assert_allclose(_np(tc, v1), _np(tc, v0), atol=tol)
        np.testing.assert_allclose(_np(tc, g1p), _np(tc, g0p), atol=tol)
        np.testing.assert_allclose(_np(tc, g1q), _np(tc, g0q), atol=tol)
        assert g1p.dtype == g0p.dtype and tuple(g1p.shape) == tuple(g0p.shape)
    assert fast.stats["fast"] >= 2
    # single argnum + vvag: per-sample gradients for the vectorised argument, summed for the shared one
    B = 3
    vv_plain = tc.backend.vvag(energy, argnums=(0, 1), vectorized_argnums=0)
    vv_fast = tc.backend.jit(tc.backend.vvag(energy, argnums=(0, 1), vectorized_argnums=0))
    for trial in range(3):
        pb = tc.backend.convert_to_tensor(rng.normal(size=[B, 2 * d, n]).astype(rdt))
        q = tc.backend.convert_to_tensor(rng.normal(size=[n]).astype(rdt))
        v0, (g0p, g0q) = vv_plain(pb, q)
        v1, (g1p, g1q) = vv_fast(pb, q)
        np.testing.assert_allclose(_np(tc, v1), _np(tc, v0), atol=tol)
        np.testing.assert_allclose(_np(tc, g1p), _np(tc, g0p), atol=tol)
        np.testing.assert_allclose(_np(tc, g1q), _np(tc, g0q), atol=tol * B)
    assert vv_fast.stats["fast"] >= 1

    def scaled(p):                                            # arithmetic on the argument: not traceable
        c = tc.Circuit(n)
        for i in range(n):
            c.rx(i, theta=2.0 * p[i])
        return tc.backend.real(c.expectation_ps(z=[0]))

    js = tc.backend.jit(tc.backend.value_and_grad(scaled))
    p1 = tc.backend.convert_to_tensor(rng.normal(size=[n]).astype(rdt))
    for _ in range(3):
        v, g = js(p1)
    np.testing.assert_allclose(_np(tc, v), np.cos(2 * _np(tc, p1)[0]), atol=tol)
    np.testing.assert_allclose(_np(tc, g)[0], -2 * np.sin(2 * _np(tc, p1)[0]), atol=tol)
    assert js.stats["fast"] == 0


def test_jit_plain_functions(tcd):
    """jit(f) of a plain energy function or of a state-returning function is traced (value only); functions that
    need device results while being probed (MPS, arithmetic on expectations) run unchanged."""
    tc = tcd
    n = 9
    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    rng = np.random.default_rng(1)

    def energy(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.ry(i, theta=p[i])
        for i in range(n - 1):
            c.cnot(i, i + 1)
        return tc.backend.real(c.expectation_ps(z=[0, n - 1]) - 0.5 * c.expectation_ps(x=[3]))

    def state(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.ry(i, theta=p[i])
        return c.wavefunction()

    def squared(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.ry(i, theta=p[i])
        return tc.backend.real(c.expectation_ps(z=[0])) ** 2

    je, js, jq = tc.backend.jit(energy), tc.backend.jit(state), tc.backend.jit(squared)
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    for _ in range(3):
        p = tc.backend.convert_to_tensor(rng.normal(size=[n]).astype(rdt))
        np.testing.assert_allclose(_np(tc, je(p)), _np(tc, energy(p)), atol=tol)
        np.testing.assert_allclose(_np(tc, js(p)), _np(tc, state(p)), atol=tol)
        np.testing.assert_allclose(_np(tc, jq(p)), _np(tc, squared(p)), atol=tol)
    assert je.stats["fast"] >= 1 and js.stats["fast"] >= 1 and jq.stats["fast"] == 0
    jv = tc.backend.jit(tc.backend.vmap(energy))
    pb = tc.backend.convert_to_tensor(rng.normal(size=[4, n]).astype(rdt))
    for _ in range(3):
        np.testing.assert_allclose(_np(tc, jv(pb)), _np(tc, tc.backend.vmap(energy)(pb)), atol=tol)
    assert jv.stats["fast"] >= 1


def test_gradient_through_a_non_unitary_gate(tcd):
    """`any` with a non-unitary matrix between two rotations (round-1 advisor finding): gradients of the parameters in
    front of it must come from checkpointed segments, not from un-computing psi with M^dagger."""
    tc = tcd
    n = 9
    m = np.array([[1.0, 0.3], [0.1j, 0.8]])

    def f(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        c.ry(0, theta=p[0])
        c.cnot(0, 1)
        c.any(0, unitary=m)
        c.rx(0, theta=p[1])
        c.cnot(1, 2)
        return tc.backend.real(c.expectation_ps(z=[0]) + c.expectation_ps(x=[1]))

    def ref(p):
        ops = [(G.H, [i]) for i in range(n)] + [(G.ry(p[0]), [0]), (G.CNOT, [0, 1]), (m, [0]), (G.rx(p[1]), [0]),
                                                 (G.CNOT, [1, 2])]
        psi = dense.run(n, ops)
        z0 = 1 - 2 * ((np.arange(2**n) >> (n - 1)) & 1)
        x1 = np.vdot(psi, psi.reshape(2, 2, -1)[:, ::-1].reshape(-1))
        return float(np.real(np.vdot(psi, z0 * psi) + x1))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    p0 = np.array([0.9, 0.4])
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0.astype(rdt)))
    np.testing.assert_allclose(float(v), ref(p0), atol=3e-5 if tc.dtypestr == "complex64" else 1e-9)
    eps = 1e-5
    fd = np.array([(ref(p0 + eps * e) - ref(p0 - eps * e)) / (2 * eps) for e in np.eye(2)])
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=5e-4 if tc.dtypestr == "complex64" else 1e-6)


def test_gradient_with_respect_to_the_input_state(tcd):
    """Circuit(inputs=psi): the reference differentiates through the input state (circuit.py:90-104).  Chained
    circuits: the parameters of the first circuit reach the loss only through the second circuit's inputs."""
    tc = tcd
    n = 9

    def f(p):
        c1 = tc.Circuit(n)
        for i in range(n):
            c1.ry(i, theta=p[i])
        c1.cnot(0, 1)
        c2 = tc.Circuit(n, inputs=c1.state())
        c2.h(2)
        c2.rx(1, theta=p[n])
        return tc.backend.real(c2.expectation_ps(z=[1]) + c2.expectation_ps(x=[2]))

    def ref(p):
        ops = [(G.ry(p[i]), [i]) for i in range(n)] + [(G.CNOT, [0, 1]), (G.H, [2]), (G.rx(p[n]), [1])]
        psi = dense.run(n, ops)
        z1 = 1 - 2 * ((np.arange(2**n) >> (n - 2)) & 1)
        x2 = np.vdot(psi, psi.reshape(4, 2, -1)[:, ::-1].reshape(-1))
        return float(np.real(np.vdot(psi, z1 * psi) + x2))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    p0 = np.random.default_rng(3).uniform(0.2, 2.5, n + 1)
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0.astype(rdt)))
    np.testing.assert_allclose(float(v), ref(p0), atol=3e-5 if tc.dtypestr == "complex64" else 1e-9)
    eps = 1e-5
    fd = np.array([(ref(p0 + eps * e) - ref(p0 - eps * e)) / (2 * eps) for e in np.eye(n + 1)])
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=5e-4 if tc.dtypestr == "complex64" else 1e-6)


def test_two_shear_reverse_sweep_on_the_gpu(monkeypatch):
    """knob shear2_bw=1: the reverse sweep with rotations in two-shear form (lambda sheared in the other order, second
    phase table with the reciprocal factors) gives the gradient of the default three-shear sweep and of the float64
    kernels; angles on both sides of the builder's |cos| threshold."""
    import tcmi as tc

    tc.set_backend("hip")
    n, d = 15, 3
    rng = np.random.default_rng(15)
    params = np.where(rng.random([2 * d, n]) < 0.7, rng.normal(0, 0.3, [2 * d, n]), rng.uniform(2.2, 4.0, [2 * d, n]))

    def grad(dt, extra):
        tc.set_dtype(dt)

        def f(p):
            c = tc.Circuit(n)
            W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            c.rx(extra, theta=0.25)      # a structure no other test compiles: the plan is built under the switch
            return _tfim(tc, c, n)

        v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr))
        return float(tc.backend.numpy(v)), tc.backend.numpy(g).astype(np.float64)

    try:
        monkeypatch.setitem(KN.VALUES, "shear2_bw", "1")
        v2, g2 = grad("complex64", 0)
        from tcmi import executor as X
        cc = next(reversed(X._CACHE.values()))
        cc = getattr(cc, "full", cc)
        recs = np.asarray(cc._adjoint(full=False)["plan"].ginfo).reshape(-1, 8)
        assert ((recs[:, 0] == 3) & (recs[:, 6] != 0)).sum() >= 10     # BK_UDAG records allowed the two-shear form
        monkeypatch.setitem(KN.VALUES, "shear2_bw", "0")
        v64, g64 = grad("complex128", 0)
    finally:
        tc.set_dtype("complex64")
    assert abs(v2 - v64) < 1e-4
    assert np.abs(g2 - g64).max() < 5e-4


def test_input_state_cotangent_through_wide_angle_rotations(tcd):
    """The packed reverse sweep pulls a sign out of rotations with cos < 0 (three-shear form) and leaves real factors
    pending after two-shear ones; neither may reach the input-state cotangent.  Chained circuits on a packed-kernel
    tile (n = 13), second circuit with rotation angles on both sides of pi; gradient of the FIRST circuit's angles
    against central differences of the dense oracle."""
    tc = tcd
    n = 13
    wide = [4.0, 0.5, 3.6, 5.9, 0.9, 2.4]

    def f(p):
        c1 = tc.Circuit(n)
        for i in range(n):
            c1.ry(i, theta=p[i])
        c1.cnot(0, 1)
        c2 = tc.Circuit(n, inputs=c1.state())
        for k, a in enumerate(wide):
            c2.rx(k, theta=p[n] * a)
        for k in range(len(wide) - 1):
            c2.rzz(k, k + 1, theta=0.3 + 0.1 * k)
        for k, a in enumerate(wide):
            c2.rx(k, theta=a)
        c2.ry(3, theta=4.4)
        return tc.backend.real(c2.expectation_ps(z=[1]) + c2.expectation_ps(x=[2]))

    def ref(p):
        ops = [(G.ry(p[i]), [i]) for i in range(n)] + [(G.CNOT, [0, 1])]
        ops += [(G.rx(p[n] * a), [k]) for k, a in enumerate(wide)]
        ops += [(G.rzz(0.3 + 0.1 * k), [k, k + 1]) for k in range(len(wide) - 1)]
        ops += [(G.rx(a), [k]) for k, a in enumerate(wide)] + [(G.ry(4.4), [3])]
        psi = dense.run(n, ops)
        z1 = 1 - 2 * ((np.arange(2**n) >> (n - 2)) & 1)
        x2 = np.vdot(psi, psi.reshape(4, 2, -1)[:, ::-1].reshape(-1))
        return float(np.real(np.vdot(psi, z1 * psi) + x2))

    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    p0 = np.concatenate([np.random.default_rng(5).uniform(0.2, 2.5, n), [1.0]])
    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0.astype(rdt)))
    np.testing.assert_allclose(float(v), ref(p0), atol=3e-5 if tc.dtypestr == "complex64" else 1e-9)
    eps = 1e-5
    fd = np.array([(ref(p0 + eps * e) - ref(p0 - eps * e)) / (2 * eps) for e in np.eye(n + 1)])
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=5e-4 if tc.dtypestr == "complex64" else 1e-6)


def test_tensor_valued_gate_matrix_is_not_silently_constant(tcd):
    """A gate matrix that is being differentiated cannot be baked into the plan as a constant: NotImplementedError
    with the reference's wording instead of a silent zero gradient."""
    tc = tcd
    import torch

    def f(m):
        c = tc.Circuit(8)
        c.any(0, unitary=m)
        return tc.backend.real(c.expectation_ps(z=[0]))

    m = torch.eye(2, dtype=torch.complex64 if tc.dtypestr == "complex64" else torch.complex128, device="cuda")
    with pytest.raises(NotImplementedError, match="Backend 'hip' has not implemented"):
        tc.backend.value_and_grad(f)(m)


def test_tensor_valued_axis_angles_of_r_cr_cu(tcd):
    """r / cr with tensor-valued alpha, phi and cu with tensor parameters (reference gates.py:661-689, 817-849,
    cu = controlled(u)): value against the dense oracle and the gradient w.r.t. ALL angles by central differences."""
    tc = tcd
    n = 5
    p0 = np.array([0.7, 1.1, -0.4, 0.9, 0.35, 2.1, 0.6, -1.3, 0.8])

    def f(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        c.r(1, theta=p[0], alpha=p[1], phi=p[2])
        c.cnot(1, 3)
        c.cr(0, 2, theta=p[3], alpha=p[4], phi=p[5])
        c.cu(4, 3, theta=p[6], phi=p[7], lbd=p[8])
        c.cr(3, 1, theta=0.3, alpha=p[1], phi=0.2)            # mixed concrete / tensor
        return tc.backend.real(c.expectation_ps(z=[1, 2]) + 0.5 * c.expectation_ps(x=[3]) - 0.25 * c.expectation_ps(y=[2], z=[4]))

    def ref(p):
        ops = [(G.H, [i]) for i in range(n)]
        ops += [(G.r(p[0], p[1], p[2]), [1]), (G.CNOT, [1, 3]), (G.cr(p[3], p[4], p[5]), [0, 2]),
                (G.controlled(G.u(p[6], p[7], p[8])), [4, 3]), (G.cr(0.3, p[1], 0.2), [3, 1])]
        psi = dense.run(n, ops)
        pe = lambda ps: dense.pauli_string_expectation(psi, n, ps).real
        return pe([0, 3, 3, 0, 0]) + 0.5 * pe([0, 0, 0, 1, 0]) - 0.25 * pe([0, 0, 2, 0, 3])

    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(p0, dtype=tc.rdtypestr))
    eps = 1e-6
    fd = np.array([(ref(p0 + eps * np.eye(9)[i]) - ref(p0 - eps * np.eye(9)[i])) / (2 * eps) for i in range(9)])
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-7
    np.testing.assert_allclose(float(tc.backend.numpy(v)), ref(p0), atol=tol)
    np.testing.assert_allclose(tc.backend.numpy(g), fd, atol=10 * tol)


def test_jit_does_not_freeze_python_that_depends_on_argument_values(tcd):
    """``backend.jit`` replaces a traceable function by a fixed pipeline after probing it with index-valued arguments.
    A function whose Python branches on the VALUE of an argument must not be frozen on the probe's branch: using the
    value in Python aborts the probe (as a concretisation error would under the reference's JAX jit) and the function
    keeps the plain path, so every call follows its own branch."""
    tc = tcd
    n = 6

    def f(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.rx(i, theta=p[i])
        if float(p[0]) > 1.0:          # Python control flow on an argument value
            c.x(0)
        return tc.backend.real(c.expectation_ps(z=[0]) + 0.5 * c.expectation_ps(z=[1], x=[2]))

    fj = tc.backend.jit(tc.backend.value_and_grad(f))
    plain = tc.backend.value_and_grad(f)
    rdt = np.float32 if tc.rdtypestr == "float32" else np.float64
    for p0 in (0.4, 1.7, 0.9, 2.5):
        p = np.linspace(0.3, 1.3, n).astype(rdt)
        p[0] = p0
        pt = tc.backend.convert_to_tensor(p)
        for _ in range(2):
            v, g = fj(pt)
            vr, gr = plain(pt)
            np.testing.assert_allclose(float(v), float(vr), atol=1e-6)
            np.testing.assert_allclose(tc.backend.numpy(g), tc.backend.numpy(gr), atol=1e-6)
    assert fj.stats["fast"] == 0          # never traced
    # the same function without the branch is traced
    def f2(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.rx(i, theta=p[i])
        return tc.backend.real(c.expectation_ps(z=[0]) + 0.5 * c.expectation_ps(z=[1], x=[2]))

    fj2 = tc.backend.jit(tc.backend.value_and_grad(f2))
    for _ in range(4):
        v, g = fj2(pt)
    assert fj2.stats["fast"] >= 1
    np.testing.assert_allclose(float(v), float(tc.backend.value_and_grad(f2)(pt)[0]), atol=1e-6)


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_tiled_pauli_sum_equals_the_flat_kernel_and_gives_the_energy(dt):
    """``tcmi_apply_pauli_sum_tiled`` (tile passes over arbitrary index bits, csrc/tcmi_hsum.hip) against the flat
    gather kernel ``tcmi_apply_pauli_sum`` on random few-body Pauli sums with X / Y / Z factors, per batch element
    weights, three passes with accumulation; Re <psi|lambda> returned by the passes = the weighted sum of the
    expectation values from the dense oracle (reference circuit.py:833-913, :899-902)."""
    import torch
    from tcmi import _lib
    from tcmi.executor import ATOMIC_COPIES, plan_pauli_passes
    from oracle import dense

    n, B = 16, 3
    cdt = torch.complex64 if dt == "complex64" else torch.complex128
    code = _lib.TCMI_C64 if dt == "complex64" else _lib.TCMI_C128
    T = int(_lib.lib().tcmi_pauli_sum_tile_bits(code))
    rng = np.random.default_rng(17)
    strings, rows = [], []
    for k in range(30):
        qs = sorted(rng.choice(n, size=int(rng.integers(1, 4)), replace=False).tolist())
        ps = [0] * n
        for q in qs:
            ps[q] = int(rng.integers(1, 4))
        if k < 4:                      # some diagonal strings
            ps = [3 if p else 0 for p in ps]
        strings.append(ps)
        xm = sum(1 << (n - 1 - q) for q in range(n) if ps[q] in (1, 2))
        zm = sum(1 << (n - 1 - q) for q in range(n) if ps[q] in (2, 3))
        rows.append((xm, zm, sum(1 for p in ps if p == 2), k))
    passes = plan_pauli_passes(n, rows, T, max_passes=8)
    assert passes is not None and len(passes) >= 2
    g = torch.Generator(device="cuda").manual_seed(5)
    psi = torch.randn(B, 2**n, dtype=cdt, device="cuda", generator=g)
    psi = psi / psi.norm(dim=1, keepdim=True)
    w = torch.from_numpy(rng.normal(size=(B, len(rows)))).cuda()
    st = torch.cuda.current_stream().cuda_stream
    # flat kernel (terms sorted by X mask)
    order = sorted(range(len(rows)), key=lambda k: rows[k][0])
    arr = np.array([[rows[k][0], rows[k][1], rows[k][2]] for k in order], dtype=np.int64).astype(np.uint32).view(np.int32)
    tdev = torch.from_numpy(arr.reshape(-1, 3).copy()).cuda()
    wf = w[:, order].contiguous()
    ref = torch.empty_like(psi)
    _lib.check(_lib.lib().tcmi_apply_pauli_sum(psi.data_ptr(), ref.data_ptr(), 2**n, B, n, tdev.data_ptr(), len(rows),
                                               wf.data_ptr(), wf.stride(0), code, st), "flat")
    out = torch.full_like(psi, float("nan"))
    dots = torch.zeros(B, ATOMIC_COPIES, dtype=torch.float64, device="cuda")
    for i, ps in enumerate(passes):
        tp = torch.tensor(ps["tilepos"], dtype=torch.int32, device="cuda")
        tr = torch.from_numpy(np.asarray(ps["rows"], dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 4).copy()).cuda()
        wp = w[:, ps["order"]].contiguous()
        _lib.check(_lib.lib().tcmi_apply_pauli_sum_tiled(psi.data_ptr(), out.data_ptr(), 2**n, B, n, tp.data_ptr(),
                                                         tr.data_ptr(), len(ps["order"]), ps["ndiag"], wp.data_ptr(),
                                                         wp.stride(0), int(i > 0), dots.data_ptr(), dots.stride(0), ATOMIC_COPIES, code,
                                                         st), "tiled")
    tol = 2e-6 if dt == "complex64" else 1e-13
    err = float((out - ref).abs().max())
    assert err < tol, err
    # energy: sum_t w_t <P_t> from the dense oracle
    e = dots.sum(1).cpu().numpy()
    for b in range(B):
        v = psi[b].cpu().numpy().astype(np.complex128)
        want = sum(float(w[b, k]) * dense.pauli_string_expectation(v, n, strings[k]).real for k in range(len(rows)))
        assert abs(e[b] - want) < (2e-5 if dt == "complex64" else 1e-11), (b, e[b], want)


def test_what_invalidates_a_jit_trace():
    """HipBackend.jit's contract: the host-side trace is keyed on tensor shapes / dtypes, on the
    VALUES of python scalar arguments, on the global dtype and contractor -- each of them re-traces -- and on nothing
    else: tensor values replay, a closure that changes after tracing is frozen (as under jax.jit)."""
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    K = tc.backend
    scale = {"v": 1.0}

    def energy(p, k):
        n = p.shape[-1]
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for i in range(n - 1):
            c.rzz(i, i + 1, theta=p[0, i])
        for i in range(n):
            c.rx(i, theta=p[1, i])
        return K.real(scale["v"] * k * c.expectation_ps(z=[0, 1]) + c.expectation_ps(x=[n - 1]))

    def ref(p, k):
        return K.value_and_grad(energy)(p, k)

    f = K.jit(K.value_and_grad(energy))
    rng = np.random.default_rng(0)
    p6 = [K.convert_to_tensor(rng.uniform(0, 3, [2, 6]).astype(np.float32)) for _ in range(4)]
    try:
        for p in p6:                      # call 1 traces and validates against the plain path, calls 2-4 replay
            v, g = f(p, 2)
            v0, g0 = ref(p, 2)
            assert abs(float(v) - float(v0)) < 1e-5 and float((g - g0).abs().max()) < 1e-5
        assert f.stats["fast"] == 3 and len(f.plans) == 1
        # a python scalar argument is static: another value is another trace, with the right result
        v, g = f(p6[0], 3)
        v0, g0 = ref(p6[0], 3)
        assert abs(float(v) - float(v0)) < 1e-5 and len(f.plans) == 2
        # another shape: another trace
        p8 = K.convert_to_tensor(rng.uniform(0, 3, [2, 8]).astype(np.float32))
        for _ in range(3):
            v, g = f(p8, 2)
        v0, g0 = ref(p8, 2)
        assert abs(float(v) - float(v0)) < 1e-5 and float((g - g0).abs().max()) < 1e-5 and len(f.plans) == 3
        # the global dtype is part of the key
        tc.set_dtype("complex128")
        p6d = K.convert_to_tensor(rng.uniform(0, 3, [2, 6]))
        v, g = f(p6d, 2)
        v0, g0 = ref(p6d, 2)
        assert abs(float(v) - float(v0)) < 1e-10 and len(f.plans) == 4
        tc.set_dtype("complex64")
        # keyword arguments: never traced
        slow = f.stats["slow"]
        f(p6[0], k=2)
        assert f.stats["slow"] == slow + 1 and len(f.plans) == 4
        # NOT part of the key: a closure that changes after the trace is frozen at its traced value (jax.jit's contract)
        fast = f.stats["fast"]
        v_before, _ = f(p6[1], 2)
        scale["v"] = 5.0
        v_after, _ = f(p6[1], 2)
        assert f.stats["fast"] == fast + 2 and abs(float(v_before) - float(v_after)) < 1e-7
        scale["v"] = 1.0
    finally:
        tc.set_dtype("complex64")
