"""GPU: reverse mode through the MPS kernels (tcmi_svd_trunc_batched / tcmi_qr_batched / tcmi_cgemm forward,
the rules of tcmi/linalg.py backward) against torch-CPU autograd of LAPACK factorizations and against
finite differences of MPSCircuit expectations (reference tests/test_mpscircuit.py:437-497)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import tcmi as tc
from tcmi import linalg as LA

from test_mps_ad_host import D, N, _rand, _simulate


@pytest.fixture
def c128():
    tc.set_backend("hip")
    tc.set_dtype("complex128")
    yield
    tc.set_dtype("complex64")


def _svd_loss(u, s, vh, rest, w1, w2):
    k = s.shape[0]
    d = torch.arange(1, k + 1, device=u.device).to(u.dtype)
    x = (u * s) @ vh
    return ((x.abs() ** 3).sum() + ((u * d) @ u.mH * w1).real.sum() + ((vh.mH * d) @ vh * w2).imag.sum()
            + (s.real ** 3).sum() + (rest.real ** 2).sum())


@pytest.mark.parametrize("shape", [(5, 5), (6, 4), (4, 7), (32, 32), (24, 40), (64, 48)])
@pytest.mark.parametrize("keep", [None, 3])
def test_svd_grad(c128, shape, keep):
    m, n = shape
    w1, w2 = _rand(m, m, 1), _rand(n, n, 2)
    a0 = _rand(m, n, 0)
    a = a0.cuda().requires_grad_(True)
    out = LA.svd_trunc(a, max_singular_values=keep)
    (g,) = torch.autograd.grad(_svd_loss(*out, w1.cuda(), w2.cuda()), a)
    assert LA.last_svd_status() == 0
    b = a0.clone().requires_grad_(True)
    u, s, vh = torch.linalg.svd(b, full_matrices=False)
    kk = min(m, n) if keep is None else keep
    sc = s.to(b.dtype)
    (g_ref,) = torch.autograd.grad(_svd_loss(u[:, :kk], sc[:kk], vh[:kk], sc[kk:], w1, w2), b)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=1e-8 * float(g_ref.abs().max()))


@pytest.mark.parametrize("shape", [(5, 5), (7, 4), (4, 7), (32, 32), (48, 20), (20, 48)])
@pytest.mark.parametrize("which", ["qr", "rq"])
def test_qr_grad(c128, shape, which):
    m, n = shape
    k = min(m, n)
    a0 = _rand(m, n, 3)

    def loss(x, y):
        d = torch.arange(1, k + 1, device=x.device).to(x.dtype)
        w = _rand(x.shape[0], y.shape[1], 4).to(x.device)
        return ((x @ y).abs() ** 3).sum() + ((x * d) @ x.mH).abs().pow(2).sum() + ((x * d) @ y * w).real.sum()

    a = a0.cuda().requires_grad_(True)
    b = a0.clone().requires_grad_(True)
    if which == "qr":
        q, r = LA.qr(a)
        rd = torch.diagonal(r).detach()
        assert float(rd.imag.abs().max()) < 1e-12 and float(rd.real.min()) >= 0
        (g,) = torch.autograd.grad(loss(q, r), a)
        (g_ref,) = torch.autograd.grad(loss(*torch.linalg.qr(b)), b)
    else:
        r, q = LA.rq(a)
        (g,) = torch.autograd.grad(loss(q.mH, r.mH), a)
        q2, r2 = torch.linalg.qr(b.mH)
        (g_ref,) = torch.autograd.grad(loss(q2, r2), b)
    np.testing.assert_allclose(g.resolve_conj().cpu().numpy(), g_ref.resolve_conj().numpy(),
                               atol=1e-9 * float(g_ref.abs().max()))


def test_matmul_grad(c128):
    a0, b0 = _rand(12, 20, 5), _rand(20, 9, 6)
    a, b = a0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
    ga, gb = torch.autograd.grad((LA.matmul(a, b).abs() ** 3).sum(), (a, b))
    a1, b1 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    ra, rb = torch.autograd.grad(((a1 @ b1).abs() ** 3).sum(), (a1, b1))
    np.testing.assert_allclose(ga.cpu().numpy(), ra.numpy(), atol=1e-9)
    np.testing.assert_allclose(gb.cpu().numpy(), rb.numpy(), atol=1e-9)


def _check_directional(expec, atol, nontrivial=True):
    params = torch.ones(3, dtype=torch.complex128, device="cuda")
    v, g = tc.backend.value_and_grad(expec)(params)
    np.testing.assert_allclose(float(expec(params)), float(v), atol=1e-10)
    dir_ = torch.tensor([1.0, 2.0, 3.0], dtype=torch.complex128, device="cuda")
    eps = 1e-6
    num = (expec(params + dir_ * eps) - expec(params - dir_ * eps)) / (2 * eps)
    np.testing.assert_allclose(float(num), float((g * dir_).sum().real), atol=atol)
    if nontrivial:
        assert abs(float(num)) > 1e-3
    return float(v)


@pytest.mark.parametrize("split", [dict(max_singular_values=D), dict()])
def test_circuits_value_and_grad(c128, split):
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(split))
        _simulate(mps, params)
        return tc.backend.real(mps.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4]))

    v = _check_directional(expec, 1e-6)
    # the untracked (kernel-truncating) route gives the same value
    with torch.no_grad():
        np.testing.assert_allclose(float(expec(torch.ones(3, dtype=torch.complex128, device="cuda"))), v, atol=1e-10)


def test_simple_circuits_ad(c128):
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(max_singular_values=D))
        mps.rx(0, theta=params[0])
        mps.cx(0, 1)
        mps.cx(1, 2)
        mps.ry(2, theta=params[1])
        mps.rzz(1, 3, theta=params[2])
        return tc.backend.real(mps.expectation_ps(**obs))

    obs = dict(x=[0, 2], z=[1])
    _check_directional(expec, 1e-6, nontrivial=False)
    obs = dict(z=[0, 3])
    _check_directional(expec, 1e-6)
    obs = dict(z=[2])
    _check_directional(expec, 1e-6)


def test_vqe_like_energy_grad_complex64():
    """A layered rzz / rx ansatz on 10 sites at bond dimension 8, single precision: value_and_grad against the
    dense state-vector path of the same framework."""
    tc.set_backend("hip")
    tc.set_dtype("complex64")
    n, layers = 10, 2

    def build(c, p):
        for i in range(n):
            c.h(i)
        for l in range(layers):
            for i in range(n - 1):
                c.rzz(i, i + 1, theta=p[l, 0, i])
            for i in range(n):
                c.rx(i, theta=p[l, 1, i])

    def e_mps(p):
        c = tc.MPSCircuit(n, split=dict(max_singular_values=16))
        build(c, p)
        e = 0.0
        for i in range(n - 1):
            e = e + tc.backend.real(c.expectation_ps(z=[i, i + 1]))
        for i in range(n):
            e = e - tc.backend.real(c.expectation_ps(x=[i]))
        return e

    def e_sv(p):
        c = tc.Circuit(n)
        build(c, p)
        e = 0.0
        for i in range(n - 1):
            e = e + tc.backend.real(c.expectation_ps(z=[i, i + 1]))
        for i in range(n):
            e = e - tc.backend.real(c.expectation_ps(x=[i]))
        return e

    g0 = torch.Generator().manual_seed(0)
    p = (torch.rand(layers, 2, n, generator=g0) * 0.8 + 0.1).cuda()
    v1, g1 = tc.backend.value_and_grad(e_mps)(p)
    v2, g2 = tc.backend.value_and_grad(e_sv)(p)
    np.testing.assert_allclose(float(v1), float(v2), atol=2e-4)
    np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), atol=2e-3)


def test_vmap_and_vvag_over_mps_circuits(c128):
    """backend.vmap / vvag of an MPSCircuit energy through the real kernels: 2-D GEMMs, SVDs and QRs of the vmapped
    chains each go out as ONE batched launch (the ``batch`` argument of the C ABI), so four chains issue as many SVD /
    QR launches as one; value_and_grad of a sum over vmapped chains (grad outside vmap) takes the batched backward
    rules."""
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(max_singular_values=D))
        mps.rx(0, theta=params[0])
        mps.cx(0, 1)
        mps.cx(1, 2)
        mps.ry(2, theta=params[1])
        mps.rzz(1, 3, theta=params[2])
        return tc.backend.real(mps.expectation_ps(z=[0, 3])) + tc.backend.real(mps.expectation_ps(z=[2]))

    g0 = torch.Generator().manual_seed(0)
    ps = (torch.rand(4, 3, generator=g0, dtype=torch.float64) * 2.0 + 0.2).cuda()
    want = torch.stack([expec(p) for p in ps])
    from tcmi import executor as EX

    def launches(fn):
        EX.EVENT_LOG = []
        try:
            out = fn()
            torch.cuda.synchronize()
            cnt = {}
            for tag, *_ in EX.EVENT_LOG:
                cnt[tag] = cnt.get(tag, 0) + 1
        finally:
            EX.EVENT_LOG = None
        return out, cnt

    got, n4 = launches(lambda: tc.backend.vmap(expec)(ps))
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=1e-11)
    _, n1 = launches(lambda: tc.backend.vmap(expec)(ps[:1]))
    assert n4.get("mps_svd", 0) > 0 and n4.get("mps_svd") == n1.get("mps_svd"), (n4, n1)
    assert n4.get("mps_qr", 0) == n1.get("mps_qr", 0), (n4, n1)
    # grad outside vmap: the decompositions record stacked matrices, their backward rules take the batch dimension
    tot = lambda q: tc.backend.sum(tc.backend.vmap(expec)(q))
    vt, gt = tc.backend.value_and_grad(tot)(ps)
    np.testing.assert_allclose(float(vt), float(want.sum()), atol=1e-10)
    for i in range(ps.shape[0]):
        _, gi = tc.backend.value_and_grad(expec)(ps[i])
        np.testing.assert_allclose(gt[i].cpu().numpy(), gi.cpu().numpy(), atol=1e-8)
    vs, gs = tc.backend.vvag(expec, argnums=0, vectorized_argnums=0)(ps)
    np.testing.assert_allclose(vs.cpu().numpy(), want.cpu().numpy(), atol=1e-11)
    vag = tc.backend.value_and_grad(expec)
    for i in range(ps.shape[0]):
        _, gi = vag(ps[i])
        np.testing.assert_allclose(gs[i].cpu().numpy(), gi.cpu().numpy(), atol=1e-8)
