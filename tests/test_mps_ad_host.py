"""Reverse-mode rules of tcmi/linalg.py (SVD, QR, RQ, GEMM) and value_and_grad through MPSCircuit, on CPU.
Only the *raw* device primitives are replaced by torch-CPU stand-ins (test infrastructure); the backward
rules, the truncation plumbing and the tensor-valued gates are the product code.  References:
``backends/jax_ops.py:18-150`` (AD-aware SVD / QR), ``tests/test_mpscircuit.py:437-497``."""

import numpy as np
import pytest
import torch

import tcmi as tc
from tcmi import linalg as LA

N = 8
D = 6


def _svd_full(mat):
    u, s, vh = torch.linalg.svd(mat, full_matrices=False)
    return u.contiguous(), s.contiguous(), vh.contiguous()


@pytest.fixture
def cpu_raw(monkeypatch):
    monkeypatch.setattr(LA, "_matmul_raw", lambda a, b: a @ b)
    monkeypatch.setattr(LA, "_site_gate_raw", lambda g, t: torch.einsum("ab,lbr->lar", g, t))
    monkeypatch.setattr(LA, "_gate_mix_raw", lambda t, g, L, R: torch.einsum(
        "xyab,labr->lxyr", g.reshape(2, 2, 2, 2), t.reshape(L, 2, 2, R)).reshape(-1))
    monkeypatch.setattr(LA, "_svd_full_raw", _svd_full)
    monkeypatch.setattr(LA, "_qr_raw", lambda m: tuple(torch.linalg.qr(m)))

    def svd_rows(mat, kmax, max_sv, max_err, relative, absorb):   # untracked truncating call
        from oracle import mps as omps
        u, s, vh, rest = omps.svd_trunc(mat.detach().numpy(), max_sv, max_err, relative)
        k = s.size
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        sv = np.concatenate([s.real, rest.real])
        if absorb == 1:
            u = u * s.reshape(1, -1)
        elif absorb == 2:
            vh = s.reshape(-1, 1) * vh
        uu = np.zeros((mat.shape[0], kmax), complex); uu[:, :k] = u
        vv = np.zeros((kmax, mat.shape[1]), complex); vv[:k] = vh
        return f(uu), f(sv), f(vv), torch.tensor([k], dtype=torch.int32), torch.zeros(1, dtype=torch.float64)

    monkeypatch.setattr(LA, "_svd_rows", svd_rows)
    tc.set_backend("hip")
    tc.set_dtype("complex128")
    yield
    tc.set_dtype("complex64")


def _rand(m, n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(m, n, dtype=torch.complex128, generator=g)


@pytest.mark.parametrize("shape", [(5, 5), (6, 4), (4, 7)])
@pytest.mark.parametrize("rule", [dict(), dict(max_singular_values=3), dict(max_truncation_err=0.8)])
@pytest.mark.parametrize("absorb", [0, 1, 2])
def test_svd_trunc_grad_matches_torch(cpu_raw, shape, rule, absorb):
    m, n = shape
    w1, w2 = _rand(m, m, 1), _rand(n, n, 2)

    def loss(u, s, vh, rest):
        k = s.shape[0]
        d = torch.arange(1, k + 1).to(u.dtype)
        if absorb == 1:     # u carries s: undo for the gauge-invariant pieces
            x = u @ vh
            y = torch.zeros(())
        elif absorb == 2:
            x = u @ vh
            y = ((u * d) @ u.mH * w1).real.sum()
        else:
            x = (u * s) @ vh
            y = ((u * d) @ u.mH * w1).real.sum() + ((vh.mH * d) @ vh * w2).imag.sum()
        return (x.abs() ** 3).sum() + y + (s.real ** 3).sum() + (rest.real ** 2).sum()

    a = _rand(m, n, 0).requires_grad_(True)
    (g,) = torch.autograd.grad(loss(*LA.svd_trunc(a, absorb=absorb, **rule)), a)

    b = a.detach().clone().requires_grad_(True)
    u, s, vh = torch.linalg.svd(b, full_matrices=False)
    from oracle import mps as omps
    kk = omps.svd_trunc(b.detach().numpy(), rule.get("max_singular_values"), rule.get("max_truncation_err"))[1].size
    sc = s.to(b.dtype)
    uk, sk, vk = u[:, :kk], sc[:kk], vh[:kk]
    if absorb == 1:
        uk = uk * sk
    elif absorb == 2:
        vk = sk[:, None] * vk
    (g_ref,) = torch.autograd.grad(loss(uk, sk, vk, sc[kk:]), b)
    np.testing.assert_allclose(g.numpy(), g_ref.numpy(), atol=1e-10)


@pytest.mark.parametrize("shape", [(5, 5), (7, 4), (4, 7), (16, 16), (20, 13)])
@pytest.mark.parametrize("which", ["qr", "rq"])
def test_qr_grad_matches_torch(cpu_raw, shape, which):
    m, n = shape
    a = _rand(m, n, 3).requires_grad_(True)
    k = min(m, n)

    def loss(x, y):          # invariant under x -> x D, y -> D^H y with D a diagonal phase
        d = torch.arange(1, k + 1).to(x.dtype)
        w = _rand(x.shape[0], y.shape[1], 4)
        return ((x @ y).abs() ** 3).sum() + ((x * d) @ x.mH).abs().pow(2).sum() + ((x * d) @ y * w).real.sum()

    if which == "qr":
        q, r = LA.qr(a)
        assert torch.diagonal(r).imag.abs().max() < 1e-12 and torch.diagonal(r).real.min() >= 0
        (g,) = torch.autograd.grad(loss(q, r), a)
        b = a.detach().clone().requires_grad_(True)
        (g_ref,) = torch.autograd.grad(loss(*torch.linalg.qr(b)), b)
    else:
        r, q = LA.rq(a)
        np.testing.assert_allclose((r @ q).detach().numpy(), a.detach().numpy(), atol=1e-12)
        loss2 = lambda x, y: loss(y.mH, x.mH)
        (g,) = torch.autograd.grad(loss2(r, q), a)
        b = a.detach().clone().requires_grad_(True)
        q2, r2 = torch.linalg.qr(b.mH)
        (g_ref,) = torch.autograd.grad(loss2(r2.mH, q2.mH), b)
    np.testing.assert_allclose(g.resolve_conj().numpy(), g_ref.resolve_conj().numpy(), atol=1e-9)


def test_triu_inv(cpu_raw):
    for n in (1, 2, 3, 8, 13, 32):
        r = torch.triu(_rand(n, n, n)) + 3 * torch.eye(n, dtype=torch.complex128)
        np.testing.assert_allclose((LA._triu_inv(r) @ r).numpy(), np.eye(n), atol=1e-10)


def _reproducible_unitary(n, param):
    """tests/test_mpscircuit.py:27-34."""
    e = 2 ** n
    A = torch.arange(e * e, device=param.device).reshape(e, e).to(torch.complex128)
    A = A + torch.sin(A) * param * 1j
    A = A - A.mH
    return torch.linalg.matrix_exp(A).reshape((2,) * (2 * n))


def _simulate(c, params):
    """tests/test_mpscircuit.py:37-55 (check=False)."""
    O1 = tc.gates.any(_reproducible_unitary(1, params[0]))
    O2 = tc.gates.any(_reproducible_unitary(2, params[1]))
    O3 = tc.gates.any(_reproducible_unitary(3, params[2]))
    c.H(0)
    for i in range(0, N - 1, 2):
        c.apply(O2.copy(), i, i + 1)
        c.apply(O1.copy(), i)
    c.apply(O3.copy(), int(N * 0.1), int(N * 0.5), int(N * 0.9))
    c.apply(O2.copy(), 1, N - 2)
    c.cz(2, 3)


def _check_directional(expec, atol, nontrivial=True):
    params = torch.ones(3, dtype=torch.complex128)
    vag = tc.backend.value_and_grad(expec)
    v, g = vag(params)
    np.testing.assert_allclose(float(expec(params)), float(v), atol=1e-10)
    dir_ = torch.tensor([1.0, 2.0, 3.0], dtype=torch.complex128)
    eps = 1e-6
    num = (expec(params + dir_ * eps) - expec(params - dir_ * eps)) / (2 * eps)
    np.testing.assert_allclose(float(num), float((g * dir_).sum().real), atol=atol)
    if nontrivial:
        assert abs(float(num)) > 1e-3


@pytest.mark.parametrize("split", [dict(max_singular_values=D), dict()])
def test_circuits_value_and_grad(cpu_raw, split):
    """tests/test_mpscircuit.py:437-466 (truncated) and the exact QR / RQ route."""
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(split))
        _simulate(mps, params)
        return tc.backend.real(mps.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4]))

    _check_directional(expec, 1e-6)


def test_simple_circuits_ad(cpu_raw):
    """tests/test_mpscircuit.py:469-497: parameterised named gates, a non-adjacent rzz."""
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(max_singular_values=D))
        mps.rx(0, theta=params[0])
        mps.cx(0, 1)
        mps.cx(1, 2)
        mps.ry(2, theta=params[1])
        mps.rzz(1, 3, theta=params[2])
        return tc.backend.real(mps.expectation_ps(**obs))

    obs = dict(x=[0, 2], z=[1])           # the reference's observable: identically zero on this circuit
    _check_directional(expec, 1e-6, nontrivial=False)
    obs = dict(z=[0, 3])
    _check_directional(expec, 1e-6)
    obs = dict(z=[2])
    _check_directional(expec, 1e-6)


def test_vmap_and_vvag_over_mps_circuits(cpu_raw):
    """backend.vmap / vvag of an MPSCircuit energy (the reference runs these under jax.vmap): the three primitives
    carry vmap rules, everything else is torch."""
    def expec(params):
        mps = tc.MPSCircuit(N, split=dict(max_singular_values=D))
        mps.rx(0, theta=params[0])
        mps.cx(0, 1)
        mps.cx(1, 2)
        mps.ry(2, theta=params[1])
        mps.rzz(1, 3, theta=params[2])
        return tc.backend.real(mps.expectation_ps(z=[0, 3])) + tc.backend.real(mps.expectation_ps(z=[2]))

    g0 = torch.Generator().manual_seed(0)
    ps = torch.rand(4, 3, generator=g0, dtype=torch.float64) * 2.0 + 0.2
    want = torch.stack([expec(p) for p in ps])
    got = tc.backend.vmap(expec)(ps)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-12)
    vs, gs = tc.backend.vvag(expec, argnums=0, vectorized_argnums=0)(ps)
    np.testing.assert_allclose(vs.numpy(), want.numpy(), atol=1e-12)
    vag = tc.backend.value_and_grad(expec)
    for i in range(ps.shape[0]):
        _, gi = vag(ps[i])
        np.testing.assert_allclose(gs[i].numpy(), gi.numpy(), atol=1e-9)
