"""GPU parity of the MPS / TEBD row (SURVEY §8a last row, config 5) through the C ABI:
tcmi_svd_trunc_batched / tcmi_qr_batched / tcmi_mps_gate_mix / tcmi_cgemm vs numpy, and tcmi.MPSCircuit
vs oracle/mps.py + the reference's KAT (tests/test_mpscircuit.py:380)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import tcmi as tc
from tcmi import linalg as LA
from oracle import mps as omps

from test_oracle_mps import D, N, dense_state, gate_list

TOL = {"complex64": 2e-5, "complex128": 1e-11}


def _rand(rng, m, n, dt, rank=None):
    a = rng.normal(size=(m, n)) + 1j * rng.normal(size=(m, n))
    if rank is not None:
        a = (rng.normal(size=(m, rank)) + 1j * rng.normal(size=(m, rank))) @ (
            rng.normal(size=(rank, n)) + 1j * rng.normal(size=(rank, n)))
    return a.astype(dt)


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
@pytest.mark.parametrize("shape", [(2, 2), (4, 4), (3, 5), (5, 3), (16, 16), (17, 33), (64, 40), (128, 128),
                                   (256, 256), (100, 300)])
def test_svd_matches_lapack(dt, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    a = _rand(rng, *shape, dt)
    u, s, vh, rest = LA.svd_trunc(torch.from_numpy(a).cuda())
    u, s, vh = u.cpu().numpy(), s.cpu().numpy(), vh.cpu().numpy()
    assert rest.numel() == 0
    s_ref = np.linalg.svd(a.astype(np.complex128), compute_uv=False)
    tol = TOL[dt] * max(1.0, s_ref[0])
    np.testing.assert_allclose(s.real, s_ref, atol=tol)
    assert np.all(np.diff(s.real) <= 1e-6 * s_ref[0])
    np.testing.assert_allclose((u * s) @ vh, a, atol=4 * tol)
    k = min(shape)
    np.testing.assert_allclose(u.conj().T @ u, np.eye(k), atol=4 * TOL[dt] * 10)
    np.testing.assert_allclose(vh @ vh.conj().T, np.eye(k), atol=4 * TOL[dt] * 10)
    assert LA.last_svd_status() == 0


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_svd_truncation_rule_and_absorb(dt):
    rng = np.random.default_rng(5)
    a = _rand(rng, 48, 64, dt)
    ag = torch.from_numpy(a).cuda()
    for kw in (dict(max_singular_values=10), dict(max_truncation_err=3.0), dict(max_truncation_err=0.2, relative=True),
               dict(max_singular_values=20, max_truncation_err=1.0)):
        u_o, s_o, vh_o, rest_o = omps.svd_trunc(a.astype(np.complex128), kw.get("max_singular_values"),
                                                kw.get("max_truncation_err"), kw.get("relative", False))
        for absorb in (0, 1, 2):
            u, s, vh, rest = LA.svd_trunc(ag, absorb=absorb, **kw)
            assert s.numel() == s_o.size and rest.numel() == rest_o.size
            np.testing.assert_allclose(s.cpu().numpy().real, s_o.real, atol=TOL[dt] * 20)
            np.testing.assert_allclose(rest.cpu().numpy().real, rest_o.real, atol=TOL[dt] * 20)
            rec = u.cpu().numpy() @ vh.cpu().numpy() if absorb else (u.cpu().numpy() * s.cpu().numpy()) @ vh.cpu().numpy()
            np.testing.assert_allclose(rec, (u_o * s_o) @ vh_o, atol=TOL[dt] * 100)


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_svd_tiny_rows_stay_finite(dt):
    """Rows far below the float range of the rotation chain (truncated TEBD tensors carry singular values
    down to 1e-20 and below): no NaN / inf, large singular values unaffected."""
    rng = np.random.default_rng(3)
    a = _rand(rng, 64, 64, dt)
    scale = np.logspace(0, -30 if dt == "complex64" else -200, 64)
    a = (a * scale[:, None]).astype(dt)
    u, s, vh, _ = LA.svd_trunc(torch.from_numpy(a).cuda())
    assert bool(torch.isfinite(u.abs()).all()) and bool(torch.isfinite(vh.abs()).all()) and bool(torch.isfinite(s.abs()).all())
    s_ref = np.linalg.svd(a.astype(np.complex128), compute_uv=False)
    np.testing.assert_allclose(s.cpu().numpy().real[:8], s_ref[:8], rtol=TOL[dt] * 20)
    assert LA.last_svd_status() == 0


@pytest.mark.parametrize("n", [128, 256])
def test_svd_factors_stay_orthonormal_for_a_spectrum_graded_over_six_decades(n):
    """complex64, singular values from 1 down to 1e-6 (TEBD-like decay): u and vh are isometries to a few 1e-6 -- also
    the rows that belong to the smallest singular values, which an absolute cut-off in the rotation test used to leave
    orthogonal only to 7e-4 -- and the values match LAPACK to float accuracy relative to the largest."""
    rng = np.random.default_rng(21)
    q1, _ = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))
    q2, _ = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))
    sv = np.logspace(0, -6, n)
    a = ((q1 * sv) @ q2).astype(np.complex64)
    u, s, vh, _ = LA.svd_trunc(torch.from_numpy(a).cuda())
    u, s, vh = u.cpu().numpy(), s.cpu().numpy().real, vh.cpu().numpy()
    eye = np.eye(n)
    assert np.abs(u.conj().T @ u - eye).max() < 2e-5
    assert np.abs(vh @ vh.conj().T - eye).max() < 2e-5
    np.testing.assert_allclose(s, np.linalg.svd(a.astype(np.complex128), compute_uv=False), atol=5e-6)
    np.testing.assert_allclose((u * s) @ vh, a, atol=5e-6)
    assert LA.last_svd_status() == 0


def test_svd_rank_deficient_and_zero():
    rng = np.random.default_rng(9)
    a = _rand(rng, 32, 32, "complex128", rank=5)
    u, s, vh, _ = LA.svd_trunc(torch.from_numpy(a).cuda())
    s = s.cpu().numpy().real
    assert np.all(s[5:] < 1e-10 * s[0])
    np.testing.assert_allclose((u.cpu().numpy() * s) @ vh.cpu().numpy(), a, atol=1e-10)
    np.testing.assert_allclose(u.cpu().numpy().conj().T @ u.cpu().numpy(), np.eye(32), atol=1e-10)
    z = torch.zeros((4, 6), dtype=torch.complex64, device="cuda")
    u, s, vh, _ = LA.svd_trunc(z)
    assert float(s.abs().max()) == 0.0 and bool(torch.isfinite(vh.abs()).all())


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
@pytest.mark.parametrize("shape", [(2, 1), (2, 2), (4, 2), (6, 9), (64, 32), (256, 128), (130, 70)])
def test_qr_and_rq(dt, shape):
    rng = np.random.default_rng(shape[0] + 31 * shape[1])
    a = _rand(rng, *shape, dt)
    ag = torch.from_numpy(a).cuda()
    q, r = LA.qr(ag)
    q, r = q.cpu().numpy(), r.cpu().numpy()
    k = min(shape)
    tol = TOL[dt] * 20
    np.testing.assert_allclose(q @ r, a, atol=tol * np.abs(a).max() * 4)
    np.testing.assert_allclose(q.conj().T @ q, np.eye(k), atol=tol)
    assert np.abs(np.tril(r, -1)).max() == 0
    rr, qq = LA.rq(ag)
    rr, qq = rr.cpu().numpy(), qq.cpu().numpy()
    np.testing.assert_allclose(rr @ qq, a, atol=tol * np.abs(a).max() * 4)
    np.testing.assert_allclose(qq @ qq.conj().T, np.eye(k), atol=tol)


def test_qr_rank_deficient_isometry():
    a = np.zeros((8, 4), dtype=np.complex128)
    a[0, 0] = 1.0
    a[:, 2] = a[:, 0] * 2
    q, r = LA.qr(torch.from_numpy(a).cuda())
    q, r = q.cpu().numpy(), r.cpu().numpy()
    np.testing.assert_allclose(q.conj().T @ q, np.eye(4), atol=1e-13)
    np.testing.assert_allclose(q @ r, a, atol=1e-13)


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_gate_mix_and_site_gate(dt):
    rng = np.random.default_rng(3)
    L, R = 7, 13
    t = (rng.normal(size=(L, 2, 2, R)) + 1j * rng.normal(size=(L, 2, 2, R))).astype(dt)
    g = (rng.normal(size=(2, 2, 2, 2)) + 1j * rng.normal(size=(2, 2, 2, 2))).astype(dt)
    out = LA.gate_mix(torch.from_numpy(t).cuda().reshape(-1), torch.from_numpy(g).cuda().reshape(-1), L, R)
    np.testing.assert_allclose(out.cpu().numpy().reshape(L, 2, 2, R), np.einsum("xyab,labr->lxyr", g, t),
                               atol=TOL[dt] * 10)
    a = (rng.normal(size=(L, 2, R)) + 1j * rng.normal(size=(L, 2, R))).astype(dt)
    g1 = (rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2))).astype(dt)
    o1 = LA.site_gate(torch.from_numpy(g1).cuda(), torch.from_numpy(a).cuda())
    np.testing.assert_allclose(o1.cpu().numpy(), np.einsum("ab,lbr->lar", g1, a), atol=TOL[dt] * 10)


def _run(ops, split=None):
    m = tc.MPSCircuit(N, split=split)
    for g, idx in ops:
        m.apply(tc.gates.Gate(np.asarray(g)), *idx)
    return m


def test_reference_truncation_kat_on_gpu():
    """reference tests/test_mpscircuit.py:380 (N=8, D=6, complex128)."""
    tc.set_dtype("complex128")
    try:
        ops = gate_list()
        w_c = dense_state(ops)
        m = _run(ops, tc.cons.split_rules(max_singular_values=D))
        m.normalize()
        real_fid = abs(np.vdot(m.wavefunction().cpu().numpy(), w_c)) ** 2
        np.testing.assert_allclose(real_fid, 0.902663090851, atol=1e-5)
        np.testing.assert_allclose(float(m._fidelity), 0.910305380327, atol=1e-5)
        assert float(m._mps.check_canonical()) < 1e-10
        ex = _run(ops)
        np.testing.assert_allclose(ex.wavefunction().cpu().numpy(), w_c, atol=1e-10)
        assert float(ex._mps.check_canonical()) < 1e-10
        e = ex.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4])
        o = omps.MPSCircuit(N)
        for g, idx in ops:
            o.apply(g, *idx)
        np.testing.assert_allclose(complex(e), o.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4]), atol=1e-9)
    finally:
        tc.set_dtype("complex64")


@pytest.mark.parametrize("dt,chi", [("complex64", 8), ("complex128", 8), ("complex64", None)])
def test_tebd_sweeps_match_oracle(dt, chi):
    """Config-5 shaped workload at a size the oracle finishes in seconds: brickwork + left-to-right sweeps of
    random SU(4) gates with max_singular_values = chi, compared with oracle/mps.py gate by gate."""
    from oracle import gates as OG

    n = 12
    tc.set_dtype(dt)
    try:
        split = tc.cons.split_rules(max_singular_values=chi) if chi else None
        osplit = omps.split_rules(max_singular_values=chi) if chi else None
        m = tc.MPSCircuit(n, split=split)
        o = omps.MPSCircuit(n, split=osplit)
        k = 0
        for sweep in range(3):
            for i in range(n - 1):
                g = OG.random_two_qubit_gate(1000 + k)
                k += 1
                m.apply(tc.gates.Gate(g.reshape(2, 2, 2, 2)), i, i + 1)
                o.apply(g, i, i + 1)
            m.rx(sweep, theta=0.3 + sweep)
            o.rx(sweep, theta=0.3 + sweep)
        assert m.get_bond_dimensions() == o.get_bond_dimensions()
        tol = 5e-5 if dt == "complex64" else 1e-9
        # compare gauge-invariant quantities: fidelity estimate, state overlap, norm
        wm = m.wavefunction().cpu().numpy().astype(np.complex128)
        wo = o.wavefunction()
        np.testing.assert_allclose(float(m._fidelity), o._fidelity, atol=tol * 10)
        np.testing.assert_allclose(abs(np.vdot(wm, wo)), np.linalg.norm(wo) ** 2, atol=tol * 10)
        np.testing.assert_allclose(float(m.get_norm()), o.get_norm(), atol=tol * 10)
    finally:
        tc.set_dtype("complex64")


def test_backend_svd_qr_rq_route_to_hip_kernels():
    """backend.svd / qr / rq (tensornetwork backend signatures with pivot_axis; reference call sites
    mps_base.py:123-168) resolve to the HIP kernels."""
    tc.set_dtype("complex128")
    try:
        rng = np.random.default_rng(0)
        a = (rng.normal(size=(3, 4, 5, 2)) + 1j * rng.normal(size=(3, 4, 5, 2)))
        t = tc.backend.convert_to_tensor(a)
        u, s, vh, rest = tc.backend.svd(t, pivot_axis=2, max_singular_values=7)
        assert tuple(u.shape) == (3, 4, 7) and tuple(vh.shape) == (7, 5, 2) and rest.numel() == 3
        uo, so, vho, resto = omps.svd_trunc(a.reshape(12, 10), 7)
        np.testing.assert_allclose(s.cpu().numpy().real, so.real, atol=1e-10)
        rec = np.tensordot(u.cpu().numpy() * s.cpu().numpy(), vh.cpu().numpy(), axes=(2, 0))
        np.testing.assert_allclose(rec.reshape(12, 10), (uo * so) @ vho, atol=1e-9)
        q, r = tc.backend.qr(t, pivot_axis=2, non_negative_diagonal=True)
        np.testing.assert_allclose(np.tensordot(q.cpu().numpy(), r.cpu().numpy(), axes=(2, 0)), a, atol=1e-10)
        d = np.diagonal(r.cpu().numpy().reshape(10, 10))
        assert np.all(np.abs(d.imag) < 1e-12) and np.all(d.real >= 0)
        r2, q2 = tc.backend.rq(t, pivot_axis=2)
        np.testing.assert_allclose(np.tensordot(r2.cpu().numpy(), q2.cpu().numpy(), axes=(2, 0)), a, atol=1e-10)
        x = rng.normal(size=(6, 4)).astype(np.float64)
        ur, sr, vhr, _ = tc.backend.svd(tc.backend.convert_to_tensor(x), pivot_axis=1)
        np.testing.assert_allclose(sr.cpu().numpy(), np.linalg.svd(x, compute_uv=False), atol=1e-10)
    finally:
        tc.set_dtype("complex64")


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_batched_svd_and_qr_through_the_abi(dt):
    """The batch argument of tcmi_svd_trunc_batched / tcmi_qr_batched (independent matrices along
    blockIdx.y, one barrier counter each), called through ctypes directly."""
    from tcmi import _lib

    lib = _lib.lib()
    code = _lib.TCMI_C64 if dt == "complex64" else _lib.TCMI_C128
    rdt = torch.float32 if dt == "complex64" else torch.float64
    rng = np.random.default_rng(11)
    B, m, n, kmax = 5, 24, 40, 10
    a = (rng.normal(size=(B, m, n)) + 1j * rng.normal(size=(B, m, n))).astype(dt)
    ag = torch.from_numpy(a).cuda()
    u = torch.empty((B, m, kmax), dtype=ag.dtype, device="cuda")
    s = torch.empty((B, m), dtype=rdt, device="cuda")
    vh = torch.empty((B, kmax, n), dtype=ag.dtype, device="cuda")
    keep = torch.empty((B,), dtype=torch.int32, device="cuda")
    tw2 = torch.empty((B,), dtype=rdt, device="cuda")
    nbytes = lib.tcmi_svd_work_bytes(m, n, B, code)
    work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.tcmi_svd_trunc_batched(ag.data_ptr(), u.data_ptr(), s.data_ptr(), vh.data_ptr(), keep.data_ptr(),
                                          tw2.data_ptr(), m, n, kmax, B, kmax, -1.0, 0, 1, 0, work.data_ptr(), nbytes,
                                          code, st), "svd")
    torch.cuda.synchronize()
    tol = TOL[dt] * 50
    for b in range(B):
        uo, so, vho, rest = omps.svd_trunc(a[b].astype(np.complex128), kmax)
        np.testing.assert_allclose(s[b].cpu().numpy(), np.linalg.svd(a[b].astype(np.complex128), compute_uv=False),
                                   atol=tol)
        np.testing.assert_allclose(u[b].cpu().numpy() @ vh[b].cpu().numpy(), (uo * so) @ vho, atol=tol * 4)
        assert int(keep[b]) == kmax
        np.testing.assert_allclose(float(tw2[b]), float(np.sum(np.abs(rest) ** 2)), rtol=1e-3, atol=tol)
    mq, nq = 18, 7
    aq = (rng.normal(size=(B, mq, nq)) + 1j * rng.normal(size=(B, mq, nq))).astype(dt)
    aqg = torch.from_numpy(aq).cuda()
    q = torch.empty((B, mq, nq), dtype=aqg.dtype, device="cuda")
    r = torch.empty((B, nq, nq), dtype=aqg.dtype, device="cuda")
    nb2 = lib.tcmi_qr_work_bytes(mq, nq, B, code)
    work2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    _lib.check(lib.tcmi_qr_batched(aqg.data_ptr(), q.data_ptr(), r.data_ptr(), mq, nq, B, work2.data_ptr(), nb2, code, st),
               "qr")
    torch.cuda.synchronize()
    for b in range(B):
        qq, rr = q[b].cpu().numpy(), r[b].cpu().numpy()
        np.testing.assert_allclose(qq @ rr, aq[b], atol=tol)
        np.testing.assert_allclose(qq.conj().T @ qq, np.eye(nq), atol=tol)


def test_measure_sample_rdm_like_the_reference_mps_test():
    """reference tests/test_mpscircuit.py:257-330 (do_test_measure / do_test_sample / do_test_rdm): with the
    same ``status`` the MPS and the state-vector circuit measure the same bits with the same probability;
    reduced density matrices (contiguous, non-contiguous, caller order) equal the dense ones."""
    tc.set_dtype("complex128")
    try:
        ops = gate_list()
        w_c = dense_state(ops)
        m = _run(ops)
        c = tc.Circuit(N)
        for g, idx in ops:
            c.any(*idx, unitary=np.asarray(g).reshape((2,) * (2 * len(idx))))
        index = [6, 5, 2, 1]
        status = tc.backend.convert_to_tensor(np.array([0.1, 0.3, 0.7, 0.9]))
        rc = c.measure(*index, with_prob=True, status=status)
        rm = m.measure(*index, with_prob=True, status=status)
        ro = omps.MPSCircuit(N)
        for g, idx in ops:
            ro.apply(g, *idx)
        so, po = ro.measure(*index, with_prob=True, status=[0.1, 0.3, 0.7, 0.9])
        np.testing.assert_allclose(rm[0].cpu().numpy(), rc[0].cpu().numpy(), atol=1e-8)
        np.testing.assert_allclose(float(rm[1]), float(rc[1]), atol=1e-8)
        np.testing.assert_allclose(rm[0].cpu().numpy(), so, atol=1e-8)
        np.testing.assert_allclose(float(rm[1]), po, atol=1e-8)
        s = m.sample(batch=10, format="sample_bin")
        assert len(s) == 10 and len(s[0]) == N
        for keep in ([1, 2, 3], [1, 3, 5], [3, 1]):
            rho = m.reduced_density_matrix(keep).cpu().numpy()
            np.testing.assert_allclose(rho, ro.reduced_density_matrix(keep), atol=1e-10)
        rho13 = m.reduced_density_matrix([1, 3]).cpu().numpy().reshape(2, 2, 2, 2)
        rho31 = m.reduced_density_matrix([3, 1]).cpu().numpy().reshape(2, 2, 2, 2)
        np.testing.assert_allclose(rho31, rho13.transpose(1, 0, 3, 2), atol=1e-12)
        rho_c = tc.quantum.reduced_density_matrix(c.wavefunction(), [0, 4, 5, 6, 7]).cpu().numpy()
        np.testing.assert_allclose(m.reduced_density_matrix([1, 2, 3]).cpu().numpy(), rho_c, atol=1e-10)
        # state sampling: frequencies follow |psi|^2
        cnt = c.sample(batch=4000, allow_state=True, format="count_vector").cpu().numpy()
        p = np.abs(w_c) ** 2
        assert np.abs(cnt / 4000.0 - p).max() < 0.05
        assert len(c.sample(batch=3, allow_state=False, format="sample_bin")) == 3
    finally:
        tc.set_dtype("complex64")


def test_svd_qr_fuzz_shapes_scales_and_ranks():
    """Randomised sweep over shapes (1..130), scales (1e-12..1e12), ranks and dtypes: the kernels stay finite,
    the barrier never times out, U S Vh and Q R reconstruct the input."""
    rng = np.random.default_rng(2024)
    for trial in range(60):
        dt = "complex64" if trial % 2 == 0 else "complex128"
        m, n = int(rng.integers(1, 131)), int(rng.integers(1, 131))
        rank = None if trial % 3 else int(rng.integers(1, min(m, n) + 1))
        scale = 10.0 ** rng.uniform(-12, 12) if dt == "complex128" else 10.0 ** rng.uniform(-6, 6)
        a = (_rand(rng, m, n, "complex128", rank=rank) * scale).astype(dt)
        ag = torch.from_numpy(a).cuda()
        u, s, vh, _ = LA.svd_trunc(ag)
        assert LA.last_svd_status() == 0, (m, n, dt)
        un, sn, vn = u.cpu().numpy(), s.cpu().numpy().real, vh.cpu().numpy()
        assert np.isfinite(un).all() and np.isfinite(sn).all() and np.isfinite(vn).all(), (m, n, dt, scale)
        amax = max(np.abs(a).max(), 1e-300)
        tol = (2e-5 if dt == "complex64" else 1e-11) * amax * max(m, n) ** 0.5
        np.testing.assert_allclose((un * sn) @ vn, a, atol=tol, err_msg=f"svd {m}x{n} {dt} rank={rank} scale={scale:.1e}")
        assert np.all(np.diff(sn) <= 1e-5 * max(sn[0], 1e-300))
        q, r = LA.qr(ag)
        qn, rn = q.cpu().numpy(), r.cpu().numpy()
        np.testing.assert_allclose(qn @ rn, a, atol=tol, err_msg=f"qr {m}x{n} {dt}")
        k = min(m, n)
        np.testing.assert_allclose(qn.conj().T @ qn, np.eye(k), atol=5e-5 if dt == "complex64" else 1e-11)


def test_qr_preconditioned_jacobi_needs_ten_sweeps_whatever_the_grading(monkeypatch):
    """linalg.SVD_PRECONDITION: A^H = Q R (two register-resident panels, block Gram-Schmidt with re-orthogonalisation),
    Jacobi on R, factors mapped back.  On a 256 x 256 complex64 matrix with a spectrum graded over six decades the plain
    kernel needs > 20 sweeps, the preconditioned one <= 12; both reconstruct the matrix, give orthonormal factors and
    the singular values of LAPACK; the truncated / absorbed forms agree with the plain path."""
    import torch
    import tcmi as tc
    from tcmi import linalg as LA

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    rng = np.random.default_rng(3)

    def haar(k):
        z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k))
        q, r = np.linalg.qr(z)
        return q * (np.diag(r) / abs(np.diag(r)))

    m = 256
    a_np = ((haar(m) * np.logspace(0, -6, m)) @ haar(m)).astype(np.complex64)
    a = torch.from_numpy(a_np).cuda()
    ref = np.linalg.svd(a_np.astype(np.complex128), compute_uv=False)
    eye = torch.eye(m, device="cuda")
    sweeps = {}
    for pre in (False, True):
        monkeypatch.setattr(LA, "SVD_PRECONDITION", pre)
        u, s, vh, _ = LA.svd_trunc(a, max_singular_values=m)
        sweeps[pre] = LA.last_svd_sweeps(a.device)
        sr = s.real.cpu().numpy()
        assert float(((u * s.reshape(1, -1)) @ vh - a).abs().max()) < 2e-5
        assert float((u.conj().t() @ u - eye).abs().max()) < 2e-5 and float((vh @ vh.conj().t() - eye).abs().max()) < 2e-5
        assert np.abs(sr - ref).max() < 1e-5 * ref[0]
        if pre:      # the small singular values keep their RELATIVE accuracy down to eps * sigma_max
            big = ref > 1e-4
            assert np.abs(sr[big] / ref[big] - 1).max() < 1e-3
    assert sweeps[False] > 20 and sweeps[True] <= 12, sweeps
    # truncation + absorption (what one TEBD bond update asks for), rectangular both ways
    for shape, absorb in (((256, 256), 1), ((128, 256), 2), ((256, 128), 1), ((64, 200), 0)):
        b = torch.from_numpy((rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)).cuda()
        k = min(shape) // 2
        monkeypatch.setattr(LA, "SVD_PRECONDITION", False)
        u0, s0, vh0, r0 = LA.svd_trunc(b, max_singular_values=k, absorb=absorb)
        monkeypatch.setattr(LA, "SVD_PRECONDITION", True)
        u1, s1, vh1, r1 = LA.svd_trunc(b, max_singular_values=k, absorb=absorb)
        assert u1.shape == u0.shape and vh1.shape == vh0.shape and s1.shape == s0.shape
        assert float((s1 - s0).abs().max()) < 1e-4 * float(s0.abs().max())
        p0 = u0 @ vh0 if absorb else (u0 * s0.reshape(1, -1)) @ vh0
        p1 = u1 @ vh1 if absorb else (u1 * s1.reshape(1, -1)) @ vh1
        assert float((p1 - p0).abs().max()) < 1e-4 * float(p0.abs().max())
        assert abs(float(r1._tcmi_tw2[0]) - float(r0._tcmi_tw2[0])) < 1e-4 * max(1.0, float(r0._tcmi_tw2[0]))



# ---- gates on three or more sites: the block update (tcmi/mpscircuit.py) against the oracle's MPO route -----------------
def _haar(k, seed):
    from scipy.stats import unitary_group

    return unitary_group.rvs(2**k, random_state=seed).reshape((2,) * (2 * k))


def _prepared(tcm, n, seed, split=None):
    """The same entangling prelude on a product MPSCircuit and on the oracle's: nearest-neighbour Haar gates, two layers."""
    ops = []
    k = 0
    for layer in range(2):
        for i in range(layer % 2, n - 1, 2):
            g = _haar(2, seed + k)
            k += 1
            ops.append((g, (i, i + 1)))
    return ops


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_three_and_four_qubit_gates_on_non_adjacent_sites_match_the_oracle(dt):
    """3- and 4-qubit gates on scattered, unsorted sites: without truncation the state equals oracle.mps (which follows the
    reference's MPO route, mpscircuit.py:386-668) and the dense oracle; with max_singular_values the block update cuts the
    same bonds in the same order as the reference's compression sweep, so the truncated states agree too."""
    from oracle import dense

    tc.set_backend("hip")
    tc.set_dtype(dt)
    try:
        n = 9
        cdt = np.complex64 if dt == "complex64" else np.complex128
        big = [(_haar(3, 11), (1, 3, 4)), (_haar(3, 12), (6, 2, 4)), (_haar(4, 13), (0, 2, 5, 8)), (_haar(3, 14), (7, 8, 5))]
        for split, tol in ((None, 3e-5 if dt == "complex64" else 1e-10),
                           (dict(max_singular_values=6), 2e-4 if dt == "complex64" else 1e-8)):
            m = tc.MPSCircuit(n, split=tc.cons.split_rules(**split) if split else None)
            o = omps.MPSCircuit(n, split=omps.split_rules(**split) if split else None)
            ops = _prepared(None, n, 100) + big + _prepared(None, n, 200)
            for g, idx in ops:
                m.apply(tc.gates.Gate(g.astype(cdt)), *idx)
                o.apply(g.astype(np.complex128), *idx)
            got = m.wavefunction().reshape(-1).cpu().numpy()
            want = np.asarray(o.wavefunction()).reshape(-1)
            assert np.abs(got - want).max() < tol, (dt, split, np.abs(got - want).max())
            assert m.is_valid()
            if split is None:
                ref = dense.run(n, [(g.reshape(2 ** len(idx), -1), list(idx)) for g, idx in ops])
                assert np.abs(got - ref).max() < tol
            else:
                assert max(m.get_bond_dimensions()) <= 6
                assert list(m.get_bond_dimensions()) == list(o.get_bond_dimensions())
            # canonical form around the centre
            ts = m.get_tensors()
            c = m.get_center_position()
            for site in range(n):
                a = ts[site].to(torch.complex128)
                if site < c:
                    gram = torch.einsum("lsr,lsq->rq", a.conj(), a)
                elif site > c:
                    gram = torch.einsum("lsr,ksr->lk", a.conj(), a)
                else:
                    continue
                eye = torch.eye(gram.shape[0], dtype=gram.dtype, device=gram.device)
                assert float((gram - eye).abs().max()) < (2e-4 if dt == "complex64" else 1e-10), site
    finally:
        tc.set_dtype("complex64")


def test_mpo_conversion_and_explicit_mpo_sweeps_like_the_reference_tests():
    """reference tests/test_mpscircuit.py:275-291 (MPO_to_gate(gate_to_MPO(g)) = g, identity tensors on skipped sites)
    and :343-372 (apply_MPO in both directions: canonical afterwards, state = the circuit's)."""
    tc.set_backend("hip")
    tc.set_dtype("complex128")
    try:
        o3 = _haar(3, 21)
        mpo3, left = tc.MPSCircuit.gate_to_MPO(tc.gates.Gate(o3), 2, 3, 4)
        assert left == 2 and len(mpo3) == 3
        np.testing.assert_allclose(tc.MPSCircuit.MPO_to_gate(mpo3).tensor.cpu().numpy(), o3, atol=1e-12)
        mpo4, left = tc.MPSCircuit.gate_to_MPO(tc.gates.Gate(o3), 1, 3, 4)
        assert left == 1 and len(mpo4) == 4
        want = np.einsum("ijkabc,pq->ipjkaqbc", o3, np.eye(2))
        np.testing.assert_allclose(tc.MPSCircuit.MPO_to_gate(mpo4).tensor.cpu().numpy(), want, atol=1e-12)
        o5 = _haar(5, 22)
        mpo5, _ = tc.MPSCircuit.gate_to_MPO(tc.gates.Gate(o5), 0, 1, 2, 3, 4)
        np.testing.assert_allclose(tc.MPSCircuit.MPO_to_gate(mpo5).tensor.cpu().numpy(), o5, atol=1e-12)
        with pytest.raises(ValueError):
            tc.MPSCircuit.gate_to_MPO(tc.gates.Gate(o3), 3, 2, 4)
        # explicit MPO, both sweep directions
        u1 = _haar(1, 23)
        u_mpo = [torch.from_numpy(u1[None, :, :, None]).cuda()] * 4
        m = tc.MPSCircuit(4)
        m.position(0)
        m.apply_MPO(u_mpo, 0, center_left=True)
        assert m.get_center_position() == 0
        m.position(3)
        m.apply_MPO(u_mpo, 0, center_left=False)
        assert m.get_center_position() == 3
        c = tc.Circuit(4)
        for _ in range(2):
            for i in range(4):
                c.any(i, unitary=u1)
        np.testing.assert_allclose(m.wavefunction().reshape(-1).cpu().numpy(), c.wavefunction().cpu().numpy(), atol=1e-12)
        # reduce_dimension on one bond: the reference's entry point, here a two-site block re-split
        m2 = tc.MPSCircuit(6)
        for g, idx in _prepared(None, 6, 300):
            m2.apply(tc.gates.Gate(g), *idx)
        before = m2.wavefunction().reshape(-1).cpu().numpy()
        m2.position(2)
        m2.reduce_dimension(2, center_left=False)
        assert m2.get_center_position() == 3
        np.testing.assert_allclose(m2.wavefunction().reshape(-1).cpu().numpy(), before, atol=1e-12)
    finally:
        tc.set_dtype("complex64")


def test_spans_too_wide_for_one_block_are_gathered_by_swaps(monkeypatch):
    """With the block budget forced down, a 3-qubit gate on sites (0, 4, 7) takes the gather / scatter route
    (consecutive_swap = two-site blocks with relabelled legs): same state as the dense oracle."""
    from oracle import dense
    from tcmi.mpscircuit import MPSCircuit

    tc.set_backend("hip")
    tc.set_dtype("complex128")
    try:
        monkeypatch.setattr(MPSCircuit, "BLOCK_MAX_SITES", 4)
        n = 8
        ops = _prepared(None, n, 400) + [(_haar(3, 31), (7, 0, 4))] + _prepared(None, n, 500) + [(_haar(4, 32), (1, 6, 3, 5))]
        m = tc.MPSCircuit(n)
        for g, idx in ops:
            m.apply(tc.gates.Gate(g), *idx)
        ref = dense.run(n, [(g.reshape(2 ** len(idx), -1), list(idx)) for g, idx in ops])
        assert np.abs(m.wavefunction().reshape(-1).cpu().numpy() - ref).max() < 1e-10
        assert abs(float(m._fidelity) - 1.0) < 1e-12
    finally:
        tc.set_dtype("complex64")
