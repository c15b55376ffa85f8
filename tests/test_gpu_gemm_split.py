"""tcmi_cgemm_split -- the complex64 join GEMM on the bf16 matrix pipe with three-piece operands -- and
tcmi_cgemm_split_f16 -- the same on the f16 pipe with two-piece operands of bounded magnitude -- against a float64 product,
next to tcmi_cgemm (exact-f32 MFMA): the split kernels must be as accurate as the f32 kernel (reference circuit.py:701-721
contracts complex64 operands with backend.tensordot, i.e. an f32 GEMM; north_star tolerance 1e-5 on amplitudes)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(M, N, K, B, graded=False, seed=0):
    import torch
    from tcmi import _lib

    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(B, K, M, 2, device="cuda", generator=g)
    b = torch.randn(B, K, N, 2, device="cuda", generator=g)
    if graded:      # magnitudes spread over twelve decades: bf16 keeps the exponent range of f32
        a = a * torch.pow(10.0, torch.rand(B, K, M, 1, device="cuda", generator=g) * 12 - 6)
        b = b * torch.pow(10.0, torch.rand(B, K, N, 1, device="cuda", generator=g) * 12 - 6)
    A = torch.view_as_complex(a.contiguous())
    Bm = torch.view_as_complex(b.contiguous())
    c32 = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    csp = torch.full((B, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.tcmi_cgemm(A.data_ptr(), Bm.data_ptr(), c32.data_ptr(), M, N, K, B, K * M, K * N, M * N, 1, 0, st), "cgemm")
    _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), csp.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "split")
    ref = torch.einsum("bkm,bkn->bmn", A.to(torch.complex128), Bm.to(torch.complex128))
    mag = torch.einsum("bkm,bkn->bmn", A.abs().to(torch.float64), Bm.abs().to(torch.float64))   # sum |a||b|: the error scale
    e32 = ((c32.to(torch.complex128) - ref).abs() / mag)
    esp = ((csp.to(torch.complex128) - ref).abs() / mag)
    return float(e32.max()), float(e32.mean()), float(esp.max()), float(esp.mean())


@pytest.mark.parametrize("shape", [(128, 128, 32, 1), (256, 384, 64, 3), (512, 256, 256, 2), (1024, 1024, 96, 1)])
def test_split_gemm_is_as_accurate_as_the_f32_mfma_gemm(shape):
    """Every tile position, ragged tile counts against the persistent grid, batch offsets; error relative to
    sum |a||b| (the scale of an f32 GEMM's rounding error)."""
    M, N, K, B = shape
    m32, a32, msp, asp = _run(M, N, K, B, seed=M + K)
    assert np.isfinite(msp)
    assert msp < 1e-6 and asp < 1e-7, (msp, asp)
    assert msp < 2.0 * m32 + 1e-8 and asp < 1.5 * a32 + 1e-9, ((m32, a32), (msp, asp))


def test_split_gemm_keeps_the_exponent_range_of_f32():
    m32, a32, msp, asp = _run(256, 256, 128, 2, graded=True, seed=3)
    assert np.isfinite(msp)
    assert msp < 2.0 * m32 + 1e-8 and asp < 1.5 * a32 + 1e-9, ((m32, a32), (msp, asp))


def test_split_gemm_refuses_shapes_it_does_not_take():
    import torch
    from tcmi import _lib

    L = _lib.lib()
    x = torch.zeros(1 << 16, dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for M, N, K in ((64, 128, 32), (128, 128, 16), (128, 100, 32)):
        assert L.tcmi_cgemm_split(x.data_ptr(), x.data_ptr(), x.data_ptr(), M, N, K, 1, K * M, K * N, M * N, st) != 0


def test_cut_join_runs_on_the_split_kernel_and_matches_the_f32_join():
    """The product path: a cut contraction's state with the join on either kernel, and against the dense oracle."""
    import tcmi as tc
    from tcmi import executor as X
    from oracle import dense, workloads as W

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("cut")
    n, d = 16, 5
    params = np.random.default_rng(5).uniform(0, 2 * np.pi, [2 * d, n])
    ref = dense.run(n, W.hea_b_ops(n, d, params))

    def state():
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype="float32"), zz=tc.gates._zz_matrix)
        cc = c._compiled()
        return cc, tc.backend.numpy(c.wavefunction())

    old = X.JOIN_GEMM
    try:
        X.JOIN_GEMM = "split"
        cc, s_split = state()
        if not isinstance(cc, X.CutCircuit):
            pytest.skip("the planner did not cut this circuit")
        X.JOIN_GEMM = "f32"
        _, s_f32 = state()
    finally:
        X.JOIN_GEMM = old
        tc.set_contractor("greedy")
    assert np.abs(s_split - ref).max() < 1e-5 and np.abs(s_f32 - ref).max() < 1e-5
    assert np.abs(s_split - ref).max() < 2 * np.abs(s_f32 - ref).max() + 1e-8


@pytest.mark.parametrize("shape", [(128, 128, 32, 1), (256, 384, 64, 3), (1024, 512, 128, 2)])
def test_split_gemm_with_a_gate_applied_to_the_product(shape):
    """tcmi_cgemm_split_epi: the product, a 4 x 4 on (lowest row bit, lowest column bit) per batch member, columns stored
    un-rotated -- against the same thing in complex128; error relative to sum |a||b| no worse than the plain split GEMM's
    bound (X is unitary here, as in the cut contraction)."""
    import torch
    from tcmi import _lib

    M, N, K, B = shape
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda", generator=g))
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda", generator=g))
    q, _ = torch.linalg.qr(torch.view_as_complex(torch.randn(B, 4, 4, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(K))))
    X64 = q.to("cuda")
    X32 = X64.to(torch.complex64).reshape(B, 16).contiguous()
    out = torch.full((B, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                      X32.data_ptr(), st), "split_epi")
    P = torch.einsum("bkm,bkn->bmn", A.to(torch.complex128), Bm.to(torch.complex128)).reshape(B, M // 2, 2, N // 2, 2)
    Y = torch.einsum("bopuv,bmucv->bmocp", X32.to(torch.complex128).reshape(B, 2, 2, 2, 2), P)   # X[b, u', v', u, v]; Y[b, m, u', c, v']
    want = Y.permute(0, 1, 2, 4, 3).reshape(B, M, N)     # column (c, v') of the product sits at v' N / 2 + c
    mag = torch.einsum("bkm,bkn->bmn", A.abs().to(torch.float64), Bm.abs().to(torch.float64)).reshape(B, M // 2, 2, N // 2, 2)
    mag = mag.sum(dim=(2, 4), keepdim=True).expand(B, M // 2, 2, N // 2, 2).permute(0, 1, 2, 4, 3).reshape(B, M, N)
    err = (out.to(torch.complex128) - want).abs() / mag
    assert torch.isfinite(out.real).all() and torch.isfinite(out.imag).all()
    assert float(err.max()) < 1e-6 and float(err.mean()) < 1e-7, (float(err.max()), float(err.mean()))
    # X = identity: the plain split product with the columns un-rotated, bit for bit
    eye = torch.eye(4, dtype=torch.complex64, device="cuda").reshape(1, 16).repeat(B, 1).contiguous()
    plain = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), plain.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "split")
    _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                      eye.data_ptr(), st), "split_epi")
    assert torch.equal(out.reshape(B, M, 2, N // 2), plain.reshape(B, M, N // 2, 2).permute(0, 1, 3, 2))


def test_cut_with_the_last_crossing_gate_applied_by_the_join():
    """The product path with a deferred gate (tcmi/cut.py): half the bond, the 4 x 4 of (ZZ, rx, rx) from
    tcmi_cut_epilogue, the join on tcmi_cgemm_split_epi -- against ``oracle.dense`` and against the plain cut."""
    import torch
    import tcmi as tc
    from tcmi import executor as X
    from oracle import dense, workloads as W

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("cut")
    try:
        n, d = 16, 6
        rng = np.random.default_rng(11)
        params = rng.uniform(0, 2 * np.pi, [3, 2 * d, n])
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(params[0], dtype="float32"), zz=tc.gates._zz_matrix)
        cc = c._compiled()
        assert isinstance(cc, X.CutCircuit) and cc.spec.epilogue is not None and cc.K == 32 and cc.spec.plain.bond_dim == 64
        got = tc.backend.numpy(c.wavefunction())
        ref = dense.run(n, W.hea_b_ops(n, d, params[0]))
        assert np.abs(got - ref).max() < 1e-5
        # a batch through the compiled object: every member has its own X; against the plain cut of the same circuits
        pv = torch.as_tensor(np.stack([np.asarray([float(v) for v in _params_of(tc, W, n, d, params[i])]) for i in range(3)]),
                             dtype=torch.float32, device="cuda")
        sd = cc.state(pv)
        sp = cc._plain_cut().state(pv)
        assert cc._plain_cut().K == 64
        assert float((sd - sp).abs().max()) < 2e-6
        for i in range(3):
            assert np.abs(sd[i].cpu().numpy() - dense.run(n, W.hea_b_ops(n, d, params[i]))).max() < 1e-5
    finally:
        tc.set_contractor("greedy")


def test_deferred_gates_of_a_mixed_circuit(defer="1"):
    """Not the HEA-B pattern: ry layers (real matrices), constant one-qubit unitaries (any 2 x 2) and a cz beside the cut in the tail, seven
    crossing rzz gates -- the last of them applied by the join, against ``oracle.dense``."""
    import os
    import tcmi as tc
    from tcmi import executor as X
    from oracle import dense, gates as G

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("cut")
    old = os.environ.get("TCMI_CUT_DEFER")
    os.environ["TCMI_CUT_DEFER"] = defer
    try:
        n, d = 16, 7
        rng = np.random.default_rng(21)
        th = rng.uniform(0, 2 * np.pi, [d, 2, n])
        us = [G.random_two_qubit_gate(s)[:2, :2] for s in range(4)]
        us = [np.linalg.qr(u)[0] for u in us]
        c = tc.Circuit(n)
        ops = []
        for i in range(n):
            c.h(i); ops.append((G.H, [i]))
        for l in range(d):
            for i in range(n - 1):
                c.rzz(i, i + 1, theta=float(th[l, 0, i])); ops.append((G.rzz(th[l, 0, i]), [i, i + 1]))
            if l == d - 1:
                c.cz(8, 9); ops.append((G.CZ, [8, 9]))
            for i in range(n):
                if l == d - 1 and 6 <= i <= 9:
                    c.any(i, unitary=us[i - 6]); ops.append((us[i - 6], [i]))
                elif l % 2:
                    c.ry(i, theta=float(th[l, 1, i])); ops.append((G.ry(th[l, 1, i]), [i]))
                else:
                    c.rx(i, theta=float(th[l, 1, i])); ops.append((G.rx(th[l, 1, i]), [i]))
        cc = c._compiled()
        assert isinstance(cc, X.CutCircuit) and cc.spec.epilogue is not None
        assert cc.K == 64
        got = tc.backend.numpy(c.wavefunction())
        assert np.abs(got - dense.run(n, ops)).max() < 1e-5
    finally:
        tc.set_contractor("greedy")
        if old is None:
            os.environ.pop("TCMI_CUT_DEFER", None)
        else:
            os.environ["TCMI_CUT_DEFER"] = old


def _params_of(tc, W, n, d, params):
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    return c._params


def test_split_gemm_on_sums_that_cancel_down_to_the_dropped_terms():
    """Adversarial operands: every k-pair is (x, -1) . (y, z) with x = 1 + 2^-9 + 2^-18 (bf16 pieces 1, 2^-9, 2^-18), y of
    the same form and z = the sum of the SIX piece products the kernel keeps, which is exact in f32.  The exact value of
    x y - z is then precisely what the kernel drops (x1 y2 + x2 y1 + x2 y2 = 2^-26 + 2^-36 per pair): the split kernel
    returns 0 where float64 returns K/2 * 2^-26 -- a 100 % RELATIVE error on a fully cancelled sum.  The bound the kernel
    meets, and this test asserts, is the absolute one of an f32 GEMM that rounds every product once:
    |c - exact| <= 2^-23 sum_k |a_k||b_k| (here the error is 2^-27 of that scale).  An exact-f32 FMA chain can do better on
    this input only by luck of its accumulation order (fl(x y) alone already loses the 2^-26); nothing in the path --
    reference tensorcircuit contracts complex64 with an f32 GEMM (circuit.py:701-721, backend.tensordot) -- relies on more."""
    import torch
    from tcmi import _lib

    L = _lib.lib()
    M = N = 128
    K, B = 64, 1
    x = 1.0 + 2.0 ** -9 + 2.0 ** -18
    kept = 1.0 + 2.0 ** -8 + 3.0 * 2.0 ** -18              # x0 y0 + x0 y1 + x1 y0 + x0 y2 + x2 y0 + x1 y1, exact in f32
    assert float(np.float32(kept)) == kept and float(np.float32(x)) == x
    a = torch.zeros(B, K, M, 2, dtype=torch.float32, device="cuda")
    b = torch.zeros(B, K, N, 2, dtype=torch.float32, device="cuda")
    a[:, 0::2, :, 0] = x
    a[:, 1::2, :, 0] = -1.0
    b[:, 0::2, :, 0] = x
    b[:, 1::2, :, 0] = kept
    # exact phases i^k on the rows of a and (-i)^k on b: the complex arithmetic (Gauss's three products) is exercised and
    # the sum stays the same
    ph = torch.tensor([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0]], device="cuda")
    for k in range(K):
        p, q = ph[(k // 2) % 4], ph[(-(k // 2)) % 4]
        ar, br = a[:, k, :, 0].clone(), b[:, k, :, 0].clone()
        a[:, k, :, 0], a[:, k, :, 1] = ar * p[0], ar * p[1]
        b[:, k, :, 0], b[:, k, :, 1] = br * q[0], br * q[1]
    A, Bm = torch.view_as_complex(a), torch.view_as_complex(b)
    c = torch.full((B, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "split")
    ref = torch.einsum("bkm,bkn->bmn", A.to(torch.complex128), Bm.to(torch.complex128))
    want = (K // 2) * (2.0 ** -26 + 2.0 ** -36)
    assert abs(complex(ref[0, 0, 0]) - want) < 1e-15             # float64 sees exactly the dropped terms
    scale = float(torch.einsum("bkm,bkn->bmn", A.abs().double(), Bm.abs().double())[0, 0, 0])
    err = float((c.to(torch.complex128) - ref).abs().max())
    assert np.isfinite(err)
    assert err <= 2.0 ** -23 * scale, (err, scale)               # the f32-GEMM bound (one rounding per product)
    assert err <= 1.01 * want                                    # and nothing beyond the dropped terms is lost


def _pow2_scale(t):
    """The largest power of two that keeps |re|, |im| and |re + im| of ``t`` (complex) under f16's 65504."""
    import torch

    m = 2.0 * float(torch.view_as_real(t).abs().max())
    return float(2.0 ** int(np.floor(np.log2(60000.0 / m))))


def _run_f16(M, N, K, B, kind="unit", seed=0, x=None):
    import torch
    from tcmi import _lib

    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(B, K, M, 2, device="cuda", generator=g)
    b = torch.randn(B, K, N, 2, device="cuda", generator=g)
    if kind == "graded":      # magnitudes spread over six decades below the operand's largest entry
        a = a * torch.pow(10.0, -6.0 * torch.rand(B, K, M, 1, device="cuda", generator=g))
        b = b * torch.pow(10.0, -6.0 * torch.rand(B, K, N, 1, device="cuda", generator=g))
    if kind == "states":      # unit-norm rows, as the half-circuit states of a cut are; the scale the executor would take
        a = a / torch.linalg.vector_norm(a, dim=(2, 3), keepdim=True)
        b = b / torch.linalg.vector_norm(b, dim=(2, 3), keepdim=True)
    A = torch.view_as_complex(a.contiguous())
    Bm = torch.view_as_complex(b.contiguous())
    sa, sb = (2.0**15, 2.0**15) if kind == "states" else (_pow2_scale(A), _pow2_scale(Bm))
    c32 = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    c16 = torch.full((B, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.tcmi_cgemm(A.data_ptr(), Bm.data_ptr(), c32.data_ptr(), M, N, K, B, K * M, K * N, M * N, 1, 0, st), "cgemm")
    _lib.check(L.tcmi_cgemm_split_f16(A.data_ptr(), Bm.data_ptr(), c16.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                      None if x is None else x.data_ptr(), sa, sb, st), "split_f16")
    return A, Bm, c32, c16


@pytest.mark.parametrize("shape", [(128, 128, 32, 1), (256, 384, 64, 3), (512, 256, 256, 2), (1024, 1024, 96, 1)])
@pytest.mark.parametrize("kind", ["unit", "states", "graded"])
def test_f16_split_gemm_is_as_accurate_as_the_f32_mfma_gemm(shape, kind):
    """Two f16 pieces, three piece products: every tile position, ragged tile counts against the persistent grid, batch
    offsets, operands of unit scale / unit-norm rows at the executor's scale 2^15 / graded over six decades (the low piece
    of the small entries goes through f16's subnormals).  Error relative to sum |a||b|: no worse than the f32 kernel's."""
    import torch

    M, N, K, B = shape
    A, Bm, c32, c16 = _run_f16(M, N, K, B, kind, seed=M + K)
    ref = torch.einsum("bkm,bkn->bmn", A.to(torch.complex128), Bm.to(torch.complex128))
    mag = torch.einsum("bkm,bkn->bmn", A.abs().to(torch.float64), Bm.abs().to(torch.float64))
    e32 = (c32.to(torch.complex128) - ref).abs() / mag
    e16 = (c16.to(torch.complex128) - ref).abs() / mag
    m32, a32, m16, a16 = float(e32.max()), float(e32.mean()), float(e16.max()), float(e16.mean())
    assert np.isfinite(m16)
    assert m16 < 3e-6 and a16 < 3e-7, (m16, a16)
    assert m16 < 2.0 * m32 + 1e-8 and a16 < 1.5 * a32 + 1e-9, ((m32, a32), (m16, a16))


def test_f16_split_gemm_with_a_gate_applied_to_the_product():
    """The 4 x 4 epilogue of tcmi_cgemm_split_f16 against the three-piece kernel's (tcmi_cgemm_split_epi, itself against
    complex128 above): unitary X per batch member, the columns un-rotated."""
    import torch
    from tcmi import _lib

    L = _lib.lib()
    for M, N, K, B in ((128, 128, 32, 1), (256, 384, 64, 3), (1024, 512, 128, 2)):
        q, _ = torch.linalg.qr(torch.view_as_complex(torch.randn(B, 4, 4, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(K))))
        X32 = q.to("cuda").to(torch.complex64).reshape(B, 16).contiguous()
        A, Bm, _, c16 = _run_f16(M, N, K, B, "states", seed=M + N + K, x=X32)
        want = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), want.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                          X32.data_ptr(), st), "split_epi")
        mag = torch.einsum("bkm,bkn->bmn", A.abs().to(torch.float64), Bm.abs().to(torch.float64)).reshape(B, M // 2, 2, N // 2, 2)
        mag = mag.sum(dim=(2, 4), keepdim=True).expand(B, M // 2, 2, N // 2, 2).permute(0, 1, 2, 4, 3).reshape(B, M, N)
        err = (c16.to(torch.complex128) - want.to(torch.complex128)).abs() / mag
        assert torch.isfinite(c16.real).all() and torch.isfinite(c16.imag).all()
        assert float(err.max()) < 1e-6 and float(err.mean()) < 1e-7, (float(err.max()), float(err.mean()))


def test_f16_split_gemm_is_loud_about_what_it_cannot_take():
    """Scales that are not powers of two and shapes the kernel does not take are refused; an operand beyond f16's range
    at the given scale gives a non-finite product (the caller's bound was wrong), never a quietly wrong one."""
    import torch
    from tcmi import _lib

    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1 << 16, dtype=torch.complex64, device="cuda")
    args = lambda M, N, K: (x.data_ptr(), x.data_ptr(), x.data_ptr(), M, N, K, 1, K * M, K * N, M * N, None)
    assert L.tcmi_cgemm_split_f16(*args(128, 128, 32), 3.0, 1.0, st) != 0
    assert L.tcmi_cgemm_split_f16(*args(128, 128, 32), 1.0, -2.0, st) != 0
    assert L.tcmi_cgemm_split_f16(*args(64, 128, 32), 1.0, 1.0, st) != 0
    assert L.tcmi_cgemm_split_f16(*args(128, 128, 16), 1.0, 1.0, st) != 0
    M = N = 128; K = 32
    A = torch.full((1, K, M), 0.01 + 0.0j, dtype=torch.complex64, device="cuda")
    Bm = torch.full((1, K, N), 0.01 + 0.0j, dtype=torch.complex64, device="cuda")
    A[0, 3, 5] = 8.0          # 8 x 2^15 is beyond 65504
    c = torch.zeros(1, M, N, dtype=torch.complex64, device="cuda")
    _lib.check(L.tcmi_cgemm_split_f16(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, 1, K * M, K * N, M * N, None,
                                      2.0**15, 2.0**15, st), "split_f16")
    assert not bool(torch.isfinite(torch.view_as_real(c[0, 5])).all())
    ok = torch.ones(M, dtype=torch.bool, device="cuda"); ok[5] = False
    np.testing.assert_allclose(c[0][ok].cpu().numpy(), K * 1e-4, rtol=1e-5)


def test_cut_join_runs_on_the_f16_kernel_when_the_cut_bounds_its_halves():
    """The product path: HEA-B's halves are unitary circuits on |0..0> with projector-like bond factors (cut.half_bounds
    = 1): the join runs on tcmi_cgemm_split_f16 at scale 2^15; against the dense oracle and against the three-piece join."""
    import tcmi as tc
    from tcmi import executor as X
    from oracle import dense, workloads as W

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("cut")
    n, d = 16, 6
    params = np.random.default_rng(9).uniform(0, 2 * np.pi, [2 * d, n])
    ref = dense.run(n, W.hea_b_ops(n, d, params))

    def state():
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype="float32"), zz=tc.gates._zz_matrix)
        return c._compiled(), tc.backend.numpy(c.wavefunction())

    old = X.JOIN_GEMM
    try:
        X.JOIN_GEMM = "split"
        cc, s_f16 = state()
        assert isinstance(cc, X.CutCircuit) and cc._f16 == (2.0**15, 2.0**15)
        X.JOIN_GEMM = "bf16"
        _, s_bf = state()
    finally:
        X.JOIN_GEMM = old
        tc.set_contractor("greedy")
    assert np.abs(s_f16 - ref).max() < 1e-5 and np.abs(s_bf - ref).max() < 1e-5
    assert np.abs(s_f16 - ref).max() < 2 * np.abs(s_bf - ref).max() + 1e-8
    assert np.abs(s_f16 - s_bf).max() < 1e-6


def test_cut_join_takes_its_f16_scales_from_non_unitary_halves():
    """Two non-unitary one-qubit gates (norms 2 and ~1.28) in the halves of a cut: ``cut.half_bounds`` grows by them, the
    operand scales shrink to the next powers of two, and the state (norm ~2.6: entries beyond what a unit-norm state has)
    still matches the dense oracle; against the three-piece join."""
    import tcmi as tc
    from tcmi import executor as X
    from oracle import dense, workloads as W

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("cut")
    n, d = 16, 5
    params = np.random.default_rng(21).uniform(0, 2 * np.pi, [2 * d, n])
    g1 = np.diag([2.0, 0.5]).astype(np.complex128)
    g2 = np.array([[1.0, 0.5], [0.0, 1.0]], dtype=np.complex128)
    ref = dense.run(n, W.hea_b_ops(n, d, params) + [(g1, [3]), (g2, [12])])

    def state():
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype="float32"), zz=tc.gates._zz_matrix)
        c.any(3, unitary=g1)
        c.any(12, unitary=g2)
        return c._compiled(), tc.backend.numpy(c.wavefunction())

    old = X.JOIN_GEMM
    try:
        X.JOIN_GEMM = "split"
        cc, s_f16 = state()
        if not isinstance(cc, X.CutCircuit):
            pytest.skip("the planner did not cut this circuit")
        assert cc._f16 == (2.0**14, 2.0**15) or cc._f16 == (2.0**15, 2.0**14) or cc._f16 == (2.0**14, 2.0**14), cc._f16
        X.JOIN_GEMM = "bf16"
        _, s_bf = state()
    finally:
        X.JOIN_GEMM = old
        tc.set_contractor("greedy")
    scale = np.abs(ref).max()
    assert np.isfinite(s_f16).all()
    assert np.abs(s_f16 - ref).max() < 1e-5 * max(1.0, scale / 1e-2) and np.abs(s_bf - ref).max() < 1e-5 * max(1.0, scale / 1e-2)
    assert np.abs(s_f16 - ref).max() < 2 * np.abs(s_bf - ref).max() + 1e-8
