"""tcmi_cgemm_split -- the complex64 join GEMM on the bf16 matrix pipe with three-piece operands -- against a float64
product, next to tcmi_cgemm (exact-f32 MFMA): the split kernel must be as accurate as the f32 kernel (reference
circuit.py:701-721 contracts complex64 operands with backend.tensordot, i.e. an f32 GEMM; north_star tolerance 1e-5 on
amplitudes)."""

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
