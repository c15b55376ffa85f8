"""The arithmetic of the join GEMM's split schemes, emulated in numpy (``oracle/split_gemm.py``): what ``include/tcmi.h``
promises about ``tcmi_cgemm_split`` (three bf16 pieces) and ``tcmi_cgemm_split_f16`` (two f16 pieces of bounded operands)
holds for the SCHEMES; the GPU tests (tests/test_gpu_gemm_split.py) measure the kernels against float64."""

import numpy as np

from oracle import split_gemm as SG


def test_three_bf16_pieces_are_exact_and_two_f16_pieces_leave_2_to_the_minus_22():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 0, 20000)).astype(np.float32)
    p = SG.pieces_bf16(x)
    assert np.array_equal(p[0] + p[1] + p[2], x)
    for q in p:      # every piece is a bf16 value
        assert not np.any(q.view(np.uint32) & np.uint32(0xFFFF))
    x = x / np.abs(x).max()
    s = 2.0**15
    h, l = SG.pieces_f16(x, s)
    t = x.astype(np.float64) * s
    big = np.abs(t) * 2.0**-11 >= 2.0**-14          # the low piece is a normal f16
    assert np.all(np.abs(t - h - l)[big] <= 2.0**-22 * np.abs(t)[big])
    assert np.all(np.abs(t - h - l)[~big] <= 2.0**-25 + 2.0**-22 * np.abs(t)[~big])      # below: half an f16 subnormal step
    assert np.isfinite(h).all() and np.abs(h).max() <= 65504


def test_real_product_of_two_f16_pieces_drops_at_most_3_times_2_to_the_minus_22():
    rng = np.random.default_rng(1)
    x = rng.uniform(0.05, 1.0, 4000).astype(np.float32) * rng.choice([-1.0, 1.0], 4000).astype(np.float32)
    y = rng.uniform(0.05, 1.0, 4000).astype(np.float32) * rng.choice([-1.0, 1.0], 4000).astype(np.float32)
    s = 2.0**15
    hx, lx = (a.astype(np.float64) for a in SG.pieces_f16(x, s))
    hy, ly = (a.astype(np.float64) for a in SG.pieces_f16(y, s))
    got = (hx * hy + hx * ly + lx * hy) / (s * s)
    want = x.astype(np.float64) * y.astype(np.float64)
    assert np.all(np.abs(got - want) <= 3.0 * 2.0**-22 * np.abs(want) * (1 + 1e-6))
    # the six kept products of three bf16 pieces: what is dropped stays under the 2 * 2^-24 + 2^-32 the header states
    px, py = SG.pieces_bf16(x), SG.pieces_bf16(y)
    got3 = sum(px[i].astype(np.float64) * py[j].astype(np.float64) for i, j in SG.KEEP_BF16X3)
    assert np.all(np.abs(got3 - want) <= (2.0 * 2.0**-24 + 2.0**-32) * np.abs(want) * (1 + 1e-6))


def test_split_complex_gemm_is_as_accurate_as_an_f32_gemm():
    """Unit-norm rows (half-circuit states) at the executor's scale 2^15, and operands graded over six decades at a scale
    taken from their largest entry: both schemes against float64, next to numpy's own complex64 product."""
    rng = np.random.default_rng(2)
    M, N, K = 96, 80, 64
    for graded in (False, True):
        a = rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))
        b = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
        if graded:
            a = a * 10.0 ** rng.uniform(-6, 0, (M, K))
            b = b * 10.0 ** rng.uniform(-6, 0, (K, N))
        a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.complex64)
        b = (b / np.linalg.norm(b, axis=0, keepdims=True)).astype(np.complex64)
        ref = a.astype(np.complex128) @ b.astype(np.complex128)
        mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        e32 = np.abs((a @ b).astype(np.complex128) - ref) / mag
        e3 = np.abs(SG.cgemm(a, b, "bf16x3") - ref) / mag
        e2 = np.abs(SG.cgemm(a, b, "f16x2", 2.0**15, 2.0**15) - ref) / mag
        if not graded:
            assert e3.max() < 1e-7 and e3.mean() < 1e-8
            assert e2.max() < 2e-7 and e2.mean() < 3e-8
        # no worse than the f32 product they stand in for (Gauss's third product, (re + im)(re' + im'), is why graded
        # operands cost both schemes and the plain f32 product alike)
        assert e3.mean() < e32.mean() and e3.max() < e32.max()
        assert e2.mean() < e32.mean() and e2.max() < e32.max()


def test_an_operand_beyond_its_bound_is_loud():
    a = np.full((4, 8), 0.01, dtype=np.complex64)
    b = np.full((8, 4), 0.01, dtype=np.complex64)
    a[1, 3] = 8.0          # 8 * 2^15 is beyond f16's 65504
    c = SG.cgemm(a, b, "f16x2", 2.0**15, 2.0**15)
    assert not np.isfinite(c[1]).all()
    assert np.isfinite(np.delete(c, 1, axis=0)).all()
