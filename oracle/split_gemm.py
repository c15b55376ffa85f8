"""TEST INFRASTRUCTURE (oracle): numpy emulation of the arithmetic of the join GEMM's two split schemes
(tensorcircuit-ng_amd/csrc/tcmi_gemm_split.hip; the join replaces the reference's complex64 ``backend.tensordot`` of the
two half-networks, circuit.py:701-721 -> cons.py:948) -- every f32 operand value as a sum of narrow pieces, a real product as
the piece products the kernel keeps, each exact in an f32 accumulator, Gauss's three real products per complex product.
Used by the CPU tests to pin the error claims of ``include/tcmi.h`` without a GPU; nothing in the product imports it.

  * ``pieces_bf16``: three bf16 pieces, round to nearest each (x = x0 + x1 + x2 exactly).
  * ``pieces_f16``: two f16 pieces of ``x * scale`` (round to nearest, the second of what the first left; numpy's float16
    has gradual underflow like the hardware's).
"""

import numpy as np


def _bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> the nearest bf16 (ties to even), as float32 (v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32)


def pieces_bf16(x: np.ndarray):
    """x (float32) -> [x0, x1, x2] float32 arrays holding bf16 values, each the nearest bf16 of what the previous ones
    left: 8 + 8 + 8 significand bits and the signs of the residues, x0 + x1 + x2 == x."""
    x = np.asarray(x, dtype=np.float32)
    p0 = _bf16_rne(x)
    r = x - p0
    p1 = _bf16_rne(r)
    p2 = _bf16_rne(r - p1)
    return [p0, p1, p2]


def pieces_f16(x: np.ndarray, scale: float):
    """x (float32), scale a power of two -> [h, l] float32 arrays holding f16 values of ``x * scale``: h the nearest f16,
    l the nearest f16 of the residue (the kernel: v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32)."""
    t = np.asarray(x, dtype=np.float32) * np.float32(scale)
    with np.errstate(over="ignore"):
        h = t.astype(np.float16).astype(np.float32)
        l = (t - h).astype(np.float16).astype(np.float32)
    return [h, l]


def real_product(pa, pb, keep) -> np.ndarray:
    """sum over the kept (i, j) of pa[i]^T-free matrix products pa[i] @ pb[j], accumulated in float64 (the kernel's f32
    accumulation adds its own rounding: the tests bound the SCHEME's error, the GPU tests measure the kernel's)."""
    acc = 0.0
    with np.errstate(invalid="ignore", over="ignore"):      # an operand beyond its bound is inf by design (the tests check it)
        for i, j in keep:
            acc = acc + pa[i].astype(np.float64) @ pb[j].astype(np.float64)
    return acc


KEEP_BF16X3 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]      # piece products of order <= 2^-16
KEEP_F16X2 = [(0, 0), (0, 1), (1, 0)]                                # h h' + h l' + l h'


def cgemm(A: np.ndarray, B: np.ndarray, scheme: str = "f16x2", scale_a: float = 1.0, scale_b: float = 1.0) -> np.ndarray:
    """A [M, K] @ B [K, N], complex64 operands, by Gauss's three real products on split operands."""
    A = np.asarray(A, dtype=np.complex64)
    B = np.asarray(B, dtype=np.complex64)
    ar, ai, br, bi = A.real.copy(), A.imag.copy(), B.real.copy(), B.imag.copy()
    if scheme == "bf16x3":
        cut_a = cut_b = pieces_bf16
        keep, unscale = KEEP_BF16X3, 1.0
        asum, bsum = ar + ai, br + bi
    else:
        cut_a = lambda x: pieces_f16(x, scale_a)
        cut_b = lambda x: pieces_f16(x, scale_b)
        keep, unscale = KEEP_F16X2, 1.0 / (scale_a * scale_b)
        asum, bsum = ar + ai, br + bi      # the kernel adds the SCALED parts: the same value, scaling by 2^k is exact
    p1 = real_product(cut_a(ar), cut_b(br), keep)
    p2 = real_product(cut_a(ai), cut_b(bi), keep)
    p3 = real_product(cut_a(asum), cut_b(bsum), keep)
    return ((p1 - p2) + 1j * (p3 - p1 - p2)) * unscale
