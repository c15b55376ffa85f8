"""tcmi_cgemm_split (bf16 matrix pipe, three-piece operands) against tcmi_cgemm (exact-f32 MFMA) and a float64 product:
error of both paths and their times on the join shape of BASELINE config 2 (M = N = 4096, K = 256, batch 8)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd"))
import numpy as np
import torch
from tcmi import _lib

L = _lib.lib()
dev = "cuda"


def run(M, N, K, B, scale="unit", reps=10):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(B, K, M, 2, device=dev, generator=g)
    b = torch.randn(B, K, N, 2, device=dev, generator=g)
    if scale == "graded":     # entries spread over 12 decades
        a = a * torch.pow(10.0, torch.rand(B, K, M, 1, device=dev, generator=g) * 12 - 6)
        b = b * torch.pow(10.0, torch.rand(B, K, N, 1, device=dev, generator=g) * 12 - 6)
    A = torch.view_as_complex(a.contiguous())
    Bm = torch.view_as_complex(b.contiguous())
    c1 = torch.empty(B, M, N, dtype=torch.complex64, device=dev)
    c2 = torch.empty(B, M, N, dtype=torch.complex64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def f32():
        _lib.check(L.tcmi_cgemm(A.data_ptr(), Bm.data_ptr(), c1.data_ptr(), M, N, K, B, K * M, K * N, M * N, 1, 0, st), "cgemm")

    def split():
        _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c2.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "cgemm_split")

    out = {}
    for name, fn in (("f32", f32), ("split", split)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    # float64 reference on a few rows of batch member 0 and B - 1
    errs = {}
    for bi in (0, B - 1):
        rows = torch.arange(0, M, max(1, M // 64), device=dev)
        ref = A[bi].to(torch.complex128).T[rows] @ Bm[bi].to(torch.complex128)
        mag = (A[bi].abs().to(torch.float64).T[rows] @ Bm[bi].abs().to(torch.float64))   # sum |a||b|: the error scale
        for name, c in (("f32", c1), ("split", c2)):
            d = (c[bi][rows].to(torch.complex128) - ref).abs()
            errs.setdefault(name, []).append((float(d.max()), float((d / mag).max()), float((d / mag).mean())))
    print(f"M={M} N={N} K={K} B={B} {scale}: f32 {out['f32']:.3f} ms  split {out['split']:.3f} ms  "
          f"ratio {out['f32'] / out['split']:.2f}")
    for name in ("f32", "split"):
        print(f"   {name:6s} max|err| {max(e[0] for e in errs[name]):.3e}  max err/sum|a||b| {max(e[1] for e in errs[name]):.3e}  "
              f"mean {np.mean([e[2] for e in errs[name]]):.3e}")
    same = float((c1 - c2).abs().max())
    print(f"   max|f32 - split| {same:.3e}")


if __name__ == "__main__":
    run(256, 256, 64, 2)
    run(4096, 4096, 256, 8)
    run(4096, 4096, 256, 8, "graded")
    run(2048, 8192, 512, 2)
