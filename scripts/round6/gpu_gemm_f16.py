"""tcmi_cgemm_split_f16 (two f16 pieces, three piece products) next to tcmi_cgemm_split (three bf16 pieces, six) and tcmi_cgemm
(exact-f32 MFMA): error against a float64 product and time, on unit-scale, graded and state-like operands."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tensorcircuit-ng_amd"))
import numpy as np
import torch
from tcmi import _lib

L = _lib.lib()
dev = "cuda"


def run(M, N, K, B, kind="unit", reps=10, epi=False):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(B, K, M, 2, device=dev, generator=g)
    b = torch.randn(B, K, N, 2, device=dev, generator=g)
    if kind == "graded":     # entries spread over 12 decades
        a = a * torch.pow(10.0, torch.rand(B, K, M, 1, device=dev, generator=g) * 12 - 6)
        b = b * torch.pow(10.0, torch.rand(B, K, N, 1, device=dev, generator=g) * 12 - 6)
    if kind == "states":     # rows of unit norm, as the half-circuit states are
        a = a / torch.linalg.vector_norm(a, dim=(2, 3), keepdim=True)
        b = b / torch.linalg.vector_norm(b, dim=(2, 3), keepdim=True)
    A = torch.view_as_complex(a.contiguous())
    Bm = torch.view_as_complex(b.contiguous())
    amax = float(torch.view_as_real(A).abs().max()) * 2.0      # |re + im| <= 2 max
    bmax = float(torch.view_as_real(Bm).abs().max()) * 2.0
    sa = 2.0 ** int(np.floor(np.log2(60000.0 / amax)))
    sb = 2.0 ** int(np.floor(np.log2(60000.0 / bmax)))
    if kind == "states":     # the bound the cut contraction has: |amplitude| <= 1
        sa = sb = 2.0**14
    c = {k: torch.empty(B, M, N, dtype=torch.complex64, device=dev) for k in ("f32", "bf16x3", "f16x2")}
    st = torch.cuda.current_stream().cuda_stream
    X = None
    if epi:
        X = torch.eye(4, dtype=torch.complex64, device=dev).reshape(1, 16).repeat(B, 1).contiguous()
    xp = X.data_ptr() if epi else None

    def f32():
        _lib.check(L.tcmi_cgemm(A.data_ptr(), Bm.data_ptr(), c["f32"].data_ptr(), M, N, K, B, K * M, K * N, M * N, 1, 0, st), "cgemm")

    def bf():
        if epi:
            _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), c["bf16x3"].data_ptr(), M, N, K, B, K * M, K * N, M * N, xp, st), "e")
        else:
            _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c["bf16x3"].data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "s")

    def f16():
        _lib.check(L.tcmi_cgemm_split_f16(A.data_ptr(), Bm.data_ptr(), c["f16x2"].data_ptr(), M, N, K, B, K * M, K * N, M * N, xp,
                                          sa, sb, st), "f16")

    out = {}
    for name, fn in (("f32", f32), ("bf16x3", bf), ("f16x2", f16)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    print(f"M={M} N={N} K={K} B={B} {kind} epi={int(epi)} scales 2^{int(np.log2(sa))} 2^{int(np.log2(sb))}: " +
          "  ".join(f"{k} {v:.3f} ms" for k, v in out.items()), flush=True)
    if epi:      # identity X: the product with its columns un-rotated; compare the two split kernels with each other only
        d = (c["bf16x3"] - c["f16x2"]).abs().max()
        print(f"   max|bf16x3 - f16x2| {float(d):.3e}  (|c| max {float(c['bf16x3'].abs().max()):.3e})")
        return
    errs = {}
    for bi in (0, B - 1):
        rows = torch.arange(0, M, max(1, M // 64), device=dev)
        ref = A[bi].to(torch.complex128).T[rows] @ Bm[bi].to(torch.complex128)
        mag = (A[bi].abs().to(torch.float64).T[rows] @ Bm[bi].abs().to(torch.float64))
        for name in c:
            d = (c[name][bi][rows].to(torch.complex128) - ref).abs()
            errs.setdefault(name, []).append((float(d.max()), float((d / mag).max()), float((d / mag).mean())))
    for name in c:
        print(f"   {name:7s} max|err| {max(e[0] for e in errs[name]):.3e}  max err/sum|a||b| {max(e[1] for e in errs[name]):.3e}  "
              f"mean {np.mean([e[2] for e in errs[name]]):.3e}")


if __name__ == "__main__":
    run(256, 256, 64, 2)
    run(4096, 4096, 128, 8)
    run(4096, 4096, 128, 8, "states")
    run(4096, 4096, 128, 8, "graded")
    run(4096, 4096, 256, 8)
    run(4096, 4096, 128, 32, "states")
    run(4096, 4096, 128, 32, "states", epi=True)
    run(4096, 4096, 128, 8, "states", epi=True)
