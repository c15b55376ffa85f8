"""complex128 GEMM (f64 MFMA) timing and accuracy vs torch (rocBLAS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import torch
from tcmi import linalg as LA
for M, N, K, ta in [(4096, 4096, 256, 0), (1024, 1024, 1024, 0), (300, 500, 77, 0), (64, 4096, 8, 0), (2048, 2048, 2048, 0)]:
    a = torch.randn(M, K, dtype=torch.complex128, device="cuda"); b = torch.randn(K, N, dtype=torch.complex128, device="cuda")
    c = LA.matmul(a, b); ref = a @ b
    err = float((c - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): LA.matmul(a, b)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): a @ b
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 5
    print(f"zgemm M={M} N={N} K={K}: {t*1e3:.3f} ms {8*M*N*K/t/1e12:.1f} TFLOP/s (torch {tr*1e3:.3f} ms {8*M*N*K/tr/1e12:.1f} TF) relerr {err:.1e}")
