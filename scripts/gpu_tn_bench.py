"""Throughput of the tensordot-engine kernels: tcmi_cgemm (MFMA f32) and tcmi_permute_bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn, _lib

tc.set_backend("hip"); tc.set_dtype("complex64")
lib = _lib.lib()
stream = torch.cuda.current_stream().cuda_stream
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (M, N, K) in [(4096, 4096, 256), (4096, 4096, 4096), (16384, 16384, 64), (1024, 1024, 1024), (16, 1 << 20, 16), (4, 1 << 22, 4), (1 << 12, 1 << 12, 16)]:
    a = torch.randn(M * K, dtype=torch.complex64, device="cuda"); b = torch.randn(K * N, dtype=torch.complex64, device="cuda")
    c = torch.empty(M * N, dtype=torch.complex64, device="cuda")
    f = lambda: _lib.check(lib.tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 1, 0, 0, 0, 0, 0, stream), "cgemm")
    ms = timeit(f)
    flops = 8.0 * M * N * K
    byts = 8.0 * (M * K + K * N + M * N)
    ref = timeit(lambda: torch.matmul(a.view(M, K), b.view(K, N)))
    err = float((c.view(M, N) - torch.matmul(a.view(M, K), b.view(K, N))).abs().max() / (K ** 0.5))
    print(f"cgemm M={M} N={N} K={K}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s  {byts/ms/1e6:.0f} GB/s   (torch/rocBLAS {ref:.3f} ms {flops/ref/1e9:.1f} TF)  relerr {err:.1e}", flush=True)
for rank in (20, 24, 26):
    t = torch.randn([2] * rank, dtype=torch.complex64, device="cuda")
    for name, perm in [("reverse", tuple(reversed(range(rank)))), ("swap-halves", tuple(range(rank // 2, rank)) + tuple(range(rank // 2))), ("move-last-to-front", (rank - 1,) + tuple(range(rank - 1))), ("swap-two-high", (1, 0) + tuple(range(2, rank)))]:
        ms = timeit(lambda: tn._permute_raw(t, perm), reps=5)
        print(f"permute rank={rank} {name}: {ms:.3f} ms  {2 * 8 * 2**rank / ms / 1e6:.0f} GB/s", flush=True)
