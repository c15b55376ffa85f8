import sys, os
sys.path.insert(0, "tensorcircuit-ng_amd")
import torch
from tcmi import _lib
L = _lib.lib()
M = N = 4096
st = torch.cuda.current_stream().cuda_stream
for K in (128, 64, 32):
  for B in (8, 32):
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
    X = torch.view_as_complex(torch.randn(B, 16, 2, device="cuda"))
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    f = lambda: _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, X.data_ptr(), st), "x")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"K={K} batch {B}: {t:.3f} ms = {t*1e3/(4*B):.2f} us per tile, {B*2**24/t/1e6:.1f} G amp/s join-only, out write {B*M*N*8/t/1e9:.2f} TB/s")
