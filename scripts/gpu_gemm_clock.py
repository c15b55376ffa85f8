"""Join-GEMM rate on random vs zero-filled operands (DVFS: the chip clocks to its power budget,
MI355X_MICROARCH.md "DVFS give-back"): gpu_gemm_clock.py [M N K batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import torch
from tcmi import _lib
M, N, K, B = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (4096, 4096, 256, 8)
st = torch.cuda.current_stream().cuda_stream
for fill in ("random", "zeros", "random"):
    a = torch.randn(B, K, M, dtype=torch.complex64, device="cuda") if fill == "random" else torch.zeros(B, K, M, dtype=torch.complex64, device="cuda")
    b = torch.randn(B, K, N, dtype=torch.complex64, device="cuda") if fill == "random" else torch.zeros(B, K, N, dtype=torch.complex64, device="cuda")
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    def run():
        _lib.check(_lib.lib().tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, B, M * K, K * N, M * N, 1, _lib.TCMI_C64, st), "g")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{fill:7s} DMA={os.environ.get('TCMI_GEMM_DMA','1')} {M}x{N}x{K} b{B}: {ms*1e3:8.1f} us  executed {6.0*M*N*K*B/ms/1e9:6.1f} TF  algorithmic {8.0*M*N*K*B/ms/1e9:6.1f} TF", flush=True)
