"""Probe: time of the split GEMM kernel alone (HIP events around the MFMA kernel are not available through the C ABI, so
the two plane passes are timed separately and subtracted) in its modes (TCMI_SPLIT_MODE = 0 full, 1 no DMA in the loop,
2 no MFMA, 3 no stores) for K = 256, 512, 1024: slope = cost per 16-k step, intercept = fixed cost per workgroup."""
import sys, os, subprocess
if len(sys.argv) == 1:
    for m in ("0", "1", "2", "3", "4"):
        env = dict(os.environ, TCMI_SPLIT_MODE=m)
        subprocess.run([sys.executable, __file__, m], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd"))
import torch
from tcmi import _lib
# the mode switch exists in the probe build only (make -C tensorcircuit-ng_amd/csrc libtcmi_probe.so)
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd", "csrc", "libtcmi_probe.so")
L = _lib.lib()
M = N = 4096; B = 8
st = torch.cuda.current_stream().cuda_stream
res = []
for K in (256, 512, 1024):
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    f = lambda: _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "x")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res.append((K, e0.elapsed_time(e1) / 10))
    if sys.argv[1] == "4":
        v = c[0, 0, 0]
        dc, dr = float(v.real), float(v.imag)
        print(f"   K={K}: workgroup 800 lived {dc:.0f} shader cycles = {dr / 100:.2f} us -> {dc / dr * 100:.0f} MHz; "
              f"{dc / (K // 16):.0f} cycles per 16-k step incl. prologue and epilogue (MFMA floor 2304)")
print("mode", sys.argv[1], " ".join(f"K={k}: {t:.3f} ms" for k, t in res),
      f" per 16-k step per workgroup: {(res[2][1] - res[0][1]) / 48 / 32 * 1e3:.3f} us (MFMA floor 0.96 us at 2.4 GHz)")
