"""Probe (libtcmi_probe.so, TCMI_SPLIT_MODE=4): time stamps of workgroup 100 after each of its tiles -- is the per-tile time
uniform, and what do the first and last tiles cost?  M = N = 4096, batch 8; K = 128 and 256."""
import sys, os
os.environ["TCMI_SPLIT_MODE"] = "4"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd"))
import torch
from tcmi import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd", "csrc", "libtcmi_probe.so")
L = _lib.lib()
M = N = 4096
st = torch.cuda.current_stream().cuda_stream
for K, B in ((128, 8), (256, 8), (128, 32)):
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    for _ in range(3):
        _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "x")
    torch.cuda.synchronize()
    nt = 4 * B
    v = torch.view_as_real(c[0, 0, : nt + 1]).cpu().numpy()
    real = v[1:, 1] / 100.0          # us (100 MHz counter)
    cyc = v[1:, 0]
    d = [real[0]] + [real[i] - real[i - 1] for i in range(1, nt)]
    print(f"K={K} batch {B}: whole life {v[0, 1] / 100:.1f} us, {v[0, 0] / v[0, 1] * 100:.0f} MHz; tiles (us): " +
          " ".join(f"{x:.1f}" for x in d[: 40]))
