"""Probe: where the epilogue of tcmi_cgemm_split_epi spends its time (M = N = 4096, K = 128, batch 8 and 32).  Modes of the
probe library (TCMI_SPLIT_MODE): plain = tcmi_cgemm_split at the same K; 0 full epilogue; 7 without the multiply-adds;
8 results as 16-byte stores of neighbouring columns; 9 no result stores."""
import sys, os, subprocess
if len(sys.argv) == 1:
    for m in ("plain", "0", "7", "8", "9"):
        env = dict(os.environ, TCMI_SPLIT_MODE="0" if m == "plain" else m)
        subprocess.run([sys.executable, __file__, m], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd"))
import torch
from tcmi import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensorcircuit-ng_amd", "csrc", "libtcmi_probe.so")
L = _lib.lib()
M = N = 4096; K = 128
st = torch.cuda.current_stream().cuda_stream
out = []
for B in (8, 32):
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
    X = torch.view_as_complex(torch.randn(B, 16, 2, device="cuda"))
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    if sys.argv[1] == "plain":
        f = lambda: _lib.check(L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, st), "x")
    else:
        f = lambda: _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                                      X.data_ptr(), st), "x")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    out.append(f"batch {B}: {t:.3f} ms = {t * 1e3 / (B * 4):.2f} us per tile")
print("mode", sys.argv[1], "; ".join(out))
