"""Probe: the split join GEMM built with 2 / 3 / 5 / 6 VALU instructions per MFMA in the sched_group_barrier pattern
(scripts/ubench/tmp_split_vpm*.so, built by hand) next to the shipped 4."""
import ctypes, glob, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
libs = [os.path.join(here, "..", "tensorcircuit-ng_amd", "csrc", "libtcmi.so")] + sorted(glob.glob(os.path.join(here, "ubench", "tmp_split_vpm*.so")))
M = N = 4096; K = 256; B = 8
A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
ref = None
for path in libs:
    L = ctypes.CDLL(path)
    L.tcmi_cgemm_split.argtypes = [vp, vp, vp, ll, ll, ll, ci, ll, ll, ll, vp]
    L.tcmi_cgemm_split.restype = ci
    f = lambda: L.tcmi_cgemm_split(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, st)
    assert f() == 0
    torch.cuda.synchronize()
    if ref is None:
        ref = c.clone()
    same = bool((c == ref).all())
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print(os.path.basename(path), " ".join(f"{t:.3f}" for t in ts), "ms; bit-identical to the shipped kernel:", same)
