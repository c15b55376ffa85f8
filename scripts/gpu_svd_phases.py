"""Phase timing inside the Jacobi SVD kernel: gpu_svd_phases.py [m] [n]
Needs the probe build (`make -C tensorcircuit-ng_amd/csrc libtcmi_probe.so`): workgroup 0 accumulates shader-clock
time per phase of a chip-wide round into control words 48..53."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi._lib as L
L.LIB_PATH = os.path.join(ROOT, "tensorcircuit-ng_amd", "csrc", "libtcmi_probe.so")
import tcmi as tc
from tcmi import linalg as LA
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tc.set_backend("hip"); tc.set_dtype("complex64")
rng = np.random.default_rng(0)
a = rng.normal(size=(m, n)) + 1j * rng.normal(size=(m, n))
A = torch.tensor(a, dtype=torch.complex64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u, s, vh, rest = LA.svd_trunc(A, max_singular_values=min(m, n) // 2, absorb=2)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
w = LA._WORK[("svd", LA._devkey(A.device))][:256].view(torch.int32).cpu().numpy()
sweeps = int((w[2:42] != 0).sum()) + 1
ph = w[48:54].astype(np.int64) * 16
names = ["row_in", "intra", "cross", "row_out", "barrier", "total(sweeps)"]
print(f"{m}x{n}: {t*1e3:.3f} ms wall, sweeps {sweeps}")
tot = ph[5]
for k, v in zip(names, ph):
    print(f"  {k:14s} {v:12d} ticks  {100.0*v/tot:5.1f} %  -> {t*1e3*v/tot:.3f} ms (if sweeps were the whole call)")
inner = w[54:61].astype(np.int64) * 16
nround = sweeps * (m // 8 - 1) * 8
print("inside a cross round (wave 0, cycles per round; the probes add s_waitcnt + s_memtime each):")
for k, v in zip(["load", "gram", "wave sums", "rotation", "apply+store", "barrier", "other"], inner):
    print(f"  {k:12s} {v / max(nround, 1):8.1f}")
hw = w[40:48].astype(np.int64)
print("HW_ID of the 8 waves of workgroup 0: " + ", ".join(f"wave {k}: simd {int(v >> 4) & 3} cu {int(v >> 8) & 15} se {int(v >> 13) & 7}" for k, v in enumerate(hw)))
