"""SVD kernel probe: gpu_svd_probe.py [m] [n] [dtype]  -- time, sweeps used (rotation counters in the control words)
for a Haar-like matrix and for a TEBD-like theta (decaying spectrum)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi._lib as L
if os.environ.get("TCMI_LIB"): L.LIB_PATH = os.environ["TCMI_LIB"]
import tcmi as tc
from tcmi import linalg as LA
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = sys.argv[3] if len(sys.argv) > 3 else "complex64"
tc.set_backend("hip"); tc.set_dtype(dt)
tdt = torch.complex64 if dt == "complex64" else torch.complex128
rng = np.random.default_rng(0)
def haar(k):
    z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k)); q, r = np.linalg.qr(z); return q * (np.diag(r) / abs(np.diag(r)))
cases = {"gaussian": rng.normal(size=(m, n)) + 1j * rng.normal(size=(m, n)),
         "decay 1e-6": (haar(m) * np.logspace(0, -6, m)) @ haar(n)[:m],
         "decay 1e-2": (haar(m) * np.logspace(0, -2, m)) @ haar(n)[:m]}
for name, a in cases.items():
    A = torch.tensor(a, dtype=tdt, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u, s, vh, rest = LA.svd_trunc(A, max_singular_values=min(m, n) // 2, absorb=2)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    w = LA._WORK[("svd", LA._devkey(A.device))][:256].view(torch.int32).cpu().numpy()
    sweeps = int((w[2:62] != 0).sum()) + 1
    sref = np.linalg.svd(a, compute_uv=False)
    err = float(np.abs(np.concatenate([s.cpu().numpy().real, rest.cpu().numpy().real]) - sref).max() / sref[0])
    uf, sf, vf, _ = LA.svd_trunc(A)                      # full thin SVD: orthogonality and reconstruction
    eye = torch.eye(min(m, n), dtype=tdt, device="cuda")
    ou = float((uf.mH @ uf - eye).abs().max()); ov = float((vf @ vf.mH - eye).abs().max())
    rec = float(((uf * sf[None, :]) @ vf - A).abs().max() / sref[0])
    print(f"{name:12s} {m}x{n} {dt}: {t*1e3:.2f} ms, sweeps {sweeps}, rotations/sweep {w[2:2+sweeps].tolist()}, sigma err {err:.1e}, |U^H U - 1| {ou:.1e}, |V V^H - 1| {ov:.1e}, recon {rec:.1e}")
