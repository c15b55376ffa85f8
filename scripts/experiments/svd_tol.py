"""Jacobi SVD convergence threshold experiment: sweeps / time / accuracy for TCMI_SVD_TOL_SCALE builds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tcmi import _lib
alt = os.environ.get("TCMI_ALT_LIB")
if alt:
    _lib.LIB_PATH = os.path.join(ROOT, alt)
from tcmi import linalg as LA
rng = np.random.default_rng(0)
for n in (64, 256):
    a = (rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))).astype(np.complex64)
    ag = torch.from_numpy(a).cuda()
    u, s, vh, _ = LA.svd_trunc(ag)
    torch.cuda.synchronize()
    ctl = LA._WORK[("svd", ag.device)][:256].view(torch.int32).cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(5): LA.svd_trunc(ag)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    sr = np.linalg.svd(a.astype(np.complex128), compute_uv=False)
    un, sn, vn = u.cpu().numpy().astype(np.complex128), s.cpu().numpy().real.astype(np.float64), vh.cpu().numpy().astype(np.complex128)
    print(f"{alt or 'default'} n={n}: sweeps={int((ctl[2:62] > 0).sum()) + 1} time={dt*1e3:.2f} ms  max|ds|/s0={np.abs(sn - sr).max()/sr[0]:.2e} "
          f"recon={np.abs((un * sn) @ vn - a).max():.2e} orthU={np.abs(un.conj().T @ un - np.eye(n)).max():.2e} orthV={np.abs(vn @ vn.conj().T - np.eye(n)).max():.2e}")
