"""One-sided Jacobi SVD (256 x 256 complex64) on spectra graded over 1 .. 6 decades: the plain kernel against the
QR-preconditioned path (linalg.SVD_PRECONDITION): sweeps, milliseconds (HIP events), accuracy."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import linalg as LA
tc.set_backend("hip"); tc.set_dtype("complex64")
rng = np.random.default_rng(0)
def haar(k):
    z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k)); q, r = np.linalg.qr(z); return q * (np.diag(r) / abs(np.diag(r)))
m = 256
eye = torch.eye(m, device="cuda")
for dec in (1, 2, 3, 4, 5, 6):
    a_np = ((haar(m) * np.logspace(0, -dec, m)) @ haar(m)).astype(np.complex64)
    a = torch.from_numpy(a_np).cuda()
    ref = np.linalg.svd(a_np.astype(np.complex128), compute_uv=False)
    out = {"decades": dec}
    for name, pre in (("plain", False), ("precond", True)):
        LA.SVD_PRECONDITION = pre
        LA.svd_trunc(a, max_singular_values=m); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); u, s, vh, _r = LA.svd_trunc(a, max_singular_values=m); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        sr = s.real.cpu().numpy()
        out[name] = {"ms": round(float(np.median(ts)), 3), "sweeps": LA.last_svd_sweeps(a.device),
                     "recon": float(((u * s.reshape(1, -1)) @ vh - a).abs().max()),
                     "sv_err": float(np.abs(sr - ref).max() / ref[0]),
                     "orth_u": float((u.conj().t() @ u - eye).abs().max()), "orth_v": float((vh @ vh.conj().t() - eye).abs().max())}
    print(json.dumps(out), flush=True)
