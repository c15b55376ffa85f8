import os, sys, time
sys.path.insert(0, "/root/repo/tensorcircuit-ng_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tcmi as tc
from tcmi import linalg as LA, _lib
tc.set_backend("hip"); tc.set_dtype("complex64")
rng = np.random.default_rng(0)
def haar(k):
    z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k)); q, r = np.linalg.qr(z); return q * (np.diag(r) / abs(np.diag(r)))
m = n = 256
for dec in (2, 4, 5, 6):
    a = (haar(m) * np.logspace(0, -dec, m)) @ haar(n)
    A = torch.tensor(a, dtype=torch.complex64, device="cuda")
    nbytes = _lib.lib().tcmi_svd_work_bytes(m, n, 1, _lib.TCMI_C64)
    work = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    u = torch.empty(m, 128, dtype=torch.complex64, device="cuda"); s = torch.empty(m, device="cuda"); vh = torch.empty(128, n, dtype=torch.complex64, device="cuda")
    keep = torch.empty(1, dtype=torch.int32, device="cuda"); tw2 = torch.empty(1, device="cuda")
    for ms in (2, 6, 10, 14, 20, 30):
        ts = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.check(_lib.lib().tcmi_svd_trunc_batched(A.data_ptr(), u.data_ptr(), s.data_ptr(), vh.data_ptr(), keep.data_ptr(), tw2.data_ptr(), m, n, 128, 1, 128, -1.0, 0, 2, ms, work.data_ptr(), work.numel(), _lib.TCMI_C64, torch.cuda.current_stream().cuda_stream), "svd")
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        w = work[:256].view(torch.int32).cpu().numpy()
        print(f"decay 1e-{dec} max_sweeps {ms}: {min(ts)*1e3:.2f} ms  rot {w[2:2+ms].tolist()}")
