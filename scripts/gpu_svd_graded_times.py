import os, sys, time
sys.path.insert(0, "/root/repo/tensorcircuit-ng_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tcmi as tc
from tcmi import linalg as LA
tc.set_backend("hip"); tc.set_dtype("complex64")
rng = np.random.default_rng(0)
def haar(k):
    z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k)); q, r = np.linalg.qr(z); return q * (np.diag(r) / abs(np.diag(r)))
m=256
for dec in (0,1,2,3,4,5,6):
    a_np = (haar(m) * np.logspace(0, -dec, m)) @ haar(m)
    a = torch.from_numpy(a_np.astype(np.complex64)).cuda()
    LA.svd_trunc(a, max_singular_values=m); torch.cuda.synchronize()
    ts=[]
    for _ in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); LA.svd_trunc(a, max_singular_values=m); e1.record(); torch.cuda.synchronize(); ts.append(round(e0.elapsed_time(e1),2))
    ctl = LA._WORK[("svd", LA._devkey(a.device))][:256].view(torch.int32).cpu().numpy()
    print(dec, ts, "sweeps", int((ctl[2:62] > 0).sum()) + 1, "rot/sweep", ctl[2:30].tolist(), flush=True)
