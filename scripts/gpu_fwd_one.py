"""One HEA-B forward configuration (state-vector plan), for profiling: gpu_fwd_one.py n d batch reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W
n, d, B, reps = (int(x) for x in sys.argv[1:5])
tc.set_backend("hip"); tc.set_dtype("complex64"); tc.set_contractor("plain")
params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n]).astype(np.float32)
c = tc.Circuit(n); W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
cc = c._compiled(); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
out = cc.state(p, full=True); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): cc.state(p, out=out, full=True)
e1.record(); torch.cuda.synchronize()
print(f"fwd n={n} d={d} B={B}: {e0.elapsed_time(e1)/reps:.3f} ms/step, {len(cc.descs)} passes")
