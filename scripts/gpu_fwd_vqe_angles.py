"""Forward plan of the VQE ansatz at VQE-like (small) angles: gpu_fwd_vqe_angles.py n d batch reps
(TCMI_SHEAR2=0 switches the two-shear form off for an A/B comparison)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W
n, d, B, reps = (int(x) for x in sys.argv[1:5])
tc.set_backend("hip"); tc.set_dtype("complex64")
params = np.random.default_rng(n).normal(0, 0.1, [2 * d, n]).astype(np.float32)
c = tc.Circuit(n); W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
cc = c._compiled(); cc = getattr(cc, "full", cc); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
out = cc.state(p, full=True); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): cc.state(p, out=out, full=True)
e1.record(); torch.cuda.synchronize()
two = int((np.asarray(cc.plan.ginfo).reshape(-1, 8)[:, 6] != 0).sum())
print(f"fwd n={n} d={d} B={B} shear2={os.environ.get('TCMI_SHEAR2', '1')}: {e0.elapsed_time(e1)/reps/B:.3f} ms/state, {len(cc.descs)} passes, "
      f"{two} two-shear records, norm {float(torch.linalg.vector_norm(out[0])):.7f} amp0 {complex(out[0, 0]):.6f} amp5 {complex(out[0, 5]):.6f}")
