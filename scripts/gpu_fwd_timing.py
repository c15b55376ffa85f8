"""Forward (state-vector plan) timing of HEA-B circuits: ms per state, achieved GB/s on the executed plan.
TCMI_VM1=1 selects the first-generation pass kernel (A/B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W

tc.set_backend("hip"); tc.set_dtype("complex64"); tc.set_contractor("plain")

def run(n, d, B, reps=5, check=False):
    params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n]).astype(np.float32)
    pt = tc.backend.convert_to_tensor(params)
    c = tc.Circuit(n); W.hea_b(c, n, d, pt, zz=tc.gates._zz_matrix)
    cc = c._compiled(); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
    st = cc.state(p, full=True); torch.cuda.synchronize()
    out = torch.empty_like(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): cc.state(p, out=out, full=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    npass = len(cc.descs)
    gbs = npass * 2 * B * 2**n * 8 / (ms * 1e-3) / 1e9
    msg = f"fwd n={n} d={d} B={B}: {ms:.3f} ms/step  {ms/B:.3f} ms/state  {npass} passes  {gbs:.0f} GB/s ({gbs/8000:.3f} of 8 TB/s)  norm={float((st[0].abs()**2).sum()):.6f}"
    if check:
        from oracle import dense
        ref = dense.run(n, W.hea_b_ops(n, d, params.astype(np.float64)))
        msg += f"  max|psi-oracle|={np.abs(st[0].cpu().numpy() - ref).max():.2e}"
    print(msg, flush=True)

print("kernel:", "vm1" if os.environ.get("TCMI_VM1") else "vm2", flush=True)
run(16, 4, 4, check=True)
run(20, 6, 4, check=True)
run(24, 8, 1)
run(24, 8, 8)
run(28, 12, 1, reps=3)
run(28, 12, 4, reps=2)
