"""Small VQE workload for profiling the adjoint sweep: n, depth, batch from argv; 3 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W

tc.set_backend("hip"); tc.set_dtype("complex64")
n, d, B = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (24, 8, 2)))

def f(p):
    c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
    e = 0.0
    for i in range(n): e += -1.0 * c.expectation((tc.gates.x(), [i]))
    for i in range(n - 1): e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
    return tc.backend.real(e)

params = tc.backend.convert_to_tensor(np.random.default_rng(n).normal(0, 0.1, [B, 2*d, n]).astype(np.float32))
fn = tc.backend.vvag(f, argnums=0, vectorized_argnums=0)
v, g = fn(params); torch.cuda.synchronize()
t0 = time.time()
for _ in range(3): v, g = fn(params)
torch.cuda.synchronize()
print(f"n={n} d={d} B={B}: {(time.time()-t0)/3*1e3:.1f} ms/step  E0={float(v[0]):.5f}")
