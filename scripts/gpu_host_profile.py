"""cProfile of the host side of a small VQE step (where the milliseconds go when kernels take microseconds)."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W
tc.set_backend("hip"); tc.set_dtype("complex64")
n, d = 10, 4
p = tc.backend.convert_to_tensor(np.random.default_rng(0).normal(size=[2 * d, n]).astype(np.float32))
def en(p):
    c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
    e = 0.0
    for i in range(n): e += -1.0 * c.expectation_ps(x=[i])
    for i in range(n - 1): e += c.expectation_ps(z=[i, i + 1])
    return tc.backend.real(e)
vg = tc.backend.value_and_grad(en)
for _ in range(3): vg(p)
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "vg"
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    if which == "vg": vg(p)
    else:
        c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix); c.wavefunction()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
