import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W
tc.set_backend("hip"); tc.set_dtype("complex64")
n, d = 10, 4
p = tc.backend.convert_to_tensor(np.random.default_rng(0).normal(size=[2 * d, n]).astype(np.float32))
c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix); cc = c._compiled()
pm = c._param_tensor().reshape(1, -1)
st = cc.state(pm, full=True); g = torch.randn_like(st)
adj = cc._adjoint()
print("forward passes", len(cc.descs), "adjoint passes", len(adj["descs"]), "cfg", adj["cfg"], "n_exec", cc.n_exec)
def timeit(f, reps=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("state ms", timeit(lambda: cc.state(pm, full=True)), "vjp ms", timeit(lambda: cc.vjp(pm, st, g)))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): cc.vjp(pm, st, g)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000])
