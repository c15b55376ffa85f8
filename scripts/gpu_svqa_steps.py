"""Per-step timing of the eager sliced reverse sweep (one slice): which tensordots cost what.
TCMI_TN_GRAPH=0 python scripts/gpu_svqa_steps.py"""
import os, sys, time, collections
os.environ["TCMI_TN_GRAPH"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn
n, d, S = 30, 8, 8
tc.set_backend("hip"); tc.set_dtype("complex64")
pt = tc.backend.convert_to_tensor(np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32))
def nodes(params):
    c = tc.Circuit(n)
    for i in range(d):
        for j in range(n - 1): c.rzz(j, j + 1, theta=params[j, i, 0])
        for j in range(n): c.rx(j, theta=params[j, i, 1])
    return c.expectation_before([tc.gates.z(), [n // 2]], reuse=False)
dc = tc.experimental.DistributedContractor(nodes, pt, {"slicing_opts": {"target_slices": S}, "max_repeats": 32, "minimize": "combo"})
arrays = dc._arrays(pt.clone().requires_grad_(True))
fop = lambda x: x.sum().real
dc.tree.contract_slices_vjp(arrays, [0], fop); torch.cuda.synchronize()
log = []
orig = tn._tensordot_raw
def timed(a, b, xa, xb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(a, b, xa, xb)
    torch.cuda.synchronize(); log.append((time.perf_counter() - t0, a.dim(), b.dim(), len(xa), r.dim()))
    return r
tn._tensordot_raw = timed
dc.tree.contract_slices_vjp(arrays, [0], fop); torch.cuda.synchronize()
tn._tensordot_raw = orig
tot = sum(x[0] for x in log)
print(f"{len(log)} tensordots, {tot*1e3:.2f} ms with per-step syncs")
big = [x for x in log if max(x[1], x[2]) >= 16]
print(f"{len(big)} with an operand of rank >= 16: {sum(x[0] for x in big)*1e3:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for t, ra, rb, nk, ro in big:
    k = (max(ra, rb), min(ra, rb), nk, ro); agg[k][0] += 1; agg[k][1] += t
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  rank {k[0]:2d} x {k[1]:2d}, {k[2]:2d} contracted -> rank {k[3]:2d}: {c:3d} calls, {t*1e6/c:7.1f} us each, {t*1e3:6.2f} ms")
