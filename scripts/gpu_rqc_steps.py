"""Per-step profile of the config-4 contraction (32-qubit 4x8 RQC amplitude): gpu_rqc_steps.py [depth] [log2_target]
Prints, for one full contraction, every tensordot step with operand ranks, contracted axes, route and HIP-event time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn as TN
from tcmi.experimental import DistributedContractor
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lt = int(sys.argv[2]) if len(sys.argv) > 2 else 27
tc.set_backend("hip"); tc.set_dtype("complex64")
rows, cols = 4, 8
gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(depth * rows * cols)]
q = lambda r, c: r * cols + c
def nodes_fn(_):
    c = tc.Circuit(rows * cols); k = 0
    for d in range(depth):
        pat = d % 4
        if pat in (0, 1): pairs = [(q(r, cc), q(r, cc + 1)) for r in range(rows) for cc in range(pat, cols - 1, 2)]
        else: pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, rows - 1, 2) for cc in range(cols)]
        for a, b in pairs:
            c.any(a, b, unitary=gates[k]); k += 1
    return c.amplitude_before("0" * (rows * cols))
dc = DistributedContractor(nodes_fn, None, cotengra_options={"slicing_opts": {"target_size": 2 ** lt}, "max_repeats": 128})
v = dc.value(None, op=lambda x: x); torch.cuda.synchronize()
log = []
orig_td, orig_sc, orig_pr, orig_gm = TN.tensordot, TN._tensordot_scattered, TN._permute_raw, TN._gemm_raw
cur = {}
def td(a, b, xa, xb):
    cur.clear(); cur.update(route="gemm", perm=0, perm_us=[])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_td(a, b, xa, xb); e1.record()
    log.append((a.dim(), b.dim(), len(xa), r.dim(), cur["route"], cur["perm"], e0, e1))
    return r
def sc(*a, **k):
    r = orig_sc(*a, **k)
    if r is not None: cur["route"] = "scat"
    return r
def pr(t, perm):
    if tuple(perm) != tuple(range(t.dim())): cur["perm"] = cur.get("perm", 0) + 1
    return orig_pr(t, perm)
TN.tensordot, TN._tensordot_scattered, TN._permute_raw = td, sc, pr
t0 = time.perf_counter(); v = dc.value(None, op=lambda x: x); torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"contract {t*1e3:.1f} ms (instrumented), nslices {dc.tree.nslices}, steps logged {len(log)}")
rows_ = [(ra, rb, nk, ro, route, npm, e0.elapsed_time(e1) * 1e3) for ra, rb, nk, ro, route, npm, e0, e1 in log]
tot = sum(r[-1] for r in rows_)
print(f"sum of step event times {tot/1e3:.1f} ms")
big = [r for r in rows_ if max(r[0], r[1]) >= 20]
print(f"steps with an operand >= 2^20: {len(big)}, {sum(r[-1] for r in big)/1e3:.1f} ms; small steps {len(rows_)-len(big)}, {sum(r[-1] for r in rows_ if max(r[0], r[1]) < 20)/1e3:.1f} ms")
import collections
agg = collections.OrderedDict()
for ra, rb, nk, ro, route, npm, us in big:
    k = (max(ra, rb), min(ra, rb), nk, ro, route, npm); a_ = agg.setdefault(k, [0, 0.0]); a_[0] += 1; a_[1] += us
print("big, small, nk, out, route, permutes: count, total us, avg us, GB/s (in+out), TF")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    by = 8.0 * (2 ** k[0] + 2 ** k[1] + 2 ** k[3]); fl = 8.0 * 2 ** (k[0] + k[1] - k[2])
    print(k, c, f"{us:.0f} {us/c:.0f} {by/(us/c*1e-6)/1e9:.0f} {fl/(us/c*1e-6)/1e12:.1f}")
if "--seq" in sys.argv:
    for r in rows_[-(len(rows_) // dc.tree.nslices if dc.tree.nslices else len(rows_)):]:
        print(r)
