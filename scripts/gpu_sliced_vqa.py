"""Sliced value_and_grad of the rzz / rx ladder <Z_0> (examples/slicing_auto_pmap_vqa.py): gpu_sliced_vqa.py n d slices"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn as TN
n, d, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tc.set_backend("hip"); tc.set_dtype("complex64")
pv = np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32)
pt = tc.backend.convert_to_tensor(pv)
def nodes(params):
    c = tc.Circuit(n)
    for i in range(d):
        for j in range(n - 1): c.rzz(j, j + 1, theta=params[j, i, 0])
        for j in range(n): c.rx(j, theta=params[j, i, 1])
    return c.expectation_before([tc.gates.z(), [n // 2]], reuse=False)
t0 = time.perf_counter()
dc = tc.experimental.DistributedContractor(nodes, pt, {"slicing_opts": {"target_slices": S}, "max_repeats": 32, "minimize": "combo"})
print(f"search {time.perf_counter()-t0:.2f}s", dc.tree_info)
steps, dep, _, _ = dc.tree._symbolic_steps()
print("steps", len(steps), "invariant", sum(1 for s in steps if not dep[s[4]]))
for mode in ("1", "0"):
    os.environ["TCMI_TN_VJP"] = mode
    v, g = dc.value_and_grad(pt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): v, g = dc.value_and_grad(pt)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    print(f"TCMI_TN_VJP={mode}: value {float(v):.6f} |g| {float(g.norm()):.4f}  {t*1e3:.1f} ms per value_and_grad")
t0 = time.perf_counter()
for _ in range(3): vv = dc.value(pt)
torch.cuda.synchronize(); print(f"value only {(time.perf_counter()-t0)/3*1e3:.1f} ms")
