"""Replay stress of the sliced value_and_grad graphs (level-batched launches, two-stream slice pairs, traced node
function): 3 parameter points visited 20 times each in random order; every result must equal the first one computed
for its point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
tc.set_backend("hip"); tc.set_dtype("complex64")
nq, dq = 20, 4
rng = np.random.default_rng(1)
pts = [tc.backend.convert_to_tensor(rng.uniform(0.2, 1.2, [nq, dq, 2]).astype(np.float32)) for _ in range(3)]
def nodes(params):
    c = tc.Circuit(nq)
    for i in range(dq):
        for j in range(nq - 1): c.rzz(j, j + 1, theta=params[j, i, 0])
        for j in range(nq): c.rx(j, theta=params[j, i, 1])
    return c.expectation_before([tc.gates.z(), [nq // 2]], reuse=False)
dc = tc.experimental.DistributedContractor(nodes, pts[0], {"slicing_opts": {"target_slices": 8}, "max_repeats": 16, "minimize": "combo"})
ref = {}
bad = 0
order = rng.integers(0, 3, 60)
for it, k in enumerate(order):
    v, g = dc.value_and_grad(pts[k])
    if it % 7 == 0: torch.cuda.synchronize()
    key = int(k)
    if key not in ref:
        ref[key] = (v.clone(), g.clone())
    else:
        dv = abs(float(v) - float(ref[key][0])); dg = float((g - ref[key][1]).abs().max())
        if dv > 1e-6 or dg > 1e-6:
            bad += 1; print("MISMATCH", it, key, dv, dg)
print(f"vjp replay stress: {len(order)} calls, {bad} mismatches, mode {dc._trace_state['mode']}, slices {dc.tree.nslices}")
