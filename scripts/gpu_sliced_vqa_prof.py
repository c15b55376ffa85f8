import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
n, d, S = 30, 8, 8
tc.set_backend("hip"); tc.set_dtype("complex64")
pt = tc.backend.convert_to_tensor(np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32))
def nodes(params):
    c = tc.Circuit(n)
    for i in range(d):
        for j in range(n - 1): c.rzz(j, j + 1, theta=params[j, i, 0])
        for j in range(n): c.rx(j, theta=params[j, i, 1])
    return c.expectation_before([tc.gates.z(), [n // 2]], reuse=False)
dc = tc.experimental.DistributedContractor(nodes, pt, {"slicing_opts": {"target_slices": S}, "max_repeats": 8, "minimize": "combo"})
dc.value_and_grad(pt); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
v, g = dc.value_and_grad(pt); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
