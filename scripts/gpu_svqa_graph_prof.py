"""Replays ONE of the four graphs of the sliced value_and_grad (g_a invariant forward, g_b slice forward, g_c slice
backward, g_d invariant backward) many times, for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 scripts/gpu_svqa_graph_prof.py g_c 50"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
which, reps = sys.argv[1], int(sys.argv[2])
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
dc.value_and_grad(pt); torch.cuda.synchronize()
c = dc.tree._vjp_graph_cache
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): c[which].replay()
e1.record(); torch.cuda.synchronize()
print(which, f"{e0.elapsed_time(e1)/reps:.3f} ms per replay")
