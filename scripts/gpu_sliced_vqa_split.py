import os, sys, time
sys.path.insert(0, "/root/repo/tensorcircuit-ng_amd"); sys.path.insert(0, "/root/repo")
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
for _ in range(3):
    t0 = time.perf_counter(); v, g = dc.value_and_grad(pt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host {1e3*(t1-t0):.1f} ms, +sync {1e3*(t2-t1):.1f} ms")
c = dc.tree._vjp_graph_cache
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name in ("g_a", "g_b", "g_c", "g_d"):
    torch.cuda.synchronize(); e0.record(); c[name].replay(); e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1):.2f} ms")
