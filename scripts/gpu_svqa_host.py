"""Host-side profile of ONE rank of eight of the sliced value_and_grad (bench.py sliced_vqa.one_rank_of_8_sharded): where do
the milliseconds between the graph replays go?  usage: python scripts/gpu_svqa_host.py [n] [d] [slices]"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tc.set_backend("hip"); tc.set_dtype("complex64")
pt = tc.backend.convert_to_tensor(np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32))
def nodes(params):
    c = tc.Circuit(n)
    for i in range(d):
        for j in range(n - 1): c.rzz(j, j + 1, theta=params[j, i, 0])
        for j in range(n): c.rx(j, theta=params[j, i, 1])
    return c.expectation_before([tc.gates.z(), [n // 2]], reuse=False)
dc = tc.experimental.DistributedContractor(nodes, pt, {"slicing_opts": {"target_slices": S}, "max_repeats": 8, "minimize": "combo"})
def timeit(tag, reps=10):
    for _ in range(3):
        dc.value_and_grad(pt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        v, g = dc.value_and_grad(pt)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per value_and_grad, value {float(v):.6f} |g| {float(g.norm()):.5f}")
timeit("all slices on this rank")
# what rank 0 of 8 executes (its slice block, its share of the invariant forest), no collectives
dc._emulate_rank = (0, 8)
keep = dc.my_slices
from tcmi import distributed as D_
dc.my_slices = [int(s) for s in D_.slice_table(int(dc.tree.nslices), 8)[0] if s >= 0]
dc.tree._vjp_graph_cache = None
timeit("rank 0 of 8 (emulated)")
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    v, g = dc.value_and_grad(pt)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
