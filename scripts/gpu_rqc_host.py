"""Host-side split of DistributedContractor.value on config 4: nodes_fn / leaf staging / graph replays / all-reduce."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn as TN
from tcmi.experimental import DistributedContractor
tc.set_backend("hip"); tc.set_dtype("complex64")
rows, cols, depth = 4, 8, 16
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
dc = DistributedContractor(nodes_fn, None, cotengra_options={"slicing_opts": {"target_size": 2 ** 27}, "max_repeats": 128})
for _ in range(2): v = dc.value(None, op=lambda x: x)
torch.cuda.synchronize()
def T(f, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
t_all, _ = T(lambda: dc.value(None, op=lambda x: x))
t_nodes, nodes = T(lambda: nodes_fn(None))
arrays = [n.tensor for n in nodes]
t_sum, _ = T(lambda: dc._local_sum(None, None))
def only_slices():
    for r in dc.tree.contract_slices(arrays, dc.my_slices): pass
t_sl, _ = T(only_slices)
print(f"value {t_all:.2f} ms | nodes_fn {t_nodes:.2f} | _local_sum (nodes_fn + slices + adds) {t_sum:.2f} | contract_slices only {t_sl:.2f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); only_slices(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
