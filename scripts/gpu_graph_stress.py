"""Stress check of the HIP-graph replay path of ContractionTree.contract_slices on config 4: repeated values with and
without the counters hook against the eager path.  gpu_graph_stress.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import tn as TN
from tcmi.experimental import DistributedContractor
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
os.environ["TCMI_TN_GRAPH"] = "0"
ref = complex(dc.value(None, op=lambda x: x))
os.environ["TCMI_TN_GRAPH"] = "1"
bad = 0
for i in range(reps):
    TN.COUNTERS = TN.new_counters() if i % 2 else None
    v = complex(dc.value(None, op=lambda x: x))
    if abs(v - ref) > 2e-9: bad += 1; print("MISMATCH", i, v, ref)
TN.COUNTERS = None
print(f"graph replay stress: {reps} values, {bad} mismatches, reference {ref}")
