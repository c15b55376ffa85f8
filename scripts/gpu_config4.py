"""Config 4: single amplitude <0^32|C|0^32> of a 32-qubit random circuit on a 4x8 grid
(brickwork of Haar-random two-qubit gates, reference gates.py:852-863), complex64, through
DistributedContractor: random-greedy path search + slicing to ``target_size``, slices summed on the
device.  Checks sliced sum == less-sliced contraction and reports time / achieved TFLOP/s."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
sys.path.insert(0, ROOT)
import tcmi as tc  # noqa: E402
from tcmi import tn  # noqa: E402
from tcmi.experimental import DistributedContractor  # noqa: E402


def rqc(rows, cols, depth, gates):
    c = tc.Circuit(rows * cols)
    q = lambda r, cc: r * cols + cc
    k = 0
    for d in range(depth):
        pat = d % 4
        pairs = []
        if pat in (0, 1):
            for r in range(rows):
                for cc in range(pat, cols - 1, 2):
                    pairs.append((q(r, cc), q(r, cc + 1)))
        else:
            for r in range(pat - 2, rows - 1, 2):
                for cc in range(cols):
                    pairs.append((q(r, cc), q(r + 1, cc)))
        for a, b in pairs:
            c.any(a, b, unitary=gates[k])
            k += 1
    return c


def main():
    rows, cols = 4, 8
    depth = int(os.environ.get("RQC_DEPTH", 16))
    logt = int(os.environ.get("RQC_LOG2_TARGET", 27))
    ngates = depth * rows * cols
    gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(ngates)]
    bits = "0" * (rows * cols)

    def nodes_fn(_):
        return rqc(rows, cols, depth, gates).amplitude_before(bits)

    out = {"grid": [rows, cols], "depth": depth, "log2_target": logt}
    t0 = time.perf_counter()
    dc = DistributedContractor(nodes_fn, None, cotengra_options={"slicing_opts": {"target_size": 2 ** logt},
                                                                 "max_repeats": 128})
    out["path_search_s"] = time.perf_counter() - t0
    tree = dc.tree
    out.update(nslices=tree.nslices, width=tree.contraction_width(),
               log2_flops_per_slice=float(np.log2(tree.total_flops() / tree.nslices)),
               log2_flops_total=float(np.log2(tree.total_flops())))
    t0 = time.perf_counter()
    v = dc.value(None, op=lambda x: x)
    torch.cuda.synchronize()
    out["first_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    v = dc.value(None, op=lambda x: x)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    out.update(contract_s=t, tflops=tree.total_flops() / t / 1e12, amplitude=[float(v.real), float(v.imag)],
               peak_mem_GiB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
    if os.environ.get("RQC_CHECK", "1") == "1":
        logt2 = int(os.environ.get("RQC_LOG2_CHECK", 30))
        dc2 = DistributedContractor(nodes_fn, None, cotengra_options={"slicing_opts": {"target_size": 2 ** logt2},
                                                                      "max_repeats": 128, "seed": 1})
        v2 = dc2.value(None, op=lambda x: x)
        out.update(check_nslices=dc2.tree.nslices, check_amplitude=[float(v2.real), float(v2.imag)],
                   abs_diff=float(abs(v - v2)), rel_diff=float(abs(v - v2) / abs(v2)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
