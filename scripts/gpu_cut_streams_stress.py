"""Stress check of the two-stream cut contraction: the same batches of parameters through backend.jit(backend.vmap(f))
with the right half on a second stream and with everything on one stream; results must be identical.
gpu_cut_streams_stress.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tc.set_backend("hip"); tc.set_dtype("complex64")
n, d = 24, 8
def f(p):
    c = tc.Circuit(n)
    W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
    return c.wavefunction()
fv = tc.backend.jit(tc.backend.vmap(f))
g = torch.Generator().manual_seed(0)
bad = 0
for B in (8, 1, 3):
    for i in range(reps):
        p = (torch.rand(B, 2 * d, n, generator=g) * 6.28).cuda()
        os.environ["TCMI_CUT_STREAMS"] = "1"
        a = fv(p)
        junk = torch.randn(1 << 22, device="cuda")          # allocator churn between the calls
        os.environ["TCMI_CUT_STREAMS"] = "0"
        b = fv(p)
        del junk
        if not torch.equal(a, b):
            bad += 1
            print("MISMATCH", B, i, float((a - b).abs().max()))
        if i % 50 == 0:
            nrm = float((a.abs() ** 2).sum(-1).max())
            assert abs(nrm - 1.0) < 1e-4, nrm
print(f"cut two-stream stress: {3 * reps} comparisons, {bad} mismatches")
