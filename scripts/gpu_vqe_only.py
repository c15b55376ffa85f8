"""ONLY the VQE step of bench.py's config-3 leg (HEA-B n qubits, depth d, TFIM value_and_grad through
backend.jit(backend.vvag(...)), micro-batch B), `steps` times after two staging calls -- the command the per-pass
rocprofv3 captures of the VQE kernels are taken over (kernel-trace stats and PMC passes): no random-cotangent vjp, no
dense-plan comparison, no other leg.  Prints the per-kernel HIP-event times of the timed steps.

    python scripts/gpu_vqe_only.py [n] [depth] [batch] [steps]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import executor as X

tc.set_backend("hip"); tc.set_dtype("complex64")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3


def energy(p):
    c = tc.templates.blocks.example_block(tc.Circuit(n), p, nlayers=d)
    e = 0.0
    for i in range(n):
        e += -1.0 * c.expectation((tc.gates.x(), [i]))
    for i in range(n - 1):
        e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
    return tc.backend.real(e)


vvag = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
params = torch.from_numpy(np.random.default_rng(28).normal(0, 0.1, [B, 2 * d, n]).astype(np.float32)).cuda()
for _ in range(2):
    v, g = vvag(params)
torch.cuda.synchronize()
X.EVENT_LOG = []
t0 = time.perf_counter()
for _ in range(steps):
    v, g = vvag(params)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
tags = {}
for tag, e0, e1, launches, work in X.EVENT_LOG:
    t = tags.setdefault(tag, [0.0, 0, 0.0])
    t[0] += e0.elapsed_time(e1) / steps; t[1] += launches / steps; t[2] += work / steps
X.EVENT_LOG = None
print(f"VQE step n={n} d={d} batch {B}: {el * 1e3:.2f} ms per call = {el * 1e3 / B:.3f} ms per sample; "
      f"E0 = {float(v[0]):.6f} |g| = {float(g.norm()):.5f}")
for k, (ms, ln, work) in tags.items():
    print(f"   {k}: {ms:.3f} ms per call, {ln:.0f} launches, {work / 1e9:.2f} GB algorithmic -> {work / ms / 1e6 / 8000:.3f} of 8 TB/s")
