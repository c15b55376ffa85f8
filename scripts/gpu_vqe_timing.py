"""Timing exploration of the VQE step (value_and_grad of TFIM energy) on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import workloads as W

tc.set_backend("hip"); tc.set_dtype("complex64")

def tfim(c, n):
    e = 0.0
    for i in range(n): e += -1.0 * c.expectation((tc.gates.x(), [i]))
    for i in range(n - 1): e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
    return tc.backend.real(e)

def run(n, d, B, reps=3):
    def f(p):
        c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return tfim(c, n)
    params = tc.backend.convert_to_tensor(np.random.default_rng(n).normal(0, 0.1, [B, 2*d, n]).astype(np.float32))
    fn = tc.backend.vvag(f, argnums=0, vectorized_argnums=0)
    t0 = time.time(); v, g = fn(params); torch.cuda.synchronize(); t_first = time.time() - t0
    from tcmi import executor as X
    X.EVENT_LOG = []
    t0 = time.time()
    for _ in range(reps): v, g = fn(params)
    torch.cuda.synchronize(); t = (time.time() - t0) / reps
    tags = {}
    for tag, e0, e1, launches, work in X.EVENT_LOG:
        tt = tags.setdefault(tag, [0.0, 0, 0]); tt[0] += e0.elapsed_time(e1) / reps; tt[1] += launches / reps; tt[2] += 1 / reps
    X.EVENT_LOG = None
    print("   per step: " + "  ".join(f"{k} {v_[0]:.1f} ms ({v_[1]:.0f} launches, {v_[2]:.0f} calls)" for k, v_ in tags.items()), flush=True)
    print(f"VQE n={n} d={d} B={B}: first {t_first:.3f}s steady {t*1e3:.1f} ms/step  E0={float(v[0]):.5f} |g|={float(g.norm()):.4f} mem={torch.cuda.max_memory_allocated()/2**30:.1f}GiB", flush=True)
    # decomposition: forward only / measure only
    c = tc.Circuit(n); W.hea_b(c, n, d, params[0], zz=tc.gates._zz_matrix)
    cc = c._compiled(); cc = getattr(cc, 'full', cc); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): st = cc.state(p, full=True)
    torch.cuda.synchronize(); tf = (time.time() - t0) / reps
    g_ = torch.randn_like(st)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): gp = cc.vjp(p, st, g_)
    torch.cuda.synchronize(); tb = (time.time() - t0) / reps
    adj = cc._adjoint(full=False)
    print(f"   forward {tf*1e3:.1f} ms ({len(cc.descs)} passes)  adjoint sweep {tb*1e3:.1f} ms ({len(adj['descs'])} passes, cfg R{adj['cfg'].R})", flush=True)

import sys
for a in (sys.argv[1:] or ['16,4,4', '20,6,4', '24,8,1', '24,8,8', '26,10,2', '28,12,1']):
    n_, d_, b_ = (int(x) for x in a.split(','))
    run(n_, d_, b_, reps=2 if n_ >= 28 else 3)
