"""Latency of small circuits (config 1 scale): Circuit API, compiled plan, hipGraph replay, value_and_grad."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi.executor import GraphedState
from oracle import workloads as W
tc.set_backend("hip")
for dt in ("complex64", "complex128"):
    tc.set_dtype(dt)
    for n, d in ((10, 4), (16, 6)):
        rdt = np.float32 if dt == "complex64" else np.float64
        p = tc.backend.convert_to_tensor(np.random.default_rng(0).normal(size=[2 * d, n]).astype(rdt))
        def wf(p):
            c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix); return c.wavefunction()
        def en(p):
            c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            e = 0.0
            for i in range(n): e += -1.0 * c.expectation_ps(x=[i])
            for i in range(n - 1): e += c.expectation_ps(z=[i, i + 1])
            return tc.backend.real(e)
        vg = tc.backend.value_and_grad(en)
        def timeit(f, reps=20):
            f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): f()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
        c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix); cc = c._compiled()
        pm = c._param_tensor().reshape(1, -1)
        gs = GraphedState(cc, 1)
        print(f"{dt} n={n} d={d}: Circuit.wavefunction {timeit(lambda: wf(p)):.3f} ms | plan.state {timeit(lambda: cc.state(pm)):.3f} ms | "
              f"hipGraph replay {timeit(lambda: gs(pm)):.3f} ms | value_and_grad {timeit(lambda: vg(p)):.3f} ms", flush=True)
# traced jit
tc.set_dtype("complex64")
for n, d in ((10, 4), (16, 6), (24, 8)):
    p = tc.backend.convert_to_tensor(np.random.default_rng(0).normal(size=[2 * d, n]).astype(np.float32))
    def en(p):
        c = tc.Circuit(n); W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n): e += -1.0 * c.expectation_ps(x=[i])
        for i in range(n - 1): e += c.expectation_ps(z=[i, i + 1])
        return tc.backend.real(e)
    plain = tc.backend.value_and_grad(en); fast = tc.backend.jit(tc.backend.value_and_grad(en))
    def timeit(f, reps=20):
        f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print(f"VQE step n={n} d={d}: value_and_grad {timeit(lambda: plain(p)):.3f} ms | jit(value_and_grad) {timeit(lambda: fast(p)):.3f} ms  {fast.stats}", flush=True)
