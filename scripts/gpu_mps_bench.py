"""Config 5: MPSCircuit n=64, chi=128, one TEBD sweep = 63 adjacent random SU(4) gates left to
right on a chi-saturated random MPS, complex64.  Prints sweeps/s, us per bond, and the SVD / QR kernel
latencies at the TEBD shapes; the CPU column is oracle/mps.py (numpy, LAPACK gesdd) on a few bonds."""
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
from tcmi import linalg as LA  # noqa: E402
from oracle import mps as omps, gates as OG  # noqa: E402


def random_mps(n, chi, rng):
    dims = [min(2 ** i, 2 ** (n - i), chi) for i in range(n + 1)]
    return [(rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1]))).astype(
        np.complex64) / np.sqrt(dims[i] * 2) for i in range(n)]


def main():
    n, chi = int(os.environ.get("MPS_N", 64)), int(os.environ.get("MPS_CHI", 128))
    rng = np.random.default_rng(64)
    tensors = random_mps(n, chi, rng)
    gates = [OG.random_two_qubit_gate(5000 + i).reshape(2, 2, 2, 2).astype(np.complex64) for i in range(n - 1)]
    split = tc.cons.split_rules(max_singular_values=chi)
    out = {"n": n, "chi": chi}

    t0 = time.perf_counter()
    m = tc.MPSCircuit(n, tensors=tensors, split=split)
    torch.cuda.synchronize()
    out["canonicalize_s"] = time.perf_counter() - t0

    def sweep():
        for i in range(n - 1):
            m.apply(tc.gates.Gate(gates[i]), i, i + 1)

    sweep()
    m.position(0)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        sweep()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        m.position(0)
        torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t.abs()).all()) for t in m.get_tensors()), "non-finite MPS tensor"
    best = min(times)
    out.update(sweep_s=best, sweeps_per_s=1 / best, us_per_bond=best / (n - 1) * 1e6,
               max_bond=max(m.get_bond_dimensions()), fidelity=float(m._fidelity))

    def timeit(f, reps=5):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    a = torch.from_numpy((rng.normal(size=(2 * chi, 2 * chi)) + 1j * rng.normal(size=(2 * chi, 2 * chi))).astype(
        np.complex64)).cuda()
    out["svd_%dx%d_ms" % (2 * chi, 2 * chi)] = timeit(lambda: LA.svd_trunc(a, max_singular_values=chi, absorb=1)) * 1e3
    LA.svd_trunc(a, max_singular_values=chi, absorb=1)
    torch.cuda.synchronize()
    ctl = LA._WORK[("svd", LA._devkey(a.device))][:256].view(torch.int32).cpu().numpy()
    out["svd_sweeps"] = int((ctl[2:62] > 0).sum()) + 1
    out["svd_barriers"] = int(ctl[0])
    b = a[:, :chi].contiguous()
    out["qr_%dx%d_ms" % (2 * chi, chi)] = timeit(lambda: LA.qr(b)) * 1e3
    an = a.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(3):
        np.linalg.svd(an, full_matrices=False)
    out["numpy_svd_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    if os.environ.get("MPS_TORCH_SVD", "1") == "1":
        out["rocsolver_svd_ms"] = timeit(lambda: torch.linalg.svd(a, full_matrices=False), reps=3) * 1e3

    # CPU column: the numpy oracle on the same sweep (bounded: first 16 bonds of the saturated region)
    o = omps.MPSCircuit(n, tensors=[t.astype(np.complex64) for t in tensors], split=omps.split_rules(max_singular_values=chi))
    o.position(n // 2 - 8)
    t0 = time.perf_counter()
    for i in range(n // 2 - 8, n // 2 + 8):
        o.apply(gates[i], i, i + 1)
    out["cpu_oracle_us_per_bond"] = (time.perf_counter() - t0) / 16 * 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
