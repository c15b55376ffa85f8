"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle.dense cross-checked
against oracle.tn).  The reference itself cannot be imported in the build container,
so these vectors are produced by the independent dense simulator and pinned to the reference only
through the known-answer tests in tests/test_oracle_kat.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dense, tn, workloads as W  # noqa: E402


def fd_grad(f, x, eps=1e-6):
    g = np.zeros_like(x)
    for i in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[i] += eps
        xm[i] -= eps
        g[i] = (f(xp) - f(xm)) / (2 * eps)
    return g


def main():
    out = {}
    # HEA-B states: seeds of section 8(d)
    for n, d in [(4, 2), (8, 3), (10, 4), (12, 4)]:
        params = np.random.default_rng(100 * n + d).normal(0, 1, [2 * d, n])
        psi = dense.run(n, W.hea_b_ops(n, d, params))
        c = tn.Circuit(n)
        W.hea_b(c, n, d, params)
        assert np.abs(c.wavefunction() - psi).max() < 1e-12
        out[f"hea_b_{n}_{d}_params"] = params
        out[f"hea_b_{n}_{d}_state"] = psi
    # config 1 (ones params, tests/test_circuit.py:941) and HEA-A
    n, d = 10, 4
    out["hea_b_10_4_ones_state"] = dense.run(n, W.hea_b_ops(n, d, np.ones([2 * d, n])))
    pa = np.random.default_rng(5).uniform(0, 2 * np.pi, [3, 9])
    out["hea_a_9_3_params"] = pa
    out["hea_a_9_3_state"] = dense.run(9, W.hea_a_ops(9, 3, pa))
    # TFIM energies + central-difference gradients
    for n, d in [(6, 2), (10, 4)]:
        params = np.random.default_rng(7 * n + d).normal(0, 0.5, [2 * d, n])
        f = lambda p, n=n, d=d: W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, p)), n)
        e = f(params)
        c = tn.Circuit(n)
        W.hea_b(c, n, d, params)
        assert abs(W.tfim_energy(c, n).real - e) < 1e-10
        out[f"tfim_{n}_{d}_params"] = params
        out[f"tfim_{n}_{d}_energy"] = np.array(e)
        out[f"tfim_{n}_{d}_grad"] = fd_grad(f, params)
    np.savez_compressed(os.path.join(HERE, "hea_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
