"""Sweep counts of cyclic one-sided Jacobi (row form, the pairing order of csrc/tcmi_mps.hip) on 256 x 256 matrices with
flat and graded spectra, raw and after QR preconditioning: python scripts/experiments/jacobi_sweeps.py (numpy, CPU)."""
import numpy as np, scipy.linalg as sla
rng = np.random.default_rng(0)
def haar(k):
    z = rng.normal(size=(k, k)) + 1j * rng.normal(size=(k, k)); q, r = np.linalg.qr(z); return q * (np.diag(r) / abs(np.diag(r)))
def sweeps(W, tol=6e-8 * 16, maxs=40):
    """row one-sided Jacobi, round-robin pairs; returns number of sweeps with any rotation"""
    W = W.copy(); p = W.shape[0]; M = p - 1
    idx = np.arange(p)
    ns = 0
    for s in range(maxs):
        rot = 0
        for R in range(M):
            # round-robin pairing: player p-1 fixed
            i = np.empty(p // 2, dtype=int); j = np.empty(p // 2, dtype=int)
            i[0], j[0] = R % M, M
            for k in range(1, p // 2):
                i[k] = (R + k) % M; j[k] = (R - k + M) % M
            x, y = W[i], W[j]
            al = (abs(x) ** 2).sum(1); be = (abs(y) ** 2).sum(1); g = (x * y.conj()).sum(1)
            ag = abs(g)
            act = (ag ** 2 > tol ** 2 * al * be) & (ag ** 2 > 1e-300)
            rot += act.sum()
            if not act.any(): continue
            ags = np.where(act, ag, 1.0)
            zeta = (be - al) / (2 * ags)
            t = np.sign(zeta + (zeta == 0)) / (abs(zeta) + np.sqrt(1 + zeta ** 2))
            c = 1 / np.sqrt(1 + t ** 2); sn = c * t
            ph = g / ags
            c = np.where(act, c, 1.0); sn = np.where(act, sn, 0.0); ph = np.where(act, ph, 1.0)
            yt = ph[:, None] * y
            W[i] = c[:, None] * x - sn[:, None] * yt
            W[j] = sn[:, None] * x + c[:, None] * yt
        ns += 1
        if rot == 0: break
    return ns
m = 256
for dec in (0, 2, 4, 6):
    a = rng.normal(size=(m, m)) + 1j * rng.normal(size=(m, m)) if dec == 0 else (haar(m) * np.logspace(0, -dec, m)) @ haar(m)
    res = {}
    res["raw"] = sweeps(a)
    nrm = (abs(a) ** 2).sum(1); res["sorted rows"] = sweeps(a[np.argsort(-nrm)])
    q, r = np.linalg.qr(a.conj().T)          # a^H = Q R  ->  a = R^H Q^H ; rows of R^H... use L = R^H
    res["QR, rows of R"] = sweeps(r)       # a^H = QR: SVD of r same singular values; rows of R
    res["QR, rows of R^H"] = sweeps(r.conj().T)
    q, r, piv = sla.qr(a.conj().T, pivoting=True)
    res["QRP, rows of R"] = sweeps(r); res["QRP, rows of R^H"] = sweeps(r.conj().T)
    print(f"decay 1e-{dec}:", res, flush=True)
