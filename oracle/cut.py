"""TEST INFRASTRUCTURE (oracle): dense numpy evaluation of the cut-contraction formula
psi[x_L, x_R] = sum_b w_b L_b[x_L] R_b[x_R] for a ``tcmi.cut.CutSpec`` -- what the reference's greedy path computes on a
shallow ladder circuit (tensorcircuit/cons.py:298-374: contract each half of the network, then join).  Used by the CPU
tests to validate ``tcmi.cut.make_cut`` against ``oracle.dense``; nothing in the product imports it."""

import numpy as np


def reference_state(spec, params: np.ndarray) -> np.ndarray:
    """Dense numpy evaluation of the cut formula (used by the CPU tests to validate make_cut)."""
    nl, nr = spec.n_left, spec.n - spec.n_left
    K = spec.bond_dim
    radices = [len(b.terms) for b in spec.bonds]
    psi = np.zeros((2**nl, 2**nr), dtype=np.complex128)

    def run(gl, n, pvec):
        st = np.zeros(2**n, dtype=np.complex128)
        st[0] = 1
        for g in gl:
            m = g.matrix(pvec)
            k = len(g.qubits)
            t = st.reshape([2] * n)
            t = np.moveaxis(t, list(g.qubits), range(k))
            shp = t.shape
            t = (m.reshape(2**k, 2**k) @ t.reshape(2**k, -1)).reshape(shp)
            st = np.ascontiguousarray(np.moveaxis(t, range(k), list(g.qubits))).reshape(-1)
        return st

    for b in range(K):
        digits, x = [], b
        for r in reversed(radices):
            digits.append(x % r)
            x //= r
        digits = digits[::-1]
        w = 1.0 + 0j
        for bond, dgt in zip(spec.bonds, digits):
            kind, ref = bond.terms[dgt][2]
            if kind == "const":
                w *= ref
            else:
                a = ref.scale * params[ref.index] + ref.offset
                w *= np.cos(a) if kind == "cos" else np.sin(a)
        pvec = np.concatenate([np.asarray(params, dtype=np.float64), np.array(digits, dtype=np.float64)])
        psi += w * np.outer(run(spec.left, nl, pvec), run(spec.right, nr, pvec))
    if getattr(spec, "right_rot", False):
        # the right half's qubits are labelled rotated by one (global qubit n_left is its LAST local qubit): back to the
        # natural column order, then the deferred gate and its one-qubit tail as an ordinary two-qubit gate
        t = psi.reshape([2**nl] + [2] * nr)
        psi = np.moveaxis(t, nr, 1).reshape(2**nl, 2**nr)
    if getattr(spec, "epilogue", None) is not None:
        e = spec.epilogue
        x = np.eye(4, dtype=np.complex128)
        for c0, c1, c2, ref in e.factors:
            m = np.array(c0, dtype=np.complex128)
            if ref is not None:
                a = ref.scale * params[ref.index] + ref.offset
                m = m + np.cos(a) * c1 + np.sin(a) * c2
            x = m @ x
        t = psi.reshape([2] * spec.n)
        t = np.moveaxis(t, [e.ql, e.qr], [0, 1])
        shp = t.shape
        t = (x @ t.reshape(4, -1)).reshape(shp)
        psi = np.ascontiguousarray(np.moveaxis(t, [0, 1], [e.ql, e.qr]))
    return psi.reshape(-1)
