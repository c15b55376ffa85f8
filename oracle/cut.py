"""TEST INFRASTRUCTURE (oracle): dense numpy evaluation of the cut-contraction formula
psi[x_L, x_R] = sum_b w_b L_b[x_L] R_b[x_R] for a ``tcmi.cut.CutSpec`` -- what the reference's greedy path computes on a
shallow ladder circuit (tensorcircuit/cons.py:298-374: contract each half of the network, then join).  Used by the CPU
tests to validate ``tcmi.cut.make_cut`` against ``oracle.dense``; nothing in the product imports it."""

import numpy as np


def reference_state(spec, params: np.ndarray) -> np.ndarray:
    """Dense numpy evaluation of the cut formula (used by the CPU tests to validate make_cut); the gates of a deferred tail
    are applied as ordinary gates on the joined state."""
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
    rot = int(getattr(spec, "right_rot", 0))
    if rot:
        # the right half's qubits are labelled rotated by `rot` (global qubits n_left .. n_left + rot - 1 are its LAST local
        # qubits): back to the natural column order, then the deferred gates as ordinary gates on the joined state
        t = psi.reshape([2**nl] + [2] * nr)
        src = list(range(nr - rot + 1, nr + 1))
        psi = np.moveaxis(t, src, list(range(1, rot + 1))).reshape(2**nl, 2**nr)
    if getattr(spec, "epilogue", None) is not None:
        st = psi.reshape(-1)
        for g in spec.epilogue.tail:
            m = g.matrix(np.asarray(params, dtype=np.float64))
            k = len(g.qubits)
            t = st.reshape([2] * spec.n)
            t = np.moveaxis(t, list(g.qubits), range(k))
            shp = t.shape
            t = (m.reshape(2**k, 2**k) @ t.reshape(2**k, -1)).reshape(shp)
            st = np.ascontiguousarray(np.moveaxis(t, range(k), list(g.qubits))).reshape(-1)
        psi = st
    return psi.reshape(-1)
