"""TEST INFRASTRUCTURE (oracle): dense numpy evaluation of the cut-contraction formula
psi[x_L, x_R] = sum_b w_b L_b[x_L] R_b[x_R] for a ``tcmi.cut.CutSpec`` -- what the reference's greedy path computes on a
shallow ladder circuit (tensorcircuit/cons.py:298-374: contract each half of the network, then join).  Used by the CPU
tests to validate ``tcmi.cut.make_cut`` against ``oracle.dense``; nothing in the product imports it."""

import numpy as np


def reference_state(spec, params: np.ndarray, program: bool = False) -> np.ndarray:
    """Dense numpy evaluation of the cut formula (used by the CPU tests to validate make_cut).  ``program``: the tail of a
    cut with two deferred crossing gates is applied the way the join kernel does it -- its ops (``TailProgram.tables``)
    on the index bits (u, r1 | v, l4) of the product, whose column index is still rotated -- instead of as ordinary gates
    on the joined state (tcmi_cgemm_split_prog's contract, csrc/tcmi_gemm_split.hip)."""
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
    if program:
        prog = spec.epilogue.program
        tabs = prog.tables(np.asarray(params, dtype=np.float64))
        M, N = psi.shape
        # product element (m, c'): u = m & 1, r1 = (m >> 1) & 1, v = c' & 1, l4 = (c' >> 1) & 1
        t = psi.reshape(M // 4, 2, 2, N // 4, 2, 2)           # [m_hi, r1, u, c_hi, l4, v]
        axis = {0: 2, 1: 1, 2: 5, 3: 4}                        # bit id -> axis
        for k, (kind, bit, _) in enumerate(prog.ops):
            if kind == "diag":
                d = tabs[k].reshape(2, 2, 2, 2)                # index u + 2 r1 + 4 v + 8 l4 -> [l4, v, r1, u]
                t = t * d.transpose(2, 3, 0, 1)[None, :, :, None, :, :]       # -> [r1, u, l4, v]
            else:
                m2 = tabs[k][:4].reshape(2, 2)
                t = np.moveaxis(np.tensordot(m2, t, axes=([1], [axis[bit]])), 0, axis[bit])
        psi = t.reshape(M, N)
        # column c' of the product is column (c' >> rot) | (block * N / 2^rot) of the state, block = v + 2 l4 or -- when v is
        # the first right-hand qubit (program.vhigh) -- 2 v + l4
        cp = np.arange(N)
        low = cp & ((1 << rot) - 1)
        if getattr(prog, "vhigh", 0):
            low = 2 * (low & 1) + (low >> 1)
        nat = (cp >> rot) | (low * (N >> rot))
        out = np.empty_like(psi)
        out[:, nat] = psi
        return out.reshape(-1)
    if rot:
        # the right half's qubits are labelled rotated by `rot` (global qubits n_left .. n_left + rot - 1 are its LAST local
        # qubits): back to the natural column order, then the deferred gates as ordinary gates on the joined state
        t = psi.reshape([2**nl] + [2] * nr)
        src = list(range(nr - rot + 1, nr + 1))
        if getattr(getattr(spec.epilogue, "program", None), "vhigh", 0):
            src = src[::-1]          # (the two rotated qubits also changed places)
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
