"""Cut contraction: a wavefunction as ONE complex GEMM of two half-circuit batches.

For a circuit whose entangling gates across a bipartition L | R are few, the contraction order that
the reference's greedy path finder picks on the CPU (opt_einsum greedy, ``tensorcircuit/cons.py:1246-1258``:
total size ~ 2 x 2^n for config 2) is: contract each half of the network on its own, then join the
halves over the "bond" indices carried by the crossing gates.  On MI355X that order maps onto

    psi[x_L, x_R] = sum_b  w_b * L_b[x_L] * R_b[x_R]           (b = one term per crossing gate)

* every crossing two-qubit gate is split by an operator-Schmidt decomposition G = sum_k A_k (x) B_k
  (exp1(ZZ) / rzz / cnot / cz: 2 terms; a generic gate: up to 4);
* ``L_b`` / ``R_b`` are ordinary half-circuits in which crossing gate k is replaced by the one-qubit
  operator ``A_{k,b_k}`` / ``B_{k,b_k}``: they run as a *batch* of K = prod r_k states through the
  tile-VM (selector gates, ``plan.BK_SELECT``);
* the join is a dense M x K . K x N complex GEMM (M = 2^|L|, N = 2^|R|) on the f32 MFMA pipe
  (``tcmi_cgemm``) that writes psi once: algorithmic bytes ~ 2^n * 8, flops 8 * 2^n * K.

The result is the same tensor (a different contraction order changes rounding only).  The executor
chooses between this and the state-vector plan with a cost model (``executor.get_compiled``).
"""

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import plan as P


def schmidt_terms(m4: np.ndarray, tol: float = 1e-13) -> List[Tuple[np.ndarray, np.ndarray]]:
    """4x4 matrix on (qa, qb) -> [(A_k, B_k)] with M = sum_k kron(A_k, B_k)."""
    m = np.asarray(m4, dtype=np.complex128).reshape(2, 2, 2, 2)      # [a_out, b_out, a_in, b_in]
    r = m.transpose(0, 2, 1, 3).reshape(4, 4)                         # [(a_out,a_in), (b_out,b_in)]
    u, s, vh = np.linalg.svd(r)
    out = []
    for k in range(4):
        if s[k] > tol * max(1.0, s[0]):
            out.append((np.sqrt(s[k]) * u[:, k].reshape(2, 2), np.sqrt(s[k]) * vh[k, :].reshape(2, 2)))
    return out


@dataclass
class Bond:
    """One crossing gate: ``terms[j] = (A_j, B_j, coef)`` with coef = ("const", c) | ("cos", ParamRef)
    | ("sin", ParamRef); the gate equals sum_j coef_j * kron(A_j, B_j) (A on the left half)."""

    terms: List[Tuple[np.ndarray, np.ndarray, tuple]]


@dataclass
class CutSpec:
    n: int
    n_left: int
    left: List[P.GateRec]      # qubits 0..n_left-1; selector parameters appended after the circuit's
    right: List[P.GateRec]     # qubits 0..n-n_left-1
    bonds: List[Bond]
    nparams: int               # circuit parameters; selector column of bond k = nparams + k

    @property
    def bond_dim(self) -> int:
        r = 1
        for b in self.bonds:
            r *= len(b.terms)
        return r


def make_cut(gates: List[P.GateRec], n: int, n_left: int, nparams: int, max_bond: int = 1 << 12) -> Optional[CutSpec]:
    """Split the gate list at qubit ``n_left``; None if a gate cannot be split (3-qubit crossing,
    parametrised crossing gate that is not of the exp1 form) or the bond exceeds ``max_bond``."""
    left, right, bonds = [], [], []
    bond = 1
    for g in gates:
        side = [q < n_left for q in g.qubits]
        if all(side):
            left.append(g)
            continue
        if not any(side):
            diag = None
            if g.diag is not None:
                diag = [P.DiagTerm(tuple(q - n_left for q in t.qubits), t.const, t.param) for t in g.diag]
            right.append(P.GateRec(tuple(q - n_left for q in g.qubits), g.c0, g.c1, g.c2, g.param, diag, g.name, g.select))
            continue
        if len(g.qubits) != 2 or g.select is not None:
            return None
        qa, qb = g.qubits
        flip = not side[0]  # first listed qubit is on the right: transpose the roles

        def ordered(m):
            m = np.asarray(m, dtype=np.complex128).reshape(4, 4)
            return m.reshape(2, 2, 2, 2).transpose(1, 0, 3, 2).reshape(4, 4) if flip else m

        terms = []
        if g.param is None:
            for a, b in schmidt_terms(ordered(g.c0)):
                terms.append((a, b, ("const", 1.0)))
        else:
            for mat, coef in ((g.c0, ("const", 1.0)), (g.c1, ("cos", g.param)), (g.c2, ("sin", g.param))):
                if np.abs(np.asarray(mat)).max() < 1e-14:
                    continue
                for a, b in schmidt_terms(ordered(mat)):
                    terms.append((a, b, coef))
        if not terms or len(terms) > 4:
            return None
        ql, qr = (qb, qa) if flip else (qa, qb)
        if len(terms) == 1:
            a, b, coef = terms[0]
            if coef[0] != "const":
                return None
            left.append(P.GateRec((ql,), c0=a, name=g.name + "-L", diag=P.diag_terms_const(a, (ql,))))
            right.append(P.GateRec((qr - n_left,), c0=b, name=g.name + "-R", diag=P.diag_terms_const(b, (qr - n_left,))))
            continue
        k = len(bonds)
        sel = P.ParamRef(nparams + k, 1.0, 0.0)
        left.append(P.GateRec((ql,), param=sel, select=[t[0] for t in terms], name=g.name + "-L"))
        right.append(P.GateRec((qr - n_left,), param=sel, select=[t[1] for t in terms], name=g.name + "-R"))
        bonds.append(Bond(terms))
        bond *= len(terms)
        if bond > max_bond:
            return None
    return CutSpec(n, n_left, left, right, bonds, nparams)
