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

**Deferred last crossing gate** (``make_cut(defer=True)``).  The LAST crossing gate G on (q_l, q_r) need not be a bond:
if every later gate that touches q_l or q_r either commutes with G (both diagonal) or is a one-qubit gate on q_l / q_r
with nothing but one-qubit gates on that qubit after it, then

    U = X . (everything else),      X = (one-qubit tail on q_l, q_r) . G     a 4 x 4 matrix per circuit,

and X is applied to the join's result where it sits in the GEMM's accumulators (``tcmi_cgemm_split_epi``: the four
amplitudes a 4 x 4 on the lowest row bit and one column bit mixes are held by ONE thread of the MFMA result layout).  The
bond dimension halves (ZZ / CNOT / CZ crossings) and with it the GEMM and the suffix batches: the HEA-B ladder of
config 2 (exp1(ZZ) ladder, then rx on every qubit) qualifies, K = 256 -> 128.  For the thread-local epilogue q_l must be
the left half's lowest index bit (q_l = n_left - 1) and q_r is made the right half's lowest bit by labelling the right
half's qubits rotated by one (``CutSpec.right_rot``; the kernel un-rotates the column index when it stores).

Two deferred crossing gates (the tail as a small gate program run by the join kernel on four index bits; K = 256 -> 64
on config 2) were built and measured in round 5 and lost to one deferred gate (1.73e11 against 1.81e11 amplitudes/s: the
program's vector work costs what the quarter bond saves); the variant was removed in round 6.
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
class Epilogue:
    """The deferred last crossing gate and the one-qubit tail absorbed into it: ``X(theta) = prod_g (c0 + cos(a_g) c1 +
    sin(a_g) c2)`` (time order = list order, later factors on the left), every factor a 4 x 4 over (q_l, q_r), q_l the
    more significant index bit.  ``factors[g] = (c0, c1, c2, ParamRef | None)``."""

    ql: int
    qr: int
    factors: List[Tuple[np.ndarray, np.ndarray, np.ndarray, Optional[P.ParamRef]]]
    tail: Optional[List[P.GateRec]] = None       # the deferred gates themselves (global qubits, time order)

    def matrix(self, params) -> np.ndarray:
        x = np.eye(4, dtype=np.complex128)
        for c0, c1, c2, ref in self.factors:
            m = np.array(c0, dtype=np.complex128)
            if ref is not None:
                a = ref.scale * float(params[ref.index]) + ref.offset
                m = m + np.cos(a) * c1 + np.sin(a) * c2
            x = m @ x
        return x


@dataclass
class CutSpec:
    n: int
    n_left: int
    left: List[P.GateRec]      # qubits 0..n_left-1; selector parameters appended after the circuit's
    right: List[P.GateRec]     # qubits 0..n-n_left-1
    bonds: List[Bond]
    nparams: int               # circuit parameters; selector column of bond k = nparams + k
    epilogue: Optional[Epilogue] = None
    right_rot: int = 0         # right half labelled rotated by this many: global qubit q is local (q - n_left - rot) mod n_right

    @property
    def bond_dim(self) -> int:
        r = 1
        for b in self.bonds:
            r *= len(b.terms)
        return r


def gate_norm_bound(g: P.GateRec) -> float:
    """An upper bound of the spectral norm of a gate over all parameter values: exactly 1.0 for unitary gates (unit-modulus
    diagonals; constant matrices and ``c0 + cos c1 + sin c2`` families that are unitary at five angles -- G(a)^+ G(a) is a
    trigonometric polynomial of degree 2: five points fix it), the largest norm among the alternatives of a selector gate
    (the operator-Schmidt factors of a crossing gate), ``|c0| + |c1| + |c2|`` otherwise."""
    def norm2(m):
        b = float(np.linalg.norm(np.asarray(m, dtype=np.complex128), 2))
        return 1.0 if abs(b - 1.0) < 1e-9 else b

    def unitary(m):
        m = np.asarray(m, dtype=np.complex128)
        return bool(np.abs(m.conj().T @ m - np.eye(m.shape[0])).max() < 1e-9)

    if g.select is not None:
        return max(norm2(m) for m in g.select)
    if g.diag is not None and not any(t.scale for t in g.diag):
        return 1.0
    dim = 1 << len(g.qubits)
    z = np.zeros((dim, dim), dtype=np.complex128)
    c0 = z if g.c0 is None else np.asarray(g.c0, dtype=np.complex128).reshape(dim, dim)
    if g.param is None:
        return 1.0 if unitary(c0) else norm2(c0)
    c1 = z if g.c1 is None else np.asarray(g.c1, dtype=np.complex128).reshape(dim, dim)
    c2 = z if g.c2 is None else np.asarray(g.c2, dtype=np.complex128).reshape(dim, dim)
    if all(unitary(c0 + np.cos(a) * c1 + np.sin(a) * c2) for a in 2 * np.pi * np.arange(5) / 5):
        return 1.0
    return norm2(c0) + norm2(c1) + norm2(c2)


def half_bounds(spec: "CutSpec") -> Tuple[float, float]:
    """(bound of |L_b[x]|, bound of |w_b R_b[x]|) for the two half-circuit batches of a cut started from a basis state: the
    product of the gates' norm bounds (a state's entries are bounded by its 2-norm), the right one times the largest
    bond weight.  What ``tcmi_cgemm_split_f16`` needs to choose its operand scales; ``inf`` when a factor is unbounded."""
    def prod(gs):
        b = 1.0
        for g in gs:
            b *= gate_norm_bound(g)
        return b

    w = 1.0
    for bond in spec.bonds:
        w *= max((abs(complex(ref)) if kind == "const" else 1.0) for _, _, (kind, ref) in bond.terms)
    return prod(spec.left), prod(spec.right) * w


def _lift(m, which):
    """2 x 2 on q_l (which = 0) / q_r (which = 1) -> 4 x 4 over (q_l, q_r)."""
    m = np.asarray(m, dtype=np.complex128).reshape(2, 2)
    return np.kron(m, np.eye(2)) if which == 0 else np.kron(np.eye(2), m)


def find_deferred(gates: List[P.GateRec], n_left: int):
    """(index of the last crossing gate, indices of the one-qubit gates absorbed with it) when that gate can be applied
    after the join (module docstring), else None."""
    cross = [i for i, g in enumerate(gates) if any(q < n_left for q in g.qubits) and any(q >= n_left for q in g.qubits)]
    if not cross:
        return None
    p = cross[-1]
    g = gates[p]
    if len(g.qubits) != 2 or g.select is not None:
        return None
    ql, qr = sorted(g.qubits)
    if ql != n_left - 1 or qr != n_left:
        return None
    dirty = {ql: not g.is_diag, qr: not g.is_diag}     # a non-diagonal factor of X sits on this qubit
    absorbed = []
    for i in range(p + 1, len(gates)):
        h = gates[i]
        hit = [q for q in h.qubits if q in dirty]
        if not hit:
            continue
        if h.select is not None:
            return None
        if len(h.qubits) == 1:
            absorbed.append(i)
            dirty[hit[0]] = dirty[hit[0]] or not h.is_diag
            continue
        # a wider gate stays in its half: it has to commute with every factor of X on the qubits they share
        if not h.is_diag or any(dirty[q] for q in hit):
            return None
    return p, absorbed


def _commute(h: P.GateRec, t: P.GateRec) -> bool:
    return not (set(h.qubits) & set(t.qubits)) or (h.is_diag and t.is_diag)


def make_cut(gates: List[P.GateRec], n: int, n_left: int, nparams: int, max_bond: int = 1 << 12,
             defer: int = 0) -> Optional[CutSpec]:
    """Split the gate list at qubit ``n_left``; None if a gate cannot be split (3-qubit crossing,
    parametrised crossing gate that is not of the exp1 form) or the bond exceeds ``max_bond``.  ``defer``: apply the
    last crossing gate after the join when the circuit allows it (``find_deferred``)."""
    left, right, bonds = [], [], []
    bond = 1
    epi, skip = None, set()
    n_right = n - n_left
    rot = 0
    if defer and epi is None:
        found = find_deferred(gates, n_left)
        if found is not None:
            p, absorbed = found
            ql, qr = n_left - 1, n_left
            factors = []
            for i in [p] + absorbed:
                g = gates[i]
                if len(g.qubits) == 2:
                    flip = g.qubits[0] != ql
                    lift = (lambda m, f=flip: None if m is None else
                            (np.asarray(m, dtype=np.complex128).reshape(2, 2, 2, 2).transpose(1, 0, 3, 2).reshape(4, 4) if f
                             else np.asarray(m, dtype=np.complex128).reshape(4, 4)))
                else:
                    lift = lambda m, w=int(g.qubits[0] == qr): None if m is None else _lift(m, w)
                z = np.zeros((4, 4), dtype=np.complex128)
                c0 = lift(g.c0)
                if g.param is None:
                    factors.append((c0, z, z, None))
                else:
                    factors.append((c0 if c0 is not None else z, lift(g.c1) if g.c1 is not None else z,
                                    lift(g.c2) if g.c2 is not None else z, g.param))
            epi = Epilogue(ql, qr, factors, [gates[i] for i in [p] + absorbed])
            skip, rot = set([p] + absorbed), 1

    def rloc(q):       # local index of global qubit q in the right half
        return (q - n_left - rot) % n_right

    for gi, g in enumerate(gates):
        if gi in skip:
            continue
        side = [q < n_left for q in g.qubits]
        if all(side):
            left.append(g)
            continue
        if not any(side):
            diag = None
            if g.diag is not None:
                diag = [P.DiagTerm(tuple(rloc(q) for q in t.qubits), t.const, t.param) for t in g.diag]
            right.append(P.GateRec(tuple(rloc(q) for q in g.qubits), g.c0, g.c1, g.c2, g.param, diag, g.name, g.select))
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
        ql_, qr_ = (qb, qa) if flip else (qa, qb)
        if len(terms) == 1:
            a, b, coef = terms[0]
            if coef[0] != "const":
                return None
            left.append(P.GateRec((ql_,), c0=a, name=g.name + "-L", diag=P.diag_terms_const(a, (ql_,))))
            right.append(P.GateRec((rloc(qr_),), c0=b, name=g.name + "-R", diag=P.diag_terms_const(b, (rloc(qr_),))))
            continue
        k = len(bonds)
        sel = P.ParamRef(nparams + k, 1.0, 0.0)
        left.append(P.GateRec((ql_,), param=sel, select=[t[0] for t in terms], name=g.name + "-L"))
        right.append(P.GateRec((rloc(qr_),), param=sel, select=[t[1] for t in terms], name=g.name + "-R"))
        bonds.append(Bond(terms))
        bond *= len(terms)
        if bond > max_bond:
            return None
    return CutSpec(n, n_left, left, right, bonds, nparams, epi, rot)
