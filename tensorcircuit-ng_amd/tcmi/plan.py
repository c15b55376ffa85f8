"""Plan compiler: gate list -> tile-VM pass programs for the HIP executor.

This replaces, for the state-vector hot path, what the reference does on every call in
Python: ``tn.copy`` of the network, ``_merge_single_gates``, the opt_einsum greedy search and
the ``contract_between`` loop (reference ``tensorcircuit/cons.py:298-374, 773-800, 845-961``;
``tensorcircuit/circuit.py:701-721``).  The result of those pairwise contractions for a circuit
acting on |0..0> is the state vector, so the plan executes it as a *state-vector-order* chain
(reference ``plain_contractor`` semantics, ``cons.py:429-463``) with two MI355X-specific
transformations:

* cache blocking: one HBM round trip (a "pass") loads a tile of 2^T amplitudes per workgroup
  (T tile bits = the low ``lowbits`` physical bits for coalescing + arbitrary others) and
  applies *every* gate that is executable on those qubits before writing the tile back;
* register blocking: inside a pass the tile lives in registers (2^R amplitudes per thread);
  gates act on "register bits" only, and an LDS exchange re-maps which tile bits are register
  bits between "rounds".  Diagonal gates (rz / phase / cz / exp1(ZZ) ...) never need their qubits
  local: they are phase polynomials evaluated from the global index.

Index conventions: qubit q <-> physical bit p = n-1-q of the flat state index (qubit 0 is the
most significant bit, reference ``tests/test_circuit.py:47-53``).

The descriptor layout written by :func:`encode_pass` is the C-ABI contract of
``tcmi_run_pass`` (``include/tcmi.h``); ``oracle/plan_emulator.py`` re-implements it in numpy for
the CPU test-suite.
"""

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

# ---- descriptor constants (mirrored in csrc/tcmi_vm.h) ---------------------------------
MAGIC = 0x54434D31  # "TCM1"
HDR_WORDS = 24
RR_WORDS = 50
R_MAX = 6
LT_MAX = 10
T_MAX = 16
OP_G1 = 1
OP_G2 = 2
OP_DIAG = 3
OP_G1M = 4
OP_EXPECT = 5
OP_DIAGB = 7   # one diagonal term on a register bit x thread bits: multiply by exp(+-i phi), sign per thread
OP_DIAGC = 6   # diagonal terms on register bits only: multiply by a 2^R table of phase factors
OP_DIAGB2 = 9  # two register-x-thread terms on the same register bit: {9, j, mask1, mask2, slot}, table of 4 factors (gen 2)
OP_DIAGCW = 10  # DIAGC whose table is picked per wave: {10, slot, nsel, m0, m1, m2}, variant = sum_k parity(wave index & m_k) << k, table of 2^nsel * 2^R factors (gen 2): register-x-thread terms whose thread bits are wave-uniform cost no multiply of their own
OP_DIAGF = 8   # backward (adjoint sweep) flush of diagonal terms in table form: see encode_pass
OP_XFOLD = 12   # reverse sweep, plan-specialised kernels only: {12, j | kind << 8, cslot, gslot}: lambda += c P psi with P = X (kind 0) or Y (kind 1) on register bit j (c = ctab[cslot], real) and gradient slot gslot += c/2 <psi|P|psi> of the tile: the term c/2 P_q of a Pauli-sum cotangent born in registers (compile_adjoint_plan ``fold``)
OP_DFOLD = 13   # reverse sweep, plan-specialised kernels only: {13, nterms, gslot, (thread-side Z mask over physical bits, register mask, cslot) * nterms}: lambda[r] += D psi[r] with D = sum_t c_t (-1)^{parity(index & zmask_t)} (the Z-only strings of a Pauli-sum cotangent, register part of every mask split off for the round's layout) and gradient slot gslot += 1/2 sum_r D |psi[r]|^2
OP_XFOLD2 = 14  # reverse sweep, plan-specialised kernels only: {14, ja | jb << 8 | ka << 16 | kb << 17, cslot, gslot}: lambda += c P_a P_b psi with P = X (k = 0) or Y (k = 1) on the register bits ja, jb, and gradient slot gslot += c/2 <psi|P_a P_b|psi> of the tile: a two-factor string (XX, YY, XY: Heisenberg-type couplings) of a Pauli-sum cotangent born in registers (compile_adjoint_plan ``fold2``)
OP_EXPECT2 = 11  # measurement, Z-only strings grouped by register mask (gen 2): see encode_measure_pass
FLAG_NOSTORE = 1
FLAG_LAMBDA_ZERO = 2   # reverse sweep: lambda is not loaded, it starts as zero and is BORN in this pass (OP_XFOLD / OP_DFOLD)
DIAG_CHUNK = 8
MAX_DIAGB = 5  # more register-x-thread terms than this in one DIAG op: the per-thread sincos path is cheaper
# G2 kinds: 0 general, 1 = CNOT(control ja, target jb), 2 = CNOT(control jb, target ja), 3 = SWAP
CONST_FLAG = 1 << 30

# builder kinds
BK_TRIG = 1  # M = C0 + cos(k*theta + off) C1 + sin(k*theta + off) C2
BK_COEF = 2  # phase coefficient in turns = k*theta + off
BK_UDAG = 3  # U^dagger (adjoint sweep)
BK_KMAT = 4  # K = (dU/dtheta) U^dagger (adjoint sweep)
BK_PHASE = 6  # one entry exp(2 pi i sum_t +-(k_t theta_t + o_t)) of a DIAGC table
BK_SELECT = 5  # M = table[round(param)]: one of several constant matrices (cut contraction bonds)


@dataclass
class ParamRef:
    """Angle = scale * params[index] + offset (params = flat per-batch parameter vector)."""

    index: int
    scale: float = 1.0
    offset: float = 0.0


@dataclass
class DiagTerm:
    """phase += coef * (-1)^{parity(bits of the listed qubits)} ; coef in radians.
    coef = const + (scale * params[index]) if param is not None."""

    qubits: Tuple[int, ...]
    const: float = 0.0
    param: Optional[ParamRef] = None
    # scale=True: not a phase but the real factor diag(c, 1/c) on the one listed qubit, c = cos(param.scale * theta +
    # param.offset): what a rotation applied as two shears leaves behind (encode_pass, two-shear form); planner-internal
    scale: bool = False
    gate: int = -1   # scale terms: index of the rotation they belong to


@dataclass
class GateRec:
    """One gate of the recorded circuit.

    Dense form: ``M = c0 + cos(a) c1 + sin(a) c2`` with ``a = param.scale*theta + param.offset``
    (``param is None`` -> constant gate ``c0``).  Matrices are ``U[out, in]`` over ``qubits`` in
    the listed order.  ``diag`` (list of DiagTerm) is set iff the gate is a unit-modulus diagonal.
    """

    qubits: Tuple[int, ...]
    c0: Optional[np.ndarray] = None
    c1: Optional[np.ndarray] = None
    c2: Optional[np.ndarray] = None
    param: Optional[ParamRef] = None
    diag: Optional[List[DiagTerm]] = None
    name: str = ""
    select: Optional[List[np.ndarray]] = None  # M = select[round(params[param.index])] (constant alternatives)

    @property
    def is_diag(self):
        return self.diag is not None

    def matrix(self, params=None):
        if self.select is not None:
            return np.array(self.select[int(round(float(params[self.param.index])))], dtype=np.complex128)
        m = np.array(self.c0, dtype=np.complex128)
        if self.param is not None:
            a = self.param.scale * float(params[self.param.index]) + self.param.offset
            m = m + np.cos(a) * self.c1 + np.sin(a) * self.c2
        return m


# ---- diagonal analysis -----------------------------------------------------------------
def walsh_terms(phases: np.ndarray, qubits: Sequence[int]):
    """phases[x] (x = bits of ``qubits`` in listed order, first = MSB) -> list of
    (qubit subset, coefficient) with phase(x) = sum coef * (-1)^{parity(x & subset)}: the Walsh-Hadamard transform of
    the phases (butterflies over the k index bits, O(k 2^k))."""
    k = len(qubits)
    c = np.asarray(phases, dtype=np.float64).reshape([2] * k) if k else np.asarray(phases, dtype=np.float64)
    for ax in range(k):
        lo, hi = np.take(c, 0, axis=ax), np.take(c, 1, axis=ax)
        c = np.stack([lo + hi, lo - hi], axis=ax)
    c = np.asarray(c).reshape(-1) / 2**k
    out = []
    for s in range(2**k):
        if abs(c[s]) > 1e-15:
            sub = tuple(qubits[i] for i in range(k) if (s >> (k - 1 - i)) & 1)
            out.append((sub, float(c[s])))
    return out


def diag_terms_vector(d: np.ndarray, qubits: Sequence[int], tol=1e-12):
    """Unit-modulus diagonal given by its 2^k entries (index bits = ``qubits`` in listed order, first = MSB) -> DiagTerm
    list, else None."""
    d = np.asarray(d, dtype=np.complex128).reshape(-1)
    if d.size != 2 ** len(qubits) or np.abs(np.abs(d) - 1).max() > tol:
        return None
    return [DiagTerm(sub, const=c) for sub, c in walsh_terms(np.angle(d), qubits)]


def diag_terms_const(matrix: np.ndarray, qubits: Sequence[int], tol=1e-12):
    """Unit-modulus diagonal constant matrix -> DiagTerm list, else None (memoised by matrix bytes)."""
    a = np.ascontiguousarray(matrix, dtype=np.complex128)
    key = a.tobytes()
    tpl = _CONST_TEMPLATES.get(key)
    if tpl is None:
        res = _diag_terms_const(a, tuple(range(len(qubits))), tol)
        tpl = False if res is None else [(t.qubits, t.const) for t in res]
        _CONST_TEMPLATES[key] = tpl
    if tpl is False:
        return None
    return [DiagTerm(tuple(qubits[i] for i in sub), const=c) for sub, c in tpl]


def _diag_terms_const(matrix: np.ndarray, qubits: Sequence[int], tol=1e-12):
    m = np.asarray(matrix, dtype=np.complex128)
    d = np.diag(m)
    if np.abs(m - np.diag(d)).max() > tol or np.abs(np.abs(d) - 1).max() > tol:
        return None
    return [DiagTerm(sub, const=c) for sub, c in walsh_terms(np.angle(d), qubits)]


_TRIG_TEMPLATES: Dict[Any, Any] = {}
_CONST_TEMPLATES: Dict[Any, Any] = {}


def diag_terms_trig(c0, c1, c2, qubits, param: ParamRef, tol=1e-12):
    """Memoised front end of ``_diag_terms_trig``: the Walsh analysis depends only on the three matrices;
    the cached template (terms on local positions, unit scale / zero offset) is remapped to ``qubits`` and
    ``param`` — recording a circuit must not redo it per gate (host time of small-circuit VQE loops)."""
    a0, a1, a2 = (np.ascontiguousarray(x, dtype=np.complex128) for x in (c0, c1, c2))
    key = (a0.tobytes(), a1.tobytes(), a2.tobytes())
    tpl = _TRIG_TEMPLATES.get(key)
    if tpl is None:
        k = len(qubits)
        res = _diag_terms_trig(a0, a1, a2, tuple(range(k)), ParamRef(0, 1.0, 0.0), tol)
        # with unit scale / zero offset: const = cg, param.scale = cf
        tpl = False if res is None else [(t.qubits, t.const, None if t.param is None else t.param.scale) for t in res]
        _TRIG_TEMPLATES[key] = tpl
    if tpl is False:
        return None
    out = []
    for sub, cg, cf in tpl:
        qs = tuple(qubits[i] for i in sub)
        if cf is None:
            out.append(DiagTerm(qs, const=cg))
        else:
            out.append(DiagTerm(qs, const=cg + cf * param.offset, param=ParamRef(param.index, cf * param.scale, 0.0)))
    return out


def _diag_terms_trig(c0, c1, c2, qubits, param: ParamRef, tol=1e-12):
    """``c0 + cos(a) c1 + sin(a) c2`` is a unit-modulus diagonal for every a iff it can be written
    diag(exp(i(f_x a + g_x))).  Recognised forms: c0 = 0, c1 = diag(e^{i g}), c2 = diag(i f e^{i g})
    with f = +-1 (covers rz / exp1(diagonal Pauli product) / rzz), and c0 = diag(p), c1 = diag(q),
    c2 = i c1 with disjoint supports (phase gate)."""
    c0, c1, c2 = (np.asarray(x, dtype=np.complex128) for x in (c0, c1, c2))
    for m in (c0, c1, c2):
        if np.abs(m - np.diag(np.diag(m))).max() > tol:
            return None
    d0, d1, d2 = np.diag(c0), np.diag(c1), np.diag(c2)
    dim = d0.size
    f = np.zeros(dim)
    g = np.zeros(dim)
    for x in range(dim):
        if abs(d0[x]) < tol and abs(abs(d1[x]) - 1) < tol:
            ratio = d2[x] / d1[x]  # must be +-i
            if abs(ratio - 1j) < tol:
                f[x] = 1.0
            elif abs(ratio + 1j) < tol:
                f[x] = -1.0
            else:
                return None
            g[x] = np.angle(d1[x])
        elif abs(abs(d0[x]) - 1) < tol and abs(d1[x]) < tol and abs(d2[x]) < tol:
            f[x] = 0.0
            g[x] = np.angle(d0[x])
        else:
            return None
    terms = {}
    for sub, c in walsh_terms(g, qubits):
        terms[sub] = [c, 0.0]
    for sub, c in walsh_terms(f, qubits):
        terms.setdefault(sub, [0.0, 0.0])[1] = c
    out = []
    for sub, (cg, cf) in terms.items():
        if cf != 0.0:
            out.append(
                DiagTerm(sub, const=cg + cf * param.offset,
                         param=ParamRef(param.index, cf * param.scale, 0.0))
            )
        else:
            out.append(DiagTerm(sub, const=cg))
    return out


# ---- scheduling ------------------------------------------------------------------------
def _scan(gates, order, qbits_of, res_of, capacity, forced, allowed):
    """One greedy scan over pending gates (program order).  Returns (chosen resource-bit set,
    chosen gate ids).  ``qbits_of(g)`` = the qubits' physical bits (ordering), ``res_of(g)`` = the
    resource bits the gate needs inside the chosen set (phys bits for the tile choice, tile bits
    for the register choice; empty for diagonal gates, which only respect ordering against the
    non-commuting dense gates)."""
    S = set(forced)
    blocked_all = set()
    blocked_dense = set()
    chosen = []
    for gi in order:
        g = gates[gi]
        qs = set(qbits_of(gi))
        if g.is_diag:
            if qs & blocked_all:
                blocked_dense |= qs
            else:
                chosen.append(gi)
            continue
        rs = set(res_of(gi))
        if qs & (blocked_all | blocked_dense) or not rs <= allowed or len(S | rs) > capacity:
            blocked_all |= qs
            continue
        S |= rs
        chosen.append(gi)
    return S, chosen


def _scan_fixed(gates, order, qmask, rmask, S, count_only=False, cap=None):
    """Gates of ``order`` (program order) executable with the resource bits ``S`` (bit mask): a dense gate runs if
    its resource bits are in S and no earlier gate on its qubits was left behind; diagonal gates need no resources.
    ``qmask[gi]`` = ordering bits of the gate, ``rmask[gi]`` = resource bits it needs (0 for diagonal gates).
    ``count_only``: return the weighted number of dense gates only (and stop once every bit of S is blocked).
    ``cap``: at most this (weighted) number of dense gates; the rest stays pending for later passes."""
    blocked_all = 0
    blocked_dense = 0
    chosen = []
    cnt = 0
    for gi in order:
        qs = qmask[gi]
        rs = rmask[gi]
        if gates[gi].is_diag:
            if qs & blocked_all:
                blocked_dense |= qs
            elif not count_only:
                chosen.append(gi)
            continue
        if (qs & (blocked_all | blocked_dense)) or (rs & ~S) or (cap is not None and cnt >= cap):
            blocked_all |= qs
            if count_only and (S & ~blocked_all) == 0:
                break
            continue
        cnt += 1 if (rs & (rs - 1)) == 0 else 3
        if not count_only:
            chosen.append(gi)
    return cnt if count_only else chosen


def _grow(gates, order, qmask, rmask, capacity, forced, allowed, window=600, tiebreak=0):
    """Choose up to ``capacity`` resource bits (bit mask, superset of ``forced``, subset of ``allowed``) greedily by
    marginal gain: the bit -- or the pair of bits of one pending two-qubit gate -- that makes the most additional
    dense gates executable per bit added.  Returns the mask (possibly with fewer than ``capacity`` bits when nothing
    more helps)."""
    S = forced
    win = order[:window]
    base = _scan_fixed(gates, win, qmask, rmask, S, True)
    while bin(S).count("1") < capacity:
        room = capacity - bin(S).count("1")
        cands = set()
        free = allowed & ~S
        b = 0
        f = free
        while f:
            if f & 1:
                cands.add(1 << b)
            f >>= 1
            b += 1
        if room >= 2:
            for gi in win:
                rs = rmask[gi]
                if rs and (rs & (rs - 1)) and not (rs & ~allowed) and bin(rs & ~S).count("1") == 2:
                    cands.add(rs & ~S)
        best, best_key = 0, None
        for c in cands:
            gain = _scan_fixed(gates, win, qmask, rmask, S | c, True) - base
            if gain <= 0:
                continue
            k = bin(c).count("1")
            key = (gain / k, -k, -c if tiebreak == 0 else c)   # equal gains: lowest / highest bits first
            if best_key is None or key > best_key:
                best, best_key = c, key
        if not best:
            break
        S |= best
        base = _scan_fixed(gates, win, qmask, rmask, S, True)
    return S


@dataclass
class Round:
    reg_tb: List[int]  # tile-bit index of each register bit
    thr_tb: List[int]  # tile-bit index of each thread bit
    gates: List[int]   # gate ids executed in this round (program order)
    fold: List[Tuple[int, float, int, int]] = field(default_factory=list)   # (register bit, coefficient, fold index, kind 0 = X / 1 = Y): OP_XFOLD ops at the start of the round
    dfold: List[Tuple[int, float]] = field(default_factory=list)       # (Z mask over physical bits, coefficient): one OP_DFOLD at the start of the round
    fold2: List[Tuple[int, int, float, int, int, int]] = field(default_factory=list)   # (register bit a, register bit b, coefficient, pair index, kind a, kind b): OP_XFOLD2 ops at the start of the round


@dataclass
class PassPlan:
    tile_bits: List[int]  # physical bit positions, ascending
    rounds: List[Round] = field(default_factory=list)
    gate_ids: List[int] = field(default_factory=list)


@dataclass
class PlanConfig:
    R: int = 5          # register bits (2^R amplitudes per thread)
    LT: int = 8         # log2(threads per workgroup)
    lowbits: int = 5    # physical low bits always in the tile (coalescing run = 2^lowbits amps)
    vec: int = 2        # amplitudes per 16-byte global access (2 for complex64, 1 for complex128)
    pass_cap: Optional[int] = None   # at most this many (weighted) dense gates per pass: a pass costs max(HBM round trip, arithmetic), so more gates than the round trip hides are better left to a later pass
    gen: int = 1        # kernel generation executing the plan: 2 = packed-f32 kernels (tcmi_vm2 / tcmi_adjoint2): extra ops
    tiebreak: int = 0   # tile growth, equal marginal gains: 0 = lowest physical bits first, 1 = highest first
    shear2: bool = True  # gen-2 plans: rotations in two-shear form where the real factor rides on a phase table (shear2_gates)

    @property
    def T(self):
        return self.R + self.LT


def schedule(gates, n: int, cfg: PlanConfig, independent: bool = False) -> List[PassPlan]:
    """``independent=True``: the items commute pairwise (measurement terms), nothing is ever blocked
    by ordering and no store layout is needed."""
    T, R = cfg.T, cfg.R
    if n < T:
        raise ValueError(f"n={n} smaller than tile bits T={T}")
    L = min(cfg.lowbits, T)
    for g in gates:
        if not g.is_diag and len(g.qubits) > min(R, 2):
            raise NotImplementedError(
                f"dense gates on more than {min(R, 2)} qubits are not supported by the HIP tile-VM"
            )
    phys = lambda gi: [n - 1 - q for q in gates[gi].qubits]
    phys_res = lambda gi: [] if gates[gi].is_diag else phys(gi)
    order_bits = (lambda gi: []) if independent else phys
    pending = list(range(len(gates)))
    passes = []
    all_bits = set(range(n))
    qmask = [0] * len(gates)
    rmask = [0] * len(gates)
    if not independent:
        for gi, g in enumerate(gates):
            for q in g.qubits:
                qmask[gi] |= 1 << (n - 1 - q)
            if not g.is_diag:
                rmask[gi] = qmask[gi]
    while pending:
        if independent:
            S, chosen = _scan(gates, pending, order_bits, phys_res, T, set(range(L)), all_bits)
        else:
            Sm = _grow(gates, pending, qmask, rmask, T, (1 << L) - 1, (1 << n) - 1, tiebreak=cfg.tiebreak)
            chosen = _scan_fixed(gates, pending, qmask, rmask, Sm, cap=cfg.pass_cap)
            S = {b for b in range(n) if (Sm >> b) & 1}
            if not any(not gates[gi].is_diag for gi in chosen) and any(not gates[gi].is_diag for gi in pending):
                # nothing dense became executable (should not happen): first-fit fallback
                S, chosen = _scan(gates, pending, order_bits, phys_res, T, set(range(L)), all_bits)
        if not chosen:
            raise RuntimeError("scheduler made no progress")
        # fill the tile with the lowest unused bits (better contiguity)
        b = 0
        while len(S) < T:
            if b not in S:
                S.add(b)
            b += 1
        tile_bits = sorted(S)
        pp = PassPlan(tile_bits=tile_bits, gate_ids=list(chosen))
        _schedule_rounds(gates, n, cfg, pp, independent)
        passes.append(pp)
        cs = set(chosen)
        pending = [gi for gi in pending if gi not in cs]
    return passes


def _schedule_rounds(gates, n, cfg: PlanConfig, pp: PassPlan, independent: bool = False):
    T, R = cfg.T, cfg.R
    tb_of_phys = {p: i for i, p in enumerate(pp.tile_bits)}

    phys = (lambda gi: []) if independent else (lambda gi: [n - 1 - q for q in gates[gi].qubits])

    def tb(gi):
        g = gates[gi]
        if g.is_diag:
            return []  # diagonal gates need no register bits
        return [tb_of_phys[n - 1 - q] for q in g.qubits]

    # constrained (load/store) layout: tile bit 0 must be a register bit when vec == 2, and the
    # low coalescing bits are pinned to the low lanes.
    lc = min(cfg.lowbits, 5, T - R + (1 if cfg.vec == 2 else 0))
    forced_c = {0} if cfg.vec == 2 else set()
    pinned = set(range(lc)) - forced_c
    allowed_c = set(range(T)) - pinned
    allowed_u = set(range(T))

    def finish_layout(S, allowed):
        S = set(S)
        for b in sorted(allowed, reverse=True):
            if len(S) >= R:
                break
            S.add(b)
        reg = sorted(S)
        thr = [b for b in range(T) if b not in S]
        assert len(reg) == R and len(thr) == T - R
        return reg, thr

    pending = list(pp.gate_ids)
    rounds = []
    # round 0: constrained
    S, chosen = _scan(gates, pending, phys, tb, R, forced_c, allowed_c)
    reg, thr = finish_layout(S, allowed_c)
    rounds.append(Round(reg, thr, chosen))
    cs = set(chosen)
    pending = [g for g in pending if g not in cs]
    while pending:
        S, chosen = _scan(gates, pending, phys, tb, R, forced_c, allowed_c)
        if len(chosen) == len(pending):
            reg, thr = finish_layout(S, allowed_c)
            rounds.append(Round(reg, thr, chosen))
            pending = []
            break
        S, chosen = _scan(gates, pending, phys, tb, R, set(), allowed_u)
        if not chosen:
            raise RuntimeError("round scheduler made no progress")
        # prefer pinned bits when filling (only unconstrained rounds can host them)
        fill = sorted(allowed_u, key=lambda b: (b not in pinned, -b))
        S = set(S)
        for b in fill:
            if len(S) >= R:
                break
            S.add(b)
        reg = sorted(S)
        thr = [b for b in range(T) if b not in S]
        rounds.append(Round(reg, thr, chosen))
        cs = set(chosen)
        pending = [g for g in pending if g not in cs]
        if not pending and not independent:
            reg, thr = finish_layout(forced_c, allowed_c)
            rounds.append(Round(reg, thr, []))
    if cfg.gen >= 2 and not independent and cfg.LT > 6:
        # thread-bit order: tile bits whose qubits share diagonal terms with this round's register qubits go to the
        # highest thread positions (>= 6: the same value for a whole wave), where a register-x-thread phase term only
        # selects a table variant (OP_DIAGCW) instead of costing a multiply of every amplitude (OP_DIAGB)
        pair = {}
        for g in gates:
            if g.is_diag:
                for t in g.diag:
                    qs = list(t.qubits)
                    for a_ in qs:
                        for b_ in qs:
                            if a_ != b_:
                                pair[(a_, b_)] = pair.get((a_, b_), 0) + 1
        qubit_of_tb = {i: n - 1 - p for i, p in enumerate(pp.tile_bits)}
        for k, rd in enumerate(rounds):
            fixed = len(pinned) if (k == 0 or k == len(rounds) - 1) else 0
            head, tail = rd.thr_tb[:fixed], rd.thr_tb[fixed:]
            regq = [qubit_of_tb[b] for b in rd.reg_tb]
            score = lambda b: sum(pair.get((qubit_of_tb[b], rq), 0) for rq in regq)  # noqa: E731
            tail = sorted(tail, key=lambda b: (score(b), b))
            rd.thr_tb = head + tail
    pp.rounds = rounds


# ---- LDS exchange maps -----------------------------------------------------------------
def exchange_masks(T, wr: Round, rd: Round, planar: bool = False):
    """Linear bijection tile-index -> LDS slot for one exchange, chosen so that the write and the read lane groups
    are bank-conflict free.  Complex slots (first-generation kernels, 8-byte elements): 16-lane write groups
    (ds_write_b64), 32-lane read groups (ds_read_b64).  ``planar`` (packed kernels: one 4-byte plane at a time,
    ds_write_b32 / ds_read_b32, 32-lane groups on 32 four-byte banks): slot bits 0..4 <- the tile bits of the
    reading round's lanes 0..4; the writing round's lane bits 0..4 that are not among them are XOR-folded into the
    low slot bits the common ones leave free, so both 5 x 5 low-bit maps are invertible."""
    nlw = min(5 if planar else 4, len(wr.thr_tb))
    nlr = min(5, len(rd.thr_tb))
    W = list(wr.thr_tb[:nlw])
    Rd = list(rd.thr_tb[:nlr])
    both = [x for x in Rd if x in W]
    rd_only = [x for x in Rd if x not in W]
    order = both + rd_only
    pos = {x: i for i, x in enumerate(order)}
    nxt = len(order)
    for x in range(T):
        if x not in pos:
            pos[x] = nxt
            nxt += 1
    used_low = {pos[x] for x in both}
    free_low = [m for m in range(5 if planar else 4) if m not in used_low]
    A = {x: 1 << pos[x] for x in range(T)}
    for x in W:
        if x not in Rd:
            A[x] |= 1 << free_low.pop(0)
    return A


# ---- encoding --------------------------------------------------------------------------
@dataclass
class Tables:
    """Accumulates the constant table and the builder program while passes are encoded."""

    ctab: List[float] = field(default_factory=list)   # shared constants (real values)
    ptab_size: int = 0                                 # per-batch table size (real values)
    ginfo: List[List[int]] = field(default_factory=list)   # builder records
    cpool: List[float] = field(default_factory=list)       # builder constants (float64)
    _slot_cache: dict = field(default_factory=dict)
    pending: list = field(default_factory=list)               # diagonal terms not emitted yet (lazy flush)
    shear2: set = field(default_factory=set)                  # gate ids applied in two-shear form (shear2_gates)
    scale_only: set = field(default_factory=set)              # ... whose factor needed a table of its own (no phase term to ride on)
    fold_slots: dict = field(default_factory=dict)            # adjoint: gradient slot of folded Pauli term i (compile_adjoint_plan ``fold``)
    gslot_param: List[int] = field(default_factory=list)     # adjoint: parameter index per slot
    gslot_factor: List[float] = field(default_factory=list)  # adjoint: d(theta)/d(slot value)

    def grad_slot(self, pidx: int, factor: float) -> int:
        self.gslot_param.append(pidx)
        self.gslot_factor.append(factor)
        return len(self.gslot_param) - 1

    def const_complex(self, m):
        off = len(self.ctab)
        for z in np.asarray(m, dtype=np.complex128).reshape(-1):
            self.ctab += [float(z.real), float(z.imag)]
        return off | CONST_FLAG

    def const_real(self, v):
        off = len(self.ctab)
        self.ctab.append(float(v))
        return off | CONST_FLAG

    def alloc(self, nreal):
        off = self.ptab_size
        self.ptab_size += nreal
        return off


TWO_PI = 2.0 * np.pi


def g1_kind(g: GateRec, tol=1e-14) -> int:
    """Structure class of a 1-qubit gate (valid for every parameter value): 1 = real matrix,
    2 = real diagonal + imaginary off-diagonal (rx-like), 0 = general.  The kernel spends 8 instead
    of 16 FMAs per amplitude pair on classes 1 and 2."""
    ms = [np.asarray(m, dtype=np.complex128) for m in (g.c0, g.c1, g.c2) if m is not None]
    if g.select is not None:
        ms = [np.asarray(m, dtype=np.complex128) for m in g.select]
    if all(np.abs(m.imag).max() < tol for m in ms):
        return 1
    if all(
        abs(m[0, 0].imag) < tol and abs(m[1, 1].imag) < tol and abs(m[0, 1].real) < tol and abs(m[1, 0].real) < tol
        for m in ms
    ):
        return 2
    return 0


_SHEAR_ANGLES = (0.3, 1.1, 2.9, 4.4, 5.7)
SHEAR_SHIFT = 20  # bit 20 + j of a G1M mask word: the gate on register bit j is in three-shear form (gen-2 plans)


def g1_shear_flavor(g: GateRec, tol=1e-12) -> int:
    """1 / 2 if the one-qubit gate is, for every parameter value, a rotation of the real class (kind 1:
    [[c, -s], [s, c]], ry-like) / of the rx-like class (kind 2: [[c, i b], [i b, c]] with c^2 + b^2 = 1), else 0.
    Such a gate is applied as three in-place shears x += u y', y' += v x, x += u y' (3 instead of 4 packed
    instructions per amplitude pair, |u| <= 1 after pulling out a global sign); csrc/tcmi_vm2.hip."""
    if g.select is not None or len(g.qubits) != 1:
        return 0
    kind = g1_kind(g)
    if kind not in (1, 2):
        return 0
    mats = [np.asarray(m, dtype=np.complex128) if m is not None else np.zeros((2, 2)) for m in (g.c0, g.c1, g.c2)]
    angles = _SHEAR_ANGLES if g.param is not None else (0.0,)
    for a_ in angles:
        m = mats[0] + np.cos(a_) * mats[1] + np.sin(a_) * mats[2]
        if kind == 1:
            a, b, c, d = m[0, 0].real, m[0, 1].real, m[1, 0].real, m[1, 1].real
            ok = abs(a * d - b * c - 1) < tol and abs(a - d) < tol and abs(b + c) < tol
        else:
            a, b, c, d = m[0, 0].real, m[0, 1].imag, m[1, 0].imag, m[1, 1].real
            ok = abs(a * d + b * c - 1) < tol and abs(a - d) < tol and abs(b - c) < tol
        if not ok:
            return 0
    return kind


SHEAR2_CMIN = 0.5      # two-shear form only while |cos| >= this (per batch element, decided by the builder kernel)
SCALE_TERM = 1 << 16   # flag in the register-mask field of a BK_PHASE term: real scale term (DiagTerm.scale)


def shear2_gates(gates: List[GateRec], order: Sequence[int], cfg: PlanConfig) -> set:
    """Gates of the schedule ``order`` that may run in TWO-shear form.  An rx-like rotation U = [[c, i b], [i b, c]]
    is diag(c, 1/c) L(i b c) S(i b / c): two in-place shears and a REAL diagonal factor (the same holds for the real
    class [[c, -s], [s, c]], but the kernels carry a two-shear body for the rx-like class only, see csrc/tcmi_vm2.hip).
    The factor is left pending like a diagonal gate (it commutes with everything that does not act on the qubit) and
    is folded into the phase table of the flush that precedes the next dense gate on the qubit -- 2 instead of 3
    packed instructions per amplitude pair when that flush multiplies by a table anyway.  Eligible: parametrised
    rotations with U[0, 0] = cos(angle) exactly, whose next gate on the qubit is diagonal (so a table flush is due)
    and which are not the last dense gate on the qubit (the factor must be gone when the plan ends).  At run time
    the builder kernel falls back to the three-shear form for batch elements with |cos| < SHEAR2_CMIN."""
    if cfg.gen < 2:
        return set()
    out = set()
    nxt_diag = {}     # qubit -> True if the following gate on it (in schedule order) is diagonal
    later_dense = set()
    for gi in reversed(list(order)):
        g = gates[gi]
        if g.is_diag:
            for q in g.qubits:
                nxt_diag[q] = True
            continue
        if (len(g.qubits) == 1 and g.param is not None and g.select is None and g1_shear_flavor(g) == 2
                and nxt_diag.get(g.qubits[0], False) and g.qubits[0] in later_dense):
            c0, c1, c2 = (np.asarray(m, dtype=np.complex128) for m in (g.c0, g.c1, g.c2))
            if abs(c0[0, 0]) < 1e-15 and abs(c1[0, 0] - 1) < 1e-15 and abs(c2[0, 0]) < 1e-15:
                out.add(gi)
        for q in g.qubits:
            nxt_diag[q] = False
            later_dense.add(q)
    return out


def _gate_slot(tables: Tables, gi: int, g: GateRec, swap: bool):
    """Table slot holding the dense matrix of gate g (optionally with its two qubits swapped)."""
    key = (gi, swap)
    if key in tables._slot_cache:
        return tables._slot_cache[key]
    dim = 2 ** len(g.qubits)

    def sw(m):
        m = np.asarray(m, dtype=np.complex128).reshape(dim, dim)
        if swap:
            m = m.reshape(2, 2, 2, 2).transpose(1, 0, 3, 2).reshape(4, 4)
        return m

    if g.select is not None:
        assert kind == BK_TRIG, "select gates exist in forward plans only"
        off = len(tables.cpool)
        tables.cpool += [float(len(g.select)), 0.0]
        for m in g.select:
            for z in sw(m).reshape(-1):
                tables.cpool += [float(z.real), float(z.imag)]
        tables.ginfo.append([BK_SELECT, slot, g.param.index, dim, off, 0, 0, 0])
        return

    if g.param is None:
        slot = tables.const_complex(sw(g.c0))
    else:
        slot = tables.alloc(2 * dim * dim)
        off = len(tables.cpool)
        tables.cpool += [g.param.scale, g.param.offset]
        for m in (g.c0, g.c1, g.c2):
            for z in sw(m).reshape(-1):
                tables.cpool += [float(z.real), float(z.imag)]
        tables.ginfo.append([BK_TRIG, slot, g.param.index, dim, off, 0, 0, 0])
    tables._slot_cache[key] = slot
    return slot


def g2_kind(g: GateRec, swap: bool, tol=1e-14) -> int:
    """Permutation classes of constant 2-qubit gates (no flops in the kernel): in the canonical
    register order (ja < jb, matrix MSB <-> ja): 1 = CNOT control ja, 2 = CNOT control jb, 3 = SWAP."""
    if g.param is not None:
        return 0
    m = np.asarray(g.c0, dtype=np.complex128).reshape(4, 4)
    if swap:
        m = m.reshape(2, 2, 2, 2).transpose(1, 0, 3, 2).reshape(4, 4)
    cx_a = np.eye(4)[[0, 1, 3, 2]]
    cx_b = np.eye(4)[[0, 3, 2, 1]]
    sw = np.eye(4)[[0, 2, 1, 3]]
    for k, ref in ((1, cx_a), (2, cx_b), (3, sw)):
        if np.abs(m - ref).max() < tol:
            return k
    return 0


def _matrix_record(tables: Tables, slot: int, g: GateRec, kind: int = BK_TRIG, swap: bool = False, shear: int = 0,
                   two: bool = False):
    """Builder record writing a dense matrix derived from g to ptab[slot ..]: the gate itself
    (BK_TRIG), its adjoint (BK_UDAG) or K = dU U^dagger (BK_KMAT).  Constant gates included, so that
    the matrices of one op can sit contiguously."""
    dim = 2 ** len(g.qubits)

    def sw(m):
        m = np.asarray(m, dtype=np.complex128).reshape(dim, dim)
        if swap:
            m = m.reshape(2, 2, 2, 2).transpose(1, 0, 3, 2).reshape(4, 4)
        return m

    if g.select is not None:
        assert kind == BK_TRIG, "select gates exist in forward plans only"
        off = len(tables.cpool)
        tables.cpool += [float(len(g.select)), 0.0]
        for m in g.select:
            for z in sw(m).reshape(-1):
                tables.cpool += [float(z.real), float(z.imag)]
        tables.ginfo.append([BK_SELECT, slot, g.param.index, dim, off, 0, 0, 0])
        return

    off = len(tables.cpool)
    if g.param is None:
        tables.cpool += [0.0, 0.0]
        mats = (g.c0, np.zeros((dim, dim)), np.zeros((dim, dim)))
        pidx = 0
    else:
        tables.cpool += [g.param.scale, g.param.offset]
        mats = (g.c0, g.c1, g.c2)
        pidx = g.param.index
    for m in mats:
        for z in sw(m).reshape(-1):
            tables.cpool += [float(z.real), float(z.imag)]
    # shear != 0: the builder writes {u, v, sign, flag} of the three-shear form of the matrix instead of its entries;
    # two: it may choose the two-shear form (flag = 2.0, sign = 1, the factor diag(c, 1/c) is a pending scale term)
    tables.ginfo.append([kind, slot, pidx, dim, off, shear, int(two), 0])


def _coef_record(tables: Tables, slot: int, t: DiagTerm):
    """Builder record writing the phase coefficient of term t (in turns) to ptab[slot]."""
    off = len(tables.cpool)
    if t.param is None:
        tables.cpool += [0.0, t.const / TWO_PI]
        tables.ginfo.append([BK_COEF, slot, 0, 1, off, 0, 0, 0])
    else:
        tables.cpool += [t.param.scale / TWO_PI, (t.const + t.param.offset) / TWO_PI]
        tables.ginfo.append([BK_COEF, slot, t.param.index, 1, off, 0, 0, 0])


def _coef_slot(tables: Tables, gi: int, ti: int, t: DiagTerm):
    key = (gi, "d", ti)
    if key in tables._slot_cache:
        return tables._slot_cache[key]
    if t.param is None:
        slot = tables.const_real(t.const / TWO_PI)
    else:
        slot = tables.alloc(1)
        off = len(tables.cpool)
        tables.cpool += [t.param.scale / TWO_PI, (t.const + t.param.offset) / TWO_PI]
        tables.ginfo.append([BK_COEF, slot, t.param.index, 1, off, 0, 0, 0])
    tables._slot_cache[key] = slot
    return slot


def encode_pass(gates: List[GateRec], n: int, cfg: PlanConfig, pp: PassPlan, tables: Tables,
                backward: bool = False, phase_tables: bool = True, final: bool = True,
                factorized_bw: bool = False) -> np.ndarray:
    """``backward=True`` encodes an adjoint-sweep pass: ``gates`` is the reversed gate list, every op
    carries U^dagger (+ K and a gradient slot for parametrised gates), see csrc/tcmi_vm.h.

    Diagonal gates are *lazy*: their terms go to ``tables.pending`` (which survives rounds and passes -- a phase
    polynomial is a function of the global index, not of a layout) and a term is emitted only when a dense gate is
    about to act on one of its qubits.  At that moment the qubit is a register bit, so every emitted term is of
    the register-only (table) or register-x-thread (sign) form and no per-thread sin / cos is evaluated; all the
    terms due before one group of one-qubit gates share a single table.  ``final`` (last pass of the plan) flushes
    what is still pending at the end.  ``factorized_bw``: backward passes use OP_DIAGF (tables from the builder,
    gradients per term) instead of the generic OP_DIAG -- only csrc/tcmi_adjoint2.hip executes it."""
    T, R, LT = cfg.T, cfg.R, cfg.LT
    assert R <= R_MAX and LT <= LT_MAX and T <= T_MAX
    words = [0] * HDR_WORDS
    words[0] = MAGIC
    words[1] = n
    words[2] = T
    words[3] = R
    words[4] = LT
    words[5] = len(pp.rounds)
    words[6] = 0
    for i, p in enumerate(pp.tile_bits):
        words[8 + i] = p
    tb_of_phys = {p: i for i, p in enumerate(pp.tile_bits)}
    nr = len(pp.rounds)
    exch = [exchange_masks(T, pp.rounds[k], pp.rounds[k + 1], planar=cfg.gen >= 2) for k in range(nr - 1)]
    for k, rd in enumerate(pp.rounds):
        rr = [0] * RR_WORDS
        for j, b in enumerate(rd.reg_tb):
            rr[2 + j] = 1 << pp.tile_bits[b]
        for i, b in enumerate(rd.thr_tb):
            rr[8 + i] = 1 << pp.tile_bits[b]
        if k > 0:
            A = exch[k - 1]
            for j, b in enumerate(rd.reg_tb):
                rr[18 + j] = A[b]
            for i, b in enumerate(rd.thr_tb):
                rr[24 + i] = A[b]
        if k < nr - 1:
            A = exch[k]
            for j, b in enumerate(rd.reg_tb):
                rr[34 + j] = A[b]
            for i, b in enumerate(rd.thr_tb):
                rr[40 + i] = A[b]
        reg_of_tb = {b: j for j, b in enumerate(rd.reg_tb)}
        regphys = {pp.tile_bits[b]: j for j, b in enumerate(rd.reg_tb)}
        ops = []
        nops = 0
        pend_g1 = {}      # register bit -> gate id: one-qubit gates on distinct bits, emitted as one G1M op
        pend_flush = []   # diagonal terms due before that group

        def flush_g1():
            nonlocal nops
            if not pend_g1:
                return
            mk = 0
            base = tables.alloc(8 * R)  # the R matrices of one G1M op are contiguous
            if backward:
                kmask, kbase, gs = 0, tables.alloc(8 * R), [0] * R
            for j, gi in pend_g1.items():
                g = gates[gi]
                mk |= (1 << j) | (g1_kind(g) << (8 + 2 * j))
                sh = g1_shear_flavor(g) if cfg.gen >= 2 else 0
                if sh:
                    mk |= 1 << (SHEAR_SHIFT + j)
                if not backward:
                    _matrix_record(tables, base + 8 * j, g, shear=sh, two=gi in tables.shear2)
                else:
                    _matrix_record(tables, base + 8 * j, g, BK_UDAG, shear=sh, two=gi in tables.shear2)
                    if g.param is not None:
                        kmask |= 1 << j
                        _matrix_record(tables, kbase + 8 * j, g, BK_KMAT)
                        gs[j] = tables.grad_slot(g.param.index, 1.0)
            if backward:
                ops.extend([OP_G1M, mk, base, kmask, kbase] + gs)
            else:
                ops.extend([OP_G1M, mk, base])
            nops += 1
            pend_g1.clear()

        def phase_terms(ts):
            """Builder constants of one table entry: {k, o, param index, register mask} per term (in turns)."""
            off = len(tables.cpool)
            for rm, t in ts:
                if t.scale:   # real factor c^(+-1), c = cos(k theta + o) in RADIANS (same arithmetic as the gate record)
                    tables.cpool += [t.param.scale, t.param.offset, float(t.param.index), float(rm | SCALE_TERM)]
                    continue
                if t.param is None:
                    tables.cpool += [0.0, t.const / TWO_PI, 0.0, float(rm)]
                else:
                    tables.cpool += [t.param.scale / TWO_PI, (t.const + t.param.offset) / TWO_PI,
                                     float(t.param.index), float(rm)]
            return off

        def emit_cnot(jc, jt):
            """CNOT between register bits (control jc, target jt): a register move of the kernels' G2 op."""
            nonlocal nops
            ja, jb, kind = (jc, jt, 1) if jc < jt else (jt, jc, 2)
            if not backward:
                ops.extend([OP_G2, ja | (kind << 8), jb, 0])
            else:
                ops.extend([OP_G2, ja | (kind << 8), jb, 0, -1, 0])
            nops += 1

        def emit_diag(terms):
            """Emit the diagonal terms ``terms`` (DiagTerm list) at the current point of this round.  The kernels' phase
            ops know three sign patterns: thread bits only, ONE register bit x thread bits, register bits only.  A term of
            three or more qubits (multi-controlled phases, the diagonal factors of synthesised dense gates, ``rzm``) can
            have two or more register bits AND thread bits: its register parity is first folded into one register bit by
            CNOTs between register bits (register moves), the term is applied as a one-register-bit term, and the CNOTs
            are undone -- P^dagger D_t P has the sign (-1)^{(P x)_t} = the parity of the folded bits."""
            if not terms:
                return
            plain, mixed = [], {}
            for t in terms:
                pb = [n - 1 - q for q in t.qubits]
                rb = sorted(regphys[p] for p in pb if p in regphys)
                if len(rb) >= 2 and any(p not in regphys for p in pb):
                    mixed.setdefault(tuple(rb), []).append(t)
                else:
                    plain.append(t)
            emit_diag_core(plain)
            phys_of_reg = {j: p for p, j in regphys.items()}
            for rb, ts in mixed.items():
                tgt = rb[0]
                drop = {n - 1 - phys_of_reg[j] for j in rb[1:]}       # qubits folded into the target bit
                for jc in rb[1:]:
                    emit_cnot(jc, tgt)
                emit_diag_core([DiagTerm(tuple(q for q in t.qubits if q not in drop), t.const, t.param) for t in ts])
                for jc in reversed(rb[1:]):
                    emit_cnot(jc, tgt)

        def emit_diag_core(terms):
            nonlocal nops
            if not terms:
                return
            A_, B_, C_ = [], [], []
            for t in terms:
                pbits = [n - 1 - q for q in t.qubits]
                rbits = [regphys[p] for p in pbits if p in regphys]
                if t.scale:   # flushed right before a dense gate on its qubit: always a register bit, always the table
                    assert len(rbits) == 1 and len(pbits) == 1, "scale term off the register bits"
                    C_.append((1 << rbits[0], t))
                    continue
                nmask = 0
                for p in pbits:
                    if p not in regphys:
                        nmask |= 1 << p
                if len(rbits) == 0:
                    A_.append((nmask, t))
                elif len(rbits) == 1:
                    B_.append((rbits[0], nmask, t))
                elif nmask == 0:
                    rm = 0
                    for j in rbits:
                        rm |= 1 << j
                    C_.append((rm, t))
                else:
                    raise NotImplementedError(
                        "diagonal term with >=2 register bits and non-register bits"
                    )
            # register-x-thread terms with the same sign function share one phase factor
            bgroups = {}
            for j, nmask, t in B_:
                bgroups.setdefault((j, nmask), []).append(t)

            def gslot(t):
                # d(phase)/d(theta) = scale * s_t(idx); dL/dtheta = -scale * sum s_t Im(conj(lambda) psi)
                if t.scale:
                    return -1   # part of its rotation gate, whose gradient is taken there
                return tables.grad_slot(t.param.index, -t.param.scale) if t.param is not None else -1

            # gen-2 kernels: a register-x-thread term whose non-register bits are the same for a whole wave (bits outside
            # the tile, thread positions >= 6) only picks a variant of the register table; up to 3 such sign functions
            wsel = []     # [(mask, [(j, term), ...])]
            if cfg.gen >= 2 and phase_tables and (not backward or factorized_bw):
                uni = 0
                for p_ in range(n):
                    if p_ not in tb_of_phys:
                        uni |= 1 << p_
                for pos_, b_ in enumerate(rd.thr_tb):
                    if pos_ >= 6:
                        uni |= 1 << pp.tile_bits[b_]
                for (j, nmask) in list(bgroups):
                    if nmask and (nmask & ~uni) == 0:
                        ent = next((w for w in wsel if w[0] == nmask), None)
                        if ent is None:
                            if len(wsel) >= 3:
                                continue
                            ent = (nmask, [])
                            wsel.append(ent)
                        ent[1].extend((j, t) for t in bgroups.pop((j, nmask)))

            def table_variants(twin=False):
                """Builder records of the (possibly wave-selected) register table; returns its slot.  ``twin`` (adjoint
                sweep, scale terms present): a second table with the reciprocal real factors follows it -- psi carries
                diag(c, 1/c), lambda its inverse, so that every gradient bilinear <lambda| . |psi> is unchanged."""
                NR = 1 << R
                nv = 1 << len(wsel)
                slot_ = tables.alloc(2 * NR * nv * (2 if twin else 1))
                terms_ = [(rm, t) for rm, t in C_]
                if not wsel and all(t.scale for _, t in C_):
                    tables.scale_only.update(t.gate for _, t in C_)
                for k_, (_m, lst) in enumerate(wsel):
                    terms_ += [((1 << j) | (1 << (R + k_)), t) for j, t in lst]
                off_ = phase_terms(terms_)
                for inv_ in range(2 if twin else 1):
                    for v_ in range(nv):
                        for r in range(NR):
                            tables.ginfo.append([BK_PHASE, slot_ + 2 * ((inv_ * nv + v_) * NR + r), 0, len(terms_), off_,
                                                 r | (v_ << R), inv_, 0])
                return slot_

            if backward and factorized_bw:
                # {8, cslot (-1: none), hasC, nB, nA, nsel, m0, m1, m2, gsC[2^R] (gradient slot of the register-only term
                #  with mask k, -1: none), B: (j, mask, slot (-1: applied elsewhere), gslot)*, A: (mask, gslot)*}: tables hold the FORWARD phase
                #  factors (2^nsel wave-selected variants of the register table), the kernel applies the conjugate;
                #  A terms (no register bit: only the final flush has them) contribute gradients only.
                twin = any(t.scale for _, t in C_)
                cslot = table_variants(twin) if (C_ or wsel) else -1
                NR_ = 1 << R
                gsc = [-1] * NR_          # gradient slot per register mask (the term's Walsh coefficient of w)
                dup = []                  # further terms with a mask already taken: gradient-only follow-up ops
                for rm, t in C_:
                    gs_ = gslot(t)
                    if gs_ < 0:
                        continue
                    if gsc[rm] < 0:
                        gsc[rm] = gs_
                    else:
                        dup.append((rm, gs_))
                body = list(gsc)
                nB = 0
                for m_, lst in wsel:          # applied through the table variant: gradient entries only
                    for j, t in lst:
                        body += [j, m_, -1, gslot(t)]
                        nB += 1
                for (j, nmask), ts in bgroups.items():
                    slot = tables.alloc(2)
                    off = phase_terms([(0, t) for t in ts])
                    tables.ginfo.append([BK_PHASE, slot, 0, len(ts), off, 0, 0, 0])
                    for i_, t in enumerate(ts):
                        body += [j, nmask, slot if i_ == 0 else -1, gslot(t)]
                        nB += 1
                for nmask, t in A_:
                    body += [nmask, gslot(t)]
                sel = [m_ for m_, _ in wsel] + [0] * (3 - len(wsel))
                # hasC: bit 0 = some register-only term has a gradient slot, bit 1 = lambda has its own table (after psi's)
                ops.extend([OP_DIAGF, cslot, int(any(g_ >= 0 for g_ in gsc)) | (2 if twin else 0), nB, len(A_), len(wsel)]
                           + sel + body)
                nops += 1
                while dup:
                    gsc2, rest = [-1] * NR_, []
                    for rm, gs_ in dup:
                        if gsc2[rm] < 0:
                            gsc2[rm] = gs_
                        else:
                            rest.append((rm, gs_))
                    ops.extend([OP_DIAGF, -1, 1, 0, 0, 0, 0, 0, 0] + gsc2)
                    nops += 1
                    dup = rest
                return
            bops = []
            if cfg.gen >= 2:
                byj = {}
                for (j, nmask), ts in bgroups.items():
                    byj.setdefault(j, []).append((nmask, ts))
                for j, lst in byj.items():
                    for i_ in range(0, len(lst) - 1, 2):
                        bops.append((j, lst[i_], lst[i_ + 1]))
                    if len(lst) % 2:
                        bops.append(((j, lst[-1][0]), lst[-1][1]))
            else:
                bops = [((j, nmask), ts) for (j, nmask), ts in bgroups.items()]
            if not backward and not A_ and len(bops) <= MAX_DIAGB and phase_tables:
                # No per-thread sincos: the builder evaluates every phase factor once, in float64.
                # Terms on register bits only -> one table of 2^R factors, the same for all threads
                # (DIAGC); terms on one register bit and thread bits -> exp(+-i phi) with the sign
                # z_j(r) * parity(thread & mask) (DIAGB).  4 lane-instructions per amplitude each.
                if wsel:
                    sel = [m_ for m_, _ in wsel] + [0] * (3 - len(wsel))
                    ops.extend([OP_DIAGCW, table_variants(), len(wsel)] + sel)
                    nops += 1
                elif C_:
                    ops.extend([OP_DIAGC, table_variants()])
                    nops += 1
                for it in bops:
                    if len(it) == 2:
                        (j, nmask), ts = it[0], it[1]
                        base = tables.alloc(2)
                        off = phase_terms([(0, t) for t in ts])
                        tables.ginfo.append([BK_PHASE, base, 0, len(ts), off, 0, 0, 0])
                        ops.extend([OP_DIAGB, j, nmask, base])
                    else:
                        # two sign functions on one register bit: factor = table[s1 + 2 s2], entry c holds
                        # exp(i(+-phi1 +- phi2)) with the signs of the bits of c (BK_PHASE: "register mask" = 1 << group)
                        j, (m1, ts1), (m2, ts2) = it
                        base = tables.alloc(8)
                        off = phase_terms([(1, t) for t in ts1] + [(2, t) for t in ts2])
                        for c_ in range(4):
                            tables.ginfo.append([BK_PHASE, base + 2 * c_, 0, len(ts1) + len(ts2), off, c_, 0, 0])
                        ops.extend([OP_DIAGB2, j, m1, m2, base])
                    nops += 1
                return
            # generic form: coefficients live contiguously in the per-batch table (bulk scalar loads), padded
            # to a multiple of DIAG_CHUNK with zero terms
            sc_ = [(rm, t) for rm, t in C_ if t.scale]
            if sc_:   # scale terms are not phases: their own table multiply first
                assert not backward
                C_keep, C_[:] = [x for x in C_ if not x[1].scale], sc_
                wsel_keep, wsel[:] = list(wsel), []
                ops.extend([OP_DIAGC, table_variants()])
                nops += 1
                C_[:] = C_keep
                wsel[:] = wsel_keep
            padA = (-len(A_)) % DIAG_CHUNK
            padB = (-len(B_)) % DIAG_CHUNK
            nA, nB, nC = len(A_) + padA, len(B_) + padB, len(C_)
            base = tables.alloc(nA + nB + nC)
            zero = DiagTerm((), 0.0, None)
            tl = [t for _, t in A_] + [zero] * padA + [t for _, _, t in B_] + [zero] * padB + [t for _, t in C_]
            for k_, t in enumerate(tl):
                _coef_record(tables, base + k_, t)
            ops.extend([OP_DIAG, nA, nB, nC, base])
            ops.extend([m for m, _ in A_] + [0] * padA)
            ops.extend([m for _, m, _ in B_] + [0] * padB)
            ops.extend([j for j, _, _ in B_] + [0] * padB)
            ops.extend([m for m, _ in C_])
            if backward:
                ops.extend([gslot(t) for t in tl])
            nops += 1

        def flush_group():
            emit_diag(pend_flush)
            pend_flush.clear()
            flush_g1()

        def take_pending(qubits):
            """Remove and return the pending terms that touch one of ``qubits``."""
            qs = set(qubits)
            hit = [t for t in tables.pending if qs & set(t.qubits)]
            if hit:
                tables.pending = [t for t in tables.pending if not (qs & set(t.qubits))]
            return hit

        if getattr(rd, "dfold", None):
            assert backward
            regmask_phys = 0
            for j_, b_ in enumerate(rd.reg_tb):
                regmask_phys |= 1 << pp.tile_bits[b_]
            ops.extend([OP_DFOLD, len(rd.dfold), tables.fold_slots["diag"]])
            for zm_, cf_ in rd.dfold:
                rm_ = 0
                for j_, b_ in enumerate(rd.reg_tb):
                    if (zm_ >> pp.tile_bits[b_]) & 1:
                        rm_ |= 1 << j_
                ops.extend([zm_ & ~regmask_phys, rm_, tables.const_real(cf_)])
            nops += 1
        for (jf, cf_, fi_, kd_) in getattr(rd, "fold", []):
            assert backward
            ops.extend([OP_XFOLD, jf | (int(kd_) << 8), tables.const_real(cf_), tables.fold_slots[fi_]])
            nops += 1
        for (ja_, jb_, cf_, pi_, ka_, kb_) in getattr(rd, "fold2", []):
            assert backward
            ops.extend([OP_XFOLD2, ja_ | (jb_ << 8) | (int(ka_) << 16) | (int(kb_) << 17), tables.const_real(cf_),
                        tables.fold_slots[("pair", pi_)]])
            nops += 1
        for gi in rd.gates:
            g = gates[gi]
            if g.is_diag:
                tables.pending.extend(g.diag)
                continue
            tbs = [tb_of_phys[n - 1 - q] for q in g.qubits]
            js = [reg_of_tb[b] for b in tbs]
            if len(js) == 1:
                due = take_pending(g.qubits)
                group_qubits = set()
                for gj in pend_g1.values():
                    group_qubits |= set(gates[gj].qubits)
                # a due term that also touches a qubit of the open group comes AFTER that group's gate
                if js[0] in pend_g1 or any(group_qubits & set(t.qubits) for t in due):
                    flush_group()
                pend_flush.extend(due)
                pend_g1[js[0]] = gi
                if gi in tables.shear2:   # its factor diag(c, 1/c) follows the gate: pending from here on
                    tables.pending.append(DiagTerm(g.qubits, 0.0, ParamRef(g.param.index, g.param.scale, g.param.offset),
                                                   scale=True, gate=gi))
            elif len(js) == 2:
                flush_group()
                emit_diag(take_pending(g.qubits))
                swap = js[0] > js[1]
                ja, jb = (js[1], js[0]) if swap else (js[0], js[1])
                if not backward:
                    ops.extend([OP_G2, ja | (g2_kind(g, swap) << 8), jb, _gate_slot(tables, gi, g, swap)])
                else:
                    uslot = tables.alloc(32)
                    _matrix_record(tables, uslot, g, BK_UDAG, swap)
                    kslot, gslot_ = -1, 0
                    if g.param is not None:
                        kslot = tables.alloc(32)
                        _matrix_record(tables, kslot, g, BK_KMAT, swap)
                        gslot_ = tables.grad_slot(g.param.index, 1.0)
                    ops.extend([OP_G2, ja | (g2_kind(g, swap) << 8), jb, uslot, kslot, gslot_])
                nops += 1
            else:
                raise NotImplementedError
        flush_group()
        if final and k == nr - 1 and tables.pending:
            emit_diag(tables.pending)
            tables.pending = []
        rr[0] = nops
        rr[1] = len(ops)
        words += rr + ops
    arr = np.array(words, dtype=np.int64)
    # masks may use bit 31: store as uint32 bit patterns in int32
    return arr.astype(np.uint32).view(np.int32)


@dataclass
class CompiledPlan:
    n: int
    cfg: PlanConfig
    passes: List[PassPlan]
    descs: List[np.ndarray]
    ctab: np.ndarray       # float64 constants (cast to the state's real dtype on upload)
    ptab_size: int
    ginfo: np.ndarray      # int32 [G, 8]
    cpool: np.ndarray      # float64
    nparams: int

    def stats(self, itemsize):
        """Algorithmic bytes/flops of the executed plan: each pass reads
        and writes the state once."""
        npass = len(self.passes)
        return {
            "passes": npass,
            "rounds": sum(len(p.rounds) for p in self.passes),
            "bytes": npass * 2 * (2**self.n) * itemsize,
        }


def compile_plan(gates: List[GateRec], n: int, cfg: PlanConfig, nparams: int = 0) -> CompiledPlan:
    passes = schedule(gates, n, cfg)
    sh2 = shear2_gates(gates, [gi for pp in passes for rd in pp.rounds for gi in rd.gates], cfg) if cfg.shear2 else set()
    while True:
        tables = Tables()
        tables.ctab += [0.0] * 8
        tables.shear2 = set(sh2)
        descs = [encode_pass(gates, n, cfg, pp, tables, final=(i == len(passes) - 1)) for i, pp in enumerate(passes)]
        if not tables.scale_only:
            break
        # a real factor that found no phase table to ride on costs a table multiply of its own -- more than the
        # third shear it saves: those rotations go back to the three-shear form
        sh2 -= tables.scale_only
    ginfo = np.array(tables.ginfo, dtype=np.int32).reshape(-1, 8)
    return CompiledPlan(
        n=n, cfg=cfg, passes=passes, descs=descs,
        ctab=np.array(tables.ctab, dtype=np.float64),
        ptab_size=tables.ptab_size, ginfo=ginfo,
        cpool=np.array(tables.cpool, dtype=np.float64), nparams=nparams,
    )


# ---- measurement plans (fused Pauli-sum expectation) ---------------------------------------------
@dataclass
class PauliTerm:
    """One Pauli string: ``x`` = qubits carrying X or Y, ``z`` = qubits carrying Z or Y (a Y qubit is
    in both).  <psi|P|psi> = i^{nY} * sum_idx (-1)^{popc(idx & zmask)} conj(psi[idx ^ xmask]) psi[idx];
    the kernel returns the sum, the host applies i^{nY}."""

    x: Tuple[int, ...]
    z: Tuple[int, ...]

    @property
    def qubits(self):
        return self.x

    is_diag = False

    @property
    def ny(self):
        return len(set(self.x) & set(self.z))


def pauli_term_from_string(ps: Sequence[int]) -> PauliTerm:
    """ps[i] in {0,1,2,3} = I,X,Y,Z on qubit i (reference abstractcircuit.py:1583-1603)."""
    x = tuple(i for i, p in enumerate(ps) if p in (1, 2))
    z = tuple(i for i, p in enumerate(ps) if p in (2, 3))
    return PauliTerm(x, z)


def encode_measure_pass(terms: List[PauliTerm], n: int, cfg: PlanConfig, pp: PassPlan) -> np.ndarray:
    T, R, LT = cfg.T, cfg.R, cfg.LT
    words = [0] * HDR_WORDS
    words[0:7] = [MAGIC, n, T, R, LT, len(pp.rounds), FLAG_NOSTORE]
    for i, p in enumerate(pp.tile_bits):
        words[8 + i] = p
    nr = len(pp.rounds)
    exch = [exchange_masks(T, pp.rounds[k], pp.rounds[k + 1], planar=cfg.gen >= 2) for k in range(nr - 1)]
    for k, rd in enumerate(pp.rounds):
        rr = [0] * RR_WORDS
        for j, b in enumerate(rd.reg_tb):
            rr[2 + j] = 1 << pp.tile_bits[b]
        for i, b in enumerate(rd.thr_tb):
            rr[8 + i] = 1 << pp.tile_bits[b]
        if k > 0:
            A = exch[k - 1]
            for j, b in enumerate(rd.reg_tb):
                rr[18 + j] = A[b]
            for i, b in enumerate(rd.thr_tb):
                rr[24 + i] = A[b]
        if k < nr - 1:
            A = exch[k]
            for j, b in enumerate(rd.reg_tb):
                rr[34 + j] = A[b]
            for i, b in enumerate(rd.thr_tb):
                rr[40 + i] = A[b]
        regphys = {pp.tile_bits[b]: j for j, b in enumerate(rd.reg_tb)}
        zs, xs = [], []
        for ti in rd.gates:
            t = terms[ti]
            xr = 0
            for q in t.x:
                xr |= 1 << regphys[n - 1 - q]
            zr, zm = 0, 0
            for q in t.z:
                p = n - 1 - q
                if p in regphys:
                    zr |= 1 << regphys[p]
                else:
                    zm |= 1 << p
            if xr == 0:
                zs += [zr, zm, ti]
            else:
                xs += [xr, zr, zm, ti]
        ops = []
        if (zs or xs) and cfg.gen >= 2:
            # {11, nX, gmask, X: (xr, zr, zm, out)*, then for every set bit k of gmask (ascending): count, (zm, out)*}:
            # the Z-only strings grouped by their register mask k -- the kernel reads the signed sum over the
            # registers from the Walsh-Hadamard transform of |a|^2 (csrc/tcmi_measure2.hip)
            groups: Dict[int, List[int]] = {}
            for i in range(0, len(zs), 3):
                groups.setdefault(zs[i], []).extend([zs[i + 1], zs[i + 2]])
            gmask = 0
            body = []
            for k in sorted(groups):
                gmask |= 1 << k
                body += [len(groups[k]) // 2] + groups[k]
            ops = [OP_EXPECT2, len(xs) // 4, gmask] + xs + body
        elif zs or xs:
            ops = [OP_EXPECT, len(zs) // 3, len(xs) // 4] + zs + xs
        rr[0] = 1 if ops else 0
        rr[1] = len(ops)
        words += rr + ops
    return np.array(words, dtype=np.int64).astype(np.uint32).view(np.int32)


@dataclass
class MeasurePlan:
    n: int
    cfg: PlanConfig
    terms: List[PauliTerm]
    passes: List[PassPlan]
    descs: List[np.ndarray]


def compile_measure_plan(terms: List[PauliTerm], n: int, cfg: PlanConfig) -> MeasurePlan:
    for t in terms:
        if len(t.x) > 2:
            raise NotImplementedError(
                "Pauli strings with more than two X/Y factors are not supported by the hip expectation kernel"
            )
    passes = schedule(terms, n, cfg, independent=True)
    descs = [encode_measure_pass(terms, n, cfg, pp) for pp in passes]
    return MeasurePlan(n, cfg, list(terms), passes, descs)


# ---- adjoint (reverse sweep) plans -----------------------------------------------------------------
@dataclass
class AdjointPlan:
    n: int
    cfg: PlanConfig
    passes: List[PassPlan]
    descs: List[np.ndarray]
    ctab: np.ndarray
    ptab_size: int
    ginfo: np.ndarray
    cpool: np.ndarray
    gslot_param: np.ndarray   # int64 [nslots]: parameter index of each gradient slot
    gslot_factor: np.ndarray  # float64 [nslots]


def gate_has_param(g: GateRec) -> bool:
    return g.param is not None or (g.diag is not None and any(t.param is not None for t in g.diag))


def fold_rounds(pp: PassPlan, cfg: PlanConfig, xterms: Sequence[Tuple[int, float, int]],
                dterms: Optional[Sequence[Tuple[int, float]]] = None,
                pairs: Optional[Sequence[Tuple[int, int, float, int, int, int]]] = None) -> None:
    """Prepend to one pass of a reverse sweep the rounds in which terms of a Pauli-sum cotangent are BORN in registers:
    ``xterms`` = [(physical bit inside the pass's tile, coefficient c, fold index, kind)] stands for lambda += c P_bit psi
    (P = X for kind 0, Y for kind 1; and the energy c / 2 <P_bit>), ``dterms`` = [(Z mask over physical bits, coefficient)] for lambda += c Z...Z psi,
    ``pairs`` = [(physical bit a, physical bit b, coefficient, pair index, kind a, kind b)] for lambda += c P_a P_b psi (both
    bits inside the tile).  The tile's bits that carry an X / Y factor are cycled through the register bits by rounds of
    their own BEFORE any gate of the pass runs, starting in the load layout of the pass's first round (which also takes the
    diagonal terms: they need no particular layout); a pair needs BOTH its bits among the register bits of one round, so
    the groups of a chain of neighbouring pairs overlap by one bit.  The pass then continues with its own rounds.  Costs one
    exchange per extra round and 24 (pair: 24) packed instructions per term and thread; saves the Pauli-sum passes
    (tcmi_apply_pauli_sum_tiled) that would have produced those terms."""
    tb_of_phys = {p: i for i, p in enumerate(pp.tile_bits)}
    todo = {tb_of_phys[p]: (i, float(c), int(kd)) for (p, c, i, kd) in xterms}
    ptodo = [(tb_of_phys[pa], tb_of_phys[pb], float(c), int(pi), int(ka), int(kb)) for (pa, pb, c, pi, ka, kb) in (pairs or [])]
    if not (todo or dterms or ptodo) or not pp.rounds:
        return
    T, R = cfg.T, cfg.R
    first = pp.rounds[0]
    r0 = Round(list(first.reg_tb), list(first.thr_tb), [])

    def place(rd, reg):
        """Every single term and every pair whose bits are register bits of ``reg`` is folded in round ``rd``."""
        nonlocal ptodo
        jof = {b: j for j, b in enumerate(reg)}
        rd.fold = [(jof[b], todo[b][1], todo[b][0], todo[b][2]) for b in reg if b in todo]
        for b in reg:
            todo.pop(b, None)
        rd.fold2 = [(jof[a], jof[b], c, pi, ka, kb) for (a, b, c, pi, ka, kb) in ptodo if a in jof and b in jof]
        ptodo = [t for t in ptodo if not (t[0] in jof and t[1] in jof)]

    place(r0, r0.reg_tb)
    r0.dfold = [(int(zm), float(c)) for zm, c in (dterms or [])]
    pre = [r0]
    while todo or ptodo:
        # the next group of <= R tile bits: seeded by the pending pair with the lowest bits (else by the lowest single),
        # grown by whatever pending pair adds the fewest new bits, then by pending singles, then filled from the top
        grp: List[int] = []
        if ptodo:
            a, b = min((t[0], t[1]) for t in ptodo)
            grp = [a, b]
            while True:
                cand = [(len({t[0], t[1]} - set(grp)), min(t[0], t[1]), t) for t in ptodo if not {t[0], t[1]} <= set(grp)]
                cand = [c_ for c_ in cand if len(grp) + c_[0] <= R]
                if not cand:
                    break
                _, _, t = min(cand, key=lambda c_: (c_[0], c_[1]))
                grp += [x for x in (t[0], t[1]) if x not in grp]
        for b in sorted(todo):
            if len(grp) >= R:
                break
            if b not in grp:
                grp.append(b)
        fill = [b for b in range(T - 1, -1, -1) if b not in grp]
        reg = sorted(grp + fill[: R - len(grp)])
        thr = [b for b in range(T) if b not in reg]
        rd = Round(reg, thr, [])
        place(rd, reg)
        pre.append(rd)
    pp.rounds = pre + pp.rounds


def compile_adjoint_plan(gates: List[GateRec], n: int, cfg: PlanConfig, factorized: bool = False,
                         drop_constant_head: bool = False, fold: Optional[Sequence[Tuple[int, float]]] = None,
                         fold_param: int = 0, dfold: Optional[Sequence[Tuple[int, float]]] = None,
                         lam_zero: bool = False, fold2: Optional[Sequence[Tuple[int, int, float, int, int]]] = None) -> AdjointPlan:
    """Plan of the reversed circuit: gates in reverse order, each applied as U^dagger to both psi and
    the cotangent lambda, with one gradient slot per parametrised gate / diagonal term.  Valid for
    unitary gates (psi is un-computed, not stored).  ``factorized``: diagonal terms as OP_DIAGF (the
    packed-f32 kernel csrc/tcmi_adjoint2.hip) instead of the generic OP_DIAG.
    ``drop_constant_head``: the constant gates BEFORE the first parametrised gate of the circuit (the Hadamard layer
    of an ansatz) end the reverse sweep and feed no gradient slot -- they are left out; psi and lambda then stop at
    the state after those gates, so callers that want the input-state cotangent must not set it."""
    rev = list(reversed(gates))
    if drop_constant_head:
        last = max((i for i, g in enumerate(rev) if gate_has_param(g)), default=-1)
        rev = rev[: last + 1]
    passes = schedule(rev, n, cfg)
    folded: List[int] = []
    folded2: List[int] = []
    if fold or dfold or fold2:
        # ``fold``: single-X terms c_i X_i of the cotangent lambda = (sum_i c_i X_i + sum_t d_t Z..Z) psi, ``dfold`` its
        # Z-only strings: born in registers instead of arriving through memory (fold_rounds).  An X term is added by the
        # FIRST pass whose tile holds its bit -- nothing that ran before touches that qubit (a gate needs its qubit in
        # the tile, a diagonal term waits for the gates before it), so everything applied so far commutes with X_i, and so
        # do the generators of the gradient events taken so far: lambda_k + c_i X_i psi_k IS the propagated full cotangent
        # where it matters.  The diagonal strings are added at the very start of the sweep.  Energies c_i / 2 <X_i> and
        # 1/2 sum_t d_t <Z..Z> arrive in gradient slots mapped to "parameter" ``fold_param``.  ``lam_zero``: every term of
        # the cotangent is folded, lambda is never loaded by the first pass (FLAG_LAMBDA_ZERO).
        assert factorized and cfg.gen >= 2
        seen_bits, touched = set(), set()
        for k, pp in enumerate(passes):
            new_bits = [b for b in pp.tile_bits if b not in seen_bits]
            xs = [(f_[0], float(f_[1]), i, int(f_[2]) if len(f_) > 2 else 0) for i, f_ in enumerate(fold or [])
                  if f_[0] in new_bits and (n - 1 - f_[0]) not in touched and i not in folded]
            ds = list(dfold) if (dfold and k == 0) else None
            # ``fold2``: two-factor strings c P_a P_b (physical bits a, b; kinds 0 = X, 1 = Y): folded by the first pass whose
            # tile holds BOTH bits while neither qubit has been touched -- the same commutation argument for the pair.  A
            # pass that touches one of the two qubits while the other is outside its tile ends the pair's chances: it is
            # never folded and stays with the tile passes (the caller reads ``folded2``).
            tset = set(pp.tile_bits)
            ps = [(f_[0], f_[1], float(f_[2]), i, int(f_[3]), int(f_[4])) for i, f_ in enumerate(fold2 or [])
                  if i not in folded2 and f_[0] in tset and f_[1] in tset
                  and (n - 1 - f_[0]) not in touched and (n - 1 - f_[1]) not in touched]
            if xs or ds or ps:
                fold_rounds(pp, cfg, xs, ds, ps)
                folded += [i for _, _, i, _ in xs]
                folded2 += [t_[3] for t_ in ps]
            seen_bits |= set(pp.tile_bits)
            for gi in pp.gate_ids:
                touched |= set(rev[gi].qubits)
    sh2 = set()
    if cfg.shear2 and factorized:
        sh2 = shear2_gates(rev, [gi for pp in passes for rd in pp.rounds for gi in rd.gates], cfg)
    while True:
        tables = Tables()
        tables.ctab += [0.0] * 8
        tables.shear2 = set(sh2)
        tables.fold_slots = {i: tables.grad_slot(int(fold_param), 1.0) for i in folded}
        for i in folded2:
            tables.fold_slots[("pair", i)] = tables.grad_slot(int(fold_param), 1.0)
        if dfold:
            tables.fold_slots["diag"] = tables.grad_slot(int(fold_param), 1.0)
        descs, crossing = [], set()
        for i, pp in enumerate(passes):
            if i > 0 and any(getattr(rd, "fold", None) or getattr(rd, "fold2", None) for rd in pp.rounds):
                # A two-shear rotation leaves a REAL factor diag(c, 1/c) pending on psi (its reciprocal on lambda) until the
                # next phase table carries it.  Across the start of a pass that folds terms into lambda that would be wrong:
                # the fold adds c X psi_memory to lambda_memory, and the two differ from the true vectors by that factor and
                # its inverse.  Gates whose factor is still pending here take the three-shear form instead.
                crossing |= {t.gate for t in tables.pending if t.scale}
            descs.append(encode_pass(rev, n, cfg, pp, tables, backward=True, final=(i == len(passes) - 1),
                                     factorized_bw=factorized))
        if crossing:
            sh2 -= crossing
            continue
        if not tables.scale_only:
            break
        sh2 -= tables.scale_only   # as in compile_plan
    ap = AdjointPlan(
        n=n, cfg=cfg, passes=passes, descs=descs, ctab=np.array(tables.ctab, dtype=np.float64),
        ptab_size=tables.ptab_size, ginfo=np.array(tables.ginfo, dtype=np.int32).reshape(-1, 8),
        cpool=np.array(tables.cpool, dtype=np.float64),
        gslot_param=np.array(tables.gslot_param, dtype=np.int64),
        gslot_factor=np.array(tables.gslot_factor, dtype=np.float64),
    )
    if lam_zero:
        assert folded or dfold or folded2
        descs[0] = descs[0].copy()
        descs[0][6] = descs[0][6] | FLAG_LAMBDA_ZERO
        ap.descs = descs
    ap.folded = sorted(folded)
    ap.folded2 = sorted(folded2)
    ap.drop_constant_head = bool(drop_constant_head)
    return ap
