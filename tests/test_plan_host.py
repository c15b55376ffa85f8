"""Host logic on CPU: the plan compiler's pass descriptors are executed by the numpy emulator
(oracle/plan_emulator.py, test infrastructure) and compared with the dense oracle."""

import numpy as np
import pytest

import tcmi as tc
from tcmi import plan as P
from tcmi.executor import pick_variant, structure_digest
from oracle import cut as oracle_cut, dense, gates as G, plan_emulator as E, workloads as W


def _mixed(n, d, seed):
    rng = np.random.default_rng(seed)
    pb = rng.uniform(0, 2 * np.pi, [2 * d, n])
    pa = rng.uniform(0, 2 * np.pi, [d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, pb, zz=tc.gates._zz_matrix)
    c.ry(0, theta=0.3)
    c.u(1, theta=0.2, phi=0.5, lbd=0.7)
    c.cnot(2, 3)
    c.s(4)
    c.cz(5, 1)
    c.swap(0, 6)
    c.cnot(7, 2)
    W.hea_a(c, n, d, pa)
    c.cphase(3, 7, theta=0.4)
    c.rzz(0, 5, theta=0.9)
    c.iswap(1, 2, theta=0.3)
    c.phase(4, theta=1.1)
    c.any(1, 6, unitary=G.random_two_qubit_gate(5))
    ops = (
        W.hea_b_ops(n, d, pb)
        + [(G.ry(0.3), [0]), (G.u(0.2, 0.5, 0.7), [1]), (G.CNOT, [2, 3]), (G.S, [4]), (G.CZ, [5, 1]),
           (G.SWAP, [0, 6]), (G.CNOT, [7, 2])]
        + W.hea_a_ops(n, d, pa)
        + [(G.controlled(G.phase(0.4)), [3, 7]), (G.rzz(0.9), [0, 5]), (G.iswap(0.3), [1, 2]),
           (G.phase(1.1), [4]), (G.random_two_qubit_gate(5), [1, 6])]
    )
    return c, ops


def _op_histogram(pl):
    seen = {}
    for desc in pl.descs:
        w = np.asarray(desc).view(np.int32)
        pc = P.HDR_WORDS
        for _ in range(int(w[5])):
            nops, nw = int(w[pc]), int(w[pc + 1])
            q = pc + P.RR_WORDS
            for _o in range(nops):
                op = int(w[q])
                seen[op] = seen.get(op, 0) + 1
                if op == P.OP_DIAG:
                    q += 5 + int(w[q + 1]) + 2 * int(w[q + 2]) + int(w[q + 3])
                else:
                    q += {P.OP_G2: 4, P.OP_G1M: 3, P.OP_DIAGC: 2, P.OP_DIAGB: 4, P.OP_DIAGB2: 5, P.OP_DIAGCW: 6}[op]
            assert q == pc + P.RR_WORDS + nw
            pc = q
    return seen


def test_generic_diag_form_still_covered():
    """Many register-x-thread terms in one diagonal layer keep the per-thread sincos op (OP_DIAG)."""
    n = 12
    c = tc.Circuit(n)
    ops = []
    for i in range(n):
        c.h(i)
        ops.append((G.H, [i]))
    rng = np.random.default_rng(0)
    for i in range(n):
        for j in range(i + 1, n):
            th = float(rng.uniform(0, 2 * np.pi))
            c.rzz(i, j, theta=th)
            ops.append((G.rzz(th), [i, j]))
    for i in range(n):
        c.rx(i, theta=0.1 * (i + 1))
        ops.append((G.rx(0.1 * (i + 1)), [i]))
    cfg = P.PlanConfig(R=3, LT=8, lowbits=5, vec=2)
    pl = P.compile_plan(c._gate_records(), n, cfg, nparams=len(c._params))
    assert _op_histogram(pl).get(P.OP_DIAG, 0) > 0
    psi = E.run_plan(pl, np.array([float(x) for x in c._params]))
    np.testing.assert_allclose(psi, dense.run(n, ops), atol=1e-12)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
@pytest.mark.parametrize("n,d", [(8, 2), (10, 3), (13, 2), (14, 2)])
def test_plan_emulated_matches_dense(dtype, n, d):
    c, ops = _mixed(n, d, seed=n)
    ne, cfg = pick_variant(n, dtype)
    assert ne == n
    pl = P.compile_plan(c._gate_records(), n, cfg, nparams=len(c._params))
    psi = E.run_plan(pl, np.array([float(x) for x in c._params]))
    np.testing.assert_allclose(psi, dense.run(n, ops), atol=1e-12)
    # every LDS exchange the planner emits is bank-conflict free: 8-byte elements for the first-generation kernels,
    # 4-byte planes (32-lane groups on 32 banks, reads and writes) for the packed ones
    if dtype == "complex64":
        for desc in pl.descs:
            assert all(w == 1 and r == 1 for w, r in E.lds_conflicts(desc, planar=cfg.gen >= 2))
    # ... also for the wider tiles of the packed kernels and for adjoint / measurement plans
    if dtype == "complex64" and n >= 13:
        for cfg2 in (P.PlanConfig(R=5, LT=8, lowbits=5, vec=2, gen=2), P.PlanConfig(R=4, LT=8, lowbits=5, vec=2, gen=2)):
            pl2 = P.compile_plan(c._gate_records(), n, cfg2, nparams=len(c._params))
            np.testing.assert_allclose(E.run_plan(pl2, np.array([float(x) for x in c._params])), dense.run(n, ops), atol=1e-12)
            for desc in pl2.descs:
                assert all(w == 1 and r == 1 for w, r in E.lds_conflicts(desc, planar=True))


@pytest.mark.parametrize("lowbits,R,LT", [(3, 2, 6), (5, 4, 8), (7, 5, 8), (5, 5, 9), (4, 3, 8)])
def test_plan_variants(lowbits, R, LT):
    n, d = max(R + LT, 12), 2
    rng = np.random.default_rng(1)
    pb = rng.uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, pb, zz=tc.gates._zz_matrix)
    c.cnot(0, n - 1)
    c.any(n - 1, 1, unitary=G.random_two_qubit_gate(3))
    cfg = P.PlanConfig(R=R, LT=LT, lowbits=lowbits, vec=2)
    pl = P.compile_plan(c._gate_records(), n, cfg, nparams=len(c._params))
    psi = E.run_plan(pl, np.array([float(x) for x in c._params]))
    ref = dense.run(n, W.hea_b_ops(n, d, pb) + [(G.CNOT, [0, n - 1]), (G.random_two_qubit_gate(3), [n - 1, 1])])
    np.testing.assert_allclose(psi, ref, atol=1e-12)
    # the diagonal layers are lowered to builder-evaluated phase tables (DIAGC: register bits only,
    # DIAGB: one register bit x thread bits); the per-thread sincos form is the fallback
    seen = _op_histogram(pl)
    assert seen.get(P.OP_DIAGC, 0) > 0 and seen.get(P.OP_DIAGB, 0) > 0
    kinds = {int(r[0]) for r in np.asarray(pl.ginfo).reshape(-1, 8)}
    assert P.BK_PHASE in kinds
    for pp in pl.passes:
        assert pp.tile_bits[: min(lowbits, cfg.T)] == list(range(min(lowbits, cfg.T)))  # coalescing run
        assert pp.rounds[0].reg_tb[0] == 0 and pp.rounds[-1].reg_tb[0] == 0          # 16-byte accesses


@pytest.mark.parametrize("scale", [0.3, 3.0])
def test_two_shear_rotations_leave_a_pending_real_factor(scale):
    """gen-2 plans apply a rotation that a diagonal gate follows as two shears; its factor diag(c, 1/c) rides on the
    phase table of the next flush (plan.shear2_gates).  Small angles: every eligible gate takes the form; large
    angles: the builder falls back to three shears wherever |cos| < SHEAR2_CMIN.  Both match the dense oracle."""
    n, d = 14, 3
    rng = np.random.default_rng(5)
    pb = rng.normal(0, scale, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, pb, zz=tc.gates._zz_matrix)
    c.ry(3, theta=0.4)
    c.rzz(3, 9, theta=0.2)
    c.ry(3, theta=-0.7)
    c.rx(9, theta=2.9)
    ops = W.hea_b_ops(n, d, pb) + [(G.ry(0.4), [3]), (G.rzz(0.2), [3, 9]), (G.ry(-0.7), [3]), (G.rx(2.9), [9])]
    th = np.array([float(x) for x in c._params])
    ref = dense.run(n, ops)
    for R in (4, 5):
        cfg = P.PlanConfig(R=R, LT=8, lowbits=5, vec=2, gen=2)
        pl = P.compile_plan(c._gate_records(), n, cfg, nparams=len(c._params))
        recs = np.asarray(pl.ginfo).reshape(-1, 8)
        allowed = recs[(recs[:, 0] == P.BK_TRIG) & (recs[:, 5] != 0) & (recs[:, 6] != 0)]
        assert len(allowed) >= (d - 1) * n // 2   # the rx layers but the last (minus those whose factor found no table), the ry before the rzz
        ptab = E.build_table(pl.ginfo, pl.cpool, th[None], pl.ptab_size)[0]
        taken = sum(ptab[int(r[1]) + 3] != 0 for r in allowed)
        assert (taken == len(allowed)) if scale < 1 else (0 < taken < len(allowed))
        np.testing.assert_allclose(E.run_plan(pl, th), ref, atol=1e-12)
        off = P.compile_plan(c._gate_records(), n, P.PlanConfig(R=R, LT=8, lowbits=5, vec=2, gen=2, shear2=False), nparams=len(c._params))
        assert not np.asarray(off.ginfo).reshape(-1, 8)[:, 6].any()
    # first-generation plans (complex128 kernels) never use it
    pl1 = P.compile_plan(c._gate_records(), n, P.PlanConfig(R=4, LT=8, lowbits=5, vec=1), nparams=len(c._params))
    assert not np.asarray(pl1.ginfo).reshape(-1, 8)[:, 6].any()


@pytest.mark.parametrize("scale", [0.3, 3.0])
def test_two_shear_adjoint_sweep(scale):
    """The reverse sweep of the packed kernels with rotations in two-shear form: psi carries the pending factor
    diag(c, 1/c), lambda its reciprocal (other shear order, second table in OP_DIAGF), so gradients are untouched.
    Checked against central differences of the forward plan; the un-computed psi and the input-state cotangent
    U^dagger g come back with the right sign (the sign pulled out of three-shear gates is applied at the store)."""
    n, d = 13, 3
    rng = np.random.default_rng(3)
    pb = rng.normal(0, scale, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, pb, zz=tc.gates._zz_matrix)
    c.ry(3, theta=0.4); c.rzz(3, 9, theta=0.2); c.ry(3, theta=-0.7); c.rx(9, theta=2.9)
    ops = W.hea_b_ops(n, d, pb) + [(G.ry(0.4), [3]), (G.rzz(0.2), [3, 9]), (G.ry(-0.7), [3]), (G.rx(2.9), [9])]
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    g = rng.normal(size=2 ** n) + 1j * rng.normal(size=2 ** n)
    lam_ref = g
    for m, qs in reversed(ops):
        lam_ref = dense.apply_gate(lam_ref, n, np.asarray(m).conj().T, qs)
    e0 = np.zeros(2 ** n)
    e0[0] = 1
    for R in (4, 5):
        cfg = P.PlanConfig(R=R, LT=8, lowbits=5, vec=2, gen=2)
        pl = P.compile_plan(recs, n, cfg, nparams=len(vals))
        psi = E.run_plan(pl, vals)
        for drop in (False, True):
            ap = P.compile_adjoint_plan(recs, n, cfg, factorized=True, drop_constant_head=drop)
            ra = np.asarray(ap.ginfo).reshape(-1, 8)
            assert ((ra[:, 0] == E.BK_UDAG) & (ra[:, 6] != 0)).sum() >= 10      # two-shear candidates
            assert ((ra[:, 0] == P.BK_PHASE) & (ra[:, 6] != 0)).any()            # lambda tables
            grad, psi_in, lam_in = E.run_adjoint_plan(ap, vals, psi, g, len(vals), return_lambda=True)
            if not drop:
                np.testing.assert_allclose(psi_in, e0, atol=1e-12)
                np.testing.assert_allclose(lam_in, lam_ref, atol=1e-11)
            for i in rng.choice(len(vals), 5, replace=False):
                vp, vm = vals.copy(), vals.copy()
                vp[i] += 1e-6
                vm[i] -= 1e-6
                fd = (np.real(np.vdot(g, E.run_plan(pl, vp))) - np.real(np.vdot(g, E.run_plan(pl, vm)))) / 2e-6
                assert abs(grad[i] - fd) < 2e-8, (R, drop, i, grad[i], fd)
        ap0 = P.compile_adjoint_plan(recs, n, P.PlanConfig(R=R, LT=8, lowbits=5, vec=2, gen=2, shear2=False), factorized=True)
        assert not np.asarray(ap0.ginfo).reshape(-1, 8)[:, 6].any()


def test_batched_table_builder():
    n, d = 10, 2
    rng = np.random.default_rng(2)
    pbs = rng.uniform(0, 2 * np.pi, [3, 2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, pbs[0], zz=tc.gates._zz_matrix)
    _, cfg = pick_variant(n, "complex64")
    pl = P.compile_plan(c._gate_records(), n, cfg, nparams=len(c._params))
    order = []
    for j in range(d):
        order += [(2 * j) * n + i for i in range(n - 1)] + [(2 * j + 1) * n + i for i in range(n)]
    pmat = pbs.reshape(3, -1)[:, order]
    for b in range(3):
        psi = E.run_plan(pl, pmat, batch_index=b)
        np.testing.assert_allclose(psi, dense.run(n, W.hea_b_ops(n, d, pbs[b])), atol=1e-12)


def test_diag_analysis():
    """rz / phase / exp1(ZZ) / cz / cphase are recognised as phase polynomials; rx / h are not."""
    pr = P.ParamRef(0, 1.0, 0.0)
    s = tc.gates.rz_spec(0.0)[0]
    t = P.diag_terms_trig(s.c0, s.c1, s.c2, (3,), P.ParamRef(0, s.scale, s.offset))
    assert len(t) == 1 and t[0].qubits == (3,) and abs(t[0].param.scale + 0.5) < 1e-15
    s = tc.gates.exp1_spec(tc.gates._zz_matrix, 0.0)[0]
    t = P.diag_terms_trig(s.c0, s.c1, s.c2, (1, 2), P.ParamRef(0, s.scale, s.offset))
    assert len(t) == 1 and t[0].qubits == (1, 2) and abs(t[0].param.scale + 1.0) < 1e-15
    s = tc.gates.phase_spec(0.0)[0]
    t = P.diag_terms_trig(s.c0, s.c1, s.c2, (0,), P.ParamRef(0, s.scale, s.offset))
    assert {x.qubits for x in t} == {(), (0,)}
    assert P.diag_terms_const(G.CZ, (0, 1)) is not None
    assert P.diag_terms_const(G.H, (0,)) is None
    s = tc.gates.rx_spec(0.0)[0]
    assert P.diag_terms_trig(s.c0, s.c1, s.c2, (0,), pr) is None
    # walsh coefficients reproduce the phases
    ph = np.array([0.1, -0.4, 0.9, 2.0])
    terms = P.walsh_terms(ph, (5, 7))
    for x in range(4):
        bits = {5: (x >> 1) & 1, 7: x & 1}
        tot = sum(c * (-1) ** sum(bits[q] for q in sub) for sub, c in terms)
        assert abs(tot - ph[x]) < 1e-14


def test_gate_kinds():
    recs = tc.Circuit(2)
    c = tc.Circuit(3)
    c.h(0); c.rx(1, theta=0.1); c.ry(2, theta=0.2); c.t(0); c.cnot(0, 1); c.cnot(2, 1); c.swap(0, 2)
    r = c._gate_records()
    assert P.g1_kind(r[0]) == 1 and P.g1_kind(r[1]) == 2 and P.g1_kind(r[2]) == 1
    assert r[3].is_diag
    assert P.g2_kind(r[4], False) == 1 and P.g2_kind(r[4], True) == 2 and P.g2_kind(r[6], False) == 3


def test_structure_cache_key():
    def build(theta):
        c = tc.Circuit(4)
        c.h(0); c.rx(1, theta=theta); c.cnot(0, 1)
        return c
    a, b = build(0.1), build(0.7)
    assert structure_digest(4, "complex64", a._gate_records()) == structure_digest(4, "complex64", b._gate_records())
    c3 = build(0.1); c3.x(2)
    assert structure_digest(4, "complex64", a._gate_records()) != structure_digest(4, "complex64", c3._gate_records())


def test_unsupported_gates_raise():
    """Dense gates on 3-8 qubits are synthesised (tcmi/synth.py); what stays unsupported fails loudly."""
    c = tc.Circuit(14)
    c.toffoli(0, 1, 2)
    _, cfg = pick_variant(14, "complex64")
    P.compile_plan(c._gate_records(), 14, cfg)
    with pytest.raises(NotImplementedError):
        c.any(0, 1, 2, unitary=np.arange(64).reshape((2,) * 6))          # non-unitary 3-qubit tensor
    c6 = tc.Circuit(14)
    c6.any(0, 1, 2, 3, 4, 5, unitary=np.eye(64)[::-1].reshape((2,) * 12))     # dense on 6 qubits: exact synthesis
    P.compile_plan(c6._gate_records(), 14, cfg)
    with pytest.raises(NotImplementedError):
        c.any(*range(9), unitary=np.eye(512))                                  # dense on 9 qubits
    raw = P.GateRec((0, 1, 2), c0=np.eye(8)[::-1].copy(), name="raw3")
    with pytest.raises(NotImplementedError):
        P.compile_plan([raw], 14, cfg)


def test_small_circuits_are_padded():
    ne, cfg = pick_variant(3, "complex64")
    assert ne == 8 and cfg.T == 8
    ne, cfg = pick_variant(28, "complex64")
    assert ne == 28 and (cfg.R, cfg.LT) == (5, 8)
    ne, cfg = pick_variant(28, "complex128")
    assert (cfg.R, cfg.LT) == (4, 8) and cfg.vec == 1


# ---- measurement and adjoint plans (emulated) -------------------------------------------------------
@pytest.mark.parametrize("n,R,LT", [(8, 2, 6), (12, 4, 8), (14, 4, 8), (13, 3, 8), (14, 5, 8)])
def test_measure_plan_emulated(n, R, LT):
    rng = np.random.default_rng(n)
    psi = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    psi /= np.linalg.norm(psi)
    strings = [ps for _, ps in W.tfim_terms(n)]
    for k in range(10):
        ps = [0] * n
        qs = rng.choice(n, 3, replace=False)
        ps[qs[0]] = int(rng.integers(1, 3)); ps[qs[1]] = int(rng.integers(1, 4)); ps[qs[2]] = 3
        strings.append(ps)
    strings.append([0] * n)  # identity
    terms = [P.pauli_term_from_string(ps) for ps in strings]
    mp = P.compile_measure_plan(terms, n, P.PlanConfig(R=R, LT=LT, lowbits=5, vec=2))
    got = E.run_measure_plan(mp, psi)
    ref = np.array([dense.pauli_string_expectation(psi, n, ps) for ps in strings])
    np.testing.assert_allclose(got, ref, atol=1e-12)
    # the second-generation descriptors (TCMI_OP_EXPECT2: Z-only strings grouped by register mask), both tile widths
    for lb in (4, 5):
        mp2 = P.compile_measure_plan(terms, n, P.PlanConfig(R=R, LT=LT, lowbits=min(lb, R + LT), vec=2, gen=2))
        ops = [int(w) for d in mp2.descs for w in np.asarray(d).view(np.int32)]
        assert P.OP_EXPECT2 in ops
        np.testing.assert_allclose(E.run_measure_plan(mp2, psi), ref, atol=1e-12)
    with pytest.raises(NotImplementedError):
        P.compile_measure_plan([P.pauli_term_from_string([1, 1, 1] + [0] * (n - 3))], n, P.PlanConfig(R=R, LT=LT))


@pytest.mark.parametrize("n,R,LT", [(8, 2, 6), (12, 4, 8), (13, 3, 8)])
def test_adjoint_plan_emulated(n, R, LT):
    """Adjoint sweep == central finite differences of the emulated forward plan; psi is un-computed
    back to |0..0>; shared parameters accumulate."""
    rng = np.random.default_rng(n)

    def build(p):
        c = tc.Circuit(n)
        k = 0
        for i in range(n):
            c.h(i)
        for i in range(n - 1):
            c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[k]); k += 1
        for i in range(n):
            c.rx(i, theta=p[k]); k += 1
        c.cnot(0, n - 1); c.ry(1, theta=p[k]); k += 1
        c.rzz(0, n - 2, theta=p[k]); k += 1
        c.phase(2, theta=p[k]); k += 1
        c.crx(n - 1, 1, theta=p[k]); k += 1
        c.rxx(2, 0, theta=p[k]); k += 1
        c.u(3, theta=p[k], phi=p[k + 1], lbd=0.3); k += 2
        c.swap(1, 4); c.rz(1, theta=p[0])  # p[0] is shared with the first exp1
        return c, k

    _, npar = build(np.zeros(200))
    c, _ = build(rng.uniform(0, 2 * np.pi, npar))
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    cfg = P.PlanConfig(R=R, LT=LT, lowbits=min(5, R + LT), vec=2)
    pl = P.compile_plan(recs, n, cfg, nparams=len(vals))
    psi = E.run_plan(pl, vals)
    g = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    ap = P.compile_adjoint_plan(recs, n, cfg)
    grad, psi_in = E.run_adjoint_plan(ap, vals, psi, g, len(vals))
    e0 = np.zeros(2**n); e0[0] = 1
    np.testing.assert_allclose(psi_in, e0, atol=1e-12)
    eps = 1e-6
    picks = rng.choice(len(vals), 8, replace=False)
    for i in picks:
        vp, vm = vals.copy(), vals.copy()
        vp[i] += eps; vm[i] -= eps
        fd = (np.real(np.vdot(g, E.run_plan(pl, vp))) - np.real(np.vdot(g, E.run_plan(pl, vm)))) / (2 * eps)
        assert abs(grad[i] - fd) < 1e-7


def test_cut_spec_and_selector_gates():
    """Cut contraction spec (tcmi/cut.py): the cut formula reproduces the dense oracle, and the
    half-circuits with selector gates (BK_SELECT) run through the emulated tile-VM."""
    from tcmi import cut

    n, d = 10, 2
    params = np.random.default_rng(0).uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    c.cnot(4, 5); c.cz(5, 4); c.any(4, 5, unitary=G.random_two_qubit_gate(1)); c.rzz(5, 4, theta=0.7)
    recs = c._gate_records()
    pv = np.array([float(x) for x in c._params])
    spec = cut.make_cut(recs, n, 5, len(pv))
    assert [len(b.terms) for b in spec.bonds] == [2, 2, 2, 2, 4, 2] and spec.bond_dim == 128
    ops = W.hea_b_ops(n, d, params) + [(G.CNOT, [4, 5]), (G.CZ, [5, 4]), (G.random_two_qubit_gate(1), [4, 5]), (G.rzz(0.7), [5, 4])]
    np.testing.assert_allclose(oracle_cut.reference_state(spec, pv), dense.run(n, ops), atol=1e-12)
    # one bond configuration of the left half through the plan compiler + emulator
    nb = len(spec.bonds)
    digits = np.array([1, 0, 1, 1, 3, 0], dtype=np.float64)
    pvec = np.concatenate([pv, digits])
    left8 = [P.GateRec(tuple(q + 3 for q in g.qubits), g.c0, g.c1, g.c2, g.param,
                       None if g.diag is None else [P.DiagTerm(tuple(q + 3 for q in t.qubits), t.const, t.param) for t in g.diag],
                       g.name, g.select) for g in spec.left]       # pad 5 -> 8 qubits
    pl = P.compile_plan(left8, 8, P.PlanConfig(R=2, LT=6, lowbits=5, vec=2), nparams=len(pvec))
    got = E.run_plan(pl, pvec)[: 2**5]
    want = np.zeros(2**5, dtype=np.complex128); want[0] = 1
    for g in spec.left:
        want = dense.apply_gate(want, 5, g.matrix(pvec), list(g.qubits))
    np.testing.assert_allclose(got, want, atol=1e-12)
    assert cut.make_cut([P.GateRec((0, 5, 9), c0=np.eye(8))], n, 5, 0) is None


def test_cut_half_bounds_dominate_every_half_state():
    """``cut.half_bounds``: what the two-piece f16 join takes its operand scales from.  Unitary gates count 1 exactly (HEA-B
    with ZZ crossings: both bounds 1), the operator-Schmidt factors of a generic crossing gate their largest norm; every
    half-circuit state of every bond configuration, the right one times its bond weight, stays under the bound; a
    non-unitary gate raises it, a parametrised non-unitary family is bounded by its coefficient norms."""
    import itertools

    from tcmi import cut

    n, d = 10, 2
    params = np.random.default_rng(3).uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    spec = cut.make_cut(c._gate_records(), n, 5, len(c._params), defer=True)
    assert cut.half_bounds(spec) == (1.0, 1.0)
    c.cnot(4, 5); c.any(4, 5, unitary=G.random_two_qubit_gate(1)); c.rzz(5, 4, theta=0.7); c.h(2)
    recs = c._gate_records()
    pv = np.array([float(x) for x in c._params])
    spec = cut.make_cut(recs, n, 5, len(pv))
    bl, br = cut.half_bounds(spec)
    assert 0.25 < bl < 4.0 and 0.25 < br < 4.0
    worst_l = worst_r = 0.0
    for digits in itertools.product(*[range(len(b.terms)) for b in spec.bonds]):
        pvec = np.concatenate([pv, np.array(digits, dtype=np.float64)])
        w = 1.0
        for b, j in zip(spec.bonds, digits):
            kind, ref = b.terms[j][2]
            a = 0.0 if kind == "const" else ref.scale * pv[ref.index] + ref.offset
            w *= complex(ref) if kind == "const" else (np.cos(a) if kind == "cos" else np.sin(a))
        for gates, nq, which in ((spec.left, 5, 0), (spec.right, n - 5, 1)):
            psi = np.zeros(2**nq, dtype=np.complex128); psi[0] = 1
            for g in gates:
                psi = dense.apply_gate(psi, nq, g.matrix(pvec), list(g.qubits))
            if which:
                worst_r = max(worst_r, np.abs(psi).max() * abs(w))
            else:
                worst_l = max(worst_l, np.abs(psi).max())
    assert worst_l <= bl * (1 + 1e-12) and worst_r <= br * (1 + 1e-12)
    assert worst_l > 0.05 * bl and worst_r > 0.01 * br          # and the bounds are not vacuous
    assert cut.gate_norm_bound(P.GateRec((0,), c0=np.diag([2.0, 0.5]))) == 2.0
    fam = P.GateRec((0,), c0=np.zeros((2, 2)), c1=np.eye(2), c2=np.diag([0.0, 3.0]), param=P.ParamRef(0, 1.0, 0.0))
    assert cut.gate_norm_bound(fam) == 4.0
    assert cut.gate_norm_bound(recs[0]) == 1.0


def test_cut_with_the_last_crossing_gate_deferred():
    """``make_cut(defer=True)``: the last crossing gate and the one-qubit tail on its qubits leave the halves (one bond
    less) and come back as a 4 x 4 on the joined state; the formula still reproduces ``oracle.dense``.  Circuits whose
    tail does not commute keep every bond."""
    from tcmi import cut

    n, d = 10, 3
    params = np.random.default_rng(1).uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    recs = c._gate_records()
    pv = np.array([float(x) for x in c._params])
    plain = cut.make_cut(recs, n, 5, len(pv))
    spec = cut.make_cut(recs, n, 5, len(pv), defer=True)
    assert plain.bond_dim == 8 and plain.epilogue is None and not plain.right_rot
    assert spec.bond_dim == 4 and spec.right_rot and (spec.epilogue.ql, spec.epilogue.qr) == (4, 5)
    assert len(spec.epilogue.factors) == 3                    # exp1(ZZ) on (4, 5), rx(4), rx(5)
    assert len(spec.left) + len(spec.right) == len(plain.left) + len(plain.right) - 4
    want = dense.run(n, W.hea_b_ops(n, d, params))
    np.testing.assert_allclose(oracle_cut.reference_state(spec, pv), want, atol=1e-12)
    np.testing.assert_allclose(oracle_cut.reference_state(plain, pv), want, atol=1e-12)
    x = spec.epilogue.matrix(pv)
    np.testing.assert_allclose(x.conj().T @ x, np.eye(4), atol=1e-12)
    # tails: rz after the ladder is absorbed as well; a cnot on (5, 6) after the last crossing ZZ does not commute with it
    c2 = tc.Circuit(n)
    W.hea_b(c2, n, 2, params[:4], zz=tc.gates._zz_matrix)
    c2.rz(5, theta=0.3); c2.ry(4, theta=1.1); c2.rzz(5, 6, theta=0.2)       # rzz after rx(5): X is dirty on qubit 5
    assert cut.find_deferred(c2._gate_records(), 5) is None
    c3 = tc.Circuit(n)
    W.hea_b(c3, n, 2, params[:4], zz=tc.gates._zz_matrix)
    c3.rzz(4, 5, theta=0.4); c3.rzz(5, 6, theta=0.2); c3.cz(3, 4); c3.rz(5, theta=0.3); c3.ry(4, theta=1.1); c3.h(7)
    r3 = c3._gate_records()
    s3 = cut.make_cut(r3, n, 5, len(c3._params), defer=True)
    assert s3.bond_dim == 4 and len(s3.epilogue.factors) == 3
    ops3 = W.hea_b_ops(n, 2, params[:4]) + [(G.rzz(0.4), [4, 5]), (G.rzz(0.2), [5, 6]), (G.CZ, [3, 4]), (G.rz(0.3), [5]),
                                            (G.ry(1.1), [4]), (G.H, [7])]
    np.testing.assert_allclose(oracle_cut.reference_state(s3, np.array([float(v) for v in c3._params])),
                               dense.run(n, ops3), atol=1e-12)
    c4 = tc.Circuit(n)
    W.hea_b(c4, n, 1, params[:2], zz=tc.gates._zz_matrix)
    c4.cnot(4, 5); c4.cnot(5, 6)
    assert cut.find_deferred(c4._gate_records(), 5) is None
    assert cut.make_cut(c4._gate_records(), n, 5, len(c4._params), defer=True).epilogue is None


def test_deferral_rules_on_random_circuits():
    """``find_deferred`` decides from commutation rules which gates may leave the half-circuits; whatever it decides on
    random mixes of diagonal and dense one- and two-qubit gates, the cut formula with the 4 x 4 tail applied afterwards has
    to reproduce ``oracle.dense``."""
    from tcmi import cut

    n, nl = 8, 4
    rng = np.random.default_rng(7)
    seen = {"tail4x4": 0, "none": 0}
    for trial in range(60):
        c = tc.Circuit(n)
        ops = []
        for i in range(n):
            c.h(i); ops.append((G.H, [i]))
        for _ in range(int(rng.integers(10, 28))):
            kind = int(rng.integers(0, 7))
            q = int(rng.integers(0, n - 1))
            th = float(rng.uniform(0, 2 * np.pi))
            if kind == 0:
                c.rzz(q, q + 1, theta=th); ops.append((G.rzz(th), [q, q + 1]))
            elif kind == 1:
                c.cz(q, q + 1); ops.append((G.CZ, [q, q + 1]))
            elif kind == 2:
                c.rx(q, theta=th); ops.append((G.rx(th), [q]))
            elif kind == 3:
                c.ry(q, theta=th); ops.append((G.ry(th), [q]))
            elif kind == 4:
                c.rz(q, theta=th); ops.append((G.rz(th), [q]))
            elif kind == 5:
                c.cnot(q, q + 1); ops.append((G.CNOT, [q, q + 1]))
            else:
                c.t(q); ops.append((G.T, [q]))
        # make sure something crosses the cut near the end now and then
        if trial % 2:
            th = float(rng.uniform(0, 2 * np.pi))
            c.rzz(nl - 1, nl, theta=th); ops.append((G.rzz(th), [nl - 1, nl]))
            for q in (nl - 2, nl - 1, nl, nl + 1):
                if rng.integers(0, 2):
                    th = float(rng.uniform(0, 2 * np.pi))
                    c.rx(q, theta=th); ops.append((G.rx(th), [q]))
        if trial % 3 == 0:       # two closing layers of a ladder around the cut: a tail of two crossing gates
            for layer in range(2):
                for q in ((nl - 1,) if layer == 0 else (nl - 2, nl - 1, nl)):
                    th = float(rng.uniform(0, 2 * np.pi))
                    c.rzz(q, q + 1, theta=th); ops.append((G.rzz(th), [q, q + 1]))
                for q in ((nl - 1, nl) if layer == 0 else (nl - 2, nl - 1, nl, nl + 1)):
                    if layer == 0 or rng.integers(0, 3):
                        th = float(rng.uniform(0, 2 * np.pi))
                        g1 = (c.rx, G.rx) if rng.integers(0, 2) else (c.ry, G.ry)
                        g1[0](q, theta=th); ops.append((g1[1](th), [q]))
        recs = c._gate_records()
        pv = np.array([float(x) for x in c._params])
        want = dense.run(n, ops)
        for defer in (0, 1):
            spec = cut.make_cut(recs, n, nl, len(pv), defer=defer)
            if spec is None:
                continue
            np.testing.assert_allclose(oracle_cut.reference_state(spec, pv), want, atol=1e-11)
            if spec.epilogue is None:
                seen["none"] += 1
            else:
                seen["tail4x4"] += 1
                x = spec.epilogue.matrix(pv)
                np.testing.assert_allclose(x.conj().T @ x, np.eye(4), atol=1e-11)
    assert seen["tail4x4"] >= 5 and seen["none"] >= 5, seen


def test_dense_three_qubit_gates_are_synthesised_exactly():
    """toffoli / fredkin / any(3 qubits) (reference gates.py sgates, basecircuit.py:183-371) are rewritten
    on the host into <= 2-qubit dense + diagonal gates (tcmi/synth.py); the compiled plan run through
    the numpy emulator of the descriptor format equals the dense oracle."""
    import tcmi as tc
    from scipy.stats import unitary_group
    from tcmi import plan as P, synth
    from oracle import dense, gates as OG, plan_emulator as E

    for u in (OG.TOFFOLI, OG.FREDKIN, unitary_group.rvs(8, random_state=3), unitary_group.rvs(16, random_state=4)):
        k = int(np.log2(u.shape[0]))
        qs = list(range(5, 5 + k))
        assert np.abs(synth.expand(synth.decompose_dense(u, qs), qs) - u).max() < 1e-13
        low = synth.lower(synth.decompose_dense(u, qs))
        assert all(len(q) <= 2 for _, q in low)
        assert np.abs(synth.expand(low, qs) - u).max() < 1e-12
    n = 8
    u3 = unitary_group.rvs(8, random_state=11)
    c = tc.Circuit(n)
    ops = []
    for i in range(n):
        c.h(i)
        ops.append((OG.H, [i]))
    c.rx(1, theta=0.4)
    ops.append((OG.rx(0.4), [1]))
    c.toffoli(0, 3, 6)
    ops.append((OG.TOFFOLI, [0, 3, 6]))
    c.fredkin(7, 2, 4)
    ops.append((OG.FREDKIN, [7, 2, 4]))
    c.any(5, 1, 3, unitary=u3.reshape((2,) * 6))
    ops.append((u3, [5, 1, 3]))
    c.cz(0, 7)
    ops.append((OG.CZ, [0, 7]))
    recs = c._gate_records()
    assert all(r.is_diag or len(r.qubits) <= 2 for r in recs)
    cfg = P.PlanConfig(R=2, LT=6, lowbits=3, vec=1)
    plan = P.compile_plan(recs, n, cfg, nparams=len(c._params))
    out = E.run_plan(plan, np.array([float(p) for p in c._params]))
    np.testing.assert_allclose(out, dense.run(n, ops), atol=1e-12)


def test_pauli_sum_tile_passes_cover_every_term_once():
    """Host planner of the tiled (sum_t w_t P_t)|psi> (executor.plan_pauli_passes): every term lands in exactly one
    pass, its X mask inside that pass's tile bits; the low bits are in every tile; the TFIM of BASELINE config 3 needs
    three passes; masks that cannot share a tile, or sums the flat kernel handles with less traffic, return None."""
    from tcmi.executor import plan_pauli_passes

    n, T = 28, 12
    rows = [(1 << (n - 1 - q), 0, 0, q) for q in range(n)]
    rows += [(0, (1 << (n - 1 - q)) | (1 << (n - 2 - q)), 0, n + q) for q in range(n - 1)]
    passes = plan_pauli_passes(n, rows, T)
    assert passes is not None and len(passes) == 3
    seen = []
    for ps in passes:
        tp = ps["tilepos"]
        assert len(tp) == T and tp == sorted(tp) and tp[:4] == [0, 1, 2, 3]
        for (xl, zm, ny, xpar), k in zip(ps["rows"], ps["order"]):
            xm, zm0, ny0, _ = rows[k]
            back = sum(1 << tp[j] for j in range(T) if (xl >> j) & 1)
            assert back == xm and zm == zm0 and (ny & 3) == ny0 and xpar == (bin(xm & zm).count("1") & 1)
            ebits = [tp[0]] + tp[9:]
            assert (ny >> 8) == sum(((zm >> b) & 1) << j for j, b in enumerate(ebits))
            seen.append(k)
        nd = ps["ndiag"]
        assert all(r[0] == 0 for r in ps["rows"][:nd]) and all(r[0] != 0 for r in ps["rows"][nd:])
        assert [r[2] >> 8 for r in ps["rows"][:nd]] == sorted(r[2] >> 8 for r in ps["rows"][:nd])
        assert [r[0] for r in ps["rows"][nd:]] == sorted(r[0] for r in ps["rows"][nd:])   # sorted by X mask
    assert sorted(seen) == list(range(len(rows)))
    assert all(k in passes[0]["order"] for k in range(n, 2 * n - 1))              # diagonal terms: first pass
    # random few-body strings on 20 qubits
    rng = np.random.default_rng(3)
    rows = []
    for k in range(40):
        qs = rng.choice(20, size=int(rng.integers(1, 4)), replace=False)
        xm = sum(1 << int(q) for q in qs[: int(rng.integers(0, len(qs) + 1))])
        zm = sum(1 << int(q) for q in qs)
        rows.append((xm, zm, int(rng.integers(0, 3)), k))
    passes = plan_pauli_passes(20, rows, 12)
    if passes is not None:
        assert sorted(k for ps in passes for k in ps["order"]) == list(range(40))
    # a mask with more high bits than a tile has free bits cannot be tiled
    assert plan_pauli_passes(28, [(sum(1 << b for b in range(8, 20)), 0, 0, 0)], 12) is None
    # small state: the flat kernel's window already holds nearly everything
    assert plan_pauli_passes(14, [(1 << (13 - q), 0, 0, q) for q in range(14)], 12) is None


def test_mpo_and_diagonal_gate_formats_through_the_plan_emulator():
    """``apply_general_gate(..., mpo=True / diagonal=True)`` (reference basecircuit.py:295-369; gate factories
    gates.py:981-1185): multicontrol (any number of controls, 1- and 2-qubit targets), a general MPO, ``diagonal``,
    ``cmz`` and ``rzm`` are lowered to the tile-VM's native operations; the compiled plan, run through the numpy
    emulator of the descriptor format, equals the dense oracle applying the gates' full matrices.  Includes the
    reference's own known answers (tests/test_circuit.py:998-1040: multicontrol_gate(X, ctrl=[1, 0]))."""
    import tcmi as tc
    from scipy.stats import unitary_group
    from tcmi import plan as P
    from oracle import dense, gates as OG, plan_emulator as E

    g = tc.gates.multicontrol_gate(tc.gates._x_matrix, ctrl=[1, 0])
    ans = np.eye(8)
    ans[[4, 5]] = ans[[5, 4]]
    np.testing.assert_allclose(g.eval_matrix(), ans, atol=1e-12)

    n = 9
    rng = np.random.default_rng(1)
    c = tc.Circuit(n)
    ops = []
    for i in range(n):
        c.h(i)
        ops.append((OG.H, [i]))
        c.rx(i, theta=0.3 + 0.1 * i)
        ops.append((OG.rx(0.3 + 0.1 * i), [i]))

    def mc_matrix(u, ctrl):
        nt = int(np.log2(u.shape[0]))
        m = np.eye(2 ** (len(ctrl) + nt), dtype=np.complex128)
        v = int("".join(str(x) for x in ctrl), 2)
        m[v * 2 ** nt:(v + 1) * 2 ** nt, v * 2 ** nt:(v + 1) * 2 ** nt] = u
        return m

    u1 = unitary_group.rvs(2, random_state=5)
    c.multicontrol(0, 4, 7, 2, 5, ctrl=[1, 0, 1, 1], unitary=u1)              # four controls, one target
    ops.append((mc_matrix(u1, [1, 0, 1, 1]), [0, 4, 7, 2, 5]))
    u2 = unitary_group.rvs(4, random_state=6)
    c.multicontrol(8, 1, 3, 6, ctrl=[0, 1], unitary=u2.reshape(2, 2, 2, 2))    # two controls, two targets
    ops.append((mc_matrix(u2, [0, 1]), [8, 1, 3, 6]))
    c.mpo(2, 0, 1, mpo=tc.gates.multicontrol_gate(tc.gates._x_matrix, ctrl=[1, 0]))
    ops.append((ans.astype(np.complex128), [2, 0, 1]))
    dvec = np.exp(1j * rng.uniform(-3, 3, 2 ** 5))
    c.diagonal(6, 0, 3, 8, 1, diag=dvec)
    ops.append((np.diag(dvec), [6, 0, 3, 8, 1]))
    c.cmz(1, 2, 3, 4, 5, 6)
    z6 = np.ones(64, dtype=np.complex128)
    z6[-1] = -1
    ops.append((np.diag(z6), [1, 2, 3, 4, 5, 6]))
    c.rzm(7, 0, 4, 2, theta=1.2)
    zs = np.array([1.0])
    for _ in range(4):
        zs = np.kron(zs, [1.0, -1.0])
    ops.append((np.diag(np.exp(-0.6j * zs)), [7, 0, 4, 2]))
    for i in range(n):
        c.ry(i, theta=0.2 * (i + 1))
        ops.append((OG.ry(0.2 * (i + 1)), [i]))
    recs = c._gate_records()
    assert all(r.is_diag or len(r.qubits) <= 2 for r in recs)
    cfg = P.PlanConfig(R=2, LT=6, lowbits=3, vec=1)
    plan = P.compile_plan(recs, n, cfg, nparams=len(c._params))
    out = E.run_plan(plan, np.array([float(p) for p in c._params]))
    np.testing.assert_allclose(out, dense.run(n, ops), atol=1e-11)
    # the reverse sweep through the same gates (CNOT register moves around the folded terms, gradient of the rzm angle
    # and of the rotations) against central differences of the emulated forward plan
    vals = np.array([float(p) for p in c._params])
    for cfg in (P.PlanConfig(R=2, LT=6, lowbits=3, vec=1), P.PlanConfig(R=3, LT=5, lowbits=3, vec=2)):
        pl = P.compile_plan(recs, n, cfg, nparams=len(vals))
        psi = E.run_plan(pl, vals)
        np.testing.assert_allclose(psi, dense.run(n, ops), atol=1e-11)
        gvec = rng.normal(size=2 ** n) + 1j * rng.normal(size=2 ** n)
        ap = P.compile_adjoint_plan(recs, n, cfg)
        grad, psi_in = E.run_adjoint_plan(ap, vals, psi, gvec, len(vals))
        e0 = np.zeros(2 ** n)
        e0[0] = 1
        np.testing.assert_allclose(psi_in, e0, atol=1e-11)
        rzm_idx = next(i for i, op_ in enumerate(c._ops) if op_.name == "rzm")
        for i in [c._ops[rzm_idx].pidx, 3, len(vals) - 2]:
            vp, vm = vals.copy(), vals.copy()
            vp[i] += 1e-6
            vm[i] -= 1e-6
            fd = (np.real(np.vdot(gvec, E.run_plan(pl, vp))) - np.real(np.vdot(gvec, E.run_plan(pl, vm)))) / 2e-6
            assert abs(grad[i] - fd) < 1e-6, (i, grad[i], fd)
    # what the backend cannot do is refused, not approximated
    with pytest.raises(NotImplementedError):
        tc.Circuit(12).cmz(*range(11))
    with pytest.raises(NotImplementedError):
        tc.Circuit(4).diagonal(0, 1, 2, diag=np.array([1, 2, 1, 1, 1, 1, 1, 1.0]))


def _expand_live(mask, tile_bits, n):
    """Physical-bit mask of the positions where an amplitude's index may be non-zero: the pass's tile bits plus the bits
    of the compact tile index that ``mask`` lets vary."""
    from tcmi import executor as X

    tb = 0
    for p in tile_bits:
        tb |= 1 << int(p)
    if mask == X.LIVE_FULL:
        return (1 << n) - 1
    free = [p for p in range(n) if not (tb >> p) & 1]
    out = tb
    for i, p in enumerate(free):
        if (mask >> i) & 1:
            out |= 1 << p
    return out


def test_live_tile_masks_cover_every_nonzero_amplitude():
    """executor.live_masks against the pass emulator (oracle/plan_emulator.py): before pass k of a state started from
    |0...0> every non-zero amplitude sits in a live tile, and in the reverse sweep the psi entering pass j (un-computed
    pass by pass) is zero outside the live tiles to rounding, so skipping the other tiles drops nothing."""
    import tcmi as tc
    from tcmi import executor as X, plan as P
    from oracle import plan_emulator as E

    tc.set_backend("hip"); tc.set_dtype("complex64")
    import torch

    n, d = 17, 3
    rng = np.random.default_rng(2)
    prm = rng.uniform(0, 2 * np.pi, 2 * d * n)
    c = tc.templates.blocks.example_block(tc.Circuit(n), torch.tensor(prm), nlayers=d)
    gates, nparams = c._gate_records(), len(c._params)
    n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, "complex64", None)
    assert cfg.gen >= 2 and len(plan.descs) >= 2
    masks, fracs = X.live_masks(plan.descs, n_exec)
    assert masks[0] == 0 and fracs[0] == 2.0 ** -(n_exec - cfg.T) and masks[-1] == X.LIVE_FULL
    state = np.zeros(2 ** n_exec, dtype=np.complex128)
    state[0] = 1.0
    ptab = E.build_table(plan.ginfo, plan.cpool, prm, plan.ptab_size)[0] if plan.ptab_size else np.zeros(0)
    idx = np.arange(2 ** n_exec)
    sparse_seen = 0
    for k, desc in enumerate(plan.descs):
        allowed = _expand_live(masks[k], plan.passes[k].tile_bits, n_exec)
        outside = (idx & ~allowed) != 0
        assert not np.any(state[outside] != 0), k              # exactly zero: nothing ever touched these amplitudes
        sparse_seen += int(masks[k] != X.LIVE_FULL)
        # the pass restricted to its live tiles gives the same state as the pass on all tiles
        full = state.copy()
        E.run_pass(full, desc, plan.ctab, ptab)
        live_only = state.copy()
        E.run_pass(live_only, desc, plan.ctab, ptab)
        live_only[outside] = 0.0
        np.testing.assert_array_equal(full, live_only)
        state = full
    assert sparse_seen >= 1
    # no zero fill (CompiledCircuit.zero_bits): memory starts as garbage with amplitude 0 = 1; a pass reads the
    # amplitudes of its live tiles that have no not-yet-touched bit set and writes its live tiles whole -- it never reads
    # garbage and the final state is complete
    touched, mem = 0, np.full(2 ** n_exec, np.nan + 0j)
    mem[0] = 1.0
    for k, desc in enumerate(plan.descs):
        tb = 0
        for p_ in plan.passes[k].tile_bits:
            tb |= 1 << int(p_)
        zb = tb & ~touched
        live = (idx & ~_expand_live(masks[k], plan.passes[k].tile_bits, n_exec)) == 0
        read = live & ((idx & zb) == 0)
        assert not np.isnan(mem[read]).any(), k
        work = np.where(read, mem, 0.0)
        E.run_pass(work, desc, plan.ctab, ptab)
        mem = np.where(live, work, mem)
        touched |= tb
    assert touched == (1 << n_exec) - 1
    np.testing.assert_array_equal(mem, state)
    # reverse sweep of the whole gate list: psi un-computed pass by pass
    cfg_a, ap = X.choose_adjoint_plan(eg, n_exec, "complex64", True)
    rmasks, rfr = X.live_masks(ap.descs, n_exec, reverse=True)
    assert rmasks[0] == X.LIVE_FULL and rfr[-1] < 1.0
    for a, b in zip(rfr, rfr[1:]):
        assert b <= a                                           # the live sets shrink along the sweep
    psi = state.copy()
    lam = (rng.normal(size=psi.shape) + 1j * rng.normal(size=psi.shape))
    ptab_a = E.build_adjoint_table(ap.ginfo, ap.cpool, prm, ap.ptab_size)[0] if ap.ptab_size else np.zeros(0)
    gout = np.zeros(max(1, len(ap.gslot_param)))
    gout_live = np.zeros_like(gout)
    psi2, lam2 = psi.copy(), lam.copy()
    for j, desc in enumerate(ap.descs):
        allowed = _expand_live(rmasks[j], ap.passes[j].tile_bits, n_exec)
        outside = (idx & ~allowed) != 0
        assert np.abs(psi[outside]).max(initial=0.0) < 1e-12, j    # un-computed to rounding
        E.run_adjoint_pass(psi, lam, desc, ap.ctab, ptab_a, gout)
        # the live-tile sweep: tiles outside never enter (psi there counts as zero, lambda there is never needed again)
        psi2[outside] = 0.0
        lam2[outside] = 0.0
        E.run_adjoint_pass(psi2, lam2, desc, ap.ctab, ptab_a, gout_live)
    np.testing.assert_allclose(gout_live, gout, atol=1e-10)
    # a short sweep that leaves the constant head out ends on a dense state: its start bits switch the masks off
    first = next(i for i, g_ in enumerate(eg) if P.gate_has_param(g_))
    start = 0
    for g_ in eg[:first]:
        for q in g_.qubits:
            start |= 1 << (n_exec - 1 - q)
    if start == (1 << n_exec) - 1:
        r = X.choose_adjoint_plan(eg, n_exec, "complex64", False)
        if r is not None:
            m2, _ = X.live_masks(r[1].descs, n_exec, start_bits=start, reverse=True)
            assert all(m == X.LIVE_FULL for m in m2)


def test_pauli_terms_folded_into_the_first_sweep_pass():
    """plan.fold_rounds / OP_XFOLD: the single-X terms of the TFIM cotangent lambda = 2 sum_t w_t P_t |psi> on the qubits of
    the sweep's first tile are added to lambda in registers by that pass.  On the CPU emulator: the gradient of the folded
    sweep started from the PARTIAL cotangent equals the gradient of the plain sweep started from the full one, and the
    folded terms' energy arrives in the extra gradient slot."""
    n, d = 14, 3
    rng = np.random.default_rng(3)
    params = rng.uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    cfg = P.PlanConfig(R=4, LT=8, lowbits=3, vec=2, gen=2, shear2=True)
    psi = dense.run(n, W.hea_b_ops(n, d, params))
    h, J = -1.0, 1.0

    def X(q, v):
        return v.reshape(2**q, 2, -1)[:, ::-1, :].reshape(-1)

    idx = np.arange(2**n)
    lam = np.zeros_like(psi)
    for q in range(n):
        lam += 2 * h * X(q, psi)
    for q in range(n - 1):
        lam += 2 * J * (1 - 2 * ((idx >> (n - 1 - q)) & 1)) * (1 - 2 * ((idx >> (n - 2 - q)) & 1)) * psi
    ap0 = P.compile_adjoint_plan(recs, n, cfg, factorized=True)
    g0, _ = E.run_adjoint_plan(ap0, vals, psi, lam, len(vals))
    fold = [(n - 1 - q, 2 * h) for q in range(n)]
    ap1 = P.compile_adjoint_plan(recs, n, cfg, factorized=True, fold=fold, fold_param=len(vals))
    tile0 = set(ap0.passes[0].tile_bits)
    assert {fold[i][0] for i in ap1.folded} >= tile0 and len(tile0) == cfg.T       # (later passes fold the bits they bring)
    assert [pp.tile_bits for pp in ap1.passes] == [pp.tile_bits for pp in ap0.passes]       # same passes, same live tiles
    part = lam.copy()
    for i in ap1.folded:
        part -= 2 * h * X(n - 1 - fold[i][0], psi)
    g1, _ = E.run_adjoint_plan(ap1, vals, psi, part, len(vals) + 1)
    assert np.abs(g1[: len(vals)] - g0).max() < 1e-12 and np.abs(g0).max() > 0.1
    want = sum(h * np.real(np.vdot(psi, X(n - 1 - fold[i][0], psi))) for i in ap1.folded)
    assert abs(g1[len(vals)] - want) < 1e-12
    # the generated kernel of the folded pass: emitted, OP_XFOLD in its source
    from tcmi import specialize as S

    src, meta = S._source("adjoint", np.asarray(ap1.descs[0]), S.adjoint_opts(cfg), 0)
    assert src.count("X fold on register bit") == cfg.T          # the first pass folds the twelve bits of its tile


def test_whole_pauli_sum_cotangent_born_in_the_sweep():
    """Every term of the TFIM cotangent folded (compile_adjoint_plan fold= / dfold= / lam_zero): X terms by the first pass
    whose tile holds their qubit (two-shear rotations whose real factor would still be pending across such a pass boundary
    fall back to three shears), ZZ strings at the start, lambda never loaded (FLAG_LAMBDA_ZERO: the emulator is handed
    noise for it).  Gradient = the plain sweep's on the full cotangent; the extra slot = the energy."""
    n, d = 16, 3
    rng = np.random.default_rng(3)
    params = rng.uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    cfg = P.PlanConfig(R=4, LT=8, lowbits=3, vec=2, gen=2, shear2=True)
    psi = dense.run(n, W.hea_b_ops(n, d, params))
    h, J = -1.0, 1.0

    def X(q, v):
        return v.reshape(2**q, 2, -1)[:, ::-1, :].reshape(-1)

    idx = np.arange(2**n)
    zz = np.zeros(2**n)
    for q in range(n - 1):
        zz += J * (1 - 2 * ((idx >> (n - 1 - q)) & 1)) * (1 - 2 * ((idx >> (n - 2 - q)) & 1))
    lam = 2 * zz * psi
    for q in range(n):
        lam = lam + 2 * h * X(q, psi)
    ap0 = P.compile_adjoint_plan(recs, n, cfg, factorized=True)
    assert len(ap0.passes) >= 2                       # X terms are folded in more than one pass
    g0, _ = E.run_adjoint_plan(ap0, vals, psi, lam, len(vals))
    fold = [(n - 1 - q, 2 * h) for q in range(n)]
    dfold = [((1 << (n - 1 - q)) | (1 << (n - 2 - q)), 2 * J) for q in range(n - 1)]
    ap1 = P.compile_adjoint_plan(recs, n, cfg, factorized=True, fold=fold, fold_param=len(vals), dfold=dfold, lam_zero=True)
    assert ap1.folded == list(range(n))
    assert int(np.asarray(ap1.descs[0]).view(np.int32)[6]) & P.FLAG_LAMBDA_ZERO
    noise = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    g1, _ = E.run_adjoint_plan(ap1, vals, psi, noise, len(vals) + 1)
    assert np.abs(g1[: len(vals)] - g0).max() < 1e-12 and np.abs(g0).max() > 0.1
    energy = sum(h * np.real(np.vdot(psi, X(q, psi))) for q in range(n)) + float(np.sum(zz * np.abs(psi) ** 2))
    assert abs(g1[len(vals)] - energy) < 1e-12
    from tcmi import specialize as S

    src, _ = S._source("adjoint", np.asarray(ap1.descs[0]), S.adjoint_opts(cfg), 0)
    assert "diagonal strings of the cotangent" in src and "lam + wg_base" not in src.split("// -- ")[1]


def test_single_y_and_z_field_terms_fold_too():
    """OP_XFOLD kind 1 (single Y) and Z-only strings of one factor: alternating X / Y fields with random weights plus a Z
    field, every term born in the sweep; against the plain sweep on the full cotangent (emulator)."""
    n, d = 14, 3
    rng = np.random.default_rng(4)
    params = rng.uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    cfg = P.PlanConfig(R=4, LT=8, lowbits=3, vec=2, gen=2, shear2=True)
    psi = dense.run(n, W.hea_b_ops(n, d, params))

    def Xq(q, v):
        return v.reshape(2**q, 2, -1)[:, ::-1, :].reshape(-1)

    def Yq(q, v):
        t = v.reshape(2**q, 2, -1)
        o = np.empty_like(t)
        o[:, 0, :], o[:, 1, :] = -1j * t[:, 1, :], 1j * t[:, 0, :]
        return o.reshape(-1)

    idx = np.arange(2**n)
    wy, wz = rng.normal(size=n), rng.normal(size=n)
    lam, energy = np.zeros_like(psi), 0.0
    for q in range(n):
        pq = (Yq if q % 2 else Xq)(q, psi)
        lam += 2 * wy[q] * pq
        energy += wy[q] * np.real(np.vdot(psi, pq))
        z = 1 - 2 * ((idx >> (n - 1 - q)) & 1)
        lam += 2 * wz[q] * z * psi
        energy += wz[q] * float(np.sum(z * np.abs(psi) ** 2))
    g0, _ = E.run_adjoint_plan(P.compile_adjoint_plan(recs, n, cfg, factorized=True), vals, psi, lam, len(vals))
    fold = [(n - 1 - q, 2 * wy[q], q % 2) for q in range(n)]
    dfold = [(1 << (n - 1 - q), 2 * wz[q]) for q in range(n)]
    ap1 = P.compile_adjoint_plan(recs, n, cfg, factorized=True, fold=fold, fold_param=len(vals), dfold=dfold, lam_zero=True)
    assert ap1.folded == list(range(n))
    g1, _ = E.run_adjoint_plan(ap1, vals, psi, rng.normal(size=2**n) + 0j, len(vals) + 1)
    assert np.abs(g1[: len(vals)] - g0).max() < 1e-12 and abs(g1[len(vals)] - energy) < 1e-12


def test_two_factor_strings_fold_into_the_sweep():
    """OP_XFOLD2: strings with two X / Y factors -- the XX + YY couplings of a Heisenberg chain
    (tensorcircuit/quantum.py:2131-2219 heisenberg_hamiltonian) plus mixed XY / YX ones -- are born in the first sweep pass
    whose tile holds both qubits while neither has been touched; the pairs that never meet that condition (one qubit
    un-computed while the other is still outside the tile) are reported as not folded and stay in the cotangent handed in
    through memory.  Gradient and energy against the plain sweep on the full cotangent (emulator), and the generated
    kernel source of a folding pass cross-checked for the op."""
    n, d = 16, 3
    rng = np.random.default_rng(9)
    params = rng.uniform(0, 2 * np.pi, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
    recs = c._gate_records()
    vals = np.array([float(x) for x in c._params])
    cfg = P.PlanConfig(R=4, LT=8, lowbits=3, vec=2, gen=2, shear2=True)
    psi = dense.run(n, W.hea_b_ops(n, d, params))

    def P1(kind, q, v):
        t = v.reshape(2**q, 2, -1)
        o = np.empty_like(t)
        if kind == 0:
            o[:, 0, :], o[:, 1, :] = t[:, 1, :], t[:, 0, :]
        else:
            o[:, 0, :], o[:, 1, :] = -1j * t[:, 1, :], 1j * t[:, 0, :]
        return o.reshape(-1)

    # nearest-neighbour XX and YY with random couplings, a few XY / YX on next-nearest neighbours
    terms = [(q, q + 1, 0, 0, rng.normal()) for q in range(n - 1)] + [(q, q + 1, 1, 1, rng.normal()) for q in range(n - 1)]
    terms += [(q, q + 2, 0, 1, rng.normal()) for q in range(0, n - 2, 3)] + [(q, q + 2, 1, 0, rng.normal()) for q in range(1, n - 2, 3)]
    idx = np.arange(2**n)
    zz = np.zeros(2**n)
    for q in range(n - 1):
        zz += 0.7 * (1 - 2 * ((idx >> (n - 1 - q)) & 1)) * (1 - 2 * ((idx >> (n - 2 - q)) & 1))
    fold2 = [(n - 1 - a, n - 1 - b, 2 * w, ka, kb) for a, b, ka, kb, w in terms]
    dfold = [((1 << (n - 1 - q)) | (1 << (n - 2 - q)), 2 * 0.7) for q in range(n - 1)]
    ap1 = P.compile_adjoint_plan(recs, n, cfg, factorized=True, fold2=fold2, fold_param=len(vals), dfold=dfold)
    assert len(ap1.passes) >= 2
    done = set(ap1.folded2)
    assert len(done) >= len(terms) // 2 and len(done) < len(terms), (len(done), len(terms))     # most fold, some cannot
    assert any(getattr(rd, "fold2", None) for pp in ap1.passes[1:] for rd in pp.rounds)           # also in later passes
    lam_full, lam_rest, e_fold = 2 * zz * psi, np.zeros_like(psi), float(np.sum(zz * np.abs(psi) ** 2))
    for k, (a, b, ka, kb, w) in enumerate(terms):
        v = P1(ka, a, P1(kb, b, psi))
        lam_full = lam_full + 2 * w * v
        if k in done:
            e_fold += w * np.real(np.vdot(psi, v))
        else:
            lam_rest = lam_rest + 2 * w * v
    g0, _ = E.run_adjoint_plan(P.compile_adjoint_plan(recs, n, cfg, factorized=True), vals, psi, lam_full, len(vals))
    g1, _ = E.run_adjoint_plan(ap1, vals, psi, lam_rest, len(vals) + 1)
    assert np.abs(g1[: len(vals)] - g0).max() < 1e-11 and np.abs(g0).max() > 0.1
    assert abs(g1[len(vals)] - e_fold) < 1e-11
    # a pair is folded only where both its qubits were untouched: re-deriving the rule from the schedule
    rev = list(reversed(recs))
    touched = set()
    for pp in ap1.passes:
        tile_q = {n - 1 - b for b in pp.tile_bits}
        for rd in pp.rounds:
            for (_, _, _, pi, _, _) in getattr(rd, "fold2", []):
                a, b = terms[pi][0], terms[pi][1]
                assert {a, b} <= tile_q and not ({a, b} & touched)
        for gi in pp.gate_ids:
            touched |= set(rev[gi].qubits)
    from tcmi import specialize as S

    src = "".join(S._source("adjoint", np.asarray(dsc), S.adjoint_opts(cfg), i)[0] for i, dsc in enumerate(ap1.descs))
    assert "XX fold on register bits" in src and "YY fold on register bits" in src and "XY fold on register bits" in src
