"""Plan-specialised pass kernels on the GPU (tcmi/specialize.py): every generated kernel is compared with the
interpreting kernel it replaces (same descriptor, same tables: the states must agree to the last bit, gradients to
summation order) and with the dense oracle."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, workloads as W  # noqa: E402
from tcmi import _knobs as KN  # noqa: E402
from tcmi import specialize as S  # noqa: E402


@pytest.fixture
def tc64():
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    old = os.environ.get("TCMI_SPECIALIZE")
    yield tc
    if old is None:
        os.environ.pop("TCMI_SPECIALIZE", None)
    else:
        os.environ["TCMI_SPECIALIZE"] = old


def _mixed(tc, n, seed):
    rng = np.random.default_rng(seed)
    c = tc.Circuit(n)
    ops = []
    for i in range(n):
        c.h(i)
        ops.append((G.H, [i]))
    for layer in range(3):
        for i in range(n - 1):
            th = float(rng.uniform(0, 6))
            c.rzz(i, i + 1, theta=th)
            ops.append((G.rzz(th), [i, i + 1]))
        for i in range(n):
            th = float(rng.uniform(0, 6))
            c.rx(i, theta=th)
            ops.append((G.rx(th), [i]))
        for i in range(layer, n, 3):
            th = float(rng.uniform(0, 6))
            c.ry(i, theta=th)
            ops.append((G.ry(th), [i]))
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n - 1))
        b = b if b < a else b + 1
        c.cnot(a, b)
        ops.append((G.CNOT, [a, b]))
        c.swap(1, n - 2)
        ops.append((G.SWAP, [1, n - 2]))
        u = G.random_two_qubit_gate(5 + layer + seed)
        c.any(3, 4 + layer, unitary=u)
        ops.append((u, [3, 4 + layer]))
        c.cz(2 + layer, 5)
        ops.append((G.CZ, [2 + layer, 5]))
    return c, ops


def _state(tc, build, flag):
    os.environ["TCMI_SPECIALIZE"] = flag
    c = build()
    cc = c._compiled()
    for k in ("_spec_fwd",):
        if hasattr(cc, k):
            setattr(cc, k, None)
    return tc.backend.numpy(c.wavefunction())


@pytest.mark.parametrize("n,seed", [(13, 0), (14, 1), (16, 2), (18, 3)])
def test_specialised_forward_passes_equal_the_interpreter_bit_for_bit(tc64, n, seed):
    tc = tc64
    from tcmi import specialize as S

    tc.set_contractor("plain")      # the state-vector plan (the cut order has its own half-circuit plans)
    try:
        ref_ops = _mixed(tc, n, seed)[1]
        a = _state(tc, lambda: _mixed(tc, n, seed)[0], "0")
        before = S.STATS["compiled"] + S.STATS["cache_hits"]
        b = _state(tc, lambda: _mixed(tc, n, seed)[0], "1")
        assert S.STATS["compiled"] + S.STATS["cache_hits"] > before, "no specialised kernel was used"
        assert np.array_equal(a, b), np.abs(a - b).max()
        if n <= 16:
            ref = dense.run(n, ref_ops)
            assert np.abs(b - ref).max() < 1e-5
    finally:
        tc.set_contractor("greedy")


@pytest.mark.parametrize("n,d", [(13, 2), (16, 3), (20, 4)])
def test_specialised_reverse_sweep_matches_the_interpreter_and_the_oracle(tc64, n, d):
    """HEA-B + TFIM value_and_grad: specialised sweep (forced) against the interpreting sweep; one gradient component
    against a central difference of the dense oracle at n <= 16."""
    tc = tc64
    rng = np.random.default_rng(n)
    params = rng.uniform(0, 2 * np.pi, [2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    res = {}
    for flag in ("0", "1"):
        os.environ["TCMI_SPECIALIZE"] = flag
        from tcmi import executor as X

        X._CACHE.clear()
        pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
        v, g = tc.backend.value_and_grad(energy)(pt)
        res[flag] = (float(v), tc.backend.numpy(g))
    assert abs(res["0"][0] - res["1"][0]) < 1e-5
    assert np.abs(res["0"][1] - res["1"][1]).max() < 5e-6
    if n <= 16:
        eps = 1e-5
        pp, pm = params.copy(), params.copy()
        pp[1, 3] += eps
        pm[1, 3] -= eps
        fd = (W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, pp)), n)
              - W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, pm)), n)) / (2 * eps)
        assert abs(res["1"][1][1, 3] - fd) < 5e-4


def test_specialised_kernels_under_vmap_and_jit(tc64):
    """The bench's call shape, small: jit(vvag(energy)) over a batch; specialised against interpreted."""
    tc = tc64
    import torch

    n, d, B = 14, 3, 4
    params = torch.from_numpy(np.random.default_rng(5).normal(0, 0.5, [B, 2 * d, n]).astype(np.float32)).cuda()

    def energy(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for j in range(d):
            for i in range(n - 1):
                c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
            for i in range(n):
                c.rx(i, theta=p[2 * j + 1, i])
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    out = {}
    for flag in ("0", "1"):
        os.environ["TCMI_SPECIALIZE"] = flag
        from tcmi import executor as X

        X._CACHE.clear()
        f = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
        for _ in range(3):
            v, g = f(params)
        out[flag] = (v.cpu().numpy(), g.cpu().numpy())
    assert np.abs(out["0"][0] - out["1"][0]).max() < 1e-5
    assert np.abs(out["0"][1] - out["1"][1]).max() < 5e-6


def test_live_tile_passes_equal_the_dense_passes(tc64):
    """executor.live_masks: a state started from |0...0> runs its first passes -- and the reverse sweep its last ones -- on
    the tiles that can be non-zero only.  Same kernels with every tile live (TCMI_SPARSE_START=0 semantics): the state is
    bit-identical, energy and gradient agree to rounding (a zero psi tile that is skipped was zero to 1e-7 only)."""
    tc = tc64
    from tcmi import executor as X

    n, d = 20, 4
    params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    os.environ["TCMI_SPECIALIZE"] = "1"
    tc.set_contractor("plain")
    old = X.SPARSE_START
    res = {}
    try:
        for flag in (True, False):
            X.SPARSE_START = flag
            X._CACHE.clear()
            pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
            c = tc.Circuit(n)
            W.hea_b(c, n, d, pt, zz=tc.gates._zz_matrix)
            cc = c._compiled()
            if flag:
                masks, fracs = cc.zero_start()
                assert any(m != X.LIVE_FULL for m in masks) and min(fracs) < 1e-2
                _, rmasks, rfracs = cc._adjoint_from_zero()
                assert any(m != X.LIVE_FULL for m in rmasks) and min(rfracs) < 1e-2
            psi = tc.backend.numpy(c.wavefunction())
            v, g = tc.backend.value_and_grad(energy)(pt)
            res[flag] = (psi, float(v), tc.backend.numpy(g))
    finally:
        X.SPARSE_START = old
        tc.set_contractor("greedy")
    assert np.array_equal(res[True][0], res[False][0])
    assert abs(res[True][1] - res[False][1]) < 1e-5
    assert np.abs(res[True][2] - res[False][2]).max() < 5e-6


def test_live_tile_kernels_against_the_dense_oracle_directly(tc64):
    """The executed fast path of the VQE leg -- plan-specialised kernels, live-tile passes, no zero fill, live-tile reverse
    sweep, XCD-aware tile order -- against ``oracle.dense`` itself (not against the product's own dense passes) at n = 21:
    the state (complex64 tolerance of BASELINE.json, 1e-5), the TFIM energy, and three gradient components against
    central differences of the oracle's energy (the reference's convention, tests/test_mpscircuit.py:452-457)."""
    tc = tc64
    from tcmi import executor as X

    n, d = 21, 4
    params = np.random.default_rng(2100).uniform(0, 2 * np.pi, [2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    os.environ["TCMI_SPECIALIZE"] = "1"
    tc.set_contractor("plain")
    try:
        X._CACHE.clear()
        pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
        c = tc.Circuit(n)
        W.hea_b(c, n, d, pt, zz=tc.gates._zz_matrix)
        cc = c._compiled()
        masks, fracs = cc.zero_start()
        assert X.SPARSE_START and any(m != X.LIVE_FULL for m in masks)      # the live-tile path is what runs
        psi = tc.backend.numpy(c.wavefunction())
        assert all(k is not None for k in cc._specialised())                # every pass on a generated kernel
        v, g = tc.backend.value_and_grad(energy)(pt)
        g = tc.backend.numpy(g)
    finally:
        tc.set_contractor("greedy")
    ref = dense.run(n, W.hea_b_ops(n, d, params))
    assert np.abs(psi - ref).max() < 1e-5
    e_ref = W.tfim_energy_dense(ref, n)
    assert abs(float(v) - e_ref) < 1e-5 * n
    eps = 1e-5
    for (r, q) in ((0, 5), (3, 11), (7, 20)):
        pp, pm = params.copy(), params.copy()
        pp[r, q] += eps
        pm[r, q] -= eps
        fd = (W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, pp)), n)
              - W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, pm)), n)) / (2 * eps)
        assert abs(g[r, q] - fd) < 2e-4, ((r, q), g[r, q], fd)


def test_generated_kernel_options_do_not_change_results(tc64):
    """Round-5 code-generation choices -- XCD-aware tile order, raised load priority, single 8-byte LDS accesses -- only move
    work in time and space: with each of them off (specialize.EXTRA_OPTS) the state is bit-identical and the gradient agrees to
    summation order."""
    tc = tc64
    from tcmi import executor as X

    n, d = 18, 3
    params = np.random.default_rng(18).uniform(0, 2 * np.pi, [2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    os.environ["TCMI_SPECIALIZE"] = "1"
    tc.set_contractor("plain")
    res = {}
    try:
        for exp in ("", "xcd=0,prio=0,single8=0", "xcd=1"):
            S.EXTRA_OPTS.clear()
            S.EXTRA_OPTS.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in exp.split(",") if kv})
            X._CACHE.clear()
            pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
            c = tc.Circuit(n)
            W.hea_b(c, n, d, pt, zz=tc.gates._zz_matrix)
            psi = tc.backend.numpy(c.wavefunction())
            v, g = tc.backend.value_and_grad(energy)(pt)
            res[exp] = (psi, float(v), tc.backend.numpy(g))
    finally:
        S.EXTRA_OPTS.clear()
        tc.set_contractor("greedy")
        X._CACHE.clear()
    for exp in ("xcd=0,prio=0,single8=0", "xcd=1"):
        assert np.array_equal(res[""][0], res[exp][0])
        assert abs(res[""][1] - res[exp][1]) < 1e-5
        assert np.abs(res[""][2] - res[exp][2]).max() < 5e-6
    ref = dense.run(n, W.hea_b_ops(n, d, params))
    assert np.abs(res["xcd=1"][0] - ref).max() < 1e-5


def test_pauli_terms_folded_into_the_sweep_match_the_plain_route_and_the_oracle(tc64):
    """executor.fold_setup: in the traced value_and_grad (jit) the single-X terms of the TFIM on the qubits of the sweep's
    first tile are added to lambda inside that pass (OP_XFOLD, generated kernels) and the Pauli-sum tile passes shrink.
    With the fold on (default) and off (TCMI_PAULI_FOLD=0): same energy, same gradient; and both against oracle.dense
    (energy, two central-difference gradient components)."""
    tc = tc64
    import torch
    from tcmi import executor as X

    n, d, B = 20, 3, 2
    rng = np.random.default_rng(2020)
    params_np = rng.uniform(0, 2 * np.pi, [B, 2 * d, n])

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    os.environ["TCMI_SPECIALIZE"] = "1"
    old = os.environ.get("TCMI_PAULI_FOLD")
    res = {}
    try:
        for flag in ("1", "0"):
            os.environ["TCMI_PAULI_FOLD"] = flag
            X._CACHE.clear()
            f = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
            p = torch.from_numpy(params_np.astype(np.float32)).cuda()
            for _ in range(3):
                v, g = f(p)
            torch.cuda.synchronize()
            X.EVENT_LOG = []                      # the fourth call runs the traced pipeline: its launches alone
            v, g = f(p)
            torch.cuda.synchronize()
            launches = sum(e[3] for e in X.EVENT_LOG if e[0] == "pauli_sum")
            X.EVENT_LOG = None
            res[flag] = (v.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64), launches)
    finally:
        X.EVENT_LOG = None
        if old is None:
            os.environ.pop("TCMI_PAULI_FOLD", None)
        else:
            os.environ["TCMI_PAULI_FOLD"] = old
        X._CACHE.clear()
    assert res["1"][2] == 0 and res["0"][2] >= 2, (res["1"][2], res["0"][2])   # the fold really ran: NO Pauli-sum pass at all
    assert np.abs(res["1"][0] - res["0"][0]).max() < 2e-5
    assert np.abs(res["1"][1] - res["0"][1]).max() < 2e-5
    p64 = params_np.astype(np.float32).astype(np.float64)
    ref = lambda q: W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, q)), n)  # noqa: E731
    assert abs(res["1"][0][0] - ref(p64[0])) < 2e-4
    eps = 1e-5
    for (r, q) in ((1, 2), (4, 17)):
        pp, pm = p64[0].copy(), p64[0].copy()
        pp[r, q] += eps
        pm[r, q] -= eps
        fd = (ref(pp) - ref(pm)) / (2 * eps)
        assert abs(res["1"][1][0, r, q] - fd) < 2e-4, ((r, q), res["1"][1][0, r, q], fd)


def test_partial_fold_with_y_fields_and_strings_that_stay_in_the_tile_passes(tc64):
    """A Hamiltonian of which only a part folds: alternating X / Y fields (OP_XFOLD kinds 0 / 1), a Z field and the ZZ chain
    (OP_DFOLD) are born in the sweep, two XX strings and one XYZ string are not -- they keep going through
    tcmi_apply_pauli_sum_tiled, and lambda IS loaded by the first pass (no FLAG_LAMBDA_ZERO).  Fold on / off agree; the
    energy equals the dense oracle's."""
    tc = tc64
    import torch
    from tcmi import executor as X

    n, d = 20, 3
    rng = np.random.default_rng(77)
    params_np = rng.uniform(0, 2 * np.pi, [2 * d, n])
    wf, wz = rng.normal(size=n), rng.normal(size=n)

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += float(wf[i]) * (c.expectation_ps(y=[i]) if i % 2 else c.expectation_ps(x=[i]))
            e += float(wz[i]) * c.expectation_ps(z=[i])
        for i in range(n - 1):
            e += 0.7 * c.expectation_ps(z=[i, i + 1])
        e += 0.31 * c.expectation_ps(x=[3, 4]) - 0.22 * c.expectation_ps(x=[11, 17]) + 0.5 * c.expectation_ps(x=[6], y=[7], z=[9])
        return tc.backend.real(e)

    os.environ["TCMI_SPECIALIZE"] = "1"
    old = os.environ.get("TCMI_PAULI_FOLD")
    res = {}
    try:
        for flag in ("1", "0"):
            os.environ["TCMI_PAULI_FOLD"] = flag
            X._CACHE.clear()
            f = tc.backend.jit(tc.backend.value_and_grad(energy))
            p = torch.from_numpy(params_np.astype(np.float32)).cuda()
            for _ in range(3):
                v, g = f(p)
            torch.cuda.synchronize()
            X.EVENT_LOG = []
            v, g = f(p)
            torch.cuda.synchronize()
            launches = sum(e[3] for e in X.EVENT_LOG if e[0] == "pauli_sum")
            X.EVENT_LOG = None
            res[flag] = (float(v), g.cpu().numpy().astype(np.float64), launches)
    finally:
        X.EVENT_LOG = None
        if old is None:
            os.environ.pop("TCMI_PAULI_FOLD", None)
        else:
            os.environ["TCMI_PAULI_FOLD"] = old
        X._CACHE.clear()
    assert 0 < res["1"][2] < res["0"][2], (res["1"][2], res["0"][2])      # fewer tile passes, but not none
    assert abs(res["1"][0] - res["0"][0]) < 3e-5 and np.abs(res["1"][1] - res["0"][1]).max() < 3e-5
    psi = dense.run(n, W.hea_b_ops(n, d, params_np.astype(np.float32).astype(np.float64)))
    ref = 0.0
    for i in range(n):
        ref += wf[i] * dense.expectation(psi, n, (G.Y if i % 2 else G.X, [i])).real + wz[i] * dense.expectation(psi, n, (G.Z, [i])).real
    for i in range(n - 1):
        ref += 0.7 * dense.expectation(psi, n, (G.Z, [i]), (G.Z, [i + 1])).real
    ref += 0.31 * dense.expectation(psi, n, (G.X, [3]), (G.X, [4])).real - 0.22 * dense.expectation(psi, n, (G.X, [11]), (G.X, [17])).real
    ref += 0.5 * dense.expectation(psi, n, (G.X, [6]), (G.Y, [7]), (G.Z, [9])).real
    assert abs(res["1"][0] - ref) < 2e-4, (res["1"][0], ref)


def test_heisenberg_couplings_folded_into_the_sweep(tc64):
    """the XX + YY + ZZ chain of a Heisenberg model (reference tensorcircuit/quantum.py:2131-2219
    heisenberg_hamiltonian; templates/measurements.py:156-191) under a traced value_and_grad: the two-factor strings whose
    qubits meet untouched in a tile of the sweep are born there (OP_XFOLD2), the ZZ strings at its start (OP_DFOLD); the
    few pairs that straddle two tiles keep going through tcmi_apply_pauli_sum_tiled.  Fold on / off: same energy and
    gradient, fewer Pauli-sum launches; and against oracle.dense (energy, two central-difference gradient components)."""
    tc = tc64
    import torch
    from tcmi import executor as X

    n, d = 20, 3
    rng = np.random.default_rng(515)
    params_np = rng.uniform(0, 2 * np.pi, [2 * d, n])
    jx, jy, jz = rng.normal(size=n - 1), rng.normal(size=n - 1), rng.normal(size=n - 1)

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n - 1):
            e += float(jx[i]) * c.expectation_ps(x=[i, i + 1]) + float(jy[i]) * c.expectation_ps(y=[i, i + 1])
            e += float(jz[i]) * c.expectation_ps(z=[i, i + 1])
        return tc.backend.real(e)

    def ref(q):
        psi = dense.run(n, W.hea_b_ops(n, d, q))
        out = 0.0
        for i in range(n - 1):
            out += jx[i] * dense.expectation(psi, n, (G.X, [i]), (G.X, [i + 1])).real
            out += jy[i] * dense.expectation(psi, n, (G.Y, [i]), (G.Y, [i + 1])).real
            out += jz[i] * dense.expectation(psi, n, (G.Z, [i]), (G.Z, [i + 1])).real
        return out

    os.environ["TCMI_SPECIALIZE"] = "1"
    old = os.environ.get("TCMI_PAULI_FOLD")
    res = {}
    try:
        for flag in ("1", "0"):
            os.environ["TCMI_PAULI_FOLD"] = flag
            X._CACHE.clear()
            f = tc.backend.jit(tc.backend.value_and_grad(energy))
            p = torch.from_numpy(params_np.astype(np.float32)).cuda()
            for _ in range(3):
                v, g = f(p)
            torch.cuda.synchronize()
            X.EVENT_LOG = []
            v, g = f(p)
            torch.cuda.synchronize()
            ev = [e for e in X.EVENT_LOG if e[0] == "pauli_sum"]
            X.EVENT_LOG = None
            res[flag] = (float(v), g.cpu().numpy().astype(np.float64), sum(e[3] for e in ev), sum(e[4] for e in ev))
    finally:
        X.EVENT_LOG = None
        if old is None:
            os.environ.pop("TCMI_PAULI_FOLD", None)
        else:
            os.environ["TCMI_PAULI_FOLD"] = old
        X._CACHE.clear()
    # the fold ran: the tile passes that remain move fewer bytes (fewer strings -> fewer index bits to cover)
    assert res["1"][3] < res["0"][3] and res["0"][2] >= 2, (res["1"][2:], res["0"][2:])
    assert abs(res["1"][0] - res["0"][0]) < 3e-5 and np.abs(res["1"][1] - res["0"][1]).max() < 3e-5
    p64 = params_np.astype(np.float32).astype(np.float64)
    assert abs(res["1"][0] - ref(p64)) < 5e-5, (res["1"][0], ref(p64))
    eps = 1e-5
    for (r, q) in ((1, 2), (4, 17)):
        pp, pm = p64.copy(), p64.copy()
        pp[r, q] += eps
        pm[r, q] -= eps
        fd = (ref(pp) - ref(pm)) / (2 * eps)
        assert abs(res["1"][1][r, q] - fd) < 5e-5, ((r, q), res["1"][1][r, q], fd)


def test_cut_suffix_reads_its_inputs_from_the_prefix_batch(tc64):
    """tcmi_spec_run_pass_from: the first pass of a cut half-circuit's suffix reads state b as weight[b] * prefix[b >> shift]
    instead of a replicated, weighted copy written by an elementwise launch.  Same state with the fused load on / off
    (knob cut_fused_rep), batched over two parameter rows, and against the dense oracle."""
    tc = tc64
    import torch
    from tcmi import executor as X

    n, d, B = 24, 3, 2          # halves of 12 qubits: the smallest tile the packed kernels (and their generated forms) run on
    params = np.random.default_rng(16).uniform(0, 2 * np.pi, [B, 2 * d, n])
    os.environ["TCMI_SPECIALIZE"] = "1"
    old = KN.VALUES.get("cut_fused_rep")
    tc.set_contractor("cut")
    res = {}
    try:
        for flag in ("1", "0"):
            KN.VALUES["cut_fused_rep"] = flag
            X._CACHE.clear()

            def wf(p):
                c = tc.Circuit(n)
                W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
                return c.wavefunction()

            pt = torch.from_numpy(params.astype(np.float32)).cuda()
            c0 = tc.Circuit(n)
            W.hea_b(c0, n, d, pt[0], zz=tc.gates._zz_matrix)
            cc = c0._compiled()
            if not isinstance(cc, X.CutCircuit) or cc.right.s == 0:
                pytest.skip("the planner did not cut this circuit into prefix / suffix batches")
            st = tc.backend.vmap(wf)(pt)
            if flag == "1":       # the generated "src" kernels of both suffixes are what ran
                assert cc.left.suffix._specialised_src() is not None and cc.right.suffix._specialised_src() is not None
            res[flag] = st.cpu().numpy()
    finally:
        KN.VALUES.pop("cut_fused_rep", None)
        if old is not None:
            KN.VALUES["cut_fused_rep"] = old
        tc.set_contractor("greedy")
        X._CACHE.clear()
    assert np.abs(res["1"] - res["0"]).max() < 1e-7
    ref = dense.run(n, W.hea_b_ops(n, d, params[1].astype(np.float32).astype(np.float64)), inplace=True)
    assert np.abs(res["1"][1] - ref).max() < 1e-5
