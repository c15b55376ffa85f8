"""Parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the public
Circuit API -> ctypes C ABI -> libtcmi.so, against the CPU oracle on the same seeded inputs and
against the committed golden vectors.  Tolerances are BASELINE.json's: complex128 1e-10,
complex64 1e-5 (max-abs on amplitudes)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, tn, workloads as W  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hea_golden.npz"))
TOL = {"complex64": 1e-5, "complex128": 1e-10}


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import torch
    import tcmi as tc
    from tcmi import _lib

    assert torch.cuda.is_available()
    _lib.lib()  # the HIP extension must be the thing that runs
    tc.set_backend("hip")
    tc.set_dtype(request.param)
    tc.set_contractor("greedy")
    yield tc
    tc.set_dtype("complex64")
    tc.set_contractor("greedy")


def _np(tc, t):
    return tc.backend.numpy(t)


def test_reference_kats_through_product(tcd):
    """reference tests/test_circuit.py:22-53, 95-99, 111-115, 397-401."""
    tc = tcd
    g = np.arange(16).reshape(2, 2, 2, 2)
    qc = tc.Circuit(2); qc.unitary(0, 1, unitary=tc.gates.Gate(g))
    assert np.real(_np(tc, qc.wavefunction())[2]) == 8
    qc = tc.Circuit(2); qc.unitary(1, 0, unitary=tc.gates.Gate(g))
    assert np.real(_np(tc, qc.wavefunction())[2]) == 4
    qc = tc.Circuit(2); qc.unitary(0, unitary=tc.gates.Gate(np.arange(4).reshape(2, 2)))
    assert np.real(_np(tc, qc.wavefunction())[2]) == 2
    c = tc.Circuit(2); c.x(0)
    np.testing.assert_allclose(_np(tc, c.amplitude("10")), 1.0)
    c.CNOT(0, 1)
    np.testing.assert_allclose(_np(tc, c.amplitude("11")), 1.0)
    c = tc.Circuit(2, inputs=np.eye(4)[2]); c.iswap(0, 1)
    np.testing.assert_allclose(_np(tc, c.state()), G.iswap(1.0)[:, 2], atol=1e-6)
    c = tc.Circuit(1); c.X(0); c.SD(0)
    np.testing.assert_allclose(_np(tc, c.state()), np.array([0.0, -1.0j]), atol=1e-6)
    c = tc.Circuit(1); c.H(0)
    np.testing.assert_allclose(_np(tc, c.state()), np.array([1, 1]) / np.sqrt(2), atol=1e-6)
    assert tuple(c.wavefunction("ket").shape) == (2, 1) and tuple(c.wavefunction("bra").shape) == (1, 2)


@pytest.mark.parametrize("n,d", [(4, 2), (8, 3), (10, 4), (12, 4)])
def test_golden_hea_b(tcd, n, d):
    tc = tcd
    params = GOLD[f"hea_b_{n}_{d}_params"]
    c = tc.Circuit(n)
    W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    psi = _np(tc, c.wavefunction())
    np.testing.assert_allclose(psi, GOLD[f"hea_b_{n}_{d}_state"], atol=TOL[tc.dtypestr])


def test_golden_config1_and_hea_a(tcd):
    tc = tcd
    c = tc.Circuit(10)
    W.hea_b(c, 10, 4, tc.backend.ones([8, 10], dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    np.testing.assert_allclose(_np(tc, c.wavefunction()), GOLD["hea_b_10_4_ones_state"], atol=TOL[tc.dtypestr])
    c = tc.Circuit(9)
    W.hea_a(c, 9, 3, GOLD["hea_a_9_3_params"])  # python-float parameters
    np.testing.assert_allclose(_np(tc, c.wavefunction()), GOLD["hea_a_9_3_state"], atol=TOL[tc.dtypestr])


@pytest.mark.parametrize("n", [1, 2, 5, 8, 11, 13, 15, 18])
def test_mixed_gate_set_vs_oracle(tcd, n):
    """Every supported gate family, arbitrary qubit pairs, both orientations; ragged sizes from a
    single qubit (padded tile) to multi-pass plans."""
    tc = tcd
    rng = np.random.default_rng(n)
    c = tc.Circuit(n)
    ops = []

    def both(name, qs, mat, **kw):
        getattr(c, name)(*qs, **kw)
        ops.append((mat, list(qs)))

    for layer in range(3):
        for q in range(n):
            t = float(rng.uniform(0, 2 * np.pi))
            kind = (q + layer) % 6
            if kind == 0: both("rx", [q], G.rx(t), theta=t)
            elif kind == 1: both("ry", [q], G.ry(t), theta=t)
            elif kind == 2: both("rz", [q], G.rz(t), theta=t)
            elif kind == 3: both("h", [q], G.H)
            elif kind == 4: both("phase", [q], G.phase(t), theta=t)
            else: both("u", [q], G.u(t, 0.3, 0.9), theta=t, phi=0.3, lbd=0.9)
        if n >= 2:
            for k in range(n):
                a, b = (int(x) for x in rng.choice(n, 2, replace=False))
                t = float(rng.uniform(0, 2 * np.pi))
                kind = (k + layer) % 9
                if kind == 0: both("cnot", [a, b], G.CNOT)
                elif kind == 1: both("cz", [a, b], G.CZ)
                elif kind == 2: both("swap", [a, b], G.SWAP)
                elif kind == 3: both("rzz", [a, b], G.rzz(t), theta=t)
                elif kind == 4: both("rxx", [a, b], G.rxx(t), theta=t)
                elif kind == 5: both("iswap", [a, b], G.iswap(0.4), theta=0.4)
                elif kind == 6: both("crx", [a, b], G.controlled(G.rx(t)), theta=t)
                elif kind == 7: both("cphase", [a, b], G.controlled(G.phase(t)), theta=t)
                else:
                    u = G.random_two_qubit_gate(100 * n + k)
                    c.any(a, b, unitary=u)
                    ops.append((u, [a, b]))
    psi = _np(tc, c.wavefunction())
    np.testing.assert_allclose(psi, dense.run(n, ops), atol=TOL[tc.dtypestr])


def test_inputs_and_negative_index(tcd):
    tc = tcd
    n = 6
    rng = np.random.default_rng(0)
    inp = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    inp /= np.linalg.norm(inp)
    c = tc.Circuit(n, inputs=inp)
    c.rx(-1, theta=0.4); c.cnot(-2, 0)
    ref = dense.run(n, [(G.rx(0.4), [n - 1]), (G.CNOT, [n - 2, 0])], inputs=inp)
    np.testing.assert_allclose(_np(tc, c.state()), ref, atol=TOL[tc.dtypestr])


@pytest.mark.parametrize("opts", [{"lowbits": 3}, {"lowbits": 7}, {"R": 4, "LT": 8}, {"R": 5, "LT": 9}, {"R": 4, "LT": 9}])
def test_plan_variants_on_device(opts):
    import tcmi as tc

    tc.set_backend("hip"); tc.set_dtype("complex64")
    tc.set_contractor("greedy", **opts)
    try:
        n, d = 16, 3
        params = np.random.default_rng(16).uniform(0, 2 * np.pi, [2 * d, n])
        c = tc.Circuit(n)
        W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
        c.cnot(0, 15); c.swap(3, 9)
        ref = dense.run(n, W.hea_b_ops(n, d, params) + [(G.CNOT, [0, 15]), (G.SWAP, [3, 9])])
        np.testing.assert_allclose(tc.backend.numpy(c.state()), ref, atol=1e-5)
    finally:
        tc.set_contractor("greedy")


def test_vmap_batched_states(tcd):
    """reference abstract_backend.py:2520-2539: vmap over the leading axis == per-sample calls,
    executed as ONE batched launch per pass."""
    tc = tcd
    n, d, B = 10, 2, 5
    pbs = np.random.default_rng(9).uniform(0, 2 * np.pi, [B, 2 * d, n])

    def f(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return c.state()

    out = _np(tc, tc.backend.vmap(f)(tc.backend.convert_to_tensor(pbs, dtype=tc.rdtypestr)))
    assert out.shape == (B, 2**n)
    for b in range(B):
        np.testing.assert_allclose(out[b], dense.run(n, W.hea_b_ops(n, d, pbs[b])), atol=TOL[tc.dtypestr])


def test_full_size_config2_vs_oracle_and_properties():
    """BASELINE config 2 at full size (n=24, d=8, complex64): max-abs parity with the oracle's TN
    contraction (complex128), unit norm, and linearity of the executor in the input state."""
    import torch
    import tcmi as tc

    tc.set_backend("hip"); tc.set_dtype("complex64")
    n, d, params = W.config_params(2)
    c = tc.Circuit(n)
    W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
    psi = c.state()
    assert abs(float((psi.abs().double() ** 2).sum()) - 1.0) < 1e-5
    oc = tn.Circuit(n, dtype=np.complex128)
    W.hea_b(oc, n, d, params.astype(np.float64))
    ref = oc.wavefunction()
    assert np.abs(tc.backend.numpy(psi) - ref).max() < 1e-5
    # linearity: U(a x + b y) = a U x + b U y on random inputs (size-independent property)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2**n, dtype=torch.complex64, device="cuda", generator=g)
    y = torch.randn(2**n, dtype=torch.complex64, device="cuda", generator=g)
    cc = c._compiled()
    p = c._param_tensor()
    ux = cc.state(p, inputs=x)[0].clone()
    uy = cc.state(p, inputs=y)[0].clone()
    uxy = cc.state(p, inputs=0.3 * x - 1.7j * y)[0]
    rel = float((uxy - (0.3 * ux - 1.7j * uy)).abs().max() / uxy.abs().max())
    assert rel < 1e-5
    # unitarity: norms are preserved
    assert abs(float(torch.linalg.vector_norm(ux) / torch.linalg.vector_norm(x)) - 1) < 1e-5


# ---- expectation (fused measurement passes, K4) ---------------------------------------------------
def test_expectation_kats(tcd):
    """reference tests/test_circuit.py:102-108, 317-323, 448-468, 550-564, 1501-1504."""
    tc = tcd
    atol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    c = tc.Circuit(2); c.x(1); c.crx(1, 0, theta=0.3)
    np.testing.assert_allclose(_np(tc, c.expectation([tc.gates._z_matrix, 0])), np.cos(0.3), atol=atol)
    np.testing.assert_allclose(_np(tc, c.expectation([tc.gates._z_matrix, 0])), 0.95533645, atol=1e-5)
    c = tc.Circuit(2); c.H(0)
    np.testing.assert_allclose(_np(tc, c.expectation((tc.gates.z(), [0]))), 0, atol=1e-7)
    for inp, want in (([0, 0, 0, 1.0], 1.0), ([0, 0, 1.0, 0], 1.0), ([1.0, 0, 0, 0], -1.0)):
        c = tc.Circuit(2, inputs=np.array(inp)); c.X(0)
        assert _np(tc, c.expectation((tc.gates.z(), [0]))) == want
    c = tc.Circuit(2); c.x(0)
    np.testing.assert_allclose(_np(tc, c.expectation_ps(z=[0])), -1, atol=atol)
    np.testing.assert_allclose(_np(tc, c.expectation_ps(z=[1])), 1, atol=atol)
    np.testing.assert_allclose(_np(tc, c.expectation_ps(ps=[3, 3])), -1, atol=atol)
    c = tc.Circuit(1); c.h(0); c.sd(0)
    np.testing.assert_allclose(_np(tc, c.expectation_ps(y=[0])), -1, atol=atol)
    with pytest.raises(ValueError, match="Cannot measure two operators in one index"):
        c.expectation((tc.gates.z(), [0]), (tc.gates.x(), [0]))


@pytest.mark.parametrize("n,d", [(6, 2), (10, 4), (13, 2), (16, 2)])
def test_tfim_energy_vs_oracle(tcd, n, d):
    """The reference's TFIM loop (benchmarks/scripts/vqe_tc.py:75-81), term by term, plus random
    Pauli strings with X/Y/Z factors and a generic (non-Pauli) two-qubit operator."""
    tc = tcd
    atol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    rng = np.random.default_rng(7 * n + d)
    params = GOLD[f"tfim_{n}_{d}_params"] if f"tfim_{n}_{d}_params" in GOLD.files else rng.normal(0, 0.5, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    e = 0.0
    for i in range(n):
        e += -1.0 * c.expectation((tc.gates.x(), [i]))
    for i in range(n - 1):
        e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
    psi = dense.run(n, W.hea_b_ops(n, d, params))
    want = W.tfim_energy_dense(psi, n)
    np.testing.assert_allclose(_np(tc, tc.backend.real(e)), want, atol=atol * n)
    if f"tfim_{n}_{d}_energy" in GOLD.files:
        np.testing.assert_allclose(_np(tc, tc.backend.real(e)), GOLD[f"tfim_{n}_{d}_energy"], atol=atol * n)
    for k in range(6):
        ps = [0] * n
        qs = rng.choice(n, 3, replace=False)
        ps[qs[0]] = int(rng.integers(1, 3)); ps[qs[1]] = int(rng.integers(1, 4)); ps[qs[2]] = 3
        got = _np(tc, c.expectation_ps(ps=ps))
        np.testing.assert_allclose(got, dense.pauli_string_expectation(psi, n, ps), atol=atol)
    m = G.random_two_qubit_gate(3) + 0.3 * G.random_two_qubit_gate(4)
    got = _np(tc, c.expectation((m, [n - 1, 1])))
    np.testing.assert_allclose(got, dense.expectation(psi, n, (m, [n - 1, 1])), atol=atol)


@pytest.mark.parametrize("n", [8, 14])
def test_three_qubit_dense_gates_and_heavy_pauli_strings(tcd, n):
    """toffoli / fredkin / any on 3 qubits (reference sgates; tests/test_mpscircuit.py:38-60 uses a 3-qubit
    dense gate) through the plan-time synthesis, and Pauli strings with more than two X/Y factors
    (reference tests/test_mpscircuit.py:176-181: x=[0,2], y=[5,3,1], z=[6,4]) through
    tcmi_apply_pauli_sum + tcmi_vdot; value and gradient against the dense oracle."""
    from scipy.stats import unitary_group

    tc = tcd
    u3 = unitary_group.rvs(8, random_state=21)
    rng = np.random.default_rng(n)
    theta = rng.uniform(0, 2 * np.pi, size=n)

    def build(c, p, ops=None):
        for i in range(n):
            c.h(i)
            c.rx(i, theta=p[i])
            if ops is not None:
                ops += [(G.H, [i]), (G.rx(float(p[i])), [i])]
        c.toffoli(0, 3, 6)
        c.fredkin(7, 2, 4)
        c.any(5, 1, 3, unitary=u3.reshape((2,) * 6))
        c.cz(0, 7)
        for i in range(n - 1):
            c.rzz(i, i + 1, theta=0.3 * (i + 1))
        if ops is not None:
            ops += [(G.TOFFOLI, [0, 3, 6]), (G.FREDKIN, [7, 2, 4]), (u3, [5, 1, 3]), (G.CZ, [0, 7])]
            ops += [(G.rzz(0.3 * (i + 1)), [i, i + 1]) for i in range(n - 1)]
        return c

    ops = []
    c = build(tc.Circuit(n), theta, ops)
    psi = dense.run(n, ops)
    np.testing.assert_allclose(_np(tc, c.state()), psi, atol=TOL[tc.dtypestr] * 2)
    ps = [0] * n
    for q in (0, 2):
        ps[q] = 1
    for q in (5, 3, 1):
        ps[q] = 2
    for q in (6, 4):
        ps[q] = 3
    want = dense.pauli_string_expectation(psi, n, ps)
    got = c.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4])
    np.testing.assert_allclose(_np(tc, got), want, atol=TOL[tc.dtypestr] * 10)
    # mixed light + heavy strings in one fused evaluation, with gradient
    def f(p):
        cc = build(tc.Circuit(n), p)
        e = cc.expectation_ps(x=[0, 2], y=[5, 3, 1], z=[6, 4]) + 0.5 * cc.expectation_ps(z=[0, 1]) \
            - 0.25 * cc.expectation_ps(x=[1, 2, 3, 4])
        return tc.backend.real(e)

    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(theta.astype(np.float64 if tc.rdtypestr == "float64" else np.float32)))
    def f_ref(p):
        o = []
        build(tc.Circuit(n), p, o)
        s = dense.run(n, o)
        ps2 = [0] * n; ps2[0] = ps2[1] = 3
        ps3 = [0] * n
        for q in (1, 2, 3, 4):
            ps3[q] = 1
        return np.real(dense.pauli_string_expectation(s, n, ps) + 0.5 * dense.pauli_string_expectation(s, n, ps2)
                       - 0.25 * dense.pauli_string_expectation(s, n, ps3))
    np.testing.assert_allclose(_np(tc, v), f_ref(theta), atol=TOL[tc.dtypestr] * 10)
    k, eps = 3, 1e-5
    tp, tm = theta.copy(), theta.copy()
    tp[k] += eps
    tm[k] -= eps
    fd = (f_ref(tp) - f_ref(tm)) / (2 * eps)
    np.testing.assert_allclose(_np(tc, g)[k], fd, atol=2e-4 if tc.dtypestr == "complex64" else 1e-7)


def test_hipgraph_replay_matches_eager():
    """GraphedState: the whole wavefunction evaluation captured in a hipGraph and replayed with new
    parameters gives the eager result (state-vector plan and cut plan)."""
    import torch
    import tcmi as tc
    from tcmi.executor import GraphedState

    tc.set_dtype("complex64")
    for n, d, contractor in [(14, 3, "plain"), (20, 4, "greedy")]:
        tc.set_contractor(contractor)
        try:
            rng = np.random.default_rng(n)
            p0 = rng.uniform(0, 6, size=[2 * d, n]).astype(np.float32)
            c = tc.Circuit(n)
            W.hea_b(c, n, d, tc.backend.convert_to_tensor(p0), zz=tc.gates._zz_matrix)
            cc = c._compiled()
            B = 3
            gs = GraphedState(cc, B)
            for trial in range(2):
                pm = torch.from_numpy(rng.uniform(0, 6, size=[B, len(c._params)]).astype(np.float32)).cuda()
                want = cc.state(pm).clone()
                got = gs(pm)
                torch.cuda.synchronize()
                assert float((got[:, : 2**n] - want[:, : 2**n]).abs().max()) < 1e-6
        finally:
            tc.set_contractor("greedy")


def test_sample_expectation_ps_and_readout_error(tcd):
    """reference tests/test_channels.py:158-196 (noise-free circuit, readout error on the measured distribution) and
    agreement of the exact / sampled estimators with expectation_ps."""
    tc = tcd
    c = tc.Circuit(3)
    c.X(0)
    np.testing.assert_allclose(float(c.sample_expectation_ps(z=[0, 1, 2])), -1.0, atol=1e-3)
    readout_error = [[0.9, 0.75], [0.4, 0.7], [0.7, 0.9]]
    np.testing.assert_allclose(float(c.sample_expectation_ps(z=[0, 1, 2], readout_error=readout_error)), 0.04, atol=1e-6)
    v = c.sample_expectation_ps(z=[0, 1, 2], readout_error=tc.backend.convert_to_tensor(np.array(readout_error)))
    np.testing.assert_allclose(float(v), 0.04, atol=1e-6)
    p = c.readouterror_bs(readout_error, c.probability())
    np.testing.assert_allclose(float(p.sum()), 1.0, atol=1e-6)
    r = c.sample(batch=2000, allow_state=True, readout_error=readout_error, format="sample_bin",
                 status=np.random.default_rng(0).uniform(size=2000))
    bits = tc.backend.numpy(r).astype(float)
    np.testing.assert_allclose(bits.mean(0), [0.75, 0.6, 0.3], atol=0.04)      # p(read 1) per qubit for |100>

    n = 6
    rng = np.random.default_rng(3)
    c = tc.Circuit(n)
    for i in range(n):
        c.ry(i, theta=float(rng.uniform(0, 3)))
    for i in range(n - 1):
        c.cnot(i, i + 1)
    for i in range(n):
        c.rx(i, theta=float(rng.uniform(0, 3)))
    for kw in (dict(z=[0, 3]), dict(x=[1], z=[4]), dict(y=[2], x=[5], z=[0])):
        exact = float(tc.backend.real(c.expectation_ps(**kw)))
        np.testing.assert_allclose(float(c.sample_expectation_ps(**kw)), exact, atol=2e-5)
        est = float(c.sample_expectation_ps(shots=20000, status=rng.uniform(size=20000), **kw))
        assert abs(est - exact) < 0.03


@pytest.mark.parametrize("n", [14, 17])
def test_many_random_pauli_strings_in_one_measurement(tcd, n):
    """One fused measurement of 60 random strings (0-2 X / Y factors, 0-3 Z factors, weights) against the dense oracle:
    every path of the measurement kernels -- Z-only strings through the Walsh transform, single-X strings, the general
    pair loop for Y factors, two X / Y factors and register Z factors -- and each string on its own."""
    tc = tcd
    atol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    rng = np.random.default_rng(100 + n)
    d = 2
    params = rng.normal(0, 0.7, [2 * d, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    psi = dense.run(n, W.hea_b_ops(n, d, params))
    strings, weights = [], []
    for k in range(60):
        ps = [0] * n
        qs = rng.permutation(n)
        nx, nz = int(rng.integers(0, 3)), int(rng.integers(0, 4))
        for q in qs[:nx]:
            ps[q] = int(rng.integers(1, 3))
        for q in qs[nx:nx + nz]:
            ps[q] = 3
        if not any(ps):
            ps[int(qs[0])] = 3
        strings.append(ps)
        weights.append(float(rng.normal()))
    e = 0.0
    for w, ps in zip(weights, strings):
        e = e + w * c.expectation_ps(ps=ps)
    want = sum(w * dense.pauli_string_expectation(psi, n, ps) for w, ps in zip(weights, strings))
    np.testing.assert_allclose(_np(tc, e), want, atol=atol * 10)
    for ps in strings[:12]:
        np.testing.assert_allclose(_np(tc, c.expectation_ps(ps=ps)), dense.pauli_string_expectation(psi, n, ps), atol=atol)
