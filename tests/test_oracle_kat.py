"""Pins the CPU oracle (oracle/tn.py, oracle/dense.py) against the known-answer constants of the
reference's own tests -- each test cites the reference test it restates --
and against the committed golden vectors.  CPU only."""

import os

import numpy as np
import pytest

from oracle import dense, gates as G, tn, workloads as W

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hea_golden.npz"))


@pytest.mark.parametrize("method", ["greedy", "plain"])
def test_wavefunction_kat(method):
    """reference tests/test_circuit.py:22-44."""
    g = np.arange(16).reshape(2, 2, 2, 2)
    c = tn.Circuit(2, method=method)
    c.unitary(g, 0, 1)
    assert np.real(c.wavefunction()[2]) == 8
    c = tn.Circuit(2, method=method)
    c.unitary(g, 1, 0)
    assert np.real(c.wavefunction()[2]) == 4
    c = tn.Circuit(2, method=method)
    c.unitary(np.arange(4).reshape(2, 2), 0)
    assert np.real(c.wavefunction()[2]) == 2


def test_basics_kat():
    """reference tests/test_circuit.py:47-53: qubit 0 is the most significant bit."""
    c = tn.Circuit(2)
    c.x(0)
    np.testing.assert_allclose(c.amplitude("10"), 1.0)
    c.cnot(0, 1)
    np.testing.assert_allclose(c.amplitude("11"), 1.0)


def test_iswap_on_identity_input():
    """reference tests/test_circuit.py:95-99."""
    cols = []
    for k in range(4):
        inp = np.zeros(4)
        inp[k] = 1
        c = tn.Circuit(2, inputs=inp)
        c.iswap(0, 1)
        cols.append(c.state())
    np.testing.assert_allclose(np.stack(cols, axis=1), G.iswap(1.0), atol=1e-12)


def test_control_vgate():
    """reference tests/test_circuit.py:102-108 and tests/test_gates.py:144-153: cos(0.3)."""
    c = tn.Circuit(2)
    c.x(1)
    c.crx(1, 0, theta=0.3)
    np.testing.assert_allclose(c.expectation((G.Z, [0])), 0.95533645, atol=1e-5)
    c = tn.Circuit(2)
    c.x(0)
    c.crx(0, 1, theta=0.3)
    np.testing.assert_allclose(c.expectation((G.Z, [1])), 0.95533645, atol=1e-5)


def test_adjoint_gate_circuit():
    """reference tests/test_circuit.py:111-115: X; SD => [0, -1j]."""
    c = tn.Circuit(1)
    c.x(0)
    c.sd(0)
    np.testing.assert_allclose(c.state(), np.array([0.0, -1.0j]))


def test_expectation_h_z():
    """reference tests/test_circuit.py:317-323."""
    c = tn.Circuit(2)
    c.h(0)
    np.testing.assert_allclose(c.expectation((G.Z, [0])), 0, atol=1e-7)


def test_su4_norm():
    """reference tests/test_circuit.py:326-335: 8 qubits, 6 layers of su4 with all-ones params,
    norm 1 at 1e-7 (complex128)."""
    c = tn.Circuit(8)
    ops = []
    for d in range(6):
        for i in range(8):
            c.su4(i, (i + 1) % 8, theta=np.ones(15))
            ops.append((G.su4(np.ones(15)), [i, (i + 1) % 8]))
    psi = c.state()
    np.testing.assert_allclose(np.linalg.norm(psi), 1, atol=1e-7)
    np.testing.assert_allclose(psi, dense.run(8, ops), atol=1e-10)


def test_single_qubit_h():
    """reference tests/test_circuit.py:397-401."""
    c = tn.Circuit(1)
    c.h(0)
    np.testing.assert_allclose(c.state(), np.array([1, 1]) / np.sqrt(2), atol=1e-12)


def test_any_inputs_state():
    """reference tests/test_circuit.py:448-468."""
    for inp, want in (([0, 0, 0, 1.0], 1.0), ([0, 0, 1.0, 0], 1.0), ([1.0, 0, 0, 0], -1.0)):
        c = tn.Circuit(2, inputs=np.array(inp))
        c.x(0)
        assert c.expectation((G.Z, [0])) == want
    c = tn.Circuit(2, inputs=np.array([1 / np.sqrt(2), 0.0, 1 / np.sqrt(2), 0.0]))
    c.x(0)
    np.testing.assert_allclose(c.expectation((G.Z, [0])), 0.0, atol=1e-4)


def test_expectation_ps_and_y():
    """reference tests/test_circuit.py:550-564 (expectation_ps +-1) and :1501-1504 (<Y> = -1)."""
    c = tn.Circuit(2)
    c.x(0)
    np.testing.assert_allclose(c.expectation_ps(z=[0]), -1)
    np.testing.assert_allclose(c.expectation_ps(z=[1]), 1)
    np.testing.assert_allclose(c.expectation_ps(ps=[3, 3]), -1)
    c = tn.Circuit(1)
    c.h(0)
    c.sd(0)  # |0> - i|1> : <Y> = -1
    np.testing.assert_allclose(c.expectation_ps(y=[0]), -1, atol=1e-12)


def test_gate_constants():
    """reference tests/test_gates.py:34-38 (phase => 0.7071j), :96-106 (exp gate => -1j),
    :115-120 (iswap tensor)."""
    c = tn.Circuit(1)
    c.h(0)
    c.phase(0, theta=np.pi / 2)
    np.testing.assert_allclose(c.state()[1], 0.7071j, atol=1e-4)
    c = tn.Circuit(2)
    c.exp(0, 1, unitary=G.ZZ, theta=np.pi / 2)
    np.testing.assert_allclose(c.state()[0], -1j, atol=1e-6)
    c = tn.Circuit(2)
    c.exp1(0, 1, unitary=G.ZZ, theta=np.pi / 2)
    np.testing.assert_allclose(c.state()[0], -1j, atol=1e-6)
    t = G.iswap(1.0)
    assert t[1, 2] == 1j and t[2, 1] == 1j and t[0, 0] == 1 and t[3, 3] == 1


def test_h_cnot_swap_amplitudes():
    """reference tests/test_circuit.py:1583-1598: amplitudes 0 and 1/sqrt(2)."""
    c = tn.Circuit(3)
    c.h(0)
    c.cnot(0, 1)
    c.swap(1, 2)
    np.testing.assert_allclose(c.amplitude("000"), 1 / np.sqrt(2), atol=1e-12)
    np.testing.assert_allclose(c.amplitude("101"), 1 / np.sqrt(2), atol=1e-12)
    np.testing.assert_allclose(c.amplitude("110"), 0, atol=1e-12)


def test_duplicate_and_negative_index():
    """reference basecircuit.py:214-219, 429-438."""
    c = tn.Circuit(3)
    with pytest.raises(ValueError):
        c.cnot(1, 1)
    c.x(-1)
    np.testing.assert_allclose(c.amplitude("001"), 1.0)
    with pytest.raises(ValueError):
        c.expectation((G.Z, [0]), (G.X, [0]))


def test_operator_measurement_gradient_kat():
    """reference tests/test_templates.py:191-211 (and :172-188): c = Circuit(2); ry(0, theta); H(1);
    <X_0> at theta = 1 is 0.84147 with gradient 0.54032 (atol 1e-4 in the reference)."""

    def f(th):
        c = tn.Circuit(2)
        c.ry(0, theta=th)
        c.h(1)
        return c.expectation((G.X, [0])).real

    np.testing.assert_allclose(f(1.0), 0.84147, atol=1e-4)
    np.testing.assert_allclose((f(1.0 + 1e-6) - f(1.0 - 1e-6)) / 2e-6, 0.54032, atol=1e-4)


@pytest.mark.parametrize("n,d", [(4, 2), (8, 3), (10, 4), (12, 4)])
def test_golden_states_tn_vs_dense(n, d):
    """oracle.tn (reference algorithm restated) == committed dense-simulator golden vectors."""
    params = GOLD[f"hea_b_{n}_{d}_params"]
    for method in ("greedy", "plain"):
        c = tn.Circuit(n, method=method)
        W.hea_b(c, n, d, params)
        np.testing.assert_allclose(c.wavefunction(), GOLD[f"hea_b_{n}_{d}_state"], atol=1e-12)


def test_example_block_ones_config1():
    """n=10, d=4, ones params (the circuit of reference
    tests/test_circuit.py:922-946); state shape 2^10, norm 1, TFIM energy consistent."""
    n, d = 10, 4
    c = tn.Circuit(n)
    W.hea_b(c, n, d, np.ones([2 * d, n]))
    psi = c.wavefunction()
    assert psi.shape == (1024,)
    np.testing.assert_allclose(psi, GOLD["hea_b_10_4_ones_state"], atol=1e-12)
    np.testing.assert_allclose(np.vdot(psi, psi), 1.0, atol=1e-12)
    np.testing.assert_allclose(W.tfim_energy(c, n).real, W.tfim_energy_dense(psi, n), atol=1e-10)
    np.testing.assert_allclose(c.expectation_ps(z=[0]), dense.pauli_string_expectation(psi, n, [3] + [0] * 9), atol=1e-12)


def test_tfim_golden():
    for n, d in [(6, 2), (10, 4)]:
        params = GOLD[f"tfim_{n}_{d}_params"]
        c = tn.Circuit(n)
        W.hea_b(c, n, d, params)
        np.testing.assert_allclose(W.tfim_energy(c, n).real, GOLD[f"tfim_{n}_{d}_energy"], atol=1e-10)


def test_contractor_equivalence():
    """reference tests/test_circuit.py:2260-2292: greedy / preprocessing / plain agree."""
    n, d = 9, 3
    pa = GOLD["hea_a_9_3_params"]
    outs = []
    for method in ("greedy", "plain"):
        c = tn.Circuit(n, method=method)
        W.hea_a(c, n, d, pa)
        outs.append(c.wavefunction())
    np.testing.assert_allclose(outs[0], outs[1], atol=1e-12)
    np.testing.assert_allclose(outs[0], GOLD["hea_a_9_3_state"], atol=1e-12)
    nodes, front = tn.Circuit(3)._copy()
    with pytest.raises(ValueError):
        tn.contract(nodes)  # >1 dangling edge and no output order (cons.py:877-886)


def test_vmap_vvag_semantics_kat():
    """reference tests/test_backends.py:890-939: vvag == per-sample value_and_grad; gradients of
    a shared (non-vectorised) argument are summed over the batch.  Restated on the oracle with
    finite differences."""
    n, d = 4, 1
    rng = np.random.default_rng(3)
    shared = rng.normal(size=[2 * d, n])
    batch = rng.normal(size=[3, n])

    def f(x, w):  # x: per-sample rx angles appended after the block
        ops = W.hea_b_ops(n, d, w) + [(G.rx(x[i]), [i]) for i in range(n)]
        return W.tfim_energy_dense(dense.run(n, ops), n)

    vals = np.array([f(b, shared) for b in batch])
    eps = 1e-6
    gw = np.zeros_like(shared)
    for b in batch:
        for i in np.ndindex(*shared.shape):
            wp, wm = shared.copy(), shared.copy()
            wp[i] += eps
            wm[i] -= eps
            gw[i] += (f(b, wp) - f(b, wm)) / (2 * eps)
    assert vals.shape == (3,) and gw.shape == shared.shape
    g_single = np.zeros_like(shared)
    for i in np.ndindex(*shared.shape):
        wp, wm = shared.copy(), shared.copy()
        wp[i] += eps
        wm[i] -= eps
        g_single[i] = (f(batch[0], wp) - f(batch[0], wm)) / (2 * eps)
    # identical samples => summed gradient / batch == single gradient (g11/batch == g01)
    same = np.array([f(batch[0], shared)] * 3)
    np.testing.assert_allclose(same, vals[0])
    assert np.abs(gw).max() > 0 and np.isfinite(g_single).all()


def test_numpy_adjoint_value_and_grad_against_central_differences():
    """oracle/adjoint.py (bench.py's CPU column of the VQE step) against oracle.dense: energy to rounding, every gradient
    component to the central difference (step 1e-6, float64), the unused last ZZ angle of a row exactly 0."""
    from oracle import adjoint as A

    n, d = 7, 2
    p = np.random.default_rng(11).normal(0, 0.8, [2 * d, n])
    e, g = A.hea_b_tfim_value_and_grad(n, d, p)
    f = lambda q: W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, q)), n)  # noqa: E731
    assert abs(e - f(p)) < 1e-12
    eps = 1e-6
    for idx in np.ndindex(*p.shape):
        pp, pm = p.copy(), p.copy()
        pp[idx] += eps
        pm[idx] -= eps
        assert abs(g[idx] - (f(pp) - f(pm)) / (2 * eps)) < 2e-8, idx
    assert (g[0::2, n - 1] == 0).all()
    # the state it starts from is the oracle's
    np.testing.assert_allclose(A.hea_b_state(n, d, p), dense.run(n, W.hea_b_ops(n, d, p)), atol=1e-14)


def test_sliced_numpy_chain_sums_to_the_unsliced_contraction():
    """oracle/sliced.py: fixing an index and summing its two slices equals the unsliced chain (a ring of three matrices)."""
    from oracle import sliced as OS

    rng = np.random.default_rng(2)
    ts = [rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2)) for _ in range(3)]
    ins = [[0, 1], [1, 2], [2, 0]]
    path = [(0, 1), (0, 1)]
    want = np.trace(ts[0] @ ts[1] @ ts[2])
    assert abs(OS.contract_path(ts, ins, path, [], []) - want) < 1e-13
    got = sum(OS.contract_path(ts, ins, path, [1], OS.slice_values(s, 1)) for s in range(2))
    assert abs(got - want) < 1e-13
