"""Small API rows of the reference's circuit front end closed in round 4, each with the reference's own known answer:
post-selection (tests/test_circuit.py:528-536), matrix-shaped inputs and Circuit.matrix (tests/test_circuit.py:539-547,
circuit.py:744-769), mpo_expectation (tests/test_templates.py:191-211), templates.blocks.example_block
(templates/blocks.py:146-185)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, workloads as W  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hea_golden.npz"))


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(request.param)
    yield tc
    tc.set_dtype("complex64")


def test_postselection_kat(tcd):
    """reference tests/test_circuit.py:528-536: s[3].real == 0.5 (the state is not renormalised)."""
    tc = tcd
    c = tc.Circuit(3)
    c.H(1)
    c.H(2)
    r = c.mid_measurement(1, 1)
    c.mid_measurement(2, 1)
    s = tc.backend.numpy(c.wavefunction())
    np.testing.assert_allclose(s[3].real, 0.5, atol=1e-6)
    np.testing.assert_allclose(np.abs(s).sum(), 0.5, atol=1e-6)       # every other amplitude is projected out
    assert int(tc.backend.numpy(r)) == 1
    assert len(c.to_qir()) == 2                                        # like the reference: not recorded in the QIR
    # post-selection inside a longer circuit, against the dense oracle (projector as a gate)
    n = 12
    rng = np.random.default_rng(3)
    c2, ops = tc.Circuit(n), []
    for i in range(n):
        c2.h(i); ops.append((G.H, [i]))
    for i in range(n - 1):
        th = float(rng.uniform(0, 6)); c2.rzz(i, i + 1, theta=th); ops.append((G.rzz(th), [i, i + 1]))
    c2.post_select(4, keep=1); ops.append((np.diag([0.0, 1.0]), [4]))
    for i in range(n):
        th = float(rng.uniform(0, 6)); c2.rx(i, theta=th); ops.append((G.rx(th), [i]))
    c2.mid_measure(9, keep=0); ops.append((np.diag([1.0, 0.0]), [9]))
    ref = dense.run(n, ops)
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    assert np.abs(tc.backend.numpy(c2.wavefunction()) - ref).max() < tol


def test_matrix_inputs_and_circuit_matrix_kat(tcd):
    """reference tests/test_circuit.py:539-547: Circuit(2, inputs=eye(4)); X(0); Y(1): wavefunction.reshape(4, 4) = X (x) Y."""
    tc = tcd
    c = tc.Circuit(2, inputs=np.eye(4))
    c.X(0)
    c.Y(1)
    answer = np.kron(G.X, G.Y)
    np.testing.assert_allclose(tc.backend.numpy(c.wavefunction()).reshape(4, 4), answer, atol=1e-4)
    # Circuit.matrix / get_quoperator: the unitary of a parametrised circuit against the dense oracle, column by column
    n = 8
    rng = np.random.default_rng(1)
    pr = rng.uniform(0, 6, [4, n])
    c3 = tc.Circuit(n)
    W.hea_b(c3, n, 2, tc.backend.convert_to_tensor(pr, dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    u = tc.backend.numpy(c3.matrix())
    ops = W.hea_b_ops(n, 2, pr)
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    for col in (0, 5, 77, 255):
        e = np.zeros(2**n, dtype=np.complex128)
        e[col] = 1.0
        assert np.abs(u[:, col] - dense.run(n, ops, inputs=e)).max() < tol
    assert np.abs(u.conj().T @ u - np.eye(2**n)).max() < 50 * tol
    np.testing.assert_allclose(tc.backend.numpy(c3.get_quoperator().eval_matrix()), u)
    # open input legs that are not a square matrix: 3 circuit qubits, 1 open leg
    inp = (rng.normal(size=16) + 1j * rng.normal(size=16))
    c4 = tc.Circuit(3, inputs=inp)
    c4.h(0); c4.cnot(0, 2); c4.rx(1, theta=0.3)
    got = tc.backend.numpy(c4.wavefunction()).reshape(8, 2)
    for j in range(2):
        want = dense.run(3, [(G.H, [0]), (G.CNOT, [0, 2]), (G.rx(0.3), [1])], inputs=inp.reshape(8, 2)[:, j])
        assert np.abs(got[:, j] - want).max() < tol * 10


def test_circuit_matrix_is_differentiable(tcd):
    """reference circuit.py:744-769: ``matrix()`` is an ordinary differentiable contraction.  d/dtheta of
    Re(U[0, 0] + 2 U[5, 3]) for U = the unitary of an rx / cnot / rzz circuit, against central differences of the dense
    oracle's columns -- the route through matrix-shaped inputs must not cut the tape."""
    tc = tcd
    n = 3
    th0 = np.array([0.4, 1.1, -0.7])

    def ops_of(th):
        return [(G.rx(th[0]), [0]), (G.CNOT, [0, 1]), (G.rx(th[1]), [2]), (G.rzz(th[2]), [1, 2]), (G.H, [1])]

    def oracle_f(th):
        u = np.stack([dense.run(n, ops_of(th), inputs=np.eye(2**n)[:, j]) for j in range(2**n)], axis=1)
        return float((u[0, 0] + 2 * u[5, 3]).real)

    def f(th):
        c = tc.Circuit(n)
        c.rx(0, theta=th[0]); c.cnot(0, 1); c.rx(2, theta=th[1]); c.rzz(1, 2, theta=th[2]); c.h(1)
        u = c.matrix()
        return tc.backend.real(u[0, 0] + 2 * u[5, 3])

    v, g = tc.backend.value_and_grad(f)(tc.backend.convert_to_tensor(th0, dtype=tc.rdtypestr))
    tol = 2e-4 if tc.dtypestr == "complex64" else 1e-7
    assert abs(float(v) - oracle_f(th0)) < tol
    eps = 1e-6
    for k in range(3):
        d = np.zeros(3); d[k] = eps
        fd = (oracle_f(th0 + d) - oracle_f(th0 - d)) / (2 * eps)
        assert abs(float(g[k]) - fd) < tol, (k, float(g[k]), fd)
    assert np.abs(tc.backend.numpy(g)).max() > 1e-2


def test_mpo_expectation_kat(tcd):
    """reference tests/test_templates.py:191-211 (the MPO branch): value 0.84147, gradient 0.54032 (atol 1e-4)."""
    tc = tcd
    mpo = tc.quantum.QuOperator.from_local_tensor(tc.gates._x_matrix, [2, 2], [0])

    def f(theta):
        c = tc.Circuit(2)
        c.ry(0, theta=theta)
        c.H(1)
        return tc.templates.measurements.operator_expectation(c, mpo)

    v, g = tc.backend.jit(tc.backend.value_and_grad(f))(tc.backend.ones([], dtype=tc.rdtypestr))
    np.testing.assert_allclose(tc.backend.numpy(v), 0.84147, atol=1e-4)
    np.testing.assert_allclose(tc.backend.numpy(g), 0.54032, atol=1e-4)
    # a two-site local tensor on non-adjacent sites, value against the dense oracle and the operator's matrix
    n = 5
    rng = np.random.default_rng(2)
    h = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
    h = h + h.conj().T
    op = tc.quantum.QuOperator.from_local_tensor(h, [2] * n, [3, 1])
    pr = rng.uniform(0, 6, [4, n])
    c = tc.Circuit(n)
    W.hea_b(c, n, 2, tc.backend.convert_to_tensor(pr, dtype=tc.rdtypestr), zz=tc.gates._zz_matrix)
    psi = dense.run(n, W.hea_b_ops(n, 2, pr))
    want = dense.expectation(psi, n, (h, [3, 1])).real
    got = float(tc.backend.numpy(tc.templates.measurements.mpo_expectation(c, op)))
    assert abs(got - want) < (1e-4 if tc.dtypestr == "complex64" else 1e-9)
    m = tc.backend.numpy(op.eval_matrix())
    assert abs(np.vdot(psi, m @ psi).real - want) < 1e-4


def test_example_block_is_the_hea_b_ansatz(tcd):
    """templates.blocks.example_block (reference blocks.py:146-185) reproduces the golden HEA-B states; with
    is_split=True (max_singular_values 2 keeps the ZZ gate exactly) too."""
    tc = tcd
    for n, d in ((8, 3), (10, 4)):
        params = GOLD[f"hea_b_{n}_{d}_params"]
        ref = GOLD[f"hea_b_{n}_{d}_state"]
        tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
        for split in (False, True):
            c = tc.templates.blocks.example_block(tc.Circuit(n), tc.backend.convert_to_tensor(params.reshape(-1), dtype=tc.rdtypestr),
                                                  nlayers=d, is_split=split)
            assert np.abs(tc.backend.numpy(c.wavefunction()) - ref).max() < tol, (n, d, split)


def test_big_diagonals_in_the_network_route(tcd):
    """Reference basecircuit.py:295-369 (mpo= / diagonal= gates as MPO nodes): cmz on 10 qubits, a 5-control
    multicontrol and a 7-qubit diagonal in the contraction engine -- amplitude, light-cone expectation and the sliced
    DistributedContractor value_and_grad -- against the state-vector route; no network node exceeds 2^10 entries
    (the gates' matrices would be 2^20 / 2^12 / 2^14)."""
    import torch

    tc = tcd
    from tcmi import tn

    n = 12
    rng = np.random.default_rng(11)
    dv = np.exp(1j * rng.uniform(0, 6, 128))
    u = G.rx(0.9) @ G.rz(0.4)

    def build(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for i in range(n):
            c.rx(i, theta=p[0, i])
        c.cmz(0, 1, 2, 3, 4, 6, 7, 8, 10, 11)
        c.multicontrol(9, 5, 3, 1, 0, 7, ctrl=[1, 1, 0, 1, 1], unitary=u)
        for i in range(n):
            c.ry(i, theta=p[1, i])
        c.diagonal(2, 4, 5, 6, 8, 9, 11, diag=dv)
        for i in range(n):
            c.rx(i, theta=p[2, i])
        return c

    p = tc.backend.convert_to_tensor(rng.uniform(0, 6, [3, n]), dtype=tc.rdtypestr)
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    c = build(p)
    psi = tc.backend.numpy(c.wavefunction())
    bits = "101100101101"
    nodes = c.amplitude_before(bits)
    assert max(nd.tensor.numel() for nd in nodes) <= 2 ** 10
    a = tn.contract_nodes(nodes).tensor
    assert abs(complex(tc.backend.numpy(a)) - psi[int(bits, 2)]) < tol
    ref = float(tc.backend.numpy(tc.backend.real(c.expectation_ps(z=[5], x=[8]))))
    e = c.expectation((tc.gates.z(), [5]), (tc.gates.x(), [8]), enable_lightcone=True)
    assert abs(float(tc.backend.numpy(tc.backend.real(e))) - ref) < tol * 4

    def nodes_fn(q):
        return build(q).expectation_before((tc.gates.z(), [5]), (tc.gates.x(), [8]), reuse=False)

    dc = tc.experimental.DistributedContractor(nodes_fn, p, {"slicing_reconf_opts": {"target_size": 2 ** 9}, "max_repeats": 4})
    v, g = dc.value_and_grad(p)
    assert abs(float(tc.backend.numpy(v)) - ref) < tol * 4

    def f(q):
        return tc.backend.real(build(q).expectation_ps(z=[5], x=[8]))

    _, g2 = tc.backend.value_and_grad(f)(p)
    assert np.abs(tc.backend.numpy(g) - tc.backend.numpy(g2)).max() < tol * 20


def test_mps_shaped_input_states_kats(tcd):
    """reference tests/test_circuit.py:470-495 (``Circuit(n, tensors=[t1, t2, t3])``: GHZ site tensors -> GHZ state) and
    :693-704 (``Circuit(2, mps_inputs=c.quvector())`` and ``replace_mps_inputs``: X X|00> = |00>); an MPSCircuit's own
    tensors as the input of a Circuit give the state of the same gates applied to the dense state (oracle.dense); the
    gradient with respect to a site tensor flows through the chain contraction (tcmi_cgemm's backward rule)."""
    tc = tcd
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    n = 3
    t1 = np.zeros([1, 2, 2], dtype=np.complex64); t1[0, 0, 0] = 1; t1[0, 1, 1] = 1
    t2 = np.zeros([2, 2, 2], dtype=np.complex64); t2[0, 0, 0] = 1; t2[1, 1, 1] = 1
    t3 = np.zeros([2, 2, 1], dtype=np.complex64); t3[0, 0, 0] = 1 / np.sqrt(2); t3[1, 1, 0] = 1 / np.sqrt(2)
    c = tc.Circuit(n, tensors=[t1, t2, t3])
    ghz = np.zeros(8); ghz[0] = ghz[7] = 1 / np.sqrt(2)
    np.testing.assert_allclose(tc.backend.numpy(c.wavefunction()), ghz, atol=1e-5)
    # reference test_circuit_add_demo
    c = tc.Circuit(2); c.x(0)
    c2 = tc.Circuit(2, mps_inputs=c.quvector()); c2.X(0)
    np.testing.assert_allclose(tc.backend.numpy(c2.wavefunction()), np.array([1.0, 0, 0, 0]), atol=1e-4)
    c3 = tc.Circuit(2); c3.X(0)
    c3.replace_mps_inputs(c.quvector())
    np.testing.assert_allclose(tc.backend.numpy(c3.wavefunction()), np.array([1.0, 0, 0, 0]), atol=1e-4)
    # a random MPS (bond dimension 4) under gates, against the dense oracle on the contracted chain
    n = 8
    rng = np.random.default_rng(8)
    dims = [1, 2, 4, 4, 4, 4, 4, 2, 1]
    ts = [rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1])) for i in range(n)]
    dense_in = ts[0].reshape(2, -1)
    for t in ts[1:]:
        dense_in = (dense_in @ t.reshape(t.shape[0], -1)).reshape(-1, t.shape[2])
    dense_in = dense_in.reshape(-1)
    nrm = np.linalg.norm(dense_in)
    ts[0] = ts[0] / nrm
    dense_in = dense_in / nrm
    cm = tc.Circuit(n, tensors=ts)
    cm.h(0); cm.cnot(0, 5); cm.rx(3, theta=0.7); cm.rzz(2, 6, theta=-0.4)
    want = dense.run(n, [(G.H, [0]), (G.CNOT, [0, 5]), (G.rx(0.7), [3]), (G.rzz(-0.4), [2, 6])], inputs=dense_in)
    assert np.abs(tc.backend.numpy(cm.wavefunction()) - want).max() < 10 * tol
    with pytest.raises(ValueError):
        tc.Circuit(5, tensors=ts)
    # differentiable: d/ds of <Z_3> for the input MPS with site 2 scaled by (1 + s A) -- against central differences of
    # the oracle
    A = rng.normal(size=ts[2].shape) + 1j * rng.normal(size=ts[2].shape)

    def z3_oracle(s):
        tt = list(ts); tt[2] = ts[2] + s * A
        v = tt[0].reshape(2, -1)
        for t in tt[1:]:
            v = (v @ t.reshape(t.shape[0], -1)).reshape(-1, t.shape[2])
        psi = dense.run(n, [(G.rx(0.7), [3])], inputs=v.reshape(-1))
        return float(np.real(dense.expectation(psi, n, (G.Z, [3]))))

    def z3(s):
        import torch

        cdt = getattr(torch, tc.dtypestr)
        tt = [torch.as_tensor(t).to(device=s.device, dtype=cdt) for t in ts]
        tt[2] = tt[2] + tc.backend.cast(s, tc.dtypestr) * torch.as_tensor(A).to(device=s.device, dtype=cdt)
        cc = tc.Circuit(n, tensors=tt)
        cc.rx(3, theta=0.7)
        return tc.backend.real(cc.expectation((tc.gates.z(), [3])))

    import torch

    s0 = torch.zeros((), dtype=getattr(torch, tc.rdtypestr), device="cuda")
    v, g = tc.backend.value_and_grad(z3)(s0)
    eps = 1e-5
    fd = (z3_oracle(eps) - z3_oracle(-eps)) / (2 * eps)
    gt = 2e-3 if tc.dtypestr == "complex64" else 1e-7
    assert abs(float(v) - z3_oracle(0.0)) < 10 * tol and abs(float(g) - fd) < gt, (float(g), fd)


def test_dense_any_gates_on_six_and_seven_qubits(tcd):
    """reference gates.py:866-890 (``any`` takes a unitary of any size): Haar-random 6- and 7-qubit gates on scattered
    qubits, between ordinary gates, against the dense oracle; nine qubits raise (a 2^18-element matrix is a state-sized
    operator, not a gate)."""
    from scipy.stats import unitary_group

    tc = tcd
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    n = 14
    for k, qs in ((6, [1, 3, 4, 8, 12, 13]), (7, [0, 2, 5, 6, 9, 10, 11])):
        u = unitary_group.rvs(2**k, random_state=40 + k)
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        c.rx(3, theta=0.3)
        c.any(*qs, unitary=u)
        c.cnot(qs[0], (qs[0] + 7) % n if (qs[0] + 7) % n not in (qs[0],) else 1)
        ops = [(G.H, [i]) for i in range(n)] + [(G.rx(0.3), [3]), (u, qs), (G.CNOT, [qs[0], (qs[0] + 7) % n])]
        want = dense.run(n, ops)
        got = tc.backend.numpy(c.wavefunction())
        assert np.abs(got - want).max() < tol, (k, np.abs(got - want).max())
    with pytest.raises(NotImplementedError):
        tc.Circuit(10).any(*range(9), unitary=np.eye(2**9))


def test_vmap_over_circuits_with_matrix_shaped_and_batched_inputs(tcd):
    """backend.vmap over a circuit whose ``inputs`` carry open legs (reference circuit.py:44-131: inputs with more legs than
    qubits are a batch of columns), batched in the parameters, in the inputs, and in both: every batch member equals the
    un-vmapped call, column by column against the dense oracle (StateFn.vmap lifts both to one batch of B x columns states)."""
    tc = tcd
    import torch

    n, B = 4, 3
    rng = np.random.default_rng(12)
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    th = rng.uniform(0, 6, [B, 2])
    mats = rng.normal(size=[B, 2**n, 2]) + 1j * rng.normal(size=[B, 2**n, 2])          # two open columns per member

    def f(t, m):
        c = tc.Circuit(n, inputs=m)
        c.h(0); c.rx(1, theta=t[0]); c.cnot(1, 2); c.rzz(2, 3, theta=t[1])
        return c.wavefunction()

    def ops(t):
        return [(G.H, [0]), (G.rx(t[0]), [1]), (G.CNOT, [1, 2]), (G.rzz(t[1]), [2, 3])]

    K = tc.backend
    tt = K.convert_to_tensor(th, dtype=tc.rdtypestr)
    mm = K.cast(K.convert_to_tensor(mats), tc.dtypestr)
    both = K.numpy(K.vmap(f, vectorized_argnums=(0, 1))(tt, mm.reshape(B, -1))).reshape(B, 2**n, 2)
    only_t = K.numpy(K.vmap(lambda t: f(t, mm[0].reshape(-1)))(tt)).reshape(B, 2**n, 2)
    only_m = K.numpy(K.vmap(lambda m: f(tt[0], m))(mm.reshape(B, -1))).reshape(B, 2**n, 2)
    for b in range(B):
        for j in range(2):
            assert np.abs(both[b, :, j] - dense.run(n, ops(th[b]), inputs=mats[b, :, j])).max() < tol * 20
            assert np.abs(only_t[b, :, j] - dense.run(n, ops(th[b]), inputs=mats[0, :, j])).max() < tol * 20
            assert np.abs(only_m[b, :, j] - dense.run(n, ops(th[0]), inputs=mats[b, :, j])).max() < tol * 20
