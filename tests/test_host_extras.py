"""Host-side logic that needs no GPU: Kraus channels, super-gates, DMCircuit recording on the doubled
circuit, sample format conversion, Pauli-sum bookkeeping, jit dispatch."""

import numpy as np
import pytest

import tcmi as tc
from oracle import dm as odm, gates as OG


def test_channels_are_trace_preserving_and_super_gate_matches_oracle():
    ch = tc.channels
    for ks in (ch.depolarizingchannel(0.1, 0.2, 0.3), ch.amplitudedampingchannel(0.4, 0.7),
               ch.phasedampingchannel(0.6), ch.resetchannel(), ch.generaldepolarizingchannel(0.05),
               ch.isotropicdepolarizingchannel(0.3)):
        ch.kraus_identity_check(ks)
        sup = ch.kraus_to_super_gate(ks)
        rng = np.random.default_rng(0)
        a = rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2))
        rho = a @ a.conj().T
        rho /= np.trace(rho)
        want = sum(np.asarray(k.tensor) @ rho @ np.asarray(k.tensor).conj().T for k in ks)
        got = (sup @ rho.reshape(-1)).reshape(2, 2)          # vec index = (ket, bra)
        np.testing.assert_allclose(got, want, atol=1e-6)
    # multi-qubit depolarizing channels (reference channels.py:103-230): 4^n Kraus operators, same as the oracle's
    from oracle import channels as OC

    with pytest.raises(ValueError):
        ch.generaldepolarizingchannel(0.1, num_qubits=2)        # 1 - 15 * 0.1 < 0, as the reference
    for got, want in ((ch.generaldepolarizingchannel(0.02, num_qubits=2), OC.generaldepolarizing(0.02, 2)),
                      (ch.isotropicdepolarizingchannel(0.3, 2), OC.isotropicdepolarizing(0.3, 2)),
                      (ch.generaldepolarizingchannel([0.004 * k for k in range(1, 16)], 2),
                       OC.generaldepolarizing([0.004 * k for k in range(1, 16)], 2))):
        assert len(got) == 16
        ch.kraus_identity_check(got)
        for a, b in zip(got, want):
            np.testing.assert_allclose(np.asarray(a.tensor).reshape(4, 4), b, atol=1e-7)


def test_dmcircuit_records_ket_and_conjugated_bra_gates():
    n = 3
    c = tc.DMCircuit(n)
    c.h(0)
    c.cnot(0, 1)
    c.rx(2, theta=0.3)
    c.s(1)
    c.depolarizing(1, px=0.1, py=0.05, pz=0.02)
    ops = c._c._ops
    assert [o.qubits for o in ops] == [(0,), (3,), (0, 1), (3, 4), (2,), (5,), (1,), (4,), (1, 4)]
    s_ket, s_bra = ops[6].matrix, ops[7].matrix
    np.testing.assert_allclose(s_bra, np.conj(s_ket))
    assert abs(s_ket[1, 1] - 1j) < 1e-12 and abs(s_bra[1, 1] + 1j) < 1e-12
    rx_ket, rx_bra = ops[4].spec, ops[5].spec
    np.testing.assert_allclose(rx_bra.c2, np.conj(rx_ket.c2))
    with pytest.raises(NotImplementedError):
        c.apply_general_kraus([np.eye(8)], [0, 1, 2])           # channels on more than two qubits
    with pytest.raises(ValueError):
        c.apply_general_kraus(tc.channels.depolarizingchannel(0.1, 0.1, 0.1), [0, 1])   # one-qubit operators, two indices
    # the same circuit on the dense oracle: the doubled state vector is vec(rho)
    rho = odm.run(2, [("u", OG.H, [0]), ("u", OG.CNOT, [0, 1])])
    np.testing.assert_allclose(np.trace(rho), 1.0)


def test_sample2all_formats():
    import torch

    bits = torch.tensor([[1, 0, 1], [0, 0, 1], [1, 0, 1]])
    q = tc.quantum
    assert q.sample2all(bits, 3, format="sample_int").tolist() == [5, 1, 5]
    assert q.sample2all(torch.tensor([5, 1, 5]), 3, format="sample_bin").tolist() == bits.tolist()
    assert q.sample2all(bits, 3, format="count_dict_bin") == {"001": 1, "101": 2}
    assert q.sample2all(bits, 3, format="count_dict_int") == {1: 1, 5: 2}
    assert q.sample2all(bits, 3, format="count_vector").tolist() == [0, 1, 0, 0, 0, 2, 0, 0]
    vals, counts = q.sample2all(bits, 3, format="count_tuple")
    assert vals.tolist() == [1, 5] and counts.tolist() == [1, 2]
    with pytest.raises(ValueError):
        q.sample2all(bits, 3, format="nope")


def test_pauli_sum_bookkeeping_and_jit_dispatch():
    h = tc.quantum.PauliStringSum2COO([[1, 0, 3], [0, 2, 0]], [0.5, -2.0])
    assert tc.backend.is_sparse(h) and len(h) == 2 and h.n == 3 and h.weights == [0.5, -2.0]
    with pytest.raises(ValueError):
        tc.quantum.PauliStringSum2COO([[1, 0]], [1.0, 2.0])
    from tcmi.jit import TracedVag

    f = lambda p: p
    assert isinstance(tc.backend.jit(tc.backend.value_and_grad(f)), TracedVag)
    assert isinstance(tc.backend.jit(tc.backend.vvag(f)), TracedVag)
    assert tc.backend.jit(f, static_argnums=(0,)) is f
    j = tc.backend.jit(f)
    x = np.arange(3.0)
    assert j(x) is x and j.stats["fast"] == 0       # not an energy: the plain function runs


def test_three_qubit_synthesis_of_controlled_gates():
    from tcmi import synth

    for u in (OG.TOFFOLI, OG.FREDKIN, np.diag(np.exp(1j * np.arange(8)))):
        ops = synth.lower(synth.decompose_dense(u, [2, 0, 1]))
        assert all(len(q) <= 2 for _, q in ops)
        np.testing.assert_allclose(synth.expand(ops, [2, 0, 1]), u, atol=1e-12)


def _lightcone_circuit(tc, pbc):
    n = 4
    ns = n if pbc else n - 1
    c = tc.Circuit(n)
    for j in range(2):
        for i in range(n):
            c.rx(i, theta=0.2, name="rx" + str(j) + "-" + str(i))
        for i in range(ns):
            c.cnot(i, (i + 1) % n, name="cnot" + str(j) + "-" + str(i))
    return c


def test_lightcone_cancellation_node_counts_and_value():
    """Reference KAT tests/test_circuit.py:1507-1533: the <Z_0> network of the 2-layer rx + cnot chain has 37 nodes
    (open chain) of which 12 lie outside the causal cone -> 25; with the periodic cnot nothing cancels (41 -> 41).
    The simplified network contracts (oracle/tn.py, numpy) to the dense-oracle expectation value."""
    import tcmi as tc
    from oracle import dense, gates as OG, tn as OT

    tc.set_backend("hip"); tc.set_dtype("complex128")
    for pbc, want in ((True, (41, 41)), (False, (37, 25))):
        c = _lightcone_circuit(tc, pbc)
        nodes = c.expectation_before([tc.gates.z(), 0], reuse=False)
        l1 = len(nodes)
        nodes = tc.simplify._full_light_cone_cancel(nodes)
        assert (l1, len(nodes)) == want
        res = OT.contract([OT.Node(np.asarray(nd.tensor.cpu()), nd.edges) for nd in nodes])
        n = 4
        ops = []
        for j in range(2):
            ops += [(OG.rx(0.2), [i]) for i in range(n)]
            ops += [(OG.CNOT, [i, (i + 1) % n]) for i in range(n if pbc else n - 1)]
        psi = dense.run(n, ops)
        ref = dense.pauli_string_expectation(psi, n, [3, 0, 0, 0])
        np.testing.assert_allclose(complex(np.asarray(res.tensor)), ref, atol=1e-12)
    # an untagged node list is returned unchanged; a non-unitary constant keeps its pair in the network
    from tcmi import tn
    plain = [tn.Node(None, [1, 2]), tn.Node(None, [1, 2])]
    assert tc.simplify._full_light_cone_cancel(plain) is plain
    c = tc.Circuit(2)
    c.any(1, unitary=np.array([[1.0, 0.0], [0.0, 0.5]]))
    nodes = c.expectation_before([tc.gates.z(), 0], reuse=False)
    assert len(tc.simplify._full_light_cone_cancel(nodes)) == len(nodes)


def _diag_circuit(tc):
    """h layer, cmz on 6 qubits, a 4-control multicontrol, a 5-qubit diagonal, rx layer; + the oracle's op list."""
    from oracle import gates as OG

    n = 8
    rng = np.random.default_rng(5)
    c, ops = tc.Circuit(n), []
    for i in range(n):
        c.h(i); ops.append((OG.H, [i]))
    c.cmz(0, 2, 3, 5, 6, 7)
    d = np.ones(64, dtype=complex); d[-1] = -1
    ops.append((np.diag(d), [0, 2, 3, 5, 6, 7]))
    u = OG.rx(0.7)
    c.multicontrol(1, 4, 6, 0, 3, ctrl=[1, 0, 1, 1], unitary=u)
    m = np.eye(32, dtype=complex)
    cv = 0b1011
    m[2 * cv:2 * cv + 2, 2 * cv:2 * cv + 2] = u
    ops.append((m, [1, 4, 6, 0, 3]))
    dv = np.exp(1j * rng.uniform(0, 6, 32))
    c.diagonal(7, 1, 2, 4, 5, diag=dv); ops.append((np.diag(dv), [7, 1, 2, 4, 5]))
    for i in range(n):
        th = float(rng.uniform(0, 6)); c.rx(i, theta=th); ops.append((OG.rx(th), [i]))
    return c, ops, n


def test_big_diagonal_gates_enter_networks_as_chains():
    """Reference basecircuit.py:295-369 wires mpo= / diagonal= gates as MPO nodes and CopyNode hyperedges; here a
    diagonal on >= 4 qubits is a chain of site nodes with dimension-2 bond bundles (circuit._diag_chain): no node of
    the network is the gate's 4^k-entry matrix, and the network contracts (oracle/tn.py) to the dense-oracle values.
    Under the light-cone cancellation the gate is one placeholder until the cancellation is done."""
    import tcmi as tc
    from tcmi import circuit as C
    from oracle import dense, tn as OT

    tc.set_backend("hip"); tc.set_dtype("complex128")
    c, ops, n = _diag_circuit(tc)
    psi = dense.run(n, ops)
    # amplitude network
    bits = "10110010"
    nodes = c.amplitude_before(bits)
    assert max(nd.tensor.numel() for nd in nodes) <= 2 ** 8, max(nd.tensor.numel() for nd in nodes)
    assert sum(1 for nd in nodes if "-" in nd.name and not nd.name.startswith("qb")) == 6 + 5 + 5
    res = OT.contract([OT.Node(np.asarray(nd.tensor.cpu()), nd.edges) for nd in nodes])
    np.testing.assert_allclose(complex(np.asarray(res.tensor)), psi[int(bits, 2)], atol=1e-12)
    # expectation network (ket + bra chains, conjugated cores on the bra side)
    nodes = c.expectation_before([tc.gates.z(), 2], [tc.gates.x(), 5], reuse=False)
    assert max(nd.tensor.numel() for nd in nodes) <= 2 ** 8
    res = OT.contract([OT.Node(np.asarray(nd.tensor.cpu()), nd.edges) for nd in nodes])
    ref = dense.pauli_string_expectation(psi, n, [0, 0, 3, 0, 0, 1, 0, 0])
    np.testing.assert_allclose(complex(np.asarray(res.tensor)), ref, atol=1e-12)
    # light cone: a trailing cmz on qubits the operator does not touch cancels as ONE gate, then the rest expands
    import torch
    c2, ops2, _ = _diag_circuit(tc)
    c2.cmz(0, 1, 3, 4)
    nodes = c2.expectation_before([tc.gates.z(), 6], reuse=False, _chains=False)
    l0 = len(nodes)
    nodes = tc.simplify._full_light_cone_cancel(nodes)
    assert l0 - len(nodes) >= 2 + 2 * 7         # the trailing cmz pair and the 7 rx pairs off qubit 6
    nodes = C._expand_chains(nodes, torch.complex128, tc.backend.device)
    assert not any(isinstance(nd.tensor, C._DiagPlaceholder) for nd in nodes)
    res = OT.contract([OT.Node(np.asarray(nd.tensor.cpu()), nd.edges) for nd in nodes])
    ref = dense.pauli_string_expectation(psi, n, [0, 0, 0, 0, 0, 0, 3, 0])
    np.testing.assert_allclose(complex(np.asarray(res.tensor)), ref, atol=1e-12)
    # the chain itself: cores reproduce the entries; a multi-controlled phase has bond dimension 2
    v = np.ones(2 ** 9, dtype=complex); v[-1] = -1
    cores = C._diag_chain(v, 9)
    assert max(a.shape[2] for a in cores) == 2
    r = cores[0][0]
    for a in cores[1:]:
        r = np.tensordot(r, a, axes=([-1], [0]))
    np.testing.assert_allclose(r.reshape(-1), v, atol=1e-13)
