"""The reference's plug-in contracts on the hip backend: the contractor callable, the custom path
finder, debug levels, the two-qubit gate split rule, and the backend ops the contractor calls."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, workloads as W  # noqa: E402


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(request.param)
    yield tc
    tc.set_dtype("complex64")
    tc.set_contractor("greedy")


def _np(tc, x):
    return tc.backend.numpy(x)


def _merge_circuit(tc, n=6, depth=4, ops=None):
    """reference tests/test_circuit.py:2260-2272 (``_build_merge_circuit``)."""
    c = tc.Circuit(n)
    for d in range(depth):
        for i in range(n):
            c.h(i); c.rz(i, theta=0.3 * (i + 1)); c.rx(i, theta=0.2 * (d + 1))
            if ops is not None:
                ops += [(G.H, [i]), (G.rz(0.3 * (i + 1)), [i]), (G.rx(0.2 * (d + 1)), [i])]
        for i in range(n - 1):
            c.cnot(i, i + 1)
            if ops is not None:
                ops.append((G.CNOT, [i, i + 1]))
    return c


def test_contractor_equivalence_kat(tcd):
    """reference tests/test_circuit.py:2274-2292: every contractor gives the same state; here additionally the node-list
    contractor call on the circuit's own network reproduces it."""
    tc = tcd
    ops = []
    expected = _np(tc, _merge_circuit(tc, ops=ops).state())
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    np.testing.assert_allclose(expected, dense.run(6, ops), atol=tol)
    for name, kw in (("greedy", {"preprocessing": True}), ("plain-experimental", {"local_steps": 3}), ("branch", {}),
                     ("optimal", {}), ("auto", {})):
        with tc.runtime_contractor(name, **kw):
            np.testing.assert_allclose(_np(tc, _merge_circuit(tc).state()), expected, rtol=1e-5, atol=1e-5)
    # the contractor callable on the uncontracted amplitude network (basecircuit.py:562-624)
    c = _merge_circuit(tc)
    bits = "010110"
    for name in ("greedy", "branch"):
        cf = tc.set_contractor(name)
        amp = cf(c.amplitude_before(bits)).tensor
        np.testing.assert_allclose(complex(_np(tc, amp)), expected[int(bits, 2)], atol=tol)
    assert tc.contractor is cf


def test_custom_path_finder_plugin(tcd):
    """set_contractor("custom", optimizer=f): f(input_sets, output_set, size_dict, memory_limit) -> linear path
    (reference cons.py:800, 1037-1040); a precomputed list path; custom_stateful with a class."""
    tc = tcd
    c = _merge_circuit(tc, n=5, depth=2)
    want = complex(_np(tc, c.state())[0b10110])
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    calls = []

    def naive(input_sets, output_set, size_dict, memory_limit=None):
        # always contract the two first tensors of the current list: a valid, bad path
        assert all(isinstance(s, list) and all(isinstance(x, str) and len(x) == 1 for x in s) for s in input_sets)
        assert set(size_dict.values()) == {2} and output_set == []
        calls.append((len(input_sets), memory_limit))
        return [(0, 1)] * (len(input_sets) - 1)

    cf = tc.set_contractor("custom", optimizer=naive, memory_limit=123)
    amp = cf(c.amplitude_before("10110")).tensor
    np.testing.assert_allclose(complex(_np(tc, amp)), want, atol=tol)
    assert calls and calls[0][1] == 123
    nn = calls[0][0]
    cf = tc.set_contractor("custom", optimizer=[(0, 1)] * (nn - 1))          # precomputed list path
    np.testing.assert_allclose(complex(_np(tc, cf(c.amplitude_before("10110")).tensor)), want, atol=tol)

    class Finder:
        def __init__(self, tag="x"):
            self.tag = tag

        def __call__(self, input_sets, output_set, size_dict, memory_limit=None):
            return [(len(input_sets) - 2 - k, len(input_sets) - 1 - k) for k in range(len(input_sets) - 1)]

    cf = tc.set_contractor("custom_stateful", optimizer=Finder, opt_conf={"tag": "y"}, max_time=10, minimize="size")
    np.testing.assert_allclose(complex(_np(tc, cf(c.amplitude_before("10110")).tensor)), want, atol=tol)
    with pytest.raises(ValueError, match="needs an `optimizer`"):
        tc.set_contractor("custom")(c.amplitude_before("10110"))


def test_contractor_call_errors_and_edge_order(tcd):
    """reference cons.py:877-896: more than one dangling edge needs output_edge_order; a wrong order is rejected;
    the result axes follow output_edge_order."""
    tc = tcd
    import torch
    from tcmi import tn

    cdt = torch.complex64 if tc.dtypestr == "complex64" else torch.complex128
    rng = np.random.default_rng(0)
    a = torch.tensor(rng.normal(size=(2, 2, 2)) + 1j * rng.normal(size=(2, 2, 2)), dtype=cdt, device="cuda")
    b = torch.tensor(rng.normal(size=(2, 2, 2)) + 1j * rng.normal(size=(2, 2, 2)), dtype=cdt, device="cuda")
    e = [tn.new_edge() for _ in range(5)]
    mk = lambda: [tn.Node(a, [e[0], e[1], e[2]]), tn.Node(b, [e[2], e[3], e[4]])]  # noqa: E731
    cf = tc.set_contractor("greedy")
    with pytest.raises(ValueError, match="more than one remaining edge"):
        cf(mk())
    with pytest.raises(ValueError, match="not equal to the remaining"):
        cf(mk(), output_edge_order=[e[0], e[1]])
    out = cf(mk(), output_edge_order=[e[4], e[0], e[3], e[1]])
    ref = np.einsum("abk,kcd->dacb", a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_allclose(_np(tc, out.tensor), ref, atol=1e-5)
    free = cf(mk(), ignore_edge_order=True)
    assert sorted(free.edges) == sorted([e[0], e[1], e[3], e[4]])


def test_debug_level_returns_zeros(tcd):
    """reference tests/test_circuit.py:922-946: example_block n=10 d=4 with debug_level=2 -> zeros of shape 2^10."""
    tc = tcd
    n, d = 10, 4

    @tc.set_function_contractor("greedy", debug_level=2, contraction_info=True)
    def small_tn():
        param = tc.backend.ones([2 * d, n])
        c = tc.Circuit(n)
        W.hea_b(c, n, d, param, zz=tc.gates._zz_matrix)
        return c.state()

    out = _np(tc, small_tn())
    assert out.shape == (2**n,)
    np.testing.assert_allclose(out, np.zeros([2**n]), atol=1e-5)
    c = tc.Circuit(4)
    c.h(0)
    z = tc.set_contractor("greedy", debug_level=2)(c.amplitude_before("0000"))
    assert complex(_np(tc, z.tensor)) == 0
    tc.set_contractor("greedy")
    assert abs(complex(_np(tc, tc.contractor(c.amplitude_before("0000")).tensor)) - 2 ** -0.5) < 1e-6


def test_split_rule_truncates_like_the_reference(tcd):
    """Circuit(split=...) (reference basecircuit.py:231-275, simplify.py:88-128): a two-qubit gate is replaced by its
    truncated operator-Schmidt form; no truncation -> unchanged."""
    tc = tcd
    n = 8
    u = np.asarray(G.random_two_qubit_gate(11)).reshape(4, 4)

    def run(split):
        c = tc.Circuit(n, split=split)
        for i in range(n):
            c.h(i)
        c.any(2, 3, unitary=u)
        c.cnot(3, 4)
        c.any(5, 6, unitary=u, split={"max_singular_values": 4})   # per-gate rule that cannot truncate
        return _np(tc, c.state())

    def oracle(mat):
        ops = [(G.H, [i]) for i in range(n)] + [(mat, [2, 3]), (G.CNOT, [3, 4]), (u, [5, 6])]
        return dense.run(n, ops)

    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    np.testing.assert_allclose(run(None), oracle(u), atol=tol)
    np.testing.assert_allclose(run({"max_singular_values": 4}), oracle(u), atol=tol)
    # rank-2 truncation on the (out0, in0) | (out1, in1) split
    t = u.reshape(2, 2, 2, 2).transpose(0, 2, 1, 3).reshape(4, 4)
    uu, sv, vh = np.linalg.svd(t)
    m2 = ((uu[:, :2] * sv[:2]) @ vh[:2]).reshape(2, 2, 2, 2).transpose(0, 2, 1, 3).reshape(4, 4)
    got = run({"max_singular_values": 2})
    cn = np.eye(4)[[0, 1, 3, 2]]
    t2 = cn.reshape(2, 2, 2, 2).transpose(0, 2, 1, 3).reshape(4, 4)
    u2, s2, v2 = np.linalg.svd(t2)   # CNOT has operator-Schmidt rank 2: unchanged by the rank-2 rule
    ops = [(G.H, [i]) for i in range(n)] + [(m2, [2, 3]), (G.CNOT, [3, 4]), (u, [5, 6])]
    np.testing.assert_allclose(got, dense.run(n, ops), atol=tol)
    # parametrised two-qubit gates with concrete angles are truncated like constants (rank-1 rule: the largest
    # operator-Schmidt component of rzz and of iswap survives); tensor-valued angles are refused
    c = tc.Circuit(4, split={"max_singular_values": 1})
    for i in range(4):
        c.h(i)
        c.rz(i, theta=0.4 * (i + 1))
    c.rzz(0, 1, theta=0.3)
    c.iswap(1, 2, theta=0.7)

    def rank1(m):
        t = np.asarray(m, dtype=np.complex128).reshape(2, 2, 2, 2).transpose(0, 2, 1, 3).reshape(4, 4)
        uu, sv, vh = np.linalg.svd(t)
        assert sv[0] - sv[1] > 1e-2
        return ((uu[:, :1] * sv[:1]) @ vh[:1]).reshape(2, 2, 2, 2).transpose(0, 2, 1, 3).reshape(4, 4)

    zz = np.diag(np.exp(-0.5j * 0.3 * np.array([1, -1, -1, 1])))
    isw = np.asarray(tc.gates.iswap_gate(0.7).tensor, dtype=np.complex128).reshape(4, 4)
    ops = []
    for i in range(4):
        ops += [(G.H, [i]), (G.rz(0.4 * (i + 1)), [i])]
    ops += [(rank1(zz), [0, 1]), (rank1(isw), [1, 2])]
    np.testing.assert_allclose(_np(tc, c.state()), dense.run(4, ops), atol=tol)
    # tensor-valued angles: a family whose operator-Schmidt rank fits the cap at every angle is exact (rzz has rank 2,
    # the reference's own max_singular_values = 2 configurations); a truncation that depends on the angle is refused
    th = tc.backend.convert_to_tensor(np.asarray(0.3, dtype=tc.rdtypestr))
    c = tc.Circuit(4, split={"max_singular_values": 2, "fixed_choice": 1})
    for i in range(4):
        c.h(i)
        c.rz(i, theta=0.4 * (i + 1))
    c.rzz(0, 1, theta=th)
    c.cphase(2, 3, theta=th)
    ops = []
    for i in range(4):
        ops += [(G.H, [i]), (G.rz(0.4 * (i + 1)), [i])]
    ops += [(zz, [0, 1]), (np.diag([1, 1, 1, np.exp(0.3j)]), [2, 3])]
    np.testing.assert_allclose(_np(tc, c.state()), dense.run(4, ops), atol=tol)
    for conf, gate in (({"max_singular_values": 1}, "rzz"), ({"max_singular_values": 2}, "iswap"),
                       ({"max_truncation_err": 0.1}, "rzz")):
        with pytest.raises(NotImplementedError, match="Backend 'hip' has not implemented"):
            c = tc.Circuit(4, split=conf)
            getattr(c, gate)(0, 1, theta=th)


def test_backend_ops_run_on_the_hip_kernels(tcd):
    """backend.tensordot / transpose / einsum / matmul on complex GPU tensors (template: reference
    backends/cupy_backend.py:85-99) against numpy."""
    tc = tcd
    import torch

    K = tc.backend
    cdt = torch.complex64 if tc.dtypestr == "complex64" else torch.complex128
    rng = np.random.default_rng(1)
    rnd = lambda *s: rng.normal(size=s) + 1j * rng.normal(size=s)  # noqa: E731
    tol = 1e-4 if tc.dtypestr == "complex64" else 1e-10
    a, b = rnd(*[2] * 12), rnd(*[2] * 5)
    ta, tb = torch.tensor(a, dtype=cdt, device="cuda"), torch.tensor(b, dtype=cdt, device="cuda")
    np.testing.assert_allclose(_np(tc, K.tensordot(ta, tb, [[3, 7], [1, 4]])), np.tensordot(a, b, [[3, 7], [1, 4]]), atol=tol)
    np.testing.assert_allclose(_np(tc, K.tensordot(ta, tb, 2)), np.tensordot(a, b, 2), atol=tol)
    perm = list(rng.permutation(12))
    np.testing.assert_allclose(_np(tc, K.transpose(ta, perm)), a.transpose(perm), atol=tol)
    m1, m2 = rnd(6, 10), rnd(10, 7)
    t1, t2 = torch.tensor(m1, dtype=cdt, device="cuda"), torch.tensor(m2, dtype=cdt, device="cuda")
    np.testing.assert_allclose(_np(tc, K.matmul(t1, t2)), m1 @ m2, atol=tol)
    g1, g2 = rnd(3, 4, 5), rnd(5, 4, 6)     # general (non-qubit) dimensions
    np.testing.assert_allclose(_np(tc, K.tensordot(torch.tensor(g1, dtype=cdt, device="cuda"),
                                                  torch.tensor(g2, dtype=cdt, device="cuda"), [[1, 2], [1, 0]])),
                               np.tensordot(g1, g2, [[1, 2], [1, 0]]), atol=tol)
    np.testing.assert_allclose(_np(tc, K.einsum("abcde,xbyd->exayc", tb, torch.tensor(a[0, 0, 0, 0, 0, 0, 0, 0], dtype=cdt, device="cuda"))),
                               np.einsum("abcde,xbyd->exayc", b, a[0, 0, 0, 0, 0, 0, 0, 0]), atol=tol)
    # real tensors / odd expressions stay on torch
    r = torch.tensor(rng.normal(size=(4, 4)), device="cuda")
    np.testing.assert_allclose(_np(tc, K.einsum("ii->", r)), np.trace(r.cpu().numpy()), atol=1e-6)


def test_expectation_between_two_states_kat(tcd):
    """reference tests/test_circuit.py:404-445 (module-level ``tc.expectation(*ops, ket=, bra=, normalization=)``):
    1j, the normalised forms, 1 and 1/sqrt(2)."""
    tc = tcd
    zp = np.array([1.0, 0.0])
    zd = np.array([0.0, 1.0])
    assert abs(complex(_np(tc, tc.expectation((tc.gates.y(), [0]), ket=zp, bra=zd))) - 1j) < 1e-7

    c = tc.Circuit(3)
    c.H(0)
    c.ry(1, theta=tc.num_to_tensor(0.8))
    c.cnot(1, 2)
    state = c.wavefunction()
    x1z2 = [(tc.gates.x(), [0]), (tc.gates.z(), [1])]
    e1 = c.expectation(*x1z2)
    e2 = tc.expectation(*x1z2, ket=state, bra=state, normalization=True)
    np.testing.assert_allclose(_np(tc, e2), _np(tc, e1), atol=1e-6)

    c = tc.Circuit(3)
    c.H(0)
    c.ry(1, theta=tc.num_to_tensor(0.8 + 0.7j))       # complex angle: a non-unitary gate, unnormalised state
    c.cnot(1, 2)
    state = c.wavefunction()
    nrm2 = float(np.linalg.norm(_np(tc, state)) ** 2)
    assert abs(nrm2 - 1) > 0.1
    # the same state from the dense oracle with the complex-angle matrix
    a = 0.5 * (0.8 + 0.7j)
    ry = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    ref = dense.run(3, [(G.H, [0]), (ry, [1]), (G.CNOT, [1, 2])])
    tol = 1e-6 if tc.dtypestr == "complex64" else 1e-12
    np.testing.assert_allclose(_np(tc, state), ref, atol=tol)
    e1 = _np(tc, c.expectation(*x1z2)) / nrm2
    e2 = tc.expectation(*x1z2, ket=state, normalization=True)
    np.testing.assert_allclose(_np(tc, e2), e1, atol=1e-6)
    want = np.vdot(ref, dense.run(3, [(G.X, [0]), (G.Z, [1])], inputs=ref)) / np.vdot(ref, ref)
    np.testing.assert_allclose(complex(_np(tc, e2)), want, atol=1e-6)

    c = tc.Circuit(2); c.X(1); s1 = c.state()
    c2 = tc.Circuit(2); c2.X(0); s2 = c2.state()
    c3 = tc.Circuit(2); c3.H(1); s3 = c3.state()
    x1x2 = [(tc.gates.x(), [0]), (tc.gates.x(), [1])]
    np.testing.assert_allclose(_np(tc, tc.expectation(*x1x2, ket=s1, bra=s2)), 1.0, atol=1e-6)
    np.testing.assert_allclose(_np(tc, tc.expectation(*x1x2, ket=s3, bra=s2)), 1.0 / np.sqrt(2), atol=1e-6)
    with pytest.raises(ValueError, match="Cannot measure two operators in one index"):
        tc.expectation((tc.gates.x(), [0]), (tc.gates.z(), [0]), ket=s1)
    # a larger state goes through tcmi_vdot; conj=False uses the bra as given
    n = 12
    rng = np.random.default_rng(2)
    k = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    b = rng.normal(size=2**n) + 1j * rng.normal(size=2**n)
    op = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))      # a non-unitary two-qubit operator
    ok = dense.run(n, [(op, [3, 7]), (G.Y, [0])], inputs=k)
    tol = 2e-3 if tc.dtypestr == "complex64" else 1e-8
    got = complex(_np(tc, tc.expectation((op, [3, 7]), (tc.gates.y(), [0]), ket=k, bra=b)))
    np.testing.assert_allclose(got, np.vdot(b, ok), rtol=tol)
    got2 = complex(_np(tc, tc.expectation((op, [3, 7]), (tc.gates.y(), [0]), ket=k, bra=b, conj=False)))
    np.testing.assert_allclose(got2, np.sum(b * ok), rtol=tol)


def test_hyperedge_and_strip_exponent_kats(tcd):
    """reference tests/test_hyperedge.py:73-138, 156-177 (CopyNode networks: 9.0, 17.0, [1, 4], reordered outer
    product) and tests/test_circuit.py:2178-2257 (strip_exponent: 0.1^400 -> (1.0, -400), ten 10.0 -> (1.0, 10),
    a 100-qubit expectation network of 2*identity gates -> 200 log10 2)."""
    tc = tcd
    from tcmi import tn

    v12 = lambda: tc.gates.num_to_tensor(np.array([1.0, 2.0]))  # noqa: E731
    cf = tc.set_contractor("greedy")
    # single hyperedge: sum_i A_i B_i C_i = 1 + 8
    e = [tn.new_edge() for _ in range(3)]
    nodes = [tn.Node(v12(), [e[0]]), tn.Node(v12(), [e[1]]), tn.Node(v12(), [e[2]]), tn.CopyNode(3, 2, edges=e)]
    np.testing.assert_allclose(_np(tc, cf(nodes).tensor), 9.0, atol=1e-5)
    # chained: A-CN1-B, CN1-CN2, C-CN2-D: sum_i A_i B_i C_i D_i = 17
    e = [tn.new_edge() for _ in range(5)]
    nodes = [tn.Node(v12(), [e[0]]), tn.Node(v12(), [e[1]]), tn.Node(v12(), [e[3]]), tn.Node(v12(), [e[4]]),
             tn.CopyNode(3, 2, edges=[e[0], e[1], e[2]]), tn.CopyNode(3, 2, edges=[e[2], e[3], e[4]])]
    np.testing.assert_allclose(_np(tc, cf(nodes).tensor), 17.0, atol=1e-5)
    # dangling hyperedge: C_i = A_i B_i
    e = [tn.new_edge() for _ in range(3)]
    res = cf([tn.Node(v12(), [e[0]]), tn.Node(v12(), [e[1]]), tn.CopyNode(3, 2, edges=e)])
    np.testing.assert_allclose(_np(tc, res.tensor), np.array([1.0, 4.0]), atol=1e-5)
    # output reordering through two rank-2 CopyNodes
    a1, a2, b1, b2 = (tn.new_edge() for _ in range(4))
    nodes = [tn.Node(v12(), [a1]), tn.Node(tc.gates.num_to_tensor(np.array([3.0, 4.0])), [b1]),
             tn.CopyNode(2, 2, edges=[a1, a2]), tn.CopyNode(2, 2, edges=[b1, b2])]
    res = cf(nodes, output_edge_order=[b2, a2])
    np.testing.assert_allclose(_np(tc, res.tensor), np.outer([3.0, 4.0], [1.0, 2.0]), atol=1e-5)
    assert tuple(res.tensor.shape) == (2, 2)
    # a custom path finder also sees hyper-indices (one symbol in three inputs)
    seen = []

    def finder(input_sets, output_set, size_dict, memory_limit=None):
        seen.append(input_sets)
        return [(0, 1)] * (len(input_sets) - 1)

    e = [tn.new_edge() for _ in range(3)]
    nodes = [tn.Node(v12(), [e[0]]), tn.Node(v12(), [e[1]]), tn.Node(v12(), [e[2]]), tn.CopyNode(3, 2, edges=e)]
    np.testing.assert_allclose(_np(tc, tc.set_contractor("custom", optimizer=finder)(nodes).tensor), 9.0, atol=1e-5)
    assert seen and seen[0][0] == seen[0][1] == seen[0][2]

    # strip_exponent
    cf = tc.set_contractor("cotengra", strip_exponent=True)
    node, ex = cf([tn.Node(tc.backend.convert_to_tensor(0.1, tc.rdtypestr), []) for _ in range(400)])
    np.testing.assert_allclose(abs(complex(_np(tc, node.tensor))), 1.0, atol=1e-5)
    np.testing.assert_allclose(ex, -400.0, atol=1e-3 if tc.dtypestr == "complex64" else 1e-8)
    cf = tc.set_contractor("custom_stateful", optimizer=lambda **kw: (lambda i, o, s, memory_limit=None: [(0, 1)] * (len(i) - 1)),
                           strip_exponent=True)
    node, ex = cf([tn.Node(tc.backend.convert_to_tensor(10.0, tc.rdtypestr), []) for _ in range(6)])
    np.testing.assert_allclose(abs(complex(_np(tc, node.tensor))), 1.0, atol=1e-5)
    np.testing.assert_allclose(ex, 6.0, atol=1e-4)
    n = 40
    c = tc.Circuit(n)
    for i in range(n):
        c.any(i, unitary=np.array([[2.0, 0], [0, 2.0]]))
    cf = tc.set_contractor("cotengra", strip_exponent=True)
    node, ex = cf(c.expectation_before([tc.gates.z(), [0]], reuse=False))
    np.testing.assert_allclose(abs(complex(_np(tc, node.tensor))), 1.0, atol=1e-5)
    np.testing.assert_allclose(ex, 2 * n * np.log10(2.0), atol=1e-3)


def test_lightcone_expectation_kat(tcd):
    """Reference KAT tests/test_circuit.py:1507-1533 on the device: ``enable_lightcone=True`` gives the same <Z_0>
    (1e-5), 37 -> 25 nodes on the open chain and 41 -> 41 with the periodic cnot; plus a deep-narrow circuit
    (n = 12, 3 brickwork layers, <X_5 Z_6>) against the dense oracle -- the cone touches 8 of the 12 qubits."""
    tc = tcd

    def construct_c(pbc=True):
        n = 4
        ns = n if pbc else n - 1
        c = tc.Circuit(n)
        for j in range(2):
            for i in range(n):
                c.rx(i, theta=0.2, name="rx" + str(j) + "-" + str(i))
            for i in range(ns):
                c.cnot(i, (i + 1) % n, name="cnot" + str(j) + "-" + str(i))
        return c

    for b in [True, False]:
        c = construct_c(b)
        m1 = c.expectation_ps(z=[0], enable_lightcone=True)
        m2 = c.expectation_ps(z=[0])
        np.testing.assert_allclose(tc.backend.numpy(m1), tc.backend.numpy(m2), atol=1e-5)
        nodes = c.expectation_before([tc.gates.z(), 0], reuse=False)
        l1 = len(nodes)
        nodes = tc.simplify._full_light_cone_cancel(nodes)
        assert (l1, len(nodes)) == ((37, 25) if b is False else (41, 41))

    n, rng = 12, np.random.default_rng(3)
    c = tc.Circuit(n)
    ops = []
    for layer in range(3):
        for i in range(n):
            th = float(rng.uniform(0, 2 * np.pi))
            c.ry(i, theta=th)
            ops.append((G.ry(th), [i]))
        for i in range(layer % 2, n - 1, 2):
            th = float(rng.uniform(0, 2 * np.pi))
            c.rzz(i, i + 1, theta=th)
            c.cnot(i, i + 1)
            ops += [(G.rzz(th), [i, i + 1]), (G.CNOT, [i, i + 1])]
    psi = dense.run(n, ops)
    ps = [0] * n
    ps[5], ps[6] = 1, 3
    ref = dense.pauli_string_expectation(psi, n, ps)
    nodes = c.expectation_before([tc.gates.x(), 5], [tc.gates.z(), 6], reuse=False)
    assert len(tc.simplify._full_light_cone_cancel(nodes)) < len(nodes) - 20
    got = c.expectation_ps(x=[5], z=[6], enable_lightcone=True)
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10
    np.testing.assert_allclose(complex(tc.backend.numpy(got)), ref, atol=tol)


def test_mpo_and_diagonal_gate_format_kats(tcd):
    """``apply_general_gate(mpo=True)`` / ``(diagonal=True)`` (reference basecircuit.py:295-369) through the HIP
    executor with the reference's own known answers: ``test_apply_mpo_gate`` / ``test_apply_multicontrol_gate``
    (tests/test_circuit.py:998-1040), ``test_circuit_diagonal_gate`` / ``_rzm_gate`` / ``_cmz_gate``
    (tests/test_hyperedge.py:532-640: the hyperedge form equals the dense diagonal matrix)."""
    tc = tcd
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    ex = lambda c, q: float(_np(tc, tc.backend.real(c.expectation([tc.gates.z(), [q]]))))  # noqa: E731
    gate = tc.gates.multicontrol_gate(tc.gates._x_matrix, ctrl=[1, 0])
    c = tc.Circuit(3)
    c.X(0)
    c.mpo(0, 1, 2, mpo=gate.copy())
    assert abs(ex(c, 2) + 1) < tol
    c = tc.Circuit(3)
    c.X(1)
    c.mpo(0, 1, 2, mpo=gate.copy())
    assert abs(ex(c, 2) - 1) < tol
    c = tc.Circuit(3)
    c.X(2)
    c.multicontrol(0, 2, 1, ctrl=[0, 1], unitary=tc.gates._x_matrix)
    assert abs(ex(c, 1) + 1) < tol
    c = tc.Circuit(3)
    c.X(0)
    c.multicontrol(0, 2, 1, ctrl=[0, 1], unitary=tc.gates._x_matrix)
    assert abs(ex(c, 1) - 1) < tol
    c = tc.Circuit(4)
    c.X(0)
    c.X(2)
    c.multicontrol(0, 1, 2, 3, ctrl=[1, 0], unitary=tc.gates.swap())
    assert abs(ex(c, 3) + 1) < tol

    # diagonal: hyperedge form == dense diagonal matrix (state and a two-operator expectation)
    n = 3
    diag = np.array([1, -1, 1j, -1j, 1, 1, -1, -1])
    c1, c2 = tc.Circuit(n), tc.Circuit(n)
    for i in range(n):
        c1.h(i)
        c2.h(i)
    c1.any(*range(n), unitary=np.diag(diag))
    c2.diagonal(*range(n), diag=tc.backend.convert_to_tensor(diag))
    for i in range(n):
        c1.rx(i, theta=0.2)
        c2.rx(i, theta=0.2)
    np.testing.assert_allclose(_np(tc, c1.state()), _np(tc, c2.state()), atol=tol)
    e1 = c1.expectation([tc.gates.z(), [0]], [tc.gates.y(), [1]])
    e2 = c2.expectation([tc.gates.z(), [0]], [tc.gates.y(), [1]])
    np.testing.assert_allclose(_np(tc, e1), _np(tc, e2), atol=tol)

    # rzm / cmz on a 14-qubit register (packed kernels, terms of up to six qubits across register and thread bits)
    # against the dense oracle
    n = 14
    c = tc.Circuit(n)
    ops = []
    for i in range(n):
        c.h(i)
        ops.append((G.H, [i]))
    theta = 1.2
    qs = [0, 5, 9, 13, 2]
    c.rzm(*qs, theta=theta)
    zs = np.array([1.0])
    for _ in qs:
        zs = np.kron(zs, [1.0, -1.0])
    ops.append((np.diag(np.exp(-0.5j * theta * zs)), qs))
    qs2 = [3, 1, 12, 7, 10, 6]
    c.cmz(*qs2)
    z6 = np.ones(64, dtype=np.complex128)
    z6[-1] = -1
    ops.append((np.diag(z6), qs2))
    u1 = G.rx(0.7) @ G.rz(0.3)
    c.multicontrol(4, 11, 8, 0, ctrl=[1, 1, 0], unitary=u1)
    mc = np.eye(16, dtype=np.complex128)
    mc[12:14, 12:14] = u1
    ops.append((mc, [4, 11, 8, 0]))
    for i in range(n):
        c.rx(i, theta=0.3 + 0.05 * i)
        ops.append((G.rx(0.3 + 0.05 * i), [i]))
    np.testing.assert_allclose(_np(tc, c.state()), dense.run(n, ops), atol=tol)


def test_gradient_through_a_multi_qubit_z_rotation(tcd):
    """d/d theta of an energy through ``rzm`` (a five-qubit parity phase whose qubits fall on register AND thread bits:
    CNOT register moves around the folded term, plan.emit_diag) and the surrounding rotations, against central
    differences of the dense oracle."""
    tc = tcd
    n = 14
    qs = [0, 5, 9, 13, 2]

    def energy(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
            c.rx(i, theta=p[i])
        c.rzm(*qs, theta=p[n])
        for i in range(n):
            c.ry(i, theta=p[n + 1 + i])
        return tc.backend.real(c.expectation((tc.gates.x(), [5])) + c.expectation((tc.gates.z(), [0]), (tc.gates.z(), [9])))

    def ref(p):
        ops = []
        for i in range(n):
            ops += [(G.H, [i]), (G.rx(p[i]), [i])]
        zs = np.array([1.0])
        for _ in qs:
            zs = np.kron(zs, [1.0, -1.0])
        ops.append((np.diag(np.exp(-0.5j * p[n] * zs)), qs))
        for i in range(n):
            ops.append((G.ry(p[n + 1 + i]), [i]))
        psi = dense.run(n, ops)
        return (dense.expectation(psi, n, (G.X, [5])) + dense.expectation(psi, n, (G.Z, [0]), (G.Z, [9]))).real

    p0 = np.random.default_rng(2).uniform(0.2, 1.4, 2 * n + 1)
    rdt = np.float32 if tc.dtypestr == "complex64" else np.float64
    v, g = tc.backend.value_and_grad(energy)(tc.backend.convert_to_tensor(p0.astype(rdt)))
    g = _np(tc, g).astype(np.float64)
    pb = p0.astype(rdt).astype(np.float64)
    assert abs(float(v) - ref(pb)) < (2e-5 if rdt == np.float32 else 1e-10)
    for i in (n, 3, n + 4, 2 * n):
        pp, pm = pb.copy(), pb.copy()
        pp[i] += 1e-6
        pm[i] -= 1e-6
        fd = (ref(pp) - ref(pm)) / 2e-6
        assert abs(g[i] - fd) < (2e-4 if rdt == np.float32 else 1e-7), (i, g[i], fd)
