"""The HIP pairwise-contraction engine (tcmi_permute_bits / tcmi_cgemm), closed networks built by
Circuit.amplitude_before / expectation_before, and DistributedContractor (single process = all
slices on one GPU) against the oracle and against the state-vector path."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, gates as G, workloads as W  # noqa: E402
from tcmi import _knobs as KN  # noqa: E402


@pytest.fixture(params=["complex64", "complex128"])
def tcd(request):
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(request.param)
    yield tc
    tc.set_dtype("complex64")


def test_permute_and_tensordot_kernels(tcd):
    import torch
    from tcmi import tn

    tc = tcd
    dt = getattr(torch, tc.dtypestr)
    g = torch.Generator(device="cuda").manual_seed(0)
    for rank, perm in [(3, (2, 0, 1)), (6, (5, 3, 0, 1, 4, 2)), (12, tuple(np.random.default_rng(1).permutation(12)))]:
        t = torch.randn([2] * rank, dtype=dt, device="cuda", generator=g)
        got = tn.permute(t, perm)
        assert torch.equal(got, t.permute(*perm).contiguous())
    tol = 1e-4 if tc.dtypestr == "complex64" else 1e-11
    for ra, rb, xa, xb in [(2, 2, [1], [0]), (4, 6, [3, 0], [1, 4]), (10, 9, [9, 2, 4, 0], [0, 8, 3, 5]),
                           (14, 14, list(range(7, 14)), list(range(7))), (8, 8, [], []), (5, 5, [0, 1, 2, 3, 4], [4, 3, 2, 1, 0]),
                           # small output, long contraction (split-K kernel): scalar, 2x2 and 8x4 outputs
                           (16, 16, list(range(16)), list(range(16))), (14, 14, list(range(1, 14)), list(range(13))),
                           (17, 16, list(range(3, 17)), list(range(2, 16)))]:
        a = torch.randn([2] * ra, dtype=dt, device="cuda", generator=g)
        b = torch.randn([2] * rb, dtype=dt, device="cuda", generator=g)
        got = tn.tensordot(a, b, xa, xb)
        want = torch.tensordot(a.to(torch.complex128), b.to(torch.complex128), dims=(xa, xb))
        scale = float(want.abs().max()) + 1e-30
        assert float((got.to(torch.complex128) - want).abs().max()) / scale < tol, (ra, rb, xa, xb)


def test_tensordot_gradients(tcd):
    """VJP of the engine (two GEMMs with conjugate-transposed operands) against torch.autograd on
    torch.tensordot."""
    import torch
    from tcmi import tn

    tc = tcd
    dt = getattr(torch, tc.dtypestr)
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn([2] * 7, dtype=dt, device="cuda", generator=g, requires_grad=True)
    b = torch.randn([2] * 6, dtype=dt, device="cuda", generator=g, requires_grad=True)
    w = torch.randn([2] * 7, dtype=dt, device="cuda", generator=g)
    xa, xb = [6, 1, 3], [0, 5, 2]
    (tn.tensordot(a, b, xa, xb) * w).sum().real.backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    (torch.tensordot(a, b, dims=(xa, xb)) * w).sum().real.backward()
    tol = 1e-4 if tc.dtypestr == "complex64" else 1e-11
    assert float((ga - a.grad).abs().max()) < tol * 10 and float((gb - b.grad).abs().max()) < tol * 10


@pytest.mark.parametrize("n,depth", [(6, 3), (12, 4)])
def test_closed_network_amplitude_and_expectation(tcd, n, depth):
    """amplitude_before / expectation_before contracted by the HIP engine == oracle, and == the
    state-vector path (reference basecircuit.py:562-624, 393-447)."""
    from tcmi import tn

    tc = tcd
    rng = np.random.default_rng(n)
    c = tc.Circuit(n)
    ops = []
    for d in range(depth):
        for i in range(d % 2, n - 1, 2):
            u = G.random_two_qubit_gate(int(rng.integers(1 << 30)))
            c.any(i, i + 1, unitary=u)
            ops.append((u, [i, i + 1]))
        for i in range(n):
            t = float(rng.uniform(0, 6))
            c.rx(i, theta=t)
            ops.append((G.rx(t), [i]))
    psi = dense.run(n, ops)
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    bits = "".join(str(int(b)) for b in rng.integers(0, 2, n))
    amp = tn.contract_nodes(c.amplitude_before(bits)).tensor
    np.testing.assert_allclose(tc.backend.numpy(amp), dense.amplitude(psi, n, bits), atol=tol)
    np.testing.assert_allclose(tc.backend.numpy(c.amplitude(bits)), dense.amplitude(psi, n, bits), atol=tol)
    for reuse in (True, False):
        nodes = c.expectation_before((tc.gates.z(), [n - 1]), (tc.gates.x(), [0]), reuse=reuse)
        e = tn.contract_nodes(nodes).tensor
        want = dense.expectation(psi, n, (G.Z, [n - 1]), (G.X, [0]))
        np.testing.assert_allclose(tc.backend.numpy(e), want, atol=tol)
    # sliced contraction == unsliced
    nodes = c.expectation_before((tc.gates.z(), [n - 1]), reuse=False)
    e = tn.contract_nodes(nodes, target_size=2**5).tensor
    np.testing.assert_allclose(tc.backend.numpy(e), dense.expectation(psi, n, (G.Z, [n - 1])), atol=tol * 4)


def test_reference_distributed_contractor_kat(tcd):
    """reference tests/test_miscs.py:275-304 verbatim: 4 qubits, rx / cnot ladder / ry, target_size 2**3;
    value == expectation_ps(z=[-1]) at 1e-6, grad["y"].shape == (4,)."""
    tc = tcd

    def nodes_fn(params):
        c = tc.Circuit(4)
        c.rx(range(4), theta=params["x"])
        c.cnot([0, 1, 2], [1, 2, 3])
        c.ry(range(4), theta=params["y"])
        return c.expectation_before([tc.gates.z(), [-1]], reuse=False)

    params = {"x": np.ones([4]), "y": 0.3 * np.ones([4])}
    dc = tc.experimental.DistributedContractor(
        nodes_fn, params,
        {"slicing_reconf_opts": {"target_size": 2**3}, "max_repeats": 8, "minimize": "write", "parallel": False},
    )
    value, grad = dc.value_and_grad(params)
    assert tuple(grad["y"].shape) == (4,)

    def baseline(params):
        c = tc.Circuit(4)
        c.rx(range(4), theta=params["x"])
        c.cnot([0, 1, 2], [1, 2, 3])
        c.ry(range(4), theta=params["y"])
        return c.expectation_ps(z=[-1])

    np.testing.assert_allclose(tc.backend.numpy(value), tc.backend.numpy(baseline(params)).real, atol=1e-6)


def test_distributed_contractor_single_process(tcd, tmp_path):
    """reference tests/test_miscs.py:275-304: target_size = 2**3 forces real slicing; the summed value
    equals expectation_ps(z=[-1]) (atol 1e-6 in the reference) and grads have the parameter shapes."""
    import torch

    tc = tcd
    n, nlayers = 6, 3
    rng = np.random.default_rng(0)
    params = {"x": tc.backend.convert_to_tensor(rng.normal(size=[nlayers, n]), dtype=tc.rdtypestr),
              "y": tc.backend.convert_to_tensor(rng.normal(size=[n]), dtype=tc.rdtypestr)}

    def build(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for j in range(nlayers):
            for i in range(n - 1):
                c.cnot(i, i + 1)
            for i in range(n):
                c.rx(i, theta=p["x"][j, i])
        for i in range(n):
            c.ry(i, theta=p["y"][i])
        return c

    def nodes_fn(p):
        return build(p).expectation_before([tc.gates.z(), [-1]], reuse=False)

    dc = tc.experimental.DistributedContractor(nodes_fn, params, {"slicing_reconf_opts": {"target_size": 2**4}})
    assert 2 <= dc.tree.nslices <= 64 and dc.tree.max_size() <= 2**4
    v = dc.value(params)
    want = build(params).expectation_ps(z=[-1])
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    np.testing.assert_allclose(tc.backend.numpy(v), tc.backend.numpy(want), atol=tol)
    v2, g = dc.value_and_grad(params)
    np.testing.assert_allclose(tc.backend.numpy(v2), tc.backend.numpy(tc.backend.real(want)), atol=tol)
    assert tuple(g["x"].shape) == (nlayers, n) and tuple(g["y"].shape) == (n,)
    # gradient == adjoint-sweep gradient of the state-vector path
    def f(p):
        return tc.backend.real(build(p).expectation_ps(z=[-1]))
    _, g_sv = tc.backend.value_and_grad(f)(params)
    gtol = 2e-4 if tc.dtypestr == "complex64" else 1e-8
    np.testing.assert_allclose(tc.backend.numpy(g["x"]), tc.backend.numpy(g_sv["x"]), atol=gtol)
    np.testing.assert_allclose(tc.backend.numpy(g["y"]), tc.backend.numpy(g_sv["y"]), atol=gtol)
    # path persistence (reference experimental.py:947-991)
    fp = str(tmp_path / "tree.pkl")
    data = tc.experimental.DistributedContractor.find_path(nodes_fn, params, {"slicing_reconf_opts": {"target_size": 2**4}}, fp)
    assert set(data) == {"inputs", "output", "size_dict", "path", "sliced_inds"}
    dc2 = tc.experimental.DistributedContractor.from_path(fp, nodes_fn, params=params)
    np.testing.assert_allclose(tc.backend.numpy(dc2.value(params)), tc.backend.numpy(v), atol=tol)


def test_config4_grid_rqc_amplitude_sliced():
    """SURVEY 8d config 4 at a size the oracle finishes in seconds: 4x5 grid, depth-8 brickwork of
    Haar-random two-qubit gates (reference gates.py:852-863), amplitude <0..0|C|0..0> through
    DistributedContractor with real slicing; equals the dense oracle and the unsliced contraction."""
    import tcmi as tc
    from oracle import dense, gates as OG

    rows, cols, depth = 4, 5, 8
    n = rows * cols
    q = lambda r, c: r * cols + c
    pairs = []
    for d in range(depth):
        pat = d % 4
        if pat in (0, 1):
            pairs += [(q(r, c), q(r, c + 1)) for r in range(rows) for c in range(pat, cols - 1, 2)]
        else:
            pairs += [(q(r, c), q(r + 1, c)) for r in range(pat - 2, rows - 1, 2) for c in range(cols)]
    mats = [OG.random_two_qubit_gate(900 + k) for k in range(len(pairs))]
    psi = dense.run(n, [(m, [a, b]) for m, (a, b) in zip(mats, pairs)])
    bits = "0" * n
    want = dense.amplitude(psi, n, [0] * n)

    def nodes_fn(_):
        c = tc.Circuit(n)
        for m, (a, b) in zip(mats, pairs):
            c.any(a, b, unitary=m.reshape(2, 2, 2, 2))
        return c.amplitude_before(bits)

    opts = {"slicing_opts": {"target_size": 2**9}, "max_repeats": 16}
    dc = tc.experimental.DistributedContractor(nodes_fn, None, opts)
    assert dc.tree.nslices >= 2 and dc.tree.max_size() <= 2**9
    v = complex(dc.value(None, op=lambda x: x))
    assert abs(v - want) < 2e-5 * max(abs(want), 1e-3), (v, want)
    dc1 = tc.experimental.DistributedContractor(nodes_fn, None, {"slicing_opts": {"target_size": 2**30}, "max_repeats": 4})
    assert dc1.tree.nslices == 1
    v1 = complex(dc1.value(None, op=lambda x: x))
    assert abs(v - v1) < 2e-5 * max(abs(want), 1e-3)


# ---- cut contraction (half-circuit batches + one MFMA GEMM) ---------------------------------------
@pytest.mark.parametrize("n,d", [(16, 3), (18, 4), (20, 5)])
def test_cut_contraction_matches_oracle_and_statevector(n, d):
    import tcmi as tc
    from tcmi.executor import CutCircuit

    tc.set_backend("hip"); tc.set_dtype("complex64")
    params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n])
    ref = dense.run(n, W.hea_b_ops(n, d, params) + [(G.CNOT, [n // 2 - 1, n // 2]), (G.random_two_qubit_gate(3), [n // 2, n // 2 - 2])])
    outs = {}
    try:
        for method in ("cut", "plain"):
            tc.set_contractor(method)
            c = tc.Circuit(n)
            W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype="float32"), zz=tc.gates._zz_matrix)
            c.cnot(n // 2 - 1, n // 2)
            c.any(n // 2, n // 2 - 2, unitary=G.random_two_qubit_gate(3))
            assert isinstance(c._compiled(), CutCircuit) == (method == "cut")
            outs[method] = tc.backend.numpy(c.wavefunction())
            np.testing.assert_allclose(outs[method], ref, atol=1e-5)
    finally:
        tc.set_contractor("greedy")
    assert np.abs(outs["cut"] - outs["plain"]).max() < 1e-5


@pytest.mark.parametrize("dt", ["complex64"])      # the cut order is a complex64 plan (executor._maybe_cut)
def test_cut_weights_kernel_matches_the_elementwise_formulation(dt, monkeypatch):
    """``tcmi_cut_weights`` (one launch: w[b][k] = prod_j coef_j(digit_j(k), theta_b) in float64) against the same
    product written with torch elementwise ops / gather / prod, for a batch of parameter vectors; the cut circuit has
    constant (cnot, dense gate) and parametrised (rzz) gates across the cut."""
    import torch
    import tcmi as tc
    from tcmi.executor import CutCircuit

    n, d = 16, 3
    tc.set_backend("hip"); tc.set_dtype(dt)
    try:
        tc.set_contractor("cut")
        rdt = "float32" if dt == "complex64" else "float64"
        params = np.random.default_rng(5).uniform(0, 2 * np.pi, [2 * d, n])
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(params, dtype=rdt), zz=tc.gates._zz_matrix)
        c.cnot(n // 2 - 1, n // 2)
        c.any(n // 2, n // 2 - 2, unitary=G.random_two_qubit_gate(3))
        cc = c._compiled()
        assert isinstance(cc, CutCircuit)
        g = torch.Generator().manual_seed(1)
        p = (torch.rand(5, max(1, cc.nparams), generator=g, dtype=torch.float64) * 6.0).to(cc.rdtype).cuda()
        w_kernel = cc._weights(p)
        # the same weights as elementwise torch operations on the tables the kernel's were packed from
        tabs = cc._wtabs
        a = p[:, tabs["pidx"]].to(torch.float64).reshape(5, tabs["nb"], tabs["rmax"]) * tabs["scale"] + tabs["offs"]
        v = tabs["const"] + tabs["cmask"] * torch.cos(a) + tabs["smask"] * torch.sin(a)      # [B, nb, rmax] complex128
        w_torch = torch.gather(v, 2, tabs["dig"].expand(5, tabs["nb"], -1)).prod(dim=1).to(cc.tdtype)
        assert w_kernel.shape == (5, cc.K) and w_kernel.dtype == w_torch.dtype
        np.testing.assert_allclose(w_kernel.cpu().numpy(), w_torch.cpu().numpy(), atol=3e-7 if dt == "complex64" else 1e-14)
        assert float(w_kernel.abs().max()) > 1e-3
    finally:
        tc.set_contractor("greedy")
        tc.set_dtype("complex64")


def test_cut_contraction_two_streams_equal_one_stream(monkeypatch):
    """The right half-circuit batch runs on a second HIP stream and joins before the GEMM: bit-identical to the
    one-stream order (knob ``cut_streams=0``) for several batches in a row, with allocator churn between the calls."""
    import torch
    import tcmi as tc
    from tcmi.executor import CutCircuit

    n, d = 20, 5
    tc.set_backend("hip"); tc.set_dtype("complex64")
    try:
        tc.set_contractor("cut")

        def f(p):
            c = tc.Circuit(n)
            W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            return c.wavefunction()

        fv = tc.backend.vmap(f)
        g = torch.Generator().manual_seed(3)
        for B in (4, 1):
            for _ in range(6):
                p = (torch.rand(B, 2 * d, n, generator=g) * 6.28).cuda()
                monkeypatch.setitem(KN.VALUES, "cut_streams", "1")
                a = fv(p)
                junk = torch.randn(1 << 20, device="cuda")
                monkeypatch.setitem(KN.VALUES, "cut_streams", "0")
                b = fv(p)
                del junk
                assert torch.equal(a, b)
                assert abs(float((a.abs() ** 2).sum(-1).max()) - 1.0) < 1e-4
        c = tc.Circuit(n)
        W.hea_b(c, n, d, tc.backend.convert_to_tensor(np.zeros([2 * d, n]), dtype="float32"), zz=tc.gates._zz_matrix)
        assert isinstance(c._compiled(), CutCircuit)
    finally:
        tc.set_contractor("greedy")


def test_cut_contraction_full_size_batch_and_grad():
    """Config 2 at full size through the cut order (auto-selected by the cost model): parity with the
    oracle's TN contraction, vmap batching, and value_and_grad (forward = cut, backward = adjoint
    sweep) against the state-vector executor."""
    import torch
    import tcmi as tc
    from tcmi.executor import CutCircuit
    from oracle import tn as otn

    tc.set_backend("hip"); tc.set_dtype("complex64"); tc.set_contractor("greedy")
    n, d, params = W.config_params(2)
    c = tc.Circuit(n)
    W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
    # (the last of the d crossing ZZ gates is applied by the join kernel: d - 1 bonds, tcmi/cut.py; TCMI_CUT_DEFER = 0 / 2:
    # none / two of them)
    ndefer = int(os.environ.get("TCMI_CUT_DEFER", "1"))
    assert isinstance(c._compiled(), CutCircuit) and c._compiled().K == 2 ** (d - ndefer)
    assert (c._compiled().spec.epilogue is not None) == (ndefer > 0)
    psi = c.state()
    oc = otn.Circuit(n, dtype=np.complex128)
    W.hea_b(oc, n, d, params.astype(np.float64))
    assert np.abs(tc.backend.numpy(psi) - oc.wavefunction()).max() < 1e-5
    assert abs(float((psi.abs().double() ** 2).sum()) - 1) < 1e-5
    # batched (vmap) and gradient at a smaller size
    n, d, B = 16, 3, 3
    pbs = np.random.default_rng(5).uniform(0, 2 * np.pi, [B, 2 * d, n]).astype(np.float32)

    def state(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return c.state()

    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return tc.backend.real(c.expectation_ps(x=[0, 1]) + c.expectation_ps(z=[7, 8]))

    res = {}
    try:
        for method in ("cut", "plain"):
            tc.set_contractor(method)
            st = tc.backend.numpy(tc.backend.vmap(state)(tc.backend.convert_to_tensor(pbs)))
            v, g = tc.backend.vvag(energy)(tc.backend.convert_to_tensor(pbs))
            res[method] = (st, tc.backend.numpy(v), tc.backend.numpy(g))
    finally:
        tc.set_contractor("greedy")
    for b in range(B):
        np.testing.assert_allclose(res["cut"][0][b], dense.run(n, W.hea_b_ops(n, d, pbs[b])), atol=1e-5)
    np.testing.assert_allclose(res["cut"][1], res["plain"][1], atol=1e-5)
    np.testing.assert_allclose(res["cut"][2], res["plain"][2], atol=2e-4)


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_tall_gemm_is_issued_in_row_chunks(dt):
    """More than 65535 row tiles (skinny products of a reconfigured contraction tree): row chunks in the
    launcher, same numbers."""
    import torch
    from tcmi import _lib

    tdt = torch.complex64 if dt == "complex64" else torch.complex128
    M, N, K = (1 << 22) + 192, 16, 4
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, dtype=tdt, device="cuda", generator=g)
    b = torch.randn(K, N, dtype=tdt, device="cuda", generator=g)
    c = torch.empty(M, N, dtype=tdt, device="cuda")
    code = _lib.TCMI_C64 if dt == "complex64" else _lib.TCMI_C128
    _lib.check(_lib.lib().tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 1, 0, 0, 0, 0, code,
                                     torch.cuda.current_stream().cuda_stream), "tcmi_cgemm")
    ref = a @ b
    tol = 1e-5 if dt == "complex64" else 1e-12
    assert float((c - ref).abs().max()) < tol * float(ref.abs().max())


@pytest.mark.parametrize("trans_a", [0, 1])
@pytest.mark.parametrize("shape", [(32, 32, 1 << 16), (64, 128, (1 << 14) + 40), (96, 32, 5000)])
def test_few_tile_long_k_gemm_splits_k(shape, trans_a):
    """Few output tiles and a long K (closing steps of a contraction tree, e.g. 32 x 32 over K = 2^20): the MFMA
    kernel splits K over the grid and sums the partial products with atomics; same numbers as one pass, with a
    batch and a ragged last chunk."""
    import torch
    from tcmi import _lib

    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(2)
    batch = 2
    a = torch.randn(batch, M, K, dtype=torch.complex64, device="cuda", generator=g) / np.sqrt(K)
    b = torch.randn(batch, K, N, dtype=torch.complex64, device="cuda", generator=g)
    a_st = a.transpose(1, 2).contiguous() if trans_a else a
    c = torch.full((batch, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    _lib.check(_lib.lib().tcmi_cgemm(a_st.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, batch, M * K, K * N, M * N,
                                     trans_a, _lib.TCMI_C64, torch.cuda.current_stream().cuda_stream), "tcmi_cgemm")
    ref = (a.to(torch.complex128) @ b.to(torch.complex128))
    assert float((c - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_tensordot_from_stored_layouts_matches_torch():
    """tcmi_tensordot_bits (MFMA kernel, bit-deposit addressing, no operand permuted) against torch.tensordot:
    random contracted axes in random pairing order, ranks 0..20, outer products, outputs narrower than a tile, few-tile
    outputs with a long K (split-K) and k bits below / above the row bits of either operand."""
    import ctypes
    import torch
    from tcmi import _lib

    rng = np.random.default_rng(7)
    g = torch.Generator(device="cuda").manual_seed(3)
    cases = [(12, 10, 4), (18, 9, 7), (9, 18, 5), (6, 6, 6), (14, 14, 12), (20, 20, 17), (8, 3, 0), (0, 5, 0),
             (5, 5, 5), (16, 7, 1), (19, 15, 13), (7, 17, 2), (13, 13, 0)]
    for ra, rb, nk in cases:
        for rep in range(3):
            xa = [int(x) for x in rng.permutation(ra)[:nk]]
            xb = [int(x) for x in rng.permutation(rb)[:nk]]
            if rep == 1 and nk:          # k bits lowest in a, highest in b
                xa, xb = list(range(ra - nk, ra)), list(range(nk))
            if rep == 2 and nk:          # k bits highest in a, lowest in b, reversed pairing
                xa, xb = list(range(nk)), list(range(rb - 1, rb - 1 - nk, -1))
            a = torch.randn([2] * ra, dtype=torch.complex64, device="cuda", generator=g)
            b = torch.randn([2] * rb, dtype=torch.complex64, device="cuda", generator=g)
            c = torch.full([2] * (ra + rb - 2 * nk), float("nan"), dtype=torch.complex64, device="cuda")
            arr_a, arr_b = (ctypes.c_int * max(nk, 1))(*xa), (ctypes.c_int * max(nk, 1))(*xb)
            _lib.check(_lib.lib().tcmi_tensordot_bits(
                a.data_ptr(), ra, b.data_ptr(), rb, ctypes.cast(arr_a, ctypes.c_void_p), ctypes.cast(arr_b, ctypes.c_void_p),
                nk, c.data_ptr(), _lib.TCMI_C64, torch.cuda.current_stream().cuda_stream), "tcmi_tensordot_bits")
            ref = np.tensordot(a.cpu().numpy().astype(np.complex128), b.cpu().numpy().astype(np.complex128), axes=(xa, xb))
            err = float(np.abs(c.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
            assert err < 3e-5, (ra, rb, nk, xa, xb, err)
    # argument errors
    one = torch.zeros(2, 2, dtype=torch.complex64, device="cuda")
    bad = (ctypes.c_int * 2)(0, 0)
    rc = _lib.lib().tcmi_tensordot_bits(one.data_ptr(), 2, one.data_ptr(), 2, ctypes.cast(bad, ctypes.c_void_p),
                                        ctypes.cast(bad, ctypes.c_void_p), 2, one.data_ptr(), _lib.TCMI_C64, 0)
    assert rc != 0


def test_fused_reverse_mode_steps_match_torch_autograd(monkeypatch):
    """tcmi_tensordot_bits_ex (conjugated operand + result stored in the operand's axis order, one launch): the two
    VJPs of a tensordot against torch's autograd of torch.tensordot and against the three-launch form; a shape the
    small-tensor kernel does not take falls back by itself; the entry point refuses such shapes."""
    import ctypes
    import torch
    from tcmi import _lib, tn

    rng = np.random.default_rng(11)
    g_ = torch.Generator(device="cuda").manual_seed(5)
    for ra, rb, nk in [(4, 4, 2), (6, 2, 1), (2, 6, 2), (8, 7, 3), (12, 4, 4), (5, 5, 5), (3, 3, 0), (10, 10, 8), (14, 4, 2)]:
        xa = [int(x) for x in rng.permutation(ra)[:nk]]
        xb = [int(x) for x in rng.permutation(rb)[:nk]]
        a = torch.randn([2] * ra, dtype=torch.complex64, device="cuda", generator=g_)
        b = torch.randn([2] * rb, dtype=torch.complex64, device="cuda", generator=g_)
        cot = torch.randn([2] * (ra + rb - 2 * nk), dtype=torch.complex64, device="cuda", generator=g_)
        a64, b64 = a.to(torch.complex128).requires_grad_(True), b.to(torch.complex128).requires_grad_(True)
        ra_, rb_ = torch.autograd.grad(torch.tensordot(a64, b64, dims=(xa, xb)), (a64, b64), cot.to(torch.complex128))
        monkeypatch.setitem(KN.VALUES, "tn_fused_vjp", "1")
        ga, gb = tn.tensordot_vjp(a, b, xa, xb, cot)
        monkeypatch.setitem(KN.VALUES, "tn_fused_vjp", "0")
        ga0, gb0 = tn.tensordot_vjp(a, b, xa, xb, cot)
        for got, plain, ref in ((ga, ga0, ra_), (gb, gb0, rb_)):
            scale = max(float(ref.abs().max()), 1e-30)
            assert tuple(got.shape) == tuple(ref.shape)
            assert float((got.to(torch.complex128) - ref).abs().max()) / scale < 3e-5, (ra, rb, nk)
            assert float((got - plain).abs().max()) / scale < 3e-5
    monkeypatch.setitem(KN.VALUES, "tn_fused_vjp", "1")
    # the tile kernels store through the permutation too (no conjugation there): plain tiles, split-K with atomics,
    # the register-accumulator kernel for <= 8 x 8 results, rows / columns narrower than a tile
    for ra, rb, nk in [(14, 6, 3), (16, 16, 12), (13, 13, 11), (15, 5, 2), (10, 15, 4), (17, 14, 13)]:
        xa = [int(x) for x in rng.permutation(ra)[:nk]]
        xb = [int(x) for x in rng.permutation(rb)[:nk]]
        rc = ra + rb - 2 * nk
        perm = [int(x) for x in rng.permutation(rc)]
        a = torch.randn([2] * ra, dtype=torch.complex64, device="cuda", generator=g_)
        b = torch.randn([2] * rb, dtype=torch.complex64, device="cuda", generator=g_)
        monkeypatch.setattr(tn, "FUSED_PERM_MAX_RANK", 31)
        got = tn._tensordot_fused(a, b, xa, xb, perm, 0)
        assert got is not None
        ref = np.tensordot(a.cpu().numpy().astype(np.complex128), b.cpu().numpy().astype(np.complex128),
                           axes=(xa, xb)).transpose(perm)
        assert float(np.abs(got.cpu().numpy() - ref).max()) / float(np.abs(ref).max()) < 3e-5, (ra, rb, nk)
    assert _lib.lib().tcmi_tensordot_bits_small_ok(12, 4, 4) == 1 and _lib.lib().tcmi_tensordot_bits_small_ok(14, 4, 2) == 0
    big = torch.zeros([2] * 14, dtype=torch.complex64, device="cuda")
    ax = (ctypes.c_int * 2)(0, 1)
    rc = _lib.lib().tcmi_tensordot_bits_ex(big.data_ptr(), 14, big.data_ptr(), 14, ctypes.cast(ax, ctypes.c_void_p),
                                           ctypes.cast(ax, ctypes.c_void_p), 2, None, 1, big.data_ptr(), _lib.TCMI_C64, 0)
    assert rc != 0


def test_many_gate_sized_tensordots_in_one_launch():
    """tcmi_tensordot_small_desc + tcmi_tensordot_small_batch: 40 independent jobs of different shapes (conjugated
    operands, permuted results, scalars, outer products, 4096-element results) in ONE launch against numpy; the
    descriptor builder refuses shapes the small kernel does not take."""
    import ctypes
    import torch
    from tcmi import _lib, tn

    L = _lib.lib()
    rng = np.random.default_rng(21)
    g_ = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(4, 4, 2), (6, 2, 1), (2, 6, 2), (8, 7, 3), (12, 4, 4), (5, 5, 5), (3, 3, 0), (10, 10, 8), (0, 4, 0), (7, 7, 1)]
    jobs, host = [], np.zeros((40, tn.SMALL_DESC_WORDS), dtype=np.int32)
    for j in range(40):
        ra, rb, nk = shapes[j % len(shapes)]
        xa = [int(x) for x in rng.permutation(ra)[:nk]]
        xb = [int(x) for x in rng.permutation(rb)[:nk]]
        rc = ra + rb - 2 * nk
        perm = [int(x) for x in rng.permutation(rc)] if j % 3 else None
        flags = j % 4
        a = torch.randn([2] * ra, dtype=torch.complex64, device="cuda", generator=g_)
        b = torch.randn([2] * rb, dtype=torch.complex64, device="cuda", generator=g_)
        c = torch.full([2] * rc, float("nan"), dtype=torch.complex64, device="cuda")
        arr_a, arr_b = (ctypes.c_int * max(nk, 1))(*xa), (ctypes.c_int * max(nk, 1))(*xb)
        arr_p = (ctypes.c_int * max(rc, 1))(*perm) if perm is not None and rc else None
        _lib.check(L.tcmi_tensordot_small_desc(
            a.data_ptr(), ra, b.data_ptr(), rb, ctypes.cast(arr_a, ctypes.c_void_p), ctypes.cast(arr_b, ctypes.c_void_p), nk,
            ctypes.cast(arr_p, ctypes.c_void_p) if arr_p is not None else None, flags, c.data_ptr(), host[j].ctypes.data),
            "tcmi_tensordot_small_desc")
        jobs.append((a, b, c, xa, xb, perm, flags))
    dev = torch.from_numpy(host.reshape(-1)).cuda()
    _lib.check(L.tcmi_tensordot_small_batch(dev.data_ptr(), 40, 12, torch.cuda.current_stream().cuda_stream),
               "tcmi_tensordot_small_batch")
    torch.cuda.synchronize()
    for a, b, c, xa, xb, perm, flags in jobs:
        an, bn = a.cpu().numpy().astype(np.complex128), b.cpu().numpy().astype(np.complex128)
        ref = np.tensordot(an.conj() if flags & 1 else an, bn.conj() if flags & 2 else bn, axes=(xa, xb))
        if perm is not None and ref.ndim:
            ref = ref.transpose(perm)
        err = float(np.abs(c.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        assert err < 3e-5, (a.dim(), b.dim(), xa, xb, perm, flags, err)
    big = torch.zeros([2] * 14, dtype=torch.complex64, device="cuda")
    ax = (ctypes.c_int * 2)(0, 1)
    assert L.tcmi_tensordot_small_desc(big.data_ptr(), 14, big.data_ptr(), 14, ctypes.cast(ax, ctypes.c_void_p),
                                       ctypes.cast(ax, ctypes.c_void_p), 2, None, 0, big.data_ptr(), host[0].ctypes.data) != 0
    assert L.tcmi_tensordot_small_batch(dev.data_ptr(), 70000, 12, 0) != 0


@pytest.mark.parametrize("dt", ["complex64", "complex128"])
def test_scattered_contraction_matches_tensordot(dt):
    """tcmi_contract_scattered (big tensor x small tensor over arbitrary axes, no permute of the big one) against
    torch.tensordot: 1..5 contracted axes anywhere, either operand order, no free axes on the small side."""
    import torch
    from tcmi import tn

    tdt = torch.complex64 if dt == "complex64" else torch.complex128
    rng = np.random.default_rng(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    rank = 22     # 2^14 free indices of the big operand are left after 8 contracted axes (tn.SCATTERED_MIN_FREE)
    big = torch.randn([2] * rank, dtype=tdt, device="cuda", generator=g)
    cases = 0
    for nk in (1, 2, 3, 4, 5, 6, 7, 8):
        for nfree in ((0, 1, 3, 5) if nk <= 5 else (0, 2, 4)):
            for big_first in (True, False):
                ax_big = sorted(rng.choice(rank, nk, replace=False).tolist(), key=lambda _: rng.random())
                small = torch.randn([2] * (nk + nfree), dtype=tdt, device="cuda", generator=g)
                ax_small = rng.permutation(nk + nfree)[:nk].tolist()
                if big_first:
                    got = tn.tensordot(big, small, ax_big, ax_small)
                    ref = torch.tensordot(big, small, (ax_big, ax_small))
                else:
                    got = tn.tensordot(small, big, ax_small, ax_big)
                    ref = torch.tensordot(small, big, (ax_small, ax_big))
                assert got.shape == ref.shape
                tol = 2e-5 if dt == "complex64" else 1e-12
                assert float((got - ref).abs().max()) < tol * max(1.0, float(ref.abs().max())), (nk, nfree, big_first)
                cases += 1
    assert cases == 58
    # the fast path was taken (no autograd, big operand of rank >= 16)
    assert tn._tensordot_scattered(big, small, ax_big, ax_small,
                                   [i for i in range(rank) if i not in ax_big],
                                   [i for i in range(small.dim()) if i not in ax_small]) is not None


def test_state_free_measurement_matches_state_route_and_scales_past_the_state(tcd):
    """measure / perfect_sampling by closed-network contractions (reference basecircuit.py:461-558) == the
    state-vector route on the same random numbers; and a 36-qubit GHZ circuit (no state vector fits the tile-VM's
    32-qubit limit) samples 00..0 / 11..1 with probability 1/2."""
    import torch

    tc = tcd
    n = 8
    rng = np.random.default_rng(5)
    c = tc.Circuit(n)
    ops_ref = []
    for d in range(3):
        for i in range(d % 2, n - 1, 2):
            u = G.random_two_qubit_gate(int(rng.integers(1 << 30)))
            c.any(i, i + 1, unitary=u)
            ops_ref.append((np.asarray(u).reshape(4, 4), [i, i + 1]))
        for i in range(n):
            th = float(rng.uniform(0, 6))
            c.rx(i, theta=th)
            ops_ref.append((G.rx(th), [i]))
    tol = 1e-5 if tc.dtypestr == "complex64" else 1e-10
    from oracle import sampling as OS

    psi_ref = dense.run(n, ops_ref)
    for seed in range(4):
        status = np.random.default_rng(seed).uniform(size=n)
        s1, p1 = c.measure(*range(n), with_prob=True, status=status, state_free=False)
        s2, p2 = c.measure(*range(n), with_prob=True, status=status, state_free=True)
        assert torch.equal(s1.cpu(), s2.cpu())
        np.testing.assert_allclose(float(p1), float(p2), rtol=20 * tol, atol=tol)
        # the oracle: conditional marginals of the dense state with the reference's comparison rule
        so, po = OS.measure(psi_ref, n, list(range(n)), status)
        assert np.array_equal(s1.cpu().numpy().astype(int), so)
        np.testing.assert_allclose(float(p1), po, rtol=20 * tol, atol=tol)
    so, po = OS.measure(psi_ref, n, [5, 2], np.array([0.3, 0.9]))
    s5, p5 = c.measure(5, 2, with_prob=True, status=np.array([0.3, 0.9]), state_free=True)
    assert np.array_equal(s5.cpu().numpy().astype(int), so)
    np.testing.assert_allclose(float(p5), po, rtol=20 * tol, atol=tol)
    s3, _ = c.measure(5, 2, status=np.array([0.3, 0.9]), state_free=True)
    s4, _ = c.measure(5, 2, status=np.array([0.3, 0.9]), state_free=False)
    assert torch.equal(s3.cpu(), s4.cpu())

    big = 36
    g = tc.Circuit(big)
    g.h(0)
    for i in range(big - 1):
        g.cnot(i, i + 1)
    seen = set()
    for seed in range(6):
        bits, p = g.perfect_sampling(status=np.random.default_rng(100 + seed).uniform(size=big))
        b = bits.cpu().numpy().astype(int)
        assert b.min() == b.max()
        np.testing.assert_allclose(float(p), 0.5, atol=1e-5)
        seen.add(int(b[0]))
    assert seen == {0, 1}


def test_graph_replay_follows_new_leaf_values_and_random_bitstrings():
    """The sliced contraction is replayed from HIP graphs that read static copies of the leaf tensors: a second and a
    third call with OTHER gate parameters and amplitudes of random bit strings must follow the new values (dense
    oracle), and repeating the first call reproduces the first value."""
    import torch
    import tcmi as tc
    from oracle import dense, gates as OG

    tc.set_backend("hip"); tc.set_dtype("complex64")
    rows, cols, depth = 4, 5, 8
    n = rows * cols
    q = lambda r, c: r * cols + c
    pairs = []
    for d in range(depth):
        pat = d % 4
        if pat in (0, 1):
            pairs += [(q(r, c), q(r, c + 1)) for r in range(rows) for c in range(pat, cols - 1, 2)]
        else:
            pairs += [(q(r, c), q(r + 1, c)) for r in range(pat - 2, rows - 1, 2) for c in range(cols)]
    mats = [OG.random_two_qubit_gate(300 + k) for k in range(len(pairs))]
    rng = np.random.default_rng(11)
    bits = "".join(str(int(b)) for b in rng.integers(0, 2, n))

    def nodes_fn(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.ry(i, theta=p[i])
        for m, (a, b) in zip(mats, pairs):
            c.any(a, b, unitary=m.reshape(2, 2, 2, 2))
        for i in range(n):
            c.rz(i, theta=p[n + i])
        return c.amplitude_before(bits)

    def ref(p):
        ops = [(OG.ry(float(p[i])), [i]) for i in range(n)] + [(m, [a, b]) for m, (a, b) in zip(mats, pairs)]
        ops += [(OG.rz(float(p[n + i])), [i]) for i in range(n)]
        return dense.amplitude(dense.run(n, ops), n, bits)

    ps = [rng.uniform(0, 2 * np.pi, 2 * n) for _ in range(3)]
    p0 = tc.backend.convert_to_tensor(ps[0], dtype="float32")
    dc = tc.experimental.DistributedContractor(nodes_fn, p0, {"slicing_opts": {"target_size": 2**9}, "max_repeats": 8})
    assert dc.tree.nslices >= 2 and len(dc.tree._symbolic_steps()[0]) >= 32      # the graph path is taken
    vals = []
    for p in ps + [ps[0]]:
        v = complex(dc.value(tc.backend.convert_to_tensor(p, dtype="float32"), op=lambda x: x))
        want = ref(p)
        assert abs(v - want) < 3e-5 * max(abs(want), 1e-3), (v, want)
        vals.append(v)
    assert getattr(dc.tree, "_graph_cache", None) is not None
    assert abs(vals[0] - vals[3]) < 1e-7


@pytest.mark.parametrize("dt,tol", [("complex64", 1e-5), ("complex128", 1e-10)])
def test_sliced_value_and_grad_on_the_fast_kernels_matches_the_adjoint_path(dt, tol, monkeypatch):
    """Really sliced network (rzz / rx ladder of examples/slicing_auto_pmap_vqa.py, 18 qubits, >= 8 slices):
    ``DistributedContractor.value_and_grad`` through the hand-written reverse sweep (``tn.contract_slices_vjp``: every
    forward and backward step on the untaped kernels) against (a) ``backend.value_and_grad`` on the state-vector
    adjoint path and (b) torch's tape over the same tree (TCMI_TN_VJP=0: ``TensordotFn``, same kernels step by step).
    Reference: experimental.py:1182-1211."""
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(dt)
    try:
        nq, dq = 18, 4
        pv = np.random.default_rng(11).uniform(0.2, 1.2, [nq, dq, 2]).astype(np.float32 if dt == "complex64" else np.float64)
        pt = tc.backend.convert_to_tensor(pv)

        def circuit(params):
            c = tc.Circuit(nq)
            for i in range(dq):
                for j in range(nq - 1):
                    c.rzz(j, j + 1, theta=params[j, i, 0])
                for j in range(nq):
                    c.rx(j, theta=params[j, i, 1])
            return c

        dc = tc.experimental.DistributedContractor(
            lambda p: circuit(p).expectation_before([tc.gates.z(), [0]], reuse=False), pt,
            {"slicing_opts": {"target_slices": 8}, "max_repeats": 16, "minimize": "combo"})
        assert dc.tree.nslices >= 8
        v, g = dc.value_and_grad(pt)
        v0, g0 = tc.backend.value_and_grad(lambda p: tc.backend.real(circuit(p).expectation_ps(z=[0])))(pt)
        # replays follow the parameters: a second point through the same graphs (static descriptor tables of the
        # batched levels, static leaf copies), then the first one again
        pt2 = pt * 0.7 + 0.1
        vb, gb = dc.value_and_grad(pt2)
        v0b, g0b = tc.backend.value_and_grad(lambda p: tc.backend.real(circuit(p).expectation_ps(z=[0])))(pt2)
        assert abs(float(vb) - float(v0b)) < tol and float((gb - g0b).abs().max()) < tol
        assert float((gb - g).abs().max()) > 1e-3
        vc, gc = dc.value_and_grad(pt)
        assert abs(float(vc) - float(v)) < 1e-6 and float((gc - g).abs().max()) < 1e-6
        cache = dc.tree._vjp_graph_cache
        if dt == "complex64":
            assert cache["batch"] is not None and cache["batch"].n > 100 and cache["batch"].launches < cache["batch"].n / 4
        monkeypatch.setenv("TCMI_TN_VJP", "0")
        v1, g1 = dc.value_and_grad(pt)
        e_adj = float((g - g0).abs().max())
        e_tape = float((g - g1).abs().max())
        print(f"sliced value_and_grad {dt}: {dc.tree.nslices} slices, |v - adjoint| {abs(float(v) - float(v0)):.2e}, "
              f"max |g - adjoint| {e_adj:.2e}, max |g - tape| {e_tape:.2e}")
        assert abs(float(v) - float(v0)) < tol and abs(float(v) - float(v1)) < tol
        assert e_adj < tol and e_tape < tol
        assert float(g0.abs().max()) > 1e-3
    finally:
        tc.set_dtype("complex64")


def test_node_function_is_traced_validated_and_given_up_when_it_must_be(tcd):
    """DistributedContractor._arrays: recipe recorded on the first call, compared bit for bit with the function on the
    second, replayed from the third on; angles that are not elements of the parameter tensor, or a function whose
    second call does something else, keep the function."""
    tc = tcd
    if tc.dtypestr != "complex64":
        pytest.skip("one dtype is enough")
    nq, dq = 10, 2
    rng = np.random.default_rng(2)
    pts = [tc.backend.convert_to_tensor(rng.uniform(0.2, 1.2, [nq, dq, 2]).astype(np.float32)) for _ in range(4)]
    calls = {"n": 0}

    def circuit(params, scale=1.0):
        c = tc.Circuit(nq)
        for i in range(dq):
            for j in range(nq - 1):
                c.rzz(j, j + 1, theta=params[j, i, 0])
            for j in range(nq):
                c.rx(j, theta=params[j, i, 1] if scale == 1.0 else params[j, i, 1] * scale)
        return c

    def ref(p, scale=1.0):
        return tc.backend.value_and_grad(lambda q: tc.backend.real(circuit(q, scale).expectation_ps(z=[0])))(p)

    def plain(p):
        calls["n"] += 1
        return circuit(p).expectation_before([tc.gates.z(), [0]], reuse=False)

    dc = tc.experimental.DistributedContractor(plain, pts[0], {"slicing_opts": {"target_slices": 2}, "max_repeats": 4})
    for k, p in enumerate(pts):
        v, g = dc.value_and_grad(p)
        v0, g0 = ref(p)
        assert abs(float(v) - float(v0)) < 2e-5 and float((g - g0).abs().max()) < 2e-4, k
        assert dc._trace_state["mode"] == ("record" if k == 0 else "replay")
    assert calls["n"] == 4        # the constructor, the recording call, the validating call and its perturbed re-run

    def computed(p):              # angles are results of arithmetic, not elements of the parameter tensor
        return circuit(p, 0.5).expectation_before([tc.gates.z(), [0]], reuse=False)

    dc2 = tc.experimental.DistributedContractor(computed, pts[0], {"slicing_opts": {"target_slices": 2}, "max_repeats": 4})
    for p in pts[:3]:
        v, g = dc2.value_and_grad(p)
        v0, g0 = ref(p, 0.5)
        assert abs(float(v) - float(v0)) < 2e-5 and float((g - g0).abs().max()) < 2e-4
    assert dc2._trace_state["mode"] == "off"

    flip = {"n": 0}

    def moody(p):                 # the same network, but the second call reads the parameters differently
        flip["n"] += 1
        return circuit(p if flip["n"] != 3 else p.flip(0).contiguous()).expectation_before([tc.gates.z(), [0]], reuse=False)

    dc3 = tc.experimental.DistributedContractor(moody, pts[0], {"slicing_opts": {"target_slices": 2}, "max_repeats": 4})
    dc3.value_and_grad(pts[0])
    v, g = dc3.value_and_grad(pts[1])            # call 3 of the function: the recipe of call 2 does not reproduce it
    v0, g0 = ref(pts[1].flip(0).contiguous())
    assert dc3._trace_state["mode"] == "off"
    assert abs(float(v) - float(v0)) < 2e-5 and float((g.flip(0) - g0).abs().max()) < 2e-4


def test_arguments_that_shape_the_networks_constants_are_never_frozen_by_the_trace(tcd):
    """`value(p0), value(p0), value(p1)` with a different amplitude bitstring in p1.  The one-hot
    caps of `amplitude_before(params["bits"])` are constants of the network built from an ARGUMENT; two identical warm-up
    calls must not freeze them (the reference jits nodes_fn, every argument stays live:
    examples/distributed_interface_amplitude.py pattern)."""
    tc = tcd
    if tc.dtypestr != "complex64":
        pytest.skip("one dtype is enough")
    import torch

    nq = 8
    rng = np.random.default_rng(4)
    th = tc.backend.convert_to_tensor(rng.uniform(0.2, 1.2, [nq, 2]).astype(np.float32))

    def circ(angles):
        c = tc.Circuit(nq)
        for j in range(nq):
            c.rx(j, theta=angles[j, 0])
        for j in range(nq - 1):
            c.cnot(j, j + 1)
        for j in range(nq):
            c.ry(j, theta=angles[j, 1])
        return c

    def nodes_fn(params):
        return circ(params["angles"]).amplitude_before(params["bits"])

    b0 = torch.tensor([0, 1, 0, 0, 1, 1, 0, 1], device="cuda")
    b1 = torch.tensor([1, 1, 0, 1, 0, 0, 0, 1], device="cuda")
    dc = tc.experimental.DistributedContractor(nodes_fn, {"angles": th, "bits": b0}, {"max_repeats": 4})
    psi = tc.backend.numpy(circ(th).wavefunction())

    def amp(bits):
        return psi[int("".join(str(int(x)) for x in bits.tolist()), 2)]

    for bits in (b0, b0, b1, b0, b1):
        v = dc.value({"angles": th, "bits": bits}, op=lambda x: x)
        assert abs(complex(tc.backend.numpy(v).reshape(-1)[0]) - amp(bits)) < 1e-5, bits
    assert dc._trace_state["mode"] == "off"


@pytest.mark.parametrize("shape", [(64, 64, 16), (64, 64, 32), (128, 192, 48), (512, 320, 256), (4096, 4096, 256),
                                   (64, 4096, 1024), (128, 128, 16), (256, 384, 80), (1024, 128, 512)])
def test_dma_pipelined_join_gemm_matches_complex128_and_the_plain_kernel(shape, monkeypatch):
    """k-major A with whole tiles and K a multiple of 16 (the cut-contraction join, cut.py) runs on the LDS-DMA
    pipelined kernels: ``cgemm_dma128_kernel`` (128 x 128 tile, 64 x 64 per wave, two stages) when M and N are
    multiples of 128, else ``cgemm_dma_kernel`` (64 x 64 tile, ring of 3 stages); XCD-aware tile walk.  Same numbers as a
    complex128 product (f32 accumulation bound) and, per element, close to the register-staged kernel
    (TCMI_GEMM_DMA is read once per process, so that comparison uses a shape the DMA kernel does not take)."""
    import torch
    from tcmi import _lib

    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    batch = 3 if M * N <= 1 << 20 else 2
    a = torch.randn(batch, K, M, dtype=torch.complex64, device="cuda", generator=g) / np.sqrt(K)   # [K][M]: k-major
    b = torch.randn(batch, K, N, dtype=torch.complex64, device="cuda", generator=g)
    c = torch.full((batch, M, N), float("nan"), dtype=torch.complex64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):   # repeated launches: a ring-buffer race would show up as run-to-run differences
        c.fill_(float("nan"))
        _lib.check(_lib.lib().tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, batch, M * K, K * N, M * N,
                                         1, _lib.TCMI_C64, st), "tcmi_cgemm")
        ref = a.transpose(1, 2).to(torch.complex128) @ b.to(torch.complex128)
        err = float((c - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-5, (shape, err)
    # ragged neighbour shape -> the register-staged kernel; its leading 64-aligned block must agree with the DMA result
    if M >= 128:
        M2 = M - 8
        a2 = a[:, :, :M2].contiguous()
        c2 = torch.empty(batch, M2, N, dtype=torch.complex64, device="cuda")
        _lib.check(_lib.lib().tcmi_cgemm(a2.data_ptr(), b.data_ptr(), c2.data_ptr(), M2, N, K, batch, M2 * K, K * N,
                                         M2 * N, 1, _lib.TCMI_C64, st), "tcmi_cgemm")
        assert float((c2 - c[:, :M2]).abs().max()) < 1e-5 * float(ref.abs().max())


def test_output_wavefunction_slicing_matches_the_full_state(tcd):
    """``experimental.sliced_state`` / ``sliced_expectation_ps`` (reference examples/slicing_wavefunction_vqa.py): the
    state projected on every mask of three cut qubits equals the corresponding amplitudes of the full state; the Pauli
    string expectation summed over the 2^3 mask pairs equals ``expectation_ps`` (X, Y and Z on cut and kept qubits);
    its gradient through the sliced route equals the adjoint sweep's."""
    import itertools
    tc = tcd
    from tcmi.experimental import sliced_expectation_ps, sliced_state

    n, d = 12, 3
    rng = np.random.default_rng(4)
    rdt = np.float32 if tc.dtypestr == "complex64" else np.float64
    pv = rng.uniform(0, 2 * np.pi, [2 * d, n]).astype(rdt)
    tol = 2e-5 if tc.dtypestr == "complex64" else 1e-10

    def circ(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        return c

    pt = tc.backend.convert_to_tensor(pv)
    full = tc.backend.numpy(circ(pt).state()).reshape([2] * n)
    cut = [1, 5, 9]
    for mask in itertools.product((0, 1), repeat=3):
        got = tc.backend.numpy(sliced_state(circ(pt), cut, list(mask)))
        idx = [slice(None)] * n
        for q, b in zip(cut, mask):
            idx[q] = b
        np.testing.assert_allclose(got, full[tuple(idx)].reshape(-1), atol=tol)
    for ps in ([0, 1, 0, 0, 3, 2, 0, 0, 0, 3, 1, 0], [3, 3] + [0] * 10, [0, 2, 2, 0, 0, 1, 0, 1, 0, 0, 0, 3]):
        want = float(tc.backend.numpy(tc.backend.real(circ(pt).expectation_ps(
            x=[i for i, p in enumerate(ps) if p == 1], y=[i for i, p in enumerate(ps) if p == 2],
            z=[i for i, p in enumerate(ps) if p == 3]))))
        got = float(sliced_expectation_ps(lambda: circ(pt), ps, cut))
        assert abs(got - want) < tol, (ps, got, want)
    ps = [0, 1, 0, 0, 3, 2, 0, 0, 0, 3, 1, 0]
    v1, g1 = tc.backend.value_and_grad(lambda p: sliced_expectation_ps(lambda: circ(p), ps, cut))(pt)
    v2, g2 = tc.backend.value_and_grad(lambda p: tc.backend.real(circ(p).expectation_ps(x=[1, 10], y=[5], z=[4, 9])))(pt)
    assert abs(float(v1) - float(v2)) < tol
    assert float((g1 - g2).abs().max()) < (2e-4 if tc.dtypestr == "complex64" else 1e-8)
