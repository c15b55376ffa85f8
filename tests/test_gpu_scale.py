"""Parity at the sizes BASELINE.json quotes: mid-size circuits against the dense oracle with
central-difference gradients, and the full-size configurations through size-independent properties
(complex64 against complex128 of the same circuit, norms, canonical forms)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense, workloads as W  # noqa: E402
from tcmi import _knobs as KN  # noqa: E402


def _energy_fn(tc, n, d):
    def energy(p):
        c = tc.Circuit(n)
        W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    return energy


@pytest.mark.parametrize("n,d", [(16, 4), (18, 3)])
@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_tfim_value_and_gradient_vs_dense_oracle(n, d, dtype):
    """SURVEY 8d config 3 at n = 16, 18: energy and central-difference gradient components (eps = 1e-6, the
    reference's own finite-difference step, tests/test_mpscircuit.py:452-457) from the dense complex128 oracle."""
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype(dtype)
    try:
        params = np.random.default_rng(n).normal(0, 0.5, [2 * d, n])
        rdt = np.float32 if dtype == "complex64" else np.float64
        v, g = tc.backend.value_and_grad(_energy_fn(tc, n, d))(tc.backend.convert_to_tensor(params.astype(rdt)))
        ref = lambda p: W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, p)), n)  # noqa: E731
        e_ref = ref(params.astype(rdt).astype(np.float64))
        # tolerances = 10-20 x the errors measured on MI355X (printed below; complex64 energy ~2e-6, gradient ~1e-6)
        e_err = abs(float(v) - e_ref)
        assert e_err < (4e-5 if dtype == "complex64" else 1e-9), e_err
        g = tc.backend.numpy(g)
        g_err = 0.0
        eps = 1e-6
        comps = [(0, 0), (0, n - 2), (1, 0), (1, n - 1), (2 * d - 2, n // 2), (2 * d - 1, n // 2), (d, 3), (d + 1, n - 4)]
        base = params.astype(rdt).astype(np.float64)
        for (a, b) in comps:
            pp, pm = base.copy(), base.copy()
            pp[a, b] += eps
            pm[a, b] -= eps
            fd = (ref(pp) - ref(pm)) / (2 * eps)
            g_err = max(g_err, abs(g[a, b] - fd))
            assert abs(g[a, b] - fd) < (2e-5 if dtype == "complex64" else 1e-7), (a, b, g[a, b], fd)
        print(f"n={n} d={d} {dtype}: energy error {e_err:.2e}, max gradient error over {len(comps)} components {g_err:.2e}")
    finally:
        tc.set_dtype("complex64")


def _full_golden():
    f = os.path.join(os.path.dirname(__file__), "golden", "full_size_golden.npz")
    if not os.path.exists(f):
        pytest.fail("tests/golden/full_size_golden.npz is missing (python tests/golden/make_golden_full.py ...)")
    return np.load(f)


def test_config3_full_size_complex64_against_complex128():
    """SURVEY 8d config 3 at full size (n = 28, depth 12, one sample of the bench's parameter generator), complex64
    against a complex128 run of the same circuit (the c128 path runs on different kernels: first-generation
    double-precision tile-VM and adjoint sweep, f64 trigonometry).  BASELINE.json's tolerance -- expectation values
    within 1e-5 -- is applied to EVERY one of the 55 terms <X_i>, <Z_i Z_i+1>.  Energy and gradient are held to 6-20 x the
    errors MEASURED at this size (profiles/r05c_config3_parity.txt: energy 8.7e-6, gradient against complex128 4.1e-6,
    against the oracle's central differences 5.1e-7 complex64 / 4.4e-9 complex128): a regression of one order of magnitude
    fails.  The measured maxima are printed."""
    import torch
    import tcmi as tc

    n, d = 28, 12
    params = np.random.default_rng(28).normal(0, 0.1, [2 * d, n])
    tc.set_backend("hip")
    out = {}
    try:
        for dtype in ("complex64", "complex128"):
            tc.set_dtype(dtype)
            rdt = np.float32 if dtype == "complex64" else np.float64
            p = tc.backend.convert_to_tensor(params.astype(np.float32).astype(rdt))
            v, g = tc.backend.jit(tc.backend.value_and_grad(_energy_fn(tc, n, d)))(p)
            c = tc.Circuit(n)
            W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            psi = c.state()
            nrm = float((psi.abs().to(torch.float64) ** 2).sum())
            terms = [float(tc.backend.real(c.expectation((tc.gates.x(), [i])))) for i in range(n)]
            terms += [float(tc.backend.real(c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1])))) for i in range(n - 1)]
            out[dtype] = (float(v), tc.backend.numpy(g).astype(np.float64), nrm, np.asarray(terms))
            del psi, c, v, g
            torch.cuda.empty_cache()
    finally:
        tc.set_dtype("complex64")
    (e64, g64, n64, t64), (e128, g128, n128, t128) = out["complex64"], out["complex128"]
    dterm, de, dg = np.abs(t64 - t128).max(), abs(e64 - e128), np.abs(g64 - g128).max()
    print(f"config 3 full size, complex64 vs complex128: max |<P_t>| difference over 55 terms {dterm:.2e}, "
          f"|E| difference {de:.2e}, max gradient difference {dg:.2e}, norms {n64:.8f} / {n128:.12f}")
    assert abs(n128 - 1) < 1e-10 and abs(n64 - 1) < 1e-4, (n64, n128)
    assert dterm < 1e-5, dterm
    # ... and against the ORACLE: the 55 terms and the energy of this very circuit from oracle.dense (complex128, gate by
    # gate on the 4 GiB state; tests/golden/make_golden_full.py config3), committed as a fixture
    full = _full_golden()
    assert np.array_equal(full["config3_params"].astype(np.float32), params.astype(np.float32))
    tor = np.concatenate([full["config3_x"], full["config3_zz"]])
    print(f"config 3 full size vs oracle.dense fixture: max |<P_t>| error complex64 {np.abs(t64 - tor).max():.2e}, "
          f"complex128 {np.abs(t128 - tor).max():.2e}; energy error {abs(e64 - float(full['config3_energy'])):.2e} / "
          f"{abs(e128 - float(full['config3_energy'])):.2e}")
    assert np.abs(t128 - tor).max() < 1e-10 and abs(e128 - float(full["config3_energy"])) < 1e-9
    assert np.abs(t64 - tor).max() < 1e-5 and abs(e64 - float(full["config3_energy"])) < 5e-5
    assert abs(e128 - (t128[n:].sum() - t128[:n].sum())) < 1e-9      # the energy is the sum of its terms
    # ... and the GRADIENT against the oracle: four components by central differences of oracle.dense's energy (step 1e-4,
    # float64: accurate to ~1e-8; the reference's convention, tests/test_mpscircuit.py:452-457), fixture config3grad
    assert "config3_grad_fd" in full.files, "full_size_golden.npz has no config3grad part (make_golden_full.py config3grad; merge)"
    comps, fd = full["config3_grad_components"], full["config3_grad_fd"]
    e64g = max(abs(g64[int(r), int(q)] - f) for (r, q), f in zip(comps, fd))
    e128g = max(abs(g128[int(r), int(q)] - f) for (r, q), f in zip(comps, fd))
    print(f"config 3 full size gradient vs oracle.dense central differences ({len(fd)} components, |g| up to "
          f"{np.abs(fd).max():.3f}): max error complex64 {e64g:.2e}, complex128 {e128g:.2e}")
    assert e128g < 5e-8, e128g
    assert e64g < 1e-5, e64g
    assert np.abs(fd).max() > 1e-2
    assert de < 5e-5, (e64, e128)
    assert dg < 4e-5, dg
    assert np.abs(g128).max() > 0.1   # the gradient is not trivially small


def test_config3_bench_call_vvag_against_per_sample_complex128():
    """The exact call bench.py times for config 3 -- ``jit(vvag(energy, argnums=0, vectorized_argnums=0))`` on rows of
    the bench's parameter batch (generator seed 28, normal(0, 0.1)), micro-batch 2 -- against per-sample complex128
    ``value_and_grad``: the batched, traced complex64 pipeline (batched passes, tiled Pauli-sum cotangent with the
    energy from <psi|lambda>, batched adjoint sweep) gives each sample's energy and gradient."""
    import torch
    import tcmi as tc

    n, d, Bg = 28, 12, 32
    params_np = np.random.default_rng(28).normal(0, 0.1, [Bg, 2 * d, n]).astype(np.float32)
    rows = [0, 17]
    tc.set_backend("hip")
    try:
        tc.set_dtype("complex64")
        energy = _energy_fn(tc, n, d)
        vvag = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
        p64 = torch.from_numpy(params_np[rows]).cuda()
        for _ in range(3):          # the third call runs the traced pipeline (the first two validate it)
            v64, g64 = vvag(p64)
        v64 = tc.backend.numpy(v64).astype(np.float64)
        g64 = tc.backend.numpy(g64).astype(np.float64)
        del vvag
        torch.cuda.empty_cache()
        tc.set_dtype("complex128")
        energy = _energy_fn(tc, n, d)
        vag = tc.backend.value_and_grad(energy)
        for k, r in enumerate(rows):
            v, g = vag(tc.backend.convert_to_tensor(params_np[r].astype(np.float64)))
            dv, dg = abs(float(v) - v64[k]), np.abs(tc.backend.numpy(g) - g64[k]).max()
            print(f"config 3 bench call, row {r}: |E64 - E128| {dv:.2e}, max gradient difference {dg:.2e}")
            assert dv < 5e-5 and dg < 4e-5, (r, dv, dg)      # 6-10 x what is measured (8.7e-6 / 4.1e-6, row 0)
            del v, g
            torch.cuda.empty_cache()
    finally:
        tc.set_dtype("complex64")


def test_config2_full_size_state_against_the_dense_oracle():
    """Config 2 at full size (n = 24, depth 8) against ``oracle.dense`` -- a gate-by-gate dense simulator, an algorithm
    independent of both the plan executor and ``oracle.tn`` (256 MiB complex128 on the host): complex64 within 1e-5,
    complex128 within 1e-10 per amplitude (BASELINE.json)."""
    import tcmi as tc

    n, d = 24, 8
    params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n]).astype(np.float32)
    ref = dense.run(n, W.hea_b_ops(n, d, params.astype(np.float64)))
    tc.set_backend("hip")
    try:
        for dtype, tol in (("complex64", 1e-5), ("complex128", 1e-10)):
            tc.set_dtype(dtype)
            rdt = np.float32 if dtype == "complex64" else np.float64
            c = tc.Circuit(n)
            W.hea_b(c, n, d, tc.backend.convert_to_tensor(params.astype(rdt)), zz=tc.gates._zz_matrix)
            got = tc.backend.numpy(c.wavefunction())
            err = np.abs(got - ref).max()
            print(f"config 2 full size vs oracle.dense, {dtype}: max amplitude error {err:.2e}")
            assert err < tol, (dtype, err)
    finally:
        tc.set_dtype("complex64")


def test_sliced_value_and_grad_replays_under_stress():
    """Reduced form of scripts/gpu_vjp_stress.py: the graphs of the sliced reverse sweep (one launch per tree level from a
    static descriptor table, slices in pairs on two streams, traced node function) visited 30 times over three
    parameter points in random order, with allocator churn in between: every result equals the first one of its point."""
    import torch
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    nq, dq = 16, 3
    rng = np.random.default_rng(1)
    pts = [tc.backend.convert_to_tensor(rng.uniform(0.2, 1.2, [nq, dq, 2]).astype(np.float32)) for _ in range(3)]

    def nodes(params):
        c = tc.Circuit(nq)
        for i in range(dq):
            for j in range(nq - 1):
                c.rzz(j, j + 1, theta=params[j, i, 0])
            for j in range(nq):
                c.rx(j, theta=params[j, i, 1])
        return c.expectation_before([tc.gates.z(), [nq // 2]], reuse=False)

    dc = tc.experimental.DistributedContractor(nodes, pts[0], {"slicing_opts": {"target_slices": 8}, "max_repeats": 8,
                                                               "minimize": "combo"})
    ref, junk = {}, []
    for it, k in enumerate(rng.integers(0, 3, 30)):
        v, g = dc.value_and_grad(pts[int(k)])
        junk.append(torch.empty(int(rng.integers(1, 1 << 20)), device="cuda"))
        if it % 5 == 0:
            junk.clear()
            torch.cuda.synchronize()
        if int(k) not in ref:
            ref[int(k)] = (v.clone(), g.clone())
        else:
            assert abs(float(v) - float(ref[int(k)][0])) < 1e-6 and float((g - ref[int(k)][1]).abs().max()) < 1e-6, it
    assert dc._trace_state["mode"] == "replay" and dc.tree._vjp_graph_cache["two"] is not None
    assert float((ref[0][1] - ref[1][1]).abs().max()) > 1e-3


def test_graph_replay_and_two_stream_paths_under_stress():
    """Reduced forms of scripts/gpu_graph_stress.py and scripts/gpu_cut_streams_stress.py (round 2 found real ordering
    bugs on these paths: memset nodes mis-ordered between two replaying graphs): (a) 30 replays of the sliced 32-qubit
    RQC amplitude, with and without the counters hook, equal the eager value; (b) 30 batches through the two-stream cut
    contraction equal the one-stream order bit for bit, with allocator churn between the calls."""
    import os
    import torch
    import tcmi as tc
    from tcmi import tn as TN
    from tcmi.experimental import DistributedContractor

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    rows, cols, depth = 4, 8, 16
    gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(depth * rows * cols)]
    q = lambda r, c: r * cols + c  # noqa: E731

    def nodes_fn(_):
        c = tc.Circuit(rows * cols)
        k = 0
        for dd in range(depth):
            pat = dd % 4
            if pat in (0, 1):
                pairs = [(q(r, cc), q(r, cc + 1)) for r in range(rows) for cc in range(pat, cols - 1, 2)]
            else:
                pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, rows - 1, 2) for cc in range(cols)]
            for a, b in pairs:
                c.any(a, b, unitary=gates[k])
                k += 1
        return c.amplitude_before("0" * (rows * cols))

    dc = DistributedContractor(nodes_fn, None, cotengra_options={"slicing_opts": {"target_size": 2 ** 27}, "max_repeats": 16})
    old = {k: os.environ.get(k) for k in ("TCMI_TN_GRAPH",)}
    old_streams = KN.VALUES.get("cut_streams")
    try:
        os.environ["TCMI_TN_GRAPH"] = "0"
        ref = complex(dc.value(None, op=lambda x: x))
        os.environ["TCMI_TN_GRAPH"] = "1"
        for i in range(30):
            TN.COUNTERS = TN.new_counters() if i % 2 else None
            v = complex(dc.value(None, op=lambda x: x))
            assert abs(v - ref) < 2e-9, (i, v, ref)
        TN.COUNTERS = None
        n, d = 24, 8

        def f(p):
            c = tc.Circuit(n)
            W.hea_b(c, n, d, p, zz=tc.gates._zz_matrix)
            return c.wavefunction()

        fv = tc.backend.jit(tc.backend.vmap(f))
        g = torch.Generator().manual_seed(0)
        for B in (8, 1, 3):
            for i in range(10):
                p = (torch.rand(B, 2 * d, n, generator=g) * 6.28).cuda()
                KN.VALUES["cut_streams"] = "1"
                a = fv(p)
                junk = torch.randn(1 << 22, device="cuda")          # allocator churn between the calls
                KN.VALUES["cut_streams"] = "0"
                b = fv(p)
                del junk
                assert torch.equal(a, b), (B, i, float((a - b).abs().max()))
    finally:
        TN.COUNTERS = None
        KN.VALUES.pop("cut_streams", None)
        if old_streams is not None:
            KN.VALUES["cut_streams"] = old_streams
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_config5_full_size_mps_sweep_properties():
    """SURVEY 8d config 5 at full size (n = 64, chi = 128, one TEBD sweep of random SU(4) gates): canonical form,
    unit norm, bond dimensions, and the complex64 fidelity estimate against a complex128 sweep on the same tensors."""
    import torch
    from scipy.stats import unitary_group
    import tcmi as tc

    n, chi = 64, 128
    rng = np.random.default_rng(64)
    dims = [min(2 ** i, 2 ** (n - i), chi) for i in range(n + 1)]
    tensors = [(rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1])))
               / np.sqrt(2 * dims[i]) for i in range(n)]
    gates = [unitary_group.rvs(4, random_state=5000 + i).reshape(2, 2, 2, 2) for i in range(n - 1)]
    tc.set_backend("hip")
    fid, spec = {}, {}
    try:
        for dtype, cdt in (("complex64", np.complex64), ("complex128", np.complex128)):
            tc.set_dtype(dtype)
            m = tc.MPSCircuit(n, tensors=[t.astype(cdt) for t in tensors],
                              split=tc.cons.split_rules(max_singular_values=chi))
            m.position(0)
            nrm0 = float(abs(m.get_norm()))
            for i in range(n - 1):
                m.apply(tc.gates.Gate(gates[i].astype(cdt)), i, i + 1)
            ts = m.get_tensors()
            assert all(bool(torch.isfinite(t.abs()).all()) for t in ts)
            assert max(m.get_bond_dimensions()) == chi
            tol = 2e-3 if dtype == "complex64" else 1e-9
            # the sweep leaves every site left of the centre left-orthogonal: sum_{l,s} conj(A[l,s,r]) A[l,s,r'] = 1
            assert m.get_center_position() == n - 2   # the last gate leaves the centre on its first site
            for site in (0, 7, 31, 61):
                a = ts[site].to(torch.complex128)
                gram = torch.einsum("lsr,lsq->rq", a.conj(), a)
                assert float((gram - torch.eye(gram.shape[0], dtype=gram.dtype, device=gram.device)).abs().max()) < tol
            fid[dtype] = (float(m._fidelity), nrm0, float(abs(m.get_norm())))
            bd = m.get_bond_dimensions()
            m.position(n // 2)
            a = m.get_tensors()[n // 2].to(torch.complex128).cpu()
            spec[dtype] = (torch.linalg.svdvals(a.reshape(a.shape[0], -1)).numpy(), bd)
    finally:
        tc.set_dtype("complex64")
    f64, f128 = fid["complex64"], fid["complex128"]
    assert abs(f64[0] - f128[0]) < 2e-3 * max(f128[0], 1e-3), (f64, f128)
    # against the ORACLE (oracle.mps: numpy + LAPACK SVD on the same tensors and gates; make_golden_full.py config5):
    # fidelity estimate, norms before / after the sweep, bond dimensions, Schmidt spectrum of the middle bond
    full = _full_golden()
    fo, n0, n1 = float(full["config5_fidelity"]), float(full["config5_norm0"]), float(full["config5_norm1"])
    print(f"config 5 full size vs oracle.mps fixture: fidelity {fo:.9f}; complex128 {f128[0]:.9f}, complex64 {f64[0]:.6f}; "
          f"norm after the sweep {n1:.9f} / {f128[2]:.9f} / {f64[2]:.6f}")
    assert abs(f128[0] - fo) < 1e-8 and abs(f128[1] - n0) < 1e-9 and abs(f128[2] - n1) < 1e-8
    assert abs(f64[0] - fo) < 2e-3 * fo and abs(f64[2] - n1) < 2e-3
    for dtype in ("complex64", "complex128"):
        assert list(spec[dtype][1]) == [int(x) for x in full["config5_bond_dims"]]
        err = np.abs(spec[dtype][0] - full["config5_mid_spectrum"]).max()
        assert err < (2e-4 if dtype == "complex64" else 1e-8), (dtype, err)
    assert 0 < f128[0] < 1
    # truncation only removes weight: the norm after the sweep is the start norm times sqrt of the kept weights
    assert f128[2] <= f128[1] * (1 + 1e-9)


def test_config4_full_size_amplitude_against_statevector():
    """SURVEY 8d config 4 at full size (the bench workload: 32 qubits on a 4x8 grid, depth 16, 448 Haar-random
    two-qubit gates, complex64): the amplitude <0^32|C|0^32> from DistributedContractor (tree search, slicing to 2^27
    elements, 8 slices summed) equals element 0 of the state vector that the plan executor computes for the same
    circuit (34 GB of state, an independent algorithm: tile passes instead of a sliced contraction tree), and the
    state is normalised.  The dense oracle cannot reach n = 32; both routes are pinned to it at n <= 20
    (test_gpu_tn.py::test_config4_grid_rqc_amplitude_sliced, test_gpu_state.py)."""
    import torch
    import tcmi as tc

    tc.set_backend("hip"); tc.set_dtype("complex64")
    rows, cols, depth = 4, 8, 16
    n = rows * cols
    gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(depth * rows * cols)]
    q = lambda r, c: r * cols + c

    def build():
        c = tc.Circuit(n)
        k = 0
        for d in range(depth):
            pat = d % 4
            if pat in (0, 1):
                pairs = [(q(r, cc), q(r, cc + 1)) for r in range(rows) for cc in range(pat, cols - 1, 2)]
            else:
                pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, rows - 1, 2) for cc in range(cols)]
            for a, b in pairs:
                c.any(a, b, unitary=gates[k])
                k += 1
        return c

    dc = tc.experimental.DistributedContractor(lambda _: build().amplitude_before("0" * n), None,
                                               cotengra_options={"slicing_opts": {"target_size": 2**27}, "max_repeats": 128})
    assert dc.tree.nslices >= 2 and dc.tree.max_size() <= 2**27
    v = complex(dc.value(None, op=lambda x: x))
    del dc
    torch.cuda.empty_cache()
    psi = build().wavefunction()
    assert psi.numel() == 2**n
    a0 = complex(psi[0])
    # norm in float64 chunks (a float32 sum of 2^32 terms would lose the digits being checked)
    nrm = 0.0
    for ch in psi.reshape(64, -1):
        nrm += float((ch.real.double() ** 2 + ch.imag.double() ** 2).sum())
    del psi
    torch.cuda.empty_cache()
    assert abs(nrm - 1.0) < 2e-4, nrm
    assert abs(v) > 1e-6                                     # a typical amplitude is 2^-16
    assert abs(v - a0) < 2e-4 * abs(a0), (v, a0)
    # ... and both against the ORACLE: the same amplitude from a numpy complex128 tensordot chain over a sliced pairwise
    # path (tests/golden/make_golden_full.py config4; the chain is checked there against oracle.tn's own greedy
    # contraction at 20 qubits)
    ao = complex(*_full_golden()["config4_amplitude"])
    print(f"config 4 full size vs the numpy oracle fixture: |DistributedContractor - oracle| / |oracle| = {abs(v - ao) / abs(ao):.2e}, "
          f"|state vector[0] - oracle| / |oracle| = {abs(a0 - ao) / abs(ao):.2e}")
    assert abs(v - ao) < 5e-4 * abs(ao) and abs(a0 - ao) < 5e-4 * abs(ao), (v, a0, ao)
