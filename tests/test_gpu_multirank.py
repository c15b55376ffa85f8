"""DistributedContractor with world_size 2 on the GPU (slice shard): results of the two-rank run that
tests/conftest.py starts at session begin (tests/multirank_slices.py) against the single-process contraction."""

import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import MULTIRANK_OUT, MULTIRANK4_OUT  # noqa: E402
from oracle import dense, gates as OG  # noqa: E402


def test_two_rank_slice_shard_equals_single_process():
    assert os.path.exists(MULTIRANK_OUT), "the two-rank run did not produce its result file (see the session log)"
    ranks = json.load(open(MULTIRANK_OUT))
    assert len(ranks) == 2 and ranks[0]["world"] == 2
    r0, r1 = ranks

    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex128")
    try:
        # ---- reference KAT (tests/test_miscs.py:275-304) ----
        c = tc.Circuit(4)
        x, y = np.ones([4]), 0.3 * np.ones([4])
        c.rx(range(4), theta=x)
        c.cnot([0, 1, 2], [1, 2, 3])
        c.ry(range(4), theta=y)
        want = float(tc.backend.numpy(c.expectation_ps(z=[-1])).real)
        for r in ranks:   # the all-reduce leaves the complete result on every rank
            assert abs(r["kat"]["value"] - want) < 1e-6
            assert abs(r["kat"]["value_only"][0] - want) < 1e-6 and abs(r["kat"]["value_only"][1]) < 1e-9
            assert len(r["kat"]["gy"]) == 4 and len(r["kat"]["gx"]) == 4
        np.testing.assert_allclose(r0["kat"]["gx"], r1["kat"]["gx"], atol=1e-12)

        def f(p):
            cc = tc.Circuit(4)
            cc.rx(range(4), theta=p["x"])
            cc.cnot([0, 1, 2], [1, 2, 3])
            cc.ry(range(4), theta=p["y"])
            return tc.backend.real(cc.expectation_ps(z=[-1]))

        _, g = tc.backend.value_and_grad(f)({"x": tc.backend.convert_to_tensor(x), "y": tc.backend.convert_to_tensor(y)})
        np.testing.assert_allclose(r0["kat"]["gx"], tc.backend.numpy(g["x"]), atol=1e-7)
        np.testing.assert_allclose(r0["kat"]["gy"], tc.backend.numpy(g["y"]), atol=1e-7)
        # disjoint shards that together hold every slice (this small network may need a single slice)
        assert sorted(r0["kat"]["mine"] + r1["kat"]["mine"]) == list(range(r0["kat"]["nslices"]))

        # ---- 4x5 RQC amplitude against the dense oracle ----
        rows, cols, depth = 4, 5, 8
        n = rows * cols
        q = lambda r, c: r * cols + c  # noqa: E731
        pairs = []
        for d in range(depth):
            pat = d % 4
            if pat in (0, 1):
                pairs += [(q(r, c), q(r, c + 1)) for r in range(rows) for c in range(pat, cols - 1, 2)]
            else:
                pairs += [(q(r, c), q(r + 1, c)) for r in range(pat - 2, rows - 1, 2) for c in range(cols)]
        mats = [OG.random_two_qubit_gate(900 + k) for k in range(len(pairs))]
        psi = dense.run(n, [(m, [a, b]) for m, (a, b) in zip(mats, pairs)])
        want = dense.amplitude(psi, n, [0] * n)
        for tag in ("rqc_sliced", "rqc_padding"):
            for r in ranks:
                v = complex(*r[tag]["value"])
                assert abs(v - want) < 1e-10 * max(1.0, abs(want)) + 1e-12, (tag, v, want)
        assert r0["rqc_sliced"]["nslices"] >= 8
        assert sorted(r0["rqc_sliced"]["mine"] + r1["rqc_sliced"]["mine"]) == list(range(r0["rqc_sliced"]["nslices"]))
        assert r0["rqc_sliced"]["mine"] and r1["rqc_sliced"]["mine"]   # both ranks contracted slices
        # one slice on two ranks: rank 1 holds only -1 padding and contributes zero (experimental.py:881-890)
        assert r0["rqc_padding"]["nslices"] == 1 and r0["rqc_padding"]["table"] == [[0], [-1]]
        assert r0["rqc_padding"]["mine"] == [0] and r1["rqc_padding"]["mine"] == []

        # ---- output-wavefunction slicing: the mask sum sharded over the two ranks equals expectation_ps ----
        from oracle import workloads as OW

        nw, dw = 12, 3
        pw = tc.backend.convert_to_tensor(np.random.default_rng(4).uniform(0, 2 * np.pi, [2 * dw, nw]))
        cw = tc.Circuit(nw)
        OW.hea_b(cw, nw, dw, pw, zz=tc.gates._zz_matrix)
        want = float(tc.backend.numpy(tc.backend.real(cw.expectation_ps(x=[1, 10], y=[5], z=[4, 9]))))
        for r in ranks:
            assert abs(r["wfslice"] - want) < 1e-10, (r["wfslice"], want)
    finally:
        tc.set_dtype("complex64")


def test_four_rank_sliced_value_and_grad_equals_the_adjoint_path():
    """World 4 (gloo on the one visible device, nccl when four are): the rzz / rx ladder of
    examples/slicing_auto_pmap_vqa.py at 16 qubits, sliced with ``slicing_opts={"target_slices": 8}`` -- every rank
    holds at least two slices -- and differentiated by the reverse sweep of ``tn.contract_slices_vjp`` (forward and
    backward on tcmi_tensordot_bits / tcmi_contract_scattered / tcmi_cgemm) against ``backend.value_and_grad`` of the
    same energy on the state-vector adjoint path of ONE process.  Tolerances: BASELINE.json (1e-5 / 1e-10)."""
    assert os.path.exists(MULTIRANK4_OUT), "the four-rank run did not produce its result file (see the session log)"
    ranks = json.load(open(MULTIRANK4_OUT))
    assert len(ranks) == 4 and all(r["world"] == 4 for r in ranks)

    import tcmi as tc

    tc.set_backend("hip")
    nq, dq = 16, 4
    pv = np.random.default_rng(5).uniform(0.2, 1.2, [nq, dq, 2])

    def energy(params):
        c = tc.Circuit(nq)
        for i in range(dq):
            for j in range(nq - 1):
                c.rzz(j, j + 1, theta=params[j, i, 0])
            for j in range(nq):
                c.rx(j, theta=params[j, i, 1])
        return tc.backend.real(c.expectation_ps(z=[0]))

    try:
        for dt, tol in (("complex64", 1e-5), ("complex128", 1e-10)):
            tc.set_dtype(dt)
            pt = tc.backend.convert_to_tensor(pv.astype(np.float32 if dt == "complex64" else np.float64))
            v, g = tc.backend.value_and_grad(energy)(pt)
            v, g = float(v), tc.backend.numpy(g).astype(np.float64)
            key = "vqa_" + dt
            assert ranks[0][key]["nslices"] >= 8
            held = sorted(sum((r[key]["mine"] for r in ranks), []))
            assert held == list(range(ranks[0][key]["nslices"])) and all(len(r[key]["mine"]) >= 2 for r in ranks)
            errs = []
            for r in ranks:          # the packed all-reduce leaves value and gradient on every rank
                assert abs(r[key]["value"] - v) < tol, (dt, r[key]["value"], v)
                assert abs(r[key]["value_only"] - v) < tol
                errs.append(np.abs(np.asarray(r[key]["grad"]) - g).max())
            print(f"sliced vqa {dt}: value {v:.8f}, max |grad - adjoint| over ranks {max(errs):.2e}")
            assert max(errs) < tol
            assert np.abs(g).max() > 1e-3      # a gradient that is really there
        # fewer slices than ranks: two ranks hold only -1 padding and still make the sharded sweep's collectives
        # (complex128 is the dtype the loop above ends on)
        few = [r["vqa_fewslices"] for r in ranks]
        assert few[0]["nslices"] == 2 and [len(f["mine"]) for f in few] == [1, 1, 0, 0], [f["mine"] for f in few]
        for f in few:
            assert abs(f["value"] - v) < 1e-10 and abs(f["value2"] - v) < 1e-10
            assert np.abs(np.asarray(f["grad"]) - g).max() < 1e-10
            assert np.abs(np.asarray(f["grad2"]) - g).max() < 1e-10
    finally:
        tc.set_dtype("complex64")


def test_abi_collective_on_a_world_of_one():
    """include/tcmi.h tcmi_comm_* / tcmi_allreduce_sum: RCCL opened by libtcmi.so itself, a communicator of ONE
    rank on this box's GPU (more ranks need more devices: RCCL refuses two ranks on one), all-reduce(SUM) in place for the
    four dtypes -- the identity on one rank, bit for bit -- ordered on the caller's stream."""
    import torch
    from tcmi import distributed as D

    uid = D.AbiCommunicator.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = D.AbiCommunicator(uid, 0, 1)
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        for dt in (torch.float32, torch.float64, torch.complex64, torch.complex128):
            x = torch.randn(1000, device="cuda", dtype=dt, generator=g)
            want = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                y = x * 2                     # ordered before the collective on the same stream
                comm.allreduce_sum_(y)
                y = y / 2
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(y, want)
        with pytest.raises(ValueError):
            comm.allreduce_sum_(torch.zeros(4, dtype=torch.int32, device="cuda"))
    finally:
        comm.close()
