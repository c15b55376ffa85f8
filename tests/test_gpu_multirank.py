"""DistributedContractor with world_size 2 on the GPU (slice shard): results of the two-rank run that
tests/conftest.py starts at session begin (tests/multirank_slices.py) against the single-process contraction."""

import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import MULTIRANK_OUT  # noqa: E402
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
    finally:
        tc.set_dtype("complex64")
