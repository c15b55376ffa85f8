"""Slice-sharded DistributedContractor on more than one rank (reference tensorcircuit/experimental.py:881-894,
1125-1152; examples/slicing_auto_pmap_vqa.py:60-72).

    python tests/multirank_slices.py <world> <out.json>      (launcher: spawns the ranks, never touches the GPU)
    python tests/multirank_slices.py --rank R <world> <port> <out.json>

Every rank is its own process with its own GPU context (`nccl` = RCCL when the box shows at least `world`
devices, otherwise `gloo` with every rank on device 0 -- the one-GPU box of the test tier).  Each rank builds the
same deterministic tree, contracts only its rows of the slice table and joins the packed all-reduce; rank 0 writes
the results.  Cases: the reference KAT (tests/test_miscs.py:275-304), a 4x5 grid RQC amplitude with 8+ slices, and
the same RQC unsliced, where rank 1 holds nothing but -1 padding (experimental.py:881-890, 1050-1052)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launcher(world, out):
    port = 29600 + (os.getpid() % 2000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(world), str(port), out],
                              env=env) for r in range(world)]
    rc = 0
    for p in procs:
        try:
            rc |= p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 1
    return rc


def worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    backend = "nccl" if ndev >= world else "gloo"
    torch.cuda.set_device(rank % max(ndev, 1))
    dist.init_process_group(backend, rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    import tcmi as tc
    from oracle import gates as OG

    tc.set_backend("hip")
    tc.set_dtype("complex128")
    res = {"backend": backend, "world": world}

    # ---- reference KAT: 4 qubits, target_size 2**3 ----
    def nodes_fn(params):
        c = tc.Circuit(4)
        c.rx(range(4), theta=params["x"])
        c.cnot([0, 1, 2], [1, 2, 3])
        c.ry(range(4), theta=params["y"])
        return c.expectation_before([tc.gates.z(), [-1]], reuse=False)

    params = {"x": np.ones([4]), "y": 0.3 * np.ones([4])}
    dc = tc.experimental.DistributedContractor(
        nodes_fn, params, {"slicing_reconf_opts": {"target_size": 2**3}, "max_repeats": 8, "minimize": "write", "parallel": False})
    v, g = dc.value_and_grad(params)
    res["kat"] = {"nslices": int(dc.tree.nslices), "mine": dc.my_slices, "value": float(v),
                  "gx": tc.backend.numpy(g["x"]).tolist(), "gy": tc.backend.numpy(g["y"]).tolist(),
                  "value_only": [float(dc.value(params).real), float(dc.value(params).imag)]}

    # ---- 4x5 RQC amplitude, sliced and unsliced ----
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

    def rqc_nodes(_):
        c = tc.Circuit(n)
        for m, (a, b) in zip(mats, pairs):
            c.any(a, b, unitary=m.reshape(2, 2, 2, 2))
        return c.amplitude_before("0" * n)

    for tag, tgt in (("rqc_sliced", 2**9), ("rqc_padding", 2**30)):
        dc = tc.experimental.DistributedContractor(rqc_nodes, None, {"slicing_opts": {"target_size": tgt}, "max_repeats": 16})
        val = complex(dc.value(None, op=lambda x: x))
        res[tag] = {"nslices": int(dc.tree.nslices), "mine": dc.my_slices, "table": dc.slice_table.tolist(),
                    "value": [val.real, val.imag]}
    # ---- the north-star workload (examples/slicing_auto_pmap_vqa.py:20-41,86-94): rzz / rx ladder, <Z_0>, sliced to
    # one-or-more slices per rank with ``target_slices``; value_and_grad through the hand-written reverse sweep ----
    nq, dq = 16, 4
    pv = np.random.default_rng(5).uniform(0.2, 1.2, [nq, dq, 2])

    def vqa_nodes(params):
        c = tc.Circuit(nq)
        for i in range(dq):
            for j in range(nq - 1):
                c.rzz(j, j + 1, theta=params[j, i, 0])
            for j in range(nq):
                c.rx(j, theta=params[j, i, 1])
        return c.expectation_before([tc.gates.z(), [0]], reuse=False)

    for dt in ("complex64", "complex128"):
        tc.set_dtype(dt)
        pt = tc.backend.convert_to_tensor(pv.astype(np.float32 if dt == "complex64" else np.float64))
        dc = tc.experimental.DistributedContractor(
            vqa_nodes, pt, {"slicing_opts": {"target_slices": 8}, "max_repeats": 16, "minimize": "combo", "parallel": True})
        v, g = dc.value_and_grad(pt)
        res["vqa_" + dt] = {"nslices": int(dc.tree.nslices), "mine": dc.my_slices, "value": float(v),
                            "grad": tc.backend.numpy(g).astype(np.float64).tolist(),
                            "value_only": float(dc.value(pt).real)}
    # ---- fewer slices than ranks (slice_table fills rows first: with world 4 ranks 2 and 3 hold only padding): the
    # padding ranks must still make the sharded sweep's collectives (all-gather of the invariant roots, all-reduce of
    # their cotangents) or the job hangs ----
    dc = tc.experimental.DistributedContractor(
        vqa_nodes, pt, {"slicing_opts": {"target_slices": 2}, "max_repeats": 16, "minimize": "combo", "parallel": True})
    v, g = dc.value_and_grad(pt)
    v2, g2 = dc.value_and_grad(pt)      # replay of the captured graphs
    res["vqa_fewslices"] = {"nslices": int(dc.tree.nslices), "mine": dc.my_slices, "value": float(v), "value2": float(v2),
                            "grad": tc.backend.numpy(g).astype(np.float64).tolist(),
                            "grad2": tc.backend.numpy(g2).astype(np.float64).tolist()}
    # ---- output-wavefunction slicing (examples/slicing_wavefunction_vqa.py): the 2^3 projections of three cut qubits
    # dealt to the ranks, one all-reduce ----
    tc.set_dtype("complex128")
    from tcmi.experimental import sliced_expectation_ps
    from oracle import workloads as OW

    nw, dw = 12, 3
    pw = tc.backend.convert_to_tensor(np.random.default_rng(4).uniform(0, 2 * np.pi, [2 * dw, nw]))

    def wcirc():
        c = tc.Circuit(nw)
        OW.hea_b(c, nw, dw, pw, zz=tc.gates._zz_matrix)
        return c

    res["wfslice"] = float(sliced_expectation_ps(wcirc, [0, 1, 0, 0, 3, 2, 0, 0, 0, 3, 1, 0], [1, 5, 9]))
    allres = [None] * world
    dist.all_gather_object(allres, res)
    if rank == 0:
        with open(out, "w") as f:
            json.dump(allres, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "--rank":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        sys.exit(launcher(int(sys.argv[1]), sys.argv[2]))
