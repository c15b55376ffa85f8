"""N > 1 path on CPU: world size 2, gloo backend.  Each rank evaluates its shard of a vmap batch
(energies and gradients of the shared parameters from the CPU oracle -- the GPU kernels are not
involved here, only the sharding / packing / reduction logic of tcmi.distributed that bench.py and
the VQE step use) and the packed all-reduce reproduces the single-process sums."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from tcmi import distributed as D
    from oracle import dense, workloads as W

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, B = 4, 1, 5
    rng = np.random.default_rng(0)
    shared = rng.normal(size=[2 * d, n])
    xs = rng.normal(size=[B, n])

    def f(x, w):
        from oracle import gates as G
        return W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, w) + [(G.rx(x[i]), [i]) for i in range(n)]), n)

    lo, hi = D.shard_range(B, rank, world)
    vals = np.zeros(B)
    gw = np.zeros_like(shared)
    eps = 1e-6
    for b in range(lo, hi):
        vals[b] = f(xs[b], shared)
        for i in np.ndindex(*shared.shape):
            wp, wm = shared.copy(), shared.copy()
            wp[i] += eps; wm[i] -= eps
            gw[i] += (f(xs[b], wp) - f(xs[b], wm)) / (2 * eps)
    v_t, g_t = D.allreduce_sum_packed([torch.from_numpy(vals), torch.from_numpy(gw)])
    q.put((rank, v_t.numpy(), g_t.numpy(), (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_batch_shard_allreduce():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # both ranks hold identical, complete results
    np.testing.assert_allclose(res[0][1], res[1][1])
    np.testing.assert_allclose(res[0][2], res[1][2])
    assert res[0][3] == (0, 3) and res[1][3] == (3, 5)
    # and they equal the single-process evaluation
    sys.path.insert(0, ROOT)
    from oracle import dense, gates as G, workloads as W

    n, d, B = 4, 1, 5
    rng = np.random.default_rng(0)
    shared = rng.normal(size=[2 * d, n])
    xs = rng.normal(size=[B, n])
    want = [W.tfim_energy_dense(dense.run(n, W.hea_b_ops(n, d, shared) + [(G.rx(x[i]), [i]) for i in range(n)]), n) for x in xs]
    np.testing.assert_allclose(res[0][1], want, atol=1e-12)
    assert np.isfinite(res[0][2]).all() and np.abs(res[0][2]).max() > 0


def test_slice_table_matches_reference_layout():
    """reference experimental.py:881-890: ceil split, row-major ids, -1 padding."""
    sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
    from tcmi import distributed as D

    t = D.slice_table(10, 4)
    assert t.shape == (4, 3)
    assert t.tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, -1, -1]]
    assert D.slice_table(8, 8).tolist() == [[i] for i in range(8)]
    assert D.slice_table(3, 8)[3:].tolist() == [[-1]] * 5
    assert [D.shard_range(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    assert D.shard_range(5, 3, 4) == (5, 5)  # empty shard


def _search_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
    sys.path.insert(0, ROOT)
    import pickle

    import torch.distributed as dist
    from tcmi import tn
    from tcmi.experimental import DistributedContractor as DC

    def nodes_fn(_):
        # a 3 x 4 grid of rank-4 tensors with periodic columns (closed network), dimension 2 everywhere
        rows, cols = 3, 4
        h = {(r, c): tn.new_edge() for r in range(rows) for c in range(cols)}
        v = {(r, c): tn.new_edge() for r in range(rows - 1) for c in range(cols)}
        nodes = []
        rng = np.random.default_rng(1)
        for r in range(rows):
            for c in range(cols):
                es = [h[(r, c)], h[(r, (c - 1) % cols)]]
                if r > 0:
                    es.append(v[(r - 1, c)])
                if r < rows - 1:
                    es.append(v[(r, c)])
                nodes.append(tn.Node(rng.normal(size=[2] * len(es)) + 0j, es))
        return nodes

    opts = {"seed": [0, 1, 2], "max_repeats": 4, "slicing_opts": {"target_size": 8}}
    os.environ["TCMI_TREE_CACHE"] = "0"          # every call below has to SEARCH (the cache has its own test)
    os.environ["TCMI_TN_SEARCH_SHARD"] = "0"
    serial = DC._get_tree_data(nodes_fn, None, opts)           # before the group exists: the plain serial search
    serial_stats = list(DC.last_search)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = DC._get_tree_data(nodes_fn, None, opts)            # switched off: every rank searches every seed, no collective
    os.environ["TCMI_TN_SEARCH_SHARD"] = "1"
    shared = DC._get_tree_data(nodes_fn, None, opts)           # seeds dealt to the ranks, the best tree broadcast
    stats = list(DC.last_search)
    q.put((rank, pickle.dumps(serial), pickle.dumps(local), pickle.dumps(shared),
           [(s["seed"], s["rank"], s["objective"]) for s in stats], [(s["seed"], s["objective"]) for s in serial_stats]))
    dist.destroy_process_group()


def test_path_search_seeds_are_dealt_to_the_ranks_and_the_best_tree_is_broadcast():
    """reference experimental.py:850-857 (rank 0 searches, broadcast_py_object): here the seeds of the hyper-search are
    dealt to the ranks; every rank ends with the tree the serial search over all seeds picks."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_search_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, serial, local, shared, stats, serial_stats in res:
        assert serial == local == shared == res[0][1]
        assert [(s, r) for s, r, _ in stats] == [(0, 0), (1, 1), (2, 0)]            # seed k searched by rank k mod 2
        assert [(s, o) for s, _, o in stats] == serial_stats                        # the same objectives, seed by seed
