"""Symbolic part of the tensor-network layer on CPU: greedy path, slicing, slice tables.  The device
tensordot/permute are replaced by numpy stand-ins (test infrastructure) so that
``ContractionTree.contract_core`` can be executed without a GPU and compared with the oracle."""

import numpy as np
import pytest

import tcmi as tc
from tcmi import tn
from tcmi import _knobs as KN
from oracle import dense, gates as G, workloads as W


@pytest.fixture
def numpy_engine(monkeypatch):
    monkeypatch.setattr(tn, "tensordot", lambda a, b, xa, xb: np.tensordot(a, b, axes=(list(xa), list(xb))))
    monkeypatch.setattr(tn, "permute", lambda t, perm: np.transpose(t, perm))
    yield


def _closed_network(n, depth, seed, bits=None, op=None):
    """A brickwork circuit as plain (numpy tensor, edges) nodes with the reference wiring, closed
    either by one-hot caps (amplitude) or by <psi|op|psi>."""
    rng = np.random.default_rng(seed)
    ops = []
    for d in range(depth):
        for i in range(d % 2, n - 1, 2):
            ops.append((G.random_two_qubit_gate(int(rng.integers(1 << 30))), [i, i + 1]))
        for i in range(n):
            ops.append((G.rx(float(rng.uniform(0, 6))), [i]))

    def ket(conj):
        nodes, front = [], []
        for q in range(n):
            e = tn.new_edge()
            nodes.append(tn.Node(np.array([1.0, 0.0], dtype=np.complex128), [e]))
            front.append(e)
        for m, qs in ops:
            k = len(qs)
            t = np.asarray(m, dtype=np.complex128).reshape([2] * (2 * k))
            oe = [tn.new_edge() for _ in range(k)]
            nodes.append(tn.Node(t.conj() if conj else t, oe + [front[q] for q in qs]))
            for j, q in enumerate(qs):
                front[q] = oe[j]
        return nodes, front

    nodes, front = ket(False)
    psi = dense.run(n, ops)
    if bits is not None:
        for b, e in zip(bits, front):
            v = np.zeros(2, dtype=np.complex128)
            v[b] = 1
            nodes.append(tn.Node(v, [e]))
        want = dense.amplitude(psi, n, bits)
    else:
        n2, f2 = ket(True)
        nodes += n2
        q = op[1]
        t = np.asarray(op[0], dtype=np.complex128).reshape(2, 2)
        nodes.append(tn.Node(t, [f2[q], front[q]]))
        ren = {f2[j]: front[j] for j in range(n) if j != q}
        for nd in nodes:
            nd.edges = [ren.get(e, e) for e in nd.edges]
        want = dense.expectation(psi, n, (op[0], [q]))
    return nodes, want


def test_greedy_path_contracts_closed_network(numpy_engine):
    nodes, want = _closed_network(8, 4, 0, bits=[0, 1, 1, 0, 1, 0, 0, 1])
    inputs, output, size_dict = tn.get_tn_info(nodes)
    assert output == []
    tree = tn.ContractionTree.from_path(inputs, output, size_dict)
    assert len(tree.path) == len(nodes) - 1
    got = tree.contract_core([nd.tensor for nd in nodes])
    np.testing.assert_allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("target", [2**3, 2**5, 2**7])
def test_sliced_sum_equals_unsliced(numpy_engine, target):
    """reference tests/test_miscs.py:275-304: forcing real slicing with a tiny target_size, the sum
    over all slices reproduces the unsliced value."""
    nodes, want = _closed_network(7, 4, 1, op=(G.Z, 6))
    inputs, output, size_dict = tn.get_tn_info(nodes)
    tree = tn.ContractionTree.from_path(inputs, output, size_dict).slice_to(target)
    assert tree.max_size() <= target and (tree.nslices >= 2 or target >= 2**7)
    arrays = [nd.tensor for nd in nodes]
    total = sum(tree.contract_core(tree.slice_arrays(arrays, i)) for i in range(tree.nslices))
    np.testing.assert_allclose(total, want, atol=1e-12)
    # cost model: slicing never lowers total flops, and the data round-trips
    t0 = tn.ContractionTree.from_path(inputs, output, size_dict)
    assert tree.total_flops() >= t0.total_flops()
    t2 = tn.ContractionTree.from_data(tree.to_data())
    assert t2.sliced_inds == tree.sliced_inds and t2.path == tree.path and t2.nslices == tree.nslices


def test_open_network_output_order(numpy_engine):
    """State contraction with dangling legs: result axes follow the requested edge order
    (reference cons.py:958-960)."""
    n = 5
    rng = np.random.default_rng(3)
    ops = [(G.H, [i]) for i in range(n)] + [(G.random_two_qubit_gate(7), [0, 3]), (G.CNOT, [4, 1]), (G.ry(0.4), [2])]
    nodes, front = [], []
    for q in range(n):
        e = tn.new_edge()
        nodes.append(tn.Node(np.array([1.0, 0.0], dtype=np.complex128), [e]))
        front.append(e)
    for m, qs in ops:
        k = len(qs)
        oe = [tn.new_edge() for _ in range(k)]
        nodes.append(tn.Node(np.asarray(m, dtype=np.complex128).reshape([2] * (2 * k)), oe + [front[q] for q in qs]))
        for j, q in enumerate(qs):
            front[q] = oe[j]
    inputs, output, size_dict = tn.get_tn_info(nodes)
    assert sorted(output) == sorted(front)
    tree = tn.ContractionTree.from_path(inputs, front, size_dict)
    psi = tree.contract_core([nd.tensor for nd in nodes]).reshape(-1)
    np.testing.assert_allclose(psi, dense.run(n, ops), atol=1e-12)
    rev = list(reversed(front))
    tree = tn.ContractionTree.from_path(inputs, rev, size_dict)
    psi_r = tree.contract_core([nd.tensor for nd in nodes])
    np.testing.assert_allclose(np.transpose(psi_r, list(reversed(range(n)))).reshape(-1), dense.run(n, ops), atol=1e-12)


def test_circuit_tn_structure():
    """Circuit.amplitude_before / expectation_before produce the reference's node counts:
    2n + gates (+ caps) for an amplitude, 2(n + gates) + ops for reuse=False."""
    import torch

    if not torch.cuda.is_available():
        # node construction touches the device; without a GPU only the symbolic helpers are checked
        t = tn.ContractionTree.from_path([[0, 1], [1, 2], [2, 0]], [], {0: 2, 1: 2, 2: 2})
        assert len(t.path) == 2 and t.max_size() <= 4
        return


# ---- subtree reconfiguration -------------------------------------------------------------------------
def _rand_net(nt, deg, seed, dangling=2):
    rng = np.random.default_rng(seed)
    inputs = [[] for _ in range(nt)]
    e = 0
    stubs = [i for i in range(nt) for _ in range(deg)]
    rng.shuffle(stubs)
    for a, b in zip(stubs[0::2], stubs[1::2]):
        if a != b:
            inputs[a].append(e)
            inputs[b].append(e)
            e += 1
    output = []
    for i in range(dangling):
        inputs[i].append(e)
        output.append(e)
        e += 1
    return inputs, output, {k: 2 for k in range(e)}


@pytest.mark.parametrize("nt,seed", [(12, 0), (30, 1), (60, 2)])
def test_reconfigure_path_is_valid_and_not_worse(nt, seed):
    from tcmi import tn

    inputs, output, sd = _rand_net(nt, 3, seed)
    p0 = tn.greedy_path(inputs, output, sd)
    m0, f0 = tn._path_stats(inputs, output, sd, p0)
    p1 = tn.reconfigure_path(inputs, output, sd, p0, subtree_size=8)
    m1, f1 = tn._path_stats(inputs, output, sd, p1)
    assert len(p1) == len(p0) and f1 <= f0
    # same value: contract with numpy along both paths
    rng = np.random.default_rng(seed)
    arrays = [rng.normal(size=[2] * len(s)) + 1j * rng.normal(size=[2] * len(s)) for s in inputs]

    def run(path):
        cur = [(a, list(s)) for a, s in zip(arrays, inputs)]
        for a, b in path:
            (tb, sb), (ta, sa) = cur.pop(b), cur.pop(a)
            live = {e for t, s in cur for e in s} | set(output)
            con = [e for e in sa if e in sb and e not in live]
            t = np.tensordot(ta, tb, ([sa.index(e) for e in con], [sb.index(e) for e in con]))
            cur.append((t, [e for e in sa if e not in con] + [e for e in sb if e not in con]))
        t, s = cur[0]
        return np.transpose(t, [s.index(e) for e in output])

    np.testing.assert_allclose(run(p1), run(p0), rtol=1e-9, atol=1e-9)
    # with a size cap no intermediate exceeds it as long as the input path respected it
    p2 = tn.reconfigure_path(inputs, output, sd, p0, subtree_size=8, max_size=m0)
    m2, f2 = tn._path_stats(inputs, output, sd, p2)
    assert m2 <= m0 and f2 <= f0


@pytest.mark.parametrize("nt,seed", [(30, 1), (60, 2), (90, 3)])
def test_native_subtree_programme_gives_the_python_loops_paths(nt, seed, monkeypatch):
    """``tcmi_subtree_dp`` and, since round 6, the whole loop around it (``tcmi_reconfigure_path``: frontier expansion,
    cost comparison, rebuild, work list) replace the Python loops of reconfigure_path: same costs, same tie-breaking, same
    node numbering, hence the identical path -- with and without a size cap, with the bytes weight of the sliced search,
    with dimensions other than 2 (weighted index sizes), at subtree sizes 6, 8 and 10.  (``tn_native_dp = 0`` switches both
    off: the pure-Python loop is the comparison.)"""
    from tcmi import tn

    assert tn._native_subtree_dp() is not None            # the library is built: this is the path the product takes
    assert tn._native_reconfigure() is not None
    inputs, output, sd = _rand_net(nt, 3, seed)
    sd3 = {k: (3 if k % 5 == 0 else 2) for k in sd}
    p0 = tn.greedy_path(inputs, output, sd)
    m0, _ = tn._path_stats(inputs, output, sd, p0)
    for dims, kw in ((sd, dict(subtree_size=8)), (sd, dict(subtree_size=10, max_size=m0)),
                     (sd, dict(subtree_size=6, alpha=32.0)), (sd3, dict(subtree_size=8, alpha=16.0))):
        monkeypatch.setitem(KN.VALUES, "tn_native_dp", "1")
        fast = tn.reconfigure_path(inputs, output, dims, p0, **kw)
        monkeypatch.setitem(KN.VALUES, "tn_native_dp", "0")
        slow = tn.reconfigure_path(inputs, output, dims, p0, **kw)
        assert fast == slow


@pytest.mark.parametrize("nt,seed", [(40, 1), (80, 4)])
def test_native_slicing_of_a_fixed_tree_picks_the_python_loops_indices(nt, seed, monkeypatch):
    """``tcmi_slice_fixed`` replaces the Python loop of ContractionTree._slice_fixed (dimension-2 networks): same scores,
    same candidate order, same choice -- identical sliced indices and flops for several targets, and the same verdict
    when the target cannot be reached within the slice budget."""
    from tcmi import tn

    assert tn._native_slice_fixed() is not None
    inputs, output, sd = _rand_net(nt, 3, seed)
    tree = tn.ContractionTree.from_path(inputs, output, sd, trials=0, seed=0)
    width = int(np.log2(tree.max_size()))
    for target_bits, max_slices in ((width - 1, 1 << 16), (width - 3, 1 << 16), (max(2, width - 6), 1 << 16), (2, 4)):
        monkeypatch.setitem(KN.VALUES, "tn_native_slice", "1")
        fast = tree._slice_fixed(list(tree.path), 2**target_bits, max_slices, 12)
        monkeypatch.setitem(KN.VALUES, "tn_native_slice", "0")
        slow = tree._slice_fixed(list(tree.path), 2**target_bits, max_slices, 12)
        assert fast == slow, (target_bits, fast, slow)
    assert slow is None or len(slow[0]) <= 2          # the last case has a budget of four slices


def test_reconfigure_improves_a_poor_path():
    from tcmi import tn

    inputs, output, sd = _rand_net(80, 3, 5, dangling=0)
    rng = np.random.default_rng(0)
    p0 = tn.greedy_path(inputs, output, sd, temperature=1.0, alpha=0.0, rng=rng)
    _, f0 = tn._path_stats(inputs, output, sd, p0)
    _, f1 = tn._path_stats(inputs, output, sd, tn.reconfigure_path(inputs, output, sd, p0, subtree_size=8))
    assert f1 < 0.5 * f0


def test_sliced_tree_is_reconfigured_within_the_cap():
    from tcmi import tn

    inputs, output, sd = _rand_net(90, 3, 9, dangling=0)
    tree = tn.ContractionTree.from_path(inputs, output, sd, trials=8, seed=1)
    target = max(4, tree.max_size() // 16)
    tree.slice_to(target)
    assert tree.max_size() <= target and tree.nslices > 1
    ref = tn.ContractionTree.from_path(inputs, output, sd, path=tn.greedy_path(inputs, output, sd))
    ref.trials = 0
    ref.slice_to(target)
    assert tree.total_flops() <= ref.total_flops()


def test_contract_slices_reuses_invariant_intermediates(monkeypatch):
    """contract_slices == contract_core per slice, with every slice-independent product computed once."""
    from tcmi import tn

    inputs, output, sd = _rand_net(40, 3, 11, dangling=1)
    tree = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
    tree.slice_to(max(4, tree.max_size() // 8))
    assert tree.nslices >= 4
    rng = np.random.default_rng(3)
    import torch
    arrays = [torch.from_numpy(rng.normal(size=[2] * len(s)) + 1j * rng.normal(size=[2] * len(s))) for s in inputs]
    calls = []

    def td(a, b, xa, xb):
        calls.append(1)
        return torch.tensordot(a, b, (list(xa), list(xb)))

    monkeypatch.setattr(tn, "tensordot", td)
    monkeypatch.setattr(tn, "permute", lambda t, perm: t.permute(*perm).contiguous())
    per_slice = [tree.contract_core(tree.slice_arrays(arrays, i)) for i in range(tree.nslices)]
    n_plain = len(calls)
    calls.clear()
    got = list(tree.contract_slices(arrays, range(tree.nslices)))
    assert len(calls) < n_plain            # shared subtrees were not recomputed
    for a, b in zip(got, per_slice):
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-12)
    steps, dep, _, _ = tree._symbolic_steps()
    n_shared = sum(1 for st in steps if not dep[st[4]])
    assert len(calls) == n_shared + tree.nslices * (len(steps) - n_shared)


def test_path_search_is_identical_across_processes():
    """Every rank of a distributed contraction derives the tree itself: search, slicing and reconfiguration must
    not depend on hash seeds or timing."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, json, hashlib; sys.path[:0] = %r\n"
        "from test_tn_host import _rand_net\n"
        "from tcmi import tn\n"
        "inputs, output, sd = _rand_net(70, 3, 21, dangling=0)\n"
        "t = tn.ContractionTree.from_path(inputs, output, sd, trials=6, seed=3)\n"
        "t.slice_to(max(4, t.max_size() // 8))\n"
        "print(hashlib.sha1(json.dumps([t.path, t.sliced_inds]).encode()).hexdigest())\n"
    ) % ([os.path.dirname(__file__)] + [p for p in sys.path if p.endswith("tensorcircuit-ng_amd") or p.endswith("repo")],)
    outs = []
    for seed in ("1", "4242"):
        env = dict(os.environ, PYTHONHASHSEED=seed)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("dangling", [0, 2])
def test_hand_written_reverse_sweep_equals_the_framework_tape(monkeypatch, dangling):
    """``contract_slices_vjp`` (the sliced value_and_grad route of DistributedContractor, reference
    experimental.py:1182-1211) against torch autograd through the same tree: torch stand-ins replace the two raw
    device launches, so this checks the axis bookkeeping of ``tensordot_vjp``, the per-slice scatter of the leaf
    cotangents and the once-only pull-back through the slice-invariant subtrees."""
    import torch
    from tcmi import tn

    inputs, output, sd = _rand_net(36, 3, 5, dangling=dangling)
    tree = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
    tree.slice_to(max(4, tree.max_size() // 8))
    assert tree.nslices >= 4
    rng = np.random.default_rng(9)
    arrays = [torch.from_numpy(rng.normal(size=[2] * len(s)) + 1j * rng.normal(size=[2] * len(s))).requires_grad_(k % 3 != 1)
              for k, s in enumerate(inputs)]
    monkeypatch.setattr(tn, "_tensordot_raw", lambda a, b, xa, xb: torch.tensordot(a, b, (list(xa), list(xb))))
    monkeypatch.setattr(tn, "_permute_raw", lambda t, perm: t.permute(*perm).contiguous())
    wts = torch.from_numpy(np.asarray(rng.normal(size=[2] * len(tree.output)) + 1j * rng.normal(size=[2] * len(tree.output))))
    fop = lambda x: (x * wts).sum().real
    ids = list(range(tree.nslices))
    total, grads = tree.contract_slices_vjp(arrays, ids, fop)
    # reference: the framework tape through contract_core, slice by slice
    monkeypatch.setattr(tn, "tensordot", lambda a, b, xa, xb: torch.tensordot(a, b, (list(xa), list(xb))))
    monkeypatch.setattr(tn, "permute", lambda t, perm: t.permute(*perm).contiguous())
    want = sum(fop(tree.contract_core(tree.slice_arrays(arrays, i))) for i in ids)
    wg = torch.autograd.grad(want, [a for a in arrays if a.requires_grad])
    np.testing.assert_allclose(float(total), float(want), rtol=1e-12)
    it = iter(wg)
    for a, g in zip(arrays, grads):
        if not a.requires_grad:
            assert g is None
            continue
        np.testing.assert_allclose(g.numpy(), next(it).numpy(), atol=1e-10)
    # a subset of the slices (what one rank of a slice shard holds)
    t2, g2 = tree.contract_slices_vjp(arrays, ids[1:3], fop)
    w2 = sum(fop(tree.contract_core(tree.slice_arrays(arrays, i))) for i in ids[1:3])
    np.testing.assert_allclose(float(t2), float(w2), rtol=1e-12)


def test_engine_ops_work_under_torch_func_transforms(monkeypatch):
    """``tn.tensordot`` / ``tn.permute`` on operands inside ``torch.func.grad`` / ``vmap`` (how backend.value_and_grad,
    vmap and vvag reach ``K.tensordot`` / ``K.transpose`` / ``K.einsum``): the two autograd Functions carry
    ``setup_context`` and a vmap rule.  Raw launches replaced by torch stand-ins."""
    import torch
    from tcmi import tn

    monkeypatch.setattr(tn, "_tensordot_raw", lambda a, b, xa, xb: torch.tensordot(a, b, (list(xa), list(xb))))
    monkeypatch.setattr(tn, "_permute_raw", lambda t, perm: t.permute(*perm).contiguous())
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.normal(size=[2] * 5) + 1j * rng.normal(size=[2] * 5))
    b = torch.from_numpy(rng.normal(size=[2] * 4) + 1j * rng.normal(size=[2] * 4))

    def f_engine(x, y):
        return (tn.permute(tn.tensordot(x, y, [1, 3], [2, 0]), (-1, 0, 3, 1, 2)).abs() ** 2).sum()

    def f_torch(x, y):
        return (torch.tensordot(x, y, ([1, 3], [2, 0])).permute(4, 0, 3, 1, 2).abs() ** 2).sum()

    (ga, gb), v = torch.func.grad_and_value(f_engine, argnums=(0, 1))(a, b)
    (wa, wb), w = torch.func.grad_and_value(f_torch, argnums=(0, 1))(a, b)
    np.testing.assert_allclose(float(v), float(w), rtol=1e-12)
    np.testing.assert_allclose(ga.numpy(), wa.numpy(), atol=1e-10)
    np.testing.assert_allclose(gb.numpy(), wb.numpy(), atol=1e-10)
    ab = torch.stack([a, 2 * a, a.conj()])
    got = torch.func.vmap(lambda x: f_engine(x, b))(ab)
    want = torch.stack([f_torch(x, b) for x in ab])
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-12)
    gv = torch.func.vmap(torch.func.grad(f_engine), in_dims=(0, None))(ab, b)
    wv = torch.stack([torch.func.grad(f_torch)(x, b) for x in ab])
    np.testing.assert_allclose(gv.numpy(), wv.numpy(), atol=1e-10)
    with pytest.raises(ValueError):
        tn.permute(a, (0, 0, 1, 2, 3))


def test_cotengra_options_are_honoured_or_refused():
    """reference experimental.py:934-946 passes ``cotengra_options`` to ``ctg.ReusableHyperOptimizer``; the
    north-star example sets ``slicing_opts={"target_slices": num_device}``, ``minimize="combo"``
    (examples/slicing_auto_pmap_vqa.py:86-94).  Here: understood keys act, everything else raises."""
    from tcmi import tn
    from tcmi.experimental import DistributedContractor as DC

    o = DC._parse_options({"slicing_opts": {"target_slices": 8}, "max_repeats": 16, "progbar": True,
                           "minimize": "combo", "parallel": True})
    assert o["target_slices"] == 8 and o["target_size"] is None and o["minimize"] == "combo" and o["max_repeats"] == 16
    assert o["seeds"] == [0] and DC._parse_options({"seed": [3, 5]})["seeds"] == [3, 5]
    assert DC._parse_options(None)["target_size"] == 2**28
    for bad in ({"optlib": "optuna"}, {"slicing_opts": {"target_overhead": 2}}, {"minimize": "foo"},
                {"methods": ["kahypar"]}, {"max_time": 3}):
        with pytest.raises(NotImplementedError):
            DC._parse_options(bad)
    inputs, output, sd = _rand_net(30, 3, 4, dangling=0)
    base = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
    for minimize in (None, "flops", "write", "combo", "size"):
        t = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
        t.minimize = minimize
        t.slice_to_slices(8)
        assert t.nslices >= 8 and t.nslices <= 16, (minimize, t.nslices)
        assert t.max_size() <= base.max_size()
    # both limits at once: the size cap first, then at least that many slices
    t = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
    t.slice_to(max(4, base.max_size() // 4))
    cap = t.max_size()
    t.slice_to_slices(16)
    assert t.nslices >= 16 and t.max_size() <= cap


def test_invariant_subtrees_are_split_over_the_ranks_once_each():
    """``ContractionTree.invariant_shards``: every slice-invariant step belongs to exactly one rank, whole subtrees stay
    together, every invariant tensor a slice-dependent step consumes is some rank's root, and the split is the same
    whoever computes it (the ranks of a slice shard derive it independently)."""
    from tcmi import tn

    inputs, output, sd = _rand_net(60, 3, 21, dangling=0)
    tree = tn.ContractionTree.from_path(inputs, output, sd, trials=4, seed=0)
    tree.slice_to(max(4, tree.max_size() // 16))
    steps, dep, last, _ = tree._symbolic_steps()
    n = len(tree.inputs)
    inv = {st[4] for st in steps if not dep[st[4]]}
    assert inv, "this network should have slice-invariant steps"
    prod = {st[4]: st for st in steps}
    for world in (2, 4, 8):
        steps_of, roots_of, loads = tree.invariant_shards(world)
        again = tree.invariant_shards(world)
        assert [sorted(x) for x in steps_of] == [sorted(x) for x in again[0]] and roots_of == again[1]
        allsteps = [t for sset in steps_of for t in sset]
        assert len(allsteps) == len(set(allsteps))
        needed = set()
        for ia, ib, xa, xb, io in steps:
            if dep[io]:
                needed |= {t for t in (ia, ib) if not dep[t] and t >= n}
        assert {r for rs in roots_of for r, _ in rs} == needed
        for k in range(world):
            for t in steps_of[k]:                      # a step's invariant operands were produced by the same rank
                for src in prod[t][:2]:
                    assert src < n or src in steps_of[k]
            for r, lg in roots_of[k]:
                assert r in steps_of[k]
        assert max(loads) <= sum(loads)


def test_native_greedy_path_equals_the_python_loop(monkeypatch):
    """``tcmi_greedy_path`` (host code of libtcmi) against the Python random-greedy loop on circuit networks: the same
    pairs in the same order for the deterministic search and for random-greedy draws (temperature / alpha / seed), with
    output indices and a disconnected component."""
    from tcmi import tn

    rng0 = np.random.default_rng(0)
    for case in range(6):
        nt = int(rng0.integers(20, 90))
        inputs, output, sd = _rand_net(nt, 3, 100 + case, dangling=case % 3)
        if case == 4:                       # a second, disconnected component
            base = max(sd) + 1
            inputs += [[base, base + 1], [base + 1, base + 2], [base + 2, base]]
            sd.update({base + k: 2 for k in range(3)})
        for temp, alpha in ((0.0, 1.0), (0.2, 1.0), (0.01, 0.5), (1.0, 1.5)):
            monkeypatch.setitem(KN.VALUES, "tn_native_greedy", "1")
            a = tn.greedy_path(inputs, output, sd, temperature=temp, alpha=alpha, rng=np.random.default_rng(case))
            monkeypatch.setitem(KN.VALUES, "tn_native_greedy", "0")
            b = tn.greedy_path(inputs, output, sd, temperature=temp, alpha=alpha, rng=np.random.default_rng(case))
            assert a == b, (case, temp, alpha)
            assert len(a) == len(inputs) - 1


def test_searched_trees_are_cached_by_network_and_options(tmp_path, monkeypatch):
    """DistributedContractor keeps searched trees (cotengra's ReusableHyperOptimizer in the reference's constructor,
    experimental.py:934-953): the second contractor of the same network and options loads the tree instead of searching;
    other options, another network or TCMI_TREE_CACHE=0 search again; a truncated file is ignored."""
    from tcmi import specialize as S, tn
    from tcmi import experimental as E

    monkeypatch.setattr(S, "CACHE_DIR", str(tmp_path / "plancache"))
    monkeypatch.setattr(S, "_user_cache_dir", lambda: None)
    monkeypatch.delenv("TCMI_TREE_CACHE", raising=False)
    nodes, _ = _closed_network(6, 3, 5, bits=[0] * 6)
    nodes2, _ = _closed_network(6, 3, 6, bits=[0] * 6)          # the same structure (other gate values): the same key
    nodes3, _ = _closed_network(6, 4, 5, bits=[0] * 6)          # another structure
    opts = {"seed": [0, 1], "max_repeats": 4, "slicing_opts": {"target_size": 8}}
    DC = E.DistributedContractor
    d1 = DC._get_tree_data(lambda _: nodes, None, opts)
    assert not DC.last_search[0].get("cached") and len(DC.last_search) == 2
    files = list((tmp_path / "plancache" / "trees").glob("*.pkl"))
    assert len(files) == 1
    d2 = DC._get_tree_data(lambda _: nodes2, None, opts)
    assert DC.last_search[0].get("cached") and [s_["seed"] for s_ in DC.last_search] == [0, 1] and d2 == d1
    DC._get_tree_data(lambda _: nodes, None, dict(opts, max_repeats=5))
    assert not DC.last_search[0].get("cached")
    DC._get_tree_data(lambda _: nodes3, None, opts)
    assert not DC.last_search[0].get("cached")
    assert len(list((tmp_path / "plancache" / "trees").glob("*.pkl"))) == 3
    monkeypatch.setenv("TCMI_TREE_CACHE", "0")
    DC._get_tree_data(lambda _: nodes, None, opts)
    assert not DC.last_search[0].get("cached")
    # cotengra's ``parallel``: the seeds on a process pool (no GPU context in this process) -- the same tree, and the same
    # cache key as the serial search
    d_par = DC._get_tree_data(lambda _: nodes, None, dict(opts, parallel=2))
    assert d_par == d1 and [s_["seed"] for s_ in DC.last_search] == [0, 1]
    assert E._tree_cache_key([[0]], [], {0: 2}, DC._parse_options(opts)) is None      # switched off: no key at all
    monkeypatch.delenv("TCMI_TREE_CACHE")
    assert E._tree_cache_key([[0]], [], {0: 2}, DC._parse_options(opts)) == \
        E._tree_cache_key([[0]], [], {0: 2}, DC._parse_options(dict(opts, parallel=True)))
    files[0].write_bytes(b"\x80")                                  # a truncated pickle
    d3 = DC._get_tree_data(lambda _: nodes, None, opts)
    assert not DC.last_search[0].get("cached") and d3 == d1        # searched again (and stored again)
    assert tn.ContractionTree.from_data(d3).nslices >= 1
