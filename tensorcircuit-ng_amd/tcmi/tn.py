"""Tensor-network layer of the hip backend: nodes, path search, slicing and the pairwise contraction
executor that drives the HIP tensordot engine (``tcmi_permute_bits`` + ``tcmi_cgemm``).

Counterpart of what the reference gets from third parties: ``tensornetwork`` nodes and
``contract_between`` (``tensorcircuit/cons.py:937-950``), ``opt_einsum.paths.greedy``
(``cons.py:1246-1258``) and cotengra's slicing (``tensorcircuit/experimental.py:863-872,1007-1008``).
Host side is symbolic (integers only) and testable without a GPU; tensors are torch-ROCm tensors
of shape ``[2] * rank`` (every circuit network has dimension-2 edges).
"""

import ctypes
import os
import heapq
import math
import itertools
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import _streams
from ._knobs import knob
from . import cons

_edge_ids = itertools.count()


def new_edge() -> int:
    return next(_edge_ids)


class Node:
    """A tensor with integer edge labels (a label shared by two nodes is a contracted edge, a label
    appearing once is dangling).  ``node.tensor`` mirrors ``tn.Node.tensor``."""

    __slots__ = ("tensor", "edges", "name", "is_dagger", "id", "is_unitary")

    def __init__(self, tensor, edges: Sequence[int], name: str = "", is_dagger: Optional[bool] = None,
                 id: Optional[int] = None, is_unitary: bool = False):
        self.tensor = tensor
        self.edges = list(edges)
        self.name = name
        # causal light-cone metadata (reference basecircuit.py:107-147): which side of <psi|O|psi> the node is on
        # and the identity of the gate it came from; None = untagged
        self.is_dagger = is_dagger
        self.id = id
        self.is_unitary = is_unitary

    def __repr__(self):
        return f"Node(name={self.name!r}, edges={self.edges})"


def node_is_unitary(nd: Node) -> bool:
    """``is_unitary`` of a node; a constant gate carries its matrix there until somebody asks (the check costs more
    than recording the gate, and only the light-cone cancellation needs it)."""
    u = nd.is_unitary
    if isinstance(u, (bool, np.bool_)):
        return bool(u)
    m = np.asarray(u)
    d = int(round(np.sqrt(m.size)))
    m = m.reshape(d, d)
    nd.is_unitary = bool(np.abs(m @ m.conj().T - np.eye(d)).max() < 1e-9)
    return nd.is_unitary


class CopyNode(Node):
    """``tn.CopyNode(rank, dimension)``: the delta tensor that identifies all its legs (a hyperedge).  It carries no
    array; the contractor merges the labels of its legs into one index that may then occur in more than two tensors
    and in the output (reference cons.py:499-541)."""

    __slots__ = ("dimension",)

    def __init__(self, rank: int, dimension: int = 2, name: str = "", edges: Optional[Sequence[int]] = None):
        edges = [new_edge() for _ in range(rank)] if edges is None else list(edges)
        assert len(edges) == rank
        super().__init__(None, edges, name)
        self.dimension = int(dimension)


def hyper_info(nodes: Sequence[Node]):
    """(arrays, input_sets, output_set, size_dict) of a network with CopyNodes: every leg label is replaced by the
    representative of its hyperedge (union-find over the legs of each CopyNode), dangling labels -- appearing once
    over all nodes, CopyNodes included -- become the output in order of appearance (reference
    ``_extract_topology``, cons.py:499-541)."""
    parent: Dict[int, int] = {}

    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    count: Dict[int, int] = {}
    order: List[int] = []
    for nd in nodes:
        for e in nd.edges:
            find(e)
            if e not in count:
                order.append(e)
            count[e] = count.get(e, 0) + 1
    for nd in nodes:
        if isinstance(nd, CopyNode):
            for e in nd.edges[1:]:
                ra, rb = find(nd.edges[0]), find(e)
                if ra != rb:
                    parent[rb] = ra
    regular = [nd for nd in nodes if not isinstance(nd, CopyNode)]
    inputs = [[find(e) for e in nd.edges] for nd in regular]
    dangling = [e for e in order if count[e] == 1]
    output = [find(e) for e in dangling]
    size_dict: Dict[int, int] = {}
    for nd in regular:
        for e, d in zip(nd.edges, nd.tensor.shape):
            size_dict[find(e)] = int(d)
    for nd in nodes:
        if isinstance(nd, CopyNode):
            size_dict.setdefault(find(nd.edges[0]), nd.dimension)
    return [nd.tensor for nd in regular], inputs, output, size_dict, dangling


def contract_hyper(arrays, inputs, output, size_dict, path=None, strip_exponent: bool = False):
    """Pairwise contraction of an einsum network whose indices may occur in any number of tensors (the reference's
    ``_algebraic_base_contraction``, cons.py:706-766).  A pair's shared index is summed only when no other tensor
    and not the output needs it; otherwise it is a batch index of one batched ``tcmi_cgemm``.  ``strip_exponent``:
    every intermediate is rescaled to unit maximum and log10 of the factors is accumulated (cotengra's
    ``tree.contract(strip_exponent=True)``): returns (mantissa, exponent) with result = mantissa * 10**exponent."""
    import torch

    from . import linalg as LA

    cdt = getattr(torch, cons.dtypestr)
    tens = [cons.backend.convert_to_tensor(t).to(device=cons.backend.device, dtype=cdt) for t in arrays]
    edges = [list(s) for s in inputs]
    for s in edges:
        if len(set(s)) != len(s):
            raise NotImplementedError("Backend 'hip' has not implemented a hyperedge joining two legs of one tensor.")
    if path is None:
        path = greedy_path(edges, output, size_dict) if len(edges) > 1 else []
    uses: Dict[int, int] = {}
    for s in edges:
        for e in s:
            uses[e] = uses.get(e, 0) + 1
    outset = set(output)
    expo = torch.zeros((), dtype=torch.float64, device=cons.backend.device)

    def strip(t):
        nonlocal expo
        if not strip_exponent:
            return t
        m = t.abs().max().to(torch.float64)
        m = torch.where(m > 0, m, torch.ones_like(m))
        expo = expo + torch.log10(m)
        return t / m.to(t.real.dtype)

    tens = [strip(t) for t in tens]
    for a, b in path:
        ta, tb, ea, eb = tens[a], tens[b], edges[a], edges[b]
        shared = [e for e in ea if e in eb]
        summed = [e for e in shared if uses[e] == 2 and e not in outset]
        batch = [e for e in shared if e not in summed]
        fa = [e for e in ea if e not in shared]
        fb = [e for e in eb if e not in shared]
        dim = lambda es: int(np.prod([size_dict[e] for e in es])) if es else 1  # noqa: E731
        A = ta.permute(tuple(ea.index(e) for e in batch + fa + summed)).reshape(dim(batch), dim(fa), dim(summed))
        B = tb.permute(tuple(eb.index(e) for e in batch + summed + fb)).reshape(dim(batch), dim(summed), dim(fb))
        C = LA.matmul(A.contiguous(), B.contiguous())           # one batched tcmi_cgemm launch
        ne = batch + fa + fb
        t = strip(C.reshape([size_dict[e] for e in ne]))
        for e in summed:
            uses[e] -= 2
        for e in batch:
            uses[e] -= 1
        tens = [x for k, x in enumerate(tens) if k not in (a, b)] + [t]
        edges = [x for k, x in enumerate(edges) if k not in (a, b)] + [ne]
    res, re_ = tens[0], edges[0]
    extra = [e for e in re_ if e not in outset]
    if extra:   # an index nobody else needs: summed out (einsum semantics)
        res = res.sum(dim=[re_.index(e) for e in extra])
        re_ = [e for e in re_ if e in outset]
    if list(re_) != list(output):
        res = res.permute(tuple(re_.index(e) for e in output))
    return (res, expo) if strip_exponent else res


# ---- symbolic part: network info, greedy path, slicing ---------------------------------------------
def get_tn_info(nodes: Sequence[Node]):
    """``cons.get_tn_info`` (reference cons.py:804): (input_sets, output_set, size_dict) with the
    nodes in their list order (the deterministic order the reference gets from ``_stable_id_``)."""
    inputs = [list(n.edges) for n in nodes]
    count: Dict[int, int] = {}
    for s in inputs:
        for e in s:
            count[e] = count.get(e, 0) + 1
    output = [e for s in inputs for e in s if count[e] == 1]
    size_dict = {e: 2 for e in count}
    return inputs, output, size_dict


def greedy_path(inputs: Sequence[Sequence[int]], output: Sequence[int], size_dict: Dict[int, int],
                memory_limit=None, temperature: float = 0.0, alpha: float = 1.0, nbranch: int = 8,
                rng=None) -> List[Tuple[int, int]]:
    """Greedy pairwise path in opt_einsum's linear format (reference cons.py:937-950: each (a, b)
    indexes the current list, both are removed, the result is appended).  Candidate pairs share an
    index; score = size(out) - alpha (size(a) + size(b)); leftovers are outer-multiplied smallest
    first.  ``temperature > 0`` turns it into opt_einsum's random-greedy: the next pair is drawn from
    the ``nbranch`` best candidates with Boltzmann weights (relative temperature)."""
    out_set = frozenset(output)
    live: Dict[int, frozenset] = {i: frozenset(s) for i, s in enumerate(inputs)}
    uses: Dict[int, int] = {}
    owners: Dict[int, set] = {}
    for i, s in live.items():
        for e in s:
            uses[e] = uses.get(e, 0) + 1
            owners.setdefault(e, set()).add(i)
    # one uniform number per step, drawn up front: the Boltzmann choice is an inverse-CDF look-up, the same in this
    # loop and in the library's (tcmi_greedy_path)
    # (numpy's Generator.choice(n, p=...) is exactly this: one uniform number, a look-up in the normalised cumulative
    # sum -- so the trees are those of the rng.choice formulation this replaces)
    uniforms = rng.random(max(1, len(inputs) - 1)) if (temperature > 0 and rng is not None) else None
    circuit_net = all(d == 2 for d in size_dict.values()) and \
        all(v == 2 or (v == 1 and e in out_set) for e, v in uses.items())
    if circuit_net and memory_limit is None:
        ssa_n = _native_greedy(inputs, out_set, alpha, temperature, nbranch, uniforms)
        if ssa_n is not None:
            return _ssa_to_linear(ssa_n, len(inputs))

    def sz(s):
        r = 1
        for e in s:
            r *= size_dict[e]
        return r

    def merged(a, b):
        sa, sb = live[a], live[b]
        keep = []
        for e in sa | sb:
            rest = uses[e] - (e in sa) - (e in sb)
            if rest > 0 or e in out_set:
                keep.append(e)
        return frozenset(keep)

    # circuit networks: every index has dimension 2 and two ends (or one end and the output) -- sizes are powers of
    # two of set lengths and the kept indices are set algebra (same results, a third of the search time)
    if all(d == 2 for d in size_dict.values()):
        def sz(s):  # noqa: F811
            return 1 << len(s)
    if all(v == 2 or (v == 1 and e in out_set) for e, v in uses.items()):
        def merged(a, b):  # noqa: F811
            sa, sb = live[a], live[b]
            return (sa ^ sb) | ((sa & sb) & out_set)

    heap: List[Tuple[int, int, int]] = []

    def push(i):
        seen = set()
        for e in live[i]:
            for j in owners.get(e, ()):
                if j != i and j in live and j not in seen:
                    seen.add(j)
                    a, b = (i, j) if i < j else (j, i)
                    m = merged(a, b)
                    heapq.heappush(heap, (sz(m) - alpha * (sz(live[a]) + sz(live[b])), a, b))

    for i in list(live):
        push(i)
    nxt = len(live)
    ssa: List[Tuple[int, int]] = []
    def pop_valid():
        while heap:
            cost, a, b = heapq.heappop(heap)
            if a not in live or b not in live:
                continue
            m = merged(a, b)
            real = sz(m) - alpha * (sz(live[a]) + sz(live[b]))
            if real != cost:
                heapq.heappush(heap, (real, a, b))
                continue
            return cost, a, b, m
        return None

    while heap:
        first = pop_valid()
        if first is None:
            break
        if temperature > 0 and rng is not None:
            cands = [first]
            while len(cands) < nbranch:
                c = pop_valid()
                if c is None:
                    break
                if any(c[1] == o[1] and c[2] == o[2] for o in cands):
                    continue
                cands.append(c)
            c0 = cands[0][0]
            scale = temperature * max(1.0, abs(c0))
            w = [math.exp(-(c[0] - c0) / scale) for c in cands]
            tot = 0.0
            for x in w:
                tot += x
            cdf, acc = [], 0.0
            for x in w:
                acc += x / tot
                cdf.append(acc)
            u = float(uniforms[len(ssa)]) if len(ssa) < len(uniforms) else 0.5
            k = len(cands) - 1
            for i, x in enumerate(cdf):
                if u < x / cdf[-1]:
                    k = i
                    break
            for i, c in enumerate(cands):
                if i != k:
                    heapq.heappush(heap, (c[0], c[1], c[2]))
            first = cands[k]
        cost, a, b, m = first
        for x in (a, b):
            for e in live[x]:
                uses[e] -= 1
                owners[e].discard(x)
            del live[x]
        live[nxt] = m
        for e in m:
            uses[e] = uses.get(e, 0) + 1
            owners.setdefault(e, set()).add(nxt)
        ssa.append((a, b))
        push(nxt)
        nxt += 1
    rest = sorted(live, key=lambda i: sz(live[i]))
    while len(rest) > 1:
        a, b = rest[0], rest[1]
        live[nxt] = live[a] | live[b]
        ssa.append((a, b))
        rest = sorted([nxt] + rest[2:], key=lambda i: sz(live[i]))
        nxt += 1
    return _ssa_to_linear(ssa, len(inputs))


def _ssa_to_linear(ssa, n: int) -> List[Tuple[int, int]]:
    ids = list(range(n))
    path = []
    k = n
    for a, b in ssa:
        ia, ib = ids.index(a), ids.index(b)
        path.append((min(ia, ib), max(ia, ib)))
        for i in sorted((ia, ib), reverse=True):
            ids.pop(i)
        ids.append(k)
        k += 1
    return path


_NATIVE_GREEDY: List[Any] = []


def _native_greedy(inputs, out_set, alpha, temperature, nbranch, uniforms):
    """``tcmi_greedy_path`` of libtcmi (host code) on a circuit network, or None (library not built / switched off with
    TCMI_TN_NATIVE_GREEDY=0): the Python loop then runs, with the same choices."""
    if knob("tn_native_greedy", "1") == "0":
        return None
    if not _NATIVE_GREEDY:
        try:
            _NATIVE_GREEDY.append(_lib.lib().tcmi_greedy_path)
        except Exception:  # noqa: BLE001
            _NATIVE_GREEDY.append(None)
    fn = _NATIVE_GREEDY[0]
    if fn is None:
        return None
    labels: Dict[int, int] = {}
    for s in inputs:
        for e in s:
            labels.setdefault(e, len(labels))
    for e in out_set:
        labels.setdefault(e, len(labels))
    W = max(1, (len(labels) + 63) // 64)
    nt = len(inputs)
    masks = np.zeros((nt, W), dtype=np.uint64)
    for i, s in enumerate(inputs):
        for e in s:
            b = labels[e]
            masks[i, b >> 6] |= np.uint64(1) << np.uint64(b & 63)
    outm = np.zeros(W, dtype=np.uint64)
    for e in out_set:
        b = labels[e]
        outm[b >> 6] |= np.uint64(1) << np.uint64(b & 63)
    ssa = np.zeros(2 * max(1, nt), dtype=np.int32)
    un = np.ascontiguousarray(uniforms, dtype=np.float64) if uniforms is not None else None
    r = fn(nt, W, masks.ctypes.data, outm.ctypes.data, float(alpha), float(temperature if un is not None else 0.0),
           int(nbranch), un.ctypes.data if un is not None else None, len(un) if un is not None else 0, ssa.ctypes.data)
    if r < 0:
        return None
    return [(int(ssa[2 * k]), int(ssa[2 * k + 1])) for k in range(r)]


def _path_stats(inputs, output, size_dict, path):
    """(max intermediate size, total flops) of a linear-format path, without building a tree object."""
    cur = [frozenset(x) for x in inputs]
    uses: Dict[int, int] = {}
    for x in cur:
        for e in x:
            uses[e] = uses.get(e, 0) + 1
    out = frozenset(output)
    mx, flops = 1, 0
    for a, b in path:
        sb = cur.pop(b)
        sa = cur.pop(a)
        keep = frozenset(e for e in sa | sb if uses[e] - (e in sa) - (e in sb) > 0 or e in out)
        for e in sa:
            uses[e] -= 1
        for e in sb:
            uses[e] -= 1
        for e in keep:
            uses[e] += 1
        f = 1
        for e in sa | sb:
            f *= size_dict[e]
        k = 1
        for e in keep:
            k *= size_dict[e]
        flops += 8 * f
        mx = max(mx, k)
        cur.append(keep)
    return mx, flops


def search_path(inputs, output, size_dict, trials: int = 0, seed: int = 0, target_size: Optional[int] = None):
    """Best of the deterministic greedy path and ``trials`` random-greedy paths (opt_einsum's
    ``RandomGreedy``; the reference reaches it through cotengra's ``greedy`` method,
    cons.py:1168-1190).  Objective: (oversize w.r.t. ``target_size``, flops)."""
    def key(st):
        mx, fl = st
        return (max(mx, target_size) if target_size else 0, fl, mx)

    best = greedy_path(inputs, output, size_dict)
    best_key = key(_path_stats(inputs, output, size_dict, best))
    rng = np.random.default_rng(seed)
    for _ in range(trials):
        t = float(10 ** rng.uniform(-2.5, 0.0))
        al = float(rng.choice([0.0, 0.5, 1.0, 1.0, 1.5]))
        p = greedy_path(inputs, output, size_dict, temperature=t, alpha=al, rng=rng)
        k = key(_path_stats(inputs, output, size_dict, p))
        if k < best_key:
            best, best_key = p, k
    return best


RECONF_SUBTREE = 10   # intermediates per re-optimised subtree (3^k / 2 splits per dynamic programme)
RECONF_ALPHA = 0.0     # cost of a step = MACs + alpha * (elements read + written)
RECONF_COMBO_ALPHAS = (16.0, 32.0, 64.0, 128.0)   # bytes weights tried on the sliced tree (see _reconfigure_sliced)
RECONF_COMBO_ROUNDS = 4                           # rounds of those reconfigurations (beam search, see _reconfigure_sliced)
RECONF_BEAM = 2
# two-roof step model of ContractionTree.model_time (measured on MI355X, profiles/r02*: the MFMA GEMM route with its
# operand permutes sustains ~100 Tflop/s on 2^27-element steps, the scattered big x small kernel 2.6 - 5 TB/s)
MODEL_TFLOPS = 100.0
MODEL_GBS = 4000.0
MODEL_STEP_S = 8e-6


_NATIVE_DP: List[Any] = []


def _native_subtree_dp():
    """``tcmi_subtree_dp`` of libtcmi (host code), or None: the planner then runs its Python loop (same results)."""
    if knob("tn_native_dp", "1") == "0":
        return None
    if not _NATIVE_DP:
        try:
            _NATIVE_DP.append(_lib.lib().tcmi_subtree_dp)
        except Exception:  # noqa: BLE001  (library not built, or a jit probe is running)
            return None
    return _NATIVE_DP[0]


_NATIVE_RECONF: List[Any] = []
_NATIVE_SLICE: List[Any] = []


def _native_slice_fixed():
    """``tcmi_slice_fixed`` of libtcmi (host code), or None: ContractionTree._slice_fixed then runs its Python loop."""
    if knob("tn_native_slice", "1") == "0":
        return None
    if not _NATIVE_SLICE:
        try:
            _NATIVE_SLICE.append(_lib.lib().tcmi_slice_fixed)
        except Exception:  # noqa: BLE001
            return None
    return _NATIVE_SLICE[0]



def _native_reconfigure():
    """``tcmi_reconfigure_path`` of libtcmi (host code), or None: the Python loop below then runs (same trees)."""
    if knob("tn_native_reconf", "1") == "0":
        return None
    if not _NATIVE_RECONF:
        try:
            _NATIVE_RECONF.append(_lib.lib().tcmi_reconfigure_path)
        except Exception:  # noqa: BLE001
            return None
    return _NATIVE_RECONF[0]


def reconfigure_path(inputs, output, size_dict, path, subtree_size: int = 8, max_size: Optional[int] = None,
                     max_passes: int = 4, max_evals: int = 4000, alpha: Optional[float] = None):
    """Subtree reconfiguration of a contraction path (the refinement cotengra applies to its trees,
    reference cons.py:1168-1190 ``optimizer_reconf`` / experimental.py ``slicing_reconf_opts``): for every
    internal node of the tree, the subtree below it is cut off at ``subtree_size`` intermediates and those are
    re-contracted in the order that an exact dynamic programme over their subsets finds cheapest (flops);
    intermediates larger than ``max_size`` elements are not allowed.  Passes repeat until nothing improves
    (at most ``max_evals`` subtree optimisations: a deterministic budget, every rank of a distributed run must
    arrive at the same tree).
    Index sets are bit masks (python ints).  Networks with an index on more than two tensors are returned
    unchanged.  Returns a path in the same linear format."""
    import math

    n = len(inputs)
    if n < 3:
        return list(path)
    eid: Dict[int, int] = {}
    occ: Dict[int, int] = {}
    for s in inputs:
        for e in s:
            eid.setdefault(e, len(eid))
            occ[e] = occ.get(e, 0) + 1
    for e in output:
        eid.setdefault(e, len(eid))
        occ[e] = occ.get(e, 0) + 1
    if any(v > 2 for v in occ.values()):
        return list(path)
    lw = [0.0] * len(eid)
    for e, k in eid.items():
        lw[k] = math.log2(size_dict[e])
    uniform = all(abs(x - 1.0) < 1e-12 for x in lw)

    def lsize(mask: int) -> float:
        if uniform:
            return float(mask.bit_count())
        t = 0.0
        while mask:
            low = mask & -mask
            t += lw[low.bit_length() - 1]
            mask ^= low
        return t

    cap = float("inf") if max_size is None else math.log2(max_size) + 1e-9
    if alpha is None:
        alpha = RECONF_ALPHA
    # SSA tree: an index kept by a contraction is one that occurs once below it (each index has two ends)
    idx: Dict[int, int] = {}
    kids: Dict[int, Tuple[int, int]] = {}
    for i, s in enumerate(inputs):
        m = 0
        for e in s:
            m ^= 1 << eid[e]
        idx[i] = m
    ids = list(range(n))
    nxt = n
    for a, b in path:
        ib = ids.pop(max(a, b))
        ia = ids.pop(min(a, b))
        idx[nxt] = idx[ia] ^ idx[ib]
        kids[nxt] = (ia, ib)
        ids.append(nxt)
        nxt += 1
    root = ids[-1] if len(ids) == 1 else None
    if root is None:          # disconnected leftovers: leave such paths alone
        return list(path)

    native = _native_subtree_dp() if subtree_size <= 16 else None
    nwords = (len(eid) + 63) // 64
    lw_c = None
    if native is not None and not uniform:
        lw_c = (ctypes.c_double * (64 * nwords))(*(lw + [0.0] * (64 * nwords - len(lw))))
    if nwords > 64:
        native = None

    def step_cost(a: int, b: int) -> float:
        c = 2.0 ** lsize(idx[a] | idx[b])
        if alpha:
            c += alpha * (2.0 ** lsize(idx[a]) + 2.0 ** lsize(idx[b]) + 2.0 ** lsize(idx[a] ^ idx[b]))
        return c

    def optimise(x: int) -> bool:
        nonlocal nxt
        # frontier: expand the most expensive internal node until subtree_size intermediates
        front = [x]
        inner = []
        while len(front) < subtree_size:
            cand = [f for f in front if f in kids]
            if not cand:
                break
            f = max(cand, key=lambda v: step_cost(*kids[v]))
            front.remove(f)
            inner.append(f)
            front.extend(kids[f])
        k = len(front)
        if k < 3:
            return False
        old = sum(step_cost(*kids[v]) for v in inner)
        masks = [idx[f] for f in front]
        full = (1 << k) - 1
        if native is not None:
            # the same dynamic programme in libtcmi (tcmi_subtree_dp, bit-identical costs and tie-breaking)
            split_c = (ctypes.c_int * (full + 1))()
            bf = ctypes.c_double()
            buf = b"".join(m.to_bytes(8 * nwords, "little") for m in masks)
            if native(k, nwords, buf, lw_c, cap, float(alpha), split_c, ctypes.byref(bf)) != 0:
                raise RuntimeError("tcmi_subtree_dp failed")
            if not bf.value < old * (1.0 - 1e-9):
                return False
            for v in inner:
                del kids[v]
                if v != x:
                    del idx[v]

            def build_c(S: int, top: bool) -> int:
                nonlocal nxt
                if S & (S - 1) == 0:
                    return front[S.bit_length() - 1]
                A = split_c[S]
                l, r = build_c(A, False), build_c(S ^ A, False)
                if top:
                    v = x
                else:
                    v = nxt
                    nxt += 1
                    m, T = 0, S
                    while T:
                        low = T & -T
                        m ^= masks[low.bit_length() - 1]
                        T ^= low
                    idx[v] = m
                kids[v] = (l, r)
                return v

            build_c(full, True)
            return True
        sidx = [0] * (full + 1)
        ssz = [0.0] * (full + 1)
        best = [float("inf")] * (full + 1)
        split = [0] * (full + 1)
        for i in range(k):
            sidx[1 << i] = masks[i]
            ssz[1 << i] = 2.0 ** lsize(masks[i]) if alpha else 0.0
            best[1 << i] = 0.0
        order = sorted(range(1, full + 1), key=lambda v: v.bit_count())
        for S in order:
            if S & (S - 1) == 0:
                continue
            low = S & -S
            sidx[S] = sidx[low] ^ sidx[S ^ low]
            ls = lsize(sidx[S])
            if alpha:
                ssz[S] = 2.0 ** ls
            if S != full and ls > cap:
                continue                         # this intermediate would not fit
            bS, sS = float("inf"), 0
            A = (S - 1) & S
            while A:
                B = S ^ A
                if A > B:
                    ca, cb = best[A], best[B]
                    if ca < float("inf") and cb < float("inf") and (sidx[A] & sidx[B]):
                        c = ca + cb + 2.0 ** lsize(sidx[A] | sidx[B])
                        if alpha:
                            c += alpha * (ssz[A] + ssz[B] + ssz[S])
                        if c < bS:
                            bS, sS = c, A
                A = (A - 1) & S
            best[S], split[S] = bS, sS
        if not best[full] < old * (1.0 - 1e-9):
            return False
        # rebuild the subtree (the root keeps its id so that its parent stays valid)
        for v in inner:
            del kids[v]
            if v != x:
                del idx[v]

        def build(S: int, top: bool) -> int:
            nonlocal nxt
            if S & (S - 1) == 0:
                return front[S.bit_length() - 1]
            A = split[S]
            l, r = build(A, False), build(S ^ A, False)
            if top:
                v = x
            else:
                v = nxt
                nxt += 1
                idx[v] = sidx[S]
            kids[v] = (l, r)
            return v

        build(full, True)
        return True

    native_loop = _native_reconfigure() if (native is not None and len(kids) == n - 1) else None
    if native_loop is not None:
        # the whole loop in libtcmi (tcmi_reconfigure_path: same frontier, costs, tie-breaking and node numbering)
        pairs = (ctypes.c_int * (2 * (n - 1)))()
        for s_ in range(n - 1):
            pairs[2 * s_], pairs[2 * s_ + 1] = kids[n + s_]
        leaf = b"".join(idx[i].to_bytes(8 * nwords, "little") for i in range(n))
        cap_n = 2 * n + 16 * max_evals + 64
        nodes_c, kids_c, cnt_c = (ctypes.c_int * cap_n)(), (ctypes.c_int * (2 * cap_n))(), ctypes.c_int()
        rc = native_loop(n, nwords, leaf, pairs, lw_c, cap, float(alpha), int(subtree_size), int(max_passes), int(max_evals),
                         nodes_c, kids_c, cap_n, ctypes.byref(cnt_c))
        if rc != 0:
            raise RuntimeError("tcmi_reconfigure_path failed")
        kids = {int(nodes_c[i]): (int(kids_c[2 * i]), int(kids_c[2 * i + 1])) for i in range(cnt_c.value)}
        max_passes = 0
    evals = 0
    for _ in range(max_passes):
        changed = False
        todo = sorted(kids, key=lambda v: (-step_cost(*kids[v]), v))
        for v in todo:
            if v not in kids:
                continue
            if optimise(v):
                changed = True
            evals += 1
            if evals >= max_evals:
                break
        if not changed or evals >= max_evals:
            break
    # back to the linear format (post-order)
    out_path: List[Tuple[int, int]] = []
    pos = list(range(n))
    stack = [(root, False)]
    order: List[int] = []
    while stack:
        v, done = stack.pop()
        if v not in kids:
            continue
        if done:
            order.append(v)
        else:
            stack.append((v, True))
            stack.append((kids[v][1], False))
            stack.append((kids[v][0], False))
    # position of a tensor in the shrinking list = the number of live slots before its slot (slot i = input i, slot
    # n + s = the result of step s): a Fenwick tree instead of list.index / list.pop (quadratic: 26 ms per call at 500
    # tensors, and a path search makes hundreds of calls)
    del pos
    nslot = 2 * n
    fen = [0] * (nslot + 1)
    for i in range(1, nslot + 1):           # build with the first n slots live
        fen[i] += 1 if i <= n else 0
        j = i + (i & -i)
        if j <= nslot:
            fen[j] += fen[i]
    slot = {i: i for i in range(n)}

    def before(sl_: int) -> int:            # live slots with index < sl_
        t, i = 0, sl_
        while i > 0:
            t += fen[i]
            i -= i & -i
        return t

    def add(sl_: int, d_: int) -> None:
        i = sl_ + 1
        while i <= nslot:
            fen[i] += d_
            i += i & -i

    for s_, v in enumerate(order):
        l, r = kids[v]
        ia, ib = before(slot[l]), before(slot[r])
        out_path.append((min(ia, ib), max(ia, ib)))
        add(slot[l], -1)
        add(slot[r], -1)
        slot[v] = n + s_
        add(n + s_, 1)
    return out_path


@dataclass
class ContractionTree:
    """Path + slicing of one network (the part of cotengra's ``ContractionTree`` the reference uses:
    ``from_path``, ``remove_ind_``, ``sliced_inds``, ``nslices``, ``slice_arrays``, ``contract_core``,
    ``total_flops / total_write / max_size``; reference experimental.py:863-872,909-919,1007-1008)."""

    inputs: List[List[int]]
    output: List[int]
    size_dict: Dict[int, int]
    path: List[Tuple[int, int]]
    sliced_inds: List[int] = field(default_factory=list)

    @classmethod
    def from_path(cls, inputs, output, size_dict, path=None, trials: int = 0, seed: int = 0):
        inputs = [list(s) for s in inputs]
        if path is None:
            path = search_path(inputs, output, size_dict, trials=trials, seed=seed)
            if trials > 0:
                path = reconfigure_path(inputs, output, size_dict, path, subtree_size=RECONF_SUBTREE)
        t = cls(inputs, list(output), dict(size_dict), [tuple(p) for p in path])
        t.trials, t.seed = trials, seed
        return t

    # -- cost model ---------------------------------------------------------------------------------
    def _walk(self):
        """Yield (set_a, set_b, set_out) of every pairwise step with the sliced indices removed."""
        sl = set(self.sliced_inds)
        cur = [frozenset(e for e in s if e not in sl) for s in self.inputs]
        uses: Dict[int, int] = {}
        for s in cur:
            for e in s:
                uses[e] = uses.get(e, 0) + 1
        out = frozenset(e for e in self.output if e not in sl)
        for a, b in self.path:
            sa, sb = cur[a], cur[b]
            keep = frozenset(e for e in sa | sb if uses[e] - (e in sa) - (e in sb) > 0 or e in out)
            for e in sa:
                uses[e] -= 1
            for e in sb:
                uses[e] -= 1
            for e in keep:
                uses[e] += 1
            yield sa, sb, keep
            cur = [s for k, s in enumerate(cur) if k not in (a, b)] + [keep]

    def _size(self, s):
        r = 1
        for e in s:
            r *= self.size_dict[e]
        return r

    @property
    def nslices(self) -> int:
        return self._size(self.sliced_inds)

    def max_size(self) -> int:
        return max([self._size(k) for _, _, k in self._walk()] + [1])

    def contraction_width(self) -> float:
        return float(np.log2(self.max_size()))

    def total_write(self) -> int:
        return self.nslices * sum(self._size(k) for _, _, k in self._walk())

    def total_flops(self) -> int:
        """Real flops (8 per complex multiply-add), all slices."""
        return self.nslices * sum(8 * self._size(sa | sb) for sa, sb, _ in self._walk())

    def algorithmic_bytes(self, itemsize: int) -> int:
        """itemsize * sum(size(A) + size(B) + size(C)) over the executed steps."""
        return self.nslices * itemsize * sum(self._size(a) + self._size(b) + self._size(c) for a, b, c in self._walk())

    # -- slicing -------------------------------------------------------------------------------------
    def remove_ind_(self, ind: int) -> None:
        if ind not in self.sliced_inds:
            self.sliced_inds.append(ind)

    def _repath(self) -> None:
        """Re-run the greedy search on the network with the sliced indices removed (the sliced
        network is smaller, so the path should adapt to it: cotengra's "slicing + reconfiguration")."""
        sl = set(self.sliced_inds)
        inputs = [[e for e in s if e not in sl] for s in self.inputs]
        output = [e for e in self.output if e not in sl]
        trials = getattr(self, "trials", 0)
        self.path = search_path(inputs, output, self.size_dict, trials=max(0, trials // 8),
                                seed=getattr(self, "seed", 0) + len(sl), target_size=getattr(self, "_target", None))

    def _steps_full(self, path):
        """(union of operand indices, output indices) of every step of ``path`` on the UNSLICED network."""
        cur = [frozenset(x) for x in self.inputs]
        uses: Dict[int, int] = {}
        for x in cur:
            for e in x:
                uses[e] = uses.get(e, 0) + 1
        out = frozenset(self.output)
        steps = []
        for a, b in path:
            sb = cur.pop(b)
            sa = cur.pop(a)
            keep = frozenset(e for e in sa | sb if uses[e] - (e in sa) - (e in sb) > 0 or e in out)
            for e in sa:
                uses[e] -= 1
            for e in sb:
                uses[e] -= 1
            for e in keep:
                uses[e] += 1
            steps.append((sa | sb, keep))
            cur.append(keep)
        return steps

    def _slice_fixed(self, path, target_size: int, max_slices: int, max_candidates: int):
        """Greedy slicing of a FIXED tree: repeatedly remove the index (of the oversize intermediates)
        that leaves the smallest (total oversize, flops).  Returns (sliced indices, total flops over all
        slices) or None if ``max_slices`` is exceeded."""
        lw = {e: float(np.log2(d)) for e, d in self.size_dict.items()}
        steps = self._steps_full(path)
        out = set(self.output)
        ltarget = float(np.log2(target_size)) + 1e-9
        sliced: List[int] = []
        sl: set = set()
        if all(d == 2 for d in self.size_dict.values()):
            # every index has dimension 2 (circuit networks): index sets as bit masks, sizes as popcounts -- the set /
            # generator form below spent half of the whole path search here
            eid = {e: i for i, e in enumerate(self.size_dict)}
            um = [sum(1 << eid[e] for e in un) for un, _ in steps]
            km = [sum(1 << eid[e] for e in keep) for _, keep in steps]
            outm = sum(1 << eid[e] for e in out)
            ids = list(self.size_dict)
            it = int(ltarget)
            native = _native_slice_fixed() if (len(ids) <= 4096 and all(isinstance(e, int) and abs(e) < (1 << 62) for e in ids)) else None
            if native is not None:
                # the same loop in libtcmi (tcmi_slice_fixed: same scores, candidate order and choice)
                W_ = (len(ids) + 63) // 64
                ub = b"".join(m.to_bytes(8 * W_, "little") for m in um)
                kb = b"".join(m.to_bytes(8 * W_, "little") for m in km)
                ob = outm.to_bytes(8 * W_, "little")
                lab = (ctypes.c_longlong * (64 * W_))(*(list(ids) + [0] * (64 * W_ - len(ids))))
                cap_ = max(1, int(max_slices).bit_length())
                bits_c, ns_c, fl_c = (ctypes.c_int * cap_)(), ctypes.c_int(), ctypes.c_double()
                if native(len(um), W_, ub, kb, ob, lab, it, int(min(max_slices, 1 << 62)), int(max_candidates), bits_c, cap_,
                          ctypes.byref(ns_c), ctypes.byref(fl_c)) != 0:
                    raise RuntimeError("tcmi_slice_fixed failed")
                if ns_c.value < 0:
                    return None
                return [ids[bits_c[i]] for i in range(ns_c.value)], 8.0 * fl_c.value * (2 ** ns_c.value)
            slm = 0
            nsl = 1
            while True:
                live = ~slm
                mx = max((k & live).bit_count() for k in km)
                if mx <= it:
                    flops = sum(2.0 ** (u & live).bit_count() for u in um)
                    return sliced, 8.0 * flops * nsl
                score: Dict[int, float] = {}
                for k in km:
                    kk = k & live
                    lk = kk.bit_count()
                    if lk > it:
                        w = 2.0 ** lk
                        kk &= ~outm
                        while kk:
                            low = kk & -kk
                            b = low.bit_length() - 1
                            score[b] = score.get(b, 0.0) + w
                            kk ^= low
                if not score:
                    return None
                cands = sorted(score, key=lambda b: (-score[b], ids[b]))[:max_candidates]
                best = None
                for b in cands:
                    lv = live & ~(1 << b)
                    over = flops = 0.0
                    for u, k in zip(um, km):
                        lk = (k & lv).bit_count()
                        flops += 2.0 ** (u & lv).bit_count()
                        if lk > it:
                            over += 2.0 ** lk
                    key = (over, flops)
                    if best is None or key < best[0]:
                        best = (key, b)
                b = best[1]
                nsl *= 2
                if nsl > max_slices:
                    return None
                sliced.append(ids[b])
                slm |= 1 << b

        def stats(extra):
            over, flops, mx = 0.0, 0.0, 0.0
            for un, keep in steps:
                lk = sum(lw[e] for e in keep if e not in sl and e != extra)
                lf = sum(lw[e] for e in un if e not in sl and e != extra)
                flops += 2.0 ** lf
                if lk > ltarget:
                    over += 2.0 ** lk
                if lk > mx:
                    mx = lk
            return over, flops, mx

        nsl = 1
        while True:
            over, flops, mx = stats(None)
            if mx <= ltarget:
                return sliced, 8.0 * flops * nsl
            score: Dict[int, float] = {}
            for _, keep in steps:
                lk = sum(lw[e] for e in keep if e not in sl)
                if lk > ltarget:
                    for e in keep:
                        if e not in sl and e not in out:
                            score[e] = score.get(e, 0.0) + 2.0 ** lk
            if not score:
                return None
            cands = sorted(score, key=lambda e: (-score[e], e))[:max_candidates]
            best = None
            for e in cands:
                o, f, _ = stats(e)
                key = (o, f)
                if best is None or key < best[0]:
                    best = (key, e)
            e = best[1]
            nsl *= self.size_dict[e]
            if nsl > max_slices:
                return None
            sliced.append(e)
            sl.add(e)

    def slice_to(self, target_size: int, max_slices: int = 1 << 16, max_candidates: int = 12) -> "ContractionTree":
        """Slice until the largest intermediate fits ``target_size`` elements (the role of cotengra's
        ``slicing_opts`` in reference experimental.py:936-953).  Every candidate path — the current one
        plus ``self.trials`` random-greedy paths — is sliced as a fixed tree; the (path, sliced indices)
        with the smallest total flops over all slices wins.  If no trial fits within ``max_slices`` the
        slice-and-re-path search (slower) is used."""
        if self.max_size() <= target_size:
            return self
        trials = getattr(self, "trials", 0)
        rng = np.random.default_rng(getattr(self, "seed", 0) + 7919)
        best = None
        paths = [list(self.path)]
        for t in range(trials + 1):
            if t > 0:
                temp = float(10 ** rng.uniform(-2.5, 0.0))
                al = float(rng.choice([0.0, 0.5, 1.0, 1.0, 1.5]))
                path = greedy_path(self.inputs, self.output, self.size_dict, temperature=temp, alpha=al, rng=rng)
            else:
                path = paths[0]
            r = self._slice_fixed(path, target_size, max_slices, max_candidates)
            if r is None:
                continue
            if best is None or r[1] < best[0]:
                best = (r[1], path, r[0])
        if best is not None:
            self.path = [tuple(x) for x in best[1]]
            start_path = list(self.path)
            self.path = [tuple(x) for x in best[1]]
            self.sliced_inds = list(best[2])
            if trials > 0:
                # second candidate: slice the search result one index at a time, reconfiguring the sliced network (for
                # flops) after every removal -- cotengra's interleaved "slicing_reconf".  Both candidates get one round
                # of the reconfiguration beam; the better one (model time) gets the rest (the 32-qubit RQC: 25.2 ms for
                # the fixed-tree slicing, 19.3 ms for the interleaved one)
                self._reconfigure_sliced(target_size, rounds=1)
                a = (self.objective(), self.path, self.sliced_inds)
                if self._slice_interleaved(start_path, target_size, max_slices, max_candidates):
                    self._reconfigure_sliced(target_size, rounds=1)
                    if self.objective() >= a[0]:
                        self.path, self.sliced_inds = a[1], a[2]
                else:
                    self.path, self.sliced_inds = a[1], a[2]
                self._reconfigure_sliced(target_size, flops_pass=False)
            return self
        return self._slice_repath(target_size, max_slices, max_candidates)

    def _slice_interleaved(self, path, target_size: int, max_slices: int, max_candidates: int) -> bool:
        """Slice ``path`` one index at a time: pick the index of the oversize intermediates that leaves the smallest
        (oversize, flops) on the current tree, then reconfigure the sliced network for flops before the next pick.
        Sets ``self.path`` / ``self.sliced_inds``; False when ``max_slices`` is exceeded."""
        self.path = [tuple(x) for x in path]
        self.sliced_inds = []
        lt = float(np.log2(target_size)) + 1e-9
        lw = {e: float(np.log2(d)) for e, d in self.size_dict.items()}
        out = set(self.output)
        while self.max_size() > target_size:
            if self.nslices * 2 > max_slices:
                return False
            steps = self._steps_full(self.path)
            sl = set(self.sliced_inds)
            score: Dict[int, float] = {}
            for _, keep in steps:
                lk = sum(lw[e] for e in keep if e not in sl)
                if lk > lt:
                    for e in keep:
                        if e not in sl and e not in out:
                            score[e] = score.get(e, 0.0) + 2.0 ** lk
            if not score:
                return False
            best = None
            for e in sorted(score, key=lambda e: (-score[e], e))[:max_candidates]:
                over = flops = 0.0
                for un, keep in steps:
                    lk = sum(lw[x] for x in keep if x not in sl and x != e)
                    flops += 2.0 ** sum(lw[x] for x in un if x not in sl and x != e)
                    if lk > lt:
                        over += 2.0 ** lk
                if best is None or (over, flops, e) < best:
                    best = (over, flops, e)
            self.sliced_inds.append(best[2])
            sl.add(best[2])
            inputs = [[e for e in s if e not in sl] for s in self.inputs]
            output = [e for e in self.output if e not in sl]
            self.path = [tuple(x) for x in reconfigure_path(inputs, output, self.size_dict, self.path, subtree_size=8,
                                                           max_passes=2)]
        return True

    def model_time(self, itemsize: int = 8) -> float:
        """Seconds one rank needs for all slices under the two-roof model of a pairwise step on the MI355X engine:
        max(flops / MODEL_TFLOPS, (|A| + |B| + |C|) itemsize / MODEL_GBS) + MODEL_STEP_S, slice-invariant steps
        counted once (``contract_slices`` computes them once).  The RQC networks of config 4 sit at the ridge: the
        flops-optimal tree moves 130 GB for 2.3 Tflop, i.e. it is bound by the bytes of its big x small steps."""
        steps, dep, _, _ = self._symbolic_steps()
        t_inv = t_sl = 0.0
        for (sa, sb, keep), st in zip(self._walk(), steps):
            fl = 8.0 * self._size(sa | sb)
            by = float(itemsize) * (self._size(sa) + self._size(sb) + self._size(keep))
            t = max(fl / (MODEL_TFLOPS * 1e12), by / (MODEL_GBS * 1e9)) + MODEL_STEP_S
            if dep[st[4]]:
                t_sl += t
            else:
                t_inv += t
        return t_inv + self.nslices * t_sl

    def objective(self) -> Any:
        """What the tree search minimises (cotengra's ``minimize``, reference experimental.py:934-946): ``None`` = the
        engine's own two-roof ``model_time``; "flops", "write", "size" (largest intermediate, then flops), "combo" /
        "combo-<f>" = scalar operations + f x elements written (cotengra's default f = 64)."""
        m = getattr(self, "minimize", None)
        if m is None:
            return self.model_time()
        if m == "flops":
            return float(self.total_flops())
        if m == "write":
            return float(self.total_write())
        if m == "size":
            return (self.max_size(), float(self.total_flops()))
        if m == "max":
            return (self.max_size(), float(self.total_write()))
        if m.startswith("combo"):
            f = float(m.split("-", 1)[1]) if "-" in m else 64.0
            return self.total_flops() / 8.0 + f * self.total_write()
        raise NotImplementedError(f"Backend 'hip' has not implemented minimize={m!r} "
                                  f"(flops / write / size / max / combo / combo-<factor>)")

    def slice_to_slices(self, target_slices: int, max_candidates: int = 12) -> "ContractionTree":
        """cotengra's ``slicing_opts={"target_slices": S}`` (reference examples/slicing_auto_pmap_vqa.py:86-94): slice
        until there are at least S slices -- one per device -- whatever the size of the intermediates.  Greedy on the
        current tree: of the indices of the largest intermediates take the one that leaves the smallest objective."""
        out = set(self.output)
        while self.nslices < int(target_slices):
            score: Dict[int, int] = {}
            for _, _, k in self._walk():
                sz = self._size(k)
                for e in k:
                    if e not in out:
                        score[e] = score.get(e, 0) + sz
            for sa, sb, _ in self._walk():          # contracted indices never appear in an output set
                for e in sa & sb:
                    if e not in out:
                        score.setdefault(e, self._size(sa | sb))
            if not score:
                break
            base = list(self.sliced_inds)
            best = None
            for e in sorted(score, key=lambda e: (-score[e], e))[:max_candidates]:
                self.sliced_inds = base + [e]
                key = (self.objective(), e)
                if best is None or key < best:
                    best = key
            self.sliced_inds = base + [best[1]]
        if getattr(self, "trials", 0) > 0 and self.sliced_inds:
            self._reconfigure_sliced(self.max_size())
        return self

    def _reconfigure_sliced(self, target_size: int, rounds: Optional[int] = None, flops_pass: bool = True) -> None:
        """Subtree reconfiguration of the sliced network under the size cap (what cotengra's
        ``slicing_reconf_opts`` does after choosing the sliced indices): first for flops, then -- the engine is
        HBM-bound on big x small steps -- with cotengra's "combo" objective flops + alpha * (elements read +
        written) for a few alpha; the tree with the smallest ``model_time`` is kept."""
        sl = set(self.sliced_inds)
        inputs = [[e for e in s if e not in sl] for s in self.inputs]
        output = [e for e in self.output if e not in sl]
        if flops_pass:
            before = self.total_flops()
            old = self.path
            self.path = [tuple(x) for x in reconfigure_path(inputs, output, self.size_dict, old,
                                                           subtree_size=RECONF_SUBTREE, max_size=target_size)]
            if self.total_flops() > before or self.max_size() > target_size:
                self.path = old
        # beam search over reconfigurations (the landscape is rugged: the greedy "always continue from the best" ends
        # 45 % above what a width-2 beam finds for the 32-qubit RQC): round 1 with small subtrees from the flops tree,
        # later rounds with the full subtree size from the two best trees so far
        def tried(path):
            self.path = path
            if self.max_size() > target_size:
                return float("inf")
            o = self.objective()
            if not isinstance(o, tuple):
                return o
            # (primary, secondary), e.g. minimize="size": (largest intermediate, flops).  The primary term is an element
            # count (an integer), so a secondary term squeezed into [0, 1) keeps the lexicographic order in one float;
            # log2 keeps a 10 % difference of the secondary term far above the ulp of the sum (o[0] * 1e30 + o[1] lost it)
            return float(o[0]) + math.log2(1.0 + float(o[1])) / 1024.0

        start = self.path
        beam = [(tried(start), start)]
        best_t, best_path = beam[0]
        seen = {tuple(start)}
        for rnd in range(RECONF_COMBO_ROUNDS if rounds is None else rounds):
            sub = 8 if (rnd == 0 and flops_pass) else RECONF_SUBTREE
            found = []
            for _, base in beam:
                for alpha in RECONF_COMBO_ALPHAS:
                    p2 = [tuple(x) for x in reconfigure_path(inputs, output, self.size_dict, base, subtree_size=sub,
                                                             max_size=target_size, alpha=alpha)]
                    if tuple(p2) in seen:
                        continue
                    seen.add(tuple(p2))
                    found.append((tried(p2), p2))
            if not found:
                break
            found.sort(key=lambda c: c[0])
            beam = found[:RECONF_BEAM]
            if beam[0][0] < best_t * (1.0 - 1e-3):
                best_t, best_path = beam[0]
            elif rnd > 0:
                break
        self.path = best_path

    def _slice_repath(self, target_size: int, max_slices: int = 1 << 16, max_candidates: int = 12) -> "ContractionTree":
        """Each step tries the (non-output) indices of the largest intermediates, re-paths the sliced
        network for every candidate and keeps the one with the smallest (oversize, total flops)."""
        out = set(self.output)
        self._target = target_size
        while self.max_size() > target_size:
            if self.nslices * 2 > max_slices:
                raise RuntimeError(
                    f"slicing to target_size={target_size} needs more than {max_slices} slices"
                )
            score: Dict[int, int] = {}
            for _, _, k in self._walk():
                s = self._size(k)
                if s > target_size:
                    for e in k:
                        if e not in out:
                            score[e] = score.get(e, 0) + s
            if not score:
                break
            cands = sorted(score, key=lambda e: (-score[e], e))[:max_candidates]
            best = None
            base_sliced, base_path = list(self.sliced_inds), list(self.path)
            for e in cands:
                self.sliced_inds = base_sliced + [e]
                self._repath()
                sizes = [self._size(k) for _, _, k in self._walk()]
                over = sum(x for x in sizes if x > target_size)   # smooth potential: progress even
                key = (max(sizes + [1]) > target_size, over, self.total_flops())  # when max stays
                if best is None or key < best[0]:
                    best = (key, e, list(self.path))
            self.sliced_inds = base_sliced + [best[1]]
            self.path = best[2]
        return self

    def slice_index_values(self, i: int) -> Dict[int, int]:
        """Mixed-radix digits of slice ``i`` (first sliced index = most significant)."""
        vals = {}
        for e in reversed(self.sliced_inds):
            d = self.size_dict[e]
            vals[e] = i % d
            i //= d
        return vals

    def slice_arrays(self, arrays: Sequence[Any], i: int) -> List[Any]:
        """Fix every sliced index to its value in slice ``i`` (host-side view selection, K8)."""
        vals = self.slice_index_values(i)
        out = []
        for arr, edges in zip(arrays, self.inputs):
            idx = tuple(vals[e] if e in vals else slice(None) for e in edges)
            out.append(arr[idx] if any(e in vals for e in edges) else arr)
        return out

    def _symbolic_steps(self):
        """Static part of the contraction (cached): per step (id_a, id_b, axes_a, axes_b, id_out), the edge order of
        every SSA tensor and which tensors depend on a sliced index."""
        key = (tuple(self.path), tuple(self.sliced_inds))
        cached = getattr(self, "_steps_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        sl = set(self.sliced_inds)
        n = len(self.inputs)
        edges = {i: [e for e in s if e not in sl] for i, s in enumerate(self.inputs)}
        dep = {i: any(e in sl for e in s) for i, s in enumerate(self.inputs)}
        uses: Dict[int, int] = {}
        for s in edges.values():
            for e in s:
                uses[e] = uses.get(e, 0) + 1
        out = [e for e in self.output if e not in sl]
        outset = set(out)
        ids = list(range(n))
        steps = []
        nxt = n
        for a, b in self.path:
            ia, ib = ids[a], ids[b]
            ea, eb = edges[ia], edges[ib]
            shared = [e for e in ea if e in eb and uses[e] == 2 and e not in outset]
            steps.append((ia, ib, [ea.index(e) for e in shared], [eb.index(e) for e in shared], nxt))
            edges[nxt] = [e for e in ea if e not in shared] + [e for e in eb if e not in shared]
            dep[nxt] = dep[ia] or dep[ib]
            for e in shared:
                uses[e] -= 2
            ids = [x for k, x in enumerate(ids) if k not in (a, b)] + [nxt]
            nxt += 1
        last = ids[0]
        final_perm = None if list(edges[last]) == out else [edges[last].index(e) for e in out]
        # last use of every tensor, to drop slice-dependent intermediates early
        plan = (steps, dep, last, final_perm)
        self._steps_cache = (key, plan)
        return plan

    def contract_slices(self, arrays: Sequence[Any], slice_ids: Sequence[int], shard=None):
        """Yield ``contract_core(slice_arrays(arrays, i))`` for every i in ``slice_ids``.  Intermediates that do
        not depend on a sliced index (most of the small early steps of a circuit network: only a few leaves carry
        the sliced indices) are computed once and reused by every slice.

        ``shard = (rank, world, group)``: the slice-invariant subtrees are split over the ranks of a slice shard
        (``invariant_shards``) and their roots exchanged with ONE all-gather, instead of every rank recomputing all of
        them (232 of 271 steps of the 32-qubit RQC: 1.7 of the 4.0 ms a rank spends on one slice).  Collective: every
        rank of the group must make the call, also ranks that hold only padding (empty ``slice_ids``).  ``group`` =
        "emulate": no collective, the other ranks' roots are computed locally once (bench.py's one-rank estimate)."""
        steps, dep, last, final_perm = self._symbolic_steps()
        slice_ids = list(slice_ids)
        if shard is not None and shard[1] > 1 and _graph_ok(arrays, len(steps), slice_ids or [0]) \
                and any(dep[st[4]] for st in steps):
            yield from self._contract_slices_graph(arrays, slice_ids, shard)
            return
        if _graph_ok(arrays, len(steps), slice_ids):
            yield from self._contract_slices_graph(arrays, slice_ids)
            return
        shared_t: Dict[int, Any] = {}
        n = len(self.inputs)
        for i in slice_ids:
            sliced = self.slice_arrays(arrays, i)
            cur: Dict[int, Any] = {}
            for k in range(n):
                if dep[k]:
                    cur[k] = sliced[k]
                elif k not in shared_t:
                    shared_t[k] = sliced[k]
            for ia, ib, xa, xb, io in steps:
                if not dep[io]:
                    if io not in shared_t:
                        shared_t[io] = tensordot(shared_t[ia], shared_t[ib], xa, xb)
                    continue
                ta = cur.pop(ia) if dep[ia] else shared_t[ia]
                tb = cur.pop(ib) if dep[ib] else shared_t[ib]
                cur[io] = tensordot(ta, tb, xa, xb)
            res = cur[last] if dep[last] else shared_t[last]
            if final_perm is not None:
                res = permute(res, final_perm)
            yield res

    def contract_slices_vjp(self, arrays: Sequence[Any], slice_ids: Sequence[int], fop, need=None, alias_ok=False,
                            hat_ok=False, shard=None, fast_key=None):
        """``sum_i fop(contract_core(slice_arrays(arrays, i)))`` and its gradient with respect to every array, by a
        reverse sweep over the step list instead of a framework tape (reference ``experimental.py:1182-1211``:
        ``value_and_grad`` of ``contract_core`` per slice, summed).  Every forward AND backward step is one launch of
        the untaped engine (``tcmi_tensordot_bits`` / ``tcmi_contract_scattered`` / ``tcmi_cgemm``): the VJP of a
        tensordot is two tensordots (``tensordot_vjp``).  Intermediates live for one slice only; the slice-invariant
        subtrees are contracted once, their cotangents accumulate over the slices and are pulled back once at the end.

        ``fop`` maps a slice result to a real scalar (linear in the result, as in the reference);
        ``need[k]`` = whether array k wants a gradient (default: ``requires_grad``).  Returns (value, grads) with
        ``grads[k]`` shaped like ``arrays[k]`` (or None).  ``alias_ok``: the gradients of the slice-invariant leaves
        may be views of the replayed graphs' static memory (valid until the next call on this tree) instead of clones
        -- a caller that consumes them at once (``DistributedContractor.value_and_grad``) saves a thousand tiny copies.
        ``hat_ok``: the graph path may return the CONJUGATES of the gradients (it sweeps conj(g), see ``_vjp_halves``) and
        says so in ``self.last_vjp_conjugated``; the caller conjugates once, after stacking them.
        ``shard = (rank, world, group)`` (graph path): the slice-invariant subtrees are split over the ranks in BOTH
        directions -- forward as in ``contract_slices`` (own subtrees, one all-gather of the roots), backward by one
        all-reduce of the roots' accumulated cotangents, after which every rank pulls back through its own subtrees only
        (reference experimental.py:1028-1063: a rank holds only its slices' work).  The returned gradients are then
        PARTIAL: their sum over the ranks (which the caller's gradient all-reduce forms anyway) is the gradient."""
        import torch

        # ``fast_key``: the caller vouches that arrays with this key have the structure (count, shapes, dtypes, device,
        # which of them are on the tape) of the call the graphs were captured on -- a traced node function replaying the
        # same recipe (DistributedContractor).  The per-array checks below (a thousand arrays: 2-3 ms of host time per
        # call) are then skipped and the captured graphs are replayed at once.
        cache = getattr(self, "_vjp_graph_cache", None)
        if fast_key is not None:
            fast_key = (fast_key, tuple(slice_ids), alias_ok, hat_ok,
                        None if shard is None else (shard[0], shard[1], shard[2] == "emulate"))
        if fast_key is not None and cache is not None and cache.get("fast_key") == fast_key \
                and not torch.cuda.is_current_stream_capturing() and os.environ.get("TCMI_TN_GRAPH", "1") != "0":
            self.last_vjp_conjugated = False
            return self._vjp_graph_replay(cache, arrays, list(slice_ids), fop, cache["need"], alias_ok, hat_ok, cache["shard3"])
        steps, dep, last, final_perm = self._symbolic_steps()
        n = len(self.inputs)
        slice_ids = list(slice_ids)
        raw = [a.detach() for a in arrays]
        if need is None:
            need = [bool(a.requires_grad) for a in arrays]
        needs = {k: bool(need[k]) for k in range(n)}
        for ia, ib, xa, xb, io in steps:
            needs[io] = needs[ia] or needs[ib]
        self.last_vjp_conjugated = False
        # A sharded run issues collectives inside the graph path (all-gather of the invariant roots, all-reduce of their
        # cotangents): the choice graph / no graph must then be the SAME on every rank, also on ranks that hold only
        # padding (slice_table fills rows first: 4 or 9 slices on 8 ranks leave ranks without any) -- those enter the
        # graph path with zero slices and still run their share of the invariant forest and both collectives.
        if shard is not None and not dep[last]:
            shard = None          # nothing is sliced: there is no invariant forest to split (decided alike on every rank)
        sharded = shard is not None and shard[1] > 1 and shard[2] != "emulate"
        if needs[last] and (slice_ids or sharded) and _graph_ok(raw, len(steps), slice_ids or [0]):
            out = self._contract_slices_vjp_graph(raw, slice_ids, fop, need, needs, alias_ok, hat_ok, shard)
            if fast_key is not None:
                self._vjp_graph_cache["fast_key"] = fast_key
            return out
        total = None
        grads: List[Any] = [None] * n
        ginv: Dict[int, Any] = {}
        inv_perm = None
        if final_perm is not None:
            inv_perm = [0] * len(final_perm)
            for i, p in enumerate(final_perm):
                inv_perm[p] = i

        def acc(store, k, g):
            store[k] = g if store.get(k) is None else store[k] + g

        with torch.no_grad():
            shared_t: Dict[int, Any] = {k: raw[k].contiguous() for k in range(n) if not dep[k]}
            for ia, ib, xa, xb, io in steps:
                if not dep[io]:
                    shared_t[io] = _tensordot_raw(shared_t[ia], shared_t[ib], xa, xb)
            for i in slice_ids:
                vals = self.slice_index_values(i)
                idx = [tuple(vals[e] if e in vals else slice(None) for e in edges) for edges in self.inputs]
                cur: Dict[int, Any] = {k: raw[k][idx[k]].contiguous() for k in range(n) if dep[k]}
                for ia, ib, xa, xb, io in steps:
                    if dep[io]:
                        cur[io] = _tensordot_raw(cur[ia] if dep[ia] else shared_t[ia], cur[ib] if dep[ib] else shared_t[ib], xa, xb)
                res = cur[last] if dep[last] else shared_t[last]
                if final_perm is not None:
                    res = _permute_raw(res, final_perm)
                with torch.enable_grad():
                    r_ = res.detach().requires_grad_(True)
                    v = fop(r_)
                    (g,) = torch.autograd.grad(v, r_)
                total = v.detach() if total is None else total + v.detach()
                if not needs[last]:
                    continue
                g = g.contiguous()
                if inv_perm is not None:
                    g = _permute_raw(g, inv_perm)
                gbar: Dict[int, Any] = {}
                acc(gbar if dep[last] else ginv, last, g)
                for ia, ib, xa, xb, io in reversed(steps):
                    if not dep[io] or io not in gbar:
                        continue
                    gio = gbar.pop(io)
                    ta = cur[ia] if dep[ia] else shared_t[ia]
                    tb = cur[ib] if dep[ib] else shared_t[ib]
                    ga, gb = tensordot_vjp(ta, tb, xa, xb, gio, needs[ia], needs[ib])
                    del cur[io]
                    if ga is not None:
                        acc(gbar if dep[ia] else ginv, ia, ga)
                    if gb is not None:
                        acc(gbar if dep[ib] else ginv, ib, gb)
                for k in range(n):
                    if dep[k] and k in gbar:
                        if grads[k] is None:
                            grads[k] = torch.zeros_like(raw[k])
                        grads[k][idx[k]] += gbar[k]
                del cur, gbar
            # pull the accumulated cotangents of the slice-invariant intermediates back to their leaves
            for ia, ib, xa, xb, io in reversed(steps):
                if dep[io] or io not in ginv:
                    continue
                gio = ginv.pop(io)
                ga, gb = tensordot_vjp(shared_t[ia], shared_t[ib], xa, xb, gio, needs[ia], needs[ib])
                if ga is not None:
                    acc(ginv, ia, ga)
                if gb is not None:
                    acc(ginv, ib, gb)
            for k in range(n):
                if not dep[k] and k in ginv and need[k]:
                    grads[k] = ginv[k].reshape(raw[k].shape)
        return total, [g if need[k] else None for k, g in enumerate(grads)]

    def _contract_slices_vjp_graph(self, raw, slice_ids, fop, need, needs, alias_ok=False, hat_ok=False, shard=None):
        """``contract_slices_vjp`` replayed from four HIP graphs (captured once per tree and operand signature): the
        slice-invariant forward steps, one slice forward, one slice backward, the invariant backward.  Between the
        slice graphs only ``fop`` and its derivative run eagerly, on the small result.  A 30-qubit depth-8 ladder is
        ~1000 forward and ~2000 backward steps of a few microseconds: issued one by one from Python they are host
        bound (0.2 s per call, scripts/gpu_sliced_vqa_prof.py)."""
        import torch

        global COUNTERS
        steps, dep, last, final_perm = self._symbolic_steps()
        n = len(self.inputs)
        vals0 = self.slice_index_values(slice_ids[0] if slice_ids else 0)    # a padding-only rank captures on slice 0
        idx0 = [tuple(vals0[e] if e in vals0 else slice(None) for e in edges) for edges in self.inputs]
        srank, sworld, sgroup = shard if shard is not None else (0, 1, None)
        if sworld <= 1:
            srank, sworld, sgroup = 0, 1, None
        sig = (tuple(self.path), tuple(self.sliced_inds), tuple(bool(x) for x in need),
               tuple((tuple(t.shape), t.dtype, t.device) for t in raw), srank, sworld, sgroup == "emulate")
        cache = getattr(self, "_vjp_graph_cache", None)
        inv_perm = None
        if final_perm is not None:
            inv_perm = [0] * len(final_perm)
            for i, p in enumerate(final_perm):
                inv_perm[p] = i
        if cache is None or cache["sig"] != sig:
            keep_counters, COUNTERS = COUNTERS, None
            try:
                with torch.no_grad():
                    st_inv = {k: raw[k].contiguous().clone() for k in range(n) if not dep[k]}
                    st_dep = {k: raw[k][idx0[k]].contiguous().clone() for k in range(n) if dep[k]}

                    # Steps by LEVEL (1 + the deepest operand; for the slice-dependent steps invariant operands count
                    # as leaves): the steps of a level are independent of each other, so under capture the gate-sized
                    # ones of a level go out as ONE launch (SmallBatch) -- forward and, level by level downwards, both
                    # halves of their VJPs.  The eager warm-up keeps the step-by-step form (B is None).
                    lvl_inv: Dict[int, int] = {}
                    lvl_dep: Dict[int, int] = {}
                    inv_levels: Dict[int, list] = {}
                    dep_levels: Dict[int, list] = {}
                    for st in steps:
                        ia, ib, xa, xb, io = st
                        if dep[io]:
                            lv = 1 + max(lvl_dep.get(ia, 0), lvl_dep.get(ib, 0))
                            lvl_dep[io] = lv
                            dep_levels.setdefault(lv, []).append(st)
                        else:
                            lv = 1 + max(lvl_inv.get(ia, 0), lvl_inv.get(ib, 0))
                            lvl_inv[io] = lv
                            inv_levels.setdefault(lv, []).append(st)
                    inv_order = [inv_levels[k] for k in sorted(inv_levels)]
                    dep_order = [dep_levels[k] for k in sorted(dep_levels)]

                    def forward_level(level, src, dst, B):
                        later = []
                        for st in level:
                            ia, ib, xa, xb, io = st
                            ta, tb = src(ia), src(ib)
                            if B is not None and B.ok(ta, tb, len(xa)):
                                dst[io] = B.add(ta, tb, xa, xb, None, 0)
                            else:
                                later.append((io, ta, tb, xa, xb))
                        if B is not None:
                            B.flush()
                        for io, ta, tb, xa, xb in later:
                            dst[io] = _tensordot_raw(ta, tb, xa, xb)

                    def backward_level(level, src, gsrc, route, B):
                        """VJP halves of the steps of one level: ``gsrc`` pops the cotangent of a step's result (None:
                        not needed), ``route(t, g)`` receives the cotangent of operand t.  All cotangents of the graphs
                        are CONJUGATED (``_vjp_halves(hat=True)``): no step conjugates an operand."""
                        later, done = [], []
                        for ia, ib, xa, xb, io in level:
                            gio = gsrc(io)
                            if gio is None:
                                continue
                            for which, x, y, ax, ay, perm, fl in _vjp_halves(src(ia), src(ib), xa, xb, gio, needs[ia],
                                                                             needs[ib], hat=True):
                                t_ = ib if which else ia
                                if B is not None and B.ok(x, y, len(ax)):
                                    done.append((t_, B.add(x, y, ax, ay, perm, fl)))
                                else:
                                    later.append((t_, x, y, ax, ay, perm, fl))
                        if B is not None:
                            B.flush()
                        for t_, x, y, ax, ay, perm, fl in later:
                            done.append((t_, _vjp_half_run(x, y, ax, ay, perm, fl)))
                        for t_, g_ in done:
                            route(t_, g_)

                    # invariant subtrees of THIS rank (sharded run): the forward computes and the backward pulls back
                    # through these only; `mine` = None: everything
                    mine = roots_of = None
                    if sworld > 1:
                        steps_of, roots_of, _ = self.invariant_shards(sworld)
                        mine = steps_of[srank]
                    inv_mine = inv_order if mine is None else \
                        [lv for lv in ([st for st in level if st[4] in mine] for level in inv_order) if lv]

                    def fwd_inv(shared, B=None, only_mine=True):
                        for level in (inv_mine if only_mine else inv_order):
                            forward_level(level, lambda t: shared[t], shared, B)

                    def fwd_slice(shared, cur, B=None):
                        for level in dep_order:
                            forward_level(level, lambda t: cur[t] if dep[t] else shared[t], cur, B)
                        res = cur[last] if dep[last] else shared[last]
                        return _permute_raw(res, final_perm) if final_perm is not None else res

                    def bwd_slice(shared, cur, g_in, gacc, B=None):
                        g0 = _conj(_permute_raw(g_in, inv_perm) if inv_perm is not None else g_in)   # the sweep carries conj(g)
                        gbar = {}
                        if dep[last]:
                            gbar[last] = g0
                        else:
                            gacc[last].add_(g0.reshape(gacc[last].shape))

                        acc_to, acc_from = [], []

                        def route(t_, g_):
                            if dep[t_]:
                                gbar[t_] = g_
                            else:       # every invariant tensor is consumed by ONE step: one contribution per slice
                                acc_to.append(gacc[t_])
                                acc_from.append(g_.reshape(gacc[t_].shape))

                        for level in reversed(dep_order):
                            backward_level(level, lambda t: cur[t] if dep[t] else shared[t],
                                           lambda io: gbar.pop(io, None), route, B)
                        if acc_to:      # one multi-tensor add instead of ~50 launches per slice
                            torch._foreach_add_(acc_to, acc_from)
                        return {k: gbar[k] for k in range(n) if dep[k] and k in gbar}

                    def bwd_inv(shared, gacc, B=None):
                        gt = dict(gacc)

                        def route(t_, g_):
                            gt[t_] = g_

                        for level in reversed(inv_mine):
                            backward_level(level, lambda t: shared[t], lambda io: gt.pop(io, None), route, B)
                        return {k: gt[k] for k in range(n) if not dep[k] and k in gt and need[k]}

                    # invariant tensors that receive cotangents from the slice sweeps: static accumulators
                    targets = set()
                    for ia, ib, xa, xb, io in steps:
                        if dep[io]:
                            targets |= {t for t in (ia, ib) if not dep[t] and needs[t]}
                    if not dep[last]:
                        targets.add(last)
                    # warm-up (kernels, bit tables), then the captures; every intermediate stays referenced so that the
                    # backward graphs read the memory the forward graphs wrote
                    shared = dict(st_inv)
                    fwd_inv(shared, only_mine=False)       # warm-up on the whole invariant part (also the other ranks' roots)
                    cur = dict(st_dep)
                    res = fwd_slice(shared, cur)
                    g_in = torch.zeros_like(res)
                    gacc = {t: torch.zeros_like(shared[t]) for t in targets}
                    bwd_slice(shared, cur, g_in, gacc)
                    bwd_inv(shared, gacc)
                    torch.cuda.synchronize()
                    warm_roots = shared
                    shared = dict(st_inv)
                    B = None
                    if knob("tn_batch", "1") != "0" and all(t.dtype == torch.complex64 for t in raw):
                        # (descriptors: 3 per invariant step; 3 per slice-dependent step and per-slice graph instance)
                        B = SmallBatch(raw[0].device, 3 * len(steps) + 3 * max(2, TN_STREAMS_SMALL) * len(steps) + 16)
                    g_a = None
                    big = gbig = None
                    views: Dict[int, Any] = {}
                    if sworld > 1:
                        # every root of the invariant forest is a view of `big` [world, cap]: this rank's graph fills its
                        # row, one all-gather fills the others (as in _contract_slices_graph)
                        cap = max(1, max(sum(1 << lg for _, lg in rs) for rs in roots_of))
                        big = torch.zeros(sworld, cap, dtype=raw[0].dtype, device=raw[0].device)
                        for r_, rs in enumerate(roots_of):
                            off = 0
                            for root, lg in rs:
                                views[root] = big[r_, off: off + (1 << lg)].view([2] * lg)
                                if sgroup == "emulate":          # the other ranks' roots, computed here once
                                    views[root].copy_(warm_roots[root])
                                off += 1 << lg
                    del warm_roots
                    if inv_mine:
                        g_a = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_a):
                            fwd_inv(shared, B)
                            if sworld > 1:
                                for root, lg in roots_of[srank]:
                                    views[root].copy_(shared[root])
                    if sworld > 1:
                        shared.update(views)
                        if sgroup != "emulate":
                            self._gather_invariants(big, srank, sworld, sgroup)   # the captures below replay on real values
                    pool = g_a.pool() if g_a is not None else None
                    if sworld > 1:
                        # the cotangents of the roots accumulate in views of ONE buffer: one all-reduce sums them over the
                        # ranks' slices before the invariant backward (leaf targets are summed with the gradients)
                        inner = sorted(t for t in targets if t >= n)
                        gbig = torch.zeros(max(1, sum(shared[t].numel() for t in inner)), dtype=raw[0].dtype,
                                           device=raw[0].device)
                        gacc, off = {}, 0
                        for t in inner:
                            gacc[t] = gbig[off: off + shared[t].numel()].view(shared[t].shape)
                            off += shared[t].numel()
                        for t in targets:
                            if t < n:
                                gacc[t] = torch.zeros_like(shared[t])
                    else:
                        gacc = {t: torch.zeros_like(shared[t]) for t in targets}
                    cur = dict(st_dep)
                    g_b = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_b, **({"pool": pool} if pool is not None else {})):
                        res = fwd_slice(shared, cur, B)
                    pool = g_b.pool()
                    g_in = torch.zeros_like(res)
                    g_c = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_c, pool=pool):
                        gleaf = bwd_slice(shared, cur, g_in, gacc, B)
                    g_d = None
                    if inv_mine:
                        g_d = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_d, pool=pool):
                            ginv_leaf = bwd_inv(shared, gacc, B)
                    else:       # nothing to pull back through on this rank (an empty capture is an error on some ROCm versions)
                        ginv_leaf = {k: gacc[k] for k in range(n) if not dep[k] and k in gacc and need[k]}
                    # a second instance of the two per-slice graphs (own leaf copies, intermediates, cotangent
                    # accumulators and memory pool) for a second stream: a slice's sweep is a chain of launches most of
                    # which cannot fill the chip (41 tile-kernel steps of ~30 us per slice backward), two slices side by
                    # side overlap them -- as contract_slices does for the forward-only replay
                    # ... and, when the slices are small (largest intermediate <= 2^22 elements: none of their kernels fills
                    # the chip), up to TN_STREAMS_SMALL instances: four slices side by side
                    extras = []
                    ninst = 1
                    if TN_STREAMS >= 2 and len(slice_ids) >= 2:
                        ninst = 2
                        if self.max_size() <= (1 << 22):
                            ninst = max(2, min(TN_STREAMS_SMALL, len(slice_ids)))
                    for _ in range(ninst - 1):
                        st_dep2 = {k: v.clone() for k, v in st_dep.items()}
                        cur2 = dict(st_dep2)
                        gacc2 = {t: torch.zeros_like(v) for t, v in gacc.items()}
                        g_b2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_b2):
                            res2 = fwd_slice(shared, cur2, B)
                        g_in2 = torch.zeros_like(res2)
                        g_c2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_c2, pool=g_b2.pool()):
                            gleaf2 = bwd_slice(shared, cur2, g_in2, gacc2, B)
                        extras.append({"st_dep": st_dep2, "cur": cur2, "gacc": gacc2, "g_b": g_b2, "res": res2, "g_in": g_in2,
                                       "g_c": g_c2, "gleaf": gleaf2, "side": _streams.side_stream(raw[0].device, len(extras))})
                    two = extras[0] if extras else None
                    if B is not None:
                        B.finish()     # the descriptor table the captured launches read: uploaded before any replay
            finally:
                COUNTERS = keep_counters
            cache = {"sig": sig, "st_inv": st_inv, "st_dep": st_dep, "shared": shared, "cur": cur, "res": res,
                     "g_in": g_in, "gacc": gacc, "gleaf": gleaf, "ginv_leaf": ginv_leaf, "g_a": g_a, "g_b": g_b,
                     "g_c": g_c, "g_d": g_d, "batch": B, "two": two, "extras": extras, "big": big, "gbig": gbig}
            cache["need"] = list(need)
            cache["shard3"] = (srank, sworld, sgroup)
            self._vjp_graph_cache = cache
        return self._vjp_graph_replay(cache, raw, slice_ids, fop, need, alias_ok, hat_ok, (srank, sworld, sgroup))

    def _vjp_graph_replay(self, cache, raw, slice_ids, fop, need, alias_ok, hat_ok, shard3):
        """One evaluation on the captured graphs of ``_contract_slices_vjp_graph``: leaf values into the static buffers,
        invariant forward, [all-gather], per slice {sliced leaves in, forward, op and its derivative, backward}, [all-reduce
        of the roots' cotangents], invariant backward."""
        import torch

        srank, sworld, sgroup = shard3
        n = len(self.inputs)
        with torch.no_grad():
            inv_k = list(cache["st_inv"])
            if inv_k:
                torch._foreach_copy_([cache["st_inv"][k] for k in inv_k], [raw[k] for k in inv_k])
            if cache["gacc"]:
                torch._foreach_zero_(list(cache["gacc"].values()))
            if cache["g_a"] is not None:
                cache["g_a"].replay()
            if sworld > 1 and sgroup != "emulate":
                self._gather_invariants(cache["big"], srank, sworld, sgroup)
            grads: List[Any] = [None] * n
            total = None
            # The leaves that carry a sliced index: their values go into STATIC full-size copies once per call, and every
            # slice's blocks of them (and of their gradients) are views built once per slice id -- a slice then costs two
            # multi-tensor launches instead of an indexing operation and a tiny kernel per leaf and direction (~60 of them:
            # with four slices in flight the host, not the device, set the pace)
            dep_keys = list(cache["st_dep"])
            g_keys = [k for k in cache["gleaf"] if need[k]]
            hs = cache.get("host_static")
            if hs is None or hs["need"] != tuple(g_keys) or any(hs["full"][k].shape != raw[k].shape for k in dep_keys):
                hs = cache["host_static"] = {
                    "need": tuple(g_keys), "full": {k: torch.empty_like(raw[k]) for k in dep_keys},
                    "gfull": {k: torch.zeros_like(raw[k]) for k in g_keys}, "views": {}}
            if dep_keys:
                torch._foreach_copy_([hs["full"][k] for k in dep_keys], [raw[k] for k in dep_keys])
            if g_keys:
                torch._foreach_zero_([hs["gfull"][k] for k in g_keys])
            for k in g_keys:
                grads[k] = hs["gfull"][k]
            extras = cache.get("extras") or []
            for inst_ in extras:
                if inst_["gacc"]:
                    torch._foreach_zero_(list(inst_["gacc"].values()))

            def slice_views(i):
                v_ = hs["views"].get(i)
                if v_ is None:
                    vals = self.slice_index_values(i)
                    idx = {k: tuple(vals[e] if e in vals else slice(None) for e in self.inputs[k])
                           for k in set(dep_keys) | set(g_keys)}
                    v_ = hs["views"][i] = ([hs["full"][k][idx[k]] for k in dep_keys], [hs["gfull"][k][idx[k]] for k in g_keys])
                return v_

            def one_slice(i, inst):
                """Forward, op and its derivative, backward of slice i on the graphs of ``inst``; returns op's value.  The
                sliced leaves' gradient blocks are NOT added here: the slices of a group differ in the last sliced indices
                only, so a leaf that carries just the earlier ones receives the SAME block from several of them -- the
                additions of all instances are issued on the main stream, after the join."""
                vin, _ = slice_views(i)
                if dep_keys:
                    torch._foreach_copy_([inst["st_dep"][k] for k in dep_keys], vin)
                inst["g_b"].replay()
                with torch.enable_grad():
                    r_ = inst["res"].detach().clone().requires_grad_(True)
                    v = fop(r_)
                    (g,) = torch.autograd.grad(v, r_)
                inst["g_in"].copy_(g)
                inst["g_c"].replay()
                return v.detach(), i

            def add_leaf_grads(inst, i):
                if g_keys:
                    torch._foreach_add_(slice_views(i)[1], [inst["gleaf"][k] for k in g_keys])

            ids = list(slice_ids)
            cur_s = torch.cuda.current_stream(raw[0].device) if extras else None
            group = 1 + len(extras)
            j = 0
            while j < len(ids):
                chunk = ids[j: j + group]
                outs = []
                for inst_, sid in zip(extras, chunk[1:]):       # the partner slices go out first, on their own streams
                    inst_["side"].wait_stream(cur_s)
                    with torch.cuda.stream(inst_["side"]):
                        v2, idx2 = one_slice(sid, inst_)
                    outs.append((inst_, v2, idx2))
                v, idx1 = one_slice(chunk[0], cache)
                add_leaf_grads(cache, idx1)
                total = v if total is None else total + v
                for inst_, v2, idx2 in outs:
                    cur_s.wait_stream(inst_["side"])
                    add_leaf_grads(inst_, idx2)       # on the main stream: ordered after the main instance's additions
                    total = total + v2
                j += len(chunk)
            for inst_ in extras:
                if inst_["gacc"]:
                    keys = list(cache["gacc"])
                    torch._foreach_add_([cache["gacc"][t] for t in keys], [inst_["gacc"][t] for t in keys])
            if sworld > 1 and sgroup != "emulate" and cache["gbig"] is not None:
                self._allreduce_complex(cache["gbig"], sgroup)     # the roots' cotangents, summed over every rank's slices
            if cache["g_d"] is not None:
                cache["g_d"].replay()
            if alias_ok:      # views of the graphs' static memory: the same objects on every call
                al = cache.get("ginv_alias")
                if al is None:
                    al = cache["ginv_alias"] = {k: gl.reshape(raw[k].shape) for k, gl in cache["ginv_leaf"].items()}
                for k, gl in al.items():
                    grads[k] = gl
            else:
                for k, gl in cache["ginv_leaf"].items():
                    grads[k] = gl.reshape(raw[k].shape).clone()
            # the graphs deliver conj(g)
            if hat_ok:
                self.last_vjp_conjugated = True
                if not alias_ok:          # the sliced leaves' gradients live in static memory too: hand out copies
                    for k in g_keys:
                        grads[k] = grads[k].clone()
            else:
                grads = [g_.conj().resolve_conj() if g_ is not None else None for g_ in grads]
        return total, [g if need[k] else None for k, g in enumerate(grads)]

    def invariant_shards(self, world: int):
        """Split of the slice-invariant steps over ``world`` ranks: the invariant part of the tree is a forest (roots =
        invariant intermediates consumed by slice-dependent steps); whole subtrees go to the least loaded rank, largest
        first, under the step model of ``model_time`` (deterministic: every rank derives the same split).  Returns
        (steps_of[r]: set of produced tensor ids, roots_of[r]: [(root id, log2 size)], model seconds per rank)."""
        steps, dep, last, _ = self._symbolic_steps()
        n = len(self.inputs)
        prod = {st[4]: st for st in steps}
        sl = set(self.sliced_inds)
        rank_of = {i: len([e for e in s if e not in sl]) for i, s in enumerate(self.inputs)}
        for ia, ib, xa, xb, io in steps:
            rank_of[io] = rank_of[ia] + rank_of[ib] - 2 * len(xa)
        roots: List[int] = []
        for ia, ib, xa, xb, io in steps:
            if dep[io]:
                for t in (ia, ib):
                    if not dep[t] and t >= n and t not in roots:
                        roots.append(t)
        trees = []
        for r in roots:
            sub, stack, cost = [], [r], 0.0
            while stack:
                t = stack.pop()
                st = prod.get(t)
                if st is not None and not dep[t]:
                    sub.append(t)
                    stack += [st[0], st[1]]
                    fl = 8.0 * 2.0 ** (rank_of[st[0]] + rank_of[st[1]] - len(st[2]))
                    by = 8.0 * (2.0 ** rank_of[st[0]] + 2.0 ** rank_of[st[1]] + 2.0 ** rank_of[t])
                    cost += max(fl / (MODEL_TFLOPS * 1e12), by / (MODEL_GBS * 1e9)) + MODEL_STEP_S
            trees.append((cost, r, sub))
        trees.sort(key=lambda c: (-c[0], c[1]))
        loads = [0.0] * world
        steps_of = [set() for _ in range(world)]
        roots_of: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
        for cost, r, sub in trees:
            k = min(range(world), key=lambda j: (loads[j], j))
            loads[k] += cost
            steps_of[k].update(sub)
            roots_of[k].append((r, rank_of[r]))
        return steps_of, roots_of, loads

    def _run_steps(self, leaves: Dict[int, Any], shared_t: Dict[int, Any], invariant: bool, only=None, batch=None):
        """The slice-invariant (``invariant``; ``only``: just the steps producing these tensors) or the slice-dependent
        steps, eagerly, on the given leaf tensors.  ``batch`` (a SmallBatch, under graph capture only): the steps run
        level by level and the gate-sized ones of a level share one launch."""
        steps, dep, last, final_perm = self._symbolic_steps()
        if batch is not None:
            # Phase 1: the steps whose whole subtree (within this part) is gate-sized -- the bottom of a circuit network's
            # tree, most of its steps -- level by level, one launch per level.  Phase 2: everything else in the tree's
            # own order (which bounds the live intermediates), consumed operands released as before.
            want = (lambda io: not dep[io] and (only is None or io in only)) if invariant else (lambda io: dep[io])
            cur = shared_t if invariant else dict(leaves)
            src = (lambda t: shared_t[t]) if invariant else (lambda t: cur[t] if dep[t] else shared_t[t])
            produced = {st[4] for st in steps if want(st[4])}
            rank: Dict[int, int] = {}
            small: Dict[int, bool] = {}
            lvl: Dict[int, int] = {}
            levels: Dict[int, list] = {}
            ok_fn = _lib.lib().tcmi_tensordot_bits_small_ok
            for st in steps:
                ia, ib, xa, xb, io = st
                if not want(io):
                    continue
                ra = rank[ia] if ia in produced else src(ia).dim()
                rb = rank[ib] if ib in produced else src(ib).dim()
                rank[io] = ra + rb - 2 * len(xa)
                small[io] = bool(ok_fn(ra, rb, len(xa))) and all(t not in produced or small[t] for t in (ia, ib))
                if small[io]:
                    lv = 1 + max(lvl.get(ia, 0), lvl.get(ib, 0))
                    lvl[io] = lv
                    levels.setdefault(lv, []).append(st)
            for lv in sorted(levels):
                for ia, ib, xa, xb, io in levels[lv]:
                    cur[io] = batch.add(src(ia), src(ib), xa, xb, None, 0)
                batch.flush()
            for ia, ib, xa, xb, io in steps:
                if not want(io) or small[io]:
                    continue
                if invariant:
                    cur[io] = tensordot(cur[ia], cur[ib], xa, xb)
                else:
                    ta = cur.pop(ia) if dep[ia] else shared_t[ia]
                    tb = cur.pop(ib) if dep[ib] else shared_t[ib]
                    cur[io] = tensordot(ta, tb, xa, xb)
            if invariant:
                return None
            res = cur[last] if dep[last] else shared_t[last]
            return permute(res, final_perm) if final_perm is not None else res
        if invariant:
            for ia, ib, xa, xb, io in steps:
                if not dep[io] and (only is None or io in only):
                    shared_t[io] = tensordot(shared_t[ia], shared_t[ib], xa, xb)
            return None
        cur = dict(leaves)
        for ia, ib, xa, xb, io in steps:
            if not dep[io]:
                continue
            ta = cur.pop(ia) if dep[ia] else shared_t[ia]
            tb = cur.pop(ib) if dep[ib] else shared_t[ib]
            cur[io] = tensordot(ta, tb, xa, xb)
        res = cur[last] if dep[last] else shared_t[last]
        if final_perm is not None:
            res = permute(res, final_perm)
        return res

    def _contract_slices_graph(self, arrays: Sequence[Any], slice_ids: Sequence[int], shard=None):
        """``contract_slices`` as two HIP graphs, captured once per (tree, operand signature) and replayed: the
        slice-invariant steps (one replay per call) and the slice-dependent steps (one replay per slice), reading
        static copies of the leaf tensors.  A sliced RQC tree is hundreds of launches of a few microseconds each;
        issued one by one from Python the launch gaps are 20 % of the wall time (profiles/r02*_rqc_steps.txt)."""
        import torch

        global COUNTERS
        steps, dep, last, final_perm = self._symbolic_steps()
        n = len(self.inputs)
        first = self.slice_arrays(arrays, slice_ids[0] if slice_ids else 0)
        srank, sworld, sgroup = shard if shard is not None else (0, 1, None)
        sig = (tuple(self.path), tuple(self.sliced_inds),
               tuple((tuple(t.shape), t.dtype, t.device) for t in first), srank, sworld, sgroup == "emulate")
        cache = getattr(self, "_graph_cache", None)
        if cache is None or cache["sig"] != sig:
            # warm-up: one eager slice (plans, bit-permutation tables and kernels are created outside the capture)
            warm: Dict[int, Any] = {k: first[k] for k in range(n) if not dep[k]}
            keep_counters, COUNTERS = COUNTERS, None
            try:
                self._run_steps({}, warm, True)
                self._run_steps({k: first[k] for k in range(n) if dep[k]}, warm, False)
                torch.cuda.synchronize()
                static = [first[k].contiguous().clone() for k in range(n)]
                shared_t: Dict[int, Any] = {k: static[k] for k in range(n) if not dep[k]}
                COUNTERS = new_counters()
                g_inv = None
                big = mine_steps = None
                B = None
                if knob("tn_batch", "1") != "0" and all(t.dtype == torch.complex64 and t.is_cuda for t in first):
                    B = SmallBatch(first[0].device, 2 * len(steps) + 16)
                if sworld > 1:
                    # invariant subtrees split over the ranks: this rank's graph computes its own subtrees and packs
                    # their roots into row `srank` of `big`; after the all-gather every root is a view of `big`
                    steps_of, roots_of, _ = self.invariant_shards(sworld)
                    cap = max(1, max(sum(1 << lg for _, lg in rs) for rs in roots_of))
                    big = torch.zeros(sworld, cap, dtype=first[0].dtype, device=first[0].device)
                    mine_steps = steps_of[srank]
                    views = {}
                    for r_, rs in enumerate(roots_of):
                        off = 0
                        for root, lg in rs:
                            views[root] = big[r_, off: off + (1 << lg)].view([2] * lg)
                            if sgroup == "emulate":      # the other ranks' roots, computed here once
                                views[root].copy_(warm[root])
                            off += 1 << lg
                    if mine_steps:
                        g_inv = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g_inv):
                            self._run_steps({}, shared_t, True, only=mine_steps, batch=B)
                            for root, lg in roots_of[srank]:
                                views[root].copy_(shared_t[root])
                    shared_t.update(views)
                elif any(not dep[st[4]] for st in steps):     # (an empty capture is an error on some ROCm versions)
                    g_inv = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_inv):
                        self._run_steps({}, shared_t, True, batch=B)
                del warm
                if sworld > 1 and sgroup != "emulate":
                    self._gather_invariants(big, srank, sworld, sgroup)   # the captures below replay on real values
                cnt_inv, COUNTERS = COUNTERS, new_counters()
                g_sl, res = None, None
                if any(dep[st[4]] for st in steps):
                    g_sl = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_sl, **({"pool": g_inv.pool()} if g_inv is not None else {})):
                        res = self._run_steps({k: static[k] for k in range(n) if dep[k]}, shared_t, False, batch=B)
                cnt_sl = COUNTERS
                # a second instance of the per-slice graph (own leaf copies, own memory pool) for a second stream: the
                # slice-dependent part is a chain of launches, many of them far too small to fill the chip, so two slices
                # side by side overlap each other's small steps
                g_sl2, res2, static2 = None, None, None
                if g_sl is not None and TN_STREAMS >= 2 and len(slice_ids) >= 2:
                    COUNTERS = None
                    static2 = {k: static[k].clone() for k in range(n) if dep[k]}
                    g_sl2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_sl2):
                        res2 = self._run_steps(dict(static2), shared_t, False, batch=B)
                if B is not None:
                    B.finish()     # descriptor table of the batched launches: uploaded before any replay
            finally:
                COUNTERS = keep_counters
            cache = {"sig": sig, "big": big, "static": static, "shared": shared_t, "g_inv": g_inv, "g_sl": g_sl, "res": res,
                     "cnt_inv": cnt_inv, "cnt_sl": cnt_sl, "g_sl2": g_sl2, "res2": res2, "static2": static2, "batch": B,
                     "side": _streams.side_stream(first[0].device, 0) if g_sl2 is not None else None}
            self._graph_cache = cache
        static = cache["static"]
        inv_k = cache.get("inv_leaves")
        if inv_k is None:
            inv_k = [k for k in range(n) if not dep[k]]
            if sworld > 1:     # only the leaves this rank's subtrees read, and those the slice-dependent steps read directly
                mine = self.invariant_shards(sworld)[0][srank]
                used = set()
                for ia, ib, xa, xb, io in steps:
                    if io in mine or dep[io]:
                        used.update(t for t in (ia, ib) if t < n and not dep[t])
                inv_k = [k for k in inv_k if k in used]
            cache["inv_leaves"] = inv_k
        dep_k = [k for k in range(n) if dep[k]]
        if inv_k:
            torch._foreach_copy_([static[k] for k in inv_k], [first[k] for k in inv_k])
        if cache["g_inv"] is not None:
            cache["g_inv"].replay()
        if sworld > 1 and sgroup != "emulate":
            self._gather_invariants(cache["big"], srank, sworld, sgroup)
        if COUNTERS is not None:
            for key, v in cache["cnt_inv"].items():
                COUNTERS[key] += v
        ids = list(slice_ids)
        two = cache.get("g_sl2") is not None
        cur_s = torch.cuda.current_stream(first[0].device) if two else None
        j = 0
        while j < len(ids):
            i = ids[j]
            i2 = ids[j + 1] if two and j + 1 < len(ids) else None
            if i2 is not None:           # the partner slice goes out first, on the second stream
                side = cache["side"]
                side.wait_stream(cur_s)  # invariant tensors, the leaves, and the clone of the previous partner result
                with torch.cuda.stream(side):
                    sl2 = self.slice_arrays(arrays, i2)
                    for k in dep_k:
                        cache["static2"][k].copy_(sl2[k])
                    cache["g_sl2"].replay()
                    del sl2
            sliced = first if i == ids[0] else self.slice_arrays(arrays, i)
            for k in dep_k:
                static[k].copy_(sliced[k])
            if cache["g_sl"] is None:      # nothing depends on a sliced index: the invariant graph did all the work
                yield self._run_steps({}, cache["shared"], False).clone()
                j += 1
                continue
            cache["g_sl"].replay()
            n_done = 1 if i2 is None else 2
            if COUNTERS is not None:
                for key, v in cache["cnt_sl"].items():
                    COUNTERS[key] += v * n_done
            r1 = cache["res"].clone()
            r2 = None
            if i2 is not None:
                cur_s.wait_stream(cache["side"])
                r2 = cache["res2"].clone()
            yield r1
            if r2 is not None:
                yield r2
            j += n_done

    @staticmethod
    def _allreduce_complex(buf, group) -> None:
        """Sum of a complex buffer over the ranks (RCCL all-reduce; through the real view, which every backend takes)."""
        import torch
        import torch.distributed as dist

        dist.all_reduce(torch.view_as_real(buf), group=group)

    @staticmethod
    def _gather_invariants(big, rank: int, world: int, group) -> None:
        """Row r of ``big`` [world, cap] <- rank r's packed invariant roots: the one collective of the sharded
        invariant part (RCCL all-gather of at most a few MB; gloo in the one-device tests)."""
        import torch
        import torch.distributed as dist

        mine = big[rank].clone()
        if dist.get_backend(group) == "nccl":
            # through the real views: every backend version takes float32 / float64, not every one takes complex
            dist.all_gather_into_tensor(torch.view_as_real(big).reshape(-1), torch.view_as_real(mine).reshape(-1), group=group)
        else:
            re = torch.view_as_real(mine).contiguous()
            outs = [torch.empty_like(re) for _ in range(world)]
            dist.all_gather(outs, re, group=group)
            for r in range(world):
                torch.view_as_real(big[r]).copy_(outs[r])

    def contract_core(self, arrays: Sequence[Any]):
        """Pairwise contraction of (already sliced) arrays along the path; returns the result with
        axes in ``output`` order (minus sliced indices)."""
        sl = set(self.sliced_inds)
        tens = list(arrays)
        edges = [[e for e in s if e not in sl] for s in self.inputs]
        uses: Dict[int, int] = {}
        for s in edges:
            for e in s:
                uses[e] = uses.get(e, 0) + 1
        out = [e for e in self.output if e not in sl]
        outset = set(out)
        for a, b in self.path:
            ea, eb = edges[a], edges[b]
            shared = [e for e in ea if e in eb and uses[e] == 2 and e not in outset]
            t = tensordot(tens[a], tens[b], [ea.index(e) for e in shared], [eb.index(e) for e in shared])
            ne = [e for e in ea if e not in shared] + [e for e in eb if e not in shared]
            for e in shared:
                uses[e] -= 2
            tens = [x for k, x in enumerate(tens) if k not in (a, b)] + [t]
            edges = [x for k, x in enumerate(edges) if k not in (a, b)] + [ne]
        res, re_ = tens[0], edges[0]
        if list(re_) != out:
            res = permute(res, [re_.index(e) for e in out])
        return res

    def to_data(self) -> Dict[str, Any]:
        """``tree_data`` dictionary of the reference (experimental.py:957-991)."""
        return {"inputs": self.inputs, "output": self.output, "size_dict": self.size_dict,
                "path": self.path, "sliced_inds": self.sliced_inds}

    @classmethod
    def from_data(cls, d):
        t = cls.from_path(d["inputs"], d["output"], d["size_dict"], d["path"])
        for e in d.get("sliced_inds", []):
            t.remove_ind_(e)
        return t


# ---- device part: permute / tensordot through the C ABI --------------------------------------------
_SRCBIT_CACHE: Dict[Tuple, Any] = {}
GRAPH_MIN_STEPS = 32   # trees with fewer steps are not worth two graph captures


def _graph_ok(arrays, nsteps: int, slice_ids) -> bool:
    """HIP-graph replay of ``contract_slices`` (TCMI_TN_GRAPH=0 disables): plain complex device tensors, nothing on
    an autograd tape or inside a functorch transform, and enough steps for the launch gaps to matter."""
    import os
    import torch

    if os.environ.get("TCMI_TN_GRAPH", "1") == "0" or nsteps < GRAPH_MIN_STEPS or not slice_ids:
        return False
    for t in arrays:
        if not (torch.is_tensor(t) and t.is_cuda and t.is_complex()) or _on_tape(t) or any(d != 2 for d in t.shape):
            return False
    return not torch.cuda.is_current_stream_capturing()


def _code(t):
    import torch

    if t.dtype == torch.complex64:
        return _lib.TCMI_C64
    if t.dtype == torch.complex128:
        return _lib.TCMI_C128
    raise TypeError(f"tcmi tensordot engine needs complex64/complex128 tensors, got {t.dtype}")


def _permute_raw(t, perm):
    """out axis i = in axis perm[i], for a contiguous [2]*rank device tensor (HIP kernel)."""
    import torch

    rank = t.dim()
    perm = tuple(int(p) for p in perm)
    if perm == tuple(range(rank)):
        return t
    assert all(s == 2 for s in t.shape), "the hip tensordot engine handles dimension-2 axes"
    key = (perm, t.device.index)
    sb = _SRCBIT_CACHE.get(key)
    if sb is None:
        from .executor import _dev

        src = np.zeros(rank, dtype=np.int32)
        for i, p in enumerate(perm):
            src[rank - 1 - i] = rank - 1 - p
        sb = _dev(src, t.device)
        _SRCBIT_CACHE[key] = sb
    t = t.contiguous()
    out = torch.empty_like(t)
    if COUNTERS is not None:
        COUNTERS["permute_launches"] += 1
        COUNTERS["permute_bytes"] += 2.0 * t.numel() * t.element_size()
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _lib.check(_lib.lib().tcmi_permute_bits(t.data_ptr(), out.data_ptr(), rank, sb.data_ptr(), 1, 0, _code(t), stream),
               "tcmi_permute_bits")
    return out


def _gemm_raw(a2, b2, M, N, K):
    import torch

    a2, b2 = a2.contiguous(), b2.contiguous()
    c = torch.empty(M * N, dtype=a2.dtype, device=a2.device)
    if COUNTERS is not None:
        COUNTERS["gemm_launches"] += 1
        COUNTERS["gemm_flops"] += 8.0 * M * N * K
        COUNTERS["gemm_bytes"] += float(M * K + K * N + M * N) * a2.element_size()
    stream = torch.cuda.current_stream(a2.device).cuda_stream
    _lib.check(_lib.lib().tcmi_cgemm(a2.data_ptr(), b2.data_ptr(), c.data_ptr(), M, N, K, 1, 0, 0, 0, 0, _code(a2), stream),
               "tcmi_cgemm")
    return c


_FN = {}

# bench.py: when a dict (see ``new_counters``), the engine tallies launches, algorithmic flops / bytes of the GEMM
# and scattered-contraction steps, and the bytes moved by stand-alone permutes (traffic that is not in B_alg).
COUNTERS = None


def new_counters():
    return {k: 0.0 for k in ("permute_launches", "permute_bytes", "gemm_launches", "gemm_flops", "gemm_bytes",
                             "scattered_launches", "scattered_flops", "scattered_bytes")}


def _vjp_axes(rank_a, rank_b, axes_a, axes_b):
    """Index bookkeeping of the two VJPs of ``c = tensordot(a, b, [axes_a, axes_b])`` (c axes = free(a), free(b)):
    gA = transpose(tensordot(g, conj b, [g's b-part], [free(b)]), perm_a), gB likewise -- each again ONE tensordot of
    the engine plus a bit permutation into the operand's own axis order."""
    fa = [i for i in range(rank_a) if i not in axes_a]
    fb = [i for i in range(rank_b) if i not in axes_b]
    sb, sa = sorted(axes_b), sorted(axes_a)
    # gA_raw axes: fa in order, then b's contracted axes in b's own order
    perm_a = [fa.index(i) if i in fa else len(fa) + sb.index(axes_b[axes_a.index(i)]) for i in range(rank_a)]
    # gB_raw axes: a's contracted axes in a's own order, then fb in order
    perm_b = [len(sa) + fb.index(j) if j in fb else sa.index(axes_a[axes_b.index(j)]) for j in range(rank_b)]
    return fa, fb, perm_a, perm_b


def _conj(t):
    return t.conj().resolve_conj()


def _vjp_halves(a, b, axes_a, axes_b, g, need_a=True, need_b=True, hat=False):
    """The VJPs of ``tensordot(a, b)`` as jobs ``(which, x, y, axes_x, axes_y, out_perm, flags)``: which = 0 / 1 for gA /
    gB, the job = permute(tensordot(conj?(x), conj?(y)), out_perm) with flags 1 / 2 = conjugate x / y.
    ``hat``: ``g`` is the CONJUGATED cotangent and so are the results -- conj(gA) = conj(g) . b and conj(gB) = a . conj(g)
    are plain tensordots, so a sweep that carries conj(g) from the result down to the leaves conjugates nothing on the
    way (flags 0; the caller conjugates once at either end)."""
    fa, fb, perm_a, perm_b = _vjp_axes(a.dim(), b.dim(), axes_a, axes_b)
    jobs = []
    if need_a:
        jobs.append((0, g, b, list(range(len(fa), len(fa) + len(fb))), fb, perm_a, 0 if hat else 2))
    if need_b:
        jobs.append((1, a, g, fa, list(range(len(fa))), perm_b, 0 if hat else 1))
    return jobs


def _vjp_half_run(x, y, ax, ay, perm, flags):
    """One such job on its own: the fused launch (gate-sized operands: conjugation and transposition; tile kernel:
    transposition, for flags == 0), else conjugate + tensordot (scattered big x small kernel included) + permute."""
    r = _tensordot_fused(x, y, ax, ay, perm, flags)
    if r is None:
        r = _tensordot_raw(_conj(x) if flags & 1 else x, _conj(y) if flags & 2 else y, ax, ay)
        r = _permute_raw(r, perm) if list(perm) != list(range(r.dim())) else r
    return r


SMALL_DESC_WORDS = 64


class SmallBatch:
    """Gate-sized tensordots of one tree LEVEL in one launch (``tcmi_tensordot_small_batch``).  Used while the HIP
    graphs of a sliced contraction are captured: ``add`` allocates the result (graph-pool memory, fixed address), writes
    the job's descriptor into a host table and returns the result tensor; ``flush`` launches the jobs added since the
    last flush -- they must be independent of each other -- reading their descriptors from a static device table;
    ``finish`` (after the captures, before the first replay) uploads the table.  Every tensor a descriptor points to is
    kept alive here."""

    def __init__(self, device, capacity: int):
        import torch

        self.host = np.zeros((max(1, capacity), SMALL_DESC_WORDS), dtype=np.int32)
        self.dev = torch.zeros(max(1, capacity) * SMALL_DESC_WORDS, dtype=torch.int32, device=device)
        self.n = 0
        self.start = 0
        self.maxlog = 0
        self.keep: List[Any] = []
        self.launches = 0

    @staticmethod
    def ok(a, b, nk: int) -> bool:
        import torch

        if knob("tn_bits", "1") == "0":
            return False
        if a.dtype != torch.complex64 or b.dtype != torch.complex64 or not a.is_cuda or not b.is_cuda:
            return False
        if any(d != 2 for d in a.shape) or any(d != 2 for d in b.shape):
            return False
        return bool(_lib.lib().tcmi_tensordot_bits_small_ok(a.dim(), b.dim(), nk))

    def add(self, a, b, axes_a, axes_b, out_perm=None, flags: int = 0):
        import torch

        if self.n >= self.host.shape[0]:
            raise RuntimeError("SmallBatch: descriptor table full")
        a, b = a.contiguous(), b.contiguous()
        nk = len(axes_a)
        rc = a.dim() + b.dim() - 2 * nk
        out = torch.empty([2] * rc, dtype=a.dtype, device=a.device)
        xa = (ctypes.c_int * max(nk, 1))(*axes_a)
        xb = (ctypes.c_int * max(nk, 1))(*axes_b)
        op = (ctypes.c_int * max(rc, 1))(*out_perm) if (out_perm is not None and rc) else None
        _lib.check(_lib.lib().tcmi_tensordot_small_desc(
            a.data_ptr(), a.dim(), b.data_ptr(), b.dim(), ctypes.cast(xa, ctypes.c_void_p), ctypes.cast(xb, ctypes.c_void_p),
            nk, ctypes.cast(op, ctypes.c_void_p) if op is not None else None, int(flags), out.data_ptr(),
            self.host[self.n].ctypes.data), "tcmi_tensordot_small_desc")
        self.n += 1
        self.maxlog = max(self.maxlog, rc)
        self.keep += [a, b, out]
        if COUNTERS is not None:
            COUNTERS["gemm_flops"] += 8.0 * (1 << (a.dim() + b.dim() - nk))
            COUNTERS["gemm_bytes"] += float(a.numel() + b.numel() + out.numel()) * a.element_size()
        return out

    def flush(self):
        import torch

        while self.start < self.n:
            cnt = min(self.n - self.start, 65535)
            stream = torch.cuda.current_stream(self.dev.device).cuda_stream
            _lib.check(_lib.lib().tcmi_tensordot_small_batch(self.dev.data_ptr() + 4 * SMALL_DESC_WORDS * self.start, cnt,
                                                             self.maxlog, stream), "tcmi_tensordot_small_batch")
            self.start += cnt
            self.launches += 1
            if COUNTERS is not None:
                COUNTERS["gemm_launches"] += 1
        self.maxlog = 0

    def finish(self):
        import torch

        assert self.start == self.n, "SmallBatch: jobs added after the last flush"
        if self.n:
            self.dev[: self.n * SMALL_DESC_WORDS].copy_(torch.from_numpy(self.host[: self.n].reshape(-1)))
            torch.cuda.synchronize(self.dev.device)


def tensordot_vjp(a, b, axes_a, axes_b, g, need_a=True, need_b=True):
    """(gA, gB) of ``tensordot(a, b, [axes_a, axes_b])`` for the cotangent ``g`` (torch's convention for complex
    tensors: gA = g . b^H), computed by the same kernels as the forward step -- the rule the reference gets from
    JAX's transpose of ``dot_general`` (``experimental.py:1182-1211`` differentiates ``contract_core``)."""
    out = [None, None]
    for which, x, y, ax, ay, perm, fl in _vjp_halves(a, b, axes_a, axes_b, g, need_a, need_b):
        out[which] = _vjp_half_run(x, y, ax, ay, perm, fl)
    return out[0], out[1]


def _tensordot_fused(a, b, axes_a, axes_b, out_perm, flags):
    """``permute(tensordot(conj?(a), conj?(b)), out_perm)`` in ONE launch (``tcmi_tensordot_bits_ex``; flags 1 / 2:
    conjugate a / b) for the gate-sized steps the small-tensor kernel takes; None for every other shape."""
    import torch

    if knob("tn_fused_vjp", "1") == "0" or knob("tn_bits", "1") == "0":
        return None
    nk = len(axes_a)
    if a.dtype != torch.complex64 or b.dtype != torch.complex64 or not a.is_cuda or not b.is_cuda:
        return None
    if any(d != 2 for d in a.shape) or any(d != 2 for d in b.shape):
        return None
    L = _lib.lib()
    if not L.tcmi_tensordot_bits_small_ok(a.dim(), b.dim(), nk):
        # bigger operands: the tile kernel stores through the permutation but does not conjugate, and the streaming
        # big x small kernel keeps the steps it is good at (their transposition stays a launch of its own)
        rc_ = a.dim() + b.dim() - 2 * nk
        if flags or a.dim() > 31 or b.dim() > 31 or list(out_perm) == list(range(rc_)):
            return None
        # a transposed store is 8-byte scattered unless the three fastest axes stay in place (64-byte runs): worth it
        # while the launch it saves costs more than the store (2.15 vs 1.70 ms per slice backward when rank-19..21
        # results were stored that way)
        if rc_ > FUSED_PERM_MAX_RANK and list(out_perm[-3:]) != list(range(rc_ - 3, rc_)):
            return None
        fa_ = [i for i in range(a.dim()) if i not in axes_a]
        fb_ = [i for i in range(b.dim()) if i not in axes_b]
        if _scattered_ok(a, b, nk, fa_, fb_):
            return None
    a, b = a.contiguous(), b.contiguous()
    rc = a.dim() + b.dim() - 2 * nk
    out = torch.empty([2] * rc, dtype=a.dtype, device=a.device)
    xa = (ctypes.c_int * max(nk, 1))(*axes_a)
    xb = (ctypes.c_int * max(nk, 1))(*axes_b)
    op = (ctypes.c_int * max(rc, 1))(*out_perm) if rc else None
    if COUNTERS is not None:
        COUNTERS["gemm_launches"] += 1
        COUNTERS["gemm_flops"] += 8.0 * (1 << (a.dim() + b.dim() - nk))
        COUNTERS["gemm_bytes"] += float(a.numel() + b.numel() + out.numel()) * a.element_size()
    stream = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(L.tcmi_tensordot_bits_ex(a.data_ptr(), a.dim(), b.data_ptr(), b.dim(), ctypes.cast(xa, ctypes.c_void_p),
                                        ctypes.cast(xb, ctypes.c_void_p), nk,
                                        ctypes.cast(op, ctypes.c_void_p) if op is not None else None, int(flags),
                                        out.data_ptr(), _lib.TCMI_C64, stream), "tcmi_tensordot_bits_ex")
    return out


def _fns():
    if _FN:
        return _FN
    import torch

    def vmap_loop(fn, info, in_dims, *args):
        outs = []
        for i in range(info.batch_size):
            outs.append(fn(*[a.select(d, i) if d is not None else a for a, d in zip(args, in_dims)]))
        return torch.stack(outs), 0

    # new-style Functions (setup_context + vmap rule): usable under torch.func.grad / vmap, which is how
    # backend.value_and_grad / vmap / vvag reach K.tensordot / K.transpose / K.einsum
    class PermuteFn(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(t, perm):
            return _permute_raw(t.contiguous(), perm)

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.perm = tuple(inputs[1])

        @staticmethod
        def vmap(info, in_dims, t, perm):
            return vmap_loop(lambda x: PermuteFn.apply(x, perm), info, in_dims[:1], t)

        @staticmethod
        def backward(ctx, g):
            inv = [0] * len(ctx.perm)
            for i, p in enumerate(ctx.perm):
                inv[p] = i
            return permute(g.contiguous(), tuple(inv)), None

    class TensordotFn(torch.autograd.Function):
        """``tensordot`` of two [2]^rank tensors with the fast kernels in BOTH directions: forward = the untaped route
        (scattered / bit-deposit MFMA / permute + GEMM), backward = two more tensordots of the same engine."""
        generate_vmap_rule = False

        @staticmethod
        def forward(a, b, axes_a, axes_b):
            return _tensordot_raw(a, b, list(axes_a), list(axes_b))

        @staticmethod
        def setup_context(ctx, inputs, output):
            a, b, axes_a, axes_b = inputs
            ctx.save_for_backward(a, b)
            ctx.axes = (tuple(axes_a), tuple(axes_b))

        @staticmethod
        def vmap(info, in_dims, a, b, axes_a, axes_b):
            return vmap_loop(lambda x, y: TensordotFn.apply(x, y, axes_a, axes_b), info, in_dims[:2], a, b)

        @staticmethod
        def backward(ctx, g):
            a, b = ctx.saved_tensors
            xa, xb = list(ctx.axes[0]), list(ctx.axes[1])
            g = g.contiguous()
            if _on_tape(a, b, g):      # double backward / nested transforms: stay differentiable
                fa, fb, perm_a, perm_b = _vjp_axes(a.dim(), b.dim(), xa, xb)
                ga = permute(tensordot(g, _conj(b), list(range(len(fa), len(fa) + len(fb))), fb), perm_a) \
                    if ctx.needs_input_grad[0] else None
                gb = permute(tensordot(_conj(a), g, fa, list(range(len(fa)))), perm_b) if ctx.needs_input_grad[1] else None
                return ga, gb, None, None
            ga, gb = tensordot_vjp(a, b, xa, xb, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            return ga, gb, None, None

    _FN.update(PermuteFn=PermuteFn, TensordotFn=TensordotFn)
    return _FN


def _transpose2(t, rows, cols):
    """[rows x cols] -> [cols x rows] for power-of-two shapes, through the bit-permute kernel."""
    r, c = int(np.log2(rows)), int(np.log2(cols))
    flat = t.reshape([2] * (r + c)) if r + c > 0 else t.reshape([])
    if r == 0 or c == 0:
        return t.reshape(-1)
    return _permute_raw(flat, tuple(range(r, r + c)) + tuple(range(r))).reshape(-1)


def permute(t, perm):
    """Differentiable axis permutation on the device (K2)."""
    perm = tuple(int(p) % max(1, t.dim()) for p in perm)
    if sorted(perm) != list(range(t.dim())):
        raise ValueError(f"permute: {perm} is not a permutation of the {t.dim()} axes")
    if perm == tuple(range(t.dim())):
        return t
    if not _on_tape(t):
        return _permute_raw(t.contiguous(), perm)        # no autograd node: half the host time of a tiny step
    return _fns()["PermuteFn"].apply(t.contiguous(), perm)


def _on_tape(*ts) -> bool:
    import torch

    for t in ts:
        if torch._C._functorch.is_functorch_wrapped_tensor(t) or (t.requires_grad and torch.is_grad_enabled()):
            return True
    return False


def tensordot(a, b, axes_a: Sequence[int], axes_b: Sequence[int]):
    """``backend.tensordot(a, b, [axes_a, axes_b])`` on the HIP engine (K1b): result axes = a's free
    axes in order, then b's (the ``contract_between`` convention).  Operands on an autograd tape (or inside a
    ``torch.func`` transform) go through ``TensordotFn``: the same kernels forward, and backward."""
    import torch

    if a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(dt), b.to(dt)
    axes_a, axes_b = [int(x) for x in axes_a], [int(x) for x in axes_b]
    if _on_tape(a, b):
        return _fns()["TensordotFn"].apply(a, b, tuple(axes_a), tuple(axes_b))
    return _tensordot_raw(a, b, axes_a, axes_b)


def _tensordot_raw(a, b, axes_a, axes_b):
    """The untaped step: scattered big x small kernel, else the bit-deposit MFMA kernel, else permute + GEMM."""
    fa = [i for i in range(a.dim()) if i not in axes_a]
    fb = [i for i in range(b.dim()) if i not in axes_b]
    r = _tensordot_scattered(a, b, axes_a, axes_b, fa, fb)
    if r is not None:
        return r
    r = _tensordot_bits(a, b, axes_a, axes_b)
    if r is not None:
        return r
    a2 = _permute_raw(a.contiguous(), fa + axes_a)
    b2 = _permute_raw(b.contiguous(), axes_b + fb)
    M, K, N = 2 ** len(fa), 2 ** len(axes_a), 2 ** len(fb)
    c = _gemm_raw(a2, b2, M, N, K)
    return c.reshape([2] * (len(fa) + len(fb)))


def _tensordot_bits(a, b, axes_a, axes_b):
    """tensordot of two plain complex64 [2]^rank device tensors straight from their stored layouts
    (``tcmi_tensordot_bits``: MFMA kernel with bit-deposit addressing, no operand permuted).  None when the step does
    not qualify (autograd tape / functorch, other dtypes or shapes): the permute + GEMM route handles it."""
    import torch

    if knob("tn_bits", "1") == "0":
        return None
    if a.dtype != torch.complex64 or not a.is_cuda or not b.is_cuda:
        return None
    if a.dim() > 31 or b.dim() > 31 or any(d != 2 for d in a.shape) or any(d != 2 for d in b.shape):
        return None
    a, b = a.contiguous(), b.contiguous()
    nk = len(axes_a)
    out = torch.empty([2] * (a.dim() + b.dim() - 2 * nk), dtype=a.dtype, device=a.device)
    xa = (ctypes.c_int * max(nk, 1))(*axes_a)
    xb = (ctypes.c_int * max(nk, 1))(*axes_b)
    if COUNTERS is not None:
        COUNTERS["gemm_launches"] += 1
        COUNTERS["gemm_flops"] += 8.0 * (1 << (a.dim() + b.dim() - nk))
        COUNTERS["gemm_bytes"] += float(a.numel() + b.numel() + out.numel()) * a.element_size()
    stream = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(_lib.lib().tcmi_tensordot_bits(a.data_ptr(), a.dim(), b.data_ptr(), b.dim(),
                                             ctypes.cast(xa, ctypes.c_void_p), ctypes.cast(xb, ctypes.c_void_p), nk,
                                             out.data_ptr(), _lib.TCMI_C64, stream), "tcmi_tensordot_bits")
    return out


TN_STREAMS = int(knob("tn_streams", "2"))   # two slices of a sliced contraction side by side (1: one stream)
TN_STREAMS_SMALL = int(knob("tn_streams_small", "4"))   # ... and up to this many when the largest intermediate has <= 2^22 elements (sliced value_and_grad only)
SCATTERED_MIN_RANK = 16    # big operand: at least 2^16 elements
SCATTERED_MAX_SMALL = 4096  # small operand: at most this many elements (it lives in LDS)
SCATTERED_MAX_NK = int(knob("tn_scat_maxk", "8"))   # more contracted axes: the MFMA bits kernel
FUSED_PERM_MAX_RANK = int(knob("tn_fused_perm_rank", "0"))   # tile kernel: transposed stores up to this result rank (measured: no gain over the permute launch, 1.71-1.75 vs 1.70 ms per slice backward at 12-14; worse above)
SCATTERED_MIN_FREE = int(knob("tn_scat_minfree", "14"))   # at least 2^14 threads (free indices of the big operand)
SCATTERED_MAX_OUT = int(knob("tn_scat_maxout", "32"))     # at most this many outputs per thread


def _scattered_ok(a, b, nk, fa, fb) -> bool:
    """Whether ``_tensordot_scattered`` takes the step."""
    if nk < 1 or nk > SCATTERED_MAX_NK:
        return False
    big_first = a.numel() >= b.numel()
    big, small = (a, b) if big_first else (b, a)
    if big.dim() < SCATTERED_MIN_RANK or small.numel() > SCATTERED_MAX_SMALL or small.numel() < (1 << nk):
        return False
    if nk > 5 and (small.numel() >> nk) > 16:
        return False
    # one thread per free index of the big operand, n outputs each: it needs many threads with little to do each.
    # (The reverse sweep of a sliced network has steps like rank 16 x rank 12 over 8 axes -- 256 threads -- or rank 17
    # x rank 12 over 5 -- 128 outputs per thread: 100 us here, 20 us on the split-K tile kernel.)
    if big.dim() - nk < SCATTERED_MIN_FREE or (small.numel() >> nk) > SCATTERED_MAX_OUT:
        return False
    if any(d != 2 for d in big.shape) or any(d != 2 for d in small.shape):
        return False
    return True


def _tensordot_scattered(a, b, axes_a, axes_b, fa, fb):
    """Big tensor x small tensor without permuting the big one (``tcmi_contract_scattered``): used when no
    gradient is needed, 1..5 axes are contracted and the small operand has at most 4096 elements.  Returns
    None when the step does not qualify (the permute + GEMM route handles it)."""
    import torch

    nk = len(axes_a)
    if not _scattered_ok(a, b, nk, fa, fb):
        return None
    big_first = a.numel() >= b.numel()
    big, small = (a, b) if big_first else (b, a)
    ax_big, ax_small = (axes_a, axes_b) if big_first else (axes_b, axes_a)
    f_small = fb if big_first else fa
    pairs = sorted(zip(ax_big, ax_small))                      # ascending big axis = descending bit position
    small2 = _permute_raw(small.contiguous(), [p[1] for p in pairs] + list(f_small)).contiguous()
    big = big.contiguous()
    rank = big.dim()
    pos = sorted(rank - 1 - p[0] for p in pairs)
    n = small.numel() >> nk
    out = torch.empty((1 << (rank - nk)) * n, dtype=big.dtype, device=big.device)
    arr = (ctypes.c_int * nk)(*pos)
    if COUNTERS is not None:
        COUNTERS["scattered_launches"] += 1
        COUNTERS["scattered_flops"] += 8.0 * out.numel() * (1 << nk)
        COUNTERS["scattered_bytes"] += float(big.numel() + small.numel() + out.numel()) * big.element_size()
    stream = torch.cuda.current_stream(big.device).cuda_stream
    _lib.check(_lib.lib().tcmi_contract_scattered(big.data_ptr(), rank, ctypes.cast(arr, ctypes.c_void_p), nk,
                                                  small2.data_ptr(), n, out.data_ptr(), int(big_first), _code(big),
                                                  stream), "tcmi_contract_scattered")
    return out.reshape([2] * (len(fa) + len(fb)))


def contract_between(na: Node, nb: Node) -> Node:
    shared = [e for e in na.edges if e in nb.edges]
    t = tensordot(na.tensor, nb.tensor, [na.edges.index(e) for e in shared], [nb.edges.index(e) for e in shared])
    return Node(t, [e for e in na.edges if e not in shared] + [e for e in nb.edges if e not in shared])


def edge_symbol(i: int) -> str:
    """opt_einsum.get_symbol: the letters a-zA-Z, then unicode from chr(192) on (the symbols a path finder plugged in
    through ``set_contractor("custom", optimizer=f)`` receives; reference cons.py:783-800)."""
    letters = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    return letters[i] if i < 52 else chr(i + 140)


def symbolic_info(nodes: Sequence[Node]):
    """(input_sets, output_set, size_dict) in the form the reference hands to a path finder: one symbol per edge,
    edges numbered by first appearance (reference cons._get_path_cache_friendly, cons.py:773-800)."""
    inputs, output, size_dict = get_tn_info(nodes)
    num: Dict[int, int] = {}
    for s_ in inputs:
        for e in s_:
            num.setdefault(e, len(num))
    sym = {e: edge_symbol(k) for e, k in num.items()}
    return ([[sym[e] for e in s_] for s_ in inputs], [sym[e] for e in output], {sym[e]: v for e, v in size_dict.items()})


def contract_nodes(nodes: Sequence[Node], output_edge_order: Optional[Sequence[int]] = None,
                   target_size: Optional[int] = None, ignore_edge_order: bool = False, optimizer: Any = None,
                   memory_limit: Any = None, debug_level: int = 0, info: bool = False, trials: int = 0,
                   strip_exponent: bool = False):
    """The contractor call of the reference (``contractor(nodes, output_edge_order=..., ignore_edge_order=...)``,
    ``_base`` at cons.py:845-961) on the HIP engine: pairwise path -> ``tcmi_cgemm`` / ``tcmi_contract_scattered`` /
    ``tcmi_permute_bits`` steps, result axes in ``output_edge_order``.  Returns a Node.

    ``optimizer``: the path-finder plug-in of ``set_contractor("custom", optimizer=...)``: a callable
    ``f(input_sets, output_set, size_dict, memory_limit=None) -> [(i, j), ...]`` in opt_einsum's linear format, or a
    precomputed list path (cons.py:1037-1040, 949-950); default: the built-in (random-)greedy search.
    ``debug_level`` 1 / 2: no arithmetic, zeros of the output shape (cons.py:928-934).  Errors as cons.py:877-896."""
    hyper = strip_exponent or any(isinstance(nd, CopyNode) for nd in nodes)
    if not hyper:
        cnt: Dict[int, int] = {}
        for nd in nodes:
            for e in nd.edges:
                cnt[e] = cnt.get(e, 0) + 1
        hyper = any(v > 2 for v in cnt.values())
    if hyper:
        # CopyNodes / indices shared by more than two tensors / strip_exponent: the einsum-style route of the reference
        # (cons.py:706-766); dangling edges of CopyNodes are valid output edges
        arrays, inputs, output, size_dict, dangling = hyper_info(nodes)
        if not ignore_edge_order:
            if output_edge_order is None:
                output_edge_order = dangling
            if set(output_edge_order) != set(dangling):
                raise ValueError("output edges are not equal to the remaining non-contracted edges of the final node.")
        else:
            output_edge_order = dangling
        rep = dict(zip(dangling, output))
        want = [rep[e] for e in output_edge_order]
        path = None
        if optimizer is not None and len(arrays) > 1:
            if isinstance(optimizer, (list, tuple)):
                path = [tuple(int(x) for x in ab) for ab in optimizer]
            else:
                num: Dict[int, int] = {}
                for s_ in inputs:
                    for e in s_:
                        num.setdefault(e, len(num))
                for e in want:
                    num.setdefault(e, len(num))
                sym = {e: edge_symbol(k) for e, k in num.items()}
                path = [tuple(int(x) for x in ab) for ab in optimizer(
                    [[sym[e] for e in s_] for s_ in inputs], [sym[e] for e in want],
                    {sym[e]: size_dict[e] for e in num}, memory_limit=memory_limit)]
            path = [ab for ab in path if len(ab) == 2]
        r = contract_hyper(arrays, inputs, want, size_dict, path=path, strip_exponent=strip_exponent)
        if strip_exponent:
            return Node(r[0], list(output_edge_order)), float(r[1])
        return Node(r, list(output_edge_order))
    inputs, output, size_dict = get_tn_info(nodes)
    if not ignore_edge_order:
        if output_edge_order is None:
            if len(output) > 1:
                raise ValueError(
                    "The final node after contraction has more than one remaining edge. "
                    "In this case `output_edge_order` has to be provided."
                )
            output_edge_order = output
        if set(output_edge_order) != set(output):
            raise ValueError("output edges are not equal to the remaining non-contracted edges of the final node.")
    else:
        output_edge_order = output
    arrays = [n.tensor for n in nodes]
    if debug_level:
        ref = arrays[0]
        z = cons.backend.zeros([size_dict[e] for e in output_edge_order], dtype=str(ref.dtype).split(".")[-1])
        return Node(z, list(output_edge_order))
    path = None
    if optimizer is not None:
        if isinstance(optimizer, (list, tuple)):
            path = [tuple(int(x) for x in ab) for ab in optimizer]
        else:
            sin, sout, ssz = symbolic_info(nodes)
            path = [tuple(int(x) for x in ab) for ab in optimizer(sin, sout, ssz, memory_limit=memory_limit)]
        path = [ab for ab in path if len(ab) == 2]   # single-element entries are skipped (cons.py:943-945)
    tree = ContractionTree.from_path(inputs, list(output_edge_order), size_dict, path=path, trials=trials)
    if info:
        print("contraction cost: log2[FLOPs] = %.3f  log2[WRITE] = %.3f  log2[SIZE] = %.1f" % (
            float(np.log2(max(tree.total_flops(), 1))), float(np.log2(max(tree.total_write(), 1))),
            float(tree.contraction_width())))
    if target_size is not None:
        tree.slice_to(target_size)
    if not tree.sliced_inds:
        return Node(tree.contract_core(arrays), list(output_edge_order))
    total = None
    for r in tree.contract_slices(arrays, range(tree.nslices)):
        total = r if total is None else total + r
    return Node(total, [e for e in output_edge_order if e not in tree.sliced_inds])
