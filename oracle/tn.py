"""Tensor-network restatement of the reference's contraction path (test infrastructure).

Follows, function by function:

* ``Circuit.__init__`` / ``all_zero_nodes``      tensorcircuit/circuit.py:44-131, basecircuit.py:51-66
* ``apply_general_gate`` wiring                  tensorcircuit/basecircuit.py:183-290
* ``_copy`` (+conj)                              tensorcircuit/basecircuit.py:150-181
* ``wavefunction``                               tensorcircuit/circuit.py:701-721
* ``expectation_before`` / ``expectation``       tensorcircuit/basecircuit.py:393-447, circuit.py:833-913
* ``amplitude_before`` / ``amplitude``           tensorcircuit/basecircuit.py:562-624
* ``custom`` contractor (greedy + preprocessing) tensorcircuit/cons.py:1007-1050
* ``_merge_single_gates``                        tensorcircuit/cons.py:298-374
* ``_base`` pairwise loop                        tensorcircuit/cons.py:845-961
* ``plain_contractor``                           tensorcircuit/cons.py:429-463

Third-party arithmetic that is NOT under /root/reference is restated from
its published behaviour: ``tensornetwork`` (pinned only in the unused freeze file
``requirements/requirements-2411.txt:198`` as ``tensornetwork-ng==0.5.0``):
``contract_between`` = ``tensordot`` over all shared edges, output axes = a's remaining
axes then b's; ``opt_einsum==3.4.0`` (``requirements-2411.txt:107``): ``paths.greedy``
picks the pair minimising ``size(out) - size(a) - size(b)`` and returns a linear-format
path.  The path changes rounding only, never the exact result.
"""

import heapq
import itertools
from collections import deque

import numpy as np

from . import gates as G


class Node:
    """A tensor with integer edge labels; a label is shared by at most two nodes
    (a dangling edge appears once)."""

    __slots__ = ("tensor", "edges")

    def __init__(self, tensor, edges):
        self.tensor = tensor
        self.edges = list(edges)
        assert tensor.ndim == len(self.edges)


def contract_between(a, b):
    """tensornetwork ``contract_between(a, b, allow_outer_product=True)``: tensordot over
    all shared edges; result axes = a's free axes in order, then b's free axes."""
    shared = [e for e in a.edges if e in b.edges]
    ax_a = [a.edges.index(e) for e in shared]
    ax_b = [b.edges.index(e) for e in shared]
    t = np.tensordot(a.tensor, b.tensor, axes=(ax_a, ax_b))
    edges = [e for e in a.edges if e not in shared] + [
        e for e in b.edges if e not in shared
    ]
    return Node(t, edges)


def _edge_owner_count(nodes):
    cnt = {}
    for nd in nodes:
        for e in nd.edges:
            cnt[e] = cnt.get(e, 0) + 1
    return cnt


def merge_single_gates(nodes):
    """tensorcircuit/cons.py:298-374 restated: every node of rank <= 2 is absorbed into
    the neighbour across its first non-dangling edge; the merged node takes the later of
    the two list slots; merged nodes of rank <= 2 go back to the front of the queue."""
    nodes = list(nodes)
    owners = {}
    for i, nd in enumerate(nodes):
        for e in nd.edges:
            owners.setdefault(e, []).append(i)
    queue = deque(i for i, nd in enumerate(nodes) if nd.tensor.ndim <= 2)
    alive = set(queue)
    while queue:
        i0 = queue.popleft()
        if i0 not in alive:
            continue
        alive.discard(i0)
        n0 = nodes[i0]
        if n0 is None or len(n0.edges) == 0:
            continue
        e0 = None
        for e in n0.edges[:2]:
            if len(owners[e]) == 2:
                e0 = e
                break
        if e0 is None:
            continue
        i1, i2 = owners[e0]
        new = contract_between(nodes[i1], nodes[i2])
        early, late = (i1, i2) if i1 < i2 else (i2, i1)
        alive.discard(i1)
        alive.discard(i2)
        nodes[early] = None
        nodes[late] = new
        for e in new.edges:
            owners[e] = [late if o in (i1, i2) else o for o in owners[e]]
        if new.tensor.ndim <= 2:
            queue.appendleft(late)
            alive.add(late)
    return [nd for nd in nodes if nd is not None]


def greedy_path(input_sets, output_set, size_dict):
    """opt_einsum ``paths.greedy`` restated (published algorithm): repeatedly contract the
    pair sharing an index that minimises ``size(out) - size(a) - size(b)``; disconnected
    leftovers are combined by outer products, smallest first.  Returns a linear-format
    path: each ``(a, b)`` indexes the *current* list, both are removed, result appended
    (tensorcircuit/cons.py:937-950)."""
    output = frozenset(output_set)
    # ssa bookkeeping
    ssa = {i: frozenset(s) for i, s in enumerate(input_sets)}
    next_id = len(ssa)

    def size(s):
        r = 1
        for x in s:
            r *= size_dict[x]
        return r

    def index_count():
        cnt = {}
        for s in ssa.values():
            for x in s:
                cnt[x] = cnt.get(x, 0) + 1
        return cnt

    cnt = index_count()

    def result_of(a, b):
        sa, sb = ssa[a], ssa[b]
        keep = set()
        for x in sa | sb:
            c = cnt[x] - (x in sa) - (x in sb)
            if c > 0 or x in output:
                keep.add(x)
        return frozenset(keep)

    heap = []
    owners = {}
    for i, s in ssa.items():
        for x in s:
            owners.setdefault(x, set()).add(i)

    def push_pairs(i):
        seen = set()
        for x in ssa[i]:
            for j in owners.get(x, ()):
                if j != i and j not in seen and j in ssa:
                    seen.add(j)
                    a, b = (i, j) if i < j else (j, i)
                    out = result_of(a, b)
                    cost = size(out) - size(ssa[a]) - size(ssa[b])
                    heapq.heappush(heap, (cost, a, b))

    for i in list(ssa):
        push_pairs(i)
    # dedupe is unnecessary: stale entries are skipped on pop
    ssa_path = []
    while heap:
        cost, a, b = heapq.heappop(heap)
        if a not in ssa or b not in ssa:
            continue
        out = result_of(a, b)
        # stale cost? recompute and re-push if it changed
        real = size(out) - size(ssa[a]) - size(ssa[b])
        if real != cost:
            heapq.heappush(heap, (real, a, b))
            continue
        for x in ssa[a]:
            cnt[x] -= 1
            owners[x].discard(a)
        for x in ssa[b]:
            cnt[x] -= 1
            owners[x].discard(b)
        del ssa[a], ssa[b]
        k = next_id
        next_id += 1
        ssa[k] = out
        for x in out:
            cnt[x] = cnt.get(x, 0) + 1
            owners.setdefault(x, set()).add(k)
        ssa_path.append((a, b))
        push_pairs(k)
    # outer products of what is left, smallest first
    rest = sorted(ssa, key=lambda i: size(ssa[i]))
    while len(rest) > 1:
        a, b = rest[0], rest[1]
        out = ssa[a] | ssa[b]
        k = next_id
        next_id += 1
        ssa[k] = frozenset(out)
        ssa_path.append((a, b))
        rest = sorted([k] + rest[2:], key=lambda i: size(ssa[i]))
    # ssa -> linear
    ids = list(range(len(input_sets)))
    path = []
    nxt = len(input_sets)
    for a, b in ssa_path:
        ia, ib = ids.index(a), ids.index(b)
        path.append((ia, ib) if ia < ib else (ib, ia))
        for i in sorted((ia, ib), reverse=True):
            ids.pop(i)
        ids.append(nxt)
        nxt += 1
    return path


def optimal_path(input_sets, output_set, size_dict):
    """opt_einsum ``paths.optimal`` for < 5 tensors (cons.py:1019-1020): exhaustive search
    over pairwise orders minimising total flops; restated as brute force."""
    n = len(input_sets)
    if n == 1:
        return []
    if n == 2:
        return [(0, 1)]
    output = set(output_set)

    def size(s):
        r = 1
        for x in s:
            r *= size_dict[x]
        return r

    best = [None, None]

    def rec(sets, path, cost):
        if best[0] is not None and cost >= best[0]:
            return
        if len(sets) == 1:
            best[0], best[1] = cost, list(path)
            return
        for i, j in itertools.combinations(range(len(sets)), 2):
            a, b = sets[i], sets[j]
            others = [s for k, s in enumerate(sets) if k not in (i, j)]
            keep = set(output)
            for s in others:
                keep |= s
            out = (a | b) & keep
            flops = size(a | b)
            rec(others + [out], path + [(i, j)], cost + flops)

    rec([set(s) for s in input_sets], [], 0)
    return best[1]


def contract(nodes, output_edge_order=None, preprocessing=True, method="greedy", stats=None):
    """``cons.custom(optimizer=greedy, preprocessing=True)`` -> ``cons._base``
    (tensorcircuit/cons.py:1007-1050, 845-961).  ``method='plain'`` follows
    ``plain_contractor`` (cons.py:429-463).  ``stats`` (dict) receives the executed plan's
    algorithmic bytes/flops as defined in SURVEY.md section 8(d)."""
    nodes = list(nodes)
    if stats is None:
        stats = {}
    stats.setdefault("elems", 0)
    stats.setdefault("flops", 0)
    stats.setdefault("steps", 0)

    def account(a, b, c):
        shared = [e for e in a.edges if e in b.edges]
        k = 1
        for e in shared:
            k *= a.tensor.shape[a.edges.index(e)]
        stats["elems"] += a.tensor.size + b.tensor.size + c.tensor.size
        stats["flops"] += 8 * c.tensor.size * k
        stats["steps"] += 1

    dangling = [e for e, c in _edge_owner_count(nodes).items() if c == 1]
    if output_edge_order is None:
        if len(dangling) > 1:
            raise ValueError(
                "The final node after contraction has more than one remaining edge. "
                "In this case `output_edge_order` has to be provided."
            )
        output_edge_order = dangling
    if set(output_edge_order) != set(dangling):
        raise ValueError(
            "output edges are not equal to the remaining non-contracted edges of the final node."
        )
    if method == "plain":
        nodes = list(reversed(nodes))
        while len(nodes) > 1:
            a, b = nodes[-1], nodes[-2]
            new = contract_between(a, b)
            account(a, b, new)
            nodes = nodes[:-2] + [new]
    else:
        if len(nodes) < 5:
            finder = optimal_path
        else:
            finder = greedy_path
            if preprocessing:
                nodes = merge_single_gates(nodes)
        if len(nodes) > 1:
            size_dict = {}
            for nd in nodes:
                for e, d in zip(nd.edges, nd.tensor.shape):
                    size_dict[e] = d
            path = finder([nd.edges for nd in nodes], list(output_edge_order), size_dict)
            for a, b in path:
                na, nb = nodes[a], nodes[b]
                new = contract_between(na, nb)
                account(na, nb, new)
                nodes = [nd for k, nd in enumerate(nodes) if k not in (a, b)] + [new]
    final = nodes[0]
    perm = [final.edges.index(e) for e in output_edge_order]
    return Node(np.transpose(final.tensor, perm), list(output_edge_order))


class Circuit:
    """The reference ``tc.Circuit`` restricted to the hot path, on numpy arrays.

    Qubit 0 is the most significant bit of the flat state (tests/test_circuit.py:47-53)."""

    def __init__(self, nqubits, inputs=None, dtype=np.complex128, method="greedy"):
        self._nqubits = nqubits
        self.dtype = np.dtype(dtype)
        self.method = method
        self._edge_counter = itertools.count()
        self._nodes = []
        self._front = []
        self.state_tensor = None
        self.stats = {}
        if inputs is None:
            # basecircuit.py:51-66: n rank-1 nodes [1, 0]
            for _ in range(nqubits):
                e = next(self._edge_counter)
                self._nodes.append(Node(np.array([1.0, 0.0], dtype=self.dtype), [e]))
                self._front.append(e)
        else:
            # circuit.py:90-104: one rank-n node
            t = np.asarray(inputs).astype(self.dtype).reshape([2] * nqubits)
            edges = [next(self._edge_counter) for _ in range(nqubits)]
            self._nodes.append(Node(t, edges))
            self._front = list(edges)

    # ---- gate application (basecircuit.py:183-290) -------------------------------
    def apply(self, matrix, *index):
        if len(index) != len(set(index)):
            raise ValueError(
                f"gate index {list(index)} has duplicate qubits; "
                "each qubit may appear at most once"
            )
        index = tuple(i if i >= 0 else self._nqubits + i for i in index)
        k = len(index)
        t = np.asarray(matrix).astype(self.dtype).reshape([2] * (2 * k))
        out_edges = [next(self._edge_counter) for _ in range(k)]
        in_edges = [self._front[q] for q in index]
        self._nodes.append(Node(t, out_edges + in_edges))
        for j, q in enumerate(index):
            self._front[q] = out_edges[j]
        self.state_tensor = None

    any = unitary = apply

    def h(self, i): self.apply(G.H, i)
    def x(self, i): self.apply(G.X, i)
    def y(self, i): self.apply(G.Y, i)
    def z(self, i): self.apply(G.Z, i)
    def s(self, i): self.apply(G.S, i)
    def t(self, i): self.apply(G.T, i)
    def sd(self, i): self.apply(G.SD, i)
    def td(self, i): self.apply(G.TD, i)
    def cnot(self, i, j): self.apply(G.CNOT, i, j)
    cx = cnot
    def cz(self, i, j): self.apply(G.CZ, i, j)
    def cy(self, i, j): self.apply(G.CY, i, j)
    def swap(self, i, j): self.apply(G.SWAP, i, j)
    def toffoli(self, i, j, k): self.apply(G.TOFFOLI, i, j, k)
    def rx(self, i, theta=0.0): self.apply(G.rx(theta), i)
    def ry(self, i, theta=0.0): self.apply(G.ry(theta), i)
    def rz(self, i, theta=0.0): self.apply(G.rz(theta), i)
    def phase(self, i, theta=0.0): self.apply(G.phase(theta), i)
    def r(self, i, theta=0.0, alpha=0.0, phi=0.0): self.apply(G.r(theta, alpha, phi), i)
    def u(self, i, theta=0.0, phi=0.0, lbd=0.0): self.apply(G.u(theta, phi, lbd), i)
    def iswap(self, i, j, theta=1.0): self.apply(G.iswap(theta), i, j)
    def cr(self, i, j, theta=0.0, alpha=0.0, phi=0.0): self.apply(G.cr(theta, alpha, phi), i, j)
    def crx(self, i, j, theta=0.0): self.apply(G.controlled(G.rx(theta)), i, j)
    def exp1(self, *index, unitary=None, theta=0.0, half=False):
        self.apply(G.exp1(unitary, theta, half), *index)
    def exp(self, *index, unitary=None, theta=0.0): self.apply(G.exp(unitary, theta), *index)
    def rzz(self, i, j, theta=0.0): self.apply(G.rzz(theta), i, j)
    def rxx(self, i, j, theta=0.0): self.apply(G.rxx(theta), i, j)
    def ryy(self, i, j, theta=0.0): self.apply(G.ryy(theta), i, j)
    def su4(self, i, j, theta=None): self.apply(G.su4(theta), i, j)

    # ---- copies (basecircuit.py:150-181) ----------------------------------------
    def _copy(self, conj=False):
        remap = {}
        nodes = []
        for nd in self._nodes:
            edges = []
            for e in nd.edges:
                if e not in remap:
                    remap[e] = next(self._edge_counter)
                edges.append(remap[e])
            t = nd.tensor.conj() if conj else nd.tensor.copy()
            nodes.append(Node(t, edges))
        return nodes, [remap[e] for e in self._front]

    def _copy_state_tensor(self, conj=False, reuse=True):
        # basecircuit.py:375-391
        if reuse:
            if self.state_tensor is None:
                nodes, front = self._copy()
                self.state_tensor = contract(nodes, front, method=self.method, stats=self.stats)
            t = self.state_tensor
            edges = [next(self._edge_counter) for _ in t.edges]
            return [Node(t.tensor.conj() if conj else t.tensor, edges)], edges
        return self._copy(conj)

    # ---- outputs ------------------------------------------------------------------
    def wavefunction(self, form="default"):
        """circuit.py:701-721."""
        nodes, front = self._copy()
        t = contract(nodes, front, method=self.method, stats=self.stats)
        out = t.tensor.reshape(-1)
        if form == "ket":
            out = out.reshape(-1, 1)
        elif form == "bra":
            out = out.reshape(1, -1)
        return out

    state = wavefunction

    def expectation_before(self, *ops, reuse=True):
        """basecircuit.py:393-447: bra[q]^op[j], ket[q]^op[j+k]; untouched ket[j]^bra[j]."""
        nq = self._nqubits
        nodes1, e1 = self._copy_state_tensor(reuse=reuse)
        nodes2, e2 = self._copy_state_tensor(conj=True, reuse=reuse)
        nodes = nodes1 + nodes2
        newdang = list(e1) + list(e2)
        rename = {}
        occupied = set()
        for op, index in ops:
            if isinstance(index, int):
                index = [index]
            index = tuple(i if i >= 0 else nq + i for i in index)
            k = len(index)
            t = np.asarray(op).astype(self.dtype).reshape([2] * (2 * k))
            edges = []
            for j, q in enumerate(index):
                if q in occupied:
                    raise ValueError(
                        f"Cannot measure two operators in one index: qubit {q} "
                        f"is already occupied by a previous operator in this "
                        f"measurement, index={index}"
                    )
                occupied.add(q)
            edges = [newdang[q + nq] for q in index] + [newdang[q] for q in index]
            nodes.append(Node(t, edges))
        for j in range(nq):
            if j not in occupied:
                rename[newdang[j + nq]] = newdang[j]
        for nd in nodes:
            nd.edges = [rename.get(e, e) for e in nd.edges]
        return nodes

    def expectation(self, *ops, reuse=True):
        """circuit.py:833-913 (noise-free branch): complex scalar."""
        nodes = self.expectation_before(*ops, reuse=reuse)
        return contract(nodes, [], method=self.method, stats=self.stats).tensor[()]

    def expectation_ps(self, x=None, y=None, z=None, ps=None, reuse=True):
        """abstractcircuit.py:1523-1603."""
        ops = []
        if ps is not None:
            x = [i for i, p in enumerate(ps) if p == 1]
            y = [i for i, p in enumerate(ps) if p == 2]
            z = [i for i, p in enumerate(ps) if p == 3]
        for lst, m in ((x, G.X), (y, G.Y), (z, G.Z)):
            if lst is not None:
                for i in lst:
                    ops.append((m, [i]))
        return self.expectation(*ops, reuse=reuse)

    def amplitude(self, l):
        """basecircuit.py:562-624: cap every output leg with onehot(l_i) (quantum.py:166-182)."""
        if isinstance(l, str):
            l = [int(ch) for ch in l]
        nodes, front = self._copy()
        for bit, e in zip(l, front):
            v = np.zeros(2, dtype=self.dtype)
            v[int(bit)] = 1.0
            nodes.append(Node(v, [e]))
        return contract(nodes, [], method=self.method, stats=self.stats).tensor[()]
