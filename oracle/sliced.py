"""numpy execution of a sliced pairwise contraction path (test infrastructure, see oracle/__init__.py; used by
tests/golden/make_golden_full.py for config 4's fixture and by bench.py's ``cpu_baseline`` of the config-4 leg).

What it restates: the reference's ``_base`` loop (tensorcircuit/cons.py:845-961: ``tn.contract_between`` = ``tensordot``
over all shared edges, pairs taken from an opt_einsum-format path) applied to one slice of a cotengra-sliced network --
every sliced index fixed to one value, the per-slice results summed by the caller
(tensorcircuit/experimental.py:999-1026,1145-1152).  The ORDER (path, sliced indices) is an input.
"""

import numpy as np


def contract_path(tensors, inputs, path, sliced, values, budget_s=None):
    """numpy tensordot chain along a linear-format pairwise path (pairs of positions in the shrinking list, the result
    appended, as opt_einsum / cotengra paths), with the ``sliced`` indices fixed to ``values``.  Closed network -> complex.
    ``budget_s``: raise TimeoutError((steps done, steps, seconds)) once the chain has run that long (bench.py's bounded
    CPU sample)."""
    import time

    t0 = time.perf_counter()
    sliced = list(sliced)
    ts, es = [], []
    for t, e in zip(tensors, inputs):
        sel = tuple(values[sliced.index(x)] if x in sliced else slice(None) for x in e)
        ts.append(t[sel])
        es.append([x for x in e if x not in sliced])
    for a, b in path:
        a, b = (a, b) if a < b else (b, a)
        tb, eb = ts.pop(b), es.pop(b)
        ta, ea = ts.pop(a), es.pop(a)
        common = [x for x in ea if x in eb]
        r = np.tensordot(ta, tb, axes=([ea.index(x) for x in common], [eb.index(x) for x in common]))
        ts.append(r)
        es.append([x for x in ea if x not in common] + [x for x in eb if x not in common])
        if budget_s is not None and time.perf_counter() - t0 > budget_s and len(ts) > 1:
            raise TimeoutError((len(path) - len(ts) + 1, len(path), time.perf_counter() - t0))
    assert len(ts) == 1 and es[0] == []
    return complex(ts[0])


def slice_values(slice_id, nsliced):
    """Bits of ``slice_id``, most significant first: the value of every sliced index in that slice."""
    return [(slice_id >> (nsliced - 1 - j)) & 1 for j in range(nsliced)]


def greedy_sliced_path(inputs, size_dict, max_width, log=None):
    """A pairwise path + sliced indices for a CLOSED network from the oracle's own tools only: ``oracle.tn.greedy_path``
    (the opt_einsum ``greedy`` restatement the reference's ``custom`` contractor uses, tensorcircuit/cons.py:1007-1050) on
    the network with some indices removed, the indices chosen one at a time -- always the one whose removal gives the
    smallest total cost 2^nsliced x (sum over steps of the product of the dimensions involved) once the largest
    intermediate fits ``max_width`` index bits, the narrowest intermediate before that.  What cotengra's slicing does for the
    reference (experimental.py:934-953), in its simplest form: no hyper-search, no reconfiguration.  Dimension 2 everywhere
    is not assumed.  Returns (path, sliced indices, log2 of the largest intermediate, log2 of the total cost)."""
    import numpy as np

    from . import tn as OT

    def stats(ins):
        path = OT.greedy_path([frozenset(s) for s in ins], frozenset(), size_dict)
        cur, width, cost = [frozenset(s) for s in ins], 0.0, 0.0
        for a, b in path:
            a, b = (a, b) if a < b else (b, a)
            sb, sa = cur.pop(b), cur.pop(a)
            r = (sa | sb) - (sa & sb)
            width = max(width, sum(np.log2(size_dict[e]) for e in r))
            cost += float(np.prod([float(size_dict[e]) for e in sa | sb]))
            cur.append(r)
        return path, width, np.log2(cost)

    cur = [list(s) for s in inputs]
    sliced = []
    shared = sorted({e for s in inputs for e in s if sum(e in t for t in inputs) == 2})
    while True:
        path, width, cost = stats(cur)
        total = cost + sum(np.log2(size_dict[e]) for e in sliced)
        if log:
            log(f"sliced {len(sliced)}: largest intermediate 2^{width:.0f}, total cost 2^{total:.2f}")
        if width <= max_width:
            return [tuple(p) for p in path], sliced, width, total
        best = None
        for e in shared:
            if e in sliced:
                continue
            _, w2, c2 = stats([[x for x in s if x != e] for s in cur])
            key = (max(w2, max_width), c2, w2)
            if best is None or key < best[0]:
                best = (key, e)
        sliced.append(best[1])
        cur = [[x for x in s if x != best[1]] for s in cur]
