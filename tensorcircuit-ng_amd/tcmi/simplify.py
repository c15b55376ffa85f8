"""Causal light-cone simplification of <psi| O |psi> networks (reference tensorcircuit/simplify.py:198-296,
used by ``Circuit.expectation(..., enable_lightcone=True)``, reference circuit.py:897-901).

The uncontracted expectation network of ``Circuit.expectation_before(reuse=False)`` holds every gate twice: U on the
ket side and U^dagger (the conjugated tensor, same leg order) on the bra side, tagged with ``is_dagger`` and the gate's
``id``.  A gate whose output legs all run straight into the same legs of its own conjugate is outside the causal cone
of every operator: U^dagger U = 1, the pair is removed and its input legs are joined.  Repeating this peels the
network back to the cone.  Nodes are ``tcmi.tn.Node`` (integer edge labels): a label shared by two nodes is a
contracted edge, so "joining" two legs is a relabelling.  Unlike the reference the pair must also be known to be
unitary (constant non-unitary matrices handed to ``c.any`` stay in the network)."""
from typing import Any, List, Tuple

from .tn import node_is_unitary


def _light_cone_cancel(nodes: List[Any]) -> Tuple[List[Any], bool]:
    """One backward scan: cancel every ket gate whose outputs meet its own conjugate (reference simplify.py:198-270)."""
    changed = False
    removed = set()
    owners = {}
    for nd in nodes:
        for e in nd.edges:
            owners.setdefault(e, []).append(nd)
    rename = {}

    def cur(e):
        while e in rename:
            e = rename[e]
        return e

    for n in reversed(nodes):
        if id(n) in removed or n.is_dagger is not False:
            continue
        noe = len(n.edges)
        if noe % 2 != 0 or noe == 0:
            continue
        k = noe // 2
        match = None
        ok = True
        for leg in range(k):
            e = cur(n.edges[leg])
            # partner on this output leg: the node (other than n) that carries the same current label at the SAME leg
            cand = [m for m in owners_of(owners, rename, e) if m is not n and id(m) not in removed]
            if len(cand) != 1:
                ok = False
                break
            m = cand[0]
            if m.is_dagger is not True or m.id != n.id or len(m.edges) != noe or cur(m.edges[leg]) != e:
                ok = False
                break
            if match is None:
                match = m
            elif match is not m:
                ok = False
                break
        if not ok or match is None or not node_is_unitary(n):
            continue
        # bypass: the input legs of n and of its conjugate become one edge
        for leg in range(k, noe):
            a, b = cur(n.edges[leg]), cur(match.edges[leg])
            if a != b:
                rename[b] = a
                owners.setdefault(a, []).extend(owners.get(b, []))
        removed.add(id(n))
        removed.add(id(match))
        changed = True
    if not changed:
        return nodes, False
    out = [nd for nd in nodes if id(nd) not in removed]
    for nd in out:
        nd.edges = [cur(e) for e in nd.edges]
    return out, True


def owners_of(owners, rename, e):
    """Nodes that carry edge label ``e`` (after the relabellings recorded so far)."""
    return owners.get(e, [])


def _full_light_cone_cancel(nodes: List[Any]) -> List[Any]:
    """Repeat ``_light_cone_cancel`` until nothing changes (reference simplify.py:276-296).  Untagged node lists are
    returned unchanged, as in the reference."""
    if not nodes:
        return nodes
    if any(getattr(n, "is_dagger", None) is None for n in nodes):
        return nodes
    nodes, changed = _light_cone_cancel(list(nodes))
    while changed:
        nodes, changed = _light_cone_cancel(nodes)
    return nodes
