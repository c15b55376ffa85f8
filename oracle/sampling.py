"""Sequential computational-basis measurement on a dense state (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the qubit branch of the reference's ``measure_jit`` (tensorcircuit/basecircuit.py:468-558): for the
k-th measured qubit the conditional probability pu of outcome 0 given the earlier outcomes is compared with the
external uniform number ``status[k]``: outcome = 1 iff ``status[k] - pu + 0.31415926e-12 > 0`` (:519-524), and the
running probability is multiplied by pu or 1 - pu (:528)."""

import numpy as np


def measure(psi, n, index, status):
    """Returns (outcomes [len(index)] of 0/1, probability of that outcome string)."""
    prob = (np.abs(np.asarray(psi, dtype=np.complex128)) ** 2).reshape([2] * n)
    p = 1.0
    out = []
    for k, j in enumerate(index):
        marg = prob.sum(axis=tuple(a for a in range(n) if a != j))
        pu = marg[0] / marg.sum()
        s = 1 if (float(status[k]) - pu + 0.31415926e-12) > 0 else 0
        out.append(s)
        p *= pu if s == 0 else (1.0 - pu)
        # condition on the outcome: keep the slice, other outcome zeroed
        sl = [slice(None)] * n
        sl[j] = 1 - s
        prob = prob.copy()
        prob[tuple(sl)] = 0.0
    return np.array(out), p
