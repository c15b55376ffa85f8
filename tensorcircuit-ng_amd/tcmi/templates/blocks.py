"""reference ``tensorcircuit/templates/blocks.py``: circuit building blocks."""

from typing import Any

from .. import cons
from .. import gates as G

Tensor = Any


def example_block(c: Any, param: Tensor, nlayers: int = 2, is_split: bool = False) -> Any:
    """reference blocks.py:146-185 (the HEA-B ansatz of BASELINE configs 2 and 3, what bench.py times): one layer of
    Hadamards, then ``nlayers`` blocks of exp(i theta Z_i Z_i+1) in a ladder followed by rx on every qubit; ``param`` holds
    2 * nlayers * n angles.  ``is_split``: the reference's SVD split of the ZZ gate (max_singular_values 2 keeps the gate
    exactly: its operator-Schmidt rank is 2)."""
    split_conf = {"max_singular_values": 2, "fixed_choice": 1} if is_split else None
    n = c._nqubits
    param = cons.backend.reshape(param, [2 * nlayers, n])
    for i in range(n):
        c.H(i)
    for j in range(nlayers):
        for i in range(n - 1):
            c.exp1(i, i + 1, unitary=G._zz_matrix, theta=param[2 * j, i], split=split_conf)
        for i in range(n):
            c.rx(i, theta=param[2 * j + 1, i])
    return c


def state_centric(f):
    """reference blocks.py:27-54: wraps a circuit-in / circuit-out block as a state-in / state-out function."""
    from functools import wraps

    from ..circuit import Circuit

    @wraps(f)
    def wrapper(s, *args, **kws):
        n = int(cons.backend.reshape(s, [-1]).shape[0]).bit_length() - 1
        c = Circuit(n, inputs=s)
        return f(c, *args, **kws).state()

    return wrapper
