"""Process-global runtime configuration, mirroring the reference ``tensorcircuit/cons.py``:
``set_backend`` (``:90-135``), ``set_dtype`` (``:185-239``), ``set_contractor`` (``:1123-1261``) and
their function-decorator / context-manager variants (``:143-182, 247-284, 1269-1314``).

As in the reference, the setters broadcast the new value into every already-imported module of the
package (reference ``cons.py:84-87,131-134,229-234``), so ``tcmi.backend`` / ``tcmi.dtypestr`` follow.
"""

import sys
from contextlib import contextmanager
from functools import wraps
from typing import Any, Callable, Dict, Iterator, Optional

import numpy as np

package_name = "tcmi"
thismodule = sys.modules[__name__]

dtypestr = "complex64"
rdtypestr = "float32"
idtypestr = "int32"
npdtype = np.complex64
backend: Any = None
contractor: Any = None
_contractor_name = "greedy"
_plan_options: Dict[str, Any] = {}


def _broadcast(name: str, value: Any) -> None:
    for modname, mod in list(sys.modules.items()):
        if mod is not None and (modname == package_name or modname.startswith(package_name + ".")):
            if hasattr(mod, name):
                setattr(mod, name, value)
    setattr(thismodule, name, value)


def set_backend(backend_name: Optional[str] = None, set_global: bool = True) -> Any:
    """reference cons.py:90-135.  The only registered backend is ``"hip"``."""
    from .backends import get_backend

    if backend_name is None:
        backend_name = "hip"
    b = get_backend(backend_name)
    if set_global:
        _broadcast("backend", b)
    return b


set_tensornetwork_backend = set_backend


def set_dtype(dtype: Optional[str] = None, set_global: bool = True):
    """reference cons.py:185-239: complex64/complex128 (or the matching float names)."""
    if not dtype:
        dtype = "complex64"
    if dtype == "complex64" or dtype == "float32":
        c, r, i = "complex64", "float32", "int32"
    elif dtype == "complex128" or dtype == "float64":
        c, r, i = "complex128", "float64", "int64"
    else:
        raise ValueError(f"Unsupported data type: {dtype}")
    if set_global:
        _broadcast("dtypestr", c)
        _broadcast("rdtypestr", r)
        _broadcast("idtypestr", i)
        _broadcast("npdtype", getattr(np, c))
    return c, r


get_dtype = lambda: (dtypestr, rdtypestr)

_KNOWN_CONTRACTORS = (
    "auto", "greedy", "branch", "optimal", "plain", "plain-experimental", "tng", "custom",
    "custom_stateful", "tilevm", "cut",
)


def set_contractor(method: Optional[str] = None, optimizer: Any = None, memory_limit: Any = None,
                   opt_conf: Any = None, set_global: bool = True, contraction_info: bool = False,
                   debug_level: int = 0, **kws: Any) -> Callable[..., Any]:
    """reference cons.py:1123-1261.  Returns (and installs as ``tc.contractor``) the callable

        contractor(nodes, output_edge_order=None, ignore_edge_order=False) -> Node

    of the reference's contractor plug-in contract (cons.py:377-382, 845-896), executed on the HIP tensordot engine
    (``tcmi.tn.contract_nodes``).  ``method``:

    * ``"custom"``: ``optimizer`` is the path finder ``f(input_sets, output_set, size_dict, memory_limit)`` or a
      precomputed list path (cons.py:1037-1040); ``"custom_stateful"``: ``optimizer`` is a class instantiated with
      ``opt_conf`` for every contraction (cons.py:1062-1081);
    * the reference's path-finder names (``greedy``, ``branch``, ``optimal``, ``auto``, ``tng``, ``plain``,
      ``cotengra*``, ``omeco*``) select the built-in search: greedy, or random-greedy + subtree reconfiguration for
      the names that ask for more than greedy; the result of a contraction does not depend on the path;
    * ``"tilevm"`` / ``"cut"`` (not in the reference) force the two state-vector execution orders of ``Circuit``.

    ``strip_exponent=True``: the callable returns ``(node, exponent)`` with result = node * 10**exponent, intermediates
    rescaled on the way (cons.py:736-740, 763-766); networks with ``tn.CopyNode`` hyperedges take the same route.
    ``debug_level`` 1 / 2: contractions return zeros of the right shape without arithmetic (cons.py:928-934);
    ``contraction_info``: cost of every contraction is printed (cons.py:1084-1120).  Whole-circuit state vectors do not
    go through this callable: ``Circuit.wavefunction`` runs the compiled tile-VM plan, whose tuning knobs may be
    passed here as ``lowbits`` / ``R`` / ``LT``."""
    if not method:
        method = "greedy"
    base = method.split("-")[0]
    if method not in _KNOWN_CONTRACTORS and base not in ("cotengra", "omeco"):
        raise ValueError("Unknown contractor type: %s" % method)
    plan_keys = ("lowbits", "R", "LT")
    # any further keyword travels, as in the reference, into the contractor call where ``_base(**kws)`` ignores it
    # (cons.py:852, 1229-1258: e.g. the ``max_time`` / ``minimize`` of tests/test_circuit.py:929-937)
    opts = {k: v for k, v in kws.items() if k in plan_keys}
    opts["debug_level"] = int(debug_level)
    opts["contraction_info"] = bool(contraction_info)
    trials = 0 if method in ("greedy", "plain", "plain-experimental", "tng", "tilevm", "cut") else 16

    def cf(nodes: Any, output_edge_order: Any = None, ignore_edge_order: bool = False, **ckws: Any) -> Any:
        from . import tn

        opt = None
        if method in ("custom", "custom_stateful") and optimizer is None:
            raise ValueError("set_contractor(%r) needs an `optimizer`" % method)
        if method == "custom":
            opt = optimizer
        elif method == "custom_stateful":
            opt = optimizer(**(opt_conf or {}))
        return tn.contract_nodes(list(nodes), output_edge_order, ignore_edge_order=ignore_edge_order, optimizer=opt,
                                 memory_limit=memory_limit, debug_level=int(ckws.get("debug_level", debug_level)),
                                 info=bool(contraction_info), trials=trials,
                                 strip_exponent=bool(ckws.get("strip_exponent", kws.get("strip_exponent", False))))

    cf.method = method  # type: ignore
    cf.plan_options = opts  # type: ignore
    if set_global:
        _broadcast("contractor", cf)
        _broadcast("_contractor_name", method)
        _broadcast("_plan_options", opts)
    return cf


def split_rules(max_singular_values: Optional[int] = None, max_truncation_err: Optional[float] = None,
                relative: bool = False) -> Dict[str, Any]:
    """reference cons.py ``split_rules``: truncation options for SVD splits (only given keys set)."""
    rules: Dict[str, Any] = {}
    if max_singular_values is not None:
        rules["max_singular_values"] = max_singular_values
    if max_truncation_err is not None:
        rules["max_truncation_err"] = max_truncation_err
    if relative is not None:  # as the reference (cons.py:1337): the key is always present
        rules["relative"] = relative
    return rules


def set_function_backend(backend_name: Optional[str] = None) -> Callable[..., Any]:
    """reference cons.py:143-166."""

    def wrapper(f: Callable[..., Any]) -> Callable[..., Any]:
        @wraps(f)
        def newf(*args: Any, **kws: Any) -> Any:
            old = backend.name if backend is not None else None
            set_backend(backend_name)
            try:
                return f(*args, **kws)
            finally:
                set_backend(old)

        return newf

    return wrapper


@contextmanager
def runtime_backend(backend_name: Optional[str] = None) -> Iterator[Any]:
    """reference cons.py:169-182."""
    old = backend.name if backend is not None else None
    K = set_backend(backend_name)
    try:
        yield K
    finally:
        set_backend(old)


def set_function_dtype(dtype: Optional[str] = None) -> Callable[..., Any]:
    """reference cons.py:247-268."""

    def wrapper(f: Callable[..., Any]) -> Callable[..., Any]:
        @wraps(f)
        def newf(*args: Any, **kws: Any) -> Any:
            old = dtypestr
            set_dtype(dtype)
            try:
                return f(*args, **kws)
            finally:
                set_dtype(old)

        return newf

    return wrapper


@contextmanager
def runtime_dtype(dtype: Optional[str] = None) -> Iterator[Any]:
    """reference cons.py:271-284."""
    old = dtypestr
    r = set_dtype(dtype)
    try:
        yield r
    finally:
        set_dtype(old)


def set_function_contractor(*confargs: Any, **confkws: Any) -> Callable[..., Any]:
    """reference cons.py:1269-1294."""

    def wrapper(f: Callable[..., Any]) -> Callable[..., Any]:
        @wraps(f)
        def newf(*args: Any, **kws: Any) -> Any:
            old_name, old_opts = _contractor_name, dict(_plan_options)
            set_contractor(*confargs, **confkws)
            try:
                return f(*args, **kws)
            finally:
                set_contractor(old_name, **{k: v for k, v in old_opts.items() if k in ("lowbits", "R", "LT")})

        return newf

    return wrapper


@contextmanager
def runtime_contractor(*confargs: Any, **confkws: Any) -> Iterator[Any]:
    """reference cons.py:1297-1314."""
    old_name, old_opts = _contractor_name, dict(_plan_options)
    r = set_contractor(*confargs, **confkws)
    try:
        yield r
    finally:
        set_contractor(old_name, **{k: v for k, v in old_opts.items() if k in ("lowbits", "R", "LT")})
