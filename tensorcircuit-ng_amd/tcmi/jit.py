"""``backend.jit`` for the variational hot loop (reference idiom ``K.jit(K.value_and_grad(f))``,
``benchmarks/scripts/vqe_tc.py:136-141``; ``K.jit(K.vvag(f))`` in the batched benchmarks).

The reference's jit removes the Python cost of re-tracing the circuit on every step.  On the hip backend the
device plan is already cached by circuit structure, but re-running the user's function still costs
milliseconds of Python (gate recording, ~100 tensor indexings and their autograd nodes) for microseconds
of kernels on small circuits.  ``TracedVag`` removes that cost for the standard pattern

    f(*args) = real( sum_k w_k <P_k> )   of ONE circuit whose gate angles are elements of tensor arguments
    f(*args) = that circuit's wavefunction()                      (value-only forms: jit(f), jit(vmap(f)))

It is found by *probing*, not by a tensor subclass: ``f`` is called twice with index-valued arguments
(element g of the concatenated arguments holds g, then 2 g + 1); an angle that reads back (g, 2 g + 1) is
element g, one that reads back the same number twice is a constant, anything else (arithmetic on the
arguments, several circuits, non-linear post-processing, aux outputs) makes the function untraceable and
the plain path is used.  The result is the fixed pipeline

    angles = concat(args)[index]  ->  tcmi state plan  ->  fused measurement  ->  Pauli-sum cotangent
           ->  adjoint sweep  ->  index_add into the argument gradients

with no autograd graph and no Python per gate.  The first real call is checked against the plain path
(value and gradients); a mismatch disables the fast path for that signature."""

from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import cons
from . import gates as G


def _as_tuple(x):
    return (x,) if isinstance(x, int) else tuple(x)


class TracedState:
    """What ``Circuit.wavefunction`` returns while ``backend.jit`` probes a function: the function's output is the
    state of this circuit (nothing was evaluated)."""

    def __init__(self, circuit, form):
        self.circuit, self.form = circuit, form


_PROBE_CLS = []


def _probe_tensor(t):
    """A probe argument: a tensor whose VALUES may not steer Python (``bool``, ``float``, ``int``, ``item``, ``tolist``,
    ``numpy``).  A function that branches on, or concretises, its arguments cannot be replaced by a fixed pipeline -- JAX
    raises a concretisation error there; here the probe aborts and the function keeps the plain path."""
    import torch
    from . import _lib

    if not _PROBE_CLS:
        class ProbeTensor(torch.Tensor):
            def _abort(self, *a, **k):
                raise _lib.TraceAbort("the traced function uses the value of an argument in Python")

            __bool__ = __float__ = __int__ = __index__ = __complex__ = _abort
            item = tolist = numpy = __array__ = _abort

        _PROBE_CLS.append(ProbeTensor)
    return t.as_subclass(_PROBE_CLS[0])


def _plain(t):
    import torch

    return t.as_subclass(torch.Tensor) if _PROBE_CLS and isinstance(t, _PROBE_CLS[0]) else t


class TracedVag:
    def __init__(self, backend, slow: Callable[..., Any], f: Callable[..., Any], argnums, has_aux: bool,
                 vectorized_argnums=None, value_only: bool = False):
        self.backend, self.slow, self.f = backend, slow, f
        self.value_only = value_only
        self.argnums_raw = argnums
        self.argnums = _as_tuple(argnums)
        self.has_aux = has_aux
        self.vec = None if vectorized_argnums is None else _as_tuple(vectorized_argnums)
        self.plans: Dict[Any, Any] = {}
        self.stats = {"fast": 0, "slow": 0}

    # ---- signature / probing ---------------------------------------------------------------------------
    def _signature(self, args):
        import torch

        sig = []
        for a in args:
            if torch.is_tensor(a):
                sig.append(("t", tuple(a.shape), str(a.dtype)))
            elif isinstance(a, np.ndarray):
                sig.append(("n", a.shape, str(a.dtype)))
            elif isinstance(a, (int, float, complex, str, bool, type(None))):
                sig.append(("s", a))
            else:
                return None
        return (tuple(sig), cons.dtypestr, cons._contractor_name)

    def _probe(self, args, mult, add):
        """Arguments with element g of the concatenation replaced by mult * g + add (float64, on the host)."""
        import torch

        out, off, spans = [], 0, []
        for i, a in enumerate(args):
            if torch.is_tensor(a) or isinstance(a, np.ndarray):
                shape = tuple(a.shape)
                if self.vec is not None and i in self.vec:
                    shape = shape[1:]
                nel = int(np.prod(shape)) if shape else 1
                vals = (torch.arange(off, off + nel, dtype=torch.float64) * mult + add).reshape(shape)
                out.append(_probe_tensor(vals))
                spans.append((i, off, nel, shape))
                off += nel
            else:
                out.append(a)
        return out, spans, off

    def _trace(self, args):
        import torch
        from .expectation import LazyExpectation

        if self.has_aux:
            return False
        runs = []
        for mult, add in ((1.0, 0.0), (2.0, 1.0)):
            pa, spans, total = self._probe(args, mult, add)
            from . import _lib

            _lib.TRACING[0] = True
            try:
                out = self.f(*pa)
            except Exception:
                return False
            finally:
                _lib.TRACING[0] = False
            if isinstance(out, TracedState):
                # f(*args) = wavefunction of one circuit: the pipeline is gather -> state plan
                if not self.value_only or out.circuit.inputs is not None:
                    return False
                c = out.circuit
                pv = []
                for v in c._params:
                    if torch.is_tensor(v):
                        if v.numel() != 1:
                            return False
                        pv.append(float(_plain(v).detach().reshape(()).to(torch.float64).cpu()))
                    else:
                        pv.append(float(np.real(v)))
                runs.append((c, out.form, None, pv, spans, total))
                continue
            if not isinstance(out, LazyExpectation) or out._value is not None or not out.is_real:
                return False
            circuits = {id(t[0]): t[0] for t in out.terms}
            if len(circuits) != 1 or not G.is_concrete(out.const):
                return False
            c = next(iter(circuits.values()))
            if c.inputs is not None:
                return False
            terms: Dict[Tuple[int, ...], float] = {}
            for _, s, w in out.terms:
                if not G.is_concrete(w):
                    return False
                terms[tuple(int(v) for v in s)] = terms.get(tuple(int(v) for v in s), 0.0) + float(np.real(w))
            pv = []
            for v in c._params:
                if torch.is_tensor(v):
                    if v.numel() != 1:
                        return False
                    pv.append(float(_plain(v).detach().reshape(()).to(torch.float64).cpu()))
                else:
                    pv.append(float(np.real(v)))
            runs.append((c, terms, float(np.real(out.const)), pv, spans, total))
            out.terms = []          # drop the pending registration, nothing is evaluated
        if len(runs) != 2:
            return False
        (c, terms, const, pa_vals, spans, total), (c2, terms2, const2, pb_vals, _, _) = runs
        if terms != terms2 or const != const2 or len(pa_vals) != len(pb_vals) or len(c._ops) != len(c2._ops):
            return False
        index, consts = [], []
        for va, vb in zip(pa_vals, pb_vals):
            if va == vb:
                index.append(total + len(consts))
                consts.append(va)
            elif vb == 2.0 * va + 1.0 and va == int(va) and 0 <= va < total:
                index.append(int(va))
            else:
                return False
        from .executor import get_measure

        cc = c._compiled()
        dev = self.backend.device
        rdt = torch.float32 if cons.rdtypestr == "float32" else torch.float64
        if const is None:   # state-valued function (terms holds the requested form)
            return {
                "cc": cc, "state_form": terms, "nq": c._nqubits, "spans": spans, "total": total, "rdt": rdt,
                "index": torch.tensor(index, dtype=torch.int64, device=dev),
                "consts": torch.tensor(consts, dtype=rdt, device=dev), "checked": False,
            }
        strings = list(terms.keys())
        if not strings:
            return False
        cm = get_measure(c._nqubits, cc.n_exec, tuple(strings), cons.dtypestr)
        dev = self.backend.device
        rdt = torch.float32 if cons.rdtypestr == "float32" else torch.float64
        return {
            "cc": cc, "cm": cm, "spans": spans, "total": total, "const": const, "rdt": rdt,
            "index": torch.tensor(index, dtype=torch.int64, device=dev),
            "consts": torch.tensor(consts, dtype=rdt, device=dev),
            "weights": torch.tensor([terms[s] for s in strings], dtype=torch.float64, device=dev),
            "weights_host": [float(terms[s]) for s in strings],
            "checked": False,
        }

    # ---- fast path --------------------------------------------------------------------------------------
    def _fast(self, plan, args):
        import torch

        dev, rdt = self.backend.device, plan["rdt"]
        batched = self.vec is not None
        B = None
        if batched:
            for i, _, _, _ in plan["spans"]:
                if i in self.vec:
                    B = int(args[i].shape[0])
                    break
        flats = []
        for i, off, nel, shape in plan["spans"]:
            a = args[i]
            a = a if torch.is_tensor(a) else torch.as_tensor(a)
            a = a.detach().to(device=dev, dtype=rdt)
            if batched:
                a = a.reshape(B, nel) if i in self.vec else a.reshape(1, nel).expand(B, nel)
            else:
                a = a.reshape(1, nel)
            flats.append(a)
        nb = B if batched else 1
        flats.append(plan["consts"].reshape(1, -1).expand(nb, -1))
        flat = torch.cat(flats, dim=1)
        params = flat.index_select(1, plan["index"]).contiguous()          # [nb, P]
        if "state_form" in plan:
            psi = plan["cc"].state(params)                                     # [nb, 2^n]
            form = plan["state_form"]
            if not batched:
                psi = psi[0]
            if form == "ket":
                psi = psi.reshape(psi.shape[:-1] + (-1, 1))
            elif form == "bra":
                psi = psi.reshape(psi.shape[:-1] + (1, -1))
            return psi
        cc, cm = plan["cc"], plan["cm"]
        state = cc.state(params, full=True)                                   # [nb, 2^n_exec]
        w = plan["weights"]
        if self.value_only:
            vals = cm.run(state)                                              # [nb, T] complex128
            value = ((vals.real * w).sum(-1) + plan["const"]).to(rdt)
            return value if batched else value[0]
        # lambda = 2 sum_t w_t P_t |psi> and, from the same launches, Re <psi|lambda> = 2 sum_t w_t <P_t>: the energy comes
        # with its cotangent, no measurement passes (executor.CompiledMeasure.apply_sum)
        # ... and the single-X terms on the qubits of the sweep's first tile are not sent through memory at all: that pass
        # adds them to lambda in registers and returns their energies (executor.fold_setup; None: nothing to fold)
        fold = cc.fold_setup(cm, plan["weights_host"]) if hasattr(cc, "fold_setup") else None
        if fold is not None:
            if fold["lam_zero"]:     # every term is born in the sweep: no tile pass, lambda is a buffer nobody initialises
                lam, dot = torch.empty_like(state), 0.0
            else:
                lam, dot = cm.apply_sum(state, w.to(torch.complex128).reshape(1, -1).expand(nb, -1), want_dot=True,
                                        skip=fold["skip"])
            gp, efold = cc.vjp(params, state, lam, consume=True, from_zero=True, fold=fold)
            value = 0.5 * dot + efold + plan["const"]
            gp = gp.to(torch.float64)
        else:
            lam, dot = cm.apply_sum(state, w.to(torch.complex128).reshape(1, -1).expand(nb, -1), want_dot=True)
            value = 0.5 * dot + plan["const"]
            gp = cc.vjp(params, state, lam, consume=True, from_zero=True).to(torch.float64)   # [nb, P]; state and lam are ours
        gflat = torch.zeros(nb, plan["total"] + plan["consts"].numel(), dtype=torch.float64, device=dev)
        gflat.index_add_(1, plan["index"], gp)
        grads = []
        for j in self.argnums:
            span = next((s for s in plan["spans"] if s[0] == j), None)
            a = args[j]
            gdt = a.dtype if torch.is_tensor(a) and a.is_floating_point() else rdt
            if span is None:
                grads.append(torch.zeros_like(torch.as_tensor(a), dtype=gdt, device=dev))
                continue
            _, off, nel, shape = span
            g = gflat[:, off:off + nel]
            if batched and j in self.vec:
                g = g.reshape((B,) + tuple(shape))
            else:
                g = g.sum(0).reshape(tuple(shape))
            grads.append(g.to(gdt))
        value = value.to(rdt)
        value = value if batched else value[0]
        g_out = grads[0] if isinstance(self.argnums_raw, int) else tuple(grads)
        return value, g_out

    def __call__(self, *args: Any, **kws: Any):
        import torch

        key = None if kws else self._signature(args)
        if key is None:
            self.stats["slow"] += 1
            return self.slow(*args, **kws)
        plan = self.plans.get(key)
        if plan is None:
            with torch.no_grad():
                plan = self._trace(args)
            self.plans[key] = plan
        if plan is False:
            self.stats["slow"] += 1
            return self.slow(*args, **kws)
        if not plan["checked"]:
            from .expectation import resolve

            ref = resolve(self.slow(*args, **kws))
            got = self._fast(plan, args)
            tol = 2e-3 if cons.dtypestr == "complex64" else 1e-7
            ok = _close(ref, got, tol)
            if not ok:
                self.plans[key] = False
                self.stats["slow"] += 1
                return ref
            plan["checked"] = True
            self.stats["slow"] += 1
            return ref
        self.stats["fast"] += 1
        return self._fast(plan, args)


def _close(a, b, tol) -> bool:
    import torch

    if isinstance(a, (tuple, list)):
        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(_close(x, y, tol) for x, y in zip(a, b))
    if torch.is_tensor(a) and torch.is_tensor(b):
        if a.shape != b.shape:
            return False
        if not a.numel():
            return True
        cdt = torch.complex128 if (a.is_complex() or b.is_complex()) else torch.float64
        # in blocks: a batch of 28-qubit states must not be copied whole into complex128 twice just to be compared (16 GiB
        # each for four states -- the validating call was the memory peak of the whole step)
        fa, fb = a.detach().reshape(-1), b.detach().reshape(-1)
        blk = 1 << 24
        amax, dmax = 0.0, 0.0
        for o in range(0, fa.numel(), blk):
            x, y = fa[o: o + blk].to(cdt), fb[o: o + blk].to(cdt)
            amax = max(amax, float(x.abs().max()))
            dmax = max(dmax, float((x - y).abs().max()))
        if amax != amax or dmax != dmax:       # a NaN on either side is not "close"
            return False
        return dmax <= tol * max(1.0, amax)
    return False
