"""Plan executor: uploads a CompiledPlan once and runs it on torch-ROCm tensors through the C ABI.

Host-side counterpart of the reference's ``contractor(nodes, output_edge_order=...)`` call inside
``Circuit.wavefunction`` (reference ``tensorcircuit/circuit.py:701-721``): the plan is compiled once
per circuit *structure* and cached, so the per-call host work is one parameter upload, one
table-builder launch and one launch per pass (this is what ``jax.jit`` buys the reference;
without it the reference re-runs ``tn.copy``, ``_merge_single_gates`` and the greedy search in
Python on every call, ``cons.py:1036,927``).
"""

import dataclasses
import hashlib
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from . import _streams
from ._knobs import knob
from . import plan as P
from . import specialize as S

# bench.py: when a list, every group of kernel launches of the executor is bracketed by HIP events on the launch
# stream and logged as (tag, start, end, launches, algorithmic bytes or flops) -- the live per-kernel durations the
# roofline figures are computed from.  None (default): no events are created.
EVENT_LOG = None
# scripts/gpu_spec_*.py: when a list, every pass launch of the reverse sweep is bracketed by its own pair of HIP events:
# (pass index, start, end)
PASS_EVENTS = None


VALU_LOG = None    # bench.py: {tag: [flops, VALU instructions x waves]} of the plan-specialised launches logged in EVENT_LOG


def _log_valu(tag, spec, fracs, n_exec, batch):
    """Arithmetic of the specialised kernels of one call, as executed: per pass ``specialize.pass_arithmetic`` (per wave) x
    waves of the live tiles x batch.  Interpreted passes are not counted (their arithmetic is not straight-line)."""
    if VALU_LOG is None:
        return
    ent = VALU_LOG.setdefault(tag, [0.0, 0.0, 0])
    for i, k in enumerate(spec):
        if k is None:
            ent[2] += 1
            continue
        ar = k.meta.get("arith")
        if not ar:
            continue
        waves = (2.0 ** (n_exec - int(k.meta["T"]))) * (1 << (int(k.meta["LT"]) - 6)) * (1.0 if fracs is None else fracs[i]) * batch
        ent[0] += ar["flops"] * waves
        ent[1] += ar["valu_instructions"] * waves


class _timed:
    def __init__(self, tag, launches, work):
        self.tag, self.launches, self.work = tag, launches, work

    def __enter__(self):
        if EVENT_LOG is not None:
            import torch

            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if EVENT_LOG is not None and exc[0] is None:
            self.e1.record()
            EVENT_LOG.append((self.tag, self.e0, self.e1, self.launches, self.work))
        return False


_T_MIN = 8
ATOMIC_COPIES = 64  # replication of atomically accumulated outputs (spreads same-address atomics)


def _dev(array, device, dtype=None):
    """Upload a numpy array as a plain device tensor.  Plans are compiled lazily, possibly inside a
    ``torch.func`` transform, where every op output (including ``.to(device)``) is a functorch
    wrapper without storage; cached plan tensors must be the underlying plain tensors."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(array))
    if dtype is not None:
        t = t.to(dtype)
    t = t.to(device)
    F = torch._C._functorch
    while F.is_functorch_wrapped_tensor(t):
        t = F.get_unwrapped(t)
    return t


def pick_variant(n: int, dtypestr: str, opts: Optional[dict] = None) -> Tuple[int, P.PlanConfig]:
    """Choose (n_exec, PlanConfig) for a circuit of n qubits: the largest tile variant that fits;
    tiny circuits are padded with spectator qubits (as most-significant bits, left in |0>)."""
    opts = opts or {}
    c64 = dtypestr == "complex64"
    n_exec = max(n, _T_MIN)
    if c64:
        variants = [(5, 8), (4, 8), (2, 6)]
    else:
        variants = [(4, 8), (3, 8), (2, 6)]
    if "R" in opts and "LT" in opts:
        variants = [(int(opts["R"]), int(opts["LT"]))] + variants
    elif knob("fwd_tile"):       # experiment switch: "R,LT" of the gate passes
        variants = [tuple(int(x) for x in knob("fwd_tile").split(","))] + variants
    for R, LT in variants:
        if R + LT <= n_exec:
            low = int(opts.get("lowbits", 5))
            low = max(1, min(low, R + LT))
            # complex64 tiles with R >= 4 run on the packed-f32 kernel (csrc/tcmi_vm2.hip), which knows OP_DIAGB2
            gen = 2 if (c64 and R >= 4 and not knob("vm1")) else 1
            cap = knob("pass_cap")
            return n_exec, P.PlanConfig(R=R, LT=LT, lowbits=low, vec=2 if c64 else 1, gen=gen,
                                        pass_cap=int(cap) if cap else None,
                                        shear2=knob("shear2", "1") != "0")
    raise ValueError("no tile variant fits")


def structure_digest(n: int, dtypestr: str, gates: List[P.GateRec]) -> str:
    h = hashlib.blake2b(digest_size=16)
    h.update(f"{n}|{dtypestr}|".encode())
    for g in gates:
        h.update(repr(g.qubits).encode())
        for m in (g.c0, g.c1, g.c2):
            h.update(b"-" if m is None else np.ascontiguousarray(m, dtype=np.complex128).tobytes())
        if g.param is not None:
            h.update(repr((g.param.index, g.param.scale, g.param.offset)).encode())
        h.update(b"d" if g.is_diag else b"n")
    return h.hexdigest()


def gate_is_unitary(g: P.GateRec, tol: float = 1e-6) -> bool:
    """Diagonal records are unit-modulus by construction; dense ones are checked (at sample angles when
    parametrised).  The adjoint sweep un-computes psi with U^dagger, which is only the inverse for unitary gates."""
    if g.is_diag:
        return True
    mats = list(g.select) if g.select is not None else None
    if mats is None:
        if g.param is None:
            mats = [np.asarray(g.c0, dtype=np.complex128)]
        else:
            mats = [np.asarray(g.c0, dtype=np.complex128) + np.cos(a) * np.asarray(g.c1) + np.sin(a) * np.asarray(g.c2)
                    for a in (0.37, 1.9, 4.1)]
    for m in mats:
        m = np.asarray(m, dtype=np.complex128)
        d = int(round(np.sqrt(m.size)))
        m = m.reshape(d, d)
        if np.abs(m @ m.conj().T - np.eye(d)).max() > tol:
            return False
    return True


def _shift_gate(g: P.GateRec, pad: int) -> P.GateRec:
    diag = None
    if g.diag is not None:
        diag = [P.DiagTerm(tuple(q + pad for q in t.qubits), t.const, t.param) for t in g.diag]
    return P.GateRec(tuple(q + pad for q in g.qubits), g.c0, g.c1, g.c2, g.param, diag, g.name)


LIVE_FULL = 0xFFFFFFFF
FOLD_CACHE_MAX = 8      # fold_setup records kept per compiled circuit (least recently used out first)
SPARSE_START = os.environ.get("TCMI_SPARSE_START", "1") != "0"
LIVE_PLAN = SPARSE_START and knob("live_plan", "1") != "0"    # plans chosen by the cost of their live tiles
NO_ZERO_FILL = knob("no_zero_fill", "1") != "0"   # |0...0> start without the zero fill (CompiledCircuit.zero_bits)


def choose_plan(n: int, gates: List[P.GateRec], nparams: int, dtypestr: str, opts: Optional[dict] = None):
    """(n_exec, cfg, plan, executed gate list) of a circuit: host work only (no device), deterministic -- also used to
    pre-compile the plan-specialised kernels of a known workload (tcmi/specialize.py, __graft_entry__.build)."""
    n_exec, cfg = pick_variant(n, dtypestr, opts)
    pad = n_exec - n
    if pad:
        gates = [_shift_gate(g, pad) for g in gates]
    plan = P.compile_plan(gates, n_exec, cfg, nparams=nparams)
    if cfg.gen >= 2 and "lowbits" not in (opts or {}) and n_exec >= 20 and len(gates) >= 64:
        # the greedy tile growth is sensitive to how many low bits are pinned and to how equal gains are broken (8 - 11
        # passes, differently balanced, at n = 28 d = 12): compile the neighbours too and keep the plan the pass model
        # likes best.  The model prices a pass on its live tiles (live_masks: the plan is for a state that starts from
        # |0...0>, the case that matters; with an input state every tile is live and the pick is a few per cent off)
        def cost(pl):
            return vm_cost_us(pl, live_masks(pl.descs, n_exec)[1] if LIVE_PLAN else None)

        best = cost(plan)
        for lb, tb in ((6, 0), (4, 0), (4, 1), (5, 1), (6, 1)):
            cfg2 = dataclasses.replace(cfg, lowbits=lb, tiebreak=tb)
            plan2 = P.compile_plan(gates, n_exec, cfg2, nparams=nparams)
            c2 = cost(plan2)
            if c2 < best * 0.995:
                best, plan, cfg = c2, plan2, cfg2
    return n_exec, cfg, plan, gates




def live_masks(descs, n: int, start_bits: int = 0, reverse: bool = False):
    """Which tiles of every pass can hold a non-zero amplitude when the circuit starts from |0...0>.

    An amplitude whose index has a 1 on a physical bit that no pass has had in its tile yet (and that no gate outside the
    plan touched: ``start_bits``) is exactly zero, and a pass acts inside its tiles (bits outside a tile only enter
    phases): a tile whose index is non-zero on such a bit is zero before and after the pass.  Per pass: a mask over the
    COMPACT tile index (bit i = the i-th physical bit outside the tile, ascending) of the bits that may vary,
    ``LIVE_FULL`` when every tile is live -- the ``live_mask`` of ``tcmi_spec_run_pass / _adjoint_pass``.
    ``reverse``: the passes are those of a reverse sweep (pass 0 un-does the END of the circuit): the psi entering sweep
    pass j carries the gates of passes j, j + 1, ... only, and a zero psi tile adds to no gradient slot whatever lambda
    holds there.  Returns (masks, fractions of live tiles)."""
    tiles = []
    for d in descs:
        w = np.asarray(d).view(np.uint32).astype(np.int64)
        tiles.append(sum(1 << int(w[8 + i]) for i in range(int(w[2]))))
    touched_before = []
    t = start_bits
    for tb in (reversed(tiles) if reverse else tiles):
        if reverse:
            t |= tb                       # psi entering this sweep pass already carries the pass's own gates
            touched_before.append(t)
        else:
            touched_before.append(t)
            t |= tb
    if reverse:
        touched_before.reverse()
    masks, fracs = [], []
    for tb, tch in zip(tiles, touched_before):
        free = [p_ for p_ in range(n) if not (tb >> p_) & 1]
        m = sum(1 << i for i, p_ in enumerate(free) if (tch >> p_) & 1)
        full = (1 << len(free)) - 1
        masks.append(LIVE_FULL if m == full else m)
        fracs.append(2.0 ** (bin(m).count("1") - len(free)))
    return masks, fracs


def choose_adjoint_plan(gates: List[P.GateRec], n_exec: int, dtypestr: str, full: bool, zero_start: bool = False):
    """(cfg, adjoint plan) of the executed gate list, or None when the short sweep has nothing to drop (use the full
    one).  Host work only, deterministic (see choose_plan).  ``zero_start`` (with ``full``): the sweep un-computes a psi
    that came from |0...0> -- candidates are priced on their live tiles (live_masks), and tiles of 8-amplitude runs are
    candidates too (the passes that stay dense are bound by VALU issue, not by how HBM likes its bursts)."""
    cfg = pick_adjoint_variant(n_exec, dtypestr, gates)
    if not full and not (gates and not P.gate_has_param(gates[0]) and any(P.gate_has_param(g) for g in gates)):
        return None
    ap = P.compile_adjoint_plan(gates, n_exec, cfg, factorized=cfg.gen >= 2, drop_constant_head=not full)
    if cfg.gen >= 2 and n_exec >= 20 and len(gates) >= 64:
        # the greedy schedule is sensitive to the pinned low bits and (short sweep) to the dropped gates: compile
        # the neighbours and keep what the pass model likes best; the short sweep may keep the full gate list
        live = zero_start and full and LIVE_PLAN

        def adj_cost(a):
            return adj_cost_us(a, live_masks(a.descs, n_exec, reverse=True)[1] if live else None)

        best = adj_cost(ap)
        force = knob("adj_force")      # experiment switch "lowbits,tiebreak[,pass cap]": that candidate, whatever the model says
        if force and full:
            lb, tb, *cap_ = (int(x) for x in force.split(","))
            cfg = dataclasses.replace(cfg, lowbits=lb, tiebreak=tb, pass_cap=cap_[0] if cap_ else None)
            return cfg, P.compile_adjoint_plan(gates, n_exec, cfg, factorized=True, drop_constant_head=False)
        for drop in ((True, False) if not full else (False,)):
            for lb, tb in ((5, 0), (4, 0), (4, 1), (5, 1)) + (((3, 0), (3, 1)) if live else ()):
                if best is None or ((lb, tb) == (cfg.lowbits, cfg.tiebreak) and drop == (not full)) \
                        or (drop is False and not full and (lb, tb) != (4, 1)):
                    continue
                cfg2 = dataclasses.replace(cfg, lowbits=lb, tiebreak=tb, pass_cap=None)
                ap2 = P.compile_adjoint_plan(gates, n_exec, cfg2, factorized=True, drop_constant_head=drop)
                c2 = adj_cost(ap2)
                if c2 is not None and c2 < best * 0.995:
                    best, ap, cfg = c2, ap2, cfg2
    return cfg, ap


def pick_adjoint_from_zero(exec_gates, n_exec: int, get_plan):
    """Host part of CompiledCircuit._adjoint_from_zero: of the short sweep (``get_plan(False)``) and the full gate list
    chosen for a psi from |0...0> (``get_plan(True)``) the one whose live tiles cost less under the pass model.
    ``get_plan(full)`` returns a record with "plan" (AdjointPlan).  Returns (record, masks, fractions)."""
    best = None
    for full in (False, True):
        adj = get_plan(full)
        if best is not None and adj is best[0]:
            continue
        ngates = sum(1 for _ in exec_gates)
        start = 0
        if not full:
            # qubits touched by the constant head a short plan may leave out (never un-computed; assumed left
            # out even where choose_adjoint_plan kept the whole gate list for the short sweep: conservative)
            first = next((i for i, g_ in enumerate(exec_gates) if P.gate_has_param(g_)), ngates)
            for g_ in exec_gates[:first]:
                for q in g_.qubits:
                    start |= 1 << (n_exec - 1 - q)
        masks, fracs = live_masks(adj["plan"].descs, n_exec, start_bits=start, reverse=True)
        cost = adj_cost_us(adj["plan"], fracs)
        if cost is None:
            cost = float(sum(fracs))
        if best is None or cost < best[3]:
            best = (adj, masks, fracs, cost)
    return best[:3]


def fold_plan_host(exec_gates, n_exec: int, adj0: dict, nparams: int, xw, dw=(), nterms: int = 0, pw=()):
    """Host part of CompiledCircuit.fold_setup (also run by specialize.precompile_circuit, without a GPU): the sweep plan
    of ``adj0`` (record of _adjoint_from_zero: its tile configuration and gate list) recompiled with terms of the Pauli-sum
    cotangent born in registers -- ``xw`` = [(term index, physical bit, weight[, kind])]: single-X (kind 0) / single-Y (kind 1) terms, each folded into the first
    pass whose tile holds its bit; ``dw`` = [(term index, Z mask over physical bits, weight)]: Z-only strings, folded at
    the start of the sweep; ``pw`` = [(term index, physical bit a, physical bit b, weight, kind a, kind b)]: strings with
    exactly two X / Y factors (XX, YY, XY: Heisenberg-type couplings), each folded by the first pass whose tile holds both
    bits while neither qubit has been touched (pairs that never meet that condition stay with the tile passes).
    ``nterms`` = number of terms of the whole sum: when every one of them is folded the first pass
    does not load lambda at all (FLAG_LAMBDA_ZERO).  Returns (AdjointPlan, indices of the folded terms, lam_zero) or None
    when fewer than four X / pair terms qualify or the schedule moved."""
    ap0 = adj0["plan"]
    seen, xs = set(), []
    for item in xw:
        k, bit, w = item[:3]
        kind = int(item[3]) if len(item) > 3 else 0      # 0: X, 1: Y
        if bit not in seen and float(w) != 0.0:          # (one folded term per qubit: a second one takes the tile passes)
            seen.add(bit)
            xs.append((k, bit, 2.0 * float(w), kind))
    ds = [(k, int(zm), 2.0 * float(w)) for k, zm, w in dw if float(w) != 0.0]
    ps = [(k, int(ba), int(bb), 2.0 * float(w), int(ka), int(kb)) for k, ba, bb, w, ka, kb in pw if float(w) != 0.0]
    if len(xs) + len(ps) < 4:
        return None
    kw = dict(factorized=True, drop_constant_head=bool(getattr(ap0, "drop_constant_head", False)),
              fold=[(bit, c, kind) for _, bit, c, kind in xs], fold_param=nparams,
              dfold=[(zm, c) for _, zm, c in ds] or None,
              fold2=[(ba, bb, c, ka, kb) for _, ba, bb, c, ka, kb in ps] or None)
    ap = P.compile_adjoint_plan(exec_gates, n_exec, adj0["cfg"], **kw)
    if [pp.tile_bits for pp in ap.passes] != [pp.tile_bits for pp in ap0.passes] or len(ap.folded) + len(ap.folded2) < 4:
        return None
    done = sorted([xs[i][0] for i in ap.folded] + [k for k, _, _ in ds] + [ps[i][0] for i in ap.folded2])
    lam_zero = bool(nterms) and len(done) == nterms
    if lam_zero:
        ap2 = P.compile_adjoint_plan(exec_gates, n_exec, adj0["cfg"], lam_zero=True, **kw)
        assert ap2.folded == ap.folded and ap2.folded2 == ap.folded2
        ap = ap2
    return ap, done, lam_zero


class CompiledCircuit:
    """A circuit structure lowered to tile-VM passes and resident on one GPU."""

    def __init__(self, n: int, gates: List[P.GateRec], nparams: int, dtypestr: str,
                 opts: Optional[dict] = None, device=None):
        import torch

        self.n = n
        self.dtypestr = dtypestr
        self.nparams = nparams
        self.n_exec, self.cfg, self.plan, gates = choose_plan(n, gates, nparams, dtypestr, opts)
        self._exec_gates = gates
        self._opts = opts
        # non-unitary gates (density-matrix channels, `any` with a non-unitary matrix): the reverse sweep cannot
        # un-compute psi through them; vjp() then works segment by segment from checkpoints
        self.nonunitary = [i for i, g in enumerate(gates) if not gate_is_unitary(g)]
        self.tdtype = torch.complex64 if dtypestr == "complex64" else torch.complex128
        self.rdtype = torch.float32 if dtypestr == "complex64" else torch.float64
        self.code = _lib.TCMI_C64 if dtypestr == "complex64" else _lib.TCMI_C128
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self._lib = _lib.lib()  # raises if the HIP extension is missing
        dev = self.device
        self.descs = [_dev(d, dev) for d in self.plan.descs]
        self.ctab = _dev(self.plan.ctab if self.plan.ctab.size else np.zeros(8), dev, self.rdtype)
        self.ginfo = _dev(self.plan.ginfo, dev)
        self.cpool = _dev(self.plan.cpool if self.plan.cpool.size else np.zeros(1), dev)
        self.nrec = int(self.plan.ginfo.shape[0])
        self.ptab_size = max(1, self.plan.ptab_size)

    _shift = staticmethod(_shift_gate)

    # ------------------------------------------------------------------------------------
    def build_ptab(self, params):
        """The per-batch table of ``state`` (gate records and phase tables from the parameter rows), built on the current
        stream: a caller that knows the parameters before it has the input states (the suffix of a cut half-circuit) builds
        it on a side stream, under the passes that produce those states, and hands it to ``state(..., ptab=...)``."""
        import torch

        params2 = params.reshape(-1, params.shape[-1]) if params.dim() > 1 else params.reshape(1, -1)
        params2 = params2.to(device=self.device, dtype=self.rdtype)
        if params2.stride(-1) != 1 or params2.stride(0) < params2.shape[1]:
            params2 = params2.contiguous()
        B = params2.shape[0]
        ptab = torch.empty(B, max(1, self.ptab_size), dtype=self.rdtype, device=self.device)
        if self.nrec:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(
                self._lib.tcmi_build_tables(
                    self.ginfo.data_ptr(), self.nrec, self.cpool.data_ptr(), params2.data_ptr(),
                    params2.stride(0), ptab.data_ptr(), ptab.stride(0), B, self.code, stream,
                ),
                "tcmi_build_tables",
            )
        return ptab

    def _specialised_src(self):
        """The generated kernel of the FIRST pass in its "src" variant (tcmi_spec_run_pass_from), or None."""
        if self.cfg.gen < 2 or self.dtypestr != "complex64" or self.n_exec != self.n or not self.plan.descs:
            return None
        if getattr(self, "_spec_src", None) is None:
            self._spec_src = S.PassSet("forward", self.plan.descs[:1], self.n_exec, {"src": 1})
        return self._spec_src.get()[0]

    def state(self, params=None, inputs=None, out=None, full=False, consume_inputs=False, ptab=None, src=None):
        """Run the plan.  ``params``: real tensor [B, P] (or [P]) on the device, or None when the
        circuit has no parameters.  Returns a complex tensor [B, 2^n].  ``consume_inputs``: the passes may run in
        place on ``inputs`` (a scratch batch the caller gives up) instead of on a copy.  ``ptab``: the table of
        ``build_ptab(params)``, already built.  ``src`` = (states [Bs, 2^n], shift, scale [B] or None): input state b is
        ``scale[b] * states[b >> shift]``, read by the first pass itself when its "src" kernel is there
        (tcmi_spec_run_pass_from), else materialised."""
        import torch

        ksrc = None
        if src is not None:
            assert inputs is None
            st_, sh_, sc_ = src
            ksrc = self._specialised_src() if (st_.dtype == self.tdtype and st_.is_contiguous()) else None
            if ksrc is None:      # no generated kernel for it: the replicated, weighted batch as an ordinary input
                Bs = st_.shape[0]
                rep_ = st_.reshape(Bs, 1, -1).expand(Bs, 1 << sh_, st_.shape[-1])
                if sc_ is not None:
                    rep_ = rep_ * sc_.reshape(Bs, 1 << sh_, 1)
                inputs, consume_inputs = rep_.reshape(Bs << sh_, -1), True      # (the caller gives ``states`` up)

        lib = self._lib
        if params is None:
            B = 1
            params2 = torch.zeros(1, 1, dtype=self.rdtype, device=self.device)
        else:
            params2 = params.reshape(-1, params.shape[-1]) if params.dim() > 1 else params.reshape(1, -1)
            params2 = params2.to(device=self.device, dtype=self.rdtype)
            if params2.stride(-1) != 1 or params2.stride(0) < params2.shape[1]:
                params2 = params2.contiguous()      # the builder takes a row stride, rows themselves must be dense
            B = params2.shape[0]
            if params2.shape[1] < self.nparams:
                raise ValueError("parameter vector shorter than the plan's parameter count")
            if params2.numel() == 0:     # an empty parameter row (constant gates on a differentiated input state)
                params2 = torch.zeros(max(B, 1), 1, dtype=self.rdtype, device=self.device)
                B = params2.shape[0]
        if inputs is not None:
            inp = inputs.reshape(-1, 2**self.n).to(device=self.device, dtype=self.tdtype)
            if params is None:
                B = inp.shape[0]
        nel = 2**self.n_exec
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if out is None:
            if (consume_inputs and inputs is not None and self.n_exec == self.n and inp.is_contiguous()
                    and inp.shape[0] == B and not inp.requires_grad):
                out = inp
            else:
                out = torch.empty(B, nel, dtype=self.tdtype, device=self.device)
        sparse = inputs is None and ksrc is None and SPARSE_START and self.cfg.gen >= 2 and self.dtypestr == "complex64"
        zbits = rfr = None
        # resolved ONCE per call: the kernel set the zero-fill decision and the byte accounting look at is the one
        # run_passes launches, and a call counts once towards the plan's hotness (specialize.PassSet.get)
        spec = self._specialised()
        if sparse and NO_ZERO_FILL and all(k is not None for k in spec):
            zb, rf, covered = self.zero_bits()
            if covered:
                zbits, rfr = zb, rf
        if ksrc is not None:
            pass                      # the first pass reads its tiles from src and writes every tile of out
        elif inputs is None and zbits is not None:
            out[:, 0] = 1.0           # everything else the passes read, earlier passes have written (zero_bits())
        elif inputs is None:
            _lib.check(
                lib.tcmi_init_zero_state(out.data_ptr(), nel, B, self.n_exec, self.code, stream),
                "tcmi_init_zero_state",
            )
        elif self.n_exec == self.n:
            if out is not inp:
                out.copy_(inp)
        else:
            out.zero_()
            out[:, : 2**self.n] = inp
        if ptab is None:
            ptab = torch.empty(B, self.ptab_size, dtype=self.rdtype, device=self.device)
            if self.nrec:
                _lib.check(
                    lib.tcmi_build_tables(
                        self.ginfo.data_ptr(), self.nrec, self.cpool.data_ptr(), params2.data_ptr(),
                        params2.stride(0), ptab.data_ptr(), ptab.stride(0), B, self.code, stream,
                    ),
                    "tcmi_build_tables",
                )
        item = 8 if self.dtypestr == "complex64" else 16
        live, units = None, 2.0 * len(self.descs)
        if sparse:
            # |0...0> start: the first passes only have a few tiles that can be non-zero (live_masks); the algorithmic
            # bytes of a pass are those of its live tiles (specialised kernels; an interpreted pass moves every tile),
            # written whole and read as far as they can be non-zero (zero_bits)
            masks, fracs = self.zero_start()
            if any(m != LIVE_FULL for m in masks) or zbits is not None:
                live = masks
                units = float(sum((f * (1.0 + (rfr[i] if rfr is not None else 1.0))) if k is not None else 2.0
                                  for i, (f, k) in enumerate(zip(fracs, spec))))
        _log_valu("pass", spec, self.zero_start()[1] if live is not None else None, self.n_exec, B)
        with _timed("pass", len(self.descs), units * B * nel * item):
            first = 0
            if ksrc is not None:
                st_, sh_, sc_ = src
                if sc_ is not None:
                    sc_ = sc_.reshape(-1).to(self.tdtype).contiguous()
                    assert sc_.numel() == B
                assert (st_.shape[0] << sh_) == B and st_.shape[1] == nel
                _lib.check(
                    lib.tcmi_spec_run_pass_from(ksrc.handle, out.data_ptr(), nel, B, self.n_exec, self.cfg.T, self.cfg.LT,
                                                self.ctab.data_ptr(), ptab.data_ptr(), ptab.stride(0), st_.data_ptr(),
                                                st_.stride(0), int(sh_), sc_.data_ptr() if sc_ is not None else None,
                                                stream),
                    "tcmi_spec_run_pass_from")
                first = 1
            self.run_passes(out, ptab, B, stream, first=first, live=live, zbits=zbits, spec=spec)
        if self.n_exec != self.n and not full:
            return out[:, : 2**self.n]
        return out

    def _specialised(self):
        spec = [None] * len(self.descs)
        if self.cfg.gen >= 2 and self.dtypestr == "complex64":
            # plan-specialised straight-line kernels where they exist (tcmi/specialize.py); the interpreter otherwise
            if getattr(self, "_spec_fwd", None) is None:
                self._spec_fwd = S.PassSet("forward", self.plan.descs, self.n_exec)
            spec = self._spec_fwd.get()
        return spec

    def zero_start(self):
        """(live masks, live fractions) of the passes for a state that starts from |0...0> (``live_masks``)."""
        if getattr(self, "_zero_start", None) is None:
            self._zero_start = live_masks(self.plan.descs, self.n_exec)
        return self._zero_start

    def zero_bits(self):
        """(per-pass ``zero_bits`` of tcmi_spec_run_pass, fraction of a live tile that is read, every bit covered?):
        the physical bits INSIDE pass k's tile that no earlier pass had in its tile -- amplitudes with one of them set are
        still zero when pass k loads its tile, so they are not read.  Pass k's live tiles are written whole, and they are
        exactly the amplitudes pass k + 1 reads: a state whose passes all run this way never reads an amplitude that was
        not written, i.e. it needs no zero fill -- provided the tiles cover every bit (else amplitudes nobody ever wrote
        would be left in the result)."""
        if getattr(self, "_zero_bits", None) is None:
            touched, zb, rf = 0, [], []
            for d in self.plan.descs:
                w = np.asarray(d).view(np.uint32).astype(np.int64)
                tb = sum(1 << int(w[8 + i]) for i in range(int(w[2])))
                u = tb & ~touched
                zb.append(u)
                rf.append(2.0 ** -bin(u).count("1"))
                touched |= tb
            self._zero_bits = (zb, rf, touched == (1 << self.n_exec) - 1)
        return self._zero_bits

    def run_passes(self, state, ptab, B, stream, first=0, last=None, live=None, zbits=None, spec=None):
        """``live``: per-pass live-tile masks (the state is |0...0> before pass 0); the specialised kernels then run on
        the live tiles only, an interpreted pass runs on all of them (zero tiles stay zero).  ``zbits``: per-pass
        ``zero_bits`` (amplitudes that are not read, see zero_bits())."""
        lib = self._lib
        nel = 2**self.n_exec
        if spec is None:
            spec = self._specialised()
        for i, (d, k) in list(enumerate(zip(self.descs, spec)))[first:last]:
            if k is not None:
                _lib.check(
                    lib.tcmi_spec_run_pass(k.handle, state.data_ptr(), nel, B, self.n_exec, self.cfg.T, self.cfg.LT,
                                           self.ctab.data_ptr(), ptab.data_ptr(), ptab.stride(0),
                                           LIVE_FULL if live is None else live[i], 0 if zbits is None else zbits[i],
                                           stream),
                    "tcmi_spec_run_pass",
                )
                continue
            _lib.check(
                lib.tcmi_run_pass(
                    state.data_ptr(), nel, B, self.n_exec, self.cfg.R, self.cfg.LT, d.data_ptr(),
                    self.ctab.data_ptr(), ptab.data_ptr(), ptab.stride(0), None, 0, 1, 0, self.code, stream,
                ),
                "tcmi_run_pass",
            )

    def stats(self):
        item = 8 if self.dtypestr == "complex64" else 16
        return self.plan.stats(item)

    # ---- reverse mode ------------------------------------------------------------------------
    def _adjoint(self, full: bool = True, zero_start: bool = False):
        """Adjoint-sweep plan (compiled lazily, own tile config: two vectors live in registers).  ``full=False``: the
        sweep may stop before the constant gates that open the circuit (no gradient slot behind them); the full plan is
        the one that also returns the input-state cotangent.  ``zero_start``: the full gate list, chosen for a psi that
        came from |0...0> (choose_adjoint_plan)."""
        import torch

        key = "_adj_zero_plan" if zero_start else ("_adj" if full else "_adj_short")
        if getattr(self, key, None) is None:
            res = choose_adjoint_plan(self._exec_gates, self.n_exec, self.dtypestr, full or zero_start, zero_start)
            if res is None:
                self._adj_short = self._adjoint(True)      # nothing to drop
                return self._adj_short
            cfg, ap = res
            dev = self.device
            setattr(self, key, {
                "plan": ap, "cfg": cfg,
                "descs": [_dev(d, dev) for d in ap.descs],
                "ctab": _dev(ap.ctab, dev, self.rdtype),
                "ginfo": _dev(ap.ginfo, dev),
                "cpool": _dev(ap.cpool if ap.cpool.size else np.zeros(1), dev),
                "gparam": _dev(ap.gslot_param, dev),
                "gfactor": _dev(ap.gslot_factor, dev),
                "nslots": len(ap.gslot_param),
            })
        return getattr(self, key)

    def _adjoint_from_zero(self):
        """The sweep plan for a psi that came from |0...0>, with its live-tile masks: psi is back to a few non-zero tiles
        in the last sweep passes (``live_masks``), so those run on a handful of workgroups.  The short sweep stops at the
        state behind the circuit's constant head -- non-zero wherever those gates acted, every pass dense for an ansatz
        that opens with a Hadamard layer -- so the full gate list is usually the cheaper sweep now; the pass model
        decides.  Returns (plan record, masks, fractions)."""
        if getattr(self, "_adj_zero", None) is None:
            self._adj_zero = pick_adjoint_from_zero(self._exec_gates, self.n_exec,
                                                    lambda full: self._adjoint(full, zero_start=full))
        return self._adj_zero

    def fold_setup(self, cm, weights):
        """Can part of the Pauli-sum cotangent of ``cm`` (weights: one float per term) be BORN in the reverse sweep instead of
        arriving through memory?  lambda = 2 sum_t w_t P_t |psi> costs 2 + 3 + 3 state transfers of tcmi_apply_pauli_sum_tiled
        for the n = 28 TFIM (one pass per 8-12 index bits that carry an X), and the sweep's first pass then reads it back.
        The single-X terms on the qubits of that first pass's TILE are added to lambda in registers there (plan.fold_rounds,
        OP_XFOLD: 24 packed instructions per term and thread), their energies arrive as gradient events, and the tile passes
        only cover the remaining terms.  The same holds for every later pass whose tile brings new qubits, and the Z-only
        strings need no neighbours at all: for the TFIM EVERY term is born in the sweep, no tile pass runs and the first pass
        does not even load lambda (8 + 1 state transfers less per sample).  Returns None (nothing to fold, or
        the plan-specialised kernels of the folded sweep are not there yet: only they execute OP_XFOLD) or a record
        {"skip": term indices left out of the tile passes, "adj": the sweep plan record, masks, fracs}."""
        if (os.environ.get("TCMI_PAULI_FOLD", "1") == "0" or not SPARSE_START or self.dtypestr != "complex64"
                or self.nonunitary or self.cfg.gen < 2 or S.mode() == "0"):
            return None
        # keyed by the terms themselves (an id() could be reused by another measurement object after this one is evicted)
        key = (tuple((tuple(t.x), tuple(t.z)) for t in cm.all_terms), tuple(float(w) for w in weights))
        # (bounded: every distinct weight vector -- an annealing schedule, re-traced energies -- pins a sweep plan's device
        # tables and its PassSet; the kernels themselves are shared through the code-object cache, their text only holds
        # table SLOTS, so an evicted entry costs two host-side plan compilations when it comes back)
        cache = self.__dict__.setdefault("_fold_cache", OrderedDict())
        if key in cache:
            cache.move_to_end(key)
        else:
            while len(cache) >= FOLD_CACHE_MAX:
                cache.popitem(last=False)
            cache[key] = None
            adj0, masks, fracs = self._adjoint_from_zero()
            if adj0["cfg"].gen >= 2 and adj0["plan"].descs and any(m != LIVE_FULL for m in masks):
                n = self.n_exec
                # single X (x = (q,), z = ()) and single Y (x = z = (q,)) strings; Z-only strings; everything else stays
                xw = [(k, n - 1 - t.x[0], float(weights[k]), 0 if not t.z else 1) for k, t in enumerate(cm.all_terms)
                      if len(t.x) == 1 and (not t.z or tuple(t.z) == tuple(t.x))]
                dw = [(k, sum(1 << (n - 1 - q) for q in t.z), float(weights[k])) for k, t in enumerate(cm.all_terms)
                      if not t.x and t.z]
                # strings with exactly two X / Y factors and no further Z (XX, YY, XY, YX): a qubit in ``x`` and ``z`` is a Y
                pw = [(k, n - 1 - t.x[0], n - 1 - t.x[1], float(weights[k]), int(t.x[0] in t.z), int(t.x[1] in t.z))
                      for k, t in enumerate(cm.all_terms) if len(t.x) == 2 and set(t.z) <= set(t.x)]
                res = fold_plan_host(self._exec_gates, n, adj0, self.nparams, xw, dw, len(cm.all_terms), pw)
                if res is not None:
                    ap, done, lam_zero = res
                    skip = frozenset(done)
                    # what is left (strings with two or more X / Y factors ...) still goes through the tile passes
                    if lam_zero or (cm._tiled_plan(skip) is not None and cm._tiled_plan() is not None
                                    and len(cm._tiled_plan(skip)) < len(cm._tiled_plan())):
                        dev = self.device
                        cache[key] = {"skip": skip, "lam_zero": lam_zero, "masks": masks, "fracs": fracs, "adj": {
                            "plan": ap, "cfg": adj0["cfg"], "descs": [_dev(d, dev) for d in ap.descs],
                            "ctab": _dev(ap.ctab, dev, self.rdtype), "ginfo": _dev(ap.ginfo, dev),
                            "cpool": _dev(ap.cpool if ap.cpool.size else np.zeros(1), dev),
                            "gparam": _dev(ap.gslot_param, dev), "gfactor": _dev(ap.gslot_factor, dev),
                            "nslots": len(ap.gslot_param)}}
        rec = cache[key]
        if rec is None:
            return None
        # only the generated kernels know OP_XFOLD: the sweep the traced pipeline runs (last pass without write-back) must be
        # loaded -- PassSet.get() also counts this call towards the plan's hotness, so a missing kernel gets compiled
        adj = rec["adj"]
        if adj.get("spec_nostore") is None:
            nd = [np.asarray(d) for d in adj["plan"].descs]
            nd[-1] = nd[-1].copy()
            nd[-1][6] |= P.FLAG_NOSTORE
            adj["spec_nostore"] = S.PassSet("adjoint", nd, self.n_exec, S.adjoint_opts(adj["cfg"]))
        if any(k is None for k in adj["spec_nostore"].get()):
            return None
        return rec

    def vjp(self, params, psi, g, chunk_bytes=48 << 30, inputs=None, want_input_grad=False, consume=False,
            from_zero=False, fold=None):
        """dL/dparams = Re <g | d psi / d params> for every batch row, by the adjoint sweep.
        params [B, P] real, psi / g [B, 2^n_exec] complex (psi = the forward output).  The sweep
        works on copies (psi is un-computed in place), processed in batch chunks to bound memory.
        ``want_input_grad``: also return the cotangent of the input state, U^dagger g = lambda after the whole
        sweep ([B, 2^n_exec]).  ``inputs``: the forward input state (needed only with non-unitary gates, whose
        segments restart from recomputed checkpoints).  ``from_zero``: psi is the state of this plan run from |0...0>
        (enables the live-tile sweep, ``_adjoint_from_zero``)."""
        import torch

        if self.nonunitary:
            return self._vjp_segmented(params, g, inputs, want_input_grad)
        lam_out = torch.empty_like(g) if want_input_grad else None
        adj = self._adjoint(full=want_input_grad)
        live, lfracs = None, None
        if fold is not None:
            # ``fold`` (a record of fold_setup): g lacks the folded terms of the cotangent, the sweep's first pass adds them;
            # returns (dL/dparams, energy of the folded terms [B])
            assert from_zero and not want_input_grad and getattr(self, "_keep_uncomputed", None) is None
            adj, live, lfracs = fold["adj"], fold["masks"], fold["fracs"]
        elif (from_zero and SPARSE_START and not want_input_grad and self.dtypestr == "complex64"
                and getattr(self, "_keep_uncomputed", None) is None):
            adj0, masks, fracs = self._adjoint_from_zero()
            if adj0["cfg"].gen == 2 and any(m != LIVE_FULL for m in masks):
                adj, live, lfracs = adj0, masks, fracs
        lib = self._lib
        B = params.shape[0]
        nel = 2**self.n_exec
        out = torch.zeros(B, max(self.nparams, 1) + (1 if fold is not None else 0), dtype=torch.float64, device=self.device)
        if (adj["nslots"] == 0 or self.nparams == 0) and not want_input_grad and fold is None:
            return out[:, : self.nparams].to(self.rdtype)
        if params is None or params.shape[-1] == 0:
            params = torch.zeros(B, 1, dtype=self.rdtype, device=self.device)
        params = params.to(device=self.device, dtype=self.rdtype).contiguous()
        item = 8 if self.dtypestr == "complex64" else 16
        cb = max(1, int(chunk_bytes // (2 * nel * item)))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        cfg = adj["cfg"]
        for b0 in range(0, B, cb):
            b1 = min(B, b0 + cb)
            nb = b1 - b0
            if consume and psi.dtype == self.tdtype and g.dtype == self.tdtype and psi.is_contiguous() and g.is_contiguous():
                # the caller owns psi and g and does not need them afterwards (the traced value_and_grad pipeline):
                # the sweep un-computes / propagates them in place, two state-sized copies per chunk less
                a, lam = psi[b0:b1], g[b0:b1]
            else:
                a = psi[b0:b1].to(self.tdtype).clone().contiguous()
                lam = g[b0:b1].to(self.tdtype).clone().contiguous()
            p = params[b0:b1]
            ptab = torch.empty(nb, max(1, adj["plan"].ptab_size), dtype=self.rdtype, device=self.device)
            _lib.check(
                lib.tcmi_build_adjoint_tables(
                    adj["ginfo"].data_ptr(), int(adj["ginfo"].shape[0]), adj["cpool"].data_ptr(),
                    p.data_ptr(), p.stride(0), ptab.data_ptr(), ptab.stride(0), nb, self.code, stream),
                "tcmi_build_adjoint_tables",
            )
            gout = torch.zeros(nb, ATOMIC_COPIES, max(1, adj["nslots"]), dtype=torch.float64, device=self.device)
            descs = adj["descs"]
            # nobody reads psi / lambda after the sweep (no input-state cotangent): the last pass of the packed kernel
            # keeps its tile to itself (FLAG_NOSTORE: two of its four state transfers less)
            nostore = cfg.gen == 2 and not want_input_grad and getattr(self, "_keep_uncomputed", None) is None and descs
            if nostore:
                if adj.get("last_nostore") is None:
                    w = descs[-1].clone()
                    w[6] = w[6] | P.FLAG_NOSTORE
                    adj["last_nostore"] = w
                descs = list(descs[:-1]) + [adj["last_nostore"]]
            spec = [None] * len(descs)
            if cfg.gen == 2 and self.dtypestr == "complex64":
                # plan-specialised straight-line kernels where they exist (tcmi/specialize.py); the interpreter otherwise
                skey = "spec_nostore" if nostore else "spec"
                if adj.get(skey) is None:
                    nd = [np.asarray(d) for d in adj["plan"].descs]
                    if nostore:
                        nd[-1] = nd[-1].copy()
                        nd[-1][6] |= P.FLAG_NOSTORE
                    adj[skey] = S.PassSet("adjoint", nd, self.n_exec, S.adjoint_opts(cfg))
                spec = adj[skey].get()
            if fold is not None and any(k is None for k in spec):
                raise RuntimeError("the folded reverse sweep needs its plan-specialised kernels (fold_setup checks this)")
            units = len(descs) * 4.0 - (2.0 if nostore else 0.0)
            if live is not None:       # algorithmic bytes of the live tiles only (interpreted passes move every tile)
                units = sum((4.0 if (i < len(descs) - 1 or not nostore) else 2.0) * (lfracs[i] if k is not None else 1.0)
                            for i, k in enumerate(spec))
                if fold is not None and fold.get("lam_zero"):
                    units -= lfracs[0]          # the first pass does not read lambda
            _log_valu("adjoint", spec, lfracs if live is not None else None, self.n_exec, nb)
            tm = _timed("adjoint", len(descs), units * nb * nel * item)
            tm.__enter__()
            for ip, (d, k) in enumerate(zip(descs, spec)):
                if PASS_EVENTS is not None:
                    pe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    if ip:
                        PASS_EVENTS[-1][2].record()
                    pe[0].record()
                    PASS_EVENTS.append((ip, pe[0], pe[1]))
                if k is not None:
                    _lib.check(
                        lib.tcmi_spec_run_adjoint_pass(
                            k.handle, a.data_ptr(), lam.data_ptr(), nel, nb, self.n_exec, cfg.T, cfg.LT,
                            adj["ctab"].data_ptr(), ptab.data_ptr(), ptab.stride(0), gout.data_ptr(), gout.stride(0),
                            ATOMIC_COPIES, gout.stride(1), LIVE_FULL if live is None else live[ip], stream),
                        "tcmi_spec_run_adjoint_pass",
                    )
                    continue
                _lib.check(
                    lib.tcmi_run_adjoint_pass(
                        a.data_ptr(), lam.data_ptr(), nel, nb, self.n_exec, cfg.R, cfg.LT, d.data_ptr(),
                        adj["ctab"].data_ptr(), ptab.data_ptr(), ptab.stride(0), gout.data_ptr(),
                        gout.stride(0), ATOMIC_COPIES, gout.stride(1), self.code, int(cfg.gen == 2), stream),
                    "tcmi_run_adjoint_pass",
                )
            if PASS_EVENTS is not None and descs:
                PASS_EVENTS[-1][2].record()
            tm.__exit__(None, None, None)
            if adj["nslots"]:
                out[b0:b1].index_add_(1, adj["gparam"], gout.sum(1) * adj["gfactor"])
            if want_input_grad:
                lam_out[b0:b1] = lam
            if getattr(self, "_keep_uncomputed", None) is not None:
                self._keep_uncomputed.append(a)
        gp = out[:, : self.nparams].to(self.rdtype)
        if fold is not None:
            return gp, out[:, self.nparams]
        return (gp, lam_out) if want_input_grad else gp

    # ---- reverse mode through non-unitary gates ----------------------------------------------------------
    def _segments(self):
        """The gate list cut at every non-unitary gate: [("U", CompiledCircuit of a unitary run) | ("N", forward
        CompiledCircuit of the gate, CompiledCircuit of its adjoint matrix)]."""
        if getattr(self, "_segs", None) is None:
            segs, run = [], []
            bad = set(self.nonunitary)

            def sub(gs):
                return CompiledCircuit(self.n_exec, gs, self.nparams, self.dtypestr, self._opts, self.device)

            for i, g in enumerate(self._exec_gates):
                if i not in bad:
                    run.append(g)
                    continue
                if run:
                    segs.append(("U", sub(run)))
                    run = []
                if g.param is not None or g.select is not None:
                    raise NotImplementedError(
                        "Backend 'hip' has not implemented gradients through a parametrised non-unitary gate.")
                m = np.asarray(g.c0, dtype=np.complex128)
                d = int(round(np.sqrt(m.size)))
                gd = P.GateRec(g.qubits, c0=m.reshape(d, d).conj().T.copy(), name=g.name + "^dagger")
                segs.append(("N", sub([g]), sub([gd])))
            if run:
                segs.append(("U", sub(run)))
            self._segs = segs
        return self._segs

    def _vjp_segmented(self, params, g, inputs, want_input_grad):
        """Adjoint sweep with checkpoints: the state entering each non-unitary gate M is recomputed by a forward
        run and kept; going back, unitary runs are un-computed as usual, at M the cotangent becomes M^dagger lambda
        and psi restarts from the checkpoint."""
        import torch

        segs = self._segments()
        B = g.shape[0]
        nel = 2**self.n_exec
        if params is None or self.nparams == 0:
            params = torch.zeros(B, max(1, self.nparams), dtype=self.rdtype, device=self.device)
        params = params.to(device=self.device, dtype=self.rdtype).contiguous()
        if inputs is None:
            x = torch.zeros(B, nel, dtype=self.tdtype, device=self.device)
            x[:, 0] = 1
        else:
            inp = inputs.reshape(-1, inputs.shape[-1]).to(device=self.device, dtype=self.tdtype)
            x = torch.zeros(B, nel, dtype=self.tdtype, device=self.device)
            x[:, : inp.shape[-1]] = inp
        ends = []   # state after each segment (psi to un-compute from) / before each non-unitary gate (checkpoint)
        for seg in segs:
            if seg[0] == "N":
                ends.append(x)
            x = seg[1].state(params, inputs=x, full=True).clone()
            if seg[0] == "U":
                ends.append(x)
        lam = g.to(self.tdtype).clone()
        grad = torch.zeros(B, max(self.nparams, 1), dtype=torch.float64, device=self.device)
        for seg, x_end in zip(reversed(segs), reversed(ends)):
            if seg[0] == "U":
                gp, lam = seg[1].vjp(params, x_end, lam, want_input_grad=True)
                grad[:, : self.nparams] += gp.to(torch.float64)
            else:
                lam = seg[2].state(params, inputs=lam, full=True).clone()
        gp = grad[:, : self.nparams].to(self.rdtype)
        return (gp, lam) if want_input_grad else gp


def pick_small_tile_variant(n_exec: int, dtypestr: str) -> P.PlanConfig:
    """Tile of the first-generation measurement / adjoint kernels: a smaller register tile than the gate passes
    (|a|^2 and cross products, or a second vector, live next to the amplitudes)."""
    c64 = dtypestr == "complex64"
    for R, LT in ([(4, 8), (2, 6)] if c64 else [(3, 8), (2, 6)]):
        if R + LT <= n_exec:
            return P.PlanConfig(R=R, LT=LT, lowbits=min(5, R + LT), vec=2 if c64 else 1)
    raise ValueError("no small-tile variant fits")


def pick_measure_variant(n_exec: int, dtypestr: str) -> P.PlanConfig:
    """Tile of the measurement passes: complex64 states of >= 13 qubits go to the packed kernel
    (csrc/tcmi_measure2.hip, TCMI_OP_EXPECT2 descriptors), everything else to the first-generation kernel."""
    c64 = dtypestr == "complex64"
    ov = knob("meas_cfg")     # experiment switch: "R,LT,lowbits" of the first-generation kernel
    if ov and c64:
        R, LT, lb = (int(x) for x in ov.split(","))
        if R + LT <= n_exec and (R, LT) != (5, 8):
            return P.PlanConfig(R=R, LT=LT, lowbits=lb, vec=2)
    if c64 and n_exec >= 13 and not knob("vm1"):
        # packed kernel csrc/tcmi_measure2.hip (TCMI_OP_EXPECT2 descriptors): 32 amplitudes per thread, 13 tile bits
        return P.PlanConfig(R=5, LT=8, lowbits=int(knob("meas_lowbits", "5")), vec=2, gen=2)
    return pick_small_tile_variant(n_exec, dtypestr)


def pick_adjoint_variant(n_exec: int, dtypestr: str, gates) -> P.PlanConfig:
    """Tile of the adjoint sweep (two vectors live in registers).  complex64 circuits of one-qubit gates and
    diagonals on >= 13 qubits run on the packed-f32 kernel (csrc/tcmi_adjoint2.hip: R = 4, 256 threads, 12 tile
    bits, four workgroups per CU -- one more pass than the 13-bit tile at n = 28, but the load / compute / store
    phases of four workgroups overlap better than those of two: 44.1 vs 47.6 ms per sample); dense two-qubit
    gates and small circuits keep the first-generation kernel."""
    # dense two-qubit gates, and diagonal terms of three or more qubits (their emission may need CNOT register moves,
    # plan.emit_diag): both are G2 ops, which only the first-generation sweep executes
    dense2 = any(((not g.is_diag) and len(g.qubits) > 1) or (g.is_diag and any(len(t.qubits) > 2 for t in g.diag))
                 for g in gates)
    if dtypestr == "complex64" and n_exec >= 13 and not dense2 and not knob("vm1"):
        # two-shear rotations in the reverse sweep: implemented and tested (TCMI_SHEAR2_BW=1), off by default -- the sweep
        # needs a second phase table for lambda (reciprocal real factors), whose scalar loads cost what the dropped
        # shears save (n = 28 d = 12: 28.8 vs 28.7 ms per pass)
        # ... in the INTERPRETING kernel.  The plan-specialised sweep prefetches its tables a segment ahead, so there the
        # dropped shears are pure gain (16 of 64 packed instructions per eligible gate): on whenever specialisation is
        sh2 = knob("shear2_bw", "1" if (S.mode() != "0" and n_exec >= S.MIN_N) else "0") == "1"
        if knob("adj_tile"):   # experiment switch "R,LT": 2^R amplitudes of psi and of lambda per thread,
            r_, lt_ = (int(x) for x in knob("adj_tile").split(","))    # workgroups of 2^LT threads (6: one wave, no barriers)
            return P.PlanConfig(R=r_, LT=lt_, lowbits=5, vec=2, gen=2, shear2=sh2)
        return P.PlanConfig(R=4, LT=8, lowbits=5, vec=2, gen=2, shear2=sh2)
    return pick_small_tile_variant(n_exec, dtypestr)


class CompiledMeasure:
    """A set of Pauli strings lowered to read-only measurement passes (fused K4 kernel)."""

    def __init__(self, n: int, n_exec: int, strings, dtypestr: str, device=None):
        import torch

        self.n, self.n_exec, self.dtypestr = n, n_exec, dtypestr
        pad = n_exec - n
        terms = []
        for ps in strings:
            t = P.pauli_term_from_string(ps)
            terms.append(P.PauliTerm(tuple(q + pad for q in t.x), tuple(q + pad for q in t.z)))
        self.cfg = pick_measure_variant(n_exec, dtypestr)
        # strings with <= 2 X/Y factors go through the fused measurement passes; heavier ones are
        # evaluated as <psi|(P psi)> (tcmi_apply_pauli_sum + tcmi_vdot), one state-sized temporary each
        self.light = [k for k, t in enumerate(terms) if len(t.x) <= 2]
        self.heavy = [k for k, t in enumerate(terms) if len(t.x) > 2]
        self.all_terms = list(terms)
        self.plan = P.compile_measure_plan([terms[k] for k in self.light], n_exec, self.cfg)
        self.nterms = len(terms)
        self.code = _lib.TCMI_C64 if dtypestr == "complex64" else _lib.TCMI_C128
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self._lib = _lib.lib()
        self.descs = [_dev(d, self.device) for d in self.plan.descs]
        rdt = torch.float32 if dtypestr == "complex64" else torch.float64
        self.dummy = _dev(np.zeros(8), self.device, rdt)
        self.phase = _dev(np.array([(1j) ** terms[k].ny for k in self.light], dtype=np.complex128), self.device)
        self._heavy_rows = []
        for k in self.heavy:
            t = terms[k]
            xm = sum(1 << (n_exec - 1 - q) for q in t.x)
            zm = sum(1 << (n_exec - 1 - q) for q in t.z)
            row = np.array([[xm, zm, t.ny]], dtype=np.int64).astype(np.uint32).view(np.int32)
            self._heavy_rows.append(_dev(row.reshape(1, 3), self.device))

    def run(self, state):
        """state: complex tensor [B, 2^n_exec] (full executor buffer).  Returns complex128 [B, nterms]
        with <psi|P_t|psi> for every term."""
        import torch

        B = state.shape[0]
        assert state.shape[1] == 2**self.n_exec and state.is_contiguous()
        nl = len(self.light)
        out = torch.zeros(B, ATOMIC_COPIES, 2 * max(nl, 1), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        item = 8 if self.dtypestr == "complex64" else 16
        with _timed("measure", len(self.descs), len(self.descs) * 1.0 * B * state.shape[1] * item):
            for d in self.descs:
                _lib.check(
                    self._lib.tcmi_run_pass(
                        state.data_ptr(), state.shape[1], B, self.n_exec, self.cfg.R, self.cfg.LT,
                        d.data_ptr(), self.dummy.data_ptr(), self.dummy.data_ptr(), 0, out.data_ptr(),
                        out.stride(0), ATOMIC_COPIES, out.stride(1), self.code, stream,
                    ),
                    "tcmi_run_pass(measure)",
                )
        vals_l = torch.view_as_complex(out.sum(1).reshape(B, max(nl, 1), 2))[:, :nl] * self.phase
        if not self.heavy:
            return vals_l
        vals = torch.zeros(B, self.nterms, dtype=torch.complex128, device=self.device)
        if nl:
            vals[:, self.light] = vals_l
        ones = torch.ones(B, 1, dtype=torch.float64, device=self.device)
        tmp = torch.empty_like(state)
        for k, row in zip(self.heavy, self._heavy_rows):
            _lib.check(
                self._lib.tcmi_apply_pauli_sum(state.data_ptr(), tmp.data_ptr(), state.shape[1], B, self.n_exec,
                                               row.data_ptr(), 1, ones.data_ptr(), ones.stride(0), self.code, stream),
                "tcmi_apply_pauli_sum")
            acc = torch.zeros(B, ATOMIC_COPIES, 2, dtype=torch.float64, device=self.device)
            _lib.check(
                self._lib.tcmi_vdot(state.data_ptr(), tmp.data_ptr(), acc.data_ptr(), state.shape[1], B, self.n_exec,
                                    ATOMIC_COPIES, acc.stride(0), self.code, stream),
                "tcmi_vdot")
            vals[:, k] = torch.view_as_complex(acc.sum(1))
        return vals

    def _tiled_plan(self, skip=frozenset()):
        """Passes of ``tcmi_apply_pauli_sum_tiled`` for this term list (``plan_pauli_passes``), or None when the flat
        gather kernel is the better (or the only) choice.  ``skip``: indices of terms that are left out (they reach the
        cotangent another way: CompiledCircuit.fold_setup)."""
        cache = self.__dict__.setdefault("_tiled_cache", {})
        if skip not in cache:
            cache[skip] = None
            if knob("pauli_tiled", "1") != "0" and self.n_exec <= 32:
                n = self.n_exec
                rows = []
                for k, t in enumerate(self.all_terms):
                    if k in skip:
                        continue
                    xm = sum(1 << (n - 1 - q) for q in t.x)
                    zm = sum(1 << (n - 1 - q) for q in t.z)
                    rows.append((xm, zm, t.ny, k))
                T = int(self._lib.tcmi_pauli_sum_tile_bits(self.code))
                passes = plan_pauli_passes(n, rows, T) if (n >= T and rows) else None
                if passes is not None:
                    cache[skip] = [
                        dict(tilepos=_dev(np.asarray(ps["tilepos"], dtype=np.int32), self.device),
                             terms=_dev(np.asarray(ps["rows"], dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 4),
                                        self.device),
                             order=_dev(np.asarray(ps["order"], dtype=np.int64), self.device), n=len(ps["order"]),
                             ndiag=int(ps["ndiag"]))
                        for ps in passes]
        return cache[skip]

    def apply_sum(self, state, gvals, want_dot=False, skip=frozenset()):
        """Cotangent of the state for L = f(<psi|P_t|psi>): 2 * sum_t Re(g_t) P_t |psi>.  state [B, 2^n_exec], gvals
        [B, nterms] complex.  Tile passes (``tcmi_apply_pauli_sum_tiled``) when every X mask fits a tile, else the flat
        gather kernel (``tcmi_apply_pauli_sum``).  ``want_dot``: also return Re <psi|lambda> per batch element
        (float64 [B]) -- for g_t = w_t that is 2 sum_t w_t <P_t>, i.e. the energy comes with its cotangent and the
        measurement passes are not needed."""
        import torch

        tiled = self._tiled_plan(skip)
        if skip and tiled is None:
            raise RuntimeError("apply_sum(skip=...) needs the tile passes (fold_setup checks _tiled_plan(skip) first)")
        if tiled is not None:
            B = state.shape[0]
            wall = 2.0 * gvals.real.to(torch.float64)
            out = torch.empty_like(state)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            item = 8 if self.dtypestr == "complex64" else 16
            dots = torch.zeros(B, ATOMIC_COPIES, dtype=torch.float64, device=self.device) if want_dot else None
            nbytes = sum((2.0 if i == 0 else 3.0) for i in range(len(tiled))) * B * state.shape[1] * item
            with _timed("pauli_sum", len(tiled), nbytes):
                for i, ps in enumerate(tiled):
                    w = wall[:, ps["order"]].contiguous()
                    _lib.check(
                        self._lib.tcmi_apply_pauli_sum_tiled(
                            state.data_ptr(), out.data_ptr(), state.shape[1], B, self.n_exec, ps["tilepos"].data_ptr(),
                            ps["terms"].data_ptr(), ps["n"], ps["ndiag"], w.data_ptr(), w.stride(0), int(i > 0),
                            dots.data_ptr() if want_dot else None, dots.stride(0) if want_dot else 0, ATOMIC_COPIES,
                            self.code, stream),
                        "tcmi_apply_pauli_sum_tiled")
            return (out, dots.sum(1)) if want_dot else out
        if getattr(self, "_sum_terms", None) is None:
            n = self.n_exec
            rows = []
            for k, t in enumerate(self.all_terms):
                xm = 0
                for q in t.x:
                    xm |= 1 << (n - 1 - q)
                zm = 0
                for q in t.z:
                    zm |= 1 << (n - 1 - q)
                rows.append((xm, zm, t.ny, k))
            rows.sort(key=lambda r: r[0])
            arr = np.array([[r[0], r[1], r[2]] for r in rows], dtype=np.int64).astype(np.uint32).view(np.int32)
            self._sum_terms = _dev(arr.reshape(-1, 3), self.device)
            self._sum_order = _dev(np.array([r[3] for r in rows], dtype=np.int64), self.device)
        B = state.shape[0]
        w = (2.0 * gvals.real.to(torch.float64))[:, self._sum_order].contiguous()
        out = torch.empty_like(state)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        item = 8 if self.dtypestr == "complex64" else 16
        with _timed("pauli_sum", 1, 2.0 * B * state.shape[1] * item):
            _lib.check(
                self._lib.tcmi_apply_pauli_sum(
                    state.data_ptr(), out.data_ptr(), state.shape[1], B, self.n_exec,
                    self._sum_terms.data_ptr(), self.nterms, w.data_ptr(), w.stride(0), self.code, stream),
                "tcmi_apply_pauli_sum",
            )
        if want_dot:
            acc = torch.zeros(B, ATOMIC_COPIES, 2, dtype=torch.float64, device=self.device)
            _lib.check(self._lib.tcmi_vdot(state.data_ptr(), out.data_ptr(), acc.data_ptr(), state.shape[1], B, self.n_exec,
                                           ATOMIC_COPIES, acc.stride(0), self.code, stream), "tcmi_vdot")
            return out, acc.sum(1)[:, 0]
        return out


def plan_pauli_passes(n: int, rows, T: int, lowbits: int = 4, max_passes: int = 6):
    """Tile passes for ``(sum_t w_t P_t)|psi>``: every pass owns tiles over the ``lowbits`` lowest index bits (whole
    128-byte segments) plus ``T - lowbits`` further bits, and applies the terms whose X mask lies inside its tile bits;
    every term is applied in exactly one pass, the diagonal ones in the first.  ``rows`` = (xmask, zmask, nY, term
    index) over physical bits.  Greedy cover: a pass is seeded with the remaining mask that reaches lowest and takes
    every other mask that still fits, fewest new bits first.  Returns a list of {tilepos, rows (xmask in tile-index
    space, zmask, nY | emask << 8, parity(x & z); the Z-only rows first, sorted by emask), ndiag, order} or None when a mask does not fit a tile, more than ``max_passes`` passes
    are needed, or the flat gather kernel moves fewer bytes (1 read per distinct mask outside its 2^T-element window
    + 1 read + 1 write, against 2 transfers for the first pass and 3 for each further one)."""
    free = T - lowbits
    lowmask = (1 << lowbits) - 1
    masks = sorted({r[0] for r in rows})
    high = {m: m & ~lowmask for m in masks}
    if any(bin(h).count("1") > free for h in high.values()):
        return None
    todo = [m for m in masks if high[m]]
    groups = []
    while todo:
        g = 0
        todo.sort(key=lambda m: ((high[m] & -high[m]), m))
        picked = []
        rest = list(todo)
        while rest:
            best = None
            for m in rest:
                new = bin(high[m] & ~g).count("1")
                if bin(g | high[m]).count("1") <= free and (best is None or (new, m) < best[0]):
                    best = ((new, m), m)
            if best is None:
                break
            g |= high[best[1]]
            picked.append(best[1])
            rest.remove(best[1])
        groups.append((g, picked))
        todo = [m for m in todo if m not in picked]
        if len(groups) > max_passes:
            return None
    if not groups:
        groups = [(0, [])]
    local = [m for m in masks if not high[m]]          # inside the low bits: first pass
    groups[0] = (groups[0][0], local + groups[0][1])
    flat_window = (1 << min(n, 12)) - 1
    flat_cost = 2 + sum(1 for m in masks if m & ~flat_window)
    if 2 + 3 * (len(groups) - 1) >= flat_cost and flat_cost <= 6:
        return None
    passes = []
    for g, picked in groups:
        bits = [b for b in range(lowbits)] + [b for b in range(lowbits, n) if (g >> b) & 1]
        fill = [b for b in range(lowbits, n) if not (g >> b) & 1]
        while len(bits) < T:                             # unused tile bits: the next free index bits
            bits.append(fill.pop(0))
        bits.sort()
        pos = {b: j for j, b in enumerate(bits)}
        ebits = [bits[0]] + bits[9:]                     # element bits of a thread: pair member, pair counter

        def emask(zm):
            return sum(((zm >> b) & 1) << j for j, b in enumerate(ebits))

        sel = [r for r in rows if r[0] in picked]
        diag = sorted((r for r in sel if r[0] == 0), key=lambda r: (emask(r[1]), r[3]))
        offd = sorted((r for r in sel if r[0] != 0), key=lambda r: (_to_local(r[0], pos), r[3]))
        passes.append({
            "tilepos": bits, "ndiag": len(diag),
            "rows": [(_to_local(r[0], pos), r[1], r[2] | (emask(r[1]) << 8), bin(r[0] & r[1]).count("1") & 1)
                     for r in diag + offd],
            "order": [r[3] for r in diag + offd],
        })
    return passes


def _to_local(xm: int, pos) -> int:
    out, b = 0, 0
    while xm >> b:
        if (xm >> b) & 1:
            out |= 1 << pos[b]
        b += 1
    return out


_MCACHE: "OrderedDict[Tuple, CompiledMeasure]" = OrderedDict()


def get_measure(n, n_exec, strings, dtypestr) -> CompiledMeasure:
    import torch

    key = (n, n_exec, dtypestr, tuple(tuple(int(p) for p in ps) for ps in strings),
           torch.cuda.current_device() if torch.cuda.is_available() else -1)
    m = _MCACHE.get(key)
    if m is None:
        m = CompiledMeasure(n, n_exec, strings, dtypestr)
        _MCACHE[key] = m
        while len(_MCACHE) > _CACHE_MAX:
            _MCACHE.popitem(last=False)
    else:
        _MCACHE.move_to_end(key)
    return m


# ---- cost model (microseconds per 2^24 amplitudes, fitted on MI355X: profiles/r01b) -------------------
VM_COST = {"pass": 19.5, "g1": 2.25, "g2": 9.0, "diag": 11.5, "exchange": 10.75}
# packed complex64 kernel (csrc/tcmi_vm2.hip), fitted to the per-pass times of n = 28, d = 12 (scripts/gpu_pass_breakdown.py):
# a pass costs max(HBM floor, fixed + gates + phase tables + exchanges), microseconds per 2^24 amplitudes
VM2_COST = {"floor": 52.0, "fixed": 16.2, "g1": 0.95, "g2": 3.8, "table": 1.1, "diag": 5.0, "exchange": 3.3}
GEMM_TFLOPS = 150.0  # tcmi_cgemm, batched cut join (3-product kernel, algorithmic flops: 154 measured, profiles/r02d)
GEMM_SPLIT_TFLOPS = 260.0   # tcmi_cgemm_split on the same shape (8 flops per complex MAC / measured time, profiles/r04f)
# "split" (default): complex64 joins on the f16 matrix pipe with two-piece operands when the cut bounds its half-circuit
# states (tcmi_cgemm_split_f16), else on the bf16 pipe with three-piece operands; "bf16": the three-piece kernel always;
# "f32": the exact-f32 MFMA kernel (tcmi_cgemm) only
JOIN_GEMM = os.environ.get("TCMI_JOIN_GEMM", "split")


# packed adjoint kernel (csrc/tcmi_adjoint2.hip), fitted to the per-pass times of n = 28, d = 12 (scripts/gpu_adj_one.py
# --passes): max(HBM floor, gates + diagonal flushes + exchanges), microseconds per 2^24 amplitudes
ADJ2_COST = {"floor": 106.0, "g1": 4.2, "flush": 8.1, "exchange": 6.3}


def adj_cost_us(ap: "P.AdjointPlan", fracs=None) -> float:
    """Estimated time of a packed (gen 2) adjoint plan for one state; None for other plans.  ``fracs``: fraction of live
    tiles per pass (``live_masks``)."""
    if ap.cfg.gen < 2:
        return None
    R = ap.cfg.R
    t = 0.0
    ipass = -1
    for desc in ap.descs:
        ipass += 1
        d = np.asarray(desc).view(np.uint32).astype(np.int64)
        pc = P.HDR_WORDS
        tp = ADJ2_COST["exchange"] * (int(d[5]) - 1)
        for _ in range(int(d[5])):
            rr = d[pc: pc + P.RR_WORDS]
            q = pc + P.RR_WORDS
            end = q + int(rr[1])
            for _o in range(int(rr[0])):
                op = int(d[q])
                if op == P.OP_G1M:
                    tp += ADJ2_COST["g1"] * bin(int(d[q + 1]) & 0xFF).count("1")
                    q += 5 + R
                elif op == P.OP_DIAGF:
                    tp += ADJ2_COST["flush"]
                    q += 9 + (1 << R) + 4 * int(d[q + 3]) + 2 * int(d[q + 4])
                else:
                    return None
            pc = end
        t += max(ADJ2_COST["floor"], tp) * (1.0 if fracs is None else fracs[ipass])
    return t * (2.0 ** ap.n) / 2.0**24


def vm_cost_us(plan: "P.CompiledPlan", fracs=None) -> float:
    """Estimated time of a tile-VM plan for one state (linear op-count model).  ``fracs``: fraction of live tiles per pass
    (``live_masks``; packed plans only)."""
    t = 0.0
    gen2 = getattr(plan.cfg, "gen", 1) >= 2 if hasattr(plan, "cfg") else False
    if gen2:
        for ipass, desc in enumerate(plan.descs):
            d = np.asarray(desc).view(np.uint32).astype(np.int64)
            pc = P.HDR_WORDS
            tp = VM2_COST["fixed"] + VM2_COST["exchange"] * (int(d[5]) - 1)
            for _ in range(int(d[5])):
                rr = d[pc: pc + P.RR_WORDS]
                q = pc + P.RR_WORDS
                for _o in range(int(rr[0])):
                    op = int(d[q])
                    if op == P.OP_G1M:
                        tp += VM2_COST["g1"] * bin(int(d[q + 1]) & 0xFF).count("1")
                        q += 3
                    elif op == P.OP_G2:
                        tp += VM2_COST["g2"] if (int(d[q + 1]) >> 8) == 0 else VM2_COST["g1"]
                        q += 4
                    elif op == P.OP_DIAG:
                        nA, nB, nC = (int(x) for x in d[q + 1: q + 4])
                        tp += VM2_COST["diag"]
                        q += 5 + nA + 2 * nB + nC
                    elif op in (P.OP_DIAGC, P.OP_DIAGB, P.OP_DIAGB2, P.OP_DIAGCW):
                        tp += VM2_COST["table"]
                        q += {P.OP_DIAGC: 2, P.OP_DIAGB: 4, P.OP_DIAGB2: 5, P.OP_DIAGCW: 6}[op]
                    else:
                        raise ValueError(op)
                pc = q
            t += max(VM2_COST["floor"], tp) * (1.0 if fracs is None else fracs[ipass])
        return t * (2.0 ** plan.n) / 2.0**24
    for pp, desc in zip(plan.passes, plan.descs):
        d = np.asarray(desc).view(np.uint32).astype(np.int64)
        pc = P.HDR_WORDS
        t += VM_COST["pass"] + VM_COST["exchange"] * (int(d[5]) - 1)
        for _ in range(int(d[5])):
            rr = d[pc: pc + P.RR_WORDS]
            q = pc + P.RR_WORDS
            for _o in range(int(rr[0])):
                op = int(d[q])
                if op == P.OP_G1M:
                    t += VM_COST["g1"] * bin(int(d[q + 1]) & 0xFF).count("1")
                    q += 3
                elif op == P.OP_G2:
                    t += VM_COST["g2"] if (int(d[q + 1]) >> 8) == 0 else VM_COST["g1"]
                    q += 4
                elif op == P.OP_DIAG:
                    nA, nB, nC = (int(x) for x in d[q + 1: q + 4])
                    t += VM_COST["diag"]
                    q += 5 + nA + 2 * nB + nC
                elif op == P.OP_DIAGC:
                    t += VM_COST["g1"]
                    q += 2
                elif op == P.OP_DIAGB:
                    t += VM_COST["g1"]
                    q += 4
                elif op == P.OP_DIAGB2:
                    t += VM_COST["g1"]
                    q += 5
                elif op == P.OP_DIAGCW:
                    t += VM_COST["g1"]
                    q += 6
                else:
                    raise ValueError(op)
            pc = q
    return t * (2.0 ** plan.n) / 2.0**24


class _HalfBatch:
    """The K = prod r_k selector variants of one half-circuit.  The variants share prefixes: with the
    bond digits ordered like the crossing gates, every state whose first s digits agree is identical up
    to the (s+1)-th selector gate.  Two levels: the prefix (gates before the (s+1)-th selector) runs once
    per K_s = prod_{k<s} r_k digit prefixes, its states are replicated K / K_s times, and the suffix
    runs on the replicas — K_s |prefix| + K |suffix| gate applications instead of K (|prefix| + |suffix|)."""

    @staticmethod
    def split_point(gates, nparams, nb, radices):
        """(s, position of the (s+1)-th selector gate) of the two-level split, s = 0: one level.  Host work only."""
        K = int(np.prod(radices)) if radices else 1
        sel_pos = [i for i, g in enumerate(gates) if g.select is not None]
        ordered = [gates[i].param.index - nparams for i in sel_pos] == list(range(len(sel_pos)))
        best = (K * len(gates), 0)
        if ordered and len(sel_pos) == nb:
            for s_ in range(1, nb):
                cut = sel_pos[s_]
                ks = int(np.prod(radices[:s_]))
                cost = ks * cut + K * (len(gates) - cut) + 0.02 * K * len(gates)  # + extra launches / copy
                if cost < best[0]:
                    best = (cost, s_)
        return best[1], (sel_pos[best[1]] if best[1] else None), K

    def __init__(self, nq, gates, nparams, nb, radices, dtypestr, opts):
        self.nq, self.radices = nq, list(radices)
        self.s, cut, K = self.split_point(gates, nparams, nb, radices)
        self.K = K
        if self.s == 0:
            self.single = CompiledCircuit(nq, gates, nparams + nb, dtypestr, opts)
            self.descs = list(self.single.descs)
        else:
            self.Ks = int(np.prod(radices[: self.s]))
            self.prefix = CompiledCircuit(nq, gates[:cut], nparams + nb, dtypestr, opts)
            self.suffix = CompiledCircuit(nq, gates[cut:], nparams + nb, dtypestr, opts)
            self.descs = list(self.prefix.descs) + list(self.suffix.descs)

    def states(self, pfull, B, scale=None, scale_ready=None):
        """pfull [B*K, nparams + nb] (digits in the last nb columns) -> [B*K, 2^nq]; ``scale`` [B, K]: every state
        multiplied by its weight (applied where the prefix states are replicated over the suffix digits: the
        replication writes the batch anyway, and the suffix passes then run in place on it).  ``scale_ready``: an event
        after which ``scale`` may be read (it is produced on another stream, under this half's prefix)."""
        if self.s == 0:
            out = self.single.state(pfull)
            if scale is not None:
                if scale_ready is not None:
                    torch_ = __import__("torch")
                    torch_.cuda.current_stream(self.single.device).wait_event(scale_ready)
                out *= scale.reshape(-1, 1)
            return out
        import torch

        K, Ks = self.K, self.Ks
        # (measured and dropped, scripts/experiments/README.md: the suffix's tables built ahead on a side stream under the
        # prefix passes -- 1.001e11 / 1.010e11 with, 1.017e11 / 1.007e11 without at 8 circuits per call, 1.78e11 / 1.79e11
        # against 1.81e11 at 32: the chains of the two halves already overlap each other)
        # the prefix parameter rows are every (K / Ks)-th row of pfull: a strided view, no gather
        ppre = pfull.reshape(B * Ks, (K // Ks) * pfull.shape[-1])[:, : pfull.shape[-1]]
        pre = self.prefix.state(ppre)                                            # [B*Ks, 2^nq]
        if scale is not None and scale_ready is not None:
            torch.cuda.current_stream(self.suffix.device).wait_event(scale_ready)
        rep_n = K // Ks
        if rep_n & (rep_n - 1) == 0 and knob("cut_fused_rep", "1") != "0":
            # state b*K + j of the suffix batch = weight[b, j] * prefix state (b*K + j) >> log2(K / Ks): read by the suffix's
            # first pass itself (tcmi_spec_run_pass_from; materialised inside state() when that kernel is not there)
            return self.suffix.state(pfull, src=(pre, rep_n.bit_length() - 1, scale))
        if scale is None:
            rep = pre.reshape(B, Ks, 1, -1).expand(B, Ks, K // Ks, pre.shape[-1]).reshape(B * K, -1)
        else:
            rep = (pre.reshape(B, Ks, 1, -1) * scale.reshape(B, Ks, K // Ks, 1)).reshape(B * K, -1)
        return self.suffix.state(pfull, inputs=rep, consume_inputs=True)


class CutCircuit:
    """Wavefunction by cut contraction (``tcmi/cut.py``): two half-circuit batches through the
    tile-VM and one MFMA complex GEMM.  Same interface as ``CompiledCircuit`` (the adjoint sweep
    is delegated to the state-vector plan of the full circuit)."""

    def __init__(self, n, gates, nparams, dtypestr, opts, spec, full_cc):
        import torch

        self.n, self.n_exec, self.nparams, self.dtypestr = n, n, nparams, dtypestr
        self.spec = spec
        self.full = full_cc            # state-vector plan (adjoint sweep, inputs != |0>, stats)
        self.cfg = full_cc.cfg
        nb = len(spec.bonds)
        radices = [len(b.terms) for b in spec.bonds]
        self.left = _HalfBatch(spec.n_left, spec.left, nparams, nb, radices, dtypestr, opts)
        self.right = _HalfBatch(n - spec.n_left, spec.right, nparams, nb, radices, dtypestr, opts)
        self.tdtype, self.rdtype, self.code = full_cc.tdtype, full_cc.rdtype, full_cc.code
        self.device = full_cc.device
        self._lib = _lib.lib()
        self.K = spec.bond_dim
        digits = np.zeros((self.K, nb), dtype=np.float64)
        for b in range(self.K):
            x = b
            for k in reversed(range(nb)):
                digits[b, k] = x % radices[k]
                x //= radices[k]
        self.digits = _dev(digits, self.device, self.rdtype)          # [K, nb]
        self.descs = self.left.descs + self.right.descs               # for bookkeeping / stats
        # deferred last crossing gate (cut.Epilogue): applied by the join kernel itself (tcmi_cgemm_split_epi), so only the
        # split-GEMM join can run this spec; joins that cannot (TCMI_JOIN_GEMM=f32) go through the plain spec's CutCircuit
        # operand scales of the two-piece f16 join: powers of two that bring the largest possible |re|, |im|, |re + im| of a
        # half-circuit state (cut.half_bounds: <= sqrt(2) x the product of the gates' norms) under f16's 65504.  Only for
        # bounds up to 2^10: the low piece of an entry x is exact to 2^-22 |x| while scale * x / 2^11 stays above f16's
        # smallest normal, below that to 2^-25 / scale absolutely -- with a scale of 32 or more that is 2^-30, under the f32
        # rounding of any amplitude that matters; a looser bound (deep stacks of non-unitary gates) would push the states
        # into f16's subnormals, and such cuts keep the three-piece bf16 join
        self._f16 = None
        from . import cut as cut_mod

        bl, br = cut_mod.half_bounds(spec)
        if np.isfinite(bl) and np.isfinite(br) and max(bl, br) <= 1024.0 and min(bl, br) > 0:
            self._f16 = tuple(float(2.0 ** int(np.floor(np.log2(46000.0 / b)))) for b in (bl, br))
        self._plain_args = (n, gates, nparams, dtypestr, opts, getattr(spec, "plain", None), full_cc)
        self._plain = None
        self._epi = None
        if spec.epilogue is not None:
            fac = spec.epilogue.factors
            tab_i = np.array([(-1 if f[3] is None else f[3].index) for f in fac], dtype=np.int32)
            tab_f = np.zeros((len(fac), 98), dtype=np.float64)
            for g_, (c0, c1, c2, ref) in enumerate(fac):
                tab_f[g_, 0], tab_f[g_, 1] = (0.0, 0.0) if ref is None else (ref.scale, ref.offset)
                for j_, m_ in enumerate((c0, c1, c2)):
                    m_ = np.asarray(m_, dtype=np.complex128).reshape(16)
                    tab_f[g_, 2 + 32 * j_: 34 + 32 * j_: 2] = m_.real
                    tab_f[g_, 3 + 32 * j_: 35 + 32 * j_: 2] = m_.imag
            self._epi = (torch.as_tensor(tab_i).to(self.device), torch.as_tensor(tab_f.reshape(-1)).to(self.device), len(fac))

    def _weights(self, params):
        """w[B, K] = prod_k coef_k(digit_k, theta): one gather + one product over the bond axis (the
        per-bond coefficient table [B, nb, rmax] is built from one cos and one sin of all bond angles)."""
        import torch

        tabs = getattr(self, "_wtabs", None)
        if tabs is None:
            nb = len(self.spec.bonds)
            rmax = max(len(b.terms) for b in self.spec.bonds)
            const = np.zeros((nb, rmax), dtype=np.complex128)
            cmask = np.zeros((nb, rmax)); smask = np.zeros((nb, rmax))
            pidx = np.zeros((nb, rmax), dtype=np.int64); scale = np.zeros((nb, rmax)); offs = np.zeros((nb, rmax))
            for k, bond in enumerate(self.spec.bonds):
                for j, (_, _, (kind, ref)) in enumerate(bond.terms):
                    if kind == "const":
                        const[k, j] = complex(ref)
                    else:
                        (cmask if kind == "cos" else smask)[k, j] = 1.0
                        pidx[k, j], scale[k, j], offs[k, j] = ref.index, ref.scale, ref.offset
            dig = self.digits.to(torch.int64)                                   # [K, nb]
            tabs = dict(const=_dev(const, self.device), cmask=_dev(cmask, self.device), smask=_dev(smask, self.device),
                        pidx=_dev(pidx.reshape(-1), self.device), scale=_dev(scale, self.device),
                        offs=_dev(offs, self.device), dig=dig.t().contiguous().unsqueeze(0), nb=nb, rmax=rmax)
            self._wtabs = tabs
            # the same tables for the one-launch kernel (tcmi_cut_weights)
            kind = (cmask + 2 * smask).astype(np.int64)
            tab_i = (kind | (pidx << 2)).astype(np.int32).reshape(-1)
            tab_f = np.stack([scale, offs, const.real, const.imag], axis=-1).astype(np.float64).reshape(-1)
            tabs["tab_i"] = torch.as_tensor(tab_i).to(self.device)
            tabs["tab_f"] = torch.as_tensor(tab_f).to(self.device)
            tabs["dig8"] = self.digits.to(torch.uint8).contiguous()              # [K, nb]
        B = params.shape[0]
        nb, rmax = tabs["nb"], tabs["rmax"]
        p = params.contiguous()
        w = torch.empty(B, self.K, dtype=self.tdtype, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.tcmi_cut_weights(p.data_ptr(), p.stride(0), B, tabs["tab_i"].data_ptr(),
                                              tabs["tab_f"].data_ptr(), tabs["dig8"].data_ptr(), self.K, nb, rmax,
                                              w.data_ptr(), self.code, stream), "tcmi_cut_weights")
        return w

    def state(self, params=None, inputs=None, out=None, full=False):
        import torch

        if inputs is not None:
            return self.full.state(params, inputs, out, full)
        if params is None:
            params = torch.zeros(1, max(1, self.nparams), dtype=self.rdtype, device=self.device)
        p = params.reshape(-1, params.shape[-1]) if params.dim() > 1 else params.reshape(1, -1)
        p = p.to(device=self.device, dtype=self.rdtype)
        B, K = p.shape[0], self.K
        if self._epi is not None and not self._split_join():
            return self._plain_cut().state(params, inputs, out, full)
        pfull = torch.cat([p[:, : self.nparams].unsqueeze(1).expand(B, K, self.nparams),
                           self.digits.unsqueeze(0).expand(B, K, -1)], dim=2).reshape(B * K, -1).contiguous()
        # The two half-circuit batches are independent and neither fills the chip (one workgroup per state: B*Ks
        # workgroups in the prefix passes), so the right half runs on a second HIP stream beside the left one.
        cur = torch.cuda.current_stream(self.device)
        two = knob("cut_streams", "1") != "0"      # also under hipGraph capture (fork / join in the graph)
        # experiment switch (scripts/experiments/README.md, round 4): sub-batches on their own streams so that the passes of
        # one hide under the join of another -- measured SLOWER (7.4e10 -> 6.9e10 / 6.4e10 amplitudes/s at 2 / 4 sub-batches:
        # the join GEMM loses more on smaller batches than the passes cost), so off
        nsplit = min(int(knob("cut_split", "1")), B) if two else 1
        if nsplit >= 2:
            return self._state_pipelined(p, pfull, B, nsplit, out)
        if two:
            side = getattr(self, "_side", None)
            if side is None:
                side = self._side = _streams.side_stream(self.device, 0)
            side.wait_stream(cur)
            # the bond weights at the head of the right half's chain (computing them on the caller's stream at the head of
            # the left half's shorter chain measured no gain eager and slower under hipGraph replay: the extra event edge
            # costs more than the 11 us it moves)
            with torch.cuda.stream(side):
                w, w_ready = self._weights(p), None
            with torch.cuda.stream(side):
                R = self.right.states(pfull, B, scale=w, scale_ready=w_ready)     # [B*K, N], each state times its weight
            pfull.record_stream(side)
            p.record_stream(side)
            w.record_stream(side)
            L = self.left.states(pfull, B)                              # [B*K, M]
            xepi = self._epilogue_matrices(p)                           # (behind the left chain, the shorter one)
            cur.wait_stream(side)
            R.record_stream(cur)
        else:
            L = self.left.states(pfull, B)
            R = self.right.states(pfull, B, scale=self._weights(p))
            xepi = self._epilogue_matrices(p)
        M, N = 2**self.spec.n_left, 2 ** (self.n - self.spec.n_left)
        if out is None:
            out = torch.empty(B, M * N, dtype=self.tdtype, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with _timed("gemm", 1, 8.0 * M * N * K * B):
            self._join(L, R, out, M, N, K, B, stream, xepi)
        return out

    def _split_join(self):
        M, N = 2**self.spec.n_left, 2 ** (self.n - self.spec.n_left)
        return self.code == _lib.TCMI_C64 and JOIN_GEMM != "f32" and M % 128 == 0 and N % 128 == 0 and self.K % 32 == 0

    def _plain_cut(self):
        """The same cut with every crossing gate a bond (joins that cannot apply the epilogue)."""
        if self._plain is None:
            n, gates, nparams, dtypestr, opts, plain, full_cc = self._plain_args
            if plain is None:
                raise RuntimeError("cut contraction with a deferred gate needs the split-GEMM join (complex64)")
            self._plain = CutCircuit(n, gates, nparams, dtypestr, opts, plain, full_cc)
        return self._plain

    def _epilogue_matrices(self, p):
        """X [B, 16] complex64 of the deferred gate (one launch, tcmi_cut_epilogue), None without one."""
        import torch

        if self._epi is None:
            return None
        tab_i, tab_f, nfac = self._epi
        p = p.contiguous()
        x = torch.empty(p.shape[0], 16, dtype=torch.complex64, device=self.device)
        _lib.check(self._lib.tcmi_cut_epilogue(p.data_ptr(), p.stride(0), p.shape[0], tab_i.data_ptr(), tab_f.data_ptr(),
                                               nfac, x.data_ptr(), self.code,
                                               torch.cuda.current_stream(self.device).cuda_stream), "tcmi_cut_epilogue")
        return x

    def _join(self, L, R, out, M, N, K, B, stream, xepi=None):
        """psi[b] = L[b]^T . R[b] (k-major halves).  complex64 joins whose shape the kernels take run on the f16 matrix
        pipe with two-piece operands (``tcmi_cgemm_split_f16``; the operand scales from the cut's norm bounds) or, when the
        cut has no bound (``TCMI_JOIN_GEMM=bf16``: always), on the bf16 pipe with three-piece operands
        (``tcmi_cgemm_split``) -- both at f32 accuracy, measured against float64 next to the f32 MFMA kernel in
        tests/test_gpu_gemm_split.py; ``TCMI_JOIN_GEMM=f32`` keeps every join on ``tcmi_cgemm``."""
        if (self._f16 is not None and JOIN_GEMM == "split" and self.code == _lib.TCMI_C64 and M % 128 == 0 and N % 128 == 0
                and K % 32 == 0):
            _lib.check(self._lib.tcmi_cgemm_split_f16(L.data_ptr(), R.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N,
                                                      M * N, None if xepi is None else xepi.data_ptr(), self._f16[0],
                                                      self._f16[1], stream), "tcmi_cgemm_split_f16(cut)")
            return
        if xepi is not None:
            _lib.check(self._lib.tcmi_cgemm_split_epi(L.data_ptr(), R.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N,
                                                      M * N, xepi.data_ptr(), stream), "tcmi_cgemm_split_epi(cut)")
            return
        if self.code == _lib.TCMI_C64 and JOIN_GEMM != "f32" and M % 128 == 0 and N % 128 == 0 and K % 32 == 0:
            _lib.check(self._lib.tcmi_cgemm_split(L.data_ptr(), R.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N,
                                                  M * N, stream), "tcmi_cgemm_split(cut)")
            return
        _lib.check(self._lib.tcmi_cgemm(L.data_ptr(), R.data_ptr(), out.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                        1, self.code, stream), "tcmi_cgemm(cut)")

    def _state_pipelined(self, p, pfull, B, nsplit, out):
        """The batch in ``nsplit`` sub-batches, each on its own HIP stream: the half-circuit passes of a sub-batch are a
        chain of small launches (one workgroup per 12-qubit state) that cannot fill the chip, the join GEMM fills it.
        With every chain on its own stream and the joins in order on the caller's stream, the first join starts after
        the SHORTEST chain (a quarter of the batch) and the other chains finish underneath it: the passes leave the
        critical path (they were 14 % of the headline step).  Fork and join become part of a hipGraph under capture."""
        import torch

        K = self.K
        M, N = 2**self.spec.n_left, 2 ** (self.n - self.spec.n_left)
        if out is None:
            out = torch.empty(B, M * N, dtype=self.tdtype, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        sides = getattr(self, "_sides", None)
        if sides is None or len(sides) < nsplit:
            sides = self._sides = [_streams.side_stream(self.device, i_) for i_ in range(nsplit)]
        w = self._weights(p)                                    # [B, K], one launch for the whole batch
        xepi = self._epilogue_matrices(p)
        bounds = [(i * B) // nsplit for i in range(nsplit + 1)]
        parts = []
        for i in range(nsplit):
            b0, b1 = bounds[i], bounds[i + 1]
            st = sides[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                pf = pfull[b0 * K: b1 * K]
                R = self.right.states(pf, b1 - b0, scale=w[b0:b1])     # [(b1-b0)*K, N], each state times its weight
                L = self.left.states(pf, b1 - b0)                      # [(b1-b0)*K, M]
            parts.append((b0, b1, L, R, st))
        for t_ in (pfull, p, w):
            for st in sides[:nsplit]:
                t_.record_stream(st)
        stream = cur.cuda_stream
        for b0, b1, L, R, st in parts:
            cur.wait_stream(st)
            L.record_stream(cur)
            R.record_stream(cur)
            with _timed("gemm", 1, 8.0 * M * N * K * (b1 - b0)):
                self._join(L, R, out[b0:b1], M, N, K, b1 - b0, stream,
                           None if xepi is None else ((xepi[0], xepi[1][b0:b1]) if isinstance(xepi, tuple) else xepi[b0:b1]))
        return out

    def vjp(self, params, psi, g, **kw):
        return self.full.vjp(params, psi, g, **kw)

    @property
    def nonunitary(self):
        return self.full.nonunitary

    def stats(self):
        item = 8 if self.dtypestr == "complex64" else 16
        M, N = 2**self.spec.n_left, 2 ** (self.n - self.spec.n_left)
        return {"passes": len(self.descs), "rounds": 0, "bond": self.K,
                "bytes": item * (self.K * (M + N) + M * N), "flops": 8.0 * M * N * self.K}


class GraphedState:
    """One wavefunction evaluation (table build + every pass / GEMM launch) captured in a hipGraph
    (``torch.cuda.CUDAGraph``) and replayed: the launch-bound regime (small batches: ~25 launches for
    < 1 ms of kernels) pays one graph launch instead of a host round trip per kernel.  Static
    buffers: ``params`` [B, P] in, ``out`` [B, 2^n] out; ``__call__(params)`` copies into the static
    input and replays.  Works for ``CompiledCircuit`` and ``CutCircuit`` (anything with ``.state``)."""

    def __init__(self, cc, batch: int):
        import torch

        self.cc, self.batch = cc, batch
        npar = max(1, getattr(cc, "nparams", 1))
        self.params = torch.zeros(batch, npar, dtype=cc.rdtype, device=cc.device)
        self.out = torch.empty(batch, 2**cc.n_exec, dtype=cc.tdtype, device=cc.device)
        cur = torch.cuda.current_stream(cc.device)
        side = _streams.side_stream(cc.device, 1)       # (not stream 0: the cut contraction forks its right half onto that one)
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # warm-up outside capture (lazy allocations, caches)
            for _ in range(2):
                cc.state(self.params, out=self.out)
        cur.wait_stream(side)
        torch.cuda.synchronize(cc.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            cc.state(self.params, out=self.out)

    def __call__(self, params):
        self.params.copy_(params.reshape(self.batch, -1)[:, : self.params.shape[1]])
        self.graph.replay()
        return self.out


_CACHE: "OrderedDict[Tuple, CompiledCircuit]" = OrderedDict()   # LRU: a training loop over changing constant
_CACHE_MAX = 64                                                  # matrices must not grow host / device memory unboundedly


def get_compiled(n, gates, nparams, dtypestr, opts) -> CompiledCircuit:
    import torch

    from . import cons as _cons

    key = (
        structure_digest(n, dtypestr, gates), nparams,
        tuple(sorted((k, v) for k, v in (opts or {}).items() if k in ("lowbits", "R", "LT"))),
        torch.cuda.current_device() if torch.cuda.is_available() else -1,
        getattr(_cons, "_contractor_name", "greedy"), os.environ.get("TCMI_CUT_DEFER", "1"),
    )
    cc = _CACHE.get(key)
    if cc is None:
        cc = CompiledCircuit(n, gates, nparams, dtypestr, opts)
        cc = _maybe_cut(cc, n, gates, nparams, dtypestr, opts)
        _CACHE[key] = cc
        while len(_CACHE) > _CACHE_MAX:
            _CACHE.popitem(last=False)
    else:
        _CACHE.move_to_end(key)
    return cc


def _maybe_cut(cc, n, gates, nparams, dtypestr, opts):
    """Pick the cheaper contraction order for ``wavefunction``: state-vector plan or cut contraction.
    ``set_contractor("plain")`` / ``"tilevm"`` forces the former, ``"cut"`` the latter."""
    best = choose_cut(n, gates, nparams, dtypestr, cc.plan)
    return cc if best is None else CutCircuit(n, gates, nparams, dtypestr, opts, best, cc)


def choose_cut(n, gates, nparams, dtypestr, plan):
    """The cut (``cut.CutSpec``) the executor would contract this circuit's wavefunction with, or None for the
    state-vector plan.  Host work only (also used to pre-compile the half-circuits' specialised kernels)."""
    from . import cons
    from . import cut as C

    method = getattr(cons, "_contractor_name", "greedy")
    if method in ("plain", "plain-experimental", "tilevm") or n < 16 or dtypestr != "complex64":
        return None
    best, best_key = None, None
    for nl in sorted({n // 2, (n + 1) // 2, n // 2 - 1, n // 2 + 1}):
        if nl < 8 or n - nl < 8:
            continue
        spec = C.make_cut(gates, n, nl, nparams)
        if spec is None or len(spec.bonds) == 0:
            continue
        # smallest bond first; then halves that fit the 12-bit tile (one pass each); then balance
        key = (spec.bond_dim, int(min(nl, n - nl) < 12), abs(2 * nl - n))
        if best is None or key < best_key:
            best, best_key = spec, key
    if best is None:
        if method == "cut":
            raise ValueError("set_contractor('cut'): this circuit cannot be cut (see tcmi/cut.py)")
        return None
    # the state-vector plan is priced on its live tiles (the circuit starts from |0...0>: live_masks), the join on the
    # kernel it will run on (tcmi_cgemm_split for complex64 joins of whole 128 x 128 tiles with K a multiple of 32)
    t_vm = vm_cost_us(plan, live_masks(plan.descs, plan.n)[1] if (LIVE_PLAN and getattr(plan.cfg, "gen", 1) >= 2) else None)
    M, N = 2 ** best.n_left, 2 ** (n - best.n_left)
    split = JOIN_GEMM != "f32" and M % 128 == 0 and N % 128 == 0 and best.bond_dim % 32 == 0
    t_gemm = 8.0 * 2.0**n * best.bond_dim / ((GEMM_SPLIT_TFLOPS if split else GEMM_TFLOPS) * 1e6)      # microseconds
    t_halves = 2 * 15.0 + best.bond_dim * 2.0 ** max(best.n_left, n - best.n_left) / 2.0**24 * 200.0  # two-level
    # config 2 (n = 24, d = 8, bond 256), per state at batch 8, round 4: cut 161 us measured (model 174), state-vector plan
    # on its live tiles 181 us (model 181); near-ties go to the cut (its join has measured better than its model so far)
    # the last crossing gate applied by the join kernel instead of being a bond (cut.py: half the bond for ZZ / CNOT / CZ)
    # when the circuit allows it and the split-GEMM join takes the smaller shape; TCMI_CUT_DEFER=0 keeps every bond
    # (two deferred gates -- the tail as a gate program of the join kernel -- were built and measured in round 5: 1.73e11
    # against 1.81e11 amplitudes/s on config 2, the program's vector instructions cost what the quarter bond saves; the
    # variant lives in the round-5 tree)
    for ndefer in [1]:
        if not split or os.environ.get("TCMI_CUT_DEFER", "1") == "0":
            break
        dspec = C.make_cut(gates, n, best.n_left, nparams, defer=ndefer)
        if dspec is not None and dspec.epilogue is not None and dspec.bond_dim % 32 == 0 and len(dspec.bonds) > 0:
            dspec.plain = best
            if method == "cut":
                return dspec
            t_gemm_d = 8.0 * 2.0**n * dspec.bond_dim / (GEMM_SPLIT_TFLOPS * 1e6)
            t_halves_d = 2 * 15.0 + dspec.bond_dim * 2.0 ** max(best.n_left, n - best.n_left) / 2.0**24 * 200.0
            if (t_gemm_d + t_halves_d) < 1.05 * t_vm:
                return dspec
    if method == "cut" or (t_gemm + t_halves) < 1.05 * t_vm:
        return best
    return None
