"""Differentiable / vmappable entry points that connect ``Circuit`` to the plan executor.

``torch.autograd.Function`` is used as plumbing only: it lets the reference's
``backend.value_and_grad / vmap / vvag`` contracts (``abstract_backend.py:2262-2293, 2520-2591``)
compose with the HIP executor.  Two primitives:

* ``StateFn``   params [P] -> full executor state [2^n_exec]; VJP = adjoint (reverse) sweep of the
  compiled plan (``CompiledCircuit.vjp``): O(1) extra states instead of the reference's
  store-every-intermediate reverse mode, which is infeasible at 28 qubits.
* ``MeasureFn`` state -> <psi|P_t|psi> for a set of Pauli strings (fused measurement passes);
  VJP = 2 * sum_t Re(g_t) P_t |psi>  (``tcmi_apply_pauli_sum``).

Under ``torch.vmap`` both map the batch axis onto the executor's batch dimension (one launch per
pass over ``[B, 2^n]``).

Second order (reference ``abstract_backend.py:2295-2492``: jvp / jacfwd / hessian differentiate the reverse pass).  The
VJP primitives have a backward of their own, exact, by the shift rule for STATES: every recorded gate is
``M(a) = c0 + cos(a) c1 + sin(a) c2`` in its angle ``a = scale * theta_j + offset`` (``plan.GateRec``), so
``d psi / d a = [psi(a + pi/2) - psi(a - pi/2)] / 2`` whatever c0, c1, c2 are.  The derivative of
``Re <g | d psi / d theta>`` along a cotangent ``c`` therefore needs two more circuit evaluations (and two more sweeps) per
non-zero entry of ``c`` -- a Hessian row costs O(1) gradients, like forward-over-reverse in the reference.  A parameter
that feeds more than one gate has no two-point rule; such circuits raise.
"""

from typing import Any, List, Optional, Tuple

import numpy as np

from . import cons

_FNS = {}


def _torch():
    import torch

    return torch


def _fns():
    if _FNS:
        return _FNS
    torch = _torch()

    class StateFn(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(params, cc, inputs):
            out = cc.state(params, inputs, full=True)
            return out[0] if params.dim() == 1 else out

        @staticmethod
        def setup_context(ctx, inputs, output):
            params, cc, inp = inputs
            ctx.cc = cc
            ctx.has_inputs = inp is not None
            ctx.inp_shape = tuple(inp.shape) if inp is not None else None
            if inp is not None:
                ctx.save_for_backward(params, output, inp)
            else:
                ctx.save_for_backward(params, output)

        @staticmethod
        def backward(ctx, grad_out):
            if ctx.has_inputs:
                params, psi, inp = ctx.saved_tensors
                gp, gin = StateVjpInFn.apply(params, psi, grad_out, ctx.cc, inp)
                return gp, None, gin
            params, psi = ctx.saved_tensors
            return StateVjpFn.apply(params, psi, grad_out, ctx.cc), None, None

        @staticmethod
        def vmap(info, in_dims, params, cc, inputs):
            B = info.batch_size
            in_b = inputs is not None and in_dims[2] is not None
            if in_dims[0] is None and not in_b:
                out = StateFn.apply(params, cc, inputs)
                return out.unsqueeze(0).expand(B, *out.shape), 0
            # parameters [.., P] and (optionally) input states [.., 2^n], either of them batched: both are lifted to
            # [B, rows, .] (rows = the columns of a matrix-shaped input, or 1) and run as ONE batch of B * rows states
            p = params.movedim(in_dims[0], 0) if in_dims[0] is not None else params.unsqueeze(0).expand(B, *params.shape)
            if inputs is None:
                lead = p.shape[:-1]
                out = StateFn.apply(p.reshape(-1, p.shape[-1]), cc, None)
                return out.reshape(*lead, out.shape[-1]), 0
            x = inputs.movedim(in_dims[2], 0) if in_b else inputs.unsqueeze(0).expand(B, *inputs.shape)
            if p.dim() == 2:
                p = p.unsqueeze(1)
            if x.dim() == 2:
                x = x.unsqueeze(1)
            rows = max(p.shape[1], x.shape[1])
            p = p.expand(B, rows, p.shape[-1]).reshape(B * rows, p.shape[-1])
            x = x.expand(B, rows, x.shape[-1]).reshape(B * rows, x.shape[-1]).contiguous()
            out = StateFn.apply(p, cc, x).reshape(B, rows, -1)
            single = (params.dim() - (1 if in_dims[0] is not None else 0)) == 1 and (inputs.dim() - (1 if in_b else 0)) == 1
            return (out[:, 0] if single else out), 0

    class StateVjpFn(torch.autograd.Function):
        """(params, psi, g) -> dL/dparams = Re <g | d psi / d params> via the adjoint sweep."""

        generate_vmap_rule = False

        @staticmethod
        def forward(params, psi, g, cc):
            single = params.dim() == 1
            gp = cc.vjp(params.reshape(-1, params.shape[-1]), psi.reshape(-1, psi.shape[-1]),
                        g.reshape(-1, g.shape[-1]), from_zero=True)       # StateFn without inputs: psi came from |0...0>
            return gp[0] if single else gp

        @staticmethod
        def setup_context(ctx, inputs, output):
            params, psi, g, cc = inputs
            ctx.cc = cc
            ctx.save_for_backward(params, g)

        @staticmethod
        def backward(ctx, c):
            # (psi is a function of params: its slot gets no gradient of its own, the params slot carries all of it)
            params, g = ctx.saved_tensors
            gp2, tang = _second_order(ctx.cc, params, g, c, ctx.needs_input_grad[0], ctx.needs_input_grad[2])
            return gp2, None, tang, None

        @staticmethod
        def vmap(info, in_dims, params, psi, g, cc):
            B = info.batch_size

            def lift(t, d):
                if d is None:
                    return t.unsqueeze(0).expand(B, *t.shape)
                return t.movedim(d, 0)

            p, s, gg = lift(params, in_dims[0]), lift(psi, in_dims[1]), lift(g, in_dims[2])
            lead = p.shape[:-1]
            out = StateVjpFn.apply(p.reshape(-1, p.shape[-1]), s.reshape(-1, s.shape[-1]).contiguous(),
                                   gg.reshape(-1, gg.shape[-1]).contiguous(), cc)
            return out.reshape(*lead, out.shape[-1]), 0

    class StateVjpInFn(torch.autograd.Function):
        """(params, psi, g, inputs) -> (dL/dparams, dL/dinputs): the cotangent of the input state of
        ``Circuit(inputs=...)`` is lambda after the whole adjoint sweep (reference circuit.py:90-104 differentiates
        through the input state)."""

        generate_vmap_rule = False

        @staticmethod
        def forward(params, psi, g, cc, inp):
            single = params.dim() == 1 and inp.dim() == 1
            s2, g2 = psi.reshape(-1, psi.shape[-1]), g.reshape(-1, g.shape[-1])
            # a circuit without parameters (all gates constant, only the input state is differentiated)
            p2 = params.reshape(-1, params.shape[-1]) if params.shape[-1] else params.new_zeros((s2.shape[0], 0))
            if p2.shape[0] != s2.shape[0]:
                p2 = p2.expand(s2.shape[0], -1)
            gp, lam = cc.vjp(p2.contiguous(), s2, g2, inputs=inp, want_input_grad=True)
            gin = lam[:, : inp.shape[-1]]
            if inp.dim() == 1:
                gin = gin.sum(0) if gin.shape[0] > 1 else gin[0]
            else:
                gin = gin.reshape(inp.shape)
            if params.dim() == 1:
                gp = gp.sum(0) if gp.shape[0] > 1 else gp[0]
            return gp, gin.to(inp.dtype)

        @staticmethod
        def setup_context(ctx, inputs, output):
            pass

        @staticmethod
        def backward(ctx, *a):
            raise NotImplementedError("Backend 'hip' has not implemented second-order derivatives.")

    class MeasureFn(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(state, cm):
            single = state.dim() == 1
            vals = cm.run(state.reshape(-1, state.shape[-1]).contiguous())
            return vals[0] if single else vals

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.cm = inputs[1]
            ctx.save_for_backward(inputs[0])

        @staticmethod
        def backward(ctx, grad_vals):
            (state,) = ctx.saved_tensors
            return MeasureVjpFn.apply(state, grad_vals, ctx.cm), None

        @staticmethod
        def vmap(info, in_dims, state, cm):
            if in_dims[0] is None:
                out = MeasureFn.apply(state, cm)
                return out.unsqueeze(0).expand(info.batch_size, *out.shape), 0
            s = state.movedim(in_dims[0], 0)
            lead = s.shape[:-1]
            out = MeasureFn.apply(s.reshape(-1, s.shape[-1]), cm)
            return out.reshape(*lead, out.shape[-1]), 0

    class MeasureVjpFn(torch.autograd.Function):
        """(state, g_vals) -> cotangent of the state: 2 * sum_t Re(g_t) P_t |psi>."""

        generate_vmap_rule = False

        @staticmethod
        def forward(state, g, cm):
            single = state.dim() == 1
            out = cm.apply_sum(state.reshape(-1, state.shape[-1]).contiguous(),
                               g.reshape(-1, g.shape[-1]))
            return out[0] if single else out

        @staticmethod
        def setup_context(ctx, inputs, output):
            state, g, cm = inputs
            ctx.cm = cm
            ctx.save_for_backward(state, g)

        @staticmethod
        def backward(ctx, t):
            """out = A psi, A = 2 sum_t Re(g_t) P_t (Hermitian).  Cotangent t of out: the state gets A t; g_t gets
            2 Re <t | P_t psi>, taken from the measurement passes by polarisation: [<t+psi|P|t+psi> - <t-psi|P|t-psi>] / 2
            with t scaled to psi's norm first."""
            state, g = ctx.saved_tensors
            cm = ctx.cm
            single = state.dim() == 1
            s2 = state.reshape(-1, state.shape[-1]).contiguous()
            t2 = t.reshape(-1, t.shape[-1]).contiguous()
            g2 = g.reshape(-1, g.shape[-1])
            gs = gg = None
            if ctx.needs_input_grad[0]:
                gs = cm.apply_sum(t2, g2)
                gs = gs[0] if single else gs.reshape(state.shape)
            if ctx.needs_input_grad[1]:
                tn = torch.linalg.vector_norm(t2, dim=-1, keepdim=True)
                sn = torch.linalg.vector_norm(s2, dim=-1, keepdim=True)
                alpha = torch.where(tn > 0, sn / tn.clamp_min(1e-300), torch.zeros_like(tn))
                th = t2 * alpha.to(t2.dtype)
                d = cm.run(th + s2) - cm.run(th - s2)
                gg = (0.5 * d / alpha.clamp_min(1e-300).to(d.dtype)) * (alpha > 0).to(d.dtype)
                gg = gg.reshape(g.shape).to(g.dtype)
            return gs, gg, None

        @staticmethod
        def vmap(info, in_dims, state, g, cm):
            B = info.batch_size

            def lift(t, d):
                if d is None:
                    return t.unsqueeze(0).expand(B, *t.shape)
                return t.movedim(d, 0)

            s, gg = lift(state, in_dims[0]), lift(g, in_dims[1])
            lead = s.shape[:-1]
            out = MeasureVjpFn.apply(s.reshape(-1, s.shape[-1]).contiguous(), gg.reshape(-1, gg.shape[-1]).contiguous(), cm)
            return out.reshape(*lead, out.shape[-1]), 0

    _FNS.update(StateFn=StateFn, StateVjpFn=StateVjpFn, StateVjpInFn=StateVjpInFn, MeasureFn=MeasureFn,
                MeasureVjpFn=MeasureVjpFn)
    return _FNS


def _shift_scales(cc):
    """(scale, count, other): scale[j] of the one gate angle a = scale * theta_j + offset that parameter j feeds, how many
    gates it feeds that way, and how many times it is read in a way the shift rule below has no form for -- through a
    diagonal term of a gate whose own ``param`` is another parameter (a fused diagonal) or as the selector of a
    ``select`` gate (cut selector columns)."""
    tab = getattr(cc, "_shift_tab", None)
    if tab is None:
        gates = getattr(cc, "full", cc)._exec_gates
        npar = max(1, cc.nparams)
        scale, count, other = np.zeros(npar), np.zeros(npar, dtype=np.int64), np.zeros(npar, dtype=np.int64)
        for g in gates:
            own = g.param.index if g.param is not None else None
            if g.param is not None and g.param.index < npar:
                if g.select is None:
                    scale[g.param.index] = g.param.scale
                    count[g.param.index] += 1
                else:
                    other[g.param.index] += 1
            for t in (g.diag or ()):
                if t.param is not None and t.param.index != own and t.param.index < npar:
                    other[t.param.index] += 1
        tab = cc._shift_tab = (scale, count, other)
    return tab


def _second_order(cc, params, g, c, need_params, need_g):
    """Backward of (params, g) -> gp = Re <g | d psi / d params> along the cotangent ``c`` of gp (module docstring):
    (d/d params of sum_j c_j gp_j, d/d g of the same = the tangent state sum_j c_j d psi / d theta_j)."""
    torch = _torch()
    if getattr(cc, "nonunitary", None):
        raise NotImplementedError("Backend 'hip' has not implemented second-order derivatives through non-unitary gates.")
    single = params.dim() == 1
    npar = params.shape[-1]
    p2 = params.reshape(-1, npar)
    g2 = g.reshape(-1, g.shape[-1])
    c2 = c.reshape(-1, npar).to(torch.float64)
    nel = g2.shape[-1]
    scale, count, other = _shift_scales(cc)
    rows = torch.nonzero(c2).cpu().numpy()                   # (batch row, parameter) of every non-zero cotangent entry
    gp_out = torch.zeros(p2.shape, dtype=torch.float64, device=p2.device) if need_params else None
    t_out = torch.zeros_like(g2) if need_g else None
    if rows.shape[0]:
        js = rows[:, 1]
        if (other[js] > 0).any():
            raise NotImplementedError("Backend 'hip': second-order derivatives need every parameter to be the angle of ONE "
                                      "gate (parameters %s are read through a fused diagonal term or select a gate)"
                                      % sorted(set(int(j) for j in js[other[js] > 0])))
        if (count[js] > 1).any():
            raise NotImplementedError("Backend 'hip': second-order derivatives need every parameter to feed ONE gate "
                                      "(parameters %s feed several)" % sorted(set(int(j) for j in js[count[js] > 1])))
        rows = rows[count[js] == 1]                          # parameters no gate reads: zero derivative
    # (the reverse-over-reverse jvp differentiates along a cotangent g that is identically zero: nothing to sweep)
    g_nonzero = bool(need_params and rows.shape[0] and (g2 != 0).any())
    item = 8 if g2.dtype == torch.complex64 else 16
    step = max(1, int((1 << 30) // (nel * item)))            # shifted states per chunk: 2 * step of them alive
    sc = torch.as_tensor(scale, dtype=torch.float64, device=p2.device)
    for r0 in range(0, rows.shape[0], step):
        bi = torch.as_tensor(rows[r0: r0 + step, 0], device=p2.device)
        ji = torch.as_tensor(rows[r0: r0 + step, 1], device=p2.device)
        m = bi.shape[0]
        ar = torch.arange(m, device=p2.device)
        delta = (0.5 * np.pi / sc[ji]).to(p2.dtype)
        pp = p2[bi].repeat(2, 1)
        pp[ar, ji] += delta
        pp[m + ar, ji] -= delta
        st = cc.state(pp, None, full=True)                   # [2 m, 2^n_exec]
        coef = 0.5 * c2[bi, ji] * sc[ji]
        if need_g:
            t_out.index_add_(0, bi, (st[:m] - st[m:]) * coef.to(st.dtype).unsqueeze(1))
        if need_params and g_nonzero:
            gv = cc.vjp(pp, st, g2[bi].repeat(2, 1).contiguous(), from_zero=True)[:, :npar].to(torch.float64)
            gp_out.index_add_(0, bi, (gv[:m] - gv[m:]) * coef.unsqueeze(1))
        del st
    if need_params:
        gp_out = (gp_out[0] if single else gp_out.reshape(params.shape)).to(params.dtype)
    if need_g:
        t_out = t_out[0] if g.dim() == 1 else t_out.reshape(g.shape)
    return gp_out, t_out


def circuit_state_full(circuit):
    """Full executor buffer [2^n_exec] of the circuit state (differentiable w.r.t. the parameters)."""
    cc = circuit._compiled()
    params = circuit._param_tensor()
    inputs = circuit._input_tensor()
    if params is None:
        if inputs is not None and _torch().is_tensor(inputs) and (
                inputs.requires_grad or _torch()._C._functorch.is_functorch_wrapped_tensor(inputs)):
            # no parametrised gate, but the input state is being differentiated: same primitive, empty parameter row
            dummy = _torch().zeros(0, dtype=cc.rdtype, device=cc.device)
            return _fns()["StateFn"].apply(dummy, cc, inputs)
        return cc.state(None, inputs, full=True)[0]
    return _fns()["StateFn"].apply(params, cc, inputs)


def circuit_state(circuit):
    full = circuit_state_full(circuit)
    n = circuit._nqubits
    return full[..., : 2**n] if full.shape[-1] != 2**n else full


def circuit_pauli_values(circuit, strings):
    from .executor import get_measure
    from .expectation import _circuit_full_state

    cc = circuit._compiled()
    cm = get_measure(circuit._nqubits, cc.n_exec, strings, cons.dtypestr)
    state = _circuit_full_state(circuit)
    return _fns()["MeasureFn"].apply(state, cm)


def circuit_expectation(circuit, ops: List[Tuple[np.ndarray, Tuple[int, ...]]]):
    from .expectation import expectation_of_ops

    return expectation_of_ops(circuit, ops)
