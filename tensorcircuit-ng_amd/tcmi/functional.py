"""Differentiable / vmappable entry points that connect ``Circuit`` to the plan executor.

``torch.autograd.Function`` is used as plumbing only: it lets the reference's
``backend.value_and_grad / vmap / vvag`` contracts (``abstract_backend.py:2262-2293, 2520-2591``)
compose with the HIP executor -- ``vmap`` maps to the executor's batch dimension (one launch over
``[B, 2^n]``), ``grad`` to the adjoint-sweep kernels.
"""

from typing import Any, List, Optional, Tuple

import numpy as np

from . import cons


def _torch():
    import torch

    return torch


_StateFn = None


def _state_fn():
    global _StateFn
    if _StateFn is not None:
        return _StateFn
    torch = _torch()

    class StateFn(torch.autograd.Function):
        """params [P] -> state [2^n]; under vmap params [B, P] -> [B, 2^n] in one batched launch."""

        generate_vmap_rule = False

        @staticmethod
        def forward(params, cc, inputs):
            out = cc.state(params, inputs)
            return out[0] if params.dim() == 1 else out

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.cc = inputs[1]

        @staticmethod
        def backward(ctx, grad_out):
            raise NotImplementedError(
                "Backend 'hip' has not implemented the VJP of `wavefunction`; differentiate "
                "`expectation` outputs (adjoint sweep) instead."
            )

        @staticmethod
        def vmap(info, in_dims, params, cc, inputs):
            if in_dims[0] is None:
                out = StateFn.apply(params, cc, inputs)
                return out.unsqueeze(0).expand(info.batch_size, *out.shape), 0
            p = params.movedim(in_dims[0], 0)
            lead = p.shape[:-1]
            out = cc.state(p.reshape(-1, p.shape[-1]), inputs)
            return out.reshape(*lead, out.shape[-1]), 0

    _StateFn = StateFn
    return StateFn


def circuit_state(cc, params, inputs=None):
    """State of a compiled circuit; ``params`` None (no parameters) or a real tensor [P]."""
    if params is None:
        return cc.state(None, inputs)[0]
    return _state_fn().apply(params, cc, inputs)


def circuit_expectation(circuit, ops: List[Tuple[np.ndarray, Tuple[int, ...]]]):
    from .expectation import expectation_of_ops

    return expectation_of_ops(circuit, ops)
