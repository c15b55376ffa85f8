"""reference ``tensorcircuit/torchnn.py:16-138``: ``QuantumNet`` (alias ``TorchLayer``), a ``torch.nn.Module``
around a quantum function ``f(inputs, *weights)``; the batch axis of the inputs is vmapped through the
hip backend (one batched plan execution, not a Python loop)."""

from typing import Any, Callable, Sequence, Tuple, Union

import torch

from . import cons
from .interfaces.torch import torch_interface

Tensor = Any


class QuantumNet(torch.nn.Module):
    def __init__(self, f: Callable[..., Any], weights_shape: Sequence[Any], initializer: Union[Any, Sequence[Any]] = None,
                 use_vmap: bool = True, vectorized_argnums: Union[int, Sequence[int]] = 0, use_interface: bool = True,
                 use_jit: bool = True, enable_dlpack: bool = False):
        super().__init__()
        if use_vmap:
            f = cons.backend.vmap(f, vectorized_argnums=vectorized_argnums)
        if use_interface:
            f = torch_interface(f, jit=use_jit, enable_dlpack=enable_dlpack)
        self.f = f
        if len(weights_shape) > 0 and isinstance(weights_shape[0], int):
            weights_shape = [tuple(weights_shape)]
        if initializer is not None and not isinstance(initializer, (list, tuple)):
            initializer = [initializer]
        self.q_weights = torch.nn.ParameterList()
        for k, ws in enumerate(weights_shape):
            init = torch.randn(tuple(ws)) if initializer is None else torch.as_tensor(initializer[k]).clone()
            self.q_weights.append(torch.nn.Parameter(init.to(torch.float32 if cons.rdtypestr == "float32" else torch.float64)))

    def forward(self, *inputs: Tensor) -> Tensor:
        return self.f(*inputs, *self.q_weights)


TorchLayer = QuantumNet
