"""reference ``tensorcircuit/interfaces/torch.py:17-125``.

On the hip backend the array container already IS a torch (ROCm) tensor and every circuit primitive is a
``torch.autograd.Function`` (``tcmi/functional.py``), so a quantum function is directly usable inside
torch autograd: the interface only moves the arguments to the backend's device / dtype conventions and
returns the result as produced (zero copy; no dlpack hop is needed)."""

from typing import Any, Callable

from .. import cons

Tensor = Any


def torch_interface(fun: Callable[..., Any], jit: bool = False, enable_dlpack: bool = False) -> Callable[..., Any]:
    """Wrap ``fun`` so that it accepts / returns torch tensors and takes part in ``torch.autograd``.
    ``jit`` is accepted for signature compatibility: plans are compiled and cached by circuit structure."""
    import torch

    def wrapped(*args: Any, **kws: Any) -> Any:
        dev = cons.backend.device
        moved = [a.to(dev) if torch.is_tensor(a) and a.device != dev else a for a in args]
        from ..expectation import resolve

        # lazily fused expectation sums become plain tensors at the torch boundary
        return torch.utils._pytree.tree_map(resolve, fun(*moved, **kws))

    wrapped.__name__ = getattr(fun, "__name__", "torch_interface_fn")
    return wrapped


pytorch_interface = torch_interface


def torch_interface_kws(f: Callable[..., Any], jit: bool = True, enable_dlpack: bool = False) -> Callable[..., Any]:
    """reference interfaces/torch.py ``torch_interface_kws``: keyword arguments are static."""
    return torch_interface(f, jit=jit, enable_dlpack=enable_dlpack)
