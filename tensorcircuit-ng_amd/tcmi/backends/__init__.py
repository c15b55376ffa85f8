"""Backend registry (reference ``tensorcircuit/backends/backend_factory.py:26-59``)."""

from typing import Any, Dict, Union

from .hip_backend import HipBackend

_BACKENDS: Dict[str, Any] = {"hip": HipBackend}
_ALIASES = {"rocm": "hip", "tcmi": "hip", "mi355x": "hip"}
_INSTANTIATED_BACKENDS: Dict[str, Any] = {}


def get_backend(backend: Union[str, Any]) -> Any:
    """Name or instance -> memoised backend instance; unknown names raise
    ``ValueError("Backend '{}' does not exist")`` as in the reference."""
    if isinstance(backend, HipBackend):
        return backend
    name = _ALIASES.get(backend, backend)
    if name not in _BACKENDS:
        raise ValueError("Backend '{}' does not exist".format(backend))
    if name in _INSTANTIATED_BACKENDS:
        return _INSTANTIATED_BACKENDS[name]
    _INSTANTIATED_BACKENDS[name] = _BACKENDS[name]()
    return _INSTANTIATED_BACKENDS[name]
