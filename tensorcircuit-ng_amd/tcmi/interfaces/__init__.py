from .torch import torch_interface, torch_interface_kws, pytorch_interface  # noqa: F401
