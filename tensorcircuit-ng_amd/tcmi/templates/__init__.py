from . import blocks, measurements  # noqa: F401
