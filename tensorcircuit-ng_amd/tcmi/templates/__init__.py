from . import measurements  # noqa: F401
