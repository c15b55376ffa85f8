import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tensorcircuit-ng_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def hipb():
    """Mirror of the reference's backend fixtures (tests/conftest.py:16-71)."""
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    yield tc
    tc.set_dtype("complex64")


@pytest.fixture
def highp():
    import tcmi as tc

    tc.set_dtype("complex128")
    yield
    tc.set_dtype("complex64")
