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


MULTIRANK_OUT = os.path.join(ROOT, ".pytest_cache", "multirank_slices.json")
MULTIRANK4_OUT = os.path.join(ROOT, ".pytest_cache", "multirank_slices_w4.json")


def pytest_sessionstart(session):
    """The two-rank and four-rank DistributedContractor runs of tests/test_gpu_multirank.py are started here, BEFORE this process
    makes any HIP call: its ranks are fresh child processes with their own GPU contexts (a process that has
    initialised the GPU must not be the one that starts them on this pool).  Counting devices does not initialise."""
    expr = session.config.getoption("-m", default="") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import subprocess

        import torch

        if torch.cuda.device_count() < 1:
            return
        os.makedirs(os.path.dirname(MULTIRANK_OUT), exist_ok=True)
        for world, out in ((2, MULTIRANK_OUT), (4, MULTIRANK4_OUT)):
            if os.path.exists(out):
                os.remove(out)
            subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_slices.py"), str(world), out],
                           timeout=900, check=False)
    except Exception as e:  # noqa: BLE001  (the test reports the missing file)
        print("multirank launcher failed:", e)


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
