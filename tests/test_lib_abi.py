"""The C-ABI library loads on a CPU-only box and exports every symbol include/tcmi.h declares
(no compute calls without a GPU)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tcmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tcmi_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_header_symbols():
    import __graft_entry__ as g
    from tcmi import _lib

    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert names, "no declarations parsed"
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/tcmi.h but not exported"
    assert sorted(_lib.exported_symbols()) == names
    assert handle.tcmi_version() == 1


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch and the message is retrievable."""
    from tcmi import _lib

    lib = _lib.lib()
    rc = lib.tcmi_run_pass(None, 0, 1, 10, 5, 8, None, None, None, 0, None, 0, 1, 0, 0, None)
    assert rc == -1
    assert b"tcmi_run_pass" in lib.tcmi_last_error()
    assert lib.tcmi_init_zero_state(None, 0, 1, 10, 0, None) == -1
    # the collective: a null communicator / buffer is refused before RCCL is even opened
    assert lib.tcmi_allreduce_sum(None, None, 4, _lib.TCMI_F64, None) == -1
    assert b"tcmi_allreduce_sum" in lib.tcmi_last_error()
    assert lib.tcmi_comm_init(None, 0, 1, None) == -1 and lib.tcmi_comm_destroy(None) == 0
    # and a generated kernel's launcher refuses a null handle (tcmi_spec_set_flags guards the "src" argument buffer)
    assert lib.tcmi_spec_set_flags(None, 1) == -1
    with pytest.raises(_lib.TcmiError):
        _lib.check(rc, "tcmi_run_pass")


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: with the shared object absent every product entry point raises TcmiError."""
    from tcmi import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libtcmi_absent.so"))
    with pytest.raises(_lib.TcmiError, match="HIP extension not built"):
        _lib.lib()
    import torch
    from tcmi import linalg as LA

    with pytest.raises(_lib.TcmiError):
        LA.matmul(torch.zeros(2, 2, dtype=torch.complex64), torch.zeros(2, 2, dtype=torch.complex64))
