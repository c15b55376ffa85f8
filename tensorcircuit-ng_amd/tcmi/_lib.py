"""ctypes binding of ``libtcmi.so`` (the C ABI declared in ``include/tcmi.h``).

The library is built in-tree by ``tensorcircuit-ng_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no fallback: if the shared object is missing or a call fails, the product raises.
"""

import ctypes
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libtcmi.so")

TCMI_C64 = 0
TCMI_C128 = 1
TCMI_F32 = 2
TCMI_F64 = 3

_lib = None


class TcmiError(RuntimeError):
    pass


_SIGNATURES = {
    "tcmi_version": (ctypes.c_int, []),
    "tcmi_last_error": (ctypes.c_char_p, []),
    "tcmi_device_count": (ctypes.c_int, []),
    "tcmi_init_zero_state": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_build_tables": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
         ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_greedy_path": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_subtree_dp": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "tcmi_cut_weights": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_cut_epilogue": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_run_pass": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_contract_scattered": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_apply_pauli_sum": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_pauli_sum_tile_bits": (ctypes.c_int, [ctypes.c_int]),
    "tcmi_apply_pauli_sum_tiled": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_build_adjoint_tables": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
         ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_run_adjoint_pass": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
         ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p],
    ),
    "tcmi_tensordot_bits": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_tensordot_bits_ex": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_tensordot_bits_small_ok": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "tcmi_tensordot_small_desc": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "tcmi_tensordot_small_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tcmi_permute_bits": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong,
         ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_cgemm": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong,
         ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_cgemm_split": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p],
    ),
    "tcmi_cgemm_split_epi": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p,
         ctypes.c_void_p],
    ),
    "tcmi_cgemm_split_f16": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p,
         ctypes.c_float, ctypes.c_float, ctypes.c_void_p],
    ),
    "tcmi_vdot": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
         ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_svd_work_bytes": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "tcmi_svd_trunc_batched": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_qr_work_bytes": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "tcmi_qr_batched": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
    "tcmi_spec_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]),
    "tcmi_spec_unload": (ctypes.c_int, [ctypes.c_void_p]),
    "tcmi_spec_set_flags": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "tcmi_spec_run_pass": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p],
    ),
    "tcmi_spec_run_pass_from": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "tcmi_spec_run_adjoint_pass": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong,
         ctypes.c_int, ctypes.c_longlong, ctypes.c_uint, ctypes.c_void_p],
    ),
    "tcmi_reconfigure_path": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p, ctypes.c_double,
         ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
         ctypes.c_int, ctypes.POINTER(ctypes.c_int)],
    ),
    "tcmi_slice_fixed": (
        ctypes.c_int,
        [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong),
         ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)],
    ),
    "tcmi_comm_load": (ctypes.c_int, [ctypes.c_char_p]),
    "tcmi_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "tcmi_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "tcmi_allreduce_sum": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]),
    "tcmi_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "tcmi_mps_gate_mix": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p],
    ),
}


def exported_symbols():
    """Names every build of libtcmi.so must export (kept in sync with include/tcmi.h)."""
    return sorted(_SIGNATURES)


class TraceAbort(Exception):
    """Raised instead of launching anything while ``backend.jit`` probes a function (tcmi/jit.py): a function
    that needs device results during the probe is not a traceable energy and keeps the plain path."""


class _Tracing(threading.local):
    """Per-thread probe flag (``TRACING[0]``): a probe in one thread must not abort device calls of another."""

    flag = False

    def __getitem__(self, i):
        return self.flag

    def __setitem__(self, i, v):
        self.flag = bool(v)


TRACING = _Tracing()


def lib():
    """Load (once) and return the ctypes handle; raises TcmiError if the HIP library is absent."""
    global _lib
    if TRACING[0]:
        raise TraceAbort("device call while tracing")
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TcmiError(
                f"HIP extension not built: {LIB_PATH} is missing. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "tensorcircuit-ng_amd/csrc`). There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().tcmi_last_error()
        raise TcmiError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
