"""Switches of the package, in one place.

**Environment switches** (supported, documented in README.md "Switches"; read where they are used):

=========================  =========================================================================================
TCMI_LIB                   path of libtcmi.so (default: tensorcircuit-ng_amd/csrc/libtcmi.so)
TCMI_SPECIALIZE            plan-specialised kernels: ``auto`` (default) / ``1`` (compile at first use) / ``0`` (never)
TCMI_SPEC_CACHE            directory of the generated code objects (default csrc/plancache, then the per-user cache)
TCMI_SPEC_HOT              ``auto``: calls of a plan before its missing kernels are compiled (3)
TCMI_SPEC_MIN_N            ``auto``: plans below this many qubits are never compiled (22)
TCMI_SPEC_KEEP             keep the generated .hip next to the code object
TCMI_SPEC_LOCK_STALE_S     age after which another process's compile lock is taken over (600)
TCMI_JOIN_GEMM             ``split`` (default: the two-piece f16 join when the cut bounds its halves, else the three-piece
                           bf16 join) / ``bf16`` (the three-piece join always) / ``f32`` (exact-f32 MFMA kernel)
TCMI_CUT_DEFER             ``1`` (default): the last crossing gate of a cut is applied by the join; ``0``: every gate a bond
TCMI_SPARSE_START          ``0``: every tile live (no live-tile passes, zero fill)
TCMI_PAULI_FOLD            ``0``: the Pauli-sum cotangent is not born in the sweep (tile passes instead)
TCMI_TN_GRAPH              ``0``: the contraction engine launches eagerly (no hipGraph replay)
TCMI_TN_TRACE              ``0``: DistributedContractor calls the node function every time (no recipe)
TCMI_TN_VJP                ``0``: sliced value_and_grad on torch's tape instead of the hand-written sweep
TCMI_TN_SHARD_INV          ``0``: every rank computes all slice-invariant subtrees
TCMI_TN_SEARCH_SHARD       ``0``: every rank runs the whole path search (no collective in the constructor)
TCMI_TREE_CACHE            ``0``: searched contraction trees are not kept in / loaded from the cache directory
TCMI_SVD_PRECOND           ``1``: QR-preconditioned Jacobi SVD (graded spectra)
TCMI_CHECK_SVD             ``1``: verify every truncated SVD against torch.linalg (debugging)
TCMI_KNOBS                 experiment knobs, see below
=========================  =========================================================================================

**Experiment knobs** (unsupported; for the measuring scripts under scripts/): ``TCMI_KNOBS="name=value,name=value"`` or
``tcmi._knobs.VALUES[name] = "value"`` from a script or a test.  They select among RESULT-PRESERVING alternatives the
cost models would not pick (tile shapes, pass caps, stream counts, kernel routes kept as fallbacks); every one is read
with :func:`knob` at its point of use, so a grep for ``knob("`` lists them all.  Kernels that compute wrong results in
order to be timed do not exist in the package (scripts/ build them into libtcmi_probe.so or patch the emitter themselves).
"""

import os
from typing import Dict, Optional

VALUES: Dict[str, str] = {}
for _kv in filter(None, os.environ.get("TCMI_KNOBS", "").split(",")):
    _k, _, _v = _kv.partition("=")
    VALUES[_k.strip().lower()] = _v.strip() if _v else "1"


def knob(name: str, default: Optional[str] = None) -> Optional[str]:
    """The string value of experiment knob ``name`` (as an environment variable would give it) or ``default``."""
    return VALUES.get(name, default)
