"""``MPSCircuit``: matrix-product-state simulator front end of the hip backend.

Mirrors reference ``tensorcircuit/mpscircuit.py`` (``split_tensor`` :35-64, ``MPSCircuit`` :72-1049) and
``tensorcircuit/mps_base.py:33-175`` (``FiniteMPS.apply_two_site_gate``) — same method names, argument
meaning, centre-position bookkeeping and error messages — plus the parts of ``tensornetwork.FiniteMPS``
those rely on (``position`` QR/RQ sweeps, ``apply_one_site_gate``, ``check_canonical``,
``canonicalize``, ``bond_dimensions``).  MPS tensors are device tensors ``[left, phys, right]``; every
contraction, SVD (with the reference truncation rule) and QR runs through the C ABI
(``tcmi/linalg.py`` -> ``tcmi_cgemm / tcmi_svd_trunc_batched / tcmi_qr_batched / tcmi_mps_gate_mix``);
there is no host linear algebra and no fallback.  Qubits only (d = 2).
"""

from functools import reduce
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import cons
from . import gates as G
from . import linalg as LA

Tensor = Any


def split_rules(max_singular_values: Optional[int] = None, max_truncation_err: Optional[float] = None,
                relative: bool = False) -> Dict[str, Any]:
    """reference cons.py ``split_rules``: only the given keys are present."""
    rules: Dict[str, Any] = {}
    if max_singular_values is not None:
        rules["max_singular_values"] = max_singular_values
    if max_truncation_err is not None:
        rules["max_truncation_err"] = max_truncation_err
    if relative is not None:  # as the reference (cons.py:1337): the key is always present
        rules["relative"] = relative
    return rules


def _torch():
    import torch

    return torch


def _dtype():
    torch = _torch()
    return torch.complex64 if cons.dtypestr == "complex64" else torch.complex128


def _device():
    b = cons.backend
    if b is not None and hasattr(b, "device"):
        return b.device
    torch = _torch()
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _t(x):
    """numpy / Gate / tensor -> device tensor of the global complex dtype."""
    torch = _torch()
    if isinstance(x, G.Gate):
        x = x.tensor
    if torch.is_tensor(x):
        return x.to(device=_device(), dtype=_dtype())
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=cons.npdtype)))
    dev = _device()
    if dev.type == "cuda":
        # a pageable host-to-device copy blocks the host until the stream has drained: inside a TEBD sweep that is one
        # full wait per gate (the host could not enqueue ahead of the 2.5 ms SVD launches).  Staged through pinned
        # memory the upload is asynchronous (torch's caching host allocator keeps the staging block alive).
        return t.pin_memory().to(dev, non_blocking=True)
    return t.to(dev)


def split_tensor(tensor: Tensor, center_left: bool = True, split: Optional[Dict[str, Any]] = None):
    """reference mpscircuit.py:35-64: SVD when truncation rules are given, QR / RQ otherwise."""
    split = split or {}
    if len(split) > 0:
        u, _, vh, _ = LA.svd_trunc(tensor, absorb=1 if center_left else 2, **split)
        return u, vh
    if center_left:
        return LA.rq(tensor)
    return LA.qr(tensor)


class FiniteMPS:
    """The slice of ``tensornetwork.FiniteMPS`` + ``tensorcircuit.mps_base.FiniteMPS`` on the path."""

    def __init__(self, tensors: Sequence[Tensor], center_position: Optional[int] = None,
                 canonicalize: bool = True):
        self.tensors = [_t(t) for t in tensors]
        self.center_position = center_position
        # (site, tensor, isometry, singular values) of the last truncating two-site update whose centre
        # tensor is isometry * diag(s): moving the centre across that bond needs no QR (see position)
        self._svd_hint = None
        if canonicalize:
            if self.center_position is None:
                self.center_position = 0
            pos = self.center_position
            self.position(len(self.tensors) - 1, normalize=False)
            self.position(0, normalize=False)
            self.position(pos, normalize=True)

    def __len__(self) -> int:
        return len(self.tensors)

    @property
    def bond_dimensions(self) -> List[int]:
        return [int(self.tensors[0].shape[0])] + [int(t.shape[2]) for t in self.tensors]

    def position(self, site: int, normalize: bool = True):
        torch = _torch()
        if self.center_position is None:
            raise ValueError("BaseMPS.center_position is `None`, cannot shift `center_position`.")
        if site >= len(self.tensors) or site < 0:
            raise ValueError("site = {} not between values 0 < site < N = {}".format(site, len(self)))
        z = None
        if site == self.center_position:
            if normalize:
                z = torch.linalg.vector_norm(self.tensors[site])
                self.tensors[site] = self.tensors[site] / z
            return z
        if site > self.center_position:
            for n in range(self.center_position, site):
                t = self.tensors[n]
                l, d, r = t.shape
                hint = self._svd_hint
                if hint is not None and hint[0] == n and hint[1] is t and not normalize:
                    # t = U diag(s) with U the isometry of the SVD that produced it: Q = U, R = diag(s)
                    # (the Householder QR the reference runs here returns the same factors up to phases)
                    _, _, u_iso, sv = hint
                    self.tensors[n] = u_iso.reshape(l, d, -1)
                    nxt = self.tensors[n + 1]
                    self.tensors[n + 1] = sv.reshape(-1, 1, 1).to(nxt.dtype) * nxt
                    self._svd_hint = None
                    continue
                q, rr = LA.qr(t.reshape(l * d, r))
                self.tensors[n] = q.reshape(l, d, -1)
                nxt = self.tensors[n + 1]
                r2, d2, r3 = nxt.shape
                nxt = LA.matmul(rr, nxt.reshape(r2, d2 * r3)).reshape(-1, d2, r3)
                if normalize:
                    nxt = nxt / torch.linalg.vector_norm(rr)
                self.tensors[n + 1] = nxt
        else:
            for n in reversed(range(site + 1, self.center_position + 1)):
                t = self.tensors[n]
                l, d, r = t.shape
                rr, q = LA.rq(t.reshape(l, d * r))
                self.tensors[n] = q.reshape(-1, d, r)
                prv = self.tensors[n - 1]
                l0, d0, _ = prv.shape
                prv = LA.matmul(prv.reshape(l0 * d0, l), rr).reshape(l0, d0, -1)
                if normalize:
                    prv = prv / torch.linalg.vector_norm(rr)
                self.tensors[n - 1] = prv
        self.center_position = site
        return z

    def apply_one_site_gate(self, gate: Tensor, site: int) -> None:
        if site < 0 or site >= len(self):
            raise ValueError("site = {} is not between 0 <= site < N={}".format(site, len(self)))
        self.tensors[site] = LA.site_gate(_t(gate).reshape(2, 2), self.tensors[site])

    def apply_two_site_gate(self, gate: Tensor, site1: int, site2: int,
                            max_singular_values: Optional[int] = None,
                            max_truncation_err: Optional[float] = None,
                            center_position: Optional[int] = None, relative: bool = False) -> Tensor:
        """reference mps_base.py:33-175 (TEBD update).  Returns the discarded singular values."""
        gate = _t(gate)
        if gate.dim() != 4:
            raise ValueError("rank of gate is {} but has to be 4".format(gate.dim()))
        if site1 < 0 or site1 >= len(self) - 1:
            raise ValueError("site1 = {} is not between 0 <= site < N - 1 = {}".format(site1, len(self)))
        if site2 < 1 or site2 >= len(self):
            raise ValueError("site2 = {} is not between 1 <= site < N = {}".format(site2, len(self)))
        if site2 <= site1:
            raise ValueError("site2 = {} has to be larger than site2 = {}".format(site2, site1))
        if site2 != site1 + 1:
            raise ValueError("Found site2 ={}, site1={}. Only nearest neighbor gates are currently"
                             "supported".format(site2, site1))
        if center_position is not None and center_position not in (site1, site2):
            raise ValueError(f"center_position = {center_position} not in {(site1, site2)} ")
        if (max_singular_values or max_truncation_err) and self.center_position not in (site1, site2):
            raise ValueError(
                "center_position = {}, but gate is applied at sites {}, {}. Truncation should only be done if the "
                "gate is applied at the center position of the MPS".format(self.center_position, site1, site2))
        use_svd = (max_truncation_err is not None) or (max_singular_values is not None)
        a, b = self.tensors[site1], self.tensors[site2]
        l, d, m = a.shape
        _, _, r = b.shape
        # theta = ncon([A, B, gate], [[-1,1,2],[2,3,-4],[-2,-3,1,3]]): bond GEMM, then the 4x4 gate
        ab = LA.matmul(a.reshape(l * d, m), b.reshape(m, d * r))
        theta = LA.gate_mix(ab.reshape(-1), gate.reshape(-1), l, r).reshape(l * d, d * r)

        def set_center(site: int) -> None:
            if self.center_position is not None:
                if self.center_position in (site1, site2):
                    self.center_position = site
                else:
                    self.center_position = None

        if center_position is None:
            center_position = site1
        hint = None
        if use_svd:
            if center_position == site2:
                left, _, right, tw = LA.svd_trunc(theta, max_singular_values, max_truncation_err, relative, 2)
            else:
                u_iso, sv, right, tw = LA.svd_trunc(theta, max_singular_values, max_truncation_err, relative, 0)
                left = u_iso * sv.reshape(1, -1)
                if theta.shape[0] <= theta.shape[1]:  # row-form SVD: u is a complete isometry even at rank loss
                    hint = (u_iso, sv)
            set_center(site2 if center_position == site2 else site1)
        else:
            tw = _torch().zeros(1, dtype=theta.dtype, device=theta.device)
            if center_position == site1:
                left, right = LA.rq(theta)
                set_center(site1)
            else:
                left, right = LA.qr(theta)
                set_center(site2)
        self.tensors[site1] = left.reshape(l, d, -1)
        self.tensors[site2] = right.reshape(-1, d, r)
        self._svd_hint = None if hint is None else (site1, self.tensors[site1], hint[0], hint[1])
        return tw

    def check_canonical(self) -> Tensor:
        torch = _torch()
        devs = []
        for site, t in enumerate(self.tensors):
            l, d, r = t.shape
            if site < self.center_position:
                mat = t.reshape(l * d, r)
                m = LA.matmul(mat.conj().t().resolve_conj(), mat)
            elif site > self.center_position:
                mat = t.reshape(l, d * r)
                m = LA.matmul(mat, mat.conj().t().resolve_conj())
            else:
                continue
            devs.append(torch.linalg.matrix_norm(m - torch.eye(m.shape[0], dtype=m.dtype, device=m.device)))
        if not devs:
            return torch.zeros((), device=_device())
        return torch.linalg.vector_norm(torch.stack(devs))

    def copy(self) -> "FiniteMPS":
        r = FiniteMPS([t.clone() for t in self.tensors], canonicalize=False)
        r.center_position = self.center_position
        return r

    def conj(self) -> "FiniteMPS":
        r = FiniteMPS([t.conj().resolve_conj() for t in self.tensors], canonicalize=False)
        r.center_position = self.center_position
        return r


_SGATES = ["i", "x", "y", "z", "h", "t", "s", "td", "sd", "wroot", "cnot", "cx", "cz", "swap", "cy", "toffoli",
           "ccnot", "ccx", "fredkin", "cswap"]
_VGATES = {
    "rx": G.rx_gate, "ry": G.ry_gate, "rz": G.rz_gate, "phase": G.phase_gate, "r": G.r_gate, "u": G.u_gate,
    "cr": G.cr_gate, "iswap": G.iswap_gate, "rxx": G.rxx_gate, "ryy": G.ryy_gate, "rzz": G.rzz_gate,
    "exp": G.exponential_gate, "exp1": G.exponential_gate_unity, "su4": G.su4_gate, "any": G.any_gate,
    "unitary": G.any_gate,
}


class MPSCircuit:
    """``MPSCircuit`` class (reference mpscircuit.py:72-1049): gates are absorbed into the MPS as they are
    applied; double gates on non-adjacent qubits move through consecutive SWAPs, n >= 3 qubit gates
    through an MPO."""

    is_mps = True

    def __init__(self, nqubits: int, center_position: Optional[int] = None,
                 tensors: Optional[Sequence[Tensor]] = None, wavefunction: Optional[Tensor] = None,
                 split: Optional[Dict[str, Any]] = None, dim: Optional[int] = None) -> None:
        if dim not in (None, 2):
            raise NotImplementedError("the hip MPSCircuit handles qubits (dim = 2) only")
        self._d = 2
        self.split = split if split is not None else {}
        if wavefunction is not None:
            if tensors is not None:
                raise ValueError("tensors and wavefunction cannot be used at input simultaneously")
            tensors = self.wavefunction_to_tensors(wavefunction, split=self.split)
            if len(tensors) != nqubits:
                raise ValueError(f"number of MPS tensors ({len(tensors)}) must match nqubits ({nqubits})")
            self._mps = FiniteMPS(tensors, canonicalize=False)
            self._mps.center_position = 0
            if center_position is not None:
                self.position(center_position)
        elif tensors is not None:
            if center_position is not None:
                self._mps = FiniteMPS(tensors, canonicalize=False)
                self._mps.center_position = center_position
            else:
                self._mps = FiniteMPS(tensors, canonicalize=True, center_position=0)
        else:
            one = np.zeros((1, 2, 1), dtype=cons.npdtype)
            one[0, 0, 0] = 1.0
            self._mps = FiniteMPS([one for _ in range(nqubits)], canonicalize=False)
            self._mps.center_position = 0 if center_position is None else center_position
        self._nqubits = nqubits
        self._fidelity: Any = 1.0
        self._qir: List[Dict[str, Any]] = []

    # ---- bookkeeping (reference :199-248)
    def get_bond_dimensions(self) -> List[int]:
        return self._mps.bond_dimensions

    def get_tensors(self) -> List[Tensor]:
        return self._mps.tensors

    def get_center_position(self) -> Optional[int]:
        return self._mps.center_position

    def set_split_rules(self, split: Dict[str, Any]) -> None:
        self.split = split

    def position(self, site: int) -> None:
        self._mps.position(site, normalize=False)

    # ---- gates (reference :250-383)
    def apply_single_gate(self, gate: Any, index: int) -> None:
        if self._mps.center_position != index:
            self.position(index)
        self._mps.apply_one_site_gate(_t(gate), index)

    def apply_adjacent_double_gate(self, gate: Any, index1: int, index2: int,
                                   center_position: Optional[int] = None,
                                   split: Optional[Dict[str, Any]] = None) -> None:
        if split is None:
            split = self.split
        if index2 - index1 != 1:
            raise ValueError(f"two-qubit gate indices must be adjacent, got index1={index1}, index2={index2}")
        diff1 = abs(index1 - self._mps.center_position)
        diff2 = abs(index2 - self._mps.center_position)
        if diff1 < diff2:
            if self._mps.center_position != index1:
                self.position(index1)
        else:
            if self._mps.center_position != index2:
                self.position(index2)
        err = self._mps.apply_two_site_gate(_t(gate).reshape(2, 2, 2, 2), index1, index2,
                                            center_position=center_position, **split)
        tw2 = getattr(err, "_tcmi_tw2", None)      # the SVD kernel's own sum of squared discarded values
        self._fidelity = self._fidelity * (1 - (tw2[0] if tw2 is not None else (err.real ** 2 + err.imag ** 2).sum()))

    def consecutive_swap(self, index_from: int, index_to: int, split: Optional[Dict[str, Any]] = None) -> None:
        """Carry the site at ``index_from`` to ``index_to`` through its neighbours (reference :329-351).  A SWAP of two
        adjacent sites needs no gate arithmetic: the two-site block is formed, its physical legs are exchanged by
        relabelling, and the block is split again (``_block`` / ``_split_block``) -- the centre travels with the site."""
        if split is None:
            split = self.split
        self.position(index_from)
        step = 1 if index_to > index_from else -1
        for i in range(index_from, index_to, step):
            lo = min(i, i + step)
            theta = self._block(lo, lo + 1)                                   # [cl, 2, 2, cr]
            theta = theta.permute(0, 2, 1, 3).contiguous()
            self._split_block(theta, lo, lo + 1, center_left=(step == -1), split=split, track=True)
        assert self._mps.center_position == index_to

    def apply_double_gate(self, gate: Any, index1: int, index2: int,
                          split: Optional[Dict[str, Any]] = None) -> None:
        gate = _t(gate).reshape(2, 2, 2, 2)
        assert index1 != index2
        if index1 > index2:
            # as in the reference (:364-367) the recursion does not forward ``split``
            self.apply_double_gate(gate.permute(1, 0, 3, 2).contiguous(), index2, index1)
            return
        if split is None:
            split = self.split
        diff1 = abs(index1 - self._mps.center_position)
        diff2 = abs(index2 - self._mps.center_position)
        if diff1 < diff2:
            self.consecutive_swap(index1, index2 - 1, split=split)
            self.apply_adjacent_double_gate(gate, index2 - 1, index2, center_position=index2 - 1, split=split)
            self.consecutive_swap(index2 - 1, index1, split=split)
        else:
            self.consecutive_swap(index2, index1 + 1, split=split)
            self.apply_adjacent_double_gate(gate, index1, index1 + 1, center_position=index1 + 1, split=split)
            self.consecutive_swap(index1 + 1, index2, split=split)

    # ---- gates on three or more sites: BLOCK UPDATE (reference :386-668 reaches the same states through an MPO) -------
    # The reference turns an n-qubit gate into an MPO by a chain of SVDs, multiplies it into the MPS site by site (one QR
    # per site on a bond of chi x D) and then compresses bond by bond.  Here the unit of work is the GPU's: the w sites the
    # gate spans are contracted into ONE block [chi_l, 2^w, chi_r] (w - 1 GEMMs, tcmi_cgemm), the gate is ONE GEMM on the
    # block's physical legs, and the block is split back by w - 1 (truncated) factorisations that peel one site at a time
    # off the far end -- the same cuts, truncated in the same order, as the reference's compression sweep, so the states
    # agree; no MPO, no intermediate bond of chi x D.  Spans too wide for one block are first gathered with
    # ``consecutive_swap``.  ``gate_to_MPO`` / ``apply_MPO`` / ``reduce_dimension`` stay as public entry points on top of
    # the same two primitives.
    BLOCK_MAX_ELEMENTS = 1 << 26          # chi_l * 2^w * chi_r of one block (512 MiB of complex64)
    BLOCK_MAX_SITES = 12                  # ... and 4^w elements of the operator on its physical legs

    def _block(self, left: int, right: int) -> Tensor:
        """Sites left..right contracted into [chi_l, 2, ..., 2, chi_r] (the centre must sit inside)."""
        ts = self._mps.tensors
        theta = ts[left]
        cl = int(theta.shape[0])
        for s_ in range(left + 1, right + 1):
            t = ts[s_]
            theta = LA.matmul(theta.reshape(-1, t.shape[0]), t.reshape(t.shape[0], -1))
        w = right - left + 1
        return theta.reshape((cl,) + (2,) * w + (int(ts[right].shape[-1]),))

    def _cut(self, rem: Tensor, center_left: bool, split: Optional[Dict[str, Any]], track: bool):
        """One cut of a block: (left factor, right factor) with the weights on the centre's side.  ``track``: the
        discarded weight of a truncating cut enters the fidelity estimate (as the reference's two-site updates do)."""
        if not (track and split):
            return split_tensor(rem, center_left=center_left, split=split)
        u, _, vh, rest = LA.svd_trunc(rem, absorb=1 if center_left else 2, **split)
        tw2 = getattr(rest, "_tcmi_tw2", None)
        self._fidelity = self._fidelity * (1 - (tw2[0] if tw2 is not None else (rest.real ** 2 + rest.imag ** 2).sum()))
        return u, vh

    def _split_block(self, theta: Tensor, left: int, right: int, center_left: bool,
                     split: Optional[Dict[str, Any]] = None, track: bool = False) -> None:
        """Write the block back as sites left..right.  ``center_left``: sites are peeled off the RIGHT end (each an
        isometry from the right, the remainder keeps the weights) and the centre ends on ``left``; otherwise mirrored.
        With truncation rules every cut is a truncated SVD of the exact remainder, else QR / RQ (``split_tensor``)."""
        ts = self._mps.tensors
        cl, cr = int(theta.shape[0]), int(theta.shape[-1])
        if center_left:
            rem = theta.reshape(-1, 2 * cr)
            for site in range(right, left, -1):
                keep_l, iso_r = self._cut(rem, True, split, track)
                k = int(iso_r.shape[0])
                ts[site] = iso_r.reshape(k, 2, -1)
                rem = keep_l.reshape(-1, 2 * k)
            ts[left] = rem.reshape(cl, 2, -1).contiguous()
            self._mps.center_position = left
        else:
            rem = theta.reshape(cl * 2, -1)
            for site in range(left, right):
                iso_l, keep_r = self._cut(rem, False, split, track)
                k = int(iso_l.shape[1])
                ts[site] = iso_l.reshape(-1, 2, k)
                rem = keep_r.reshape(k * 2, -1)
            ts[right] = rem.reshape(-1, 2, cr).contiguous()
            self._mps.center_position = right
        self._mps._svd_hint = None

    def _block_fits(self, left: int, right: int, extra_bond: int = 1) -> bool:
        w = right - left + 1
        ts = self._mps.tensors
        return w <= self.BLOCK_MAX_SITES and int(ts[left].shape[0]) * int(ts[right].shape[-1]) * (1 << w) * extra_bond \
            <= self.BLOCK_MAX_ELEMENTS

    def _block_update(self, op: Tensor, sites: Sequence[int], left: int, right: int, center_left: bool,
                      split: Optional[Dict[str, Any]] = None) -> None:
        """``op`` [2^k, 2^k] (rows = outputs) acts on the sorted ``sites`` inside the block left..right."""
        torch = _torch()
        self.position(left if center_left else right)
        theta = self._block(left, right)
        w, k = right - left + 1, len(sites)
        legs = [1 + (q - left) for q in sites]
        rest = [a for a in range(w + 2) if a not in legs]
        perm = legs + rest
        moved = theta.permute(perm).contiguous()
        shp = moved.shape
        moved = LA.matmul(_t(op).reshape(2**k, 2**k), moved.reshape(2**k, -1)).reshape(shp)
        inv = [0] * len(perm)
        for i_, a in enumerate(perm):
            inv[a] = i_
        theta = moved.permute(inv).contiguous()
        del moved
        # end1 = where the reference's sweep starts: its compression runs back towards it, the centre ends there
        self._split_block(theta, left, right, center_left=center_left, split=split)

    @classmethod
    def gate_to_MPO(cls, gate: Any, *index: int) -> Tuple[List[Tensor], int]:
        """An exact MPO of the gate on the strictly increasing sites ``index`` (identity tensors on the sites in between):
        tensors [D_l, 2 (out), 2 (in), D_r] and the leftmost site.  No decomposition is computed: the sites left of the
        middle one hand their (out, in) leg pairs to the right through the bond, the sites right of it to the left, and
        the middle site holds the gate -- bonds 4, 16, ... , 4^floor(k/2), the worst case of an SVD-built MPO."""
        torch = _torch()
        if len(index) == 0:
            raise ValueError("`index` must contain at least one site.")
        if not all(index[i] < index[i + 1] for i in range(len(index) - 1)):
            raise ValueError("`index` must be strictly increasing.")
        k = len(index)
        g = _t(gate).reshape((2,) * (2 * k))
        m = k // 2                                # the site that holds the gate
        # legs (o_1..o_k, i_1..i_k) -> (pairs left of m | o_m, i_m | pairs right of m)
        order = []
        for j in list(range(m)) + [m] + list(range(m + 1, k)):
            order += [j, k + j]
        g = g.permute(order).contiguous()
        eye4 = torch.eye(4, dtype=g.dtype, device=g.device)
        mains: List[Tensor] = []
        for j in range(k):
            if j < m:      # [4^j, o, i, 4^(j+1)]: appends its (o, i) pair to the bond
                dl = 4**j
                t = torch.einsum("ab,cd->acbd", torch.eye(dl, dtype=g.dtype, device=g.device), eye4)     # [dl, 4, dl, 4]
                mains.append(t.reshape(dl, 2, 2, dl * 4))
            elif j == m:
                mains.append(g.reshape(4**m, 2, 2, 4 ** (k - 1 - m)))
            else:          # [4^(k-j), o, i, 4^(k-1-j)]: takes the first (o, i) pair off the bond
                dr = 4 ** (k - 1 - j)
                t = torch.einsum("cd,ab->cadb", eye4, torch.eye(dr, dtype=g.dtype, device=g.device))     # [4, dr, 4, dr]
                mains.append(t.reshape(4 * dr, 2, 2, dr))
        tensors: List[Tensor] = []
        for j, site in enumerate(index):
            if j > 0:
                bond = int(tensors[-1].shape[-1])
                for _ in range(index[j - 1] + 1, site):
                    fill = torch.einsum("ab,cd->acdb", torch.eye(bond, dtype=g.dtype, device=g.device),
                                        torch.eye(2, dtype=g.dtype, device=g.device))
                    tensors.append(fill.contiguous())
            tensors.append(mains[j].contiguous())
        return tensors, int(index[0])

    @classmethod
    def MPO_to_gate(cls, tensors: Sequence[Tensor]) -> Any:
        """reference :464-486: the MPO contracted back into a gate tensor with legs (out_1..out_w, in_1..in_w)."""
        op = cls._mpo_operator([_t(t) for t in tensors])
        w = len(tensors)
        return G.Gate(op.reshape((2,) * (2 * w)))

    @staticmethod
    def _mpo_operator(tensors: Sequence[Tensor]) -> Tensor:
        """[2^w, 2^w] matrix (rows = outputs) of an MPO whose outer bonds have dimension 1."""
        acc = tensors[0]
        if int(acc.shape[0]) != 1 or int(tensors[-1].shape[-1]) != 1:
            raise ValueError("MPO with open outer bonds")
        acc = acc.reshape(2, 2, -1)                               # [O, I, bond]
        for t in tensors[1:]:
            dl, _, _, dr = t.shape
            nxt = LA.matmul(acc.reshape(-1, dl), t.reshape(dl, -1))                  # [O I, o i dr]
            o_, i_ = int(acc.shape[0]), int(acc.shape[1])
            acc = nxt.reshape(o_, i_, 2, 2, dr).permute(0, 2, 1, 3, 4).contiguous().reshape(o_ * 2, i_ * 2, dr)
        return acc.reshape(acc.shape[0], acc.shape[1])

    @classmethod
    def reduce_tensor_dimension(cls, tensor_left: Tensor, tensor_right: Tensor, center_left: bool = True,
                                split: Optional[Dict[str, Any]] = None) -> Tuple[Tensor, Tensor]:
        """Two neighbouring site tensors re-split across their bond under the truncation rules (reference :488-520)."""
        a, b = _t(tensor_left), _t(tensor_right)
        pair = LA.matmul(a.reshape(-1, a.shape[-1]), b.reshape(b.shape[0], -1))      # [cl * 2, 2 * cr]
        lft, rgt = split_tensor(pair, center_left=center_left, split=split or {})
        return lft.reshape(a.shape[0], a.shape[1], -1), rgt.reshape(-1, b.shape[1], b.shape[2])

    def reduce_dimension(self, index_left: int, center_left: bool = True,
                         split: Optional[Dict[str, Any]] = None) -> None:
        """Compress the bond (index_left, index_left + 1); the centre must sit on one of the two sites (reference :522-550)."""
        if split is None:
            split = self.split
        if self._mps.center_position not in (index_left, index_left + 1):
            raise ValueError("reduce_dimension: the centre must be on one of the two sites")
        self._split_block(self._block(index_left, index_left + 1), index_left, index_left + 1,
                          center_left=center_left, split=split)

    def apply_MPO(self, tensors: Sequence[Tensor], index_left: int, center_left: bool = True,
                  split: Optional[Dict[str, Any]] = None) -> None:
        """An MPO on the sites index_left .. index_left + len(tensors) - 1 (reference :552-634); ``center_left``: the
        sweep starts -- and the centre ends -- on the left end.  The MPO is contracted into its operator on the block's
        physical legs and applied as one block update."""
        if split is None:
            split = self.split
        tensors = [_t(t) for t in tensors]
        w = len(tensors)
        right = index_left + w - 1
        if not self._block_fits(index_left, right):
            raise NotImplementedError(f"Backend 'hip' has not implemented apply_MPO over {w} sites at these bond dimensions "
                                      f"(one block of more than {self.BLOCK_MAX_ELEMENTS} elements)")
        self._block_update(self._mpo_operator(tensors), list(range(index_left, right + 1)), index_left, right,
                           center_left=center_left, split=split)

    def apply_nqubit_gate(self, gate: Any, *index: int, split: Optional[Dict[str, Any]] = None) -> None:
        """A gate on three or more sites (reference :636-668), by one block update over the sites it spans; a span too
        wide for one block is gathered next to its first site with ``consecutive_swap`` and scattered again afterwards."""
        if split is None:
            split = self.split
        k = len(index)
        order = sorted(range(k), key=lambda j: index[j])
        sites = [int(index[j]) for j in order]
        op = _t(gate).reshape((2,) * (2 * k)).permute(order + [k + j for j in order]).contiguous().reshape(2**k, 2**k)
        left, right = sites[0], sites[-1]
        if self._block_fits(left, right):
            near_left = abs(left - self._mps.center_position) < abs(right - self._mps.center_position)
            self._block_update(op, sites, left, right, center_left=near_left, split=split)
            return
        home = list(sites)
        for j in range(1, k):                       # gather: site j of the gate next to site j - 1
            self.consecutive_swap(sites[j], left + j, split=split)
        self._block_update(op, [left + j for j in range(k)], left, left + k - 1, center_left=True, split=split)
        for j in range(k - 1, 0, -1):               # scatter, last moved first
            self.consecutive_swap(left + j, home[j], split=split)

    def apply_general_gate(self, gate: Any, *index: int, name: Optional[str] = None,
                           split: Optional[Dict[str, Any]] = None, mpo: bool = False,
                           diagonal: bool = False, ir_dict: Optional[Dict[str, Any]] = None) -> None:
        """reference :670-724."""
        if split is None:
            split = self.split
        self._qir.append({"gate": gate, "index": index, "name": name or "", "split": split, "mpo": mpo})
        if len(index) != len(set(index)):
            raise ValueError(f"gate index {list(index)} has duplicate qubits; each qubit may appear at most once")
        if mpo is not False:
            raise NotImplementedError("MPO not implemented for MPS")
        if diagonal is not False:
            raise NotImplementedError("diagonal hyperedge not implemented for MPS")
        noe = len(index)
        if noe == 1:
            self.apply_single_gate(gate, *index)
        elif noe == 2:
            self.apply_double_gate(gate, *index, split=split)
        else:
            self.apply_nqubit_gate(gate, *index, split=split)

    apply = apply_general_gate

    def mid_measurement(self, index: int, keep: int = 0) -> None:
        """reference :726-744: z-basis projector, state left unnormalised."""
        gate = np.zeros((2, 2), dtype=cons.npdtype)
        gate[keep, keep] = 1.0
        self.apply_single_gate(gate, index)

    def is_valid(self) -> bool:
        mps = self._mps
        if len(mps) != self._nqubits:
            return False
        for i in range(self._nqubits):
            if mps.tensors[i].dim() != 3:
                return False
        for i in range(self._nqubits - 1):
            if mps.tensors[i].shape[-1] != mps.tensors[i + 1].shape[0]:
                return False
        return True

    # ---- outputs (reference :765-1049)
    @classmethod
    def wavefunction_to_tensors(cls, wavefunction: Tensor, dim_phys: Optional[int] = None, norm: bool = True,
                                split: Optional[Dict[str, Any]] = None) -> List[Tensor]:
        dim_phys = dim_phys if dim_phys is not None else 2
        split = split or {}
        w = _t(wavefunction).reshape(-1, 1)
        n_tensors = int(np.round(np.log(w.shape[0]) / np.log(dim_phys)))
        tensors: List[Tensor] = []
        for _ in range(n_tensors):
            nright = w.shape[1]
            w = w.reshape(-1, nright * dim_phys)
            w, q = split_tensor(w, center_left=True, split=split)
            tensors.insert(0, q.reshape(-1, dim_phys, nright))
        if tuple(w.shape) != (1, 1):
            raise ValueError(f"expected scalar wavefunction of shape (1, 1), got {tuple(w.shape)}")
        if not norm:
            tensors[0] = tensors[0] * w[0, 0]
        return tensors

    def wavefunction(self, form: str = "default") -> Tensor:
        torch = _torch()
        LA.svd_health_check(_device())
        result = torch.ones((1, 1), dtype=_dtype(), device=_device())
        for t in self._mps.tensors:
            j, b, k = t.shape
            result = LA.matmul(result, t.reshape(j, b * k)).reshape(-1, k)
        shape = {"default": [-1], "ket": [-1, 1], "bra": [1, -1]}[form]
        return result.reshape(shape)

    state = wavefunction

    def copy_without_tensor(self) -> "MPSCircuit":
        r = MPSCircuit.__new__(MPSCircuit)
        r._d = self._d
        r.split = dict(self.split)
        r._nqubits = self._nqubits
        r._fidelity = self._fidelity
        r._qir = list(self._qir)
        return r

    def copy(self) -> "MPSCircuit":
        r = self.copy_without_tensor()
        r._mps = self._mps.copy()
        return r

    def conj(self) -> "MPSCircuit":
        r = self.copy_without_tensor()
        r._mps = self._mps.conj()
        return r

    def get_norm(self) -> Tensor:
        return _torch().linalg.vector_norm(self._mps.tensors[self._mps.center_position])

    def normalize(self) -> None:
        c = self._mps.center_position
        self._mps.tensors[c] = self._mps.tensors[c] / self.get_norm()

    def amplitude(self, l: str) -> Tensor:
        assert len(l) == self._nqubits
        LA.svd_health_check(_device())
        mats = [self._mps.tensors[i][:, int(ch), :] for i, ch in enumerate(l)]
        return reduce(LA.matmul, mats)[0, 0]

    def proj_with_mps(self, other: "MPSCircuit", conj: bool = True) -> Tensor:
        """<other|self> (reference :905-939), contracted from the right end."""
        bra = other.conj() if conj else other.copy()
        ket = self.copy()
        assert bra._nqubits == ket._nqubits
        for _ in range(bra._nqubits, 1, -1):
            bra_b = bra._mps.tensors[-1]
            ket_a, ket_b = ket._mps.tensors[-2:]
            k = bra_b.shape[0]
            l = ket_b.shape[0]
            proj_b = LA.matmul(bra_b.reshape(k, -1), ket_b.reshape(l, -1).t())        # "kbm,lbm->kl"
            j, a, _ = ket_a.shape
            new_ka = LA.matmul(ket_a.reshape(j * a, l), proj_b.t()).reshape(j, a, k)  # "jal,kl->jak"
            bra._mps.tensors.pop()
            ket._mps.tensors.pop()
            ket._mps.tensors[-1] = new_ka
        return (bra._mps.tensors[0] * ket._mps.tensors[0]).sum()

    def slice(self, begin: int, end: int) -> "MPSCircuit":
        nq = end - begin + 1
        tensors = [t.clone() for t in self._mps.tensors[begin:end + 1]]
        cp = None
        c = self._mps.center_position
        if c is not None and begin <= c <= end:
            cp = c - begin
        return self.__class__(nq, tensors=tensors, center_position=cp, split=dict(self.split))

    def expectation(self, *ops: Tuple[Any, List[int]], reuse: bool = True, other: Optional["MPSCircuit"] = None,
                    conj: bool = True, normalize: bool = False, split: Optional[Dict[str, Any]] = None,
                    **kws: Any) -> Tensor:
        if split is None:
            split = {}
        LA.svd_health_check(_device())    # no result is handed out from factors of a timed-out decomposition
        ops2 = [[op[0], [op[1]] if isinstance(op[1], int) else list(op[1])] for op in ops]
        all_sites = np.concatenate([op[1] for op in ops2])
        if other is None:
            site_begin, site_end = int(np.min(all_sites)), int(np.max(all_sites))
            if self._mps.center_position < site_begin:
                self.position(site_begin)
            elif self._mps.center_position > site_end:
                self.position(site_end)
        else:
            assert isinstance(other, MPSCircuit), "the bra has to be a MPSCircuit"
        mps = self.copy()
        mps.set_split_rules(split)
        for gate, index in ops2:
            mps.apply(gate, *index)
        if other is None:
            ket = mps.slice(site_begin, site_end)
            bra = self.slice(site_begin, site_end)
        else:
            ket, bra = mps, other
        value = ket.proj_with_mps(bra, conj=conj)
        if normalize:
            n1 = self.get_norm()
            n2 = n1 if other is None else other.get_norm()
            value = value / (n1 * n2).sqrt()
        return value

    # ---- sampling and reduced states (reference :1061-1288) -------------------------------------------
    def measure(self, *index: int, with_prob: bool = False, status: Optional[Tensor] = None):
        """z-basis measurement of the given sites, one after the other on a copy of the MPS (the centre is
        moved to the site, its tensor projected on the outcome).  ``status`` (one uniform number per site)
        makes it deterministic, with the ``backend.probability_sample`` rule of the reference
        (abstract_backend.py:1849-1861): outcome = searchsorted(cumsum(p), status_k)."""
        torch = _torch()
        mps = self.copy()
        rdt = torch.float32 if cons.rdtypestr == "float32" else torch.float64
        if status is None:
            status = cons.backend.implicit_randu(shape=[len(index)])
        st = cons.backend.numpy(cons.backend.convert_to_tensor(status)).reshape(-1)
        p = torch.ones((), dtype=rdt, device=_device())
        outcomes = []
        for k, site in enumerate(index):
            mps.position(site)
            t = mps._mps.tensors[site]
            ps = (t.real ** 2 + t.imag ** 2).sum(dim=(0, 2)).to(rdt)
            ps = ps / ps.sum()
            cum = torch.cumsum(ps, 0)
            r = cum[-1] * float(st[k])
            outcome = int(torch.searchsorted(cum, r).clamp(max=1).item())
            p = p * ps[outcome]
            mps._mps.tensors[site] = t[:, outcome, :].unsqueeze(1).contiguous()
            outcomes.append(outcome)
        sample = torch.tensor(outcomes, dtype=rdt, device=_device())
        return (sample, p) if with_prob else (sample, -1.0)

    def reduced_density_matrix(self, subsystem_to_keep: Optional[Sequence[int]] = None, *,
                               subsystems_to_trace_out: Optional[Sequence[int]] = None) -> Tensor:
        """rho over ``subsystem_to_keep`` (indices in the caller's order).  Contiguous subsystems use the
        canonical form (centre moved inside: the environments are identities); otherwise the whole chain
        is contracted, tracing the other sites.  Every contraction is a ``tcmi_cgemm``."""
        torch = _torch()
        n = self._nqubits
        if (subsystem_to_keep is None) == (subsystems_to_trace_out is None):
            raise ValueError("MPSCircuit.reduced_density_matrix: give exactly one of subsystem_to_keep / "
                             "subsystems_to_trace_out")
        if subsystem_to_keep is None:
            tr = {int(i) % n for i in subsystems_to_trace_out}
            keep = [i for i in range(n) if i not in tr]
        else:
            keep = [int(i) % n for i in subsystem_to_keep]
        if not keep:
            raise ValueError("Must keep at least one qubit index.")
        ks = sorted(keep)
        contiguous = all(ks[i + 1] - ks[i] == 1 for i in range(len(ks) - 1))
        if contiguous:
            cp = self._mps.center_position
            if cp is None or cp < ks[0]:
                self.position(ks[0])
            elif cp > ks[-1]:
                self.position(ks[-1])
            sites = ks
        else:
            sites = list(range(n))
        # R[x, y, j, k]: x / y = kept physical indices of ket / bra, j / k = open right bonds
        first = self._mps.tensors[sites[0]]
        l0 = first.shape[0]
        eye = torch.eye(l0, dtype=first.dtype, device=first.device)
        R = eye.reshape(1, 1, l0, l0)
        for i in sites:
            t = self._mps.tensors[i]
            tc_ = t.conj().resolve_conj()
            if i in ks:
                tmp = LA.einsum2("xyjk,jcl->xykcl", R, t)
                R = LA.einsum2("xykcl,kdm->xcydlm", tmp, tc_)
                x, c, y, d, l, m = R.shape
                R = R.reshape(x * c, y * d, l, m)
            else:
                tmp = LA.einsum2("xyjk,jcl->xykcl", R, t)
                x, y, k, c, l = tmp.shape
                R = LA.matmul(tmp.permute(0, 1, 4, 2, 3).reshape(x * y * l, k * c),
                              tc_.reshape(k * c, -1)).reshape(x, y, l, -1)
        rho = torch.diagonal(R, dim1=2, dim2=3).sum(-1)
        nk = len(keep)
        if keep != ks:
            order = {v: i for i, v in enumerate(ks)}
            perm = [order[q] for q in keep] + [nk + order[q] for q in keep]
            rho = rho.reshape([2] * (2 * nk)).permute(perm).reshape(2 ** nk, 2 ** nk)
        return rho.contiguous()

    def sample(self, batch: Optional[int] = None, allow_state: bool = False, readout_error: Any = None,
               format: Optional[str] = None, random_generator: Any = None, status: Optional[Tensor] = None,
               jittable: bool = True) -> Any:
        """reference :1241-1288: ``batch`` independent full measurements; formats of ``quantum.sample2all``."""
        from .quantum import sample2all

        torch = _torch()
        if allow_state:
            raise ValueError("MPSCircuit.sample does not support allow_state=True")
        n = self._nqubits
        if batch is None:
            if status is None:
                status = cons.backend.implicit_randu(shape=[n])
            r = self.measure(*range(n), status=status)[0]
            if format is None:
                return r
            ch = r.reshape(1, -1).to(torch.int32)
        else:
            if status is None:
                status = cons.backend.implicit_randu(shape=[batch, n])
            st = cons.backend.convert_to_tensor(status)
            r = torch.stack([self.measure(*range(n), status=st[i])[0] for i in range(batch)])
            if format is None:
                return r
            ch = r.to(torch.int32)
        return sample2all(ch, n, format=format)

    def expectation_ps(self, x: Optional[Sequence[int]] = None, y: Optional[Sequence[int]] = None,
                       z: Optional[Sequence[int]] = None, **kws: Any) -> Tensor:
        """reference abstractcircuit.py:1523-1603."""
        ops = []
        for mk, idx in ((G.x, x), (G.y, y), (G.z, z)):
            for i in (idx or []):
                ops.append((mk(), [i]))
        return self.expectation(*ops, **kws)


def _make_sgate(name):
    def f(self, *index, **kw):
        self.apply_general_gate(getattr(G, name)(), *index, name=name, **kw)

    f.__name__ = name
    return f


_SPECS = {
    "rx": G.rx_spec, "ry": G.ry_spec, "rz": G.rz_spec, "phase": G.phase_spec, "r": G.r_spec, "u": G.u_spec,
    "cr": G.cr_spec, "iswap": G.iswap_spec, "exp1": G.exp1_spec,
    "rxx": lambda theta=0.0: G.exp1_spec(G._xx_matrix, theta, half=True),
    "ryy": lambda theta=0.0: G.exp1_spec(G._yy_matrix, theta, half=True),
    "rzz": lambda theta=0.0: G.exp1_spec(G._zz_matrix, theta, half=True),
}


def _tensor_gate(name: str, kw: Dict[str, Any]) -> Tensor:
    """Gate matrix as a device tensor when a parameter is a tensor (so that it stays on the autograd
    tape): the same ``C0 + cos(a) C1 + sin(a) C2`` factors the state-vector path records."""
    torch = _torch()
    if name in ("any", "unitary"):
        return _t(kw["unitary"])
    if name == "exp":
        u = _t(kw["unitary"])
        d = int(round(np.sqrt(u.numel())))
        return torch.linalg.matrix_exp(-1j * _t(kw["theta"]) * u.reshape(d, d))
    if name == "su4":
        gen = torch.einsum("i,iab->ab", _t(kw["theta"]).reshape(15), _t(G._su4_generators))
        return torch.linalg.matrix_exp(-1j * gen)
    mat = None
    for spec in _SPECS[name](**kw):
        th = spec.theta
        if torch.is_tensor(th):
            a = spec.scale * th.to(_device()).real + spec.offset
            m = _t(spec.c0) + torch.cos(a) * _t(spec.c1) + torch.sin(a) * _t(spec.c2)
        else:
            m = _t(spec.matrix())
        mat = m if mat is None else LA.matmul(m, mat)
    return mat


def _make_vgate(name, factory):
    def f(self, *index, split=None, **kw):
        if any(_torch().is_tensor(v) for v in kw.values()):
            gate = _tensor_gate(name, kw)
        else:
            gate = factory(**kw)
        self.apply_general_gate(gate, *index, name=name, split=split)

    f.__name__ = name
    return f


for _n in _SGATES:
    setattr(MPSCircuit, _n, _make_sgate(_n))
    setattr(MPSCircuit, _n.upper(), _make_sgate(_n))
for _n, _f in _VGATES.items():
    setattr(MPSCircuit, _n, _make_vgate(_n, _f))
    setattr(MPSCircuit, _n.upper(), _make_vgate(_n, _f))
