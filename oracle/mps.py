"""CPU oracle for the MPS / TEBD row of the hot path (SURVEY.md §8a last row, config 5).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  numpy restatement of

* ``tensorcircuit/mpscircuit.py`` — ``split_tensor`` (:35-64), ``MPSCircuit`` (:72-1049):
  construction, ``position``, single / adjacent-double / swapped double gates (:250-383),
  ``gate_to_MPO`` (:386-451), ``reduce_dimension`` (:481-535), ``apply_MPO`` (:537-634),
  ``apply_nqubit_gate`` (:636-668), ``wavefunction_to_tensors`` (:765-807), ``wavefunction`` (:809-832),
  ``get_norm/normalize/amplitude`` (:883-903), ``proj_with_mps`` (:905-939), ``slice`` (:941-963),
  ``expectation`` (:965-1049);
* ``tensorcircuit/mps_base.py:33-175`` — ``FiniteMPS.apply_two_site_gate`` (TEBD update with the
  centre-position fix);
* the truncation rule of ``backend.svd`` as stated in-tree at ``backends/jax_backend.py:62-112``;
* the parts of the third-party ``tensornetwork.FiniteMPS`` the above rely on (package absent from
  ``/root/reference``; unpinned, historical pin ``tensornetwork-ng==0.5.0``): ``position`` by QR / RQ
  sweeps, ``apply_one_site_gate`` (``ncon([gate, A], [[-2, 1], [-1, 1, -3]])``), ``check_canonical``,
  ``canonicalize``, ``bond_dimensions`` — restated from its published behaviour.

Pinned by the reference's own known-answer test ``tests/test_mpscircuit.py:22-60,109-131,380``
(N=8, D=6: real fidelity 0.902663090851, estimated fidelity 0.910305380327) in
``tests/test_oracle_mps.py``.
"""

from functools import reduce
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import gates as G

DTYPE = np.complex128


# ----------------------------------------------------------------------------- decompositions
def svd_trunc(mat, max_singular_values=None, max_truncation_err=None, relative=False):
    """backend.svd(pivot_axis=1) with the truncation rule of jax_backend.py:62-112."""
    u, s, vh = np.linalg.svd(mat, full_matrices=False)
    if max_singular_values is None:
        max_singular_values = s.size
    if max_truncation_err is not None:
        trunc_errs = np.sqrt(np.cumsum(np.square(s[::-1])))
        abs_err = max_truncation_err * s[0] if relative else max_truncation_err
        num_err = int(np.count_nonzero(trunc_errs > abs_err))
    else:
        num_err = max_singular_values
    keep = min(max_singular_values, num_err)
    return u[:, :keep], s[:keep].astype(mat.dtype), vh[:keep, :], s[keep:].astype(mat.dtype)


def qr(mat):
    return np.linalg.qr(mat)


def rq(mat):
    q, r = np.linalg.qr(mat.conj().T)
    return r.conj().T, q.conj().T


def split_tensor(mat, center_left=True, split=None):
    """mpscircuit.py:35-64."""
    split = split or {}
    if len(split) > 0:
        u, s, vh, _ = svd_trunc(mat, **split)
        if center_left:
            return u * s.reshape(1, -1), vh
        return u, s.reshape(-1, 1) * vh
    if center_left:
        return rq(mat)
    return qr(mat)


# ----------------------------------------------------------------------------- FiniteMPS
class FiniteMPS:
    """The slice of tensornetwork.FiniteMPS + tensorcircuit.mps_base.FiniteMPS the path uses."""

    def __init__(self, tensors, center_position=None, canonicalize=True):
        self.tensors = [np.asarray(t) for t in tensors]
        self.center_position = center_position
        if canonicalize:
            if self.center_position is None:
                self.center_position = 0
            pos = self.center_position
            self.position(len(self.tensors) - 1, normalize=False)
            self.position(0, normalize=False)
            self.position(pos, normalize=True)

    def __len__(self):
        return len(self.tensors)

    @property
    def bond_dimensions(self):
        return [self.tensors[0].shape[0]] + [t.shape[2] for t in self.tensors]

    def position(self, site, normalize=True):
        if self.center_position is None:
            raise ValueError("BaseMPS.center_position is `None`, cannot shift `center_position`.")
        if site == self.center_position:
            z = np.linalg.norm(self.tensors[site])
            if normalize:
                self.tensors[site] = self.tensors[site] / z
            return z
        if site > self.center_position:
            for n in range(self.center_position, site):
                t = self.tensors[n]
                l, d, r = t.shape
                q, rr = qr(t.reshape(l * d, r))
                self.tensors[n] = q.reshape(l, d, -1)
                self.tensors[n + 1] = np.tensordot(rr, self.tensors[n + 1], axes=(1, 0))
                z = np.linalg.norm(rr)
                if normalize:
                    self.tensors[n + 1] = self.tensors[n + 1] / z
            self.center_position = site
        else:
            for n in reversed(range(site + 1, self.center_position + 1)):
                t = self.tensors[n]
                l, d, r = t.shape
                rr, q = rq(t.reshape(l, d * r))
                self.tensors[n] = q.reshape(-1, d, r)
                self.tensors[n - 1] = np.tensordot(self.tensors[n - 1], rr, axes=(2, 0))
                z = np.linalg.norm(rr)
                if normalize:
                    self.tensors[n - 1] = self.tensors[n - 1] / z
            self.center_position = site
        return z

    def apply_one_site_gate(self, gate, site):
        # ncon([gate, A], [[-2, 1], [-1, 1, -3]]): gate[out, in]
        self.tensors[site] = np.einsum("ab,lbr->lar", gate, self.tensors[site])

    def apply_two_site_gate(self, gate, site1, site2, max_singular_values=None,
                            max_truncation_err=None, center_position=None, relative=False):
        """mps_base.py:33-175."""
        if gate.ndim != 4:
            raise ValueError("rank of gate is {} but has to be 4".format(gate.ndim))
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
            raise ValueError("center_position = {}, but gate is applied at sites {}, {}. Truncation should "
                             "only be done if the gate is applied at the center position of the MPS".format(
                                 self.center_position, site1, site2))
        use_svd = (max_truncation_err is not None) or (max_singular_values is not None)
        # ncon([A, B, gate], [[-1,1,2],[2,3,-4],[-2,-3,1,3]])
        theta = np.einsum("lam,mbr,xyab->lxyr", self.tensors[site1], self.tensors[site2], gate)
        l, _, _, r = theta.shape
        d = theta.shape[1]

        def set_center(site):
            if self.center_position is not None:
                if self.center_position in (site1, site2):
                    self.center_position = site
                else:
                    self.center_position = None

        if center_position is None:
            center_position = site1
        mat = theta.reshape(l * d, d * r)
        if use_svd:
            u, s, vh, tw = svd_trunc(mat, max_singular_values, max_truncation_err, relative)
            if center_position == site2:
                left, right = u, s.reshape(-1, 1) * vh
                set_center(site2)
            else:
                left, right = u * s.reshape(1, -1), vh
                set_center(site1)
        else:
            tw = np.zeros(1, dtype=mat.dtype)
            if center_position == site1:
                left, right = rq(mat)
                set_center(site1)
            else:
                left, right = qr(mat)
                set_center(site2)
        self.tensors[site1] = left.reshape(l, d, -1)
        self.tensors[site2] = right.reshape(-1, d, r)
        return tw

    def check_canonical(self):
        devs = []
        for site in range(len(self.tensors)):
            t = self.tensors[site]
            if site < self.center_position:
                m = np.einsum("abi,abj->ij", t, t.conj())
            elif site > self.center_position:
                m = np.einsum("iab,jab->ij", t, t.conj())
            else:
                continue
            devs.append(np.linalg.norm(m - np.eye(m.shape[0])))
        return np.linalg.norm(np.array(devs)) if devs else 0.0

    def copy(self):
        r = FiniteMPS([t.copy() for t in self.tensors], canonicalize=False)
        r.center_position = self.center_position
        return r

    def conj(self):
        r = FiniteMPS([t.conj() for t in self.tensors], canonicalize=False)
        r.center_position = self.center_position
        return r


# ----------------------------------------------------------------------------- MPSCircuit
_FIXED = {
    "i": G.I2, "x": G.X, "y": G.Y, "z": G.Z, "h": G.H, "s": G.S, "t": G.T, "sd": G.SD, "td": G.TD,
    "cnot": G.CNOT, "cx": G.CNOT, "cz": G.CZ, "cy": G.CY, "swap": G.SWAP, "toffoli": G.TOFFOLI,
    "fredkin": G.FREDKIN,
}
_PARAM = {"rx": G.rx, "ry": G.ry, "rz": G.rz, "phase": G.phase, "rzz": G.rzz, "rxx": G.rxx,
          "ryy": G.ryy, "iswap": G.iswap}


def split_rules(max_singular_values=None, max_truncation_err=None, relative=False):
    """cons.py split_rules: only the given keys are set."""
    r: Dict[str, Any] = {}
    if max_singular_values is not None:
        r["max_singular_values"] = max_singular_values
    if max_truncation_err is not None:
        r["max_truncation_err"] = max_truncation_err
    if relative is not None:  # as the reference (cons.py:1337): the key is always present
        r["relative"] = relative
    return r


class MPSCircuit:
    """mpscircuit.py:72-1049 (qubits only, d = 2)."""

    def __init__(self, nqubits, center_position=None, tensors=None, wavefunction=None, split=None):
        self.split = split or {}
        if wavefunction is not None:
            tensors = self.wavefunction_to_tensors(np.asarray(wavefunction, dtype=DTYPE), split=self.split)
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
            tensors = [np.array([1.0, 0.0], dtype=DTYPE)[None, :, None] for _ in range(nqubits)]
            self._mps = FiniteMPS(tensors, canonicalize=False)
            self._mps.center_position = 0 if center_position is None else center_position
        self._nqubits = nqubits
        self._fidelity = 1.0

    # -- bookkeeping
    def get_bond_dimensions(self):
        return self._mps.bond_dimensions

    def get_tensors(self):
        return self._mps.tensors

    def get_center_position(self):
        return self._mps.center_position

    def set_split_rules(self, split):
        self.split = split

    def position(self, site):
        self._mps.position(site, normalize=False)

    # -- gates
    def apply_single_gate(self, gate, index):
        if self._mps.center_position != index:
            self.position(index)
        self._mps.apply_one_site_gate(np.asarray(gate).reshape(2, 2), index)

    def apply_adjacent_double_gate(self, gate, index1, index2, center_position=None, split=None):
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
        err = self._mps.apply_two_site_gate(np.asarray(gate).reshape(2, 2, 2, 2), index1, index2,
                                            center_position=center_position, **split)
        self._fidelity *= 1 - np.real(np.sum(err ** 2))

    def consecutive_swap(self, index_from, index_to, split=None):
        if split is None:
            split = self.split
        self.position(index_from)
        swap = G.SWAP.reshape(2, 2, 2, 2)
        if index_from < index_to:
            for i in range(index_from, index_to):
                self.apply_adjacent_double_gate(swap, i, i + 1, center_position=i + 1, split=split)
        elif index_from > index_to:
            for i in range(index_from, index_to, -1):
                self.apply_adjacent_double_gate(swap, i - 1, i, center_position=i - 1, split=split)
        assert self._mps.center_position == index_to

    def apply_double_gate(self, gate, index1, index2, split=None):
        gate = np.asarray(gate).reshape(2, 2, 2, 2)
        assert index1 != index2
        if index1 > index2:
            # NB the reference drops ``split`` on this recursion (mpscircuit.py:366)
            self.apply_double_gate(gate.transpose(1, 0, 3, 2), index2, index1)
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

    @classmethod
    def gate_to_MPO(cls, gate, *index):
        """mpscircuit.py:386-451."""
        if len(index) == 0:
            raise ValueError("`index` must contain at least one site.")
        if not all(index[i] < index[i + 1] for i in range(len(index) - 1)):
            raise ValueError("`index` must be strictly increasing.")
        index_left = int(np.min(index))
        nindex = len(index)
        dim = 2
        gate = np.asarray(gate).reshape((dim,) * (2 * nindex))
        order = tuple(np.arange(2 * nindex).reshape(2, nindex).T.flatten().tolist())
        gate = gate.transpose(order).reshape((dim * dim,) * nindex)
        main_tensors = cls.wavefunction_to_tensors(gate, dim_phys=dim * dim, norm=False)
        tensors: List[np.ndarray] = []
        previous_i = None
        for i, main in zip(np.array(index, dtype=int) - index_left, main_tensors):
            if previous_i is not None:
                for _ in range(int(previous_i) + 1, int(i)):
                    bond = tensors[-1].shape[-1]
                    i4 = np.eye(bond * dim, dtype=tensors[-1].dtype).reshape(bond, dim, bond, dim)
                    tensors.append(i4.transpose(0, 1, 3, 2))
            nleft, _, nright = main.shape
            tensors.append(main.reshape(nleft, dim, dim, nright))
            previous_i = int(i)
        return tensors, index_left

    @classmethod
    def reduce_tensor_dimension(cls, tl, tr, center_left=True, split=None):
        split = split or {}
        ni, di = tl.shape[0], tr.shape[1]
        nk, dk = tr.shape[-1], tr.shape[-2]
        t = np.einsum("iaj,jbk->iabk", tl, tr).reshape(ni * di, nk * dk)
        nl, nr = split_tensor(t, center_left=center_left, split=split)
        return nl.reshape(ni, di, -1), nr.reshape(-1, dk, nk)

    def reduce_dimension(self, index_left, center_left=True, split=None):
        if split is None:
            split = self.split
        index_right = index_left + 1
        assert self._mps.center_position in [index_left, index_right]
        nl, nr = self.reduce_tensor_dimension(self._mps.tensors[index_left], self._mps.tensors[index_right],
                                              center_left=center_left, split=split)
        self._mps.tensors[index_left] = nl
        self._mps.tensors[index_right] = nr
        self._mps.center_position = index_left if center_left else index_right

    def apply_MPO(self, tensors, index_left, center_left=True, split=None):
        """mpscircuit.py:537-634."""
        if split is None:
            split = self.split
        nindex = len(tensors)
        index_right = index_left + nindex - 1
        if center_left:
            end1, end2, step = index_left, index_right, 1
        else:
            end1, end2, step = index_right, index_left, -1
        n_list = np.arange(nindex)[::step]
        idx_list = np.arange(index_left, index_right + 1)[::step]
        self.position(end1)
        residue = None
        for i, idx in zip(n_list, idx_list):
            o = tensors[i]
            t = self._mps.tensors[idx]
            ni, d_out, _, nj = o.shape
            nk, _, nl = t.shape
            ot = np.einsum("iabj,kbl->ikajl", o, t).reshape(ni * nk, d_out, nj * nl)
            if residue is not None:
                if step == 1:
                    ot = np.einsum("ab,bcd->acd", residue, ot)
                else:
                    ot = np.einsum("abc,cd->abd", ot, residue)
            s0, s1, s2 = ot.shape
            if idx != end2:
                if step == 1:
                    q, r = qr(ot.reshape(s0 * s1, -1))
                    self._mps.tensors[idx] = q.reshape(s0, s1, -1)
                    residue = r
                    self._mps.center_position = idx + 1
                else:
                    q_t, r_t = qr(ot.transpose(2, 1, 0).reshape(s2 * s1, -1))
                    self._mps.tensors[idx] = q_t.reshape(s2, s1, -1).transpose(2, 1, 0)
                    residue = r_t.T
                    self._mps.center_position = idx - 1
            else:
                self._mps.tensors[idx] = ot
                self._mps.center_position = end2
        for i in idx_list[::-1][:-1]:
            self.reduce_dimension(min(i, i - step), center_left=center_left, split=split)

    def apply_nqubit_gate(self, gate, *index, split=None):
        gate = np.asarray(gate)
        if not np.all(np.diff(index) > 0):
            order = np.argsort(index)
            order_all = order.tolist() + (order + len(index)).tolist()
            gate = gate.reshape((2,) * (2 * len(index))).transpose(order_all)
            self.apply_nqubit_gate(gate, *np.sort(index).tolist(), split=split)
            return
        if split is None:
            split = self.split
        mpo, index_left = self.gate_to_MPO(gate, *index)
        index_right = index_left + len(mpo) - 1
        diff_left = abs(index_left - self._mps.center_position)
        diff_right = abs(index_right - self._mps.center_position)
        self.apply_MPO(mpo, index_left, center_left=diff_left < diff_right, split=split)

    def apply(self, gate, *index, split=None):
        """apply_general_gate, mpscircuit.py:670-724."""
        if len(index) != len(set(index)):
            raise ValueError(f"gate index {list(index)} has duplicate qubits; each qubit may appear at most once")
        if split is None:
            split = self.split
        if len(index) == 1:
            self.apply_single_gate(gate, *index)
        elif len(index) == 2:
            self.apply_double_gate(gate, *index, split=split)
        else:
            self.apply_nqubit_gate(gate, *index, split=split)

    def __getattr__(self, name):
        low = name.lower()
        if low in _FIXED:
            return lambda *idx, **kw: self.apply(_FIXED[low], *idx, **kw)
        if low in _PARAM:
            def f(*idx, split=None, **kw):
                self.apply(_PARAM[low](**kw), *idx, split=split)
            return f
        raise AttributeError(name)

    # -- outputs
    @classmethod
    def wavefunction_to_tensors(cls, wavefunction, dim_phys=2, norm=True, split=None):
        """mpscircuit.py:765-807."""
        split = split or {}
        w = np.asarray(wavefunction).reshape(-1, 1)
        n_tensors = int(np.round(np.log(w.shape[0]) / np.log(dim_phys)))
        tensors: List[np.ndarray] = []
        for _ in range(n_tensors):
            nright = w.shape[1]
            w = w.reshape(-1, nright * dim_phys)
            w, q = split_tensor(w, center_left=True, split=split)
            tensors.insert(0, q.reshape(-1, dim_phys, nright))
        if w.shape != (1, 1):
            raise ValueError(f"expected scalar wavefunction of shape (1, 1), got {w.shape}")
        if not norm:
            tensors[0] = tensors[0] * w[0, 0]
        return tensors

    def wavefunction(self, form="default"):
        result = np.ones((1, 1, 1), dtype=DTYPE)
        for t in self._mps.tensors:
            result = np.einsum("iaj,jbk->iabk", result, t)
            ni, na, nb, nk = result.shape
            result = result.reshape(ni, na * nb, nk)
        return result.reshape({"default": [-1], "ket": [-1, 1], "bra": [1, -1]}[form])

    state = wavefunction

    def copy(self):
        r = MPSCircuit.__new__(MPSCircuit)
        r.split = dict(self.split)
        r._nqubits = self._nqubits
        r._fidelity = self._fidelity
        r._mps = self._mps.copy()
        return r

    def conj(self):
        r = self.copy()
        r._mps = self._mps.conj()
        return r

    def get_norm(self):
        return np.linalg.norm(self._mps.tensors[self._mps.center_position])

    def normalize(self):
        c = self._mps.center_position
        self._mps.tensors[c] = self._mps.tensors[c] / self.get_norm()

    def amplitude(self, l):
        assert len(l) == self._nqubits
        mats = [self._mps.tensors[i][:, int(ch), :] for i, ch in enumerate(l)]
        return reduce(np.matmul, mats)[0, 0]

    def proj_with_mps(self, other, conj=True):
        bra = other.conj() if conj else other.copy()
        ket = self.copy()
        assert bra._nqubits == ket._nqubits
        for _ in range(bra._nqubits, 1, -1):
            bra_b = bra._mps.tensors[-1]
            ket_a, ket_b = ket._mps.tensors[-2:]
            proj_b = np.einsum("kbm,lbm->kl", bra_b, ket_b)
            new_ka = np.einsum("jal,kl->jak", ket_a, proj_b)
            bra._mps.tensors.pop()
            ket._mps.tensors.pop()
            ket._mps.tensors[-1] = new_ka
        return np.sum(bra._mps.tensors[0] * ket._mps.tensors[0])

    def slice(self, begin, end):
        nq = end - begin + 1
        tensors = [t.copy() for t in self._mps.tensors[begin:end + 1]]
        cp = None
        c = self._mps.center_position
        if c is not None and begin <= c <= end:
            cp = c - begin
        r = MPSCircuit(nq, tensors=tensors, center_position=cp, split=dict(self.split))
        return r

    def expectation(self, *ops, other=None, conj=True, normalize=False, split=None):
        """mpscircuit.py:965-1049."""
        split = split or {}
        ops = [[np.asarray(g), [i] if isinstance(i, int) else list(i)] for g, i in ops]
        all_sites = np.concatenate([op[1] for op in ops])
        if other is None:
            site_begin, site_end = int(np.min(all_sites)), int(np.max(all_sites))
            if self._mps.center_position < site_begin:
                self.position(site_begin)
            elif self._mps.center_position > site_end:
                self.position(site_end)
        mps = self.copy()
        mps.set_split_rules(split)
        for gate, index in ops:
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
            value = value / np.sqrt(n1 * n2)
        return value

    def measure(self, *index, with_prob=False, status=None):
        """mpscircuit.py:1061-1115 with the ``status`` branch of ``backend.probability_sample``
        (abstract_backend.py:1849-1861): outcome = searchsorted(cumsum(p), status_k)."""
        mps = self.copy()
        p = 1.0
        sample = []
        for k, site in enumerate(index):
            mps.position(site)
            t = mps._mps.tensors[site]
            ps = np.real(np.einsum("iaj,iaj->a", t, t.conj()))
            ps = ps / np.sum(ps)
            cum = np.cumsum(ps)
            r = cum[-1] * float(status[k])
            outcome = int(np.searchsorted(cum, r))
            p = p * ps[outcome]
            mps._mps.tensors[site] = t[:, outcome, :][:, None, :]
            sample.append(outcome)
        return np.array(sample, dtype=np.float64), (p if with_prob else -1.0)

    def reduced_density_matrix(self, keep):
        """mpscircuit.py:1117-1240 (dense restatement through the wavefunction)."""
        n = self._nqubits
        w = self.wavefunction().reshape([2] * n)
        keep = list(keep)
        rest = [i for i in range(n) if i not in keep]
        m = np.transpose(w, keep + rest).reshape(2 ** len(keep), -1)
        return m @ m.conj().T

    def expectation_ps(self, x=None, y=None, z=None, **kw):
        """abstractcircuit.py:1523-1603 (Pauli-string shortcut)."""
        ops = []
        for mat, idx in ((G.X, x), (G.Y, y), (G.Z, z)):
            for i in (idx or []):
                ops.append((mat, [i]))
        return self.expectation(*ops, **kw)
