"""``DMCircuit``: density-matrix simulator (reference ``tensorcircuit/densitymatrix.py``).

The reference doubles the tensor network (a ket and a bra copy of every gate, Kraus operators contracted
between them).  Here the doubled network is a 2n-qubit *state-vector plan*: rho is stored vectorised,
qubit i = ket index i, qubit n + i = bra index i, so

    U rho U^dagger          ->  U on qubit i, conj(U) on qubit n + i          (unitary gates, any arity)
    sum_k K_k rho K_k^dagger ->  the 4 x 4 super-gate sum_k K_k (x) conj(K_k) on the pair (i, n + i)

and every gate runs through the same tile-VM passes (``tcmi_run_pass``) as ``Circuit``.  One-qubit channels are one
non-unitary two-qubit gate of the doubled plan; two-qubit channels are a sum of branches (``apply_general_kraus``)."""

from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import channels as CH
from . import cons
from . import gates as G
from .circuit import Circuit, sgates, vgates, gate_aliases

Tensor = Any


class DMCircuit:
    is_dm = True

    def __init__(self, nqubits: int, empty: bool = False, inputs: Optional[Tensor] = None,
                 dminputs: Optional[Tensor] = None, **kws: Any) -> None:
        for k in ("mps_inputs", "mpo_dminputs", "tensors"):
            if kws.get(k) is not None:
                raise NotImplementedError(f"Backend 'hip' has not implemented DMCircuit({k}=...).")
        if kws.get("dim") not in (None, 2):
            raise NotImplementedError("the hip backend supports qubits (dim=2) only")
        self._nqubits = nqubits
        n = nqubits
        vec = None
        if inputs is not None:
            psi = cons.backend.cast(cons.backend.convert_to_tensor(inputs), cons.dtypestr).reshape(-1)
            vec = (psi.reshape(-1, 1) * psi.conj().reshape(1, -1)).reshape(-1)
        elif dminputs is not None:
            vec = cons.backend.cast(cons.backend.convert_to_tensor(dminputs), cons.dtypestr).reshape(-1)
        self._c = Circuit(2 * n, inputs=vec)
        self._qir: List[Any] = []

    # ---- gates ---------------------------------------------------------------------------------------
    def _pair(self, index: Sequence[int]) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
        n = self._nqubits
        ket = tuple(int(i) % n for i in index)
        return ket, tuple(i + n for i in ket)

    def apply_general_gate(self, gate: Any, *index: int, name: Optional[str] = None, **kws: Any) -> None:
        ket, bra = self._pair(index)
        self._c.apply_general_gate(gate, *ket, name=name, **kws)
        self._c._conj = True
        try:
            self._c.apply_general_gate(gate, *bra, name=name, **kws)
        finally:
            self._c._conj = False

    apply = apply_general_gate

    def apply_general_kraus(self, kraus: Sequence[Any], index: Sequence[int], **kws: Any) -> None:
        """reference densitymatrix.py:222-244: rho -> sum_k K_k rho K_k^dagger on the given qubits."""
        index = [index] if isinstance(index, int) else list(index)
        ket, bra = self._pair(index)
        if len(index) == 1:
            sup = CH.kraus_to_super_gate(kraus)                       # [ket', bra'; ket, bra]
            self._c._record_const(sup, (ket[0], bra[0]), "kraus", split_conf={})
            return
        if len(index) > 2:
            raise NotImplementedError("Backend 'hip' has not implemented Kraus channels on more than two qubits.")
        # two-qubit channel: the super-gate sum_k K_k (x) conj(K_k) is a non-unitary 4-qubit operator, which the
        # tile-VM has no dense op for.  It is applied as what it is -- a sum of branches: the vectorised rho so far is
        # materialised, every branch applies K_k to the ket pair and conj(K_k) to the bra pair (two dense two-qubit
        # gates, non-unitary allowed) and the branch states are added; the sum is the input state of the plan that
        # records the rest of the circuit (differentiable through ``Circuit(inputs=...)``).
        d = 2 ** len(index)
        from .circuit import _constant_value

        mats = [np.asarray(_constant_value(k.tensor if isinstance(k, G.Gate) else k, "Kraus operator"),
                           dtype=np.complex128).reshape(d, d) for k in kraus]
        rho = self._c.wavefunction().reshape(-1)
        total = None
        for m in mats:
            br = Circuit(2 * self._nqubits, inputs=rho)
            br._record_const(m, ket, "kraus-ket", split_conf={})
            br._record_const(m.conj(), bra, "kraus-bra", split_conf={})
            out = br.wavefunction().reshape(-1)
            total = out if total is None else total + out
        self._c = Circuit(2 * self._nqubits, inputs=total)

    general_kraus = apply_general_kraus

    @staticmethod
    def check_kraus(kraus: Sequence[Any]) -> bool:
        CH.kraus_identity_check(kraus)
        return True

    # ---- outputs -------------------------------------------------------------------------------------
    def densitymatrix(self, check: bool = False, reuse: bool = True) -> Tensor:
        """reference densitymatrix.py:279-300: [2^n, 2^n]."""
        d = 2 ** self._nqubits
        dm = self._c.wavefunction().reshape(d, d)
        if check:
            self.check_density_matrix(dm)
        return dm

    state = densitymatrix

    def wavefunction(self) -> Tensor:
        """reference densitymatrix.py:302-318: for a pure state only — the dominant eigenvector is not
        computed here; use ``Circuit`` for pure-state simulation."""
        raise NotImplementedError("Backend 'hip' has not implemented DMCircuit.wavefunction.")

    @staticmethod
    def check_density_matrix(dm: Tensor) -> None:
        import torch

        assert torch.allclose(dm, dm.conj().t(), atol=1e-5), "density matrix is not Hermitian"
        assert abs(complex(torch.trace(dm)) - 1.0) < 1e-4, "density matrix has trace != 1"

    def expectation(self, *ops: Tuple[Any, List[int]], reuse: bool = True, **kws: Any) -> Tensor:
        """Tr(rho prod ops) (reference densitymatrix.py:331-368, noise-free branch): the operators act on the
        ket indices of a copy of the plan, the trace is the diagonal sum of the resulting matrix."""
        import torch

        if kws.get("noise_conf") is not None:
            raise NotImplementedError("Backend 'hip' has not implemented noise_conf in DMCircuit.expectation.")
        c2 = Circuit.__new__(Circuit)
        c2.__dict__.update(self._c.__dict__)
        c2._ops, c2._params, c2._qir = list(self._c._ops), list(self._c._params), list(self._c._qir)
        c2.state_tensor = None
        if hasattr(c2, "_pending"):
            c2._pending = {}
        for op, idx in ops:
            idx = [idx] if isinstance(idx, int) else list(idx)
            ket, _ = self._pair(idx)
            t = op.tensor if isinstance(op, G.Gate) else op
            if not isinstance(t, np.ndarray):
                t = cons.backend.numpy(cons.backend.convert_to_tensor(t))
            c2._record_const(np.asarray(t), ket, "op")
        d = 2 ** self._nqubits
        m = c2.wavefunction().reshape(d, d)
        return torch.diagonal(m).sum()

    def expectation_ps(self, x: Optional[Sequence[int]] = None, y: Optional[Sequence[int]] = None,
                       z: Optional[Sequence[int]] = None, **kws: Any) -> Tensor:
        ops = []
        for mk, idx in ((G.x, x), (G.y, y), (G.z, z)):
            for i in (idx or []):
                ops.append((mk(), [i]))
        return self.expectation(*ops, **kws)

    def to_qir(self):
        return self._c._qir


DMCircuit2 = DMCircuit


def _make_gate(name):
    def f(self, *index, **kw):
        for idx in Circuit._bcast(index):
            ket, bra = self._pair(idx)
            getattr(self._c, name)(*ket, **kw)
            self._c._conj = True
            try:
                getattr(self._c, name)(*bra, **kw)
            finally:
                self._c._conj = False

    f.__name__ = name
    return f


def _make_channel(name, factory):
    def f(self, *index, **kw):
        for idx in Circuit._bcast(index):
            self.apply_general_kraus(factory(**kw), list(idx))

    f.__name__ = name
    return f


for _n in list(sgates) + list(vgates):
    if _n == "any":
        continue
    setattr(DMCircuit, _n, _make_gate(_n))
    setattr(DMCircuit, _n.upper(), _make_gate(_n))
for _a, _b in [(a[1], a[0]) for a in gate_aliases]:
    if hasattr(DMCircuit, _b):
        setattr(DMCircuit, _a, getattr(DMCircuit, _b))
        setattr(DMCircuit, _a.upper(), getattr(DMCircuit, _b))


def _any(self, *index, **kw):
    self.apply_general_gate(kw.get("unitary"), *index, name=kw.get("name", "any"))


DMCircuit.any = _any
DMCircuit.unitary = _any
for _n, _f in (("depolarizing", CH.depolarizingchannel), ("amplitudedamping", CH.amplitudedampingchannel),
               ("phasedamping", CH.phasedampingchannel), ("generaldepolarizing", CH.generaldepolarizingchannel),
               ("isotropicdepolarizing", CH.isotropicdepolarizingchannel)):
    setattr(DMCircuit, _n, _make_channel(_n, _f))


def _reset(self, *index):
    for idx in Circuit._bcast(index):
        self.apply_general_kraus(CH.resetchannel(), list(idx))


DMCircuit.reset = _reset
