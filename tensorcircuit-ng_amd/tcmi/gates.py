"""Gate library of the HIP path (mirrors the reference ``tensorcircuit/gates.py`` surface that the
hot path uses: constant matrices ``:45-174``, ``Gate`` ``:185-224``, parametrised gates
``:584-978``).

Every parametrised gate is expressed in the *trig-linear* form the device-side table builder
evaluates per batch element (``tcmi_build_tables``):

    M(theta) = C0 + cos(k*theta + o) * C1 + sin(k*theta + o) * C2

so a gate whose angle is a device tensor never needs a host round trip; gates with several tensor
parameters are decomposed into single-parameter factors.  Unit-modulus diagonal gates additionally
carry their phase polynomial (``plan.DiagTerm``) and are applied without being tile-local.
"""

from dataclasses import dataclass
from typing import Any, List, Optional, Sequence

import numpy as np

from . import plan as P

# ---- constants (reference gates.py:45-174) ---------------------------------------------
_i_matrix = np.eye(2, dtype=np.complex128)
_x_matrix = np.array([[0.0, 1.0], [1.0, 0.0]], dtype=np.complex128)
_y_matrix = np.array([[0.0, -1j], [1j, 0.0]], dtype=np.complex128)
_z_matrix = np.array([[1.0, 0.0], [0.0, -1.0]], dtype=np.complex128)
_h_matrix = np.array([[1.0, 1.0], [1.0, -1.0]], dtype=np.complex128) / np.sqrt(2.0)
_s_matrix = np.diag([1.0, 1j]).astype(np.complex128)
_t_matrix = np.diag([1.0, np.exp(0.25j * np.pi)]).astype(np.complex128)
_sd_matrix = _s_matrix.conj().T
_td_matrix = _t_matrix.conj().T
_wroot_matrix = np.array(
    [[1.0, -(1.0 + 1.0j) / np.sqrt(2.0)], [(1.0 - 1.0j) / np.sqrt(2.0), 1.0]], dtype=np.complex128
) / np.sqrt(2.0)
_i00 = np.diag([1.0, 0.0]).astype(np.complex128)
_i11 = np.diag([0.0, 1.0]).astype(np.complex128)
_i01 = np.array([[0.0, 1.0], [0.0, 0.0]], dtype=np.complex128)
_i10 = _i01.T.copy()


def _kron(*ms):
    out = np.eye(1, dtype=np.complex128)
    for m in ms:
        out = np.kron(out, m)
    return out


_ii_matrix = _kron(_i_matrix, _i_matrix)
_xx_matrix = _kron(_x_matrix, _x_matrix)
_yy_matrix = _kron(_y_matrix, _y_matrix)
_zz_matrix = _kron(_z_matrix, _z_matrix)
_cnot_matrix = _kron(_i00, _i_matrix) + _kron(_i11, _x_matrix)
_cy_matrix = _kron(_i00, _i_matrix) + _kron(_i11, _y_matrix)
_cz_matrix = _kron(_i00, _i_matrix) + _kron(_i11, _z_matrix)
_swap_matrix = np.eye(4, dtype=np.complex128)[[0, 2, 1, 3]]
_toffoli_matrix = np.eye(8, dtype=np.complex128)
_toffoli_matrix[6:, 6:] = _x_matrix
_fredkin_matrix = np.eye(8, dtype=np.complex128)[[0, 1, 2, 3, 4, 6, 5, 7]]
_pauli = [_i_matrix, _x_matrix, _y_matrix, _z_matrix]
# generator order of su4 (reference gates.py:86-104)
_su4_generators = np.stack(
    [_kron(_pauli[a], _pauli[b]) for a in range(4) for b in range(4) if (a, b) != (0, 0)]
)

_CONST_GATES = {
    "i": _i_matrix, "x": _x_matrix, "y": _y_matrix, "z": _z_matrix, "h": _h_matrix,
    "s": _s_matrix, "t": _t_matrix, "sd": _sd_matrix, "td": _td_matrix, "wroot": _wroot_matrix,
    "cnot": _cnot_matrix, "cx": _cnot_matrix, "cy": _cy_matrix, "cz": _cz_matrix,
    "swap": _swap_matrix, "toffoli": _toffoli_matrix, "ccnot": _toffoli_matrix,
    "ccx": _toffoli_matrix, "fredkin": _fredkin_matrix, "cswap": _fredkin_matrix,
}


class Gate:
    """Minimal stand-in for the reference ``Gate(tn.Node)`` (gates.py:185-224): a named tensor of
    shape ``[2] * 2k`` with axes ``[out.., in..]``."""

    def __init__(self, tensor, name="__unnamed_node__"):
        # a backend (device) tensor is kept as it is, so that it can stay on the autograd tape
        self.tensor = tensor if _is_device_tensor(tensor) else np.asarray(tensor)
        self.name = name

    def copy(self):
        return Gate(self.tensor, self.name)

    def __repr__(self):
        return f"Gate(name={self.name!r}, tensor=\n{self.tensor!r})"


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "requires_grad")


def _as_gate(m, name):
    if _is_device_tensor(m):
        k = int(round(np.log2(m.numel()))) // 2
        return Gate(m.reshape([2] * (2 * k)), name=name)
    m = np.asarray(m)
    k = int(round(np.log2(m.size))) // 2
    return Gate(m.reshape([2] * (2 * k)), name=name)


def _npdtype():
    from . import cons

    return cons.npdtype


def _make_const(name):
    def f():
        return _as_gate(_CONST_GATES[name].astype(_npdtype()), name)

    f.__name__ = name
    return f


for _n in _CONST_GATES:
    globals()[_n] = _make_const(_n)
    globals()[_n + "_gate"] = globals()[_n]


# ---- parameter handling ----------------------------------------------------------------
def is_concrete(v):
    """True for python / numpy numbers (usable on the host without a device round trip)."""
    return isinstance(v, (int, float, complex, np.number)) or (
        isinstance(v, np.ndarray) and v.ndim == 0
    )


@dataclass
class TrigSpec:
    """One factor ``C0 + cos(k*theta+o) C1 + sin(k*theta+o) C2`` on ``nq`` qubits."""

    c0: np.ndarray
    c1: np.ndarray
    c2: np.ndarray
    theta: Any          # python number or backend tensor (0-d / batched 0-d)
    scale: float = 1.0
    offset: float = 0.0
    name: str = ""

    def matrix(self, theta=None):
        t = self.theta if theta is None else theta
        a = self.scale * float(t) + self.offset
        return self.c0 + np.cos(a) * self.c1 + np.sin(a) * self.c2


def _z(d):
    return np.zeros((d, d), dtype=np.complex128)


def rx_spec(theta):
    """reference gates.py:692-707."""
    return [TrigSpec(_z(2), _i_matrix, -1j * _x_matrix, theta, 0.5, name="rx")]


def ry_spec(theta):
    return [TrigSpec(_z(2), _i_matrix, -1j * _y_matrix, theta, 0.5, name="ry")]


def rz_spec(theta):
    return [TrigSpec(_z(2), _i_matrix, -1j * _z_matrix, theta, 0.5, name="rz")]


def phase_spec(theta):
    """reference gates.py:584-603: diag(1, e^{i theta})."""
    return [TrigSpec(_i00, _i11, 1j * _i11, theta, 1.0, name="phase")]


def exp1_spec(unitary, theta, half=False, name="exp1"):
    """reference gates.py:920-953: cos(t) I - i sin(t) U (t = theta/2 iff half)."""
    u = np.asarray(unitary, dtype=np.complex128)
    d = int(round(np.sqrt(u.size)))
    u = u.reshape(d, d)
    return [TrigSpec(_z(d), np.eye(d, dtype=np.complex128), -1j * u, theta, 0.5 if half else 1.0, name=name)]


def iswap_spec(theta):
    """reference gates.py:788-814."""
    d1 = np.diag([1.0, 0, 0, 1.0]).astype(np.complex128)
    d2 = np.diag([0, 1.0, 1.0, 0]).astype(np.complex128)
    od = np.zeros((4, 4), dtype=np.complex128)
    od[1, 2] = od[2, 1] = 1.0
    return [TrigSpec(d1, d2, 1j * od, theta, np.pi / 2, name="iswap")]


def _axis(alpha, phi):
    return (
        np.sin(alpha) * np.cos(phi) * _x_matrix
        + np.sin(alpha) * np.sin(phi) * _y_matrix
        + np.cos(alpha) * _z_matrix
    )


def _axis_frame_specs(alpha, phi, inverse):
    """V = rz(phi) ry(alpha) maps the Z axis onto n = (sin a cos p, sin a sin p, cos a): factors of V (or of V^dagger)
    in application order, each with one angle -- the form tensor-valued ``alpha`` / ``phi`` take on the plan."""
    ry = lambda a, sc: TrigSpec(_z(2), _i_matrix, -1j * _y_matrix, a, sc, name="ry")
    rz = lambda a, sc: TrigSpec(_z(2), _i_matrix, -1j * _z_matrix, a, sc, name="rz")
    return [rz(phi, -0.5), ry(alpha, -0.5)] if inverse else [ry(alpha, 0.5), rz(phi, 0.5)]


def r_spec(theta, alpha, phi):
    """reference gates.py:661-689: R = cos(theta) I - i sin(theta) n.sigma = exp(-i theta n.sigma).  Concrete
    ``alpha`` / ``phi``: one factor; tensor-valued: V exp(-i theta Z) V^dagger with V = rz(phi) ry(alpha), five
    one-angle factors in application order (all differentiable / batchable)."""
    if is_concrete(alpha) and is_concrete(phi):
        return [TrigSpec(_z(2), _i_matrix, -1j * _axis(float(alpha), float(phi)), theta, 1.0, name="r")]
    core = TrigSpec(_z(2), _i_matrix, -1j * _z_matrix, theta, 1.0, name="r")
    return _axis_frame_specs(alpha, phi, True) + [core] + _axis_frame_specs(alpha, phi, False)


def cr_spec(theta, alpha, phi):
    """reference gates.py:817-849 (concrete alpha / phi; the tensor-valued form is assembled in
    ``Circuit._vgate``: frame change on the target, controlled exp(-i theta Z), frame change back)."""
    if not (is_concrete(alpha) and is_concrete(phi)):
        raise NotImplementedError("cr_spec needs concrete alpha / phi; use Circuit.cr for tensor-valued angles")
    ax = _axis(float(alpha), float(phi))
    return [TrigSpec(_kron(_i00, _i_matrix), _kron(_i11, _i_matrix), -1j * _kron(_i11, ax), theta, 1.0, name="cr")]


def cu_spec(theta, phi, lbd):
    """reference gates.py cu = controlled(u): controlled(phase(phi) ry(theta) phase(lbd)) = cphase(phi) cry(theta)
    cphase(lbd), in application order."""
    return [cphase_spec(lbd)[0], controlled_rot_spec(_y_matrix, theta, "cry")[0], cphase_spec(phi)[0]]


def controlled_rot_spec(pauli, theta, name):
    """crx / cry / crz = |0><0| x I + |1><1| x exp(-i theta/2 P) (reference gates.py:313-346 applied
    to rx/ry/rz)."""
    return [TrigSpec(_kron(_i00, _i_matrix), _kron(_i11, _i_matrix), -1j * _kron(_i11, pauli), theta, 0.5, name=name)]


def cphase_spec(theta):
    p11 = _kron(_i11, _i11)
    return [TrigSpec(np.eye(4, dtype=np.complex128) - p11, p11, 1j * p11, theta, 1.0, name="cphase")]


def u_spec(theta, phi, lbd):
    """reference gates.py:630-658.  U = phase(phi) . ry(theta) . phase(lbd) exactly; returned in
    application order (first factor is applied first)."""
    return [phase_spec(lbd)[0], ry_spec(theta)[0], phase_spec(phi)[0]]


# concrete-matrix helpers used by the reference-style functional API (tc.gates.rx_gate(0.3) ...)
def _concrete(specs):
    m = None
    for s in specs:
        mm = s.matrix()
        m = mm if m is None else mm @ m
    return m


def rx_gate(theta=0.0):
    return _as_gate(_concrete(rx_spec(theta)).astype(_npdtype()), "rx")


def ry_gate(theta=0.0):
    return _as_gate(_concrete(ry_spec(theta)).astype(_npdtype()), "ry")


def rz_gate(theta=0.0):
    return _as_gate(_concrete(rz_spec(theta)).astype(_npdtype()), "rz")


def phase_gate(theta=0.0):
    return _as_gate(_concrete(phase_spec(theta)).astype(_npdtype()), "phase")


def r_gate(theta=0.0, alpha=0.0, phi=0.0):
    return _as_gate(_concrete(r_spec(theta, alpha, phi)).astype(_npdtype()), "r")


def u_gate(theta=0.0, phi=0.0, lbd=0.0):
    return _as_gate(_concrete(u_spec(theta, phi, lbd)).astype(_npdtype()), "u")


def iswap_gate(theta=1.0):
    return _as_gate(_concrete(iswap_spec(theta)).astype(_npdtype()), "iswap")


def cr_gate(theta=0.0, alpha=0.0, phi=0.0):
    return _as_gate(_concrete(cr_spec(theta, alpha, phi)).astype(_npdtype()), "cr")


def exponential_gate_unity(unitary, theta, half=False, name="none"):
    return _as_gate(_concrete(exp1_spec(unitary, theta, half)).astype(_npdtype()), "exp1-" + name)


exp1_gate = exponential_gate_unity


def exponential_gate(unitary, theta, name="none"):
    """reference gates.py:893-914: expm(-i theta U) (concrete theta only)."""
    from scipy.linalg import expm

    u = np.asarray(unitary, dtype=np.complex128)
    d = int(round(np.sqrt(u.size)))
    return _as_gate(expm(-1j * float(theta) * u.reshape(d, d)).astype(_npdtype()), "exp-" + name)


exp_gate = exponential_gate


def rzz_gate(theta=0.0):
    return exponential_gate_unity(_zz_matrix, theta, half=True, name="rzz")


def rxx_gate(theta=0.0):
    return exponential_gate_unity(_xx_matrix, theta, half=True, name="rxx")


def ryy_gate(theta=0.0):
    return exponential_gate_unity(_yy_matrix, theta, half=True, name="ryy")


def su4_gate(theta, name="su(4)"):
    """reference gates.py:956-972 (concrete theta only)."""
    from scipy.linalg import expm

    th = np.asarray(theta, dtype=np.float64).reshape(15)
    gen = np.einsum("i,iab->ab", th.astype(np.complex128), _su4_generators)
    return _as_gate(expm(-1j * gen).astype(_npdtype()), name)


def any_gate(unitary, name="any"):
    """reference gates.py:866-890."""
    if isinstance(unitary, Gate):
        return unitary
    if _is_device_tensor(unitary):
        return _as_gate(unitary, name)
    return _as_gate(np.asarray(unitary).astype(_npdtype()), name)


def random_two_qubit_gate(seed=None):
    """reference gates.py:852-863."""
    from scipy.stats import unitary_group

    return _as_gate(unitary_group.rvs(4, random_state=seed).astype(_npdtype()), "R2Q")


def random_single_qubit_gate(seed=None):
    rng = np.random.default_rng(seed)
    theta, alpha, phi = rng.random(3) * 2 * np.pi
    return r_gate(theta, alpha, phi)


def matrix_for_gate(gate):
    """reference gates.py: reshape a Gate tensor to its square matrix."""
    t = gate.tensor
    if _is_device_tensor(t):
        t = t.detach().cpu().numpy()
    t = np.asarray(t)
    d = int(round(np.sqrt(t.size)))
    return t.reshape(d, d)


# ---- gates in MPO / diagonal (hyperedge) form: reference gates.py:981-1190 ----------------------------------------------
class Operator:
    """What the reference's MPO-format gate factories return (``QuOperator`` / ``QuVector``, quantum.py): a small tensor
    network -- ``nodes`` = [(tensor, edge labels)], dangling ``out_edges`` then ``in_edges`` -- that a circuit applies
    with ``c.mpo(*index, mpo=op)`` or, for the diagonal forms, ``c.diagonal(...)``.  Here it is a host-side description:
    ``eval_matrix()`` contracts it with numpy; the circuit lowers it to the tile-VM's native operations
    (``Circuit.apply_general_gate``)."""

    def __init__(self, nodes, out_edges, in_edges=(), kind=None):
        self.nodes = [(np.asarray(t), list(e)) for t, e in nodes]
        self.out_edges, self.in_edges = list(out_edges), list(in_edges)
        self.kind = kind        # ("multicontrol", unitary, ctrl) | ("diagonal", vector) | None

    def copy(self):
        return Operator(self.nodes, self.out_edges, self.in_edges, self.kind)

    def adjoint(self):
        return Operator([(np.conj(t), e) for t, e in self.nodes], self.in_edges or self.out_edges,
                        self.out_edges if self.in_edges else (), None)

    def eval(self):
        """The contracted tensor with axes (out_edges..., in_edges...)."""
        labels = {}
        for _, es in self.nodes:
            for e in es:
                labels.setdefault(e, len(labels))
        import string

        sym = (string.ascii_letters * 4)
        expr = ",".join("".join(sym[labels[e]] for e in es) for _, es in self.nodes)
        outs = "".join(sym[labels[e]] for e in self.out_edges + self.in_edges)
        return np.einsum(expr + "->" + outs, *[t for t, _ in self.nodes])

    def eval_matrix(self):
        t = self.eval()
        d = 2 ** len(self.out_edges)
        return t.reshape(d, -1) if self.in_edges else t.reshape(-1)


def multicontrol_gate(unitary, ctrl=1):
    """reference gates.py:981-1055: U on the target qubits iff the control qubits read ``ctrl``; an MPO of bond
    dimension 2 (the same node tensors as the reference's)."""
    if isinstance(unitary, Gate):
        unitary = unitary.tensor
    if _is_device_tensor(unitary):
        unitary = unitary.detach().cpu().numpy()
    u = np.asarray(unitary, dtype=np.complex128)
    d = int(round(np.sqrt(u.size)))
    u = u.reshape(d, d)
    l = int(round(np.log2(d)))
    ctrl = [int(ctrl)] if isinstance(ctrl, (int, np.integer)) else [int(round(float(np.real(c)))) for c in ctrl]
    rend = np.stack([u, np.eye(d)]).reshape([2] + [2] * (2 * l))
    nodes = []
    left = np.zeros([2, 2, 2])
    if ctrl[0] == 1:
        left[1, 1, 0] = 1
        left[0, 0, 1] = 1
    else:
        left[0, 0, 0] = 1
        left[1, 1, 1] = 1
    eid = iter(range(10 ** 6))
    bond = next(eid)
    outs, ins = [next(eid)], [next(eid)]
    nodes.append((left, [outs[0], ins[0], bond]))
    for c in ctrl[1:]:
        mid = np.zeros([2, 2, 2, 2])
        if c == 1:
            mid[0, 1, 1, 0] = mid[1, 0, 0, 1] = mid[1, 1, 1, 1] = mid[0, 0, 0, 1] = 1
        else:
            mid[0, 0, 0, 0] = mid[1, 1, 1, 1] = mid[1, 0, 0, 1] = mid[0, 1, 1, 1] = 1
        o, i_, nb = next(eid), next(eid), next(eid)
        nodes.append((mid, [bond, o, i_, nb]))
        outs.append(o)
        ins.append(i_)
        bond = nb
    to, ti = [next(eid) for _ in range(l)], [next(eid) for _ in range(l)]
    nodes.append((rend, [bond] + to + ti))
    return Operator(nodes, outs + to, ins + ti, kind=("multicontrol", u, list(ctrl)))


def diagonal_gate(diag, dim=2, name="diagonal"):
    """reference gates.py:1058-1075: the coefficient tensor [2]^k of a diagonal gate (applied through hyperedges)."""
    if dim != 2:
        raise NotImplementedError("Backend 'hip' has not implemented qudit gates")
    if _is_device_tensor(diag):
        k = int(round(np.log2(diag.numel())))
        return Gate(diag.reshape([2] * k), name=name)
    d = np.asarray(diag).astype(_npdtype())
    k = int(round(np.log2(d.size)))
    return Gate(d.reshape([2] * k), name=name)


def rzm_gate(theta, n, dim=2, name="rzm"):
    """reference gates.py:1078-1131: exp(-i theta/2 Z...Z) on n qubits as an MPS of diagonal coefficients (chi = 2)."""
    if n < 2:
        raise ValueError("Gate requires at least 2 qubits.")
    if dim != 2:
        raise ValueError("rzm gate only supports dim=2 at the moment.")
    th = float(np.real(theta if is_concrete(theta) else np.asarray(theta.detach().cpu())))
    c, s_ = np.cos(th / 2), np.sin(th / 2)
    m1 = np.array([c, -1j * s_, c, 1j * s_]).reshape(2, 2)
    mk = np.zeros((2, 2, 2), dtype=np.complex128)
    mk[0, 0, 0] = mk[0, 1, 0] = mk[1, 0, 1] = 1.0
    mk[1, 1, 1] = -1.0
    mn = np.array([[1.0, 1.0], [1.0, -1.0]], dtype=np.complex128)
    return _diag_mps([m1] + [mk] * (n - 2) + [mn], name, None)


def cmz_gate(n, dim=2, name="cmz"):
    """reference gates.py:1134-1185: the multi-controlled Z on n qubits as an MPS of diagonal coefficients (chi = 2)."""
    if n < 2:
        raise ValueError("Gate requires at least 2 qubits.")
    if dim != 2:
        raise ValueError("cmz gate only supports dim=2 at the moment.")
    m1 = np.array([[1.0, 0.0], [1.0, -2.0]], dtype=np.complex128)
    mk = np.zeros((2, 2, 2), dtype=np.complex128)
    mk[0, 0, 0] = mk[0, 1, 0] = mk[1, 1, 1] = 1.0
    mn = np.array([[1.0, 1.0], [0.0, 1.0]], dtype=np.complex128)
    return _diag_mps([m1] + [mk] * (n - 2) + [mn], name, None)


def _diag_mps(tensors, name, kind):
    n = len(tensors)
    outs = list(range(n))
    bonds = [n + i for i in range(n - 1)]
    nodes = [(tensors[0], [outs[0], bonds[0]])]
    for i in range(1, n - 1):
        nodes.append((tensors[i], [bonds[i - 1], outs[i], bonds[i]]))
    nodes.append((tensors[-1], [bonds[-1], outs[-1]]))
    op = Operator(nodes, outs, (), kind)
    op.kind = ("diagonal", op.eval_matrix())
    return op


def mpo_gate(mpo, name="mpo"):
    return mpo


# short names of the parameterised gates (reference gates.py:1192-1232: tc.gates.rx, tc.gates.any ...)
for _n in ("rx", "ry", "rz", "phase", "r", "u", "iswap", "cr", "exp", "exp1", "rzz", "rxx", "ryy", "su4", "any",
           "multicontrol", "mpo", "diagonal", "rzm", "cmz"):
    globals().setdefault(_n, globals()[_n + "_gate"])


def num_to_tensor(*num, dtype=None):
    """reference gates.py:227-262: python numbers -> backend tensors of ``dtype`` (default: the complex dtype)."""
    from . import cons

    dtype = dtype or cons.dtypestr
    out = [cons.backend.cast(cons.backend.convert_to_tensor(np.asarray(n)), dtype) for n in num]
    return out[0] if len(out) == 1 else out


array_to_tensor = num_to_tensor
