"""Gate matrices, restated from the reference (test infrastructure, see oracle/__init__.py).

All matrices are complex128 ``U[out, in]`` with the qubit order of the gate's
``index`` arguments; a k-qubit gate tensor is ``U.reshape([2] * 2k)`` with axes
``[out_0..out_{k-1}, in_0..in_{k-1}]`` (reference ``tensorcircuit/basecircuit.py:288-290``,
``tensorcircuit/gates.py:505-508``).
"""

import numpy as np
from scipy.linalg import expm

# constants: tensorcircuit/gates.py:45-174
I2 = np.eye(2, dtype=np.complex128)
X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
H = np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2)
S = np.array([[1, 0], [0, 1j]], dtype=np.complex128)
T = np.array([[1, 0], [0, np.exp(1j * np.pi / 4)]], dtype=np.complex128)
SD = S.conj().T
TD = T.conj().T
WROOT = (
    1
    / np.sqrt(2)
    * np.array(
        [[1, -1 / np.sqrt(2) * (1 + 1.0j)], [1 / np.sqrt(2) * (1 - 1.0j), 1]],
        dtype=np.complex128,
    )
)
PAULI = [I2, X, Y, Z]

CNOT = np.array(
    [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.complex128
)
CZ = np.diag([1, 1, 1, -1]).astype(np.complex128)
CY = np.array(
    [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, -1j], [0, 0, 1j, 0]], dtype=np.complex128
)
SWAP = np.array(
    [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128
)
TOFFOLI = np.eye(8, dtype=np.complex128)
TOFFOLI[6:, 6:] = X
FREDKIN = np.eye(8, dtype=np.complex128)
FREDKIN[5:7, 5:7] = X

XX = np.kron(X, X)
YY = np.kron(Y, Y)
ZZ = np.kron(Z, Z)

# tensorcircuit/gates.py:86-104 -- generator order of the su4 gate
SU4_GENERATORS = np.stack(
    [
        np.kron(a, b)
        for a, b in [
            (I2, X), (I2, Y), (I2, Z),
            (X, I2), (X, X), (X, Y), (X, Z),
            (Y, I2), (Y, X), (Y, Y), (Y, Z),
            (Z, I2), (Z, X), (Z, Y), (Z, Z),
        ]
    ]
)


def rx(theta):
    """tensorcircuit/gates.py:692-707: cos(t/2) I - i sin(t/2) X."""
    return np.cos(theta / 2.0) * I2 - 1j * np.sin(theta / 2.0) * X


def ry(theta):
    """tensorcircuit/gates.py:710-725."""
    return np.cos(theta / 2.0) * I2 - 1j * np.sin(theta / 2.0) * Y


def rz(theta):
    """tensorcircuit/gates.py:728-743."""
    return np.cos(theta / 2.0) * I2 - 1j * np.sin(theta / 2.0) * Z


def phase(theta):
    """tensorcircuit/gates.py:584-603: diag(1, e^{i theta})."""
    return np.array([[1, 0], [0, np.exp(1j * theta)]], dtype=np.complex128)


def u(theta=0.0, phi=0.0, lbd=0.0):
    """tensorcircuit/gates.py:630-658 (OpenQASM 3 U gate)."""
    return np.array(
        [
            [np.cos(theta / 2), -np.exp(1j * lbd) * np.sin(theta / 2)],
            [
                np.exp(1j * phi) * np.sin(theta / 2),
                np.exp(1j * (phi + lbd)) * np.cos(theta / 2),
            ],
        ],
        dtype=np.complex128,
    )


def r(theta=0.0, alpha=0.0, phi=0.0):
    """tensorcircuit/gates.py:661-689."""
    return (
        np.cos(theta) * I2
        - 1j * np.cos(phi) * np.sin(alpha) * np.sin(theta) * X
        - 1j * np.sin(phi) * np.sin(alpha) * np.sin(theta) * Y
        - 1j * np.sin(theta) * np.cos(alpha) * Z
    )


def cr(theta=0.0, alpha=0.0, phi=0.0):
    """tensorcircuit/gates.py:817-849: |0><0| x I + |1><1| x r(theta, alpha, phi)."""
    up = np.diag([1.0, 0.0]).astype(np.complex128)
    dn = np.diag([0.0, 1.0]).astype(np.complex128)
    return np.kron(up, I2) + np.kron(dn, r(theta, alpha, phi))


def iswap(theta=1.0):
    """tensorcircuit/gates.py:788-814 (theta in units of pi/2)."""
    c, s = np.cos(theta * np.pi / 2), np.sin(theta * np.pi / 2)
    m = np.zeros((4, 4), dtype=np.complex128)
    m[0, 0] = m[3, 3] = 1
    m[1, 1] = m[2, 2] = c
    m[1, 2] = m[2, 1] = 1j * s
    return m


def exp1(unitary, theta, half=False):
    """tensorcircuit/gates.py:920-953: cos(t') I - i sin(t') U, t' = t/2 iff half."""
    unitary = np.asarray(unitary, dtype=np.complex128)
    dim = int(round(np.sqrt(unitary.size)))
    unitary = unitary.reshape(dim, dim)
    if half:
        theta = theta / 2.0
    return np.cos(theta) * np.eye(dim) - 1j * np.sin(theta) * unitary


def exp(unitary, theta):
    """tensorcircuit/gates.py:893-914: expm(-i theta U)."""
    unitary = np.asarray(unitary, dtype=np.complex128)
    dim = int(round(np.sqrt(unitary.size)))
    return expm(-1j * theta * unitary.reshape(dim, dim))


def rzz(theta):
    """tensorcircuit/gates.py:975-978: exp1(ZZ, theta, half=True)."""
    return exp1(ZZ, theta, half=True)


def rxx(theta):
    return exp1(XX, theta, half=True)


def ryy(theta):
    return exp1(YY, theta, half=True)


def su4(theta):
    """tensorcircuit/gates.py:956-972: expm(-i sum_k theta_k G_k)."""
    theta = np.asarray(theta, dtype=np.float64).reshape(15)
    gen = np.einsum("i,iab->ab", theta.astype(np.complex128), SU4_GENERATORS)
    return expm(-1j * gen)


def controlled(u_mat):
    """tensorcircuit/gates.py:313-346: |0><0| x I + |1><1| x U (control first)."""
    u_mat = np.asarray(u_mat, dtype=np.complex128)
    d = u_mat.shape[0]
    out = np.eye(2 * d, dtype=np.complex128)
    out[d:, d:] = u_mat
    return out


def ocontrolled(u_mat):
    """tensorcircuit/gates.py:349-380: |0><0| x U + |1><1| x I."""
    u_mat = np.asarray(u_mat, dtype=np.complex128)
    d = u_mat.shape[0]
    out = np.eye(2 * d, dtype=np.complex128)
    out[:d, :d] = u_mat
    return out


def random_two_qubit_gate(seed):
    """tensorcircuit/gates.py:852-863 with an explicit seed (scipy unitary_group)."""
    from scipy.stats import unitary_group

    return unitary_group.rvs(4, random_state=seed).astype(np.complex128)
