"""reference ``tensorcircuit/templates/measurements.py:156-191``: Hamiltonian expectation helpers."""

from typing import Any

from .. import cons
from ..quantum import PauliSum, QuOperator

Tensor = Any


def sparse_expectation(c: Any, hamiltonian: PauliSum) -> Tensor:
    """<psi|H|psi> for a matrix-free Pauli sum: every string goes through ``expectation_ps`` lazily, so the
    whole sum is ONE fused measurement (and one Pauli-sum cotangent in the backward pass)."""
    if not isinstance(hamiltonian, PauliSum):
        raise TypeError("Backend 'hip': sparse Hamiltonians are PauliSum objects (tc.quantum.PauliStringSum2COO)")
    if hamiltonian.n != c._nqubits:
        raise ValueError("Hamiltonian and circuit act on different numbers of qubits")
    e = 0.0
    for s, w in zip(hamiltonian.structures, hamiltonian.weights):
        x = [q for q, p in enumerate(s) if p == 1]
        y = [q for q, p in enumerate(s) if p == 2]
        z = [q for q, p in enumerate(s) if p == 3]
        if not (x or y or z):
            e = e + w
        else:
            e = e + w * c.expectation_ps(x=x, y=y, z=z)
    return cons.backend.real(e)


def mpo_expectation(c: Any, mpo: QuOperator) -> Tensor:
    """reference measurements.py:194-208: real <psi| O |psi> for an operator in ``QuOperator`` form.  A local-tensor
    operator is measured as what it is -- ``c.expectation((tensor, sites))``, i.e. the fused Pauli-sum measurement and
    its cotangent kernel in reverse mode; a dense one through the state."""
    if not isinstance(mpo, QuOperator):
        raise TypeError("mpo_expectation expects a tc.quantum.QuOperator")
    if mpo.n != c._nqubits:
        raise ValueError("operator and circuit act on different numbers of qubits")
    b = cons.backend
    if mpo.local is not None:
        return b.real(c.expectation((mpo.local, list(mpo.loc))))
    w = c.state(form="ket")
    # H|psi> and <psi|H psi> on tcmi_cgemm (backend.matmul), not torch's ``@`` (rocBLAS)
    hw = b.matmul(b.cast(mpo.dense, cons.dtypestr), w)
    return b.real(b.matmul(b.adjoint(w), hw)[0, 0])


def operator_expectation(c: Any, hamiltonian: Any) -> Tensor:
    """reference measurements.py:156-172: dense matrix, sparse (here: PauliSum) or MPO (QuOperator) Hamiltonian."""
    if isinstance(hamiltonian, PauliSum):
        return sparse_expectation(c, hamiltonian)
    if isinstance(hamiltonian, QuOperator):
        return mpo_expectation(c, hamiltonian)
    b = cons.backend
    w = c.state(form="ket")
    h = b.cast(b.convert_to_tensor(hamiltonian), cons.dtypestr)
    e = b.matmul(b.adjoint(w), b.matmul(h, w))[0, 0]      # tcmi_cgemm, differentiable through its own backward rule
    return b.real(e)
