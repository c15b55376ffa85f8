"""Drop-in surface on CPU: set_backend / set_dtype / set_contractor behaviour, gate library values,
circuit recording and the reference's error behaviour."""

import numpy as np
import pytest

import tcmi as tc
from oracle import gates as G


def test_set_backend_registry():
    """reference backends/backend_factory.py:39-59, cons.py:90-135."""
    b = tc.set_backend("hip")
    assert b.name == "hip" and tc.backend is b and tc.get_backend("hip") is b
    import tcmi.circuit as mod

    assert tc.cons.backend is b
    with pytest.raises(ValueError, match="Backend 'nonexistent' does not exist"):
        tc.set_backend("nonexistent")
    with pytest.raises(NotImplementedError, match="Backend 'hip' has not implemented `sparse_dense_matmul`."):
        b.sparse_dense_matmul(1, 2)


def test_set_dtype_broadcast():
    """reference cons.py:185-239: complex128 => float64 / int64, broadcast to loaded modules."""
    assert tc.set_dtype("complex128") == ("complex128", "float64")
    import tcmi.gates as gm

    assert tc.dtypestr == "complex128" and tc.rdtypestr == "float64" and tc.idtypestr == "int64"
    assert tc.cons.npdtype is np.complex128 and tc.gates.x().tensor.dtype == np.complex128
    assert tc.set_dtype("float32") == ("complex64", "float32")
    assert tc.gates.x().tensor.dtype == np.complex64
    with pytest.raises(ValueError):
        tc.set_dtype("int8")
    with tc.runtime_dtype("complex128"):
        assert tc.dtypestr == "complex128"
    assert tc.dtypestr == "complex64"

    @tc.set_function_dtype("complex128")
    def f():
        return tc.dtypestr

    assert f() == "complex128" and tc.dtypestr == "complex64"


def test_set_contractor_names():
    """reference cons.py:1123-1261: known names accepted (incl. cotengra-*), unknown rejected."""
    for name in ("greedy", "plain", "auto", "custom", "cotengra-30-64", "omeco-8-100"):
        tc.set_contractor(name)
    with pytest.raises(ValueError, match="Unknown contractor type"):
        tc.set_contractor("no-such")
    tc.set_contractor("greedy", lowbits=6)
    assert tc.cons._plan_options["lowbits"] == 6
    with tc.runtime_contractor("plain", lowbits=4):
        assert tc.cons._plan_options["lowbits"] == 4
    assert tc.cons._plan_options["lowbits"] == 6
    tc.set_contractor("greedy")


def test_gate_library_matches_oracle():
    tc.set_dtype("complex128")
    try:
        pairs = [
            (tc.gates.h(), G.H), (tc.gates.x(), G.X), (tc.gates.y(), G.Y), (tc.gates.z(), G.Z),
            (tc.gates.s(), G.S), (tc.gates.t(), G.T), (tc.gates.sd(), G.SD), (tc.gates.td(), G.TD),
            (tc.gates.wroot(), G.WROOT), (tc.gates.cnot(), G.CNOT), (tc.gates.cz(), G.CZ),
            (tc.gates.cy(), G.CY), (tc.gates.swap(), G.SWAP), (tc.gates.toffoli(), G.TOFFOLI),
            (tc.gates.fredkin(), G.FREDKIN), (tc.gates.rx_gate(0.37), G.rx(0.37)),
            (tc.gates.ry_gate(0.37), G.ry(0.37)), (tc.gates.rz_gate(0.37), G.rz(0.37)),
            (tc.gates.phase_gate(0.37), G.phase(0.37)), (tc.gates.r_gate(0.3, 0.4, 0.5), G.r(0.3, 0.4, 0.5)),
            (tc.gates.u_gate(0.3, 0.4, 0.5), G.u(0.3, 0.4, 0.5)), (tc.gates.iswap_gate(0.6), G.iswap(0.6)),
            (tc.gates.cr_gate(0.3, 0.4, 0.5), G.cr(0.3, 0.4, 0.5)), (tc.gates.rzz_gate(0.8), G.rzz(0.8)),
            (tc.gates.rxx_gate(0.8), G.rxx(0.8)), (tc.gates.ryy_gate(0.8), G.ryy(0.8)),
            (tc.gates.exp1_gate(tc.gates._zz_matrix, 0.8), G.exp1(G.ZZ, 0.8)),
            (tc.gates.exp_gate(tc.gates._xx_matrix, 0.8), G.exp(G.XX, 0.8)),
            (tc.gates.su4_gate(np.arange(15) * 0.1), G.su4(np.arange(15) * 0.1)),
        ]
        for got, want in pairs:
            k = int(np.log2(want.shape[0]))
            assert got.tensor.shape == (2,) * (2 * k)
            np.testing.assert_allclose(tc.gates.matrix_for_gate(got), want, atol=1e-14, err_msg=got.name)
    finally:
        tc.set_dtype("complex64")


def test_circuit_recording_and_errors():
    c = tc.Circuit(4)
    c.H(0); c.h(1); c.CNOT(0, 1); c.cx(1, 2); c.rx(3, theta=0.5); c.unitary(0, unitary=G.X)
    c.exp1(1, 2, unitary=tc.gates._zz_matrix, theta=0.2)
    c.x([2, 3])                       # index broadcasting (abstractcircuit.py:161-183)
    c.rz(-1, theta=0.1)               # negative index (basecircuit.py:219)
    assert c.gate_count() == 10 and c._ops[-1].qubits == (3,)
    assert [q["name"] for q in c.to_qir()][:4] == ["h", "h", "cnot", "cnot"]
    with pytest.raises(ValueError, match="duplicate qubits"):
        c.cnot(1, 1)
    with pytest.raises(ValueError):
        c.h(7)
    with pytest.raises(ValueError, match="Cannot measure two operators in one index"):
        c.expectation((tc.gates.z(), [0]), (tc.gates.x(), [0]))
    with pytest.raises(NotImplementedError):
        tc.Circuit(2, dim=3)


def test_no_cpu_fallback():
    """Without a GPU the product path raises instead of silently computing on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = tc.Circuit(3)
    c.h(0)
    with pytest.raises(Exception):
        c.wavefunction()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: no product module may import it."""
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorcircuit-ng_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
