"""Full-size golden fixtures (BASELINE.json configs 3, 4, 5) from the CPU oracle, for tests/test_gpu_scale.py.

VERDICT round 3 ("pin full-size parity to the oracle, not to the product"): a few hundred bytes of numbers that the
oracle can only produce once, in the build container (62 GB of host memory, tens of minutes), committed as
tests/golden/full_size_golden.npz:

  config3   HEA-B n = 28, depth 12, row 0 of the bench's parameter batch (seed 28, normal(0, 0.1), rounded to float32 as
            the bench uploads it): the 28 <X_i>, the 27 <Z_i Z_i+1> and the TFIM energy, from oracle.dense (complex128,
            gate by gate on the 4 GiB state; in-place strided updates, oracle/dense.py::apply_gate_inplace).
  config4   the amplitude <0^32|C|0^32> of the bench's 32-qubit 4x8-grid depth-16 random circuit: numpy complex128
            tensordot chain over a sliced pairwise path, slice by slice.  Path AND sliced indices come from the ORACLE's own
            tools since round 6 (oracle.sliced.greedy_sliced_path: oracle.tn.greedy_path on the network with indices
            removed one at a time -- the unsliced greedy order has a 2^38-element intermediate here; two sliced indices
            bring it to 2^28 at a total cost of 2^43.6, 12 x the product's tree): nothing of the product is imported.
            The same chain is first checked against oracle.tn's own greedy contraction at 20 qubits.
  config5   MPSCircuit n = 64, chi = 128, one TEBD sweep of 63 random SU(4) gates on the bench's random MPS: the fidelity
            estimate, the norm before / after and the Schmidt spectrum of the middle bond, from oracle.mps (numpy, LAPACK
            SVD).

  config3grad   four components of the gradient of that energy with respect to the parameters of row 0, by central
            differences of oracle.dense's energy (the reference's own convention for checking gradients,
            tests/test_mpscircuit.py:452-457), step h = 1e-4 in float64: truncation ~ h^2 |E(3)| / 6 ~ 1e-8, rounding
            ~ 1e-13 / 2h ~ 5e-10 -- below the complex128 tolerance (1e-7) the GPU test asserts.  Components: one ZZ angle
            of the first layer, one rx angle of the first layer, one rx angle in the middle, one ZZ angle of the last layer.

    python tests/golden/make_golden_full.py config3 | config3grad | config4 | config5 | merge
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dense, gates as G, workloads as W  # noqa: E402

OUT = os.path.join(HERE, "full_size_golden.npz")


def _part(name):
    return os.path.join(HERE, f"_full_{name}.npz")


def config3():
    n, d = 28, 12
    params = np.random.default_rng(28).normal(0, 0.1, [32, 2 * d, n]).astype(np.float32)[0].astype(np.float64)
    t0 = time.time()
    psi = dense.run(n, W.hea_b_ops(n, d, params), inplace=True)
    print("state done", time.time() - t0, "s; norm", float(np.vdot(psi, psi).real), flush=True)
    xs, zz = [], []
    for i in range(n):                       # <X_i> = 2 Re sum conj(a[..0..]) a[..1..]
        v = psi.reshape(2**i, 2, 2 ** (n - 1 - i))
        xs.append(2.0 * float(np.real(np.vdot(v[:, 0, :], v[:, 1, :]))))
    prob = (psi.real**2 + psi.imag**2)
    del psi
    for i in range(n - 1):                   # <Z_i Z_i+1> = sum |a|^2 z_i z_i+1
        v = prob.reshape(2**i, 2, 2, 2 ** (n - 2 - i))
        zz.append(float(v[:, 0, 0, :].sum() + v[:, 1, 1, :].sum() - v[:, 0, 1, :].sum() - v[:, 1, 0, :].sum()))
    energy = float(np.sum(zz) - np.sum(xs))
    np.savez(_part("config3"), config3_params=params, config3_x=np.array(xs), config3_zz=np.array(zz),
             config3_energy=np.array(energy))
    print("config3", energy, time.time() - t0, "s")


GRAD_COMPONENTS = [(0, 3), (1, 0), (13, 14), (22, 26)]     # (row of the [2d, n] parameter block, qubit)
GRAD_STEP = 1e-4


def _tfim_terms(psi, n):
    """(<X_i>, <Z_i Z_i+1>) of a flat state through strided views (no index arrays: 4 GiB states)."""
    xs, zz = [], []
    for i in range(n):                       # <X_i> = 2 Re sum conj(a[..0..]) a[..1..]
        v = psi.reshape(2**i, 2, 2 ** (n - 1 - i))
        xs.append(2.0 * float(np.real(np.vdot(v[:, 0, :], v[:, 1, :]))))
    prob = (psi.real**2 + psi.imag**2)
    for i in range(n - 1):                   # <Z_i Z_i+1> = sum |a|^2 z_i z_i+1
        v = prob.reshape(2**i, 2, 2, 2 ** (n - 2 - i))
        zz.append(float(v[:, 0, 0, :].sum() + v[:, 1, 1, :].sum() - v[:, 0, 1, :].sum() - v[:, 1, 0, :].sum()))
    return xs, zz


def _config3_fd(args):
    k, sign = args
    n, d = 28, 12
    params = np.random.default_rng(28).normal(0, 0.1, [32, 2 * d, n]).astype(np.float32)[0].astype(np.float64)
    r, q = GRAD_COMPONENTS[k]
    params[r, q] += sign * GRAD_STEP
    t0 = time.time()
    psi = dense.run(n, W.hea_b_ops(n, d, params), inplace=True)
    xs, zz = _tfim_terms(psi, n)
    e = float(np.sum(zz) - np.sum(xs))
    print("  component", k, "sign", sign, "energy", e, time.time() - t0, "s", flush=True)
    return k, sign, e


def config3grad():
    import multiprocessing as mp

    t0 = time.time()
    jobs = [(k, s_) for k in range(len(GRAD_COMPONENTS)) for s_ in (+1, -1)]
    with mp.get_context("spawn").Pool(int(os.environ.get("GOLDEN_WORKERS", "4"))) as pool:
        res = pool.map(_config3_fd, jobs, chunksize=1)
    e = {(k, s_): v for k, s_, v in res}
    fd = np.array([(e[(k, 1)] - e[(k, -1)]) / (2 * GRAD_STEP) for k in range(len(GRAD_COMPONENTS))])
    np.savez(_part("config3grad"), config3_grad_components=np.array(GRAD_COMPONENTS), config3_grad_fd=fd,
             config3_grad_step=np.array(GRAD_STEP))
    print("config3grad", fd, time.time() - t0, "s")


def _rqc_network(rows, cols, depth, dtype=np.complex128):
    """Tensors + index lists of <0..0|C|0..0> for the bench's brickwork circuit (bench.py::rqc_leg): every gate a
    [2,2,2,2] tensor (out_a, out_b, in_a, in_b), |0> caps at both ends; edges are integers."""
    n = rows * cols
    q = lambda r, c: r * cols + c  # noqa: E731
    tensors, inputs = [], []
    nxt = [0]

    def new():
        nxt[0] += 1
        return nxt[0] - 1

    wire = []
    zero = np.array([1.0, 0.0], dtype=dtype)
    for _ in range(n):
        e = new()
        wire.append(e)
        tensors.append(zero)
        inputs.append([e])
    k = 0
    for dd in range(depth):
        pat = dd % 4
        if pat in (0, 1):
            pairs = [(q(r, cc), q(r, cc + 1)) for r in range(rows) for cc in range(pat, cols - 1, 2)]
        else:
            pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, rows - 1, 2) for cc in range(cols)]
        for a, b in pairs:
            u = np.asarray(G.random_two_qubit_gate(7000 + k), dtype=dtype).reshape(2, 2, 2, 2)
            k += 1
            ea, eb = new(), new()
            tensors.append(u)
            inputs.append([ea, eb, wire[a], wire[b]])
            wire[a], wire[b] = ea, eb
    for i in range(n):
        tensors.append(zero)
        inputs.append([wire[i]])
    return tensors, inputs


def _contract_path(tensors, inputs, path, sliced, values):
    from oracle import sliced as OS

    return OS.contract_path(tensors, inputs, path, sliced, values)


def _tree(inputs, max_width):
    """(path, sliced indices) from the oracle's own search (oracle/sliced.py); nothing of the product is imported."""
    from oracle import sliced as OS

    size_dict = {e: 2 for s in inputs for e in s}
    path, sliced, width, cost = OS.greedy_sliced_path(inputs, size_dict, max_width, log=lambda m: print("  search:", m, flush=True))
    return path, sliced, (width, cost)


def config4():
    from oracle import tn as OT

    # the chain against oracle.tn's own greedy contraction at 20 qubits (4 x 5 grid, depth 8)
    ts, ins = _rqc_network(4, 5, 8)
    path, sliced, _ = _tree(ins, 9)
    assert sliced, "the 20-qubit check must exercise the slice sum"
    tot = 0.0
    for s_ in range(2 ** len(sliced)):
        vals = [(s_ >> (len(sliced) - 1 - j)) & 1 for j in range(len(sliced))]
        tot += _contract_path(ts, ins, path, sliced, vals)
    nodes = [OT.Node(t, list(e)) for t, e in zip(ts, ins)]
    # oracle.tn nodes share Edge identity through equal labels: rebuild through its Circuit instead
    c = OT.Circuit(20)
    q = lambda r, cc: r * 5 + cc  # noqa: E731
    k = 0
    for dd in range(8):
        pat = dd % 4
        if pat in (0, 1):
            pairs = [(q(r, cc), q(r, cc + 1)) for r in range(4) for cc in range(pat, 4, 2)]
        else:
            pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, 3, 2) for cc in range(5)]
        for a, b in pairs:
            c.apply(G.random_two_qubit_gate(7000 + k), a, b)
            k += 1
    ref = complex(c.amplitude("0" * 20))
    print("20-qubit check: sliced chain", tot, "oracle.tn greedy", ref, flush=True)
    assert abs(tot - ref) < 1e-12
    # full size
    t0 = time.time()
    ts, ins = _rqc_network(4, 8, 16)
    path, sliced, (width, cost) = _tree(ins, 28)
    print("tree: slices", 2 ** len(sliced), "log2 cost %.2f" % cost, "max size 2^%d" % int(width), "search", time.time() - t0, "s", flush=True)
    tot = 0.0
    for s_ in range(2 ** len(sliced)):
        vals = [(s_ >> (len(sliced) - 1 - j)) & 1 for j in range(len(sliced))]
        tot += _contract_path(ts, ins, path, sliced, vals)
        print("  slice", s_, tot, time.time() - t0, "s", flush=True)
    np.savez(_part("config4"), config4_amplitude=np.array([tot.real, tot.imag]))
    print("config4", tot, time.time() - t0, "s")


def config5():
    from scipy.stats import unitary_group
    from oracle import mps as OM

    n, chi = 64, 128
    rng = np.random.default_rng(64)
    dims = [min(2 ** i, 2 ** (n - i), chi) for i in range(n + 1)]
    tensors = [(rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1])))
               / np.sqrt(2 * dims[i]) for i in range(n)]
    gates = [unitary_group.rvs(4, random_state=5000 + i).reshape(2, 2, 2, 2) for i in range(n - 1)]
    t0 = time.time()
    m = OM.MPSCircuit(n, tensors=[t.astype(np.complex128) for t in tensors], split=OM.split_rules(max_singular_values=chi))
    m.position(0)
    nrm0 = float(abs(m.get_norm()))
    for i in range(n - 1):
        m.apply(gates[i].astype(np.complex128), i, i + 1)
    nrm1 = float(abs(m.get_norm()))
    fid = float(m._fidelity)
    # Schmidt spectrum of the middle bond: singular values of the centre-canonical two-block split
    m.position(n // 2)
    a = np.asarray(m.get_tensors()[n // 2])
    sv = np.linalg.svd(a.reshape(a.shape[0], -1), compute_uv=False)
    np.savez(_part("config5"), config5_fidelity=np.array(fid), config5_norm0=np.array(nrm0), config5_norm1=np.array(nrm1),
             config5_mid_spectrum=sv, config5_bond_dims=np.array(m.get_bond_dimensions()))
    print("config5 fidelity", fid, "norms", nrm0, nrm1, "spectrum head", sv[:4], time.time() - t0, "s")


def merge():
    out = {}
    if os.path.exists(OUT):           # parts that are not regenerated keep their committed values
        with np.load(OUT) as z:
            out.update({k: z[k] for k in z.files})
    for name in ("config3", "config3grad", "config4", "config5"):
        if os.path.exists(_part(name)):
            with np.load(_part(name)) as z:
                out.update({k: z[k] for k in z.files})
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    {"config3": config3, "config3grad": config3grad, "config4": config4, "config5": config5, "merge": merge}[sys.argv[1]]()
