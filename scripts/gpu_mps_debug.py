import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from scipy.stats import unitary_group
import tcmi as tc
tc.set_backend("hip"); tc.set_dtype(sys.argv[3] if len(sys.argv) > 3 else "complex128")
cdt = np.complex64 if tc.dtypestr == "complex64" else np.complex128
n, chi = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(64)
dims = [min(2 ** i, 2 ** (n - i), chi) for i in range(n + 1)]
tensors = [((rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1]))) / np.sqrt(2 * dims[i])) for i in range(n)]
tensors = [t.astype(cdt) for t in tensors]
m = tc.MPSCircuit(n, tensors=tensors, split=tc.cons.split_rules(max_singular_values=chi))
print("center at start", m.get_center_position(), "canon dev", float(m._mps.check_canonical()))
m.position(0)
print("after position(0): center", m.get_center_position(), "canon dev", float(m._mps.check_canonical()), "norm", float(abs(m.get_norm())))
for i in range(n - 1):
    m.apply(tc.gates.Gate(unitary_group.rvs(4, random_state=5000 + i).reshape(2, 2, 2, 2).astype(cdt)), i, i + 1)
    if i < 3 or i == n - 2:
        print(i, "center", m.get_center_position(), "canon dev", float(m._mps.check_canonical()), "fid", float(m._fidelity))
ts = m.get_tensors()
a = ts[0].to(torch.complex128); print("site0 gram", torch.einsum("lsr,lsq->rq", a.conj(), a))
