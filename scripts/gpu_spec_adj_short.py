"""One specialised (or interpreted) reverse sweep at n=28 d=12, batch 1, for PMC collection."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import executor as X
tc.set_backend("hip"); tc.set_dtype("complex64")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = 1
params = torch.from_numpy(np.random.default_rng(28).normal(0, 1.0, [B, 2 * d, n]).astype(np.float32)).cuda()
def circ(p):
    c = tc.Circuit(n)
    for i in range(n): c.h(i)
    for j in range(d):
        for i in range(n - 1): c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
        for i in range(n): c.rx(i, theta=p[2 * j + 1, i])
    return c
c = circ(params[0]); cc = c._compiled()
while not isinstance(cc, X.CompiledCircuit):
    cc = cc.full_cc if hasattr(cc, "full_cc") else cc.cc
pt = torch.stack([circ(params[b])._param_tensor().reshape(-1) for b in range(B)])
psi = cc.state(pt, full=True).clone()
g = torch.randn(B, psi.shape[1], device="cuda").to(torch.complex64)
for _ in range(2):
    cc.vjp(pt, psi, g)
torch.cuda.synchronize()
print("done")
