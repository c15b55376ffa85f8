"""Time of the fused TFIM measurement (55 strings) on a resident n=26 state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi.executor import get_measure
tc.set_backend("hip"); tc.set_dtype("complex64")
n = 26
strings = []
for i in range(n):
    s = [0] * n; s[i] = 1; strings.append(tuple(s))
for i in range(n - 1):
    s = [0] * n; s[i] = 3; s[i + 1] = 3; strings.append(tuple(s))
cm = get_measure(n, n, tuple(strings), "complex64")
st = torch.randn(2, 2**n, dtype=torch.complex64, device="cuda"); st /= st.norm(dim=1, keepdim=True)
v = cm.run(st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): v = cm.run(st)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
w = torch.ones(2, len(strings), dtype=torch.complex128, device="cuda")
cm.apply_sum(st, w); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): cm.apply_sum(st, w)
torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 5
print(f"measure n={n} B=2: {t*1e3:.2f} ms ({len(cm.descs)} passes)  pauli_sum {t2*1e3:.2f} ms  sumZ0Z1={float(v[0, n].real):.6f}")
