"""Pre-compile the plan-specialised kernels of an HEA-B circuit (H layer, d x (ZZ ladder + rx layer)) on the build host.
usage: python scripts/precompile_hea.py n depth [n depth ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import torch
import tcmi as tc
from tcmi import specialize as S
tc.set_dtype("complex64")
a = [int(x) for x in sys.argv[1:]]
for n, d in zip(a[::2], a[1::2]):
    c = tc.Circuit(n)
    p = torch.zeros(2 * d, n)
    for i in range(n): c.h(i)
    for j in range(d):
        for i in range(n - 1): c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
        for i in range(n): c.rx(i, theta=p[2 * j + 1, i])
    t0 = time.time()
    print(n, d, S.precompile_circuit(c), "%.1f s" % (time.time() - t0), S.STATS)
