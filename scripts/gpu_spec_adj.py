"""Plan-specialised reverse sweep against the interpreting kernel: gradients, un-computed psi, lambda; sweep times.
usage: python scripts/gpu_spec_adj.py [n] [depth] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
import tcmi.specialize as _SPX; _SPX.ALLOW_PROBE = True   # this script times kernels, also the wrong-result variants of TCMI_SPEC_EXP
from tcmi import specialize as S, executor as X
tc.set_backend("hip"); tc.set_dtype("complex64")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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
os.environ["TCMI_SPECIALIZE"] = "0"
psi = cc.state(pt, full=True).clone()
g = torch.randn(B, psi.shape[1], dtype=torch.float32, device="cuda").to(torch.complex64) + 1j * torch.randn(B, psi.shape[1], device="cuda")
g = g / g.abs().pow(2).sum(1, keepdim=True).sqrt()

def run(flag, want_input_grad):
    os.environ["TCMI_SPECIALIZE"] = flag
    for k in ("_adj", "_adj_short"):
        a = getattr(cc, k, None)
        if a is not None:
            a.pop("spec", None); a.pop("spec_nostore", None)
    t0 = time.time()
    r = cc.vjp(pt, psi, g, want_input_grad=want_input_grad)
    torch.cuda.synchronize()
    return r, time.time() - t0

for wig in (False, True):
    r0, _ = run("0", wig)
    r1, t1 = run("1", wig)
    g0, g1 = (r0[0], r1[0]) if wig else (r0, r1)
    scale = g0.abs().max().item()
    print(f"want_input_grad={wig}: first specialised call {t1:.1f} s {S.STATS}")
    print("   max |grad_interp - grad_spec| = %.3e  (max |grad| %.3e)" % ((g0 - g1).abs().max().item(), scale))
    assert (g0 - g1).abs().max().item() < 2e-5 * max(1.0, scale) or os.environ.get('TCMI_SPEC_EXP')
    if wig:
        e = (r0[1] - r1[1]).abs().max().item()
        print("   max |lambda_interp - lambda_spec| = %.3e, bitwise %s" % (e, bool(torch.equal(r0[1], r1[1]))))
        assert e < 1e-6 or os.environ.get('TCMI_SPEC_EXP')

def timeit(flag, reps=3):
    os.environ["TCMI_SPECIALIZE"] = flag
    for k in ("_adj", "_adj_short"):
        a = getattr(cc, k, None)
        if a is not None:
            a.pop("spec", None); a.pop("spec_nostore", None)
    cc.vjp(pt, psi, g); torch.cuda.synchronize()
    X.EVENT_LOG = []
    X.PASS_EVENTS = []
    t0 = time.time()
    for _ in range(reps): cc.vjp(pt, psi, g)
    torch.cuda.synchronize()
    el = (time.time() - t0) / reps / B * 1e3
    ms = sum(e0.elapsed_time(e1) for tag, e0, e1, *_ in X.EVENT_LOG if tag == "adjoint") / reps / B
    X.EVENT_LOG = None
    npass = 1 + max(i for i, _, _ in X.PASS_EVENTS)
    per = [0.0] * npass
    for i, e0, e1 in X.PASS_EVENTS:
        per[i] += e0.elapsed_time(e1) / reps / B
    X.PASS_EVENTS = None
    print("   mode", flag, "per-pass ms per sample:", " ".join("%.2f" % t for t in per))
    return el, ms
a = timeit("0"); b = timeit("1")
print("interpreter  sweep ms per sample: wall %.2f  kernels %.2f" % a)
print("specialised  sweep ms per sample: wall %.2f  kernels %.2f   (x%.2f)" % (b + (a[1] / b[1],)))
