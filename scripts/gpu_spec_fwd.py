"""Plan-specialised forward passes against the interpreting kernel: same state (bitwise or to rounding), pass times.
usage: python scripts/gpu_spec_fwd.py [n] [depth] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
import tcmi.specialize as _SPX; _SPX.ALLOW_PROBE = True   # this script times kernels, also the wrong-result variants of TCMI_SPEC_EXP
from tcmi import specialize as S
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
flat = torch.stack([params[b].reshape(-1) for b in range(B)])
pf = cc_params = None
def run(modeflag):
    os.environ["TCMI_SPECIALIZE"] = modeflag
    cc._spec_fwd = None
    # parameters in the plan's order: go through the public path once for correctness of ordering
    outs = []
    for b in range(B):
        outs.append(circ(params[b]).wavefunction().clone())
    torch.cuda.synchronize()
    return torch.stack(outs)
ref = run("0")
t0 = time.time(); got = run("1"); print("first specialised run (incl. compile) %.1f s" % (time.time() - t0), S.STATS)
err = (ref - got).abs().max().item()
print("max |interp - spec| =", err, " norm", got[0].abs().pow(2).sum().item(), " bitwise equal:", bool(torch.equal(ref, got)))
assert err < 2e-6 or os.environ.get('TCMI_SPEC_EXP')
pt = torch.stack([circ(params[b])._param_tensor().reshape(-1) for b in range(B)])
def timeit(modeflag, reps=5):
    os.environ["TCMI_SPECIALIZE"] = modeflag
    cc._spec_fwd = None
    out = torch.empty(B, 2 ** cc.n_exec, dtype=torch.complex64, device="cuda")
    cc.state(pt, out=out); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        cc.state(pt, out=out)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps / B * 1e3, out
ti, oi = timeit("0")
oi = oi.clone()
ts, os_ = timeit("1")
print("plan:", type(cc).__name__, "passes", len(cc.descs), " batched state equal:", bool(torch.equal(oi, os_)))
print("interpreter  ms per state: %.3f" % ti)
print("specialised  ms per state: %.3f   (x%.2f)" % (ts, ti / ts))
from tcmi import executor as X
for flag in ("0", "1"):
    os.environ["TCMI_SPECIALIZE"] = flag
    cc._spec_fwd = None
    out = torch.empty(B, 2 ** cc.n_exec, dtype=torch.complex64, device="cuda")
    ptab = torch.empty(B, cc.ptab_size, dtype=torch.float32, device="cuda")
    cc.state(pt, out=out)
    st = torch.cuda.current_stream().cuda_stream
    from tcmi import _lib
    p2 = pt.contiguous()
    _lib.check(cc._lib.tcmi_build_tables(cc.ginfo.data_ptr(), cc.nrec, cc.cpool.data_ptr(), p2.data_ptr(), p2.stride(0),
                                         ptab.data_ptr(), ptab.stride(0), B, cc.code, st), "build")
    ts_ = []
    for i in range(len(cc.descs)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): cc.run_passes(out, ptab, B, st, first=i, last=i + 1)
        e1.record(); torch.cuda.synchronize()
        ts_.append(e0.elapsed_time(e1) / 3 / B)
    print("mode", flag, "per-pass ms per state:", " ".join("%.3f" % t for t in ts_), " sum %.3f" % sum(ts_))
