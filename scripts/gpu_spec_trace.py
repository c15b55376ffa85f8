"""Per-workgroup timestamps of one specialised forward pass (TCMI_SPEC_EXP=trace): when does a workgroup start, when has
its tile arrived, when is its compute done -- are the workgroups that share a CU in lock step?
usage: TCMI_SPEC_EXP=trace python scripts/gpu_spec_trace.py [pass index]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
os.environ["TCMI_SPECIALIZE"] = "1"
import numpy as np, torch
import tcmi as tc
from tcmi import executor as X, _lib
tc.set_backend("hip"); tc.set_dtype("complex64")
n, d, B = 28, 12, 1
ip = int(sys.argv[1]) if len(sys.argv) > 1 else 1
params = torch.from_numpy(np.random.default_rng(28).normal(0, 1.0, [B, 2 * d, n]).astype(np.float32)).cuda()
def circ(p):
    c = tc.Circuit(n)
    for i in range(n): c.h(i)
    for j in range(d):
        for i in range(n - 1): c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
        for i in range(n): c.rx(i, theta=p[2 * j + 1, i])
    return c
c = circ(params[0]); cc = c._compiled()
pt = torch.stack([circ(params[b])._param_tensor().reshape(-1) for b in range(B)])
out = torch.empty(B, 2 ** n, dtype=torch.complex64, device="cuda")
ntile = 2 ** (n - cc.cfg.T)
buf = torch.zeros(32 * ntile * B + 64, dtype=torch.int64, device="cuda")
cc.ctab = buf.view(torch.float32)          # the trace build writes its timestamps through the (unused) constant-table pointer
cc.state(pt, out=out)
ptab = torch.empty(B, cc.ptab_size, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
p2 = pt.contiguous()
_lib.check(cc._lib.tcmi_build_tables(cc.ginfo.data_ptr(), cc.nrec, cc.cpool.data_ptr(), p2.data_ptr(), p2.stride(0),
                                     ptab.data_ptr(), ptab.stride(0), B, cc.code, st), "build")
torch.cuda.synchronize(); buf.zero_()
cc.run_passes(out, ptab, B, st, first=ip, last=ip + 1)
torch.cuda.synchronize()
t = buf[: 32 * ntile].view(ntile, 32).cpu().numpy().astype(np.int64)
t0 = t[:, 0].min()
start, arr, done, hw = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0, t[:, 3]   # us (100 MHz)
print("pass", ip, "workgroups", ntile, " kernel span %.1f us" % done.max())
print("load wait   us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (np.mean(arr - start), *np.percentile(arr - start, [10, 50, 90])))
print("compute     us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % (np.mean(done - arr), *np.percentile(done - arr, [10, 50, 90])))
# HW_ID: wave_id [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]  (xcc via blockIdx % 8)
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
xcd = np.arange(ntile) % 8
key = xcd * 1000 + cu
print("distinct (xcd, se, sh, cu):", len(np.unique(key)), " wave ids seen:", sorted(set((hw & 0xF).tolist())))
# phase coherence on one CU: starts of the workgroups that ran there, in time order
k0 = key[0]
sel = np.where(key == k0)[0]
order = sel[np.argsort(start[sel])]
print("CU of workgroup 0 ran", len(sel), "workgroups; first 24 (start, arrived, done, wave id):")
for i in order[:24]:
    print("   %8.1f %8.1f %8.1f  w%d" % (start[i], arr[i], done[i], hw[i] & 0xF))
# how many workgroups of the whole chip are waiting for their tile at a given time
ts = np.linspace(0, done.max(), 400)
nload = [(np.sum((start <= x) & (arr > x))) for x in ts]
ncomp = [(np.sum((arr <= x) & (done > x))) for x in ts]
print("workgroups waiting for their tile over time (40 samples):", " ".join(str(v) for v in nload[::10]))
print("workgroups computing over time              (40 samples):", " ".join(str(v) for v in ncomp[::10]))

nr = len(cc.plan.passes[ip].rounds)
prev = arr
print("per round (compute us, exchange us), means over workgroups:")
for k in range(nr - 1):
    a_, b_ = (t[:, 4 + 2 * k] - t0) / 100.0, (t[:, 5 + 2 * k] - t0) / 100.0
    print("   round %d: compute %.2f  exchange %.2f" % (k, np.mean(a_ - prev), np.mean(b_ - a_)))
    prev = b_
print("   round %d: compute %.2f" % (nr - 1, np.mean(done - prev)))
