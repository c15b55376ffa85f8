"""Per-pass times of the forward passes and of the zero-start reverse sweep with live tiles (executor.live_masks), next to
the live fraction of every pass.  usage: python scripts/gpu_live_passes.py [n] [depth] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
import tcmi.specialize as _SPX; _SPX.ALLOW_PROBE = True   # this script times kernels, also the wrong-result variants of TCMI_SPEC_EXP
from tcmi import executor as X
tc.set_backend("hip"); tc.set_dtype("complex64")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
params = torch.from_numpy(np.random.default_rng(28).normal(0, 0.1, [B, 2 * d * n]).astype(np.float32)).cuda()
c = tc.templates.blocks.example_block(tc.Circuit(n), params[0], nlayers=d)
cc = c._compiled()
cc = getattr(cc, "full", cc)
pt = torch.stack([tc.templates.blocks.example_block(tc.Circuit(n), params[b], nlayers=d)._param_tensor().reshape(-1) for b in range(B)])
for _ in range(4):                      # hot: specialised kernels are loaded
    psi = cc.state(pt, full=True)
g = torch.randn(B, psi.shape[1], device="cuda").to(torch.complex64)
for _ in range(2):
    cc.vjp(pt, psi.clone(), g.clone(), from_zero=True)
torch.cuda.synchronize()
adj, masks, fracs = cc._adjoint_from_zero()
X.PASS_EVENTS = []
cc.vjp(pt, psi.clone(), g.clone(), from_zero=True)
torch.cuda.synchronize()
ts = [e0.elapsed_time(e1) for _, e0, e1 in X.PASS_EVENTS]
X.PASS_EVENTS = None
print(f"reverse sweep n={n} d={d} batch {B}: lowbits {adj['cfg'].lowbits}, {len(ts)} passes, {sum(ts) / B:.2f} ms per sample")
for j, (t, f) in enumerate(zip(ts, fracs)):
    print(f"   pass {j}: live fraction {f:.5f}  {t / B:.3f} ms per sample")
# forward passes one by one
fm, ff = cc.zero_start()
zb, rf, cov = cc.zero_bits()
spec = cc._specialised()
out = torch.empty(B, 2 ** cc.n_exec, dtype=torch.complex64, device="cuda")
ptab = torch.empty(B, cc.ptab_size, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
from tcmi import _lib
_lib.check(_lib.lib().tcmi_build_tables(cc.ginfo.data_ptr(), cc.nrec, cc.cpool.data_ptr(), pt.data_ptr(), pt.stride(0), ptab.data_ptr(), ptab.stride(0), B, cc.code, st), "tables")
tf = []
for rep in range(2):
    out[:, 0] = 1.0
    tf = []
    for i in range(len(cc.descs)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cc.run_passes(out, ptab, B, st, first=i, last=i + 1, live=fm, zbits=zb if cov else None)
        e1.record(); torch.cuda.synchronize()
        tf.append(e0.elapsed_time(e1))
print(f"forward n={n} d={d}: lowbits {cc.cfg.lowbits}, {len(tf)} passes, {sum(tf) / B:.2f} ms per state; max |psi - passes| {float((out - psi).abs().max()):.1e}")
for i, (t, f, r) in enumerate(zip(tf, ff, rf)):
    print(f"   pass {i}: live fraction {f:.5f}, read fraction of a live tile {r:.4f}  {t / B:.3f} ms per state")
