"""Does the time of the memory-heavy sweep passes depend on where lambda sits relative to psi?  Both vectors are walked
at identical offsets at the same time; if their base addresses differ by a multiple of the channel / bank interleave
period the two streams collide.  psi and lambda are carved from ONE allocation with a controlled byte offset between
the end of psi and the start of lambda.   usage: python scripts/gpu_alias_probe.py [n] [depth] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import executor as X
tc.set_backend("hip"); tc.set_dtype("complex64")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
params = torch.from_numpy(np.random.default_rng(28).normal(0, 0.1, [B, 2 * d * n]).astype(np.float32)).cuda()
c = tc.templates.blocks.example_block(tc.Circuit(n), params[0], nlayers=d)
cc = c._compiled(); cc = getattr(cc, "full", cc)
pt = torch.stack([tc.templates.blocks.example_block(tc.Circuit(n), params[b], nlayers=d)._param_tensor().reshape(-1) for b in range(B)])
for _ in range(3):
    psi0 = cc.state(pt, full=True)
nel = psi0.shape[1]
g0 = torch.randn(B, nel, device="cuda").to(torch.complex64)
pad_max = 1 << 22      # complex elements
buf = torch.empty(2 * B * nel + pad_max, dtype=torch.complex64, device="cuda")
print("base address of the allocation: %#x" % buf.data_ptr())
for off_bytes in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096 + 256):
    off = off_bytes // 8
    psi = buf[: B * nel].view(B, nel)
    lam = buf[B * nel + off: B * nel + off + B * nel].view(B, nel)
    ts_all = []
    for rep in range(2):
        psi.copy_(psi0); lam.copy_(g0)
        torch.cuda.synchronize()
        X.PASS_EVENTS = []
        cc.vjp(pt, psi, lam, from_zero=True, consume=True)
        torch.cuda.synchronize()
        ts_all = [e0.elapsed_time(e1) / B for _, e0, e1 in X.PASS_EVENTS]
        X.PASS_EVENTS = None
    print(f"lambda at psi_end + {off_bytes:8d} B: sweep {sum(ts_all):6.2f} ms per sample; passes " + " ".join(f"{t:.2f}" for t in ts_all[:7]))
# separate allocations, as the pipeline makes them
for rep in range(3):
    psi, lam = psi0.clone(), g0.clone()
    torch.cuda.synchronize()
    X.PASS_EVENTS = []
    cc.vjp(pt, psi, lam, from_zero=True, consume=True)
    torch.cuda.synchronize()
    ts_all = [e0.elapsed_time(e1) / B for _, e0, e1 in X.PASS_EVENTS]
    X.PASS_EVENTS = None
    print(f"separate allocations (psi %#x, lambda %#x): sweep {sum(ts_all):6.2f}; passes " % (psi.data_ptr(), lam.data_ptr()) + " ".join(f"{t:.2f}" for t in ts_all[:7]))
    del psi, lam
