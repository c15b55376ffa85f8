"""One adjoint sweep configuration for profiling: gpu_adj_one.py n d batch reps  (prints per-pass times with op mix)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import plan as P, _lib
from tcmi.executor import ATOMIC_COPIES
from oracle import workloads as W
n, d, B, reps = (int(x) for x in sys.argv[1:5])
tc.set_backend("hip"); tc.set_dtype("complex64"); tc.set_contractor("plain")
params = np.random.default_rng(n).normal(0, 0.1, [2 * d, n]).astype(np.float32)
c = tc.Circuit(n); W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
cc = c._compiled(); cc = getattr(cc, "full", cc); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
st = cc.state(p, full=True); g = torch.randn_like(st); torch.cuda.synchronize()
adj = cc._adjoint(full=False); cfg = adj["cfg"]; lib = cc._lib
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cc.vjp(p, st, g); torch.cuda.synchronize()
e0.record()
for _ in range(reps): cc.vjp(p, st, g)
e1.record(); torch.cuda.synchronize()
print(f"adjoint n={n} d={d} B={B}: {e0.elapsed_time(e1)/reps:.3f} ms/sweep, {len(adj['descs'])} passes, R={cfg.R} LT={cfg.LT}")
if "--passes" in sys.argv:
    def mix(desc):
        w = np.asarray(desc).view(np.int32); pc = P.HDR_WORDS; m = dict(g1=0, G1M=0, DIAG=0, terms=0, rounds=int(w[5]))
        for _ in range(int(w[5])):
            nops = int(w[pc]); q = pc + P.RR_WORDS
            for _o in range(nops):
                op = int(w[q])
                if op == P.OP_G1M: m["g1"] += bin(int(w[q + 1]) & 0xff).count("1"); m["G1M"] += 1; q += 5 + cfg.R
                elif op == P.OP_DIAG:
                    nA, nB, nC = int(w[q + 1]), int(w[q + 2]), int(w[q + 3]); m["DIAG"] += 1
                    gs = w[q + 5 + nA + 2 * nB + nC: q + 5 + 2 * nA + 3 * nB + 2 * nC]; m["terms"] += int((gs >= 0).sum())
                    q += 5 + 2 * nA + 3 * nB + 2 * nC
                elif op == P.OP_G2: q += 6
                elif op == P.OP_DIAGF:
                    nB, nA = int(w[q + 3]), int(w[q + 4]); m["DIAG"] += 1
                    m["terms"] += int((w[q + 9:q + 9 + 2**cfg.R] >= 0).sum()) + nB + nA
                    q += 9 + 2**cfg.R + 4 * nB + 2 * nA
            pc = q
        return m
    nel = 2**cc.n_exec; stream = torch.cuda.current_stream().cuda_stream
    a = st.clone(); lam = g.clone()
    ptab = torch.empty(B, max(1, adj["plan"].ptab_size), dtype=cc.rdtype, device=cc.device)
    _lib.check(lib.tcmi_build_adjoint_tables(adj["ginfo"].data_ptr(), int(adj["ginfo"].shape[0]), adj["cpool"].data_ptr(), p.data_ptr(), p.stride(0), ptab.data_ptr(), ptab.stride(0), B, cc.code, stream), "b")
    gout = torch.zeros(B, ATOMIC_COPIES, adj["nslots"], dtype=torch.float64, device=cc.device)
    tot = 0
    for i, dsc in enumerate(adj["descs"]):
        torch.cuda.synchronize(); e0.record()
        _lib.check(lib.tcmi_run_adjoint_pass(a.data_ptr(), lam.data_ptr(), nel, B, cc.n_exec, cfg.R, cfg.LT, dsc.data_ptr(), adj["ctab"].data_ptr(), ptab.data_ptr(), ptab.stride(0), gout.data_ptr(), gout.stride(0), ATOMIC_COPIES, gout.stride(1), cc.code, int(cfg.gen == 2), stream), "p")
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1); tot += ms
        print(f"pass {i:2d}: {ms*1e3:8.1f} us  {4*B*2**n*8/(ms*1e-3)/1e9:6.0f} GB/s  {mix(adj['plan'].descs[i])}", flush=True)
    print(f"total {tot:.2f} ms")
