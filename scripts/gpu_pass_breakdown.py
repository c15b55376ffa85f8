"""Per-pass timing of the state-vector plan next to each pass's op mix: gpu_pass_breakdown.py n d [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from tcmi import plan as P, _lib
from oracle import workloads as W
n, d = int(sys.argv[1]), int(sys.argv[2]); B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tc.set_backend("hip"); tc.set_dtype("complex64"); tc.set_contractor("plain", **({"lowbits": int(os.environ["LOWBITS"])} if os.environ.get("LOWBITS") else {}))
params = np.random.default_rng(n).uniform(0, 2 * np.pi, [2 * d, n]).astype(np.float32)
c = tc.Circuit(n); W.hea_b(c, n, d, tc.backend.convert_to_tensor(params), zz=tc.gates._zz_matrix)
cc = c._compiled(); p = c._param_tensor().reshape(1, -1).repeat(B, 1)
out = cc.state(p, full=True); torch.cuda.synchronize()

def mix(desc):
    w = np.asarray(desc).view(np.int32); pc = P.HDR_WORDS; m = dict(g1=0, G1M=0, DIAGC=0, DIAGB=0, DIAG=0, G2=0, rounds=int(w[5]))
    for _ in range(int(w[5])):
        nops = int(w[pc]); q = pc + P.RR_WORDS
        for _o in range(nops):
            op = int(w[q])
            if op == P.OP_G1M: m["g1"] += bin(int(w[q + 1]) & 0xff).count("1"); m["G1M"] += 1; q += 3
            elif op == P.OP_DIAG: m["DIAG"] += 1; q += 5 + int(w[q + 1]) + 2 * int(w[q + 2]) + int(w[q + 3])
            elif op == P.OP_DIAGC: m["DIAGC"] += 1; q += 2
            elif op == P.OP_DIAGB: m["DIAGB"] += 1; q += 4
            elif op == P.OP_DIAGB2: m["DIAGB"] += 1; q += 5
            elif op == P.OP_DIAGCW: m["DIAGC"] += 1; q += 6
            elif op == P.OP_G2: m["G2"] += 1; q += 4
        pc = q
    return m

ptab = torch.empty(B, cc.ptab_size, dtype=cc.rdtype, device=cc.device)
stream = torch.cuda.current_stream().cuda_stream
_lib.check(cc._lib.tcmi_build_tables(cc.ginfo.data_ptr(), cc.nrec, cc.cpool.data_ptr(), p.data_ptr(), p.stride(0), ptab.data_ptr(), ptab.stride(0), B, cc.code, stream), "build")
tot = 0.0
for i in range(len(cc.descs)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cc.run_passes(out, ptab, B, stream, i, i + 1); torch.cuda.synchronize()
    e0.record()
    for _ in range(3): cc.run_passes(out, ptab, B, stream, i, i + 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3; tot += ms
    gbs = 2 * B * 2**n * 8 / (ms * 1e-3) / 1e9
    print(f"pass {i:2d}: {ms*1e3:8.1f} us  {gbs:6.0f} GB/s  {mix(cc.plan.descs[i])}", flush=True)
print(f"total {tot:.3f} ms")
