"""Host-only op mix of the forward / adjoint plans of the HEA-B workload: plan_opmix.py n d [fwd|adj]
(per pass: rounds, G1M gates by structure class, diagonal ops with their term counts) -- no GPU needed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np
import tcmi as tc
from tcmi import plan as P
from oracle import workloads as W
n, d = int(sys.argv[1]), int(sys.argv[2]); which = sys.argv[3] if len(sys.argv) > 3 else "adj"
tc.set_backend("numpy") if hasattr(tc, "set_backend") and False else None
params = np.random.default_rng(n).normal(0, 0.1, [2 * d, n])
c = tc.Circuit(n); W.hea_b(c, n, d, params, zz=tc.gates._zz_matrix)
recs = c._gate_records()
if which == "adj":
    cfg = P.PlanConfig(R=4, LT=int(os.environ.get("LT", 9)), lowbits=5, vec=2, gen=2)
    pl = P.compile_adjoint_plan(recs, n, cfg, factorized=True)
else:
    cfg = P.PlanConfig(R=5, LT=8, lowbits=5, vec=2, gen=2)
    pl = P.compile_plan(recs, n, cfg, nparams=len(c._params))
R = cfg.R; NR = 1 << R
tot = {}
for i, desc in enumerate(pl.descs):
    w = np.asarray(desc).view(np.int32); pc = P.HDR_WORDS
    m = dict(rounds=int(w[5]), g1=0, shear=0, G1M=0, grads=0, DIAGF=0, hasC=0, Cg=0, B=0, Bapply=0, A=0, tab=0, other={})
    for _ in range(int(w[5])):
        nops = int(w[pc]); q = pc + P.RR_WORDS; end = q + int(w[pc + 1])
        for _o in range(nops):
            op = int(w[q])
            if op == P.OP_G1M:
                mk = int(w[q + 1]); m["G1M"] += 1; m["g1"] += bin(mk & 0xff).count("1"); m["shear"] += bin((mk >> 20) & (mk & 0xff)).count("1")
                if which == "adj":
                    m["grads"] += bin(int(w[q + 3]) & 0xff).count("1"); q += 5 + R
                else:
                    q += 3
            elif op == P.OP_DIAGF:
                cs, hasC, nB, nA = int(w[q + 1]), int(w[q + 2]), int(w[q + 3]), int(w[q + 4])
                m["DIAGF"] += 1; m["hasC"] += hasC; m["tab"] += cs >= 0
                m["Cg"] += int((w[q + 9:q + 9 + NR] >= 0).sum()); m["B"] += nB; m["A"] += nA
                bb = w[q + 9 + NR:q + 9 + NR + 4 * nB].reshape(-1, 4); m["Bapply"] += int((bb[:, 2] >= 0).sum())
                q += 9 + NR + 4 * nB + 2 * nA
            else:
                m["other"][op] = m["other"].get(op, 0) + 1
                q = end; break
        pc = end
    print(i, m)
    for k, v in m.items():
        if k != "other": tot[k] = tot.get(k, 0) + v
print("total", tot)
