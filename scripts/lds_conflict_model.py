"""Host-only model of the LDS bank conflicts of the generated kernels' exchanges (MI355X_MICROARCH.md, LDS table):
ds_write_b64 = 4 contiguous 16-lane groups on 32 four-byte banks, ds_read_b64 = 2 x 32 lanes on 64 banks,
ds_write_b32 / ds_read_b32 = 2 x 32 lanes on 32 banks.  Prints, per pass and exchange of the n-qubit HEA-B plans, the
LDS-array cycles per wave instruction against the conflict-free count.

    python scripts/lds_conflict_model.py [n] [depth]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))


def cycles(addrs, nbytes, groups, bankmod):
    """addrs: byte address per lane (64); -> LDS-array cycles of one wave instruction."""
    tot = 0
    for g in groups:
        per_bank = {}
        for ln in g:
            a = int(addrs[ln])
            for dw in range(nbytes // 4):
                per_bank.setdefault(((a >> 2) + dw) % bankmod, set()).add((a >> 2) + dw)
        tot += max(len(v) for v in per_bank.values())
    return tot


G16 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
G32 = [list(range(0, 32)), list(range(32, 64))]


def xor_masks(masks, v):
    out = 0
    for i, m in enumerate(masks):
        if (v >> i) & 1:
            out ^= m
    return out


def exchange_cost(em, k, elem_bytes):
    from tcmi import plan as P

    wr, rdn = em.rounds[k], em.rounds[k + 1]
    reg_wr, thr_wr, reg_rd, thr_rd = wr.reg_wr, wr.thr_wr, rdn.reg_rd, rdn.thr_rd
    if elem_bytes == 8 and em.opts.get("own_slots", True):
        tb = {1 << p: i for i, p in enumerate(em.tile_bits)}

        class _R:
            pass

        a_, b_ = _R(), _R()
        a_.reg_tb, a_.thr_tb = [tb[m] for m in wr.reg_phys], [tb[m] for m in wr.thr_phys]
        b_.reg_tb, b_.thr_tb = [tb[m] for m in rdn.reg_phys], [tb[m] for m in rdn.thr_phys]
        A = P.exchange_masks(em.T, a_, b_, planar=False)
        reg_wr, thr_wr = [A[x] for x in a_.reg_tb], [A[x] for x in a_.thr_tb]
        reg_rd, thr_rd = [A[x] for x in b_.reg_tb], [A[x] for x in b_.thr_tb]
    sh = {4: 2, 8: 3}[elem_bytes]
    wc = rc = 0
    NW = 1 << (em.LT - 6)
    for wv in range(NW):
        lanes = np.arange(64) + 64 * wv
        for r in range(em.NR):
            wa = [(xor_masks(thr_wr, int(t)) ^ xor_masks(reg_wr, r)) << sh for t in lanes]
            ra = [(xor_masks(thr_rd, int(t)) ^ xor_masks(reg_rd, r)) << sh for t in lanes]
            if elem_bytes == 8:
                wc += cycles(wa, 8, G16, 32)
                rc += cycles(ra, 8, G32, 64)
            else:
                wc += cycles(wa, 4, G32, 32)
                rc += cycles(ra, 4, G32, 32)
    ideal_w = NW * em.NR * (4 if elem_bytes == 8 else 2)
    ideal_r = NW * em.NR * 2
    return wc, ideal_w, rc, ideal_r


def main():
    import torch

    import tcmi as tc
    from tcmi import cons, executor as X, specialize as S

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    tc.set_dtype("complex64")
    c = tc.templates.blocks.example_block(tc.Circuit(n), torch.zeros(2 * d * n), nlayers=d)
    gates, nparams = c._gate_records(), len(c._params)
    n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, cons.dtypestr, cons._plan_options)
    print(f"forward plan: {len(plan.descs)} passes, R={cfg.R} LT={cfg.LT} T={cfg.T}")
    for i, dsc in enumerate(plan.descs):
        em = S._Forward(dsc)
        tw = ti = 0
        for k in range(em.nrounds - 1):
            wc, iw, rc, ir = exchange_cost(em, k, 4)
            tw += wc + rc
            ti += iw + ir
        print(f"  fwd pass {i}: rounds {em.nrounds}  LDS cycles {tw} (conflict-free {ti})")
    for full, zero in ((True, True),):
        acfg, ap = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, full, zero)
        print(f"adjoint plan (full={full}, zero_start={zero}): {len(ap.descs)} passes, R={acfg.R} LT={acfg.LT} T={acfg.T}")
        for i, dsc in enumerate(ap.descs):
            em = S._Adjoint(np.asarray(dsc), S.adjoint_opts(acfg))
            line = []
            tw = ti = 0
            for k in range(em.nrounds - 1):
                wc, iw, rc, ir = exchange_cost(em, k, 8)
                tw += wc + rc
                ti += iw + ir
                line.append(f"w{wc / iw:.2f}/r{rc / ir:.2f}")
            print(f"  adj pass {i}: rounds {em.nrounds}  LDS cycles x2 vectors {2 * tw} (conflict-free {2 * ti})  " + " ".join(line))


if __name__ == "__main__":
    main()
