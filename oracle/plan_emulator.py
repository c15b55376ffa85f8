"""Numpy emulator of the tile-VM pass descriptors (TEST INFRASTRUCTURE, see oracle/__init__.py).

Interprets exactly the int32 descriptor words that ``tcmi.plan.encode_pass`` hands to the HIP
kernel (``tcmi_run_pass``), at thread granularity: per-thread register arrays, the LDS exchange
through the descriptor's slot masks, the builder program that turns parameters into gate tables.
It lets the CPU test-suite validate the plan compiler (scheduling, layouts, diagonal-term
classification, table slots) without a GPU, and reports LDS bank-conflict counts of a plan.
It is never imported by the product package.
"""

import numpy as np

MAGIC = 0x54434D31
HDR_WORDS = 24
RR_WORDS = 50
OP_G1, OP_G2, OP_DIAG, OP_G1M, OP_EXPECT, OP_DIAGC, OP_DIAGB = 1, 2, 3, 4, 5, 6, 7
OP_DIAGF, OP_DIAGB2, OP_DIAGCW, OP_EXPECT2, OP_XFOLD, OP_DFOLD, OP_XFOLD2 = 8, 9, 10, 11, 12, 13, 14
R_MAX = 6
CONST_FLAG = 1 << 30
BK_TRIG, BK_COEF, BK_SELECT, BK_PHASE = 1, 2, 5, 6
SHEAR2_CMIN = 0.5      # plan.SHEAR2_CMIN / TCMI_SHEAR2_CMIN
SCALE_TERM = 1 << 16   # plan.SCALE_TERM


def _phase_entry(cpool, off, nterms, r, params, invert=False):
    """One BK_PHASE table entry [B] (complex): exp(2 pi i sum_t s_t(r) (k_t theta + o_t)) times the real scale terms
    c^(+-1), c = cos(k theta + o) -- 1 where |c| < SHEAR2_CMIN (the gate then ran in three-shear form).
    ``invert``: reciprocal scale (the lambda table of the adjoint sweep)."""
    B = params.shape[0]
    phi = np.zeros(B)
    mag = np.ones(B)
    for t in range(nterms):
        kt, ot, pt, rm = cpool[off + 4 * t: off + 4 * t + 4]
        rm = int(rm)
        odd = bin(r & rm & (SCALE_TERM - 1)).count("1") & 1
        if rm & SCALE_TERM:
            c = np.cos(kt * params[:, int(pt)] + ot)
            ok = np.abs(c) >= SHEAR2_CMIN
            f = np.where(ok, c, 1.0)
            mag *= (1.0 / f) if (odd != invert) else f
            continue
        phi += (-1.0 if odd else 1.0) * (kt * params[:, int(pt)] + ot)
    return mag * np.exp(2j * np.pi * phi)


def build_table(ginfo, cpool, params, ptab_size):
    """The builder kernel (``tcmi_build_tables``): params [B, P] -> ptab [B, ptab_size] float64."""
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    B = params.shape[0]
    ptab = np.zeros((B, ptab_size), dtype=np.float64)
    for rec in np.asarray(ginfo).reshape(-1, 8):
        kind, slot, pidx, dim, off = (int(x) for x in rec[:5])
        k, o = cpool[off], cpool[off + 1]
        th = params[:, pidx]
        if kind == BK_PHASE:
            e = _phase_entry(cpool, off, dim, int(rec[5]), params)
            ptab[:, slot] = e.real
            ptab[:, slot + 1] = e.imag
        elif kind == BK_TRIG:
            a = k * th + o
            nn = dim * dim
            c = [
                (cpool[off + 2 + 2 * nn * i: off + 2 + 2 * nn * (i + 1)]).reshape(nn, 2)
                for i in range(3)
            ]
            c = [x[:, 0] + 1j * x[:, 1] for x in c]
            m = c[0][None, :] + np.cos(a)[:, None] * c[1][None, :] + np.sin(a)[:, None] * c[2][None, :]
            if int(rec[5]):
                ptab[:, slot: slot + 8] = _shear_params(m, int(rec[5]), two=bool(rec[6]))
                continue
            ptab[:, slot: slot + 2 * nn: 2] = m.real
            ptab[:, slot + 1: slot + 2 * nn: 2] = m.imag
        elif kind == BK_COEF:
            ptab[:, slot] = k * th + o
        elif kind == BK_SELECT:
            nn = dim * dim
            for b in range(B):
                idx = int(np.clip(np.rint(th[b]), 0, int(k) - 1))
                t = cpool[off + 2 + 2 * nn * idx: off + 2 + 2 * nn * (idx + 1)]
                ptab[b, slot: slot + 2 * nn] = t
        else:
            raise ValueError(kind)
    return ptab


def _shear_params(m, flavor, two=False):
    """{u, v, sign, flag, 0...} of the shear form of the rotation matrices m [B, 4] (row-major 2x2), as the builder
    kernels write them: sign * m = S(u) L(v) S(u) on (x, y) (flavor 1) or on (x, i y) (flavor 2), flag = 0; with
    ``two`` and |m00| >= SHEAR2_CMIN the two-shear form m = diag(m00, 1 / m00) L(v) S(u), sign = 1, flag = 2."""
    m = np.asarray(m).reshape(-1, 4)
    out = np.zeros((m.shape[0], 8))
    a = m[:, 0].real
    c = m[:, 2].real if flavor == 1 else m[:, 2].imag
    sg = np.where(a < 0, -1.0, 1.0)
    a3, c3 = a * sg, c * sg
    safe = np.abs(c3) > 1e-30
    num = (a3 - 1.0) if flavor == 1 else (1.0 - a3)
    out[:, 0] = np.where(safe, num / np.where(safe, c3, 1.0), 0.0)
    out[:, 1] = c3
    out[:, 2] = sg
    if two:
        ok = np.abs(a) >= SHEAR2_CMIN
        asafe = np.where(ok, a, 1.0)
        out[:, 0] = np.where(ok, (-c if flavor == 1 else c) / asafe, out[:, 0])
        out[:, 1] = np.where(ok, c * a, out[:, 1])
        out[:, 2] = np.where(ok, 1.0, out[:, 2])
        out[:, 3] = np.where(ok, 2.0, 0.0)
    return out


def _apply_shear(regs, r0, r1, u, v, flavor, two=False):
    """x += u y', y' += v x, x += u y' with y' = y (flavor 1) or i y (flavor 2: y += v (i x)); ``two``: without the
    third step (the real factor diag(c, 1/c) that completes the rotation is a pending scale term)."""
    f = 1.0 if flavor == 1 else 1j
    x, y = regs[..., r0].copy(), regs[..., r1].copy()
    x = x + u * f * y
    y = y + v * f * x
    if not two:
        x = x + u * f * y
    regs[..., r0] = x
    regs[..., r1] = y


def _xor_masks(idx, masks):
    out = np.zeros_like(idx)
    for i, m in enumerate(masks):
        out ^= np.where((idx >> i) & 1, np.uint32(m), np.uint32(0)).astype(np.uint32)
    return out


def _parity(x):
    x = x.astype(np.uint64)
    p = np.zeros_like(x)
    for s in (32, 16, 8, 4, 2, 1):
        x = x ^ (x >> np.uint64(s))
    return (x & np.uint64(1)).astype(np.int64)


def run_pass(state, desc, ctab, ptab_row, eout=None):
    """Apply one pass descriptor in place to ``state`` (complex array of 2^n).  Measurement passes
    accumulate into ``eout`` (complex array, one entry per Pauli term) and do not store."""
    d = np.asarray(desc).view(np.uint32).astype(np.int64)
    assert d[0] == MAGIC
    n, T, R, LT, nrounds = (int(x) for x in d[1:6])
    flags = int(d[6])
    assert T == R + LT and state.size == 2**n
    tile_bits = [int(x) for x in d[8: 8 + T]]
    assert tile_bits == sorted(set(tile_bits))
    nwg, nth, NR = 2 ** (n - T), 2**LT, 2**R
    wg = np.arange(nwg, dtype=np.uint64)
    for p in tile_bits:
        low = np.uint64((1 << p) - 1)
        wg = ((wg & ~low) << np.uint64(1)) | (wg & low)
    wg_base = wg.astype(np.uint32)
    tid = np.arange(nth, dtype=np.uint32)
    rid = np.arange(NR, dtype=np.uint32)

    def tab(slot, cnt):
        slot = int(slot)
        if slot & CONST_FLAG:
            return np.asarray(ctab)[(slot & ~CONST_FLAG): (slot & ~CONST_FLAG) + cnt]
        return np.asarray(ptab_row)[slot: slot + cnt]

    pc = HDR_WORDS
    regs = None
    lds = None
    pass_sign = 1.0   # product of the signs pulled out of the shear-form gates, applied before the store
    for k in range(nrounds):
        rr = d[pc: pc + RR_WORDS]
        nops, opwords = int(rr[0]), int(rr[1])
        reg_pm, thr_pm = rr[2: 2 + R], rr[8: 8 + LT]
        reg_rm, thr_rm = rr[18: 18 + R], rr[24: 24 + LT]
        reg_wm, thr_wm = rr[34: 34 + R], rr[40: 40 + LT]
        tphys = _xor_masks(tid, thr_pm)
        rphys = _xor_masks(rid, reg_pm)
        # every (thread, reg) pair must address a distinct tile element
        allphys = (tphys[:, None] | rphys[None, :]).reshape(-1)
        assert np.unique(allphys).size == nth * NR
        tile_mask = 0
        for p in tile_bits:
            tile_mask |= 1 << p
        assert int(np.bitwise_or.reduce(allphys)) == tile_mask
        gidx = (wg_base[:, None, None] | tphys[None, :, None] | rphys[None, None, :]).astype(np.int64)
        if k == 0:
            regs = state[gidx]
        else:
            slot = (_xor_masks(tid, thr_rm)[:, None] ^ _xor_masks(rid, reg_rm)[None, :]).astype(np.int64)
            assert np.unique(slot).size == nth * NR and slot.max() < nth * NR
            regs = lds[:, slot]
        # ops
        q = pc + RR_WORDS
        for _ in range(nops):
            op = int(d[q])
            if op in (OP_G1, OP_G1M):
                if op == OP_G1:
                    todo = [(int(d[q + 1]) & 0xFF, int(d[q + 1]) >> 8, d[q + 2])]
                    q += 3
                else:
                    mk, base_slot = int(d[q + 1]), int(d[q + 2])
                    todo = [(j, (mk >> (8 + 2 * j)) & 3, base_slot + 8 * j) for j in range(R) if (mk >> j) & 1]
                    q += 3
                for j, kind, slot in todo:
                    if op == OP_G1M and (mk >> (20 + j)) & 1:   # three-shear form {u, v, sign}
                        tb8 = tab(slot, 8)
                        bit = (rid >> j) & 1
                        r0 = rid[bit == 0]
                        _apply_shear(regs, r0, r0 | (1 << j), tb8[0], tb8[1], kind, two=tb8[3] != 0)
                        pass_sign *= tb8[2]
                        continue
                    m = tab(slot, 8)
                    m = (m[0::2] + 1j * m[1::2]).reshape(2, 2)
                    if kind == 1:      # the kernel reads only the real parts
                        m = m.real.astype(np.complex128)
                    elif kind == 2:    # real diagonal, imaginary off-diagonal
                        m = np.array([[m[0, 0].real, 1j * m[0, 1].imag], [1j * m[1, 0].imag, m[1, 1].real]])
                    bit = (rid >> j) & 1
                    r0 = rid[bit == 0]
                    r1 = r0 | (1 << j)
                    a0, a1 = regs[..., r0].copy(), regs[..., r1].copy()
                    regs[..., r0] = m[0, 0] * a0 + m[0, 1] * a1
                    regs[..., r1] = m[1, 0] * a0 + m[1, 1] * a1
            elif op == OP_G2:
                ja, kind, jb, slot = int(d[q + 1]) & 0xFF, int(d[q + 1]) >> 8, int(d[q + 2]), d[q + 3]
                assert ja < jb
                m = tab(slot, 32)
                m = (m[0::2] + 1j * m[1::2]).reshape(4, 4)
                if kind == 1:
                    m = np.eye(4)[[0, 1, 3, 2]]
                elif kind == 2:
                    m = np.eye(4)[[0, 3, 2, 1]]
                elif kind == 3:
                    m = np.eye(4)[[0, 2, 1, 3]]
                base = rid[(((rid >> ja) & 1) == 0) & (((rid >> jb) & 1) == 0)]
                idx = [base | (xa << ja) | (xb << jb) for xa in (0, 1) for xb in (0, 1)]
                a = [regs[..., ix].copy() for ix in idx]
                for o in range(4):
                    regs[..., idx[o]] = sum(m[o, i] * a[i] for i in range(4))
                q += 4
            elif op == OP_DIAG:
                nA, nB, nC, base_slot = (int(x) for x in d[q + 1: q + 5])
                assert nA % 8 == 0 and nB % 8 == 0
                q += 5
                cf = np.asarray(ptab_row)[base_slot: base_slot + nA + nB + nC]
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)  # [nwg, nth]
                phi = np.zeros((nwg, nth, NR), dtype=np.float64)
                for e in range(nA):
                    sgn = 1 - 2 * _parity(tidx & np.uint64(int(d[q + e])))
                    phi += (float(cf[e]) * sgn)[:, :, None]
                q += nA
                for e in range(nB):
                    mask, j = int(d[q + e]), int(d[q + nB + e])
                    sgn = 1 - 2 * _parity(tidx & np.uint64(mask))
                    z = 1 - 2 * ((rid >> j) & 1).astype(np.int64)
                    phi += (float(cf[nA + e]) * sgn)[:, :, None] * z[None, None, :]
                q += 2 * nB
                for e in range(nC):
                    z = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(int(d[q + e])))
                    phi += float(cf[nA + nB + e]) * z[None, None, :]
                q += nC
                regs = regs * np.exp(2j * np.pi * phi).astype(regs.dtype)
            elif op == OP_DIAGB:
                j, mask, slot = int(d[q + 1]), int(d[q + 2]), int(d[q + 3])
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
                sgn = (1 - 2 * _parity(tidx & np.uint64(mask)))[:, :, None] * (
                    1 - 2 * ((rid >> j) & 1).astype(np.int64))[None, None, :]
                cs, sn = float(ptab_row[slot]), float(ptab_row[slot + 1])
                regs = regs * (cs + 1j * sn * sgn).astype(regs.dtype)
                q += 4
            elif op == OP_DIAGC:
                tb = np.asarray(ptab_row)[int(d[q + 1]): int(d[q + 1]) + 2 * NR]
                regs = regs * (tb[0::2] + 1j * tb[1::2]).astype(regs.dtype)[None, None, :]
                q += 2
            elif op == OP_DIAGCW:
                # table variant picked per thread (per wave on the device) by the parities of the selector masks
                slot, nsel = int(d[q + 1]), int(d[q + 2])
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
                v = np.zeros(tidx.shape, dtype=np.int64)
                for k_ in range(nsel):
                    v += _parity(tidx & np.uint64(int(d[q + 3 + k_]))).astype(np.int64) << k_
                tb = np.asarray(ptab_row)[slot: slot + 2 * NR * (1 << nsel)]
                tbc = (tb[0::2] + 1j * tb[1::2]).reshape(1 << nsel, NR)
                regs = regs * tbc[v].astype(regs.dtype)
                # the selector bits must be uniform over every wave of 64 threads
                for k_ in range(nsel):
                    pk = _parity(tidx & np.uint64(int(d[q + 3 + k_]))).reshape(nwg, -1, min(64, nth))
                    assert (pk == pk[:, :, :1]).all()
                q += 6
            elif op == OP_DIAGB2:
                # factor = table[s1 + 2 s2] for register bit clear, its conjugate for bit set
                j, m1, m2, slot = int(d[q + 1]), int(d[q + 2]), int(d[q + 3]), int(d[q + 4])
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
                c = (_parity(tidx & np.uint64(m1)) + 2 * _parity(tidx & np.uint64(m2))).astype(np.int64)
                tb = np.asarray(ptab_row)[slot: slot + 8]
                e = (tb[0::2] + 1j * tb[1::2])[c]                      # [nwg, nth]
                bit = ((rid >> j) & 1).astype(bool)
                regs = regs * np.where(bit[None, None, :], np.conj(e)[:, :, None], e[:, :, None]).astype(regs.dtype)
                q += 5
            elif op == OP_EXPECT:
                nZ, nX = int(d[q + 1]), int(d[q + 2])
                q += 3
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
                for _z in range(nZ):
                    zr, zm, oi = int(d[q]), int(d[q + 1]), int(d[q + 2])
                    sr = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(zr))
                    st = 1 - 2 * _parity(tidx & np.uint64(zm))
                    eout[oi] += np.sum((np.abs(regs) ** 2) * sr[None, None, :] * st[:, :, None])
                    q += 3
                for _x in range(nX):
                    xr, zr, zm, oi = (int(v) for v in d[q: q + 4])
                    assert bin(xr).count("1") in (1, 2)
                    sr = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(zr))
                    st = 1 - 2 * _parity(tidx & np.uint64(zm))
                    eout[oi] += np.sum(np.conj(regs[..., rid ^ xr]) * regs * sr[None, None, :] * st[:, :, None])
                    q += 4
            elif op == OP_EXPECT2:
                # {11, nX, gmask, X: (xr, zr, zm, out)*, per set bit k of gmask: count, (zm, out)*}
                nX, gmask = int(d[q + 1]), int(np.uint32(d[q + 2]))
                q += 3
                tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
                for _x in range(nX):
                    xr, zr, zm, oi = (int(v) for v in d[q: q + 4])
                    assert bin(xr).count("1") in (1, 2)
                    sr = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(zr))
                    st = 1 - 2 * _parity(tidx & np.uint64(zm))
                    eout[oi] += np.sum(np.conj(regs[..., rid ^ xr]) * regs * sr[None, None, :] * st[:, :, None])
                    q += 4
                for kreg in range(NR):
                    if not (gmask >> kreg) & 1:
                        continue
                    cnt = int(d[q])
                    q += 1
                    sr = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(kreg))
                    for _z in range(cnt):
                        zm, oi = int(d[q]), int(d[q + 1])
                        st = 1 - 2 * _parity(tidx & np.uint64(zm))
                        eout[oi] += np.sum((np.abs(regs) ** 2) * sr[None, None, :] * st[:, :, None])
                        q += 2
            else:
                raise ValueError(f"bad opcode {op} at word {q}")
        assert q == pc + RR_WORDS + opwords
        pc = q
        if k < nrounds - 1:
            slot = (_xor_masks(tid, thr_wm)[:, None] ^ _xor_masks(rid, reg_wm)[None, :]).astype(np.int64)
            assert np.unique(slot).size == nth * NR and slot.max() < nth * NR
            lds = np.zeros((nwg, nth * NR), dtype=regs.dtype)
            lds[:, slot] = regs
        elif not (flags & 1):
            state[gidx] = regs * pass_sign
    assert pc == d.size
    return state


def run_plan(plan, params=None, dtype=np.complex128, batch_index=0):
    """Execute a CompiledPlan on |0..0> and return the flat state."""
    state = np.zeros(2**plan.n, dtype=dtype)
    state[0] = 1.0
    if plan.ptab_size:
        if params is None:
            params = np.zeros(1)
        ptab = build_table(plan.ginfo, plan.cpool, params, plan.ptab_size)[batch_index]
    else:
        ptab = np.zeros(0)
    for desc in plan.descs:
        run_pass(state, desc, plan.ctab, ptab)
    return state


def lds_conflicts(desc, planar=False):
    """Worst-case LDS bank multiplicity of every exchange in a pass: (write_ways, read_ways) per
    exchange.  8-byte elements (first-generation kernels): ds_write_b64, 16-lane groups on 32 banks; ds_read_b64,
    32-lane groups on 64 banks.  ``planar`` (packed kernels, one 4-byte plane at a time): ds_write_b32 and
    ds_read_b32, 32-lane groups on 32 four-byte banks."""
    d = np.asarray(desc).view(np.uint32).astype(np.int64)
    n, T, R, LT, nrounds = (int(x) for x in d[1:6])
    pc = HDR_WORDS
    out = []
    lanes = np.arange(min(64, 2**LT), dtype=np.uint32)
    rid = np.arange(2**R, dtype=np.uint32)
    recs = []
    for k in range(nrounds):
        rr = d[pc: pc + RR_WORDS]
        recs.append(rr)
        pc += RR_WORDS + int(rr[1])

    def ways(thr_m, reg_m, group, nslots):
        worst = 1
        base = _xor_masks(lanes, thr_m)
        for r in rid:
            s = base ^ _xor_masks(np.array([r], dtype=np.uint32), reg_m)[0]
            for g0 in range(0, lanes.size, group):
                b = (s[g0: g0 + group].astype(np.int64)) % nslots
                worst = max(worst, int(np.bincount(b).max()))
        return worst

    for k in range(nrounds - 1):
        w = ways(recs[k][40: 40 + LT], recs[k][34: 34 + R], 32 if planar else 16, 32 if planar else 16)
        r_ = ways(recs[k + 1][24: 24 + LT], recs[k + 1][18: 18 + R], 32, 32)
        out.append((w, r_))
    return out


def run_measure_plan(mplan, state):
    """Execute a MeasurePlan on a flat state; returns <psi|P_t|psi> per term (complex128)."""
    eout = np.zeros(len(mplan.terms), dtype=np.complex128)
    st = np.array(state, dtype=np.complex128)
    for desc in mplan.descs:
        run_pass(st, desc, np.zeros(8), np.zeros(0), eout)
    return np.array([eout[i] * (1j) ** t.ny for i, t in enumerate(mplan.terms)])


# ---- adjoint sweep (tcmi_run_adjoint_pass / tcmi_build_adjoint_tables) -----------------------------
BK_UDAG, BK_KMAT = 3, 4


def build_adjoint_table(ginfo, cpool, params, ptab_size):
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    B = params.shape[0]
    ptab = np.zeros((B, ptab_size), dtype=np.float64)
    for rec in np.asarray(ginfo).reshape(-1, 8):
        kind, slot, pidx, dim, off = (int(x) for x in rec[:5])
        k, o = cpool[off], cpool[off + 1]
        if kind == BK_PHASE:
            e = _phase_entry(cpool, off, dim, int(rec[5]), params, invert=bool(rec[6]))
            ptab[:, slot] = e.real
            ptab[:, slot + 1] = e.imag
            continue
        for b in range(B):
            a = k * params[b, pidx] + o
            if kind == BK_COEF:
                ptab[b, slot] = a
                continue
            nn = dim * dim
            c = [(cpool[off + 2 + 2 * nn * i: off + 2 + 2 * nn * (i + 1)]).reshape(nn, 2) for i in range(3)]
            c = [(x[:, 0] + 1j * x[:, 1]).reshape(dim, dim) for x in c]
            u = c[0] + np.cos(a) * c[1] + np.sin(a) * c[2]
            du = k * (-np.sin(a) * c[1] + np.cos(a) * c[2])
            m = u.conj().T if kind == BK_UDAG else du @ u.conj().T
            if int(rec[5]) and kind == BK_UDAG:
                ptab[b, slot: slot + 8] = _shear_params(m.reshape(1, 4), int(rec[5]), two=bool(rec[6]))[0]
                continue
            ptab[b, slot: slot + 2 * nn: 2] = m.real.reshape(-1)
            ptab[b, slot + 1: slot + 2 * nn: 2] = m.imag.reshape(-1)
    return ptab


def run_adjoint_pass(psi, lam, desc, ctab, ptab_row, gout):
    """One backward pass on (psi, lam) in place; gradient slots accumulate into ``gout``."""
    d = np.asarray(desc).view(np.uint32).astype(np.int64)
    dsig = np.asarray(desc).astype(np.int64)  # signed view (slots may be -1)
    n, T, R, LT, nrounds = (int(x) for x in d[1:6])
    nwg, nth, NR = 2 ** (n - T), 2**LT, 2**R
    tile_bits = [int(x) for x in d[8: 8 + T]]
    wg = np.arange(nwg, dtype=np.uint64)
    for p in tile_bits:
        low = np.uint64((1 << p) - 1)
        wg = ((wg & ~low) << np.uint64(1)) | (wg & low)
    wg_base = wg.astype(np.uint32)
    tid = np.arange(nth, dtype=np.uint32)
    rid = np.arange(NR, dtype=np.uint32)

    def tab(slot, cnt):
        slot = int(slot)
        if slot & CONST_FLAG:
            return np.asarray(ctab)[(slot & ~CONST_FLAG): (slot & ~CONST_FLAG) + cnt]
        return np.asarray(ptab_row)[slot: slot + cnt]

    def cm(slot, dim):
        m = tab(slot, 2 * dim * dim)
        return (m[0::2] + 1j * m[1::2]).reshape(dim, dim)

    pc = HDR_WORDS
    regs = None  # [2, nwg, nth, NR]: psi, lambda
    lds = None
    pass_sign = 1.0
    for k in range(nrounds):
        rr = d[pc: pc + RR_WORDS]
        nops, opwords = int(rr[0]), int(rr[1])
        tphys = _xor_masks(tid, rr[8: 8 + LT])
        rphys = _xor_masks(rid, rr[2: 2 + R])
        gidx = (wg_base[:, None, None] | tphys[None, :, None] | rphys[None, None, :]).astype(np.int64)
        if k == 0:
            # FLAG_LAMBDA_ZERO (header word 6, bit 1): lambda is born in this pass, nothing is loaded
            regs = np.stack([psi[gidx], np.zeros_like(lam[gidx]) if (int(d[6]) & 2) else lam[gidx]])
        else:
            slot = (_xor_masks(tid, rr[24: 24 + LT])[:, None] ^ _xor_masks(rid, rr[18: 18 + R])[None, :]).astype(np.int64)
            regs = lds[:, :, slot]
        q = pc + RR_WORDS
        tidx = (wg_base[:, None] | tphys[None, :]).astype(np.uint64)
        for _ in range(nops):
            op = int(d[q])
            if op == OP_G1M:
                mk, ubase, kmask, kbase = int(d[q + 1]), int(d[q + 2]), int(d[q + 3]), int(d[q + 4])
                for j in range(R):
                    if not (mk >> j) & 1:
                        continue
                    shear = (mk >> (20 + j)) & 1
                    ud = None if shear else cm(ubase + 8 * j, 2)
                    r0 = rid[((rid >> j) & 1) == 0]
                    r1 = r0 | (1 << j)
                    if (kmask >> j) & 1:
                        kk = cm(kbase + 8 * j, 2)
                        a0, a1 = regs[0][..., r0], regs[0][..., r1]
                        t0, t1 = kk[0, 0] * a0 + kk[0, 1] * a1, kk[1, 0] * a0 + kk[1, 1] * a1
                        gout[int(dsig[q + 5 + j])] += np.sum(np.real(np.conj(regs[1][..., r0]) * t0 + np.conj(regs[1][..., r1]) * t1))
                    if shear:  # the pulled-out sign is common to psi and lambda: no gradient sees it, the store applies it
                        tb8 = tab(ubase + 8 * j, 8)
                        pass_sign *= tb8[2]
                        flv = (mk >> (8 + 2 * j)) & 3
                        if tb8[3] != 0:
                            # two-shear form: psi <- L(v) S(u) psi, pending factor diag(c, 1/c); lambda takes the other
                            # order, lambda <- S(b) L(g) lambda with pending diag(1/c, c): (g, b) = (-u, -v) for the
                            # real class, (u, v) for the rx-like class
                            _apply_shear(regs[0], r0, r1, tb8[0], tb8[1], flv, two=True)
                            f = 1.0 if flv == 1 else 1j
                            g_, b_ = (-tb8[0], -tb8[1]) if flv == 1 else (tb8[0], tb8[1])
                            x, y = regs[1][..., r0].copy(), regs[1][..., r1].copy()
                            y = y + g_ * f * x
                            x = x + b_ * f * y
                            regs[1][..., r0] = x
                            regs[1][..., r1] = y
                            continue
                        _apply_shear(regs, r0, r1, tb8[0], tb8[1], flv)
                        continue
                    x0, x1 = regs[..., r0].copy(), regs[..., r1].copy()
                    regs[..., r0] = ud[0, 0] * x0 + ud[0, 1] * x1
                    regs[..., r1] = ud[1, 0] * x0 + ud[1, 1] * x1
                q += 5 + R
            elif op == OP_DFOLD:
                # {13, nterms, gslot, (thread-side Z mask, register mask, cslot) * nterms}: lambda += D psi, D = sum_t c_t sign_t
                nt_, gs_ = int(d[q + 1]), int(dsig[q + 2])
                D = np.zeros((nwg, nth, NR))
                for t_ in range(nt_):
                    zm_, rm_, c_ = int(d[q + 3 + 3 * t_]), int(d[q + 4 + 3 * t_]), float(tab(d[q + 5 + 3 * t_], 1)[0])
                    st_ = 1 - 2 * _parity(tidx & np.uint64(zm_))
                    sr_ = 1 - 2 * _parity(rid & np.uint32(rm_)).astype(np.int64)
                    D += c_ * st_[:, :, None] * sr_[None, None, :]
                gout[gs_] += 0.5 * np.sum(D * np.abs(regs[0]) ** 2)
                regs[1] = regs[1] + D * regs[0]
                q += 3 + 3 * nt_
            elif op == OP_XFOLD:
                # {12, j, cslot, gslot}: lambda += c X_j psi on register bit j, energy slot += c sum_pairs Re(conj(psi_x) psi_y)
                j, kd_, c_, gs_ = int(d[q + 1]) & 0xFF, int(d[q + 1]) >> 8, float(tab(d[q + 2], 1)[0]), int(dsig[q + 3])
                r0 = rid[((rid >> j) & 1) == 0]
                r1 = r0 | (1 << j)
                a0, a1 = regs[0][..., r0].copy(), regs[0][..., r1].copy()
                f01, f10 = (1.0, 1.0) if kd_ == 0 else (-1j, 1j)        # P[0, 1], P[1, 0] of X / Y
                regs[1][..., r0] += c_ * f01 * a1
                regs[1][..., r1] += c_ * f10 * a0
                gout[gs_] += c_ * np.sum(np.real(np.conj(a0) * f01 * a1))
                q += 4
            elif op == OP_XFOLD2:
                # {14, ja | jb << 8 | ka << 16 | kb << 17, cslot, gslot}: lambda += c P_a P_b psi on the register bits ja, jb
                # (P = X / Y), energy slot += c / 2 Re <psi| P_a P_b |psi> of the tile
                w1 = int(d[q + 1])
                ja, jb, ka, kb = w1 & 0xFF, (w1 >> 8) & 0xFF, (w1 >> 16) & 1, (w1 >> 17) & 1
                c_, gs_ = float(tab(d[q + 2], 1)[0]), int(dsig[q + 3])
                pm = {0: np.array([[0, 1], [1, 0]], dtype=np.complex128), 1: np.array([[0, -1j], [1j, 0]], dtype=np.complex128)}
                a = regs[0].copy()
                out = np.zeros_like(a)
                for r_ in rid:
                    xa, xb = (int(r_) >> ja) & 1, (int(r_) >> jb) & 1
                    src = int(r_) ^ (1 << ja) ^ (1 << jb)
                    out[..., int(r_)] = pm[ka][xa, 1 - xa] * pm[kb][xb, 1 - xb] * a[..., src]
                regs[1] = regs[1] + c_ * out
                gout[gs_] += 0.5 * c_ * np.sum(np.real(np.conj(a) * out))
                q += 4
            elif op == OP_G2:
                ja, kind, jb = int(d[q + 1]) & 0xFF, int(d[q + 1]) >> 8, int(d[q + 2])
                uslot, kslot, gslot = int(dsig[q + 3]), int(dsig[q + 4]), int(dsig[q + 5])
                ud = cm(uslot, 4)
                if kind == 1:
                    ud = np.eye(4)[[0, 1, 3, 2]]
                elif kind == 2:
                    ud = np.eye(4)[[0, 3, 2, 1]]
                elif kind == 3:
                    ud = np.eye(4)[[0, 2, 1, 3]]
                base = rid[(((rid >> ja) & 1) == 0) & (((rid >> jb) & 1) == 0)]
                idx = [base | (xa << ja) | (xb << jb) for xa in (0, 1) for xb in (0, 1)]
                if kslot >= 0:
                    kk = cm(kslot, 4)
                    a = [regs[0][..., ix] for ix in idx]
                    acc = 0.0
                    for row in range(4):
                        t = sum(kk[row, col] * a[col] for col in range(4))
                        acc += np.sum(np.real(np.conj(regs[1][..., idx[row]]) * t))
                    gout[gslot] += acc
                x = [regs[..., ix].copy() for ix in idx]
                for row in range(4):
                    regs[..., idx[row]] = sum(ud[row, col] * x[col] for col in range(4))
                q += 6
            elif op == OP_DIAG:
                nA, nB, nC, base_slot = (int(x) for x in d[q + 1: q + 5])
                q += 5
                cf = np.asarray(ptab_row)[base_slot: base_slot + nA + nB + nC]
                mA, mB, jB, mC = d[q: q + nA], d[q + nA: q + nA + nB], d[q + nA + nB: q + nA + 2 * nB], d[q + nA + 2 * nB: q + nA + 2 * nB + nC]
                g0 = q + nA + 2 * nB + nC
                gs = dsig[g0: g0 + nA + nB + nC]
                q = g0 + nA + nB + nC
                w = np.imag(np.conj(regs[1]) * regs[0])  # [nwg, nth, NR]
                phi = np.zeros((nwg, nth, NR))
                for e in range(nA + nB + nC):
                    if e < nA:
                        sgn = (1 - 2 * _parity(tidx & np.uint64(int(mA[e]))))[:, :, None] * np.ones(NR)[None, None, :]
                    elif e < nA + nB:
                        z = 1 - 2 * ((rid >> int(jB[e - nA])) & 1).astype(np.int64)
                        sgn = (1 - 2 * _parity(tidx & np.uint64(int(mB[e - nA]))))[:, :, None] * z[None, None, :]
                    else:
                        z = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(int(mC[e - nA - nB])))
                        sgn = np.ones((nwg, nth))[:, :, None] * z[None, None, :]
                    phi += float(cf[e]) * sgn
                    if int(gs[e]) >= 0:
                        gout[int(gs[e])] += np.sum(sgn * w)
                regs = regs * np.exp(-2j * np.pi * phi)[None]
            elif op == OP_DIAGF:
                cslot, nC, nB, nA, nsel = int(dsig[q + 1]), int(d[q + 2]), int(d[q + 3]), int(d[q + 4]), int(d[q + 5])
                sel_masks = [int(d[q + 6 + k_]) for k_ in range(nsel)]
                q += 9
                w = np.imag(np.conj(regs[1]) * regs[0])  # [nwg, nth, NR]
                for rm in range(NR):      # gsC[rm]: gradient slot of the register-only term with mask rm
                    gs = int(dsig[q + rm])
                    if gs >= 0:
                        assert nC & 1
                        z = 1 - 2 * _parity(rid.astype(np.uint64) & np.uint64(rm))
                        gout[gs] += np.sum(w * z[None, None, :])
                q += NR
                if cslot >= 0:
                    v = np.zeros(tidx.shape, dtype=np.int64)
                    for k_, m_ in enumerate(sel_masks):
                        v += _parity(tidx & np.uint64(m_)).astype(np.int64) << k_
                    ntab = 2 if nC & 2 else 1       # bit 1: lambda's table (reciprocal real factors) follows psi's
                    tb = np.asarray(ptab_row)[cslot: cslot + 2 * NR * (1 << nsel) * ntab]
                    tbc = (tb[0::2] + 1j * tb[1::2]).reshape(ntab, 1 << nsel, NR)
                    regs = np.stack([regs[0] * np.conj(tbc[0][v]), regs[1] * np.conj(tbc[-1][v])])
                for e in range(nB):
                    j, mask, slot, gs = int(d[q]), int(d[q + 1]), int(dsig[q + 2]), int(dsig[q + 3])
                    q += 4
                    sgn = (1 - 2 * _parity(tidx & np.uint64(mask)))[:, :, None] * (
                        1 - 2 * ((rid >> j) & 1).astype(np.int64))[None, None, :]
                    if gs >= 0:
                        gout[gs] += np.sum(w * sgn)
                    if slot >= 0:
                        cs, sn = float(ptab_row[slot]), float(ptab_row[slot + 1])
                        regs = regs * (cs - 1j * sn * sgn)[None]
                for e in range(nA):
                    mask, gs = int(d[q]), int(dsig[q + 1])
                    q += 2
                    if gs >= 0:
                        gout[gs] += np.sum(w * (1 - 2 * _parity(tidx & np.uint64(mask)))[:, :, None])
            else:
                raise ValueError(f"bad backward opcode {op}")
        assert q == pc + RR_WORDS + opwords
        pc = q
        if k < nrounds - 1:
            slot = (_xor_masks(tid, rr[40: 40 + LT])[:, None] ^ _xor_masks(rid, rr[34: 34 + R])[None, :]).astype(np.int64)
            lds = np.zeros((2, nwg, nth * NR), dtype=regs.dtype)
            lds[:, :, slot] = regs
        else:
            psi[gidx] = regs[0] * pass_sign
            lam[gidx] = regs[1] * pass_sign
    assert pc == d.size


def run_adjoint_plan(aplan, params, psi, lam, nparams, return_lambda=False):
    """Returns (dL/dparams, psi_in): Re<lam | d psi / d params> by the adjoint sweep (float64)."""
    psi = np.array(psi, dtype=np.complex128)
    lam = np.array(lam, dtype=np.complex128)
    params = np.asarray(params, dtype=np.float64).reshape(-1) if nparams else np.zeros(1)
    ptab = build_adjoint_table(aplan.ginfo, aplan.cpool, params, max(1, aplan.ptab_size))[0]
    gout = np.zeros(len(aplan.gslot_param))
    for desc in aplan.descs:
        run_adjoint_pass(psi, lam, desc, aplan.ctab, ptab, gout)
    g = np.zeros(max(nparams, 1))
    np.add.at(g, aplan.gslot_param, gout * aplan.gslot_factor)
    if return_lambda:
        return g[:nparams], psi, lam
    return g[:nparams], psi
