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
OP_G1, OP_G2, OP_DIAG, OP_G1M, OP_EXPECT = 1, 2, 3, 4, 5
R_MAX = 6
CONST_FLAG = 1 << 30
BK_TRIG, BK_COEF = 1, 2


def build_table(ginfo, cpool, params, ptab_size):
    """The builder kernel (``tcmi_build_tables``): params [B, P] -> ptab [B, ptab_size] float64."""
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    B = params.shape[0]
    ptab = np.zeros((B, ptab_size), dtype=np.float64)
    for rec in np.asarray(ginfo).reshape(-1, 8):
        kind, slot, pidx, dim, off = (int(x) for x in rec[:5])
        k, o = cpool[off], cpool[off + 1]
        th = params[:, pidx]
        if kind == BK_TRIG:
            a = k * th + o
            nn = dim * dim
            c = [
                (cpool[off + 2 + 2 * nn * i: off + 2 + 2 * nn * (i + 1)]).reshape(nn, 2)
                for i in range(3)
            ]
            c = [x[:, 0] + 1j * x[:, 1] for x in c]
            m = c[0][None, :] + np.cos(a)[:, None] * c[1][None, :] + np.sin(a)[:, None] * c[2][None, :]
            ptab[:, slot: slot + 2 * nn: 2] = m.real
            ptab[:, slot + 1: slot + 2 * nn: 2] = m.imag
        elif kind == BK_COEF:
            ptab[:, slot] = k * th + o
        else:
            raise ValueError(kind)
    return ptab


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
            state[gidx] = regs
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


def lds_conflicts(desc, elem_bytes=8):
    """Worst-case LDS bank multiplicity of every exchange in a pass: (write_ways, read_ways) per
    exchange, for 8-byte elements (ds_write_b64: 16-lane groups on 32 banks; ds_read_b64: 32-lane
    groups on 64 banks)."""
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
        w = ways(recs[k][40: 40 + LT], recs[k][34: 34 + R], 16, 16)
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
