#!/usr/bin/env python3
"""Generates tcmi_vm2_asm.inc: the hand-scheduled gfx950 instruction bodies of the complex64 tile-VM.

Why assembly bodies (measured on MI355X, scripts/ubench/gen_operand_forms.py, profiles/r02a_*):
  * a 32-bit VALU instruction with an SGPR (or constant) operand issues every 4 cycles per SIMD, the same
    instruction with VGPR operands only every 2; `v_pk_*_f32` always takes 4 cycles for its two lanes'
    worth of work.  Gate coefficients are wave-uniform (scalar loads), so the amplitude arithmetic is
    written with packed f32 instructions on (re, im) register pairs: full FP32 rate with SGPR matrices.
  * hipcc copies the whole amplitude register array at every multi-way control-flow merge that follows a
    modification of it (64 `v_mov` per gate in the round-1 kernel).  Here every body is one asm statement whose
    amplitude operands are tied ("+v"), and the callers select bodies with chains of independent skip tests
    (tcmi_vm2.hip): two-way merges of in-place updates, which the register coalescer resolves without copies.

Packed-operand conventions used below (V_PK_{MUL,FMA}_F32, CDNA3/4 ISA): result.lo uses source halves chosen
by op_sel, result.hi by op_sel_hi; neg_lo / neg_hi negate a source for the lo / hi result.  With an amplitude
A = (re, im) and a coefficient pair P:
    c*A + C        v_pk_fma_f32 D, A, P, C op_sel_hi:[1,0,1]                              (c = P.lo)
    s*(iA) + C     v_pk_fma_f32 D, A, P, C op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]  (s = P.hi, iA = (-im, re))

    python3 gen_vm2_asm.py > tcmi_vm2_asm.inc
"""


def mul_re(D, A, P, hi=False):
    """D = P.x * A   (x = hi / lo half of P)"""
    h = 1 if hi else 0
    return f"v_pk_mul_f32 {D}, {A}, {P} op_sel:[0,{h}] op_sel_hi:[1,{h}]"


def mul_im(D, A, P, hi=True, conj=False):
    """D = P.x * (iA)   (conj: P.x * (-iA))"""
    h = 1 if hi else 0
    neg = "neg_hi:[1,0]" if conj else "neg_lo:[1,0]"
    return f"v_pk_mul_f32 {D}, {A}, {P} op_sel:[1,{h}] op_sel_hi:[0,{h}] {neg}"


def fma_re(D, A, P, C, hi=False):
    """D = P.x * A + C"""
    h = 1 if hi else 0
    return f"v_pk_fma_f32 {D}, {A}, {P}, {C} op_sel:[0,{h},0] op_sel_hi:[1,{h},1]"


def fma_im(D, A, P, C, hi=True, conj=False):
    """D = P.x * (iA) + C"""
    h = 1 if hi else 0
    neg = "neg_hi:[1,0,0]" if conj else "neg_lo:[1,0,0]"
    return f"v_pk_fma_f32 {D}, {A}, {P}, {C} op_sel:[1,{h},0] op_sel_hi:[0,{h},1] {neg}"


# ---- one-qubit gate on an amplitude pair (X, Y); P0..P3 = (m00, m01, m10, m11) as (re, im) pairs ------------
def g1_rx(X, Y, T, U, P):
    """real diagonal, imaginary off-diagonal (rx, y ...): x' = m00r x + m01i (iy), y' = m10i (ix) + m11r y"""
    return [mul_im(T, Y, P[1]), mul_re(Y, Y, P[3]), fma_im(Y, X, P[2], Y), fma_re(X, X, P[0], T)]


def g1_real(X, Y, T, U, P):
    """real matrix (h, ry, x, z ...)"""
    return [mul_re(T, Y, P[1]), mul_re(Y, Y, P[3]), fma_re(Y, X, P[2], Y), fma_re(X, X, P[0], T)]


def g1_gen(X, Y, T, U, P):
    """general complex 2x2: x' in T, y' in place through U, then X <- T"""
    return [
        mul_re(T, X, P[0]), mul_im(U, Y, P[3]),
        fma_im(T, X, P[0], T), fma_re(U, Y, P[3], U),
        fma_re(T, Y, P[1], T), fma_re(U, X, P[2], U),
        fma_im(T, Y, P[1], T), fma_im(Y, X, P[2], U),
        f"v_mov_b64 {X}, {T}",
    ]


def interleave(seqs):
    """Round-robin merge of instruction sequences (independent chains back to back)."""
    out = []
    for i in range(max(len(s) for s in seqs)):
        for s in seqs:
            if i < len(s):
                out.append(s[i])
    return out


def gen_gate8(name, fn, ntmp):
    """Eight amplitude pairs through one 2x2 matrix of a fixed structure class (no branches: the class is chosen
    by the caller with flag tests the optimiser cannot fuse into a switch)."""
    NP = 8
    P = [f"%[p{k}]" for k in range(4)]
    lines = []
    for g in range(0, NP, 2):
        lines += interleave([fn(f"%[x{p}]", f"%[y{p}]", f"%[t{p % 2}]", f"%[u{p % 2}]", P) for p in (g, g + 1)])
    args = ", ".join(f"v2f& x{p}, v2f& y{p}" for p in range(NP)) + ", v2f p0, v2f p1, v2f p2, v2f p3"
    outs = []
    for p in range(NP):
        outs += [f'[x{p}] "+v"(x{p})', f'[y{p}] "+v"(y{p})']
    tmps = ["t0", "t1"] + (["u0", "u1"] if ntmp == 4 else [])
    outs += [f'[{t}] "=&v"({t})' for t in tmps]
    ins = [f'[p{k}] "s"(p{k})' for k in range(4)]
    emit_named(name, args, lines, outs, ins, tmps, clobbers=())


def emit_named(name, args, lines, outs, ins, tmps, clobbers=("scc",)):
    body = "\n".join(f'      "{l}\\n\\t"' for l in lines)
    c = ", ".join(f'"{x}"' for x in clobbers)
    print(f"__device__ __forceinline__ void {name}({args}) {{")
    if tmps:
        print("  v2f " + ", ".join(tmps) + ";")
    print("  asm volatile(")
    print(body)
    print("      : " + ", ".join(outs))
    print("      : " + ", ".join(ins))
    print(f"      : {c});")
    print("}\n")


def gen_shear8(name, flavor, steps=3):
    """Eight amplitude pairs through a rotation in three-shear form, p = (u, v): x += u y', y' += v x, x += u y'
    with y' = y (flavor "real") or i y (flavor "rx": y += v (i x)); 3 packed instructions per pair, no temporaries.
    steps = 2: the two-shear form (no third step; the real factor diag(c, 1/c) is a pending scale term of the plan);
    steps = "third": the third step alone (the forward kernel runs the three-shear form as the two-shear body + this:
    fewer instructions of kernel text -- past a certain size of the op loop hipcc drains every tile load before
    the loop instead of waiting at first use, 6 % of the pass time); steps = "lambda": see below."""
    f = fma_re if flavor == "real" else (lambda D, A, P, C, hi=False: fma_im(D, A, P, C, hi=hi))
    seqs = []
    for p_ in range(8):
        X, Y = f"%[x{p_}]", f"%[y{p_}]"
        if steps == "lambda":
            # the cotangent's side of a two-shear U^dagger (adjoint sweep): lower shear first, lambda carries the
            # reciprocal factor diag(1/c, c).  y' += g x, x += b y' with (g, b) = (-u, -v) [real] or (u, v) [rx-like]
            neg = " neg_lo:[0,1,0] neg_hi:[0,1,0]" if flavor == "real" else ""
            seqs.append([f(Y, X, "%[p]", Y, hi=False) + neg, f(X, Y, "%[p]", X, hi=True) + neg])
            continue
        full = [f(X, Y, "%[p]", X, hi=False), f(Y, X, "%[p]", Y, hi=True), f(X, Y, "%[p]", X, hi=False)]
        seqs.append(full[2:] if steps == "third" else full[:steps])
    lines = interleave(seqs[:4]) + interleave(seqs[4:])
    args = ", ".join(f"v2f& x{p_}, v2f& y{p_}" for p_ in range(8)) + ", v2f p"
    outs = []
    for p_ in range(8):
        outs += [f'[x{p_}] "+v"(x{p_})', f'[y{p_}] "+v"(y{p_})']
    emit_named(name, args, lines, outs, ['[p] "s"(p)'], [], clobbers=())


def gen_scale8():
    """a_k *= p.x (real scalar in an SGPR pair): the sign pulled out of the shear-form gates of a pass."""
    lines = [mul_re(f"%[a{k}]", f"%[a{k}]", "%[p]") for k in range(8)]
    args = ", ".join(f"v2f& a{k}" for k in range(8)) + ", v2f p"
    outs = [f'[a{k}] "+v"(a{k})' for k in range(8)]
    emit_named("vm2_scale8", args, lines, outs, ['[p] "s"(p)'], [], clobbers=())


def gen_cmul8s():
    """a_k *= (c_k + i s_k), coefficient pairs in SGPRs (DIAGC: one table entry per register index)."""
    seqs = []
    for k in range(8):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[a{k}]", f"%[p{k}]"), fma_re(f"%[a{k}]", f"%[a{k}]", f"%[p{k}]", T)])
    lines = []
    for g in range(0, 8, 2):
        lines += interleave(seqs[g:g + 2])
    args = ", ".join(f"v2f& a{k}" for k in range(8)) + ", " + ", ".join(f"v2f p{k}" for k in range(8))
    outs = [f'[a{k}] "+v"(a{k})' for k in range(8)] + ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    ins = [f'[p{k}] "s"(p{k})' for k in range(8)]
    emit_named("vm2_cmul8s", args, lines, outs, ins, ["t0", "t1"], clobbers=())


def gen_cmul8s_conj():
    """a_k *= (c_k - i s_k), coefficient pairs in SGPRs (inverse of a DIAGC table, adjoint sweep)."""
    seqs = []
    for k in range(8):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[a{k}]", f"%[p{k}]", conj=True), fma_re(f"%[a{k}]", f"%[a{k}]", f"%[p{k}]", T)])
    lines = []
    for g in range(0, 8, 2):
        lines += interleave(seqs[g:g + 2])
    args = ", ".join(f"v2f& a{k}" for k in range(8)) + ", " + ", ".join(f"v2f p{k}" for k in range(8))
    outs = [f'[a{k}] "+v"(a{k})' for k in range(8)] + ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    ins = [f'[p{k}] "s"(p{k})' for k in range(8)]
    emit_named("vm2_cmul8s_conj", args, lines, outs, ins, ["t0", "t1"], clobbers=())


def gen_cmul8v():
    """a_k *= (c_k + i s_k), per-thread coefficient pairs in VGPRs (general DIAG op)."""
    seqs = []
    for k in range(8):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[a{k}]", f"%[e{k}]"), fma_re(f"%[a{k}]", f"%[a{k}]", f"%[e{k}]", T)])
    lines = []
    for g in range(0, 8, 2):
        lines += interleave(seqs[g:g + 2])
    args = ", ".join(f"v2f& a{k}" for k in range(8)) + ", " + ", ".join(f"v2f e{k}" for k in range(8))
    outs = [f'[a{k}] "+v"(a{k})' for k in range(8)] + ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    ins = [f'[e{k}] "v"(e{k})' for k in range(8)]
    emit_named("vm2_cmul8v", args, lines, outs, ins, ["t0", "t1"], clobbers=())


def gen_cmul44v():
    """a_k *= (c + i s) for the four a's, b_k *= (c - i s) for the four b's; (c, s) per thread (DIAGB)."""
    seqs = []
    for k in range(4):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[a{k}]", "%[e]"), fma_re(f"%[a{k}]", f"%[a{k}]", "%[e]", T)])
    for k in range(4):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[b{k}]", "%[e]", conj=True), fma_re(f"%[b{k}]", f"%[b{k}]", "%[e]", T)])
    lines = []
    for g in range(0, 8, 2):
        lines += interleave(seqs[g:g + 2])
    args = ", ".join(f"v2f& a{k}" for k in range(4)) + ", " + ", ".join(f"v2f& b{k}" for k in range(4)) + ", v2f e"
    outs = [f'[a{k}] "+v"(a{k})' for k in range(4)] + [f'[b{k}] "+v"(b{k})' for k in range(4)]
    outs += ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    emit_named("vm2_cmul44v", args, lines, outs, ['[e] "v"(e)'], ["t0", "t1"], clobbers=())


def gen_g2():
    """Dense two-qubit gate on two amplitude quads (index = (bit ja << 1) | bit jb): out_i = sum_j M[i][j] v_j,
    16 coefficient pairs in SGPRs, 32 packed instructions + 4 moves per quad."""
    lines = []
    for qd in ("a", "b"):
        rows = []
        for i in range(4):
            T = f"%[t{i}]"
            seq = [mul_re(T, f"%[{qd}0]", f"%[m{4 * i}]"), fma_im(T, f"%[{qd}0]", f"%[m{4 * i}]", T)]
            for j in range(1, 4):
                seq += [fma_re(T, f"%[{qd}{j}]", f"%[m{4 * i + j}]", T), fma_im(T, f"%[{qd}{j}]", f"%[m{4 * i + j}]", T)]
            rows.append(seq)
        lines += interleave(rows)
        lines += [f"v_mov_b64 %[{qd}{i}], %[t{i}]" for i in range(4)]
    args = (", ".join(f"v2f& a{i}" for i in range(4)) + ", " + ", ".join(f"v2f& b{i}" for i in range(4)) + ", "
            + ", ".join(f"v2f m{k}" for k in range(16)))
    outs = [f'[a{i}] "+v"(a{i})' for i in range(4)] + [f'[b{i}] "+v"(b{i})' for i in range(4)]
    outs += [f'[t{i}] "=&v"(t{i})' for i in range(4)]
    ins = [f'[m{k}] "s"(m{k})' for k in range(16)]
    emit_named("vm2_g2x2", args, lines, outs, ins, ["t0", "t1", "t2", "t3"], clobbers=())


def gen_swap4():
    """Four register swaps a_k <-> b_k (CNOT / SWAP on register bits)."""
    lines = []
    for k in range(4):
        lines += [f"v_mov_b64 %[t{k % 2}], %[a{k}]", f"v_mov_b64 %[a{k}], %[b{k}]", f"v_mov_b64 %[b{k}], %[t{k % 2}]"]
    args = ", ".join(f"v2f& a{k}, v2f& b{k}" for k in range(4))
    outs = []
    for k in range(4):
        outs += [f'[a{k}] "+v"(a{k})', f'[b{k}] "+v"(b{k})']
    outs += ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    emit_named("vm2_swap4", args, lines, outs, [], ["t0", "t1"], clobbers=())


# ---- adjoint sweep (tcmi_adjoint2.hip): gradient inner products and conjugate phases ----------------------------
def gen_grad4_init(name, kind):
    """As gen_grad4, but the accumulators START here (first pair by v_pk_mul instead of a zeroed register + fma): the
    plan-specialised sweep opens every gradient sum with this body and continues with the accumulating one."""
    lines = []
    for p in range(4):
        if kind == "rx":
            m3, m2 = "op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]"
        else:
            m3, m2 = "op_sel:[0,0,0] op_sel_hi:[1,1,1]", "op_sel:[0,0] op_sel_hi:[1,1]"
        if p == 0:
            lines.append(f"v_pk_mul_f32 %[c0], %[lx{p}], %[ay{p}] {m2}")
            lines.append(f"v_pk_mul_f32 %[c1], %[ly{p}], %[ax{p}] {m2}")
        else:
            lines.append(f"v_pk_fma_f32 %[c0], %[lx{p}], %[ay{p}], %[c0] {m3}")
            lines.append(f"v_pk_fma_f32 %[c1], %[ly{p}], %[ax{p}], %[c1] {m3}")
    args = (", ".join(f"v2f ax{p}, v2f ay{p}" for p in range(4)) + ", " + ", ".join(f"v2f lx{p}, v2f ly{p}" for p in range(4))
            + ", v2f& c0, v2f& c1")
    outs = ['[c0] "=&v"(c0)', '[c1] "=&v"(c1)']
    ins = []
    for p in range(4):
        ins += [f'[ax{p}] "v"(ax{p})', f'[ay{p}] "v"(ay{p})', f'[lx{p}] "v"(lx{p})', f'[ly{p}] "v"(ly{p})']
    emit_named(name, args, lines, outs, ins, [], clobbers=())


def gen_grad4(name, kind):
    """Accumulate the gradient inner product of one gate over four amplitude pairs (x = index with the gate's bit
    clear, y = set) of psi (ax, ay) and lambda (lx, ly); nothing is modified but the two accumulators.
      rx-like  K = i kappa X:  acc0 += (lx.re ay.im, -lx.im ay.re), acc1 += (ly.re ax.im, -ly.im ax.re)
      real     K = [[0, k01], [k10, 0]]:  acc0 += lx * ay (per component), acc1 += ly * ax
    The caller combines the halves and scales by the generator coefficient."""
    lines = []
    for p in range(4):
        if kind == "rx":
            mods = "op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
        else:
            mods = "op_sel:[0,0,0] op_sel_hi:[1,1,1]"
        lines.append(f"v_pk_fma_f32 %[c0], %[lx{p}], %[ay{p}], %[c0] {mods}")
        lines.append(f"v_pk_fma_f32 %[c1], %[ly{p}], %[ax{p}], %[c1] {mods}")
    args = (", ".join(f"v2f ax{p}, v2f ay{p}" for p in range(4)) + ", " + ", ".join(f"v2f lx{p}, v2f ly{p}" for p in range(4))
            + ", v2f& c0, v2f& c1")
    outs = ['[c0] "+v"(c0)', '[c1] "+v"(c1)']
    ins = []
    for p in range(4):
        ins += [f'[ax{p}] "v"(ax{p})', f'[ay{p}] "v"(ay{p})', f'[lx{p}] "v"(lx{p})', f'[ly{p}] "v"(ly{p})']
    emit_named(name, args, lines, outs, ins, [], clobbers=())


def gen_grad4_gen():
    """General generator K (complex 2x2 in SGPR pairs k0..k3): acc0 += lx * (K (ax, ay))_0, acc1 += ly * (K (ax, ay))_1
    componentwise (the real part of conj(lambda) K psi is the sum of the halves)."""
    lines = []
    K = [f"%[k{i}]" for i in range(4)]
    for p in range(0, 4, 2):
        seqs = []
        for w in (0, 1):
            X, Y, T, U = f"%[ax{p + w}]", f"%[ay{p + w}]", f"%[t{w}]", f"%[u{w}]"
            seqs.append([
                mul_re(T, X, K[0]), mul_re(U, X, K[2]),
                fma_im(T, X, K[0], T), fma_im(U, X, K[2], U),
                fma_re(T, Y, K[1], T), fma_re(U, Y, K[3], U),
                fma_im(T, Y, K[1], T), fma_im(U, Y, K[3], U),
                f"v_pk_fma_f32 %[c0], %[lx{p + w}], {T}, %[c0]",
                f"v_pk_fma_f32 %[c1], %[ly{p + w}], {U}, %[c1]",
            ])
        lines += interleave(seqs)
    args = (", ".join(f"v2f ax{p}, v2f ay{p}" for p in range(4)) + ", " + ", ".join(f"v2f lx{p}, v2f ly{p}" for p in range(4))
            + ", v2f k0, v2f k1, v2f k2, v2f k3, v2f& c0, v2f& c1")
    outs = ['[c0] "+v"(c0)', '[c1] "+v"(c1)'] + [f'[{t}] "=&v"({t})' for t in ("t0", "t1", "u0", "u1")]
    ins = []
    for p in range(4):
        ins += [f'[ax{p}] "v"(ax{p})', f'[ay{p}] "v"(ay{p})', f'[lx{p}] "v"(lx{p})', f'[ly{p}] "v"(ly{p})']
    ins += [f'[k{i}] "s"(k{i})' for i in range(4)]
    emit_named("vm2_grad4_gen", args, lines, outs, ins, ["t0", "t1", "u0", "u1"], clobbers=())


def gen_cross8():
    """t_k = (l_k.re a_k.im, l_k.im a_k.re): Im(conj(l) a) = t.x - t.y (diagonal-term gradients)."""
    lines = [f"v_pk_mul_f32 %[t{k}], %[l{k}], %[a{k}] op_sel:[0,1] op_sel_hi:[1,0]" for k in range(8)]
    args = ", ".join(f"v2f a{k}" for k in range(8)) + ", " + ", ".join(f"v2f l{k}" for k in range(8)) + ", " + ", ".join(f"v2f& t{k}" for k in range(8))
    outs = [f'[t{k}] "=v"(t{k})' for k in range(8)]
    ins = [f'[a{k}] "v"(a{k})' for k in range(8)] + [f'[l{k}] "v"(l{k})' for k in range(8)]
    emit_named("vm2_cross8", args, lines, outs, ins, [], clobbers=())


def gen_cmul8v_conj():
    """a_k *= (c_k - i s_k), per-thread coefficient pairs in VGPRs (inverse phases of the adjoint sweep)."""
    seqs = []
    for k in range(8):
        T = f"%[t{k % 2}]"
        seqs.append([mul_im(T, f"%[a{k}]", f"%[e{k}]", conj=True), fma_re(f"%[a{k}]", f"%[a{k}]", f"%[e{k}]", T)])
    lines = []
    for g in range(0, 8, 2):
        lines += interleave(seqs[g:g + 2])
    args = ", ".join(f"v2f& a{k}" for k in range(8)) + ", " + ", ".join(f"v2f e{k}" for k in range(8))
    outs = [f'[a{k}] "+v"(a{k})' for k in range(8)] + ['[t0] "=&v"(t0)', '[t1] "=&v"(t1)']
    ins = [f'[e{k}] "v"(e{k})' for k in range(8)]
    emit_named("vm2_cmul8v_conj", args, lines, outs, ins, ["t0", "t1"], clobbers=())


# ---- bodies with an internal wave-uniform skip (plan-specialised kernels, tcmi/specialize.py) -------------------
# A C-level `if` around an asm body splits the kernel into basic blocks, and hipcc's register allocator then spills
# scalar operands at the block boundaries (240 v_writelane / v_readlane pairs in a 9-round pass).  With the skip inside
# the asm statement the whole pass stays ONE basic block.
def gen_shear23(name="vm2_shear23_8_rx"):
    """Eight amplitude pairs through an rx-like rotation whose form the table builder chose per batch element:
    two shears always, the third unless bit 30 of `f` is set (flag word 2.0f = two-shear form, plan.shear2_gates)."""
    fm = lambda D, A, P, C, hi=False: fma_im(D, A, P, C, hi=hi)  # noqa: E731
    first, third = [], []
    for p_ in range(8):
        X, Y = f"%[x{p_}]", f"%[y{p_}]"
        first.append([fm(X, Y, "%[p]", X, hi=False), fm(Y, X, "%[p]", Y, hi=True)])
        third.append([fm(X, Y, "%[p]", X, hi=False)])
    lines = ["s_bitcmp1_b32 %[f], 30"] + interleave(first[:4]) + interleave(first[4:]) + ["s_cbranch_scc1 1f"]
    lines += interleave(third) + ["1:"]
    args = ", ".join(f"v2f& x{p_}, v2f& y{p_}" for p_ in range(8)) + ", v2f p, uint32_t f"
    outs = []
    for p_ in range(8):
        outs += [f'[x{p_}] "+v"(x{p_})', f'[y{p_}] "+v"(y{p_})']
    emit_named(name, args, lines, outs, ['[p] "s"(p)', '[f] "s"(f)'], [], clobbers=("scc",))


def gen_shear23l(name="vm2_shear23l_8_rx"):
    """The cotangent's side of gen_shear23 (reverse sweep): the three-shear form of U^dagger unless bit 30 of `f` is set,
    else the two shears in the other order (lower first; lambda carries the reciprocal factor diag(1/c, c), see
    gen_shear8 steps="lambda")."""
    fm = lambda D, A, P, C, hi=False: fma_im(D, A, P, C, hi=hi)  # noqa: E731
    three, two = [], []
    for p_ in range(8):
        X, Y = f"%[x{p_}]", f"%[y{p_}]"
        three.append([fm(X, Y, "%[p]", X, hi=False), fm(Y, X, "%[p]", Y, hi=True), fm(X, Y, "%[p]", X, hi=False)])
        two.append([fm(Y, X, "%[p]", Y, hi=False), fm(X, Y, "%[p]", X, hi=True)])
    lines = ["s_bitcmp1_b32 %[f], 30", "s_cbranch_scc1 2f"] + interleave(three[:4]) + interleave(three[4:])
    lines += ["s_branch 3f", "2:"] + interleave(two[:4]) + interleave(two[4:]) + ["3:"]
    args = ", ".join(f"v2f& x{p_}, v2f& y{p_}" for p_ in range(8)) + ", v2f p, uint32_t f"
    outs = []
    for p_ in range(8):
        outs += [f'[x{p_}] "+v"(x{p_})', f'[y{p_}] "+v"(y{p_})']
    emit_named(name, args, lines, outs, ['[p] "s"(p)', '[f] "s"(f)'], [], clobbers=("scc",))


def gen_negate16_if():
    """a_k = -a_k for sixteen amplitudes when `f` is non-zero (the sign pulled out of the shear-form gates of a pass)."""
    lines = ["s_cmp_eq_u32 %[f], 0", "s_cbranch_scc1 1f"]
    lines += [f"v_pk_mul_f32 %[a{k}], %[a{k}], -1.0 op_sel_hi:[1,0]" for k in range(16)]
    lines += ["1:"]
    args = ", ".join(f"v2f& a{k}" for k in range(16)) + ", uint32_t f"
    outs = [f'[a{k}] "+v"(a{k})' for k in range(16)]
    emit_named("vm2_negate16_if", args, lines, outs, ['[f] "s"(f)'], [], clobbers=("scc",))


if __name__ == "__main__":
    print("// GENERATED by gen_vm2_asm.py -- do not edit; see that file for the conventions.")
    print("#ifndef TCMI_VM2_ASM_INC\n#define TCMI_VM2_ASM_INC\n")
    gen_gate8("vm2_gate8_rx", g1_rx, 2)
    gen_gate8("vm2_gate8_real", g1_real, 2)
    gen_gate8("vm2_gate8_gen", g1_gen, 4)
    gen_shear8("vm2_shear8_real", "real")
    gen_shear8("vm2_shear8_rx", "rx")
    gen_shear8("vm2_shear2_8_rx", "rx", steps=2)
    gen_shear8("vm2_shear3rd_8_rx", "rx", steps="third")
    gen_shear8("vm2_shear2l_8_rx", "rx", steps="lambda")
    gen_scale8()
    gen_cmul8s()
    gen_cmul8v()
    gen_cmul44v()
    gen_g2()
    gen_swap4()
    gen_grad4("vm2_grad4_rx", "rx")
    gen_grad4("vm2_grad4_real", "real")
    gen_grad4_gen()
    gen_cross8()
    gen_cmul8v_conj()
    gen_cmul8s_conj()
    gen_shear23()
    gen_shear23l()
    gen_negate16_if()
    gen_grad4_init("vm2_grad4i_rx", "rx")
    gen_grad4_init("vm2_grad4i_real", "real")
    print("#endif")
