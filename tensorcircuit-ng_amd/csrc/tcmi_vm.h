// Internal layout constants of the tile-VM pass descriptor (mirrored in tcmi/plan.py).
//
// desc[0]=MAGIC desc[1]=n desc[2]=T desc[3]=R desc[4]=LT desc[5]=nrounds desc[8..8+T)=tile bit
// positions (ascending physical bits).  Then per round a record of TCMI_RR_WORDS words
//   [0] nops  [1] op words  [2..8) reg phys masks  [8..18) thread phys masks
//   [18..24) reg LDS-read masks  [24..34) thread LDS-read masks   (exchange into this round)
//   [34..40) reg LDS-write masks [40..50) thread LDS-write masks  (exchange out of this round)
// followed by the round's ops:
//   G1M  {4, mask | kinds<<8, base}   1-qubit gates on every register bit j in mask; matrix j at
//                                     ptab[base + 8j ..]; 2 kind bits per j at bit 8+2j (0 general,
//                                     1 real, 2 real-diagonal/imaginary-off-diagonal)
//   G2   {2, ja | kind<<8, jb, slot}  2-qubit gate, matrix index = (bit ja << 1) | bit jb, ja < jb
//                                     (kind: 0 dense, 1 CNOT control ja, 2 CNOT control jb, 3 SWAP)
//   DIAG {3, nA, nB, nC, base, maskA[nA], maskB[nB], jB[nB], rmaskC[nC]}  phase polynomial in turns,
//        coefficients contiguous at ptab[base ..] in the order A, B, C (nA, nB multiples of 8):
//        A: coef * (-1)^parity(thread_index & mask)
//        B: coef * (-1)^parity(thread_index & mask) * z_j(r)
//        C: coef * (-1)^parity(r & rmask)
//   EXPECT {5, nZ, nX, Z[nZ] = {zr, zmask, out}, X[nX] = {xr, zr, zmask, out}}  Pauli-string partial sums
//        sum_r (-1)^parity(r & zr) (-1)^parity(thread_index & zmask) conj(a[r ^ xr]) a[r]  accumulated
//        (atomicAdd, double) into eout[2*out], eout[2*out+1]; xr = register-bit mask of the X/Y bits
//        (one or two bits), zr / zmask = sign bits (Z and Y) inside / outside the register bits.
// Backward (adjoint sweep) programs, executed by tcmi_run_adjoint_pass on (psi, lambda):
//   G1M  {4, mask | kinds<<8, ubase, kmask, kbase, gslot[R]}   U^dagger_j at ptab[ubase + 8j], K_j at
//        ptab[kbase + 8j] for the bits in kmask; gout[gslot[j]] += Re <lambda|K_j|psi>, then both
//        vectors are multiplied by U^dagger_j
//   G2   {2, ja | kind<<8, jb, uslot, kslot (-1: constant gate), gslot}
//   DIAG {3, nA, nB, nC, base, maskA, maskB, jB, rmaskC, gsA[nA], gsB[nB], gsC[nC]}  forward
//        coefficients at ptab[base..]; gout[gs] += sum_idx s_t(idx) Im(conj(lambda) psi) for gs >= 0,
//        then both vectors are multiplied by the conjugate phase
// desc[6] flags: bit 0 = do not store the tile (measurement pass).
// slot = offset (in reals) into the per-batch table, or into the constant table if TCMI_CONST_FLAG.
#ifndef TCMI_VM_H
#define TCMI_VM_H

#include "../../include/tcmi.h"

#define TCMI_MAGIC 0x54434D31
#define TCMI_HDR_WORDS 24
#define TCMI_RR_WORDS 50
#define TCMI_OP_G1 1
#define TCMI_OP_G2 2
#define TCMI_OP_DIAG 3
#define TCMI_OP_G1M 4
#define TCMI_DIAG_CHUNK 8
#define TCMI_OP_EXPECT 5
#define TCMI_OP_DIAGB 7 /* {7, j, thread mask, slot}: a[r] *= exp(+-i phi), sign = z_j(r) * parity(thread index & mask); table = {cos, sin} */
#define TCMI_OP_DIAGC 6 /* {6, slot}: a[r] *= table[r], 2^R complex factors in the per-batch table */
#define TCMI_OP_DIAGF 8 /* backward flush of diagonal terms in table form (tcmi_adjoint2.hip): {8, cslot (-1: none), hasC, nB, nA, nsel, m0, m1, m2, gsC[2^R] (gradient slot of the register-only term with mask k or -1; hasC = any), B: (j, mask, slot (-1: applied elsewhere), gslot)*, A: (mask, gslot)*}; cslot = 2^nsel wave-selected variants (as OP_DIAGCW) of the register table; tables = FORWARD phase factors, the conjugate is applied; A terms: gradient only */
#define TCMI_OP_DIAGB2 9 /* {9, j, mask1, mask2, slot}: a[r] *= table[s1 + 2 s2] (conjugate for z_j(r) = -1), s_i = parity(thread index & mask_i); 4 complex factors (second-generation kernels only) */
#define TCMI_OP_EXPECT2 11 /* measurement (second-generation kernel tcmi_measure2.hip): {11, nX, gmask, X strings (xr, zr, zm, out)*nX as in EXPECT, then for every set bit k of gmask (ascending): count, (zm, out)*count -- the Z-only strings whose register mask is k */
#define TCMI_OP_DIAGCW 10 /* {10, slot, nsel, m0, m1, m2}: a[r] *= table[v][r], v = sum_k parity(wave's thread index & m_k) << k (the masks touch wave-uniform bits only), 2^nsel tables of 2^R complex factors (second-generation kernels only) */
#define TCMI_FLAG_NOSTORE 1
#define TCMI_CONST_FLAG (1 << 30)
#define TCMI_BK_TRIG 1
#define TCMI_BK_COEF 2
#define TCMI_BK_UDAG 3  /* U^dagger of C0 + cos C1 + sin C2 */
#define TCMI_BK_KMAT 4  /* K = (dU/dtheta) U^dagger */
#define TCMI_BK_PHASE 6 /* exp(2 pi i sum_t +-(k_t theta_t + o_t)) for one register index: rec = {6, slot, 0, nterms, off, r}, cpool = {k, o, param index, register mask} per term */
#define TCMI_SCALE_TERM (1 << 16) /* BK_PHASE term flag (register-mask field): real factor c^(+-1), c = cos(k theta + o) in radians: what a rotation applied in two-shear form leaves behind (plan.shear2_gates); rec[6] = 1: reciprocal (lambda tables of the adjoint sweep) */
#define TCMI_SHEAR2_CMIN 0.5      /* two-shear form only while |cos| >= this, else the three-shear form and factor 1 */
#define TCMI_SMALL_DESC_WORDS 64 /* words of one job of tcmi_tensordot_small_batch (tcmi_tensordot.hip) */
#define TCMI_BK_SELECT 5 /* M = table[round(theta)]: cpool = {count, 0, matrices...} */

#endif
