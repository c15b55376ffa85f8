// tcmi tile-VM, complex64 gate passes, second generation (gfx950 / MI355X only).
//
// Same pass descriptors and tables as tcmi_vm.hip (layout: tcmi_vm.h, written by tcmi/plan.py, emulated on the
// CPU by oracle/plan_emulator.py), i.e. the same stand-in for the reference's tn.contract_between ->
// backend.tensordot chain (tensorcircuit/cons.py:937-960).  What changed is how the amplitude arithmetic is
// issued, after measuring the VALU on MI355X (scripts/ubench/gen_operand_forms.py, profiles/r02a_*):
//
//   * a 32-bit VALU instruction that reads an SGPR issues every 4 cycles, v_pk_{mul,fma}_f32 also every 4 but
//     does two lanes' worth of work -> amplitudes are (re, im) register PAIRS and every gate / phase is a short
//     sequence of packed instructions with the wave-uniform coefficients still in SGPRs (tcmi_vm2_asm.inc);
//   * hipcc copies the whole amplitude array at every control-flow merge behind a modification of it (the
//     round-1 kernel issued more v_mov than gate arithmetic).  Here the op loop body is straight-line for the
//     compiler: the gate-kind dispatch is a scalar branch inside the asm statements (tied operands), and the
//     per-op-type sections are skipped with independent one-hot flags the optimiser cannot merge into a switch.
//
// The kernel is launched by tcmi_run_pass (tcmi_vm.hip) for complex64 gate passes with R >= 4 (n >= 12).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tcmi_vm.h"
#include "tcmi_dev.h"

namespace tcmi {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v2f TCMI_K* KV2;

#include "tcmi_vm2_asm.inc"

// k-th register index with bit J clear / with bits JA < JB clear
__host__ __device__ constexpr int ins0(int k, int J) { return ((k >> J) << (J + 1)) | (k & ((1 << J) - 1)); }
__host__ __device__ constexpr int ins00(int k, int JA, int JB) { return ins0(ins0(k, JA), JB); }

// 1 << v, opaque to the optimiser: the sections of the op loop are guarded by independent-looking flag tests,
// which keeps them a chain of skips instead of a switch (whose merge would copy the amplitude registers)
__device__ __forceinline__ int onehot(int v) {
  int f;
  asm("s_lshl_b32 %0, 1, %1" : "=s"(f) : "s"(v) : "scc");
  return f;
}
// scheduling-region boundary: keeps the compiler from hoisting all 2^R slot addresses of an exchange phase ahead
// of the accesses (32 extra live registers, which pushed the kernel over the 128-VGPR / 4-waves-per-SIMD budget)
#define TCMI_SCHED_FENCE() asm volatile("" ::: "memory")

// a wave-uniform value in a VGPR (keeps later VALU uses free of SGPR operands: 2 instead of 4 issue cycles)
__device__ __forceinline__ uint32_t to_vgpr(uint32_t v) {
  uint32_t r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}
__device__ __forceinline__ uint32_t to_vgpr_v(uint32_t v) {  // not CSE-able: bounds the live range to its phase
  uint32_t r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}
// on ? 1 << v : 0   (on = 0 / 1)
__device__ __forceinline__ int onehot_if(int on, int v) {
  int f;
  asm("s_lshl_b32 %0, %1, %2" : "=s"(f) : "s"(on), "s"(v) : "scc");
  return f;
}

// one-qubit gate on register bit J; kf = one-hot structure class (1 general, 2 real, 4 rx-like, 0 = no gate)
template <int NR, int J>
__device__ __forceinline__ void vm2_g1(v2f (&a)[NR], int kf, v2f p0, v2f p1, v2f p2, v2f p3) {
  constexpr int B = 1 << J;
#define TCMI_G8(FN)                                                                                               \
  _Pragma("unroll") for (int g = 0; g < NR / 2; g += 8) {                                                         \
    const int r0 = ins0(g, J), r1 = ins0(g + 1, J), r2 = ins0(g + 2, J), r3 = ins0(g + 3, J);                     \
    const int r4 = ins0(g + 4, J), r5 = ins0(g + 5, J), r6 = ins0(g + 6, J), r7 = ins0(g + 7, J);                 \
    FN(a[r0], a[r0 | B], a[r1], a[r1 | B], a[r2], a[r2 | B], a[r3], a[r3 | B], a[r4], a[r4 | B], a[r5], a[r5 | B], \
       a[r6], a[r6 | B], a[r7], a[r7 | B], p0, p1, p2, p3);                                                       \
  }
  // Every skipped body is a TAKEN branch, and taken branches are what this dispatch costs (one more test per register
  // bit was 5 % of the pass time): two groups, so a gate passes at most four tests and an empty bit one (the caller's).
  if (kf & 7) {
    if (kf & 4) { TCMI_G8(vm2_gate8_rx) }
    if (kf & 2) { TCMI_G8(vm2_gate8_real) }
    if (kf & 1) { TCMI_G8(vm2_gate8_gen) }
  }
#undef TCMI_G8
#define TCMI_S8(FN)                                                                                               \
  _Pragma("unroll") for (int g = 0; g < NR / 2; g += 8) {                                                         \
    const int r0 = ins0(g, J), r1 = ins0(g + 1, J), r2 = ins0(g + 2, J), r3 = ins0(g + 3, J);                     \
    const int r4 = ins0(g + 4, J), r5 = ins0(g + 5, J), r6 = ins0(g + 6, J), r7 = ins0(g + 7, J);                 \
    FN(a[r0], a[r0 | B], a[r1], a[r1 | B], a[r2], a[r2 | B], a[r3], a[r3 | B], a[r4], a[r4 | B], a[r5], a[r5 | B], \
       a[r6], a[r6 | B], a[r7], a[r7 | B], p0);                                                                   \
  }
  // rotations in three-shear form, p0 = (u, v) (plan.g1_shear_flavor; the pulled-out sign is handled by the caller);
  // the rx-like class also in TWO-shear form (class 4: the builder's choice per batch element, plan.shear2_gates): the
  // first two shears, then the third for class 6 only
  if (kf & (16 | 32 | 64)) {
    if (kf & (16 | 64)) { TCMI_S8(vm2_shear2_8_rx) }
    if (kf & 64) { TCMI_S8(vm2_shear3rd_8_rx) }
    if (kf & 32) { TCMI_S8(vm2_shear8_real) }
  }
#undef TCMI_S8
}

template <int NR, int J>
__device__ __forceinline__ void vm2_diagb(v2f (&a)[NR], v2f e) {
#pragma unroll
  for (int g = 0; g < NR / 2; g += 4) {
    constexpr int B = 1 << J;
    const int r0 = ins0(g, J), r1 = ins0(g + 1, J), r2 = ins0(g + 2, J), r3 = ins0(g + 3, J);
    vm2_cmul44v(a[r0], a[r1], a[r2], a[r3], a[r0 | B], a[r1 | B], a[r2 | B], a[r3 | B], e);
  }
}

// KIND 0 dense, 1 CNOT control JA, 2 CNOT control JB, 3 SWAP      (index = (bit JA << 1) | bit JB, JA < JB)
template <int NR, int JA, int JB>
__device__ __forceinline__ void vm2_g2(v2f (&a)[NR], int kflag, KV2 m) {
  constexpr int A = 1 << JA, B = 1 << JB;
  if (kflag & 1) {
    const v2f m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7];
    const v2f m8 = m[8], m9 = m[9], m10 = m[10], m11 = m[11], m12 = m[12], m13 = m[13], m14 = m[14], m15 = m[15];
#pragma unroll
    for (int g = 0; g < NR / 4; g += 2) {
      const int r0 = ins00(g, JA, JB), r1 = ins00(g + 1, JA, JB);
      vm2_g2x2(a[r0], a[r0 | B], a[r0 | A], a[r0 | A | B], a[r1], a[r1 | B], a[r1 | A], a[r1 | A | B],
               m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15);
    }
  }
  if (kflag & 2) {
#pragma unroll
    for (int g = 0; g < NR / 4; g += 4) {
      const int r0 = ins00(g, JA, JB), r1 = ins00(g + 1, JA, JB), r2 = ins00(g + 2, JA, JB), r3 = ins00(g + 3, JA, JB);
      vm2_swap4(a[r0 | A], a[r0 | A | B], a[r1 | A], a[r1 | A | B], a[r2 | A], a[r2 | A | B], a[r3 | A], a[r3 | A | B]);
    }
  }
  if (kflag & 4) {
#pragma unroll
    for (int g = 0; g < NR / 4; g += 4) {
      const int r0 = ins00(g, JA, JB), r1 = ins00(g + 1, JA, JB), r2 = ins00(g + 2, JA, JB), r3 = ins00(g + 3, JA, JB);
      vm2_swap4(a[r0 | B], a[r0 | A | B], a[r1 | B], a[r1 | A | B], a[r2 | B], a[r2 | A | B], a[r3 | B], a[r3 | A | B]);
    }
  }
  if (kflag & 8) {
#pragma unroll
    for (int g = 0; g < NR / 4; g += 4) {
      const int r0 = ins00(g, JA, JB), r1 = ins00(g + 1, JA, JB), r2 = ins00(g + 2, JA, JB), r3 = ins00(g + 3, JA, JB);
      vm2_swap4(a[r0 | B], a[r0 | A], a[r1 | B], a[r1 | A], a[r2 | B], a[r2 | A], a[r3 | B], a[r3 | A]);
    }
  }
}

#define TCMI_DIAGB_OPS ((1 << TCMI_OP_DIAGB) | (1 << TCMI_OP_DIAGB2) | (1 << TCMI_OP_DIAG))
#define TCMI_DIAG_OPS (TCMI_DIAGB_OPS | (1 << TCMI_OP_DIAGC) | (1 << TCMI_OP_DIAGCW))
// TOFF: type of the per-thread byte offset (uint32_t while every tile bit is below bit 29, i.e. n <= 29)
template <int R, int LT, typename TOFF>
__global__ __launch_bounds__(1 << LT, (1024 >> LT)) void pass2_kernel(v2f* __restrict__ state, long long state_stride,
                                                           const int* __restrict__ desc_g,
                                                           const float* __restrict__ ctab_g,
                                                           const float* __restrict__ ptab_g, long long ptab_stride) {
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);

  const uint32_t tid = threadIdx.x;
  state += (long long)blockIdx.y * state_stride;
  const KInt desc = (KInt)desc_g;
  const KPtr<float> ctab = (KPtr<float>)ctab_g;
  const KPtr<float> ptab = (KPtr<float>)(ptab_g + (long long)blockIdx.y * ptab_stride);

  const int nrounds = desc[5];
  unsigned long long x = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < T; ++i) {
    const int p = desc[8 + i];
    const unsigned long long low = (1ull << p) - 1ull;
    x = ((x & ~low) << 1) | (x & low);
  }
  const uint32_t wg_base = (uint32_t)x;

  v2f a[NR];
  int pc = TCMI_HDR_WORDS;
  uint32_t tphys;
  {  // tile load, layout of round 0 (register bit 0 = tile bit 0: two amplitudes per 16-byte access).
     // Address = uniform base (workgroup + register-index part, SGPRs) + one 32-bit per-thread byte offset.
    const KInt rr = desc + pc;
    tphys = xor_masks<LT>(tid, rr + 8);
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rr[2 + j];
    const char* __restrict__ base = reinterpret_cast<const char*>(state + wg_base);
    const TOFF toff = (TOFF)tphys * sizeof(v2f);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      const v4f v = *reinterpret_cast<const v4f*>(base + (unsigned long long)reg_mask<R>(r, rpm) * sizeof(v2f) + toff);
      a[r] = v.xy;
      a[r + 1] = v.zw;
    }
  }

  int pc_cur = pc;
  uint32_t sgnbits = 0;  // sign pulled out of the shear-form gates of this pass (bit 31)
#pragma unroll 1
  for (int k = 0;; ++k) {
    pc_cur = pc;
    const KInt rr = desc + pc;
    const int nops = rr[0];
    int q = pc + TCMI_RR_WORDS;
#pragma unroll 1
    for (int o = 0; o < nops; ++o) {
      const int f = onehot(desc[q]);
      int qn = q;
      if (f & (1 << TCMI_OP_G1M)) {
        const int mk = desc[q + 1];
        const KV2 mp = (KV2)(ptab + desc[q + 2]);
        qn = q + 3;
        // every matrix of the op in one burst of scalar loads (one wait per op, not one per gate)
        v2f cf[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) cf[i] = mp[i];
        // structure class per register bit, one-hot and opaque: 1 general, 2 real, 4 rx-like, 0 = no gate on this bit
#define TCMI_G1(J)                                                                                                  \
  if constexpr (R > J) {                                                                                             \
    const int sh = (mk >> (20 + J)) & (mk >> J) & 1; /* three-shear form: classes 5 (real) and 6 (rx-like) */          \
    const int two = sh & (int)(__float_as_uint(cf[4 * J + 1].y) >> 30); /* two-shear form: class 4 (rx-like only) */      \
    const int kf = onehot_if((mk >> J) & 1, ((mk >> (8 + 2 * J)) & 3) + 4 * sh - 2 * two);                           \
    if (kf) vm2_g1<NR, J>(a, kf, cf[4 * J], cf[4 * J + 1], cf[4 * J + 2], cf[4 * J + 3]);                            \
    sgnbits ^= sh ? (__float_as_uint(cf[4 * J + 1].x) & 0x80000000u) : 0u;                                           \
  }
        TCMI_G1(0) TCMI_G1(1) TCMI_G1(2) TCMI_G1(3) TCMI_G1(4) TCMI_G1(5)
#undef TCMI_G1
      }
      // diagonal ops in two nested groups (skipped tests are taken branches, see vm2_g1)
      if (f & TCMI_DIAG_OPS) {
        if (f & ((1 << TCMI_OP_DIAGC) | (1 << TCMI_OP_DIAGCW))) {
          // register table; OP_DIAGCW: one of 2^nsel variants, picked by sign functions that are uniform over the wave
          // (register-x-thread phase terms on wave-uniform bits ride on this multiply instead of needing their own)
          int toff = desc[q + 1];
          qn = q + 2;
          if (f & (1 << TCMI_OP_DIAGCW)) {
              const int nsel = desc[q + 2];
              const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wg_base | tphys));
              int v = 0;
  #pragma unroll
              for (int k2 = 0; k2 < 3; ++k2)
                if (k2 < nsel) v |= (__popc(widx & (uint32_t)desc[q + 3 + k2]) & 1) << k2;
              toff += 2 * NR * v;
              qn = q + 6;
          }
          const KV2 tp = (KV2)(ptab + toff);
  #pragma unroll
          for (int h = 0; h < NR; h += 16) {  // 32 scalar registers of table per burst
            v2f t[16];
  #pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = tp[h + i];
  #pragma unroll
            for (int r = 0; r < 16; r += 8)
              vm2_cmul8s(a[h + r], a[h + r + 1], a[h + r + 2], a[h + r + 3], a[h + r + 4], a[h + r + 5], a[h + r + 6],
                         a[h + r + 7], t[r], t[r + 1], t[r + 2], t[r + 3], t[r + 4], t[r + 5], t[r + 6], t[r + 7]);
          }
        }
        if (f & TCMI_DIAGB_OPS) {
          if (f & (1 << TCMI_OP_DIAGB)) {
            const int fj = onehot(desc[q + 1]);
            const uint32_t m = (uint32_t)desc[q + 2];
            const KPtr<float> tp = ptab + desc[q + 3];
            qn = q + 4;
            v2f e;
            e.x = tp[0];
            const float sn = tp[1];
            e.y = (__popc((wg_base | tphys) & m) & 1) ? -sn : sn;
    #define TCMI_DB(J) \
      if constexpr (R > J) { if (fj & (1 << J)) vm2_diagb<NR, J>(a, e); }
            TCMI_DB(0) TCMI_DB(1) TCMI_DB(2) TCMI_DB(3) TCMI_DB(4) TCMI_DB(5)
    #undef TCMI_DB
          }
          if (f & (1 << TCMI_OP_DIAGB2)) {
            // two sign functions on one register bit: factor = table[s1 + 2 s2] (4 builder-evaluated entries)
            const int fj = onehot(desc[q + 1]);
            const uint32_t m1 = (uint32_t)desc[q + 2], m2 = (uint32_t)desc[q + 3];
            const KV2 tp = (KV2)(ptab + desc[q + 4]);
            qn = q + 5;
            const v2f t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
            const uint32_t tidx = wg_base | tphys;
            const bool s1 = __popc(tidx & m1) & 1, s2 = __popc(tidx & m2) & 1;
            v2f e;
            e.x = s2 ? (s1 ? t3.x : t2.x) : (s1 ? t1.x : t0.x);
            e.y = s2 ? (s1 ? t3.y : t2.y) : (s1 ? t1.y : t0.y);
    #define TCMI_DB(J) \
      if constexpr (R > J) { if (fj & (1 << J)) vm2_diagb<NR, J>(a, e); }
            TCMI_DB(0) TCMI_DB(1) TCMI_DB(2) TCMI_DB(3) TCMI_DB(4) TCMI_DB(5)
    #undef TCMI_DB
          }
          if (f & (1 << TCMI_OP_DIAG)) {
            // general phase polynomial (many thread x register terms): per-thread phases in turns, hardware sin / cos
            const int nA = desc[q + 1], nB = desc[q + 2], nC = desc[q + 3];
            const KPtr<float> cf = ptab + desc[q + 4];
            int qq = q + 5;
            const uint32_t tidx = wg_base | tphys;
            double phi = 0.0;
    #pragma unroll 1
            for (int e = 0; e < nA; e += TCMI_DIAG_CHUNK) {
              uint32_t mk[TCMI_DIAG_CHUNK];
              float cc[TCMI_DIAG_CHUNK];
    #pragma unroll
              for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
                mk[i] = (uint32_t)desc[qq + e + i];
                cc[i] = cf[e + i];
              }
    #pragma unroll
              for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
                const double c = (double)cc[i];
                phi += (__popc(tidx & mk[i]) & 1) ? -c : c;
              }
            }
            qq += nA;
            double cj[R];
    #pragma unroll
            for (int j = 0; j < R; ++j) cj[j] = 0.0;
    #pragma unroll 1
            for (int e = 0; e < nB; e += TCMI_DIAG_CHUNK) {
              uint32_t mk[TCMI_DIAG_CHUNK];
              int jj[TCMI_DIAG_CHUNK];
              float cc[TCMI_DIAG_CHUNK];
    #pragma unroll
              for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
                mk[i] = (uint32_t)desc[qq + e + i];
                jj[i] = desc[qq + nB + e + i];
                cc[i] = cf[nA + e + i];
              }
    #pragma unroll
              for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
                const double c = (double)cc[i];
                const double sgn = (__popc(tidx & mk[i]) & 1) ? -c : c;
    #pragma unroll
                for (int j = 0; j < R; ++j) cj[j] += (j == jj[i]) ? sgn : 0.0;
              }
            }
            qq += 2 * nB;
            // phases of eight register indices at a time (the full 2^R table would cost 32 more live registers)
            const float ph0 = (float)(phi - rint(phi));
            float cjf[R];
    #pragma unroll
            for (int j = 0; j < R; ++j) cjf[j] = (float)(cj[j] - rint(cj[j]));
            const KInt mC = desc + qq;
            const KPtr<float> cC = cf + nA + nB;
            qn = qq + nC;
    #pragma unroll
            for (int h = 0; h < NR; h += 8) {
              float ph[8];
    #pragma unroll
              for (int i = 0; i < 8; ++i) {
                float v = ph0;
    #pragma unroll
                for (int j = 0; j < R; ++j) v += (((h + i) >> j) & 1) ? -cjf[j] : cjf[j];
                ph[i] = v;
              }
    #pragma unroll 1
              for (int e = 0; e < nC; ++e) {
                const uint32_t rmask = (uint32_t)mC[e];
                const float c = cC[e];
    #pragma unroll
                for (int i = 0; i < 8; ++i) ph[i] += (__popc((uint32_t)(h + i) & rmask) & 1) ? -c : c;
              }
              v2f e8[8];
    #pragma unroll
              for (int i = 0; i < 8; ++i) {
                float sn, cs;
                sincos_turns<float>(ph[i], &sn, &cs);
                e8[i].x = cs;
                e8[i].y = sn;
              }
              vm2_cmul8v(a[h], a[h + 1], a[h + 2], a[h + 3], a[h + 4], a[h + 5], a[h + 6], a[h + 7], e8[0], e8[1], e8[2],
                         e8[3], e8[4], e8[5], e8[6], e8[7]);
            }
          }
        }
      }
      if (f & (1 << TCMI_OP_G2)) {
        const int jak = desc[q + 1], jb = desc[q + 2];
        const KV2 m = (KV2)tab_ptr<float>(desc[q + 3], ctab, ptab);
        qn = q + 4;
        const int ja = jak & 0xff;
        const int kflag = onehot(jak >> 8);
        const int fab = onehot(ja * 5 + jb);  // one-hot over the (ja, jb) pairs, ja < jb <= 5
#define TCMI_G2(A, B) \
  if constexpr (R > B) { if (fab & (1 << (A * 5 + B))) vm2_g2<NR, A, B>(a, kflag, m); }
        TCMI_G2(0, 1) TCMI_G2(0, 2) TCMI_G2(0, 3) TCMI_G2(0, 4) TCMI_G2(0, 5)
        TCMI_G2(1, 2) TCMI_G2(1, 3) TCMI_G2(1, 4) TCMI_G2(1, 5)
        TCMI_G2(2, 3) TCMI_G2(2, 4) TCMI_G2(2, 5)
        TCMI_G2(3, 4) TCMI_G2(3, 5)
        TCMI_G2(4, 5)
#undef TCMI_G2
      }
      q = qn;
    }
    pc += TCMI_RR_WORDS + rr[1];
    if (k == nrounds - 1) break;

    // ---- LDS exchange into the layout of round k + 1: real parts, then imaginary parts, through one 2^T-float
    // buffer (32 KiB at T = 13 -> four workgroups per CU; the 8-byte form allowed two).  Slot addresses follow a
    // Gray-code walk over the register index: one VGPR-only v_xor per access (2 issue cycles; the same xor with
    // the mask in an SGPR takes 4) and no table of 2^R addresses kept alive next to the amplitudes. ----
    {
      const KInt rn = desc + pc;
      tphys = xor_masks<LT>(tid, rn + 8);
      const uint32_t wslot = xor_masks<LT>(tid, rr + 40) << 2, rslot = xor_masks<LT>(tid, rn + 24) << 2;
      char* const lb = reinterpret_cast<char*>(lds);
      uint32_t ad;
      uint32_t mv[R];  // the masks of the current phase in VGPRs (re-materialised per phase: 5 moves, 5 fewer live registers)
#define TCMI_MASKS(SRC, OFF)                                                  \
  _Pragma("unroll") for (int j = 0; j < R; ++j) mv[j] = to_vgpr_v((uint32_t)SRC[OFF + j] << 2);
#define TCMI_WALK(BASE, STMT)                                                 \
  ad = BASE;                                                                  \
  _Pragma("unroll") for (int g = 0; g < NR; ++g) {                            \
    if (g) ad ^= mv[__builtin_ctz(g)];                                        \
    constexpr_for_r(g ^ (g >> 1));                                            \
    STMT;                                                                     \
  }
#define constexpr_for_r(X) const int r = (X)
      TCMI_MASKS(rr, 34)
      TCMI_WALK(wslot, *reinterpret_cast<float*>(lb + ad) = a[r].x)
      __syncthreads();
      TCMI_MASKS(rn, 18)
      TCMI_WALK(rslot, a[r].x = *reinterpret_cast<const float*>(lb + ad))
      __syncthreads();
      TCMI_MASKS(rr, 34)
      TCMI_WALK(wslot, *reinterpret_cast<float*>(lb + ad) = a[r].y)
      __syncthreads();
      TCMI_MASKS(rn, 18)
      TCMI_WALK(rslot, a[r].y = *reinterpret_cast<const float*>(lb + ad))
      __syncthreads();  // all reads done before the next exchange overwrites the buffer
#undef TCMI_MASKS
#undef TCMI_WALK
#undef constexpr_for_r
    }
  }

  if (sgnbits) {  // wave-uniform
    v2f m1;
    m1.x = -1.f;
    m1.y = -1.f;
#pragma unroll
    for (int r = 0; r < NR; r += 8) vm2_scale8(a[r], a[r + 1], a[r + 2], a[r + 3], a[r + 4], a[r + 5], a[r + 6], a[r + 7], m1);
  }
  {  // tile store, layout of the last round (constrained like round 0)
    const KInt rl = desc + pc_cur;
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rl[2 + j];
    char* __restrict__ base = reinterpret_cast<char*>(state + wg_base);
    const TOFF toff = (TOFF)tphys * sizeof(v2f);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      v4f v;
      v.xy = a[r];
      v.zw = a[r + 1];
      *reinterpret_cast<v4f*>(base + (unsigned long long)reg_mask<R>(r, rpm) * sizeof(v2f) + toff) = v;
    }
  }
}

template <int R, int LT>
static int launch_pass2(void* state, long long state_stride, int batch, int n, const int* desc, const void* ctab,
                        const void* ptab, long long ptab_stride, hipStream_t st) {
  constexpr int T = R + LT;
  const size_t lds = sizeof(float) << T;
  auto kern = n <= 29 ? pass2_kernel<R, LT, uint32_t> : pass2_kernel<R, LT, unsigned long long>;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return TCMI_ERR_HIP;
  }
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(1u << LT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<v2f*>(state), state_stride, desc,
                     reinterpret_cast<const float*>(ctab), reinterpret_cast<const float*>(ptab), ptab_stride);
  return hipGetLastError() == hipSuccess ? TCMI_OK : TCMI_ERR_HIP;
}

// complex64 gate pass (no measurement output).  Returns -1 when (R, LT) has no second-generation variant.
int run_pass2_c64(void* state, long long state_stride, int batch, int n, int R, int LT, const int* desc,
                  const void* ctab, const void* ptab, long long ptab_stride, hipStream_t st) {
  if (R == 5 && LT == 8) return launch_pass2<5, 8>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, st);
  if (R == 4 && LT == 8) return launch_pass2<4, 8>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, st);
  if (R == 4 && LT == 9) return launch_pass2<4, 9>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, st);
  if (R == 5 && LT == 9) return launch_pass2<5, 9>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, st);
  return -1;
}

}  // namespace tcmi
