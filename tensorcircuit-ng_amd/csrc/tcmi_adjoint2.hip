// tcmi adjoint sweep, complex64, second generation (gfx950 / MI355X only).
//
// Same backward pass descriptors as tcmi_adjoint.hip (layout: tcmi_vm.h "backward ops"; semantics: psi is
// un-computed gate by gate with U^dagger while the cotangent lambda follows, every parametrised gate adds
// Re<lambda|K|psi> to its gradient slot -- the reverse-mode rule that replaces framework AD through stored
// intermediates, reference tensorcircuit/backends/pytorch_backend.py:775-786), issued the way tcmi_vm2.hip
// issues the forward pass: both vectors as (re, im) register pairs, packed-f32 instruction bodies with tied
// operands (tcmi_vm2_asm.inc), no register-array copies at control-flow merges, planar LDS exchange.
//
// Tile: R = 4 register bits, 512 threads (16 + 16 amplitude pairs per thread, 8 waves per workgroup, two
// workgroups per CU = 4 waves per SIMD).  Handles one-qubit gate ops and table-form diagonal flushes (OP_DIAGF); plans with dense
// two-qubit gates stay on the first-generation kernel (the host checks, tcmi/executor.py).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tcmi_vm.h"
#include "tcmi_dev.h"

namespace tcmi {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v2f TCMI_K* KV2;

#include "tcmi_vm2_asm.inc"

__host__ __device__ constexpr int ains0(int k, int J) { return ((k >> J) << (J + 1)) | (k & ((1 << J) - 1)); }

__device__ __forceinline__ int aonehot_if(int on, int v) {
  int f;
  asm("s_lshl_b32 %0, %1, %2" : "=s"(f) : "s"(on), "s"(v) : "scc");
  return f;
}
__device__ __forceinline__ int aonehot(int v) {
  int f;
  asm("s_lshl_b32 %0, 1, %1" : "=s"(f) : "s"(v) : "scc");
  return f;
}
__device__ __forceinline__ uint32_t ato_vgpr_v(uint32_t v) {
  uint32_t r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}

// U^dagger on register bit J of one vector; kf = one-hot structure class (1 general, 2 real, 4 rx-like, 32 / 64 shear forms)
template <int NR, int J>
__device__ __forceinline__ void adj_apply(v2f (&a)[NR], int kf, v2f p0, v2f p1, v2f p2, v2f p3) {
  constexpr int B = 1 << J;
#define TCMI_A8(FN, ...)                                                                                              \
  _Pragma("unroll") for (int g = 0; g < NR / 2; g += 8) {                                                             \
    const int r0 = ains0(g, J), r1 = ains0(g + 1, J), r2 = ains0(g + 2, J), r3 = ains0(g + 3, J);                     \
    const int r4 = ains0(g + 4, J), r5 = ains0(g + 5, J), r6 = ains0(g + 6, J), r7 = ains0(g + 7, J);                 \
    FN(a[r0], a[r0 | B], a[r1], a[r1 | B], a[r2], a[r2 | B], a[r3], a[r3 | B], a[r4], a[r4 | B], a[r5], a[r5 | B],     \
       a[r6], a[r6 | B], a[r7], a[r7 | B], __VA_ARGS__);                                                              \
  }
  if (kf & 4) { TCMI_A8(vm2_gate8_rx, p0, p1, p2, p3) }
  if (kf & 2) { TCMI_A8(vm2_gate8_real, p0, p1, p2, p3) }
  if (kf & 1) { TCMI_A8(vm2_gate8_gen, p0, p1, p2, p3) }
  // U^dagger as three shears, p0 = (u, v); its sign is common to psi and lambda and cancels in every gradient
  if (kf & 32) { TCMI_A8(vm2_shear8_real, p0) }
  if (kf & 64) { TCMI_A8(vm2_shear8_rx, p0) }
#undef TCMI_A8
}

// the same on psi and lambda with one test per structure class; (p0 .. p3) = the record of the gate (shear forms: p0)
template <int NR, int J>
__device__ __forceinline__ void adj_apply2(v2f (&a)[NR], v2f (&l)[NR], int kf, v2f p0, v2f p1, v2f p2, v2f p3) {
  constexpr int B = 1 << J;
#define TCMI_A8(V, FN, ...)                                                                                           \
  _Pragma("unroll") for (int g = 0; g < NR / 2; g += 8) {                                                             \
    const int r0 = ains0(g, J), r1 = ains0(g + 1, J), r2 = ains0(g + 2, J), r3 = ains0(g + 3, J);                     \
    const int r4 = ains0(g + 4, J), r5 = ains0(g + 5, J), r6 = ains0(g + 6, J), r7 = ains0(g + 7, J);                 \
    FN(V[r0], V[r0 | B], V[r1], V[r1 | B], V[r2], V[r2 | B], V[r3], V[r3 | B], V[r4], V[r4 | B], V[r5], V[r5 | B],     \
       V[r6], V[r6 | B], V[r7], V[r7 | B], __VA_ARGS__);                                                              \
  }
  // two groups of tests: every skipped body is a taken branch (see vm2_g1 in tcmi_vm2.hip)
  if (kf & (16 | 32 | 64)) {
    // three shears, (u, v) only
    if (kf & 64) { TCMI_A8(a, vm2_shear8_rx, p0) TCMI_A8(l, vm2_shear8_rx, p0) }
    // two-shear form (plan.shear2_gates, rx-like class only): psi is left with the pending factor diag(c, 1/c), lambda
    // -- sheared in the other order -- with its reciprocal, so Im(conj(lambda) psi) and every gradient bilinear stay
    // what they are
    if (kf & 16) { TCMI_A8(a, vm2_shear2_8_rx, p0) TCMI_A8(l, vm2_shear2l_8_rx, p0) }
    if (kf & 32) { TCMI_A8(a, vm2_shear8_real, p0) TCMI_A8(l, vm2_shear8_real, p0) }
  }
  if (kf & 7) {
    if (kf & 4) { TCMI_A8(a, vm2_gate8_rx, p0, p1, p2, p3) TCMI_A8(l, vm2_gate8_rx, p0, p1, p2, p3) }
    if (kf & 2) { TCMI_A8(a, vm2_gate8_real, p0, p1, p2, p3) TCMI_A8(l, vm2_gate8_real, p0, p1, p2, p3) }
    if (kf & 1) { TCMI_A8(a, vm2_gate8_gen, p0, p1, p2, p3) TCMI_A8(l, vm2_gate8_gen, p0, p1, p2, p3) }
  }
#undef TCMI_A8
}

// a[r] *= (e.x + i e.y) for z_J(r) = +1, the conjugate for z_J(r) = -1
template <int NR, int J>
__device__ __forceinline__ void adj_diagb(v2f (&a)[NR], v2f e) {
  constexpr int B = 1 << J;
#pragma unroll
  for (int g = 0; g < NR / 2; g += 4) {
    const int r0 = ains0(g, J), r1 = ains0(g + 1, J), r2 = ains0(g + 2, J), r3 = ains0(g + 3, J);
    vm2_cmul44v(a[r0], a[r1], a[r2], a[r3], a[r0 | B], a[r1 | B], a[r2 | B], a[r3 | B], e);
  }
}

// gradient of the gate on register bit J: Re<lambda|K|psi> summed over this thread's pairs
template <int NR, int J>
__device__ __forceinline__ float adj_grad(const v2f (&a)[NR], const v2f (&l)[NR], int kf, v2f k0, v2f k1, v2f k2, v2f k3) {
  constexpr int B = 1 << J;
  v2f c0 = {0.f, 0.f}, c1 = {0.f, 0.f}, d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
  float g = 0.f;
#define TCMI_GR(FN, ...)                                                                                            \
  _Pragma("unroll") for (int g8 = 0; g8 < NR / 2; g8 += 8) {                                                        \
    const int r0 = ains0(g8, J), r1 = ains0(g8 + 1, J), r2 = ains0(g8 + 2, J), r3 = ains0(g8 + 3, J);               \
    const int r4 = ains0(g8 + 4, J), r5 = ains0(g8 + 5, J), r6 = ains0(g8 + 6, J), r7 = ains0(g8 + 7, J);           \
    FN(a[r0], a[r0 | B], a[r1], a[r1 | B], a[r2], a[r2 | B], a[r3], a[r3 | B], l[r0], l[r0 | B], l[r1], l[r1 | B],   \
       l[r2], l[r2 | B], l[r3], l[r3 | B], ##__VA_ARGS__, c0, c1);                                                  \
    FN(a[r4], a[r4 | B], a[r5], a[r5 | B], a[r6], a[r6 | B], a[r7], a[r7 | B], l[r4], l[r4 | B], l[r5], l[r5 | B],   \
       l[r6], l[r6 | B], l[r7], l[r7 | B], ##__VA_ARGS__, d0, d1);                                                  \
  }
  if (kf & 4) {  // K = i kappa X, kappa = Im K01
    TCMI_GR(vm2_grad4_rx)
    g = -k1.y * ((c0.x + c0.y) + (c1.x + c1.y) + (d0.x + d0.y) + (d1.x + d1.y));
  }
  if (kf & 2) {  // real antisymmetric K = [[0, k01], [k10, 0]]
    TCMI_GR(vm2_grad4_real)
    g = k1.x * ((c0.x + c0.y) + (d0.x + d0.y)) + k2.x * ((c1.x + c1.y) + (d1.x + d1.y));
  }
  if (kf & 1) {
    TCMI_GR(vm2_grad4_gen, k0, k1, k2, k3)
    g = (c0.x + c0.y) + (c1.x + c1.y) + (d0.x + d0.y) + (d1.x + d1.y);
  }
#undef TCMI_GR
  return g;
}

template <int R, int LT, typename TOFF>
__global__ __launch_bounds__(1 << LT, (R >= 5 ? 512 : 1024) >> LT) void adjoint2_kernel(v2f* __restrict__ psi, v2f* __restrict__ lam,
                                                                       long long state_stride,
                                                                       const int* __restrict__ desc_g,
                                                                       const float* __restrict__ ctab_g,
                                                                       const float* __restrict__ ptab_g,
                                                                       long long ptab_stride, double* __restrict__ gout,
                                                                       long long gout_stride, int gcopies,
                                                                       long long gcopy_stride) {
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  // gradient contributions of the workgroup's waves meet in LDS (ds_add_f32) and leave as ONE f64 atomic per slot
  // and workgroup: gacc[i] / gidx[i] = sum and output slot of the i-th gradient event of the pass (events are
  // numbered in program order, identical in every wave because the op loop is descriptor-driven)
  constexpr int GACC = 256;
  float* const gacc = lds + (2 << T);
  int* const gidx = reinterpret_cast<int*>(gacc + GACC);
  int gev = 0;
  for (int i = threadIdx.x; i < GACC; i += (1 << LT)) gacc[i] = 0.f;
  __syncthreads();
#define TCMI_GFLUSH()                                                                  \
  {                                                                                    \
    __syncthreads();                                                                   \
    for (int i = threadIdx.x; i < gev; i += (1 << LT)) {                               \
      atomicAdd(gout + gidx[i], (double)gacc[i]);                                      \
      gacc[i] = 0.f;                                                                   \
    }                                                                                  \
    gev = 0;                                                                           \
    __syncthreads();                                                                   \
  }
#define TCMI_GADD(SLOT, VAL)                             \
  {                                                      \
    if (gev == GACC) TCMI_GFLUSH() /* workgroup-uniform */ \
    if (lane0) {                                         \
      atomicAdd(gacc + gev, VAL);                        \
      gidx[gev] = SLOT; /* same value from every wave */ \
    }                                                    \
    ++gev;                                               \
  }

  const uint32_t tid = threadIdx.x;
  psi += (long long)blockIdx.y * state_stride;
  lam += (long long)blockIdx.y * state_stride;
  gout += (long long)blockIdx.y * gout_stride + (long long)(blockIdx.x % (unsigned)gcopies) * gcopy_stride;
  const KInt desc = (KInt)desc_g;
  const KPtr<float> ptab = (KPtr<float>)(ptab_g + (long long)blockIdx.y * ptab_stride);
  const bool lane0 = (tid & 63) == 0;

  const int nrounds = desc[5];
  unsigned long long x = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < T; ++i) {
    const int p = desc[8 + i];
    const unsigned long long low = (1ull << p) - 1ull;
    x = ((x & ~low) << 1) | (x & low);
  }
  const uint32_t wg_base = (uint32_t)x;

  v2f a[NR], l[NR];
  int pc = TCMI_HDR_WORDS;
  uint32_t tphys;
  {
    const KInt rr = desc + pc;
    tphys = xor_masks<LT>(tid, rr + 8);
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rr[2 + j];
    const char* __restrict__ ba = reinterpret_cast<const char*>(psi + wg_base);
    const char* __restrict__ bl = reinterpret_cast<const char*>(lam + wg_base);
    const TOFF toff = (TOFF)tphys * sizeof(v2f);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      const unsigned long long ro = (unsigned long long)reg_mask<R>(r, rpm) * sizeof(v2f);
      const v4f v = *reinterpret_cast<const v4f*>(ba + ro + toff);
      const v4f w = *reinterpret_cast<const v4f*>(bl + ro + toff);
      a[r] = v.xy;
      a[r + 1] = v.zw;
      l[r] = w.xy;
      l[r + 1] = w.zw;
    }
  }

  int pc_cur = pc;
  uint32_t sgnbits = 0;  // sign pulled out of the three-shear gates of this pass (bit 31): psi and lambda both carry it
#pragma unroll 1
  for (int k = 0;; ++k) {
    pc_cur = pc;
    const KInt rr = desc + pc;
    const int nops = rr[0];
    int q = pc + TCMI_RR_WORDS;
#pragma unroll 1
    for (int o = 0; o < nops; ++o) {
      const int f = aonehot(desc[q]);
      int qn = q;
      if (f & (1 << TCMI_OP_G1M)) {
        // {4, mask | kinds << 8, ubase, kmask, kbase, gslot[R]}
        const int mk = desc[q + 1], kmask = desc[q + 3];
        const KV2 up = (KV2)(ptab + desc[q + 2]);
        const KV2 kp = (KV2)(ptab + desc[q + 4]);
        qn = q + 5 + R;
        // every record of the op in ONE burst of scalar loads: U^dagger (8 floats per register bit; the shear forms use
        // the first four: u, v, sign, form flag) and K.  One exposed load latency per op instead of three per gate --
        // the waves of this kernel spend more time waiting for dependent scalar loads than issuing VALU work.
        v2f uf[4 * R], kf_[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) uf[i] = up[i];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) kf_[i] = kp[i];
#define TCMI_BW(J)                                                                                           \
  if constexpr (R > J) {                                                                                     \
    if (aonehot_if((mk >> J) & 1, 0)) { /* register bits without a gate cost two scalar instructions */       \
      const int kd = (mk >> (8 + 2 * J)) & 3;                                                                \
      const int gf = aonehot_if((kmask >> J) & 1, kd); /* generator class: never shear */                     \
      const int sh = (mk >> (20 + J)) & 1;                                                                   \
      const v2f sf = uf[4 * J + 1]; /* shear records: {sign, form flag} */                                     \
      const int two = sh & (int)(__float_as_uint(sf.y) >> 30);                                               \
      sgnbits ^= sh ? (__float_as_uint(sf.x) & 0x80000000u) : 0u;                                            \
      const int kf = aonehot(kd + 4 * sh - 2 * two);                                                         \
      if (gf) {                                                                                              \
        float g = adj_grad<NR, J>(a, l, gf, kf_[4 * J], kf_[4 * J + 1], kf_[4 * J + 2], kf_[4 * J + 3]);     \
        g = wave_sum_uniform(g);                                                                             \
        TCMI_GADD(desc[q + 5 + J], g)                                                                        \
      }                                                                                                      \
      adj_apply2<NR, J>(a, l, kf, uf[4 * J], uf[4 * J + 1], uf[4 * J + 2], uf[4 * J + 3]);                   \
    }                                                                                                        \
  }
        TCMI_BW(0) TCMI_BW(1) TCMI_BW(2) TCMI_BW(3) TCMI_BW(4)
#undef TCMI_BW
      }
      if (f & (1 << TCMI_OP_DIAGF)) {
        // {8, cslot, hasC, nB, nA, nsel, m0, m1, m2, gsC[2^R], B: (j, mask, slot, gslot)*, A: (mask, gslot)*}
        const int cslot = desc[q + 1], hasC = desc[q + 2], nB = desc[q + 3], nA = desc[q + 4], nsel = desc[q + 5];
        int qq = q + 9 + NR;
        qn = qq + 4 * nB + 2 * nA;
        const uint32_t tidx = wg_base | tphys;
        int tvar = 0;   // wave-selected variant of the register table (OP_DIAGCW semantics)
        if (nsel > 0) {
          const uint32_t widx = (uint32_t)__builtin_amdgcn_readfirstlane((int)tidx);
#pragma unroll
          for (int k2 = 0; k2 < 3; ++k2)
            if (k2 < nsel) tvar |= (__popc(widx & (uint32_t)desc[q + 6 + k2]) & 1) << k2;
        }
        // w[r] = Im(conj(lambda) psi): invariant under the phases applied here.  Every term's gradient is a signed sum of
        // it over the registers: after the Walsh-Hadamard transform below w[k] = sum_r (-1)^{|r & k|} w[r], so the term
        // with register mask k reads its sum from w[k] (w[0]: register-independent terms, w[1 << j]: one register bit).
        float w[NR];
#pragma unroll
        for (int h = 0; h < NR; h += 8) {
          v2f t[8];
          vm2_cross8(a[h], a[h + 1], a[h + 2], a[h + 3], a[h + 4], a[h + 5], a[h + 6], a[h + 7], l[h], l[h + 1], l[h + 2],
                     l[h + 3], l[h + 4], l[h + 5], l[h + 6], l[h + 7], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
#pragma unroll
          for (int i = 0; i < 8; ++i) w[h + i] = t[i].x - t[i].y;
        }
        if ((hasC & 1) | nB | nA) {
#pragma unroll
          for (int j = 0; j < R; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r)
              if (!((r >> j) & 1)) {
                const float lo = w[r], hi = w[r | (1 << j)];
                w[r] = lo + hi;
                w[r | (1 << j)] = lo - hi;
              }
        }
        if (hasC & 1) {
          int gsc[NR];
#pragma unroll
          for (int k3 = 0; k3 < NR; ++k3) gsc[k3] = desc[q + 9 + k3];
#pragma unroll
          for (int k3 = 1; k3 < NR; ++k3)
            if (gsc[k3] >= 0) {  // wave-uniform
              const float sv = wave_sum_uniform(w[k3]);
              TCMI_GADD(gsc[k3], sv)
            }
        }
        // one of two table multiplies, picked by independent skip tests on an opaque one-hot flag (a nested if / else
        // would merge two modified copies of the amplitude arrays: 40 more live registers)
        const int tf = aonehot_if(cslot >= 0 ? 1 : 0, (hasC >> 1) & 1);
        if (tf & 1) {
          const KV2 tp = (KV2)(ptab + cslot + 2 * NR * tvar);
          v2f t[NR];
#pragma unroll
          for (int i = 0; i < NR; ++i) t[i] = tp[i];
#pragma unroll
          for (int h = 0; h < NR; h += 8) {
            vm2_cmul8s_conj(a[h], a[h + 1], a[h + 2], a[h + 3], a[h + 4], a[h + 5], a[h + 6], a[h + 7], t[h], t[h + 1],
                            t[h + 2], t[h + 3], t[h + 4], t[h + 5], t[h + 6], t[h + 7]);
            vm2_cmul8s_conj(l[h], l[h + 1], l[h + 2], l[h + 3], l[h + 4], l[h + 5], l[h + 6], l[h + 7], t[h], t[h + 1],
                            t[h + 2], t[h + 3], t[h + 4], t[h + 5], t[h + 6], t[h + 7]);
          }
        }
        if (tf & 2) {
          // real scale terms in the table (two-shear rotations): lambda's copy, with the reciprocal factors, follows
          // the 2^nsel variants of psi's
          const KV2 tp = (KV2)(ptab + cslot + 2 * NR * tvar);
          const KV2 tl = tp + (NR << nsel);
          constexpr int CH = NR < 16 ? NR : 16;  // both tables of a chunk in one burst of scalar loads (64 SGPRs)
#pragma unroll
          for (int h0 = 0; h0 < NR; h0 += CH) {
            v2f t[CH], u[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
              t[i] = tp[h0 + i];
              u[i] = tl[h0 + i];
            }
#pragma unroll
            for (int h = 0; h < CH; h += 8) {
              vm2_cmul8s_conj(a[h0 + h], a[h0 + h + 1], a[h0 + h + 2], a[h0 + h + 3], a[h0 + h + 4], a[h0 + h + 5],
                              a[h0 + h + 6], a[h0 + h + 7], t[h], t[h + 1], t[h + 2], t[h + 3], t[h + 4], t[h + 5], t[h + 6],
                              t[h + 7]);
              vm2_cmul8s_conj(l[h0 + h], l[h0 + h + 1], l[h0 + h + 2], l[h0 + h + 3], l[h0 + h + 4], l[h0 + h + 5],
                              l[h0 + h + 6], l[h0 + h + 7], u[h], u[h + 1], u[h + 2], u[h + 3], u[h + 4], u[h + 5], u[h + 6],
                              u[h + 7]);
            }
          }
        }
        if (nB > 0 || nA > 0) {
          const float w0 = w[0];
          float wj[R];
#pragma unroll
          for (int j = 0; j < R; ++j) wj[j] = w[1 << j];
#pragma unroll 1
          for (int e = 0; e < nB; ++e, qq += 4) {
            const int jj = desc[qq], slot = desc[qq + 2], gs = desc[qq + 3];
            const bool neg = __popc(tidx & (uint32_t)desc[qq + 1]) & 1;
            if (gs >= 0) {
              float ws = wj[0];
#pragma unroll
              for (int j = 1; j < R; ++j) ws = (j == jj) ? wj[j] : ws;
              ws = wave_sum_uniform(neg ? -ws : ws);
              TCMI_GADD(gs, ws)
            }
            if (slot >= 0) {
              const KPtr<float> tp = ptab + slot;
              v2f ev;
              ev.x = tp[0];
              const float sn = tp[1];
              ev.y = neg ? sn : -sn;  // the inverse phase: (cs - i ys) on z = +1
              const int fj = aonehot(jj);
#define TCMI_DB(J)                                                   \
  if constexpr (R > J) {                                             \
    if (fj & (1 << J)) {                                             \
      adj_diagb<NR, J>(a, ev);                                       \
      adj_diagb<NR, J>(l, ev);                                       \
    }                                                                \
  }
              TCMI_DB(0) TCMI_DB(1) TCMI_DB(2) TCMI_DB(3) TCMI_DB(4)
#undef TCMI_DB
            }
          }
#pragma unroll 1
          for (int e = 0; e < nA; ++e, qq += 2) {
            const int gs = desc[qq + 1];
            if (gs >= 0) {
              const bool neg = __popc(tidx & (uint32_t)desc[qq]) & 1;
              const float v = wave_sum_uniform(neg ? -w0 : w0);
              TCMI_GADD(gs, v)
            }
          }
        }
      }
      q = qn;
    }
    pc += TCMI_RR_WORDS + rr[1];
    if (k == nrounds - 1) break;

    // ---- LDS exchange, one VECTOR at a time as 8-byte (re, im) values through one 2^T x 8-byte buffer (32 KiB at
    // T = 12: still four workgroups per CU).  Against the four real planes of the first version: half the LDS
    // instructions, address walks and barriers (ds_write_b64 6 / ds_read_b64 2 LDS cycles per 8 bytes instead of
    // 2 x 4 / 2 x 2); the slot permutation is conflict-free for 8-byte accesses exactly when it is for 4-byte ones ----
    {
      const KInt rn = desc + pc;
      tphys = xor_masks<LT>(tid, rn + 8);
      const uint32_t wslot = xor_masks<LT>(tid, rr + 40) << 3, rslot = xor_masks<LT>(tid, rn + 24) << 3;
      char* const lb = reinterpret_cast<char*>(lds);
      uint32_t ad;
      uint32_t mv[R];
#define TCMI_MASKS(SRC, OFF) \
  _Pragma("unroll") for (int j = 0; j < R; ++j) mv[j] = ato_vgpr_v((uint32_t)SRC[OFF + j] << 3);
#define TCMI_WALK(BASE, STMT)                      \
  ad = BASE;                                       \
  _Pragma("unroll") for (int g = 0; g < NR; ++g) { \
    if (g) ad ^= mv[__builtin_ctz(g)];             \
    const int r = g ^ (g >> 1);                    \
    STMT;                                          \
  }
#define TCMI_VECTOR(V)                                                        \
  TCMI_MASKS(rr, 34)                                                          \
  TCMI_WALK(wslot, *reinterpret_cast<v2f*>(lb + ad) = V[r])                   \
  __syncthreads();                                                            \
  TCMI_MASKS(rn, 18)                                                          \
  TCMI_WALK(rslot, V[r] = *reinterpret_cast<const v2f*>(lb + ad))             \
  __syncthreads();
      TCMI_VECTOR(a)
      TCMI_VECTOR(l)
#undef TCMI_VECTOR
#undef TCMI_MASKS
#undef TCMI_WALK
    }
  }

  TCMI_GFLUSH()
#undef TCMI_GADD
#undef TCMI_GFLUSH
  // TCMI_FLAG_NOSTORE on the LAST pass of a sweep whose un-computed psi and propagated lambda nobody reads (a
  // value_and_grad step without the input-state cotangent): only the gradient sums leave the tile
  if (!(desc[6] & TCMI_FLAG_NOSTORE)) {
    if (sgnbits) {  // wave-uniform; no gradient sees it, the stored psi and the input-state cotangent do
      v2f m1;
      m1.x = -1.f;
      m1.y = -1.f;
#pragma unroll
      for (int h = 0; h < NR; h += 8) {
        vm2_scale8(a[h], a[h + 1], a[h + 2], a[h + 3], a[h + 4], a[h + 5], a[h + 6], a[h + 7], m1);
        vm2_scale8(l[h], l[h + 1], l[h + 2], l[h + 3], l[h + 4], l[h + 5], l[h + 6], l[h + 7], m1);
      }
    }
    const KInt rl = desc + pc_cur;
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rl[2 + j];
    char* __restrict__ ba = reinterpret_cast<char*>(psi + wg_base);
    char* __restrict__ bl = reinterpret_cast<char*>(lam + wg_base);
    const TOFF toff = (TOFF)tphys * sizeof(v2f);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      const unsigned long long ro = (unsigned long long)reg_mask<R>(r, rpm) * sizeof(v2f);
      v4f v, w;
      v.xy = a[r];
      v.zw = a[r + 1];
      w.xy = l[r];
      w.zw = l[r + 1];
      *reinterpret_cast<v4f*>(ba + ro + toff) = v;
      *reinterpret_cast<v4f*>(bl + ro + toff) = w;
    }
  }
}

template <int R, int LT>
static int launch_adjoint2(void* psi, void* lam, long long stride, int batch, int n, const int* desc, const void* ctab,
                           const void* ptab, long long ptab_stride, double* gout, long long gout_stride, int gcopies,
                           long long gcopy_stride, hipStream_t st) {
  constexpr int T = R + LT;
  if (n < T) return -1;
  const size_t lds = (2 * sizeof(float) << T) + 256 * (sizeof(float) + sizeof(int));
  auto kern = n <= 29 ? adjoint2_kernel<R, LT, uint32_t> : adjoint2_kernel<R, LT, unsigned long long>;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TCMI_ERR_HIP;
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(1u << LT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<v2f*>(psi), reinterpret_cast<v2f*>(lam), stride, desc,
                     reinterpret_cast<const float*>(ctab), reinterpret_cast<const float*>(ptab), ptab_stride, gout,
                     gout_stride, gcopies, gcopy_stride);
  return hipGetLastError() == hipSuccess ? TCMI_OK : TCMI_ERR_HIP;
}

// complex64 adjoint pass, (R, LT) = (4, 9) or (5, 8).  Returns -1 when there is no second-generation variant.
int run_adjoint2_c64(void* psi, void* lam, long long stride, int batch, int n, int R, int LT, const int* desc,
                     const void* ctab, const void* ptab, long long ptab_stride, double* gout, long long gout_stride,
                     int gcopies, long long gcopy_stride, hipStream_t st) {
  if (R == 4 && LT == 9)
    return launch_adjoint2<4, 9>(psi, lam, stride, batch, n, desc, ctab, ptab, ptab_stride, gout, gout_stride, gcopies, gcopy_stride, st);
  if (R == 4 && LT == 8)
    return launch_adjoint2<4, 8>(psi, lam, stride, batch, n, desc, ctab, ptab, ptab_stride, gout, gout_stride, gcopies, gcopy_stride, st);
  if (R == 5 && LT == 8)
    return launch_adjoint2<5, 8>(psi, lam, stride, batch, n, desc, ctab, ptab, ptab_stride, gout, gout_stride, gcopies, gcopy_stride, st);
  return -1;
}

}  // namespace tcmi
