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
// The kernel is launched by tcmi_run_pass (tcmi_vm.hip) for complex64 gate passes with R = 5 (n >= 13).

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

template <int NR, int J>
__device__ __forceinline__ void vm2_g1(v2f (&a)[NR], int kind, KV2 mp) {
  const v2f p0 = mp[0], p1 = mp[1], p2 = mp[2], p3 = mp[3];
#pragma unroll
  for (int g = 0; g < NR / 2; g += 4) {
    constexpr int B = 1 << J;
    const int r0 = ins0(g, J), r1 = ins0(g + 1, J), r2 = ins0(g + 2, J), r3 = ins0(g + 3, J);
    vm2_gate4(a[r0], a[r0 | B], a[r1], a[r1 | B], a[r2], a[r2 | B], a[r3], a[r3 | B], p0, p1, p2, p3, kind);
  }
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

template <int R, int LT>
__global__ __launch_bounds__(1 << LT, 2) void pass2_kernel(v2f* __restrict__ state, long long state_stride,
                                                           const int* __restrict__ desc_g,
                                                           const float* __restrict__ ctab_g,
                                                           const float* __restrict__ ptab_g, long long ptab_stride) {
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v2f* lds = reinterpret_cast<v2f*>(smem);

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
  {  // tile load, layout of round 0 (register bit 0 = tile bit 0: two amplitudes per 16-byte access)
    const KInt rr = desc + pc;
    tphys = xor_masks<LT>(tid, rr + 8);
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rr[2 + j];
    const v2f* __restrict__ src = state + (wg_base | tphys);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      const v4f v = *reinterpret_cast<const v4f*>(src + reg_mask<R>(r, rpm));
      a[r] = v.xy;
      a[r + 1] = v.zw;
    }
  }

  int pc_cur = pc;
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
        // kind per register bit: 0 general, 1 real, 2 rx-like, 3 = no gate on this bit
#define TCMI_G1(J) \
  if constexpr (R > J) vm2_g1<NR, J>(a, ((mk >> J) & 1) ? ((mk >> (8 + 2 * J)) & 3) : 3, mp + 4 * J);
        TCMI_G1(0) TCMI_G1(1) TCMI_G1(2) TCMI_G1(3) TCMI_G1(4) TCMI_G1(5)
#undef TCMI_G1
      }
      if (f & (1 << TCMI_OP_DIAGC)) {
        const KV2 tp = (KV2)(ptab + desc[q + 1]);
        qn = q + 2;
#pragma unroll
        for (int r = 0; r < NR; r += 8)
          vm2_cmul8s(a[r], a[r + 1], a[r + 2], a[r + 3], a[r + 4], a[r + 5], a[r + 6], a[r + 7], tp[r], tp[r + 1],
                     tp[r + 2], tp[r + 3], tp[r + 4], tp[r + 5], tp[r + 6], tp[r + 7]);
      }
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
        float ph[NR];
        ph[0] = (float)(phi - rint(phi));
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const float c = (float)(cj[j] - rint(cj[j]));
#pragma unroll
          for (int r = 0; r < (1 << j); ++r) {
            ph[r | (1 << j)] = ph[r] - c;
            ph[r] += c;
          }
        }
#pragma unroll 1
        for (int e = 0; e < nC; ++e) {
          const uint32_t rmask = (uint32_t)desc[qq + e];
          const float c = cf[nA + nB + e];
#pragma unroll
          for (int r = 0; r < NR; ++r) ph[r] += (__popc((uint32_t)r & rmask) & 1) ? -c : c;
        }
        qq += nC;
        qn = qq;
#pragma unroll
        for (int r = 0; r < NR; r += 8) {
          v2f e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float s, c;
            sincos_turns<float>(ph[r + i], &s, &c);
            e[i].x = c;
            e[i].y = s;
          }
          vm2_cmul8v(a[r], a[r + 1], a[r + 2], a[r + 3], a[r + 4], a[r + 5], a[r + 6], a[r + 7], e[0], e[1], e[2], e[3],
                     e[4], e[5], e[6], e[7]);
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

    // ---- LDS exchange into the layout of round k + 1 ----
    {
      const uint32_t tslot = xor_masks<LT>(tid, rr + 40);
      uint32_t wsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) wsm[j] = (uint32_t)rr[34 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) lds[tslot ^ reg_mask<R>(r, wsm)] = a[r];
    }
    __syncthreads();
    {
      const KInt rn = desc + pc;
      tphys = xor_masks<LT>(tid, rn + 8);
      const uint32_t tslot = xor_masks<LT>(tid, rn + 24);
      uint32_t rsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) rsm[j] = (uint32_t)rn[18 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) a[r] = lds[tslot ^ reg_mask<R>(r, rsm)];
    }
    __syncthreads();  // all reads done before the next exchange overwrites the tile
  }

  {  // tile store, layout of the last round (constrained like round 0)
    const KInt rl = desc + pc_cur;
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rl[2 + j];
    v2f* __restrict__ dst = state + (wg_base | tphys);
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      v4f v;
      v.xy = a[r];
      v.zw = a[r + 1];
      *reinterpret_cast<v4f*>(dst + reg_mask<R>(r, rpm)) = v;
    }
  }
}

template <int R, int LT>
static int launch_pass2(void* state, long long state_stride, int batch, int n, const int* desc, const void* ctab,
                        const void* ptab, long long ptab_stride, hipStream_t st) {
  constexpr int T = R + LT;
  const size_t lds = sizeof(v2f) << T;
  auto kern = pass2_kernel<R, LT>;
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
  if (R == 5 && LT == 9) return launch_pass2<5, 9>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, st);
  return -1;
}

}  // namespace tcmi
