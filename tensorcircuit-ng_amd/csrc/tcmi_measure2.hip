// tcmi measurement passes, complex64, second generation (gfx950 / MI355X only).
//
// <psi| P_t |psi> for a list of Pauli strings (reference circuit.py:833-913 expectation -> expectation_before ->
// contractor, one contraction per string) as read-only tile passes: same descriptors as the gate passes (tile
// bits, rounds, planar LDS exchange: tcmi_vm.h, written by plan.encode_measure_pass, emulated by
// oracle/plan_emulator.py), one op kind, TCMI_OP_EXPECT2.  The first-generation kernel (tcmi_vm.hip, MODE 1) spent
// 60-100 VALU instructions per string, thread and tile and was VALU bound at 1.2 TB/s on the 55-term TFIM energy;
// here
//   * Z-only strings read their signed sum over the registers from the Walsh-Hadamard transform of |a|^2 (one
//     transform per round, then a select + sign + wave reduction per string);
//   * a string with one X on a register bit and no register Z is 2 Re sum_pairs conj(a0) a1: one packed FMA per pair;
//   * the waves' partial sums meet in LDS, one f64 atomic per string and workgroup leaves the pass.
// Strings with two X / Y factors or Z factors on register bits take the general pair loop (register-index templates).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tcmi_vm.h"
#include "tcmi_dev.h"

namespace tcmi {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mins0(int k, int J) { return ((k >> J) << (J + 1)) | (k & ((1 << J) - 1)); }

__device__ __forceinline__ int monehot(int v) {
  int f;
  asm("s_lshl_b32 %0, 1, %1" : "=s"(f) : "s"(v) : "scc");
  return f;
}
__device__ __forceinline__ uint32_t mto_vgpr_v(uint32_t v) {
  uint32_t r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}

// sum_r sgn(r & zr) conj(a[r ^ XR]) a[r]   (XR: compile-time register mask of the X / Y factors)
template <int NR, int XR>
__device__ __forceinline__ void expect_pairs(const v2f (&a)[NR], uint32_t zr, float& re, float& im) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const v2f b = a[r ^ XR], v = a[r];
    const float sr = (__popc((uint32_t)r & zr) & 1) ? -1.f : 1.f;
    re = __builtin_fmaf(sr, __builtin_fmaf(b.y, v.y, b.x * v.x), re);
    im = __builtin_fmaf(sr, __builtin_fmaf(-b.y, v.x, b.x * v.y), im);
  }
}

template <int R, int LT, typename TOFF>
__global__ __launch_bounds__(1 << LT, (1024 >> LT)) void measure2_kernel(const v2f* __restrict__ state, long long state_stride,
                                                              const int* __restrict__ desc_g, double* __restrict__ eout,
                                                              long long eout_stride, int ecopies, long long ecopy_stride) {
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  // partial sums of the pass, numbered in program order (identical in every wave: descriptor-driven control flow)
  constexpr int EACC = 512;
  float* const eacc = lds + (1 << T);
  int* const eidx = reinterpret_cast<int*>(eacc + EACC);
  int eev = 0;

  const uint32_t tid = threadIdx.x;
  const bool lane0 = (tid & 63) == 0;
  state += (long long)blockIdx.y * state_stride;
  eout += (long long)blockIdx.y * eout_stride + (long long)(blockIdx.x % (unsigned)ecopies) * ecopy_stride;
  const KInt desc = (KInt)desc_g;
  for (int i = tid; i < EACC; i += (1 << LT)) eacc[i] = 0.f;
  __syncthreads();
#define TCMI_EFLUSH()                                  \
  {                                                    \
    __syncthreads();                                   \
    for (int i = tid; i < eev; i += (1 << LT)) {       \
      atomicAdd(eout + eidx[i], (double)eacc[i]);      \
      eacc[i] = 0.f;                                   \
    }                                                  \
    eev = 0;                                           \
    __syncthreads();                                   \
  }
#define TCMI_EADD(IDX, VAL)                               \
  {                                                       \
    if (eev == EACC) TCMI_EFLUSH() /* workgroup-uniform */  \
    if (lane0) {                                          \
      atomicAdd(eacc + eev, VAL);                         \
      eidx[eev] = IDX;                                    \
    }                                                     \
    ++eev;                                                \
  }

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
  {
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

#pragma unroll 1
  for (int k = 0;; ++k) {
    const KInt rr = desc + pc;
    const int nops = rr[0];
    int q = pc + TCMI_RR_WORDS;
    const uint32_t tidx = wg_base | tphys;
#pragma unroll 1
    for (int o = 0; o < nops; ++o) {
      // the host writes TCMI_OP_EXPECT2 ops only into these passes
      const int nX = desc[q + 1];
      const uint32_t gmask = (uint32_t)desc[q + 2];
      q += 3;
#pragma unroll 1
      for (int e = 0; e < nX; ++e, q += 4) {
        // the amplitudes do not change inside this loop, so the optimiser would hoist every string-independent pair
        // product of every case out of it (64 live values per case: hundreds of spills); an empty asm that "modifies"
        // the registers ends their invariance at no instruction cost
#pragma unroll
        for (int r = 0; r < NR; r += 8)
          asm volatile("" : "+v"(a[r]), "+v"(a[r + 1]), "+v"(a[r + 2]), "+v"(a[r + 3]), "+v"(a[r + 4]), "+v"(a[r + 5]),
                       "+v"(a[r + 6]), "+v"(a[r + 7]));
        const int xr = desc[q], oi = desc[q + 3];
        const uint32_t zr = (uint32_t)desc[q + 1], zm = (uint32_t)desc[q + 2];
        const bool neg = __popc(tidx & zm) & 1;
        const int fx = monehot(xr);
        if (zr == 0 && (xr & (xr - 1)) == 0) {
          // one X on register bit J, no register Z: 2 Re sum_pairs conj(a0) a1, imaginary part exactly zero
          v2f acc = {0.f, 0.f};
#define TCMI_X1(J)                                                            \
  if constexpr (R > J) {                                                      \
    if (fx & (1 << (1 << J))) {                                               \
      asm volatile("" ::: "memory");                                          \
      _Pragma("unroll") for (int g = 0; g < NR / 2; ++g) {                    \
        const int r0 = mins0(g, J);                                           \
        acc = __builtin_elementwise_fma(a[r0], a[r0 | (1 << J)], acc);        \
      }                                                                       \
    }                                                                         \
  }
          TCMI_X1(0) TCMI_X1(1) TCMI_X1(2) TCMI_X1(3) TCMI_X1(4)
#undef TCMI_X1
          float v = 2.f * (acc.x + acc.y);
          v = wave_sum_uniform(neg ? -v : v);
          TCMI_EADD(2 * oi, v)
        } else {
          float re = 0.f, im = 0.f;
#define TCMI_XP(XR)                                                 \
  if constexpr (XR < NR) {                                          \
    if (fx & (1u << XR)) {                                          \
      asm volatile("" ::: "memory"); /* keeps the 15 bodies from being if-converted and computed all at once */ \
      expect_pairs<NR, XR>(a, zr, re, im);                          \
    }                                                               \
  }
          TCMI_XP(1) TCMI_XP(2) TCMI_XP(4) TCMI_XP(8) TCMI_XP(16) TCMI_XP(3) TCMI_XP(5) TCMI_XP(6) TCMI_XP(9) TCMI_XP(10)
          TCMI_XP(12) TCMI_XP(17) TCMI_XP(18) TCMI_XP(20) TCMI_XP(24)
#undef TCMI_XP
          re = wave_sum_uniform(neg ? -re : re);
          im = wave_sum_uniform(neg ? -im : im);
          TCMI_EADD(2 * oi, re)
          TCMI_EADD(2 * oi + 1, im)
        }
      }
      if (gmask) {
        // w[k] = sum_r (-1)^{|r & k|} |a[r]|^2: the sum a Z-only string with register mask k needs
        float w[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) w[r] = __builtin_fmaf(a[r].x, a[r].x, a[r].y * a[r].y);
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (!((r >> j) & 1)) {
              const float lo = w[r], hi = w[r | (1 << j)];
              w[r] = lo + hi;
              w[r | (1 << j)] = lo - hi;
            }
#pragma unroll
        for (int kk = 0; kk < NR; ++kk) {
          if ((gmask >> kk) & 1u) {  // wave-uniform
            const int cnt = desc[q];
            ++q;
#pragma unroll 1
            for (int e = 0; e < cnt; ++e, q += 2) {
              const bool neg = __popc(tidx & (uint32_t)desc[q]) & 1;
              const float v = wave_sum_uniform(neg ? -w[kk] : w[kk]);
              TCMI_EADD(2 * desc[q + 1], v)
            }
          }
        }
      }
    }
    pc += TCMI_RR_WORDS + rr[1];
    if (k == nrounds - 1) break;

    {  // planar LDS exchange (re plane, then im plane), as in tcmi_vm2.hip
      const KInt rn = desc + pc;
      tphys = xor_masks<LT>(tid, rn + 8);
      const uint32_t wslot = xor_masks<LT>(tid, rr + 40) << 2, rslot = xor_masks<LT>(tid, rn + 24) << 2;
      char* const lb = reinterpret_cast<char*>(lds);
      uint32_t ad;
      uint32_t mv[R];
#define TCMI_MASKS(SRC, OFF) \
  _Pragma("unroll") for (int j = 0; j < R; ++j) mv[j] = mto_vgpr_v((uint32_t)SRC[OFF + j] << 2);
#define TCMI_WALK(BASE, STMT)                      \
  ad = BASE;                                       \
  _Pragma("unroll") for (int g = 0; g < NR; ++g) { \
    if (g) ad ^= mv[__builtin_ctz(g)];             \
    const int r = g ^ (g >> 1);                    \
    STMT;                                          \
  }
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
      __syncthreads();
#undef TCMI_MASKS
#undef TCMI_WALK
    }
  }
  TCMI_EFLUSH()
#undef TCMI_EADD
#undef TCMI_EFLUSH
}

template <int R, int LT>
static int launch_measure2(const void* state, long long state_stride, int batch, int n, const int* desc, double* eout,
                           long long eout_stride, int ecopies, long long ecopy_stride, hipStream_t st) {
  constexpr int T = R + LT;
  if (n < T) return -1;
  const size_t lds = (sizeof(float) << T) + 512 * (sizeof(float) + sizeof(int));
  auto kern = n <= 29 ? measure2_kernel<R, LT, uint32_t> : measure2_kernel<R, LT, unsigned long long>;
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(1u << LT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const v2f*>(state), state_stride, desc, eout, eout_stride,
                     ecopies < 1 ? 1 : ecopies, ecopy_stride);
  return hipGetLastError() == hipSuccess ? TCMI_OK : TCMI_ERR_HIP;
}

// complex64 measurement pass whose descriptors hold TCMI_OP_EXPECT2 ops: (R, LT) = (5, 8).  -1: no such variant.
int run_measure2_c64(const void* state, long long state_stride, int batch, int n, int R, int LT, const int* desc,
                     double* eout, long long eout_stride, int ecopies, long long ecopy_stride, hipStream_t st) {
  if (R == 5 && LT == 8)
    return launch_measure2<5, 8>(state, state_stride, batch, n, desc, eout, eout_stride, ecopies, ecopy_stride, st);
  return -1;
}

}  // namespace tcmi
