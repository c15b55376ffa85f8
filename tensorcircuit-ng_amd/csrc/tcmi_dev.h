// Device-side helpers shared by the tile-VM kernels (gate passes, measurement passes, adjoint sweep).
#ifndef TCMI_DEV_H
#define TCMI_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tcmi_vm.h"

namespace tcmi {

template <typename F> struct Cx;
template <> struct Cx<float> { using type = float2; };
template <> struct Cx<double> { using type = double2; };

template <typename F> __device__ __forceinline__ void sincos_turns(F x, F* s, F* c);
template <> __device__ __forceinline__ void sincos_turns<float>(float x, float* s, float* c) {
#ifdef TCMI_PRECISE_SINCOS
  sincospif(2.0f * x, s, c);
#else
  // v_sin_f32 / v_cos_f32 take their argument in turns; reduce to [-0.5, 0.5] first
  x -= __builtin_rintf(x);
  *s = __builtin_amdgcn_sinf(x);
  *c = __builtin_amdgcn_cosf(x);
#endif
}
template <> __device__ __forceinline__ void sincos_turns<double>(double x, double* s, double* c) {
  sincospi(2.0 * x, s, c);
}

// One BK_PHASE table entry (both builder kernels): exp(2 pi i sum_t s_t(r) (k_t theta_t + o_t)) times the real scale
// terms c^(+-1) of rotations applied in two-shear form (TCMI_SCALE_TERM; the same sincos and threshold as the gate
// record, so the two always agree on the form).
template <typename F>
__device__ inline void phase_entry(const double* __restrict__ tp, int nterms, int r, const F* __restrict__ prow,
                                   bool invert, F* __restrict__ out) {
  double phi = 0.0, mag = 1.0;
  for (int t = 0; t < nterms; ++t) {
    const double th = (double)prow[(int)tp[4 * t + 2]];
    const double v = fma(tp[4 * t], th, tp[4 * t + 1]);  // explicit: the gate records form the angle the same way
    const unsigned rm = (unsigned)tp[4 * t + 3];
    const bool odd = __popc((unsigned)r & rm & (TCMI_SCALE_TERM - 1)) & 1;
    if (rm & TCMI_SCALE_TERM) {
      double s, c;
      sincos(v, &s, &c);
      if (fabs(c) >= TCMI_SHEAR2_CMIN) mag *= (odd != invert) ? 1.0 / c : c;
      continue;
    }
    phi += odd ? -v : v;
  }
  phi -= rint(phi);
  double s, c;
  sincospi(2.0 * phi, &s, &c);
  out[0] = (F)(c * mag);
  out[1] = (F)(s * mag);
}

// Tables and descriptors are read-only for the whole launch and every access is wave-uniform:
// read them through the constant address space so they become s_load (SGPR) operands.
#define TCMI_K __attribute__((address_space(4)))
template <typename F> using KPtr = const F TCMI_K*;
using KInt = const int TCMI_K*;

template <typename F>
__device__ __forceinline__ KPtr<F> tab_ptr(int slot, KPtr<F> ctab, KPtr<F> ptab) {
  return (slot & TCMI_CONST_FLAG) ? ctab + (slot & ~TCMI_CONST_FLAG) : ptab + slot;
}

// XOR of mask[i] over the set bits of v (wave-uniform masks, per-lane v)
template <int NB>
__device__ __forceinline__ uint32_t xor_masks(uint32_t v, KInt masks) {
  uint32_t out = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) out ^= (0u - ((v >> i) & 1u)) & (uint32_t)masks[i];
  return out;
}

// XOR of mask[j] over the set bits of a compile-time register index
template <int R>
__device__ __forceinline__ uint32_t reg_mask(int r, const uint32_t (&m)[R]) {
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < R; ++j)
    if ((r >> j) & 1) out ^= m[j];
  return out;
}

template <typename F> __device__ __forceinline__ F fma_(F a, F b, F c);
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// acc += m * v (complex), 4 FMAs
template <typename F, typename C>
__device__ __forceinline__ void cfma(F mr, F mi, const C& v, F& re, F& im) {
  re = fma_<F>(mr, v.x, re);
  re = fma_<F>(-mi, v.y, re);
  im = fma_<F>(mr, v.y, im);
  im = fma_<F>(mi, v.x, im);
}

// KIND 0: general complex 2x2 (16 FMA per pair)
// KIND 1: real matrix (h, ry, x, z ...)                         (8 per pair)
// KIND 2: real diagonal, imaginary off-diagonal (rx, y ...)     (8 per pair)
template <typename F, int NR, int J, int KIND>
__device__ __forceinline__ void apply_g1(typename Cx<F>::type (&a)[NR], const F (&m)[8]) {
  const F m00r = m[0], m00i = m[1], m01r = m[2], m01i = m[3];
  const F m10r = m[4], m10i = m[5], m11r = m[6], m11i = m[7];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    const auto x = a[r];
    const auto y = a[r1];
    if constexpr (KIND == 1) {
      a[r].x = fma_<F>(m01r, y.x, m00r * x.x);
      a[r].y = fma_<F>(m01r, y.y, m00r * x.y);
      a[r1].x = fma_<F>(m11r, y.x, m10r * x.x);
      a[r1].y = fma_<F>(m11r, y.y, m10r * x.y);
    } else if constexpr (KIND == 2) {
      a[r].x = fma_<F>(-m01i, y.y, m00r * x.x);
      a[r].y = fma_<F>(m01i, y.x, m00r * x.y);
      a[r1].x = fma_<F>(-m10i, x.y, m11r * y.x);
      a[r1].y = fma_<F>(m10i, x.x, m11r * y.y);
    } else {
      F re0 = m00r * x.x, im0 = m00r * x.y, re1 = m10r * x.x, im1 = m10r * x.y;
      re0 = fma_<F>(-m00i, x.y, re0);
      im0 = fma_<F>(m00i, x.x, im0);
      re1 = fma_<F>(-m10i, x.y, re1);
      im1 = fma_<F>(m10i, x.x, im1);
      cfma<F>(m01r, m01i, y, re0, im0);
      cfma<F>(m11r, m11i, y, re1, im1);
      a[r].x = re0; a[r].y = im0; a[r1].x = re1; a[r1].y = im1;
    }
  }
}

template <typename F, int NR, int JA, int JB>
__device__ __forceinline__ void apply_g2(typename Cx<F>::type (&a)[NR], const F (&m)[32]) {
  // matrix index = (bit JA << 1) | bit JB, JA < JB
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (((r >> JA) & 1) || ((r >> JB) & 1)) continue;
    const int i0 = r, i1 = r | (1 << JB), i2 = r | (1 << JA), i3 = r | (1 << JA) | (1 << JB);
    const typename Cx<F>::type v[4] = {a[i0], a[i1], a[i2], a[i3]};
    typename Cx<F>::type o[4];
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      F re = m[8 * row] * v[0].x, im = m[8 * row] * v[0].y;
      re = fma_<F>(-m[8 * row + 1], v[0].y, re);
      im = fma_<F>(m[8 * row + 1], v[0].x, im);
#pragma unroll
      for (int col = 1; col < 4; ++col) cfma<F>(m[2 * (4 * row + col)], m[2 * (4 * row + col) + 1], v[col], re, im);
      o[row].x = re;
      o[row].y = im;
    }
    a[i0] = o[0];
    a[i1] = o[1];
    a[i2] = o[2];
    a[i3] = o[3];
  }
}

// permutation gates: CNOT (either orientation) and SWAP move amplitudes, no arithmetic
template <typename F, int NR, int JA, int JB, int KIND>
__device__ __forceinline__ void apply_perm2(typename Cx<F>::type (&a)[NR]) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (((r >> JA) & 1) || ((r >> JB) & 1)) continue;
    const int i1 = r | (1 << JB), i2 = r | (1 << JA), i3 = r | (1 << JA) | (1 << JB);
    if constexpr (KIND == 1) { const auto t = a[i2]; a[i2] = a[i3]; a[i3] = t; }       // control JA
    else if constexpr (KIND == 2) { const auto t = a[i1]; a[i1] = a[i3]; a[i3] = t; }  // control JB
    else { const auto t = a[i1]; a[i1] = a[i2]; a[i2] = t; }                            // SWAP
  }
}

template <typename F>
__device__ __forceinline__ F wave_sum(F v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Wave-wide sum with a wave-uniform result.  float: DPP butterfly inside each row of 16 lanes, row_bcast15 /
// row_bcast31 across rows, v_readlane 63 — no LDS crossbar (ds_bpermute) traffic.  double: shuffles.
template <int CTRL, int RM>
__device__ __forceinline__ float dpp_add_c(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xf, false));
}
__device__ __forceinline__ float wave_sum_uniform(float v) {
  v = dpp_add_c<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add_c<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add_c<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add_c<0x140, 0xf>(v);  // row_mirror: every lane of a row holds the row sum
  v = dpp_add_c<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_add_c<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum_uniform(double v) { return wave_sum<double>(v); }

// Two / four wave-wide sums at once.  The first DPP steps fold the values into ONE register (lane % 2 resp. lane % 4
// selects the quantity), the rest of the butterfly keeps that residue (row_ror 4 / 8, then the gfx950 row swaps
// v_permlane16_swap / v_permlane32_swap), so four sums cost 21 VALU instructions instead of 4 x 13.
__device__ __forceinline__ float swap_add16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float fold2(float a, float b, bool odd) {  // even lanes: a pair sum, odd lanes: b pair sum
  const float keep = odd ? b : a, give = odd ? a : b;
  return keep + dpp_get<0xB1>(give);
}
__device__ __forceinline__ void wave_sum2_uniform(float& a, float& b, int lane) {
  float r = fold2(a, b, lane & 1);
  r = dpp_add_c<0x4E, 0xf>(r);   // quad_perm [2,3,0,1]: keeps the lane parity
  r = dpp_add_c<0x124, 0xf>(r);  // row_ror 4
  r = dpp_add_c<0x128, 0xf>(r);  // row_ror 8
  r = swap_add32(swap_add16(r));
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 0));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 1));
}
__device__ __forceinline__ void wave_sum4_uniform(float& a, float& b, float& c, float& d, int lane) {
  const float r0 = fold2(a, b, lane & 1), r1 = fold2(c, d, lane & 1);
  const bool hi = lane & 2;
  const float keep = hi ? r1 : r0, give = hi ? r0 : r1;
  float r = keep + dpp_get<0x4E>(give);  // lane % 4: 0 a, 1 b, 2 c, 3 d (sums over the quad)
  r = dpp_add_c<0x124, 0xf>(r);
  r = dpp_add_c<0x128, 0xf>(r);
  r = swap_add32(swap_add16(r));
  a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 0));
  b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 1));
  c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 2));
  d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 3));
}
// Eight wave-wide sums at once, NOT made uniform: afterwards every lane with lane % 8 == k holds the sum of v_k over the
// whole wave.  Three folding steps (strides 1, 2, 4; 3 instructions each per pair of quantities), then the rest of the
// butterfly on the one folded register: 29 VALU instructions for eight sums, no v_readlane / v_writelane.  The
// plan-specialised reverse sweep (tcmi/specialize.py) parks the result in the lanes of an accumulator register
// (park8) -- its gradient events are static, so lane e % 64 of accumulator e / 64 IS event e.
__device__ __forceinline__ float fold_q2(float a, float b, bool hi) {  // lane bit 1 clear: a, set: b (partner = lane ^ 2)
  const float keep = hi ? b : a, give = hi ? a : b;
  return keep + dpp_get<0x4E>(give);
}
__device__ __forceinline__ float fold_r4(float a, float b, bool hi) {  // lane bit 2 clear: a, set: b (partner: row_ror 4)
  const float keep = hi ? b : a, give = hi ? a : b;
  return keep + dpp_get<0x124>(give);
}
__device__ __forceinline__ float wave_fold8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                            int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  const float r01 = fold2(v0, v1, b0), r23 = fold2(v2, v3, b0), r45 = fold2(v4, v5, b0), r67 = fold2(v6, v7, b0);
  const float q0 = fold_q2(r01, r23, b1), q1 = fold_q2(r45, r67, b1);
  float r = fold_r4(q0, q1, b2);   // lane % 8 = k: partial sum of v_k over the lanes {l, l ^ 1, l ^ 2, l ^ 3, ...} it met
  r = dpp_add_c<0x128, 0xf>(r);    // row_ror 8: the other half of the row
  return swap_add32(swap_add16(r));
}
// acc[lane] = r[lane] for the lanes of group G (lane >> 3 == G), unchanged elsewhere: one DPP move
template <int G>
__device__ __forceinline__ int park8(int acc, float r) {
  return __builtin_amdgcn_update_dpp(acc, __float_as_int(r), 0xE4, 1 << (G >> 1), 3 << (2 * (G & 1)), false);
}
__device__ __forceinline__ void wave_sum2_uniform(double& a, double& b, int) {
  a = wave_sum<double>(a);
  b = wave_sum<double>(b);
}
__device__ __forceinline__ void wave_sum4_uniform(double& a, double& b, double& c, double& d, int) {
  a = wave_sum<double>(a);
  b = wave_sum<double>(b);
  c = wave_sum<double>(c);
  d = wave_sum<double>(d);
}

}  // namespace tcmi
#endif
