// tcmi: (sum_t w_t P_t) |psi> as tile passes (gfx950 / MI355X only).
//
// The cotangent of an energy L = sum_t w_t <psi|P_t|psi> is lambda = 2 sum_t w_t P_t |psi> -- the backward of
// Circuit.expectation (reference tensorcircuit/circuit.py:899-902 under backend.value_and_grad), and the same product
// is the matrix-free H|psi> of PauliStringSum2MVP-style Hamiltonians (tensorcircuit/quantum.py).  The flat kernel
// (tcmi_adjoint.hip, pauli_sum_kernel) stages 4096 consecutive amplitudes in LDS and gathers the partner of every X
// mask that leaves the tile from global memory: 17 tile reads per tile written for the 55-term TFIM at n = 28
// (profiles/r02o_vqe_n28_d12_pmc.txt: 7.6 x the algorithmic traffic, 0.10 of the HBM roofline).
//
// Here a pass owns tiles over ARBITRARY index bits (the 4 lowest, for 128-byte segments, plus a group of high bits chosen
// by the host): every term whose X mask lies inside the tile bits is applied from LDS, and a term is applied in exactly
// one pass; pass 0 writes lambda, later passes add to it.  TFIM at n = 28 with 12 tile bits: 3 passes = 3 reads of psi,
// 2 reads and 3 writes of lambda -- 8 state transfers instead of 18, all of them coalesced 16-byte accesses.
// The pass can also return Re <psi| (its part of lambda) >: summed over the passes that is sum_t w_t <P_t>, the energy
// itself, so a value_and_grad step needs no separate measurement passes.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tcmi_dev.h"

extern "C" int tcmi_set_error_(int code, const char* msg);

namespace tcmi {

// T tile bits, 256 threads, EPT = 2^T / 256 elements per thread, held as EPT / 2 neighbouring pairs (tile bit 0 is
// physical bit 0: one 16-byte global access per pair).  Host-side LDS index e = pair member | thread << 1 | k << 9; the
// kernel stores member-planar (see the staging loop).
// terms: int32 [nterms][4] = {X mask in LDS-index space, Z/Y sign mask (physical bits), number of Y | emask << 8,
// parity(X & Z)}; the first ndiag rows are the Z-only strings, sorted by emask, the rest is sorted by X mask.
template <typename F, int T>
__global__ __launch_bounds__(256) void pauli_tile_kernel(const typename Cx<F>::type* __restrict__ in,
                                                         typename Cx<F>::type* __restrict__ out, long long stride,
                                                         const int* __restrict__ tilepos_g, const int* __restrict__ terms,
                                                         int nterms, int ndiag, const double* __restrict__ w,
                                                         long long wstride, int accumulate, double* __restrict__ eout, long long estride,
                                                         int ecopies, int nt) {
  using C = typename Cx<F>::type;
  constexpr int NE = 1 << T, EPT = NE / 256, KB = T - 9;   // KB = index bits carried by the pair counter k
  static_assert(EPT >= 2 && (EPT & 1) == 0, "at least one pair per thread");
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  C* tile = reinterpret_cast<C*>(smem_);
  const KInt tp = (KInt)tilepos_g;
  const KInt tm = (KInt)terms;
  in += (long long)blockIdx.y * stride;
  out += (long long)blockIdx.y * stride;
  const KPtr<double> wk = (KPtr<double>)(w + (long long)blockIdx.y * wstride);
  // workgroup base: blockIdx.x spread over the non-tile bits (tile positions ascending)
  unsigned long long x = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < T; ++i) {
    const int p = tp[i];
    const unsigned long long low = (1ull << p) - 1ull;
    x = ((x & ~low) << 1) | (x & low);
  }
  const unsigned long long wg_base = x;
  const uint32_t tid = threadIdx.x;
  // physical offset of the thread bits (LDS index bits 1..8) and of every pair counter value (bits 9..T-1)
  unsigned long long tphys = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) tphys |= (unsigned long long)((tid >> b) & 1u) << tp[1 + b];
  unsigned long long kphys[EPT / 2];
#pragma unroll
  for (int k = 0; k < EPT / 2; ++k) {
    unsigned long long v = 0;
#pragma unroll
    for (int b = 0; b < KB; ++b)
      if ((k >> b) & 1) v |= 1ull << tp[9 + b];
    kphys[k] = v;
  }
  // stage the tile (registers are transient: the partner values and, at the end, the thread's own values come from LDS)
#pragma unroll
  for (int k = 0; k < EPT / 2; ++k) {
    const C* src = in + (wg_base | tphys | kphys[k]);
    // LDS layout: the two members of a pair live in two PLANES (index = thread | k << 8 | member << (T - 1)): the lanes of
    // a wave are then 8 (16) bytes apart in every partner read below.  With the members side by side the lanes sat 16
    // (32) bytes apart and every 8-byte read hit each bank twice -- the round-3 PMC run counted more bank-conflict cycles
    // than active LDS cycles in this kernel.
    if constexpr (sizeof(F) == 4) {
      // states that no cache holds until the next pass (nt: 2^26 amplitudes and up) are streamed with the nontemporal hint
      typedef float v4f_nt __attribute__((ext_vector_type(4)));
      const v4f_nt q = nt ? __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(src))
                          : *reinterpret_cast<const v4f_nt*>(src);
      C a0, a1;
      a0.x = q.x; a0.y = q.y; a1.x = q.z; a1.y = q.w;
      tile[tid + 256 * k] = a0;
      tile[tid + 256 * k + NE / 2] = a1;
    } else {
      tile[tid + 256 * k] = src[0];
      tile[tid + 256 * k + NE / 2] = src[1];
    }
  }
  __syncthreads();
  F re[EPT], im[EPT], dg[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    re[e] = 0;
    im[e] = 0;
    dg[e] = 0;
  }
  const uint32_t tlo = (uint32_t)tphys, blo = (uint32_t)wg_base;   // n <= 32: the masks are 32-bit
  // sign of a term at element e of this thread = s_thread * s_base * (-1)^{|e & emask|}: emask = the Z/Y mask restricted
  // to the element bits (pair member, pair counter), computed by the host (row word 2, bits 8..)
  int t = 0;
  // ---- Z-only strings (rows 0 .. ndiag-1, sorted by emask): dg[e] = sum_t c_t s_t(e).  One add per TERM into the
  // accumulator of its emask, then one Walsh-Hadamard transform over the element bits (the same regrouping as the
  // measurement kernel's): the 27 ZZ terms of the n = 28 TFIM cost ~200 instructions instead of 27 x 32
#pragma unroll
  for (int m = 0; m < EPT; ++m) {
#pragma unroll 1
    while (t < ndiag && ((tm[4 * t + 2] >> 8) & (EPT - 1)) == m) {
      const uint32_t zm = (uint32_t)tm[4 * t + 1];
      const F c0 = (F)wk[t];
      dg[m] += ((__popc(tlo & zm) + __popc(blo & zm)) & 1) ? -c0 : c0;
      ++t;
    }
  }
  if (ndiag > 0) {
#pragma unroll
    for (int j = 0; j < T - 8; ++j)
#pragma unroll
      for (int e = 0; e < EPT; ++e)
        if (!((e >> j) & 1)) {
          const F lo = dg[e], hi = dg[e | (1 << j)];
          dg[e] = lo + hi;
          dg[e | (1 << j)] = lo - hi;
        }
  }
  // ---- strings with X / Y factors, sorted by X mask: partner values from LDS
#pragma unroll 1
  for (; t < nterms; ++t) {
    const uint32_t xm0 = (uint32_t)tm[4 * t], zm = (uint32_t)tm[4 * t + 1];
    const uint32_t xm = ((xm0 & 1u) << (T - 1)) | (xm0 >> 1);     // the host's mask (member bit lowest) in plane layout
    const int w2 = tm[4 * t + 2], ny = w2 & 3, em = (w2 >> 8) & (EPT - 1), xpar = tm[4 * t + 3] & 1;
    const F c0 = (F)wk[t];
    const F ct = ((__popc(tlo & zm) + __popc(blo & zm) + xpar) & 1) ? -c0 : c0;
    if (em == 0 && ny == 0) {   // no sign inside the thread (every pure-X string): two FMAs per element
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const C v = tile[(tid + 256 * (e >> 1) + (e & 1) * (NE / 2)) ^ xm];
        re[e] = fma_<F>(ct, v.x, re[e]);
        im[e] = fma_<F>(ct, v.y, im[e]);
      }
      continue;
    }
    const bool flip = (ny == 2) || (ny == 3);          // i^2 = -1, i^3 = -i
    const bool rot = (ny & 1) != 0;                    // odd number of Y: multiply by i
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const C v = tile[(tid + 256 * (e >> 1) + (e & 1) * (NE / 2)) ^ xm];
      const bool neg = ((__popc((uint32_t)(e & em)) & 1) != 0) != flip;   // uniform
      const F c = neg ? -ct : ct;
      if (rot) {   // i v = (-v.y, v.x)
        re[e] = fma_<F>(-c, v.y, re[e]);
        im[e] = fma_<F>(c, v.x, im[e]);
      } else {
        re[e] = fma_<F>(c, v.x, re[e]);
        im[e] = fma_<F>(c, v.y, im[e]);
      }
    }
  }
  // diagonal part and Re <psi | this pass's part of lambda>
  double acc = 0;
#pragma unroll
  for (int k = 0; k < EPT / 2; ++k) {
    C o0, o1;
    o0 = tile[tid + 256 * k];
    o1 = tile[tid + 256 * k + NE / 2];
    re[2 * k] = fma_<F>(dg[2 * k], o0.x, re[2 * k]);
    im[2 * k] = fma_<F>(dg[2 * k], o0.y, im[2 * k]);
    re[2 * k + 1] = fma_<F>(dg[2 * k + 1], o1.x, re[2 * k + 1]);
    im[2 * k + 1] = fma_<F>(dg[2 * k + 1], o1.y, im[2 * k + 1]);
    acc += (double)(o0.x * re[2 * k] + o0.y * im[2 * k]) + (double)(o1.x * re[2 * k + 1] + o1.y * im[2 * k + 1]);
  }
  if (eout) {
    acc = wave_sum<double>(acc);
    if ((tid & 63) == 0) atomicAdd(eout + (long long)blockIdx.y * estride + (blockIdx.x % (unsigned)ecopies), acc);
  }
#pragma unroll
  for (int k = 0; k < EPT / 2; ++k) {
    C* dst = out + (wg_base | tphys | kphys[k]);
    if constexpr (sizeof(F) == 4) {
      float4 o = make_float4(re[2 * k], im[2 * k], re[2 * k + 1], im[2 * k + 1]);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dst);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      typedef float v4f_nt __attribute__((ext_vector_type(4)));
      const v4f_nt ov = {o.x, o.y, o.z, o.w};
      if (nt) __builtin_nontemporal_store(ov, reinterpret_cast<v4f_nt*>(dst));
      else *reinterpret_cast<v4f_nt*>(dst) = ov;
    } else {
      C o0, o1;
      o0.x = re[2 * k]; o0.y = im[2 * k]; o1.x = re[2 * k + 1]; o1.y = im[2 * k + 1];
      if (accumulate) {
        o0.x += dst[0].x; o0.y += dst[0].y; o1.x += dst[1].x; o1.y += dst[1].y;
      }
      dst[0] = o0;
      dst[1] = o1;
    }
  }
}

}  // namespace tcmi

extern "C" {

// 12 / 11 tile bits: 16 elements per thread, 32 KiB of LDS, four workgroups per CU (13 bits would allow 256-byte
// segments with the same three passes for the n = 28 TFIM, but 32 accumulator pairs per thread leave one wave per SIMD:
// 14.0 ms against 5.6 ms, scripts/gpu_pauli_tiled.py)
int tcmi_pauli_sum_tile_bits(int dtype) { return dtype == TCMI_C64 ? 12 : (dtype == TCMI_C128 ? 11 : -1); }

int tcmi_apply_pauli_sum_tiled(const void* in, void* out, long long state_stride, int batch, int n, const int* tilepos,
                               const int* terms, int nterms, int ndiag, const double* weights, long long weights_stride,
                               int accumulate, double* eout, long long eout_stride, int ecopies, int dtype,
                               void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int T = tcmi_pauli_sum_tile_bits(dtype);
  if (!in || !out || !tilepos || !terms || !weights || batch < 1 || nterms < 0 || ndiag < 0 || ndiag > nterms || T < 0 || n < T || n > 32 ||
      (eout && ecopies < 1) || in == out)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_apply_pauli_sum_tiled: bad argument");
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL((tcmi::pauli_tile_kernel<float, 12>), grid, block, sizeof(float2) << 12, st,
                       reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(out), state_stride, tilepos, terms,
                       nterms, ndiag, weights, weights_stride, accumulate, eout, eout_stride, ecopies, n >= 26 ? 1 : 0);
  else
    hipLaunchKernelGGL((tcmi::pauli_tile_kernel<double, 11>), grid, block, sizeof(double2) << 11, st,
                       reinterpret_cast<const double2*>(in), reinterpret_cast<double2*>(out), state_stride, tilepos,
                       terms, nterms, ndiag, weights, weights_stride, accumulate, eout, eout_stride, ecopies, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

}  // extern "C"
