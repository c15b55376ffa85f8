// MPS / TEBD kernels for gfx950 (SURVEY.md §8a last row, K7): the decompositions behind
// reference tensorcircuit/mps_base.py:33-175 (FiniteMPS.apply_two_site_gate), mpscircuit.py:35-64
// (split_tensor) and the tensornetwork FiniteMPS.position sweeps.
//
//  * svd_block_kernel  — thin SVD of a row-major p x q matrix (p <= q) by one-sided (Hestenes) Jacobi
//    in *row* form: unitary rotations from the left make the rows of W = Y a mutually orthogonal,
//    W = Sigma Vh, U = Y^H.  Rows are contiguous (coalesced).  Blocked: a workgroup keeps two blocks of B
//    rows of [W | Y] in LDS, one wave per row pair, B pair-rounds per block pairing between
//    __syncthreads; the block pairings form a round-robin tournament whose rounds are separated by a
//    device-scope barrier between the (few) workgroups of that matrix, so a whole SVD is ONE launch.
//    Sorting, truncation count (reference jax_backend.py:62-112) and the optional absorption of S into
//    U or Vh are fused into the tail of the same kernel.
//  * qr_householder_kernel — Householder QR (complete isometry even for rank-deficient input, which
//    |0..0> product states are), one workgroup per matrix.
//  * mps_gate_mix_kernel — theta[l,a',b',r] = sum_ab G[a',b',a,b] T[l,a,b,r] (the 4x4 gate on the two
//    physical legs after the A.B bond GEMM).
#include <hip/hip_runtime.h>

#include "../../include/tcmi.h"
#include "tcmi_dev.h"

namespace tcmi {

// convergence: a pair is rotated while |<x,y>| > TCMI_SVD_TOL_SCALE * sqrt(q) * eps * |x| |y|
#ifndef TCMI_SVD_TOL_SCALE
#define TCMI_SVD_TOL_SCALE 1.0
#endif
constexpr int SVD_CTL_WORDS = 64;     // [0] barrier counter, [1] error flag, [2..] rotations per sweep
constexpr int SVD_MAX_SWEEPS = 60;
// Stopping rule: every pair whose cosine exceeds the tolerance is rotated, but a sweep only counts as "not converged"
// when one of its rotations had a cosine above SVD_STOP_SCALE x the tolerance.  A sweep of nothing but smaller
// rotations leaves second-order cosines behind (quadratic convergence), so the extra sweep that would only confirm
// "no rotation needed" (1 of 11-12 at 256 x 256) is not run.
#ifndef TCMI_SVD_STOP_SCALE
#define TCMI_SVD_STOP_SCALE 4.0
#endif
constexpr double SVD_STOP_SCALE = TCMI_SVD_STOP_SCALE;
constexpr unsigned SPIN_LIMIT = 1u << 21;

template <typename F>
struct Eps;
template <>
struct Eps<float> {
  static constexpr float v = 5.9604645e-8f;
  // |<x,y>|^2 below this is zero (keeps rsq / rcp away from denormals).  Just above the smallest normal number: with
  // 1e-30 here, rows with sigma < 3e-5 sigma_max (|x|^2 |y|^2 tol^2 < 1e-30) were never rotated against each other and
  // the vh rows of a spectrum graded over six decades came out orthogonal only to 7e-4 (now 1e-6, like the others).
  static constexpr float tiny = 1e-36f;
};
template <>
struct Eps<double> {
  static constexpr double v = 1.1102230246251565e-16;
  static constexpr double tiny = 1e-280;
};

// Inter-workgroup data (rows of W / Y, norms, sorted s) is exchanged INSIDE the launch.  Per-XCD L2s are
// not coherent and a CU's L1 is never refreshed by other CUs, so every shared word is written with
// write-through agent-scope stores (sc1) and read with agent-scope loads (sc1, L1 bypass) — the
// "8-byte agent atomics on both sides" form of the hand-off rules; no release/acquire fences (3.5 us
// each) are needed, the barrier only has to drain the stores of every wave before it arrives.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
#define TCMI_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ float2 ld_sc1(const float2* p) {
  const unsigned long long v = __hip_atomic_load((gu64*)p, TCMI_RLX);
  float2 r;
  r.x = __uint_as_float((unsigned)v);
  r.y = __uint_as_float((unsigned)(v >> 32));
  return r;
}
__device__ __forceinline__ void st_sc1(float2* p, float2 v) {
  __hip_atomic_store((gu64*)p, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), TCMI_RLX);
}
__device__ __forceinline__ double2 ld_sc1(const double2* p) {
  double2 r;
  r.x = __longlong_as_double((long long)__hip_atomic_load((gu64*)p, TCMI_RLX));
  r.y = __longlong_as_double((long long)__hip_atomic_load((gu64*)p + 1, TCMI_RLX));
  return r;
}
__device__ __forceinline__ void st_sc1(double2* p, double2 v) {
  __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v.x), TCMI_RLX);
  __hip_atomic_store((gu64*)p + 1, (unsigned long long)__double_as_longlong(v.y), TCMI_RLX);
}
__device__ __forceinline__ float ld_sc1(const float* p) { return __uint_as_float(__hip_atomic_load((gu32*)p, TCMI_RLX)); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store((gu32*)p, __float_as_uint(v), TCMI_RLX); }
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, TCMI_RLX));
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), TCMI_RLX);
}

// Barrier between the nwg workgroups of one matrix (all resident: the host sizes every launch by svd_resident_wgs).
// Every wave drains its write-through stores, one lane arrives on a monotonic counter and polls it
// relaxed with s_sleep; the spin is bounded (a timeout sets ctl[1] and ends the kernel).
__device__ __forceinline__ bool grid_barrier(unsigned* ctl, unsigned nwg, unsigned& epoch, unsigned* s_dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (nwg == 1) {
    __syncthreads();
    return true;
  }
  ++epoch;
  __syncthreads();
  if (threadIdx.x == 0) {
    gu32* c = (gu32*)ctl;
    __hip_atomic_fetch_add(c, 1u, TCMI_RLX);
    const unsigned target = epoch * nwg;
    unsigned spins = 0, dead = 0;
    while (__hip_atomic_load(c, TCMI_RLX) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || __hip_atomic_load(c + 1, TCMI_RLX)) {
        __hip_atomic_store(c + 1, 1u, TCMI_RLX);
        dead = 1;
        break;
      }
    }
    *s_dead = dead;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  return *s_dead == 0;
}

// Rotation annihilating <x, y> = gr + i gi between rows of squared norms al, be (all wave-uniform).
// c^2 + s^2 = 1 only to rounding; the accumulated drift is common to W and Y and is divided out at the
// end (sigma_i = |W_i| / |Y_i|).
template <typename F>
__device__ __forceinline__ void jacobi_rotation(F al, F be, F gr, F gi, F& c, F& sn, F& pr, F& pi) {
  const F ag = sqrt(gr * gr + gi * gi);
  const F zeta = (be - al) / (2 * ag);
  const F t = (zeta >= 0 ? (F)1 : (F)-1) / (fabs(zeta) + sqrt(1 + zeta * zeta));
  c = 1 / sqrt(1 + t * t);
  sn = c * t;
  const F ia = 1 / ag;
  pr = gr * ia;
  pi = gi * ia;
}
// complex64: the hardware reciprocal / rsqrt / sqrt approximations (1 ulp) instead of the IEEE
// division and square-root expansions — this scalar chain sits on the critical path of every pair
// round, and the 1e-7 loss of unitarity is what the final sigma = |W_i| / |Y_i| already divides out.
template <>
__device__ __forceinline__ void jacobi_rotation<float>(float al, float be, float gr, float gi, float& c, float& sn,
                                                       float& pr, float& pi) {
  const float ia = __builtin_amdgcn_rsqf(gr * gr + gi * gi);
  const float zeta = (be - al) * 0.5f * ia;
  const float az = fabsf(zeta);
  const float t = __builtin_copysignf(__builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(1.0f + zeta * zeta)), zeta);
  c = __builtin_amdgcn_rsqf(1.0f + t * t);
  sn = c * t;
  pr = gr * ia;
  pi = gi * ia;
}

// wave-wide sums (result uniform): wave_sum_uniform in tcmi_dev.h (DPP for float, shuffles for double)


// Row transfer between global memory (write-through / L1-bypassing, see above) and LDS.  Rows whose byte
// length is a multiple of 16 move as 16-byte units: global -> LDS by the DMA path (global_load_lds, no
// VGPR round trip), LDS -> global by ds_read_b128 + global_store_dwordx4 sc1.
typedef __attribute__((address_space(1))) const char gchar;
typedef __attribute__((address_space(3))) char lchar;

template <typename Ct>
__device__ __forceinline__ void row_in(Ct* dst, const Ct* src, int nelem, int lane) {
  const int nbytes = nelem * (int)sizeof(Ct);
  int done = 0;
  if ((nbytes & 15) == 0) {
    const char* g = reinterpret_cast<const char*>(src);
    char* l = reinterpret_cast<char*>(dst);
    for (; done + 1024 <= nbytes; done += 1024)
      __builtin_amdgcn_global_load_lds((gchar*)(g + done + lane * 16), (lchar*)(l + done), 16, 0, 16);
    if (done < nbytes) {
      if (lane * 16 < nbytes - done)
        __builtin_amdgcn_global_load_lds((gchar*)(g + done + lane * 16), (lchar*)(l + done), 16, 0, 16);
      done = nbytes;
    }
  }
  for (int c = done / (int)sizeof(Ct) + lane; c < nelem; c += 64) dst[c] = ld_sc1(src + c);
}

template <typename Ct>
__device__ __forceinline__ void row_out(Ct* dst, const Ct* src, int nelem, int lane) {
  const int nbytes = nelem * (int)sizeof(Ct);
  int done = 0;
  if ((nbytes & 15) == 0) {
    char* g = reinterpret_cast<char*>(dst);
    const char* l = reinterpret_cast<const char*>(src);
    for (; done < nbytes; done += 1024) {
      if (done + lane * 16 < nbytes) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = *reinterpret_cast<const v4f*>(l + done + lane * 16);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g + done + lane * 16), "v"(v) : "memory");
      }
    }
    done = nbytes;
  }
  for (int c = done / (int)sizeof(Ct) + lane; c < nelem; c += 64) st_sc1(dst + c, src[c]);
}

// One pair of LDS-resident rows [W row (q) | Y row (P2)], stride ld: measure <x, y> on the W part, rotate
// the whole row.  Returns 1 if a rotation was applied (wave-uniform).
template <typename F>
__device__ __forceinline__ int rotate_pair(typename Cx<F>::type* x, typename Cx<F>::type* y, int q, int ld, int lane,
                                           F tol2) {
  using Ct = typename Cx<F>::type;
  F al = 0, be = 0, gr = 0, gi = 0;
  for (int c = lane; c < q; c += 64) {
    const Ct xv = x[c], yv = y[c];
    al = fma_<F>(xv.x, xv.x, fma_<F>(xv.y, xv.y, al));
    be = fma_<F>(yv.x, yv.x, fma_<F>(yv.y, yv.y, be));
    gr = fma_<F>(xv.x, yv.x, fma_<F>(xv.y, yv.y, gr));   // gamma += x conj(y)
    gi = fma_<F>(xv.y, yv.x, fma_<F>(-xv.x, yv.y, gi));
  }
  al = wave_sum_uniform(al);
  be = wave_sum_uniform(be);
  gr = wave_sum_uniform(gr);
  gi = wave_sum_uniform(gi);
  const F g2 = gr * gr + gi * gi;
  if (!(g2 > tol2 * al * be && g2 > Eps<F>::tiny)) return 0;
  F c, sn, pr, pi;  // y~ = e^{i phi} y;  x' = c x - s y~;  y' = s x + c y~
  jacobi_rotation<F>(al, be, gr, gi, c, sn, pr, pi);
  for (int col = lane; col < ld; col += 64) {
    const Ct xv = x[col], yv = y[col];
    const F tr = pr * yv.x - pi * yv.y, ti = pr * yv.y + pi * yv.x;
    Ct nx, ny;
    nx.x = c * xv.x - sn * tr;
    nx.y = c * xv.y - sn * ti;
    ny.x = sn * xv.x + c * tr;
    ny.y = sn * xv.y + c * ti;
    x[col] = nx;
    y[col] = ny;
  }
  return g2 > (F)(SVD_STOP_SCALE * SVD_STOP_SCALE) * tol2 * al * be ? 1 : 0;
}

// Cross rounds with the wave's own row held in registers (row length ld <= 64 * E): the partner row is
// read from LDS once and written once per round, the own row only returns to LDS after the last round —
// a third of the LDS traffic of rotate_pair.
template <typename F, int E>
__device__ __forceinline__ int rotate_pair_reg(typename Cx<F>::type (&x)[E], typename Cx<F>::type* yrow, int q, int ld,
                                               int lane, F tol2) {
  using Ct = typename Cx<F>::type;
  Ct y[E];
  F al = 0, be = 0, gr = 0, gi = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = lane + 64 * e;
    if (c < ld) {
      y[e] = yrow[c];
    } else {
      y[e].x = 0;
      y[e].y = 0;
    }
    if (c < q) {
      al = fma_<F>(x[e].x, x[e].x, fma_<F>(x[e].y, x[e].y, al));
      be = fma_<F>(y[e].x, y[e].x, fma_<F>(y[e].y, y[e].y, be));
      gr = fma_<F>(x[e].x, y[e].x, fma_<F>(x[e].y, y[e].y, gr));
      gi = fma_<F>(x[e].y, y[e].x, fma_<F>(-x[e].x, y[e].y, gi));
    }
  }
  al = wave_sum_uniform(al);
  be = wave_sum_uniform(be);
  gr = wave_sum_uniform(gr);
  gi = wave_sum_uniform(gi);
  const F g2 = gr * gr + gi * gi;
  if (!(g2 > tol2 * al * be && g2 > Eps<F>::tiny)) return 0;
  F c, sn, pr, pi;
  jacobi_rotation<F>(al, be, gr, gi, c, sn, pr, pi);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int col = lane + 64 * e;
    if (col < ld) {
      const F tr = pr * y[e].x - pi * y[e].y, ti = pr * y[e].y + pi * y[e].x;
      Ct nx, ny;
      nx.x = c * x[e].x - sn * tr;
      nx.y = c * x[e].y - sn * ti;
      ny.x = sn * x[e].x + c * tr;
      ny.y = sn * x[e].y + c * ti;
      x[e] = nx;
      yrow[col] = ny;
    }
  }
  return g2 > (F)(SVD_STOP_SCALE * SVD_STOP_SCALE) * tol2 * al * be ? 1 : 0;
}

// The same rounds for rows of exactly 64 * E elements of which the first 64 * EQ are the W part:
// * no bounds tests, so the E partner loads are all in flight at once (with the tests each ds_read waited for the one
//   before: 8 serial LDS round trips per round);
// * the squared row norms are measured once per chip-wide round and then follow the rotations analytically
//   (|x'|^2 = |x|^2 - t |gamma|, |y'|^2 = |y|^2 + t |gamma|, as LAPACK's xGESVJ does between its refreshes): a round
//   only measures gamma, two fused wave sums instead of four;
// the rounds of a workgroup are VALU-issue bound (8 waves x ~230 instructions on 4 SIMDs), so instructions are time.
template <typename F>
__device__ __forceinline__ void jacobi_rotation_t(F al, F be, F gr, F gi, F& c, F& sn, F& pr, F& pi, F& tg) {
  jacobi_rotation<F>(al, be, gr, gi, c, sn, pr, pi);
  tg = (sn / c) * sqrt(gr * gr + gi * gi);
}
template <>
__device__ __forceinline__ void jacobi_rotation_t<float>(float al, float be, float gr, float gi, float& c, float& sn,
                                                         float& pr, float& pi, float& tg) {
  // half-angle form: two dependent transcendentals on the critical path (rsq, rsq) instead of four (rsq, sqrt, rcp, rsq)
  const float g2 = gr * gr + gi * gi;
  const float ia = __builtin_amdgcn_rsqf(g2);
  const float d = 0.5f * (be - al), ad = fabsf(d);
  const float h2 = fmaf(d, d, g2);                 // h^2 = d^2 + |gamma|^2
  const float ih = __builtin_amdgcn_rsqf(h2);
  const float u = fmaf(0.5f * ad, ih, 0.5f);       // c^2 = (h + |d|) / 2h
  const float ic = __builtin_amdgcn_rsqf(u);
  c = u * ic;
  sn = __builtin_copysignf(0.5f * (g2 * ia) * ih * ic, d);
  pr = gr * ia;
  pi = gi * ia;
  tg = __builtin_copysignf(g2 * __builtin_amdgcn_rcpf(ad + h2 * ih), d);  // t |gamma| = |gamma|^2 / (|d| + h)
}

#ifdef TCMI_SVD_TIMING
#define TCMI_TI(k)                                                 \
  {                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    tin[k] += now_ - tin[7];                                       \
    tin[7] = now_;                                                 \
  }
#define TCMI_TI_ARG , unsigned long long* tin
#define TCMI_TI_PASS , tin
#else
#define TCMI_TI(k)
#define TCMI_TI_ARG
#define TCMI_TI_PASS
#endif
template <typename F, int E, int EQ>
__device__ __forceinline__ int rotate_pair_reg_exact(typename Cx<F>::type (&x)[E], typename Cx<F>::type* yrow, F& al,
                                                     F* be_slot, int lane, F tol2 TCMI_TI_ARG) {
  using Ct = typename Cx<F>::type;
  Ct y[E];
  TCMI_TI(6)
#pragma unroll
  for (int e = 0; e < E; ++e) y[e] = yrow[lane + 64 * e];
  const F be = *be_slot;
  TCMI_TI(0)
  F gr = 0, gi = 0;
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    gr = fma_<F>(x[e].x, y[e].x, fma_<F>(x[e].y, y[e].y, gr));
    gi = fma_<F>(x[e].y, y[e].x, fma_<F>(-x[e].x, y[e].y, gi));
  }
  TCMI_TI(1)
  wave_sum2_uniform(gr, gi, lane);
  const F g2 = gr * gr + gi * gi;
  TCMI_TI(2)
  if (!(g2 > tol2 * al * be && g2 > Eps<F>::tiny)) return 0;
  F c, sn, pr, pi, tg;
  const int big = g2 > (F)(SVD_STOP_SCALE * SVD_STOP_SCALE) * tol2 * al * be ? 1 : 0;
  jacobi_rotation_t<F>(al, be, gr, gi, c, sn, pr, pi, tg);
#ifdef TCMI_SVD_TIMING
  asm volatile("" : "+v"(c), "+v"(sn), "+v"(pr), "+v"(pi));
#endif
  TCMI_TI(3)
  al = al - tg > 0 ? al - tg : 0;
  if (lane == 0) *be_slot = be + tg > 0 ? be + tg : 0;
  if constexpr (sizeof(F) == 4) {  // packed f32: (re, im) pairs, 6 v_pk instructions per element instead of 12
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 c2 = {c, c}, s2 = {sn, sn}, pr2 = {pr, pr}, pi2 = {-pi, pi};
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const f2 yv = {y[e].x, y[e].y}, ys = {y[e].y, y[e].x}, xv = {x[e].x, x[e].y};
      const f2 t = __builtin_elementwise_fma(pr2, yv, pi2 * ys);
      const f2 nx = __builtin_elementwise_fma(c2, xv, -(s2 * t));
      const f2 ny = __builtin_elementwise_fma(s2, xv, c2 * t);
      x[e].x = nx.x;
      x[e].y = nx.y;
      Ct o;
      o.x = ny.x;
      o.y = ny.y;
      yrow[lane + 64 * e] = o;
    }
    TCMI_TI(4)
    return big;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const F tr = pr * y[e].x - pi * y[e].y, ti = pr * y[e].y + pi * y[e].x;
    Ct nx, ny;
    nx.x = c * x[e].x - sn * tr;
    nx.y = c * x[e].y - sn * ti;
    ny.x = sn * x[e].x + c * tr;
    ny.y = sn * x[e].y + c * ti;
    x[e] = nx;
    yrow[lane + 64 * e] = ny;
  }
  return big;
}

template <typename F, int B, int E, int EQ>
__device__ __forceinline__ int cross_rounds_exact(typename Cx<F>::type* L, F* nrm, int wave, int lane,
                                                  F tol2 TCMI_TI_ARG) {
  using Ct = typename Cx<F>::type;
  constexpr int ld = 64 * E;
  Ct* xrow = L + wave * ld;
  const Ct* prow = L + (B + wave) * ld;
  Ct xr[E];
  int rot = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) xr[e] = xrow[lane + 64 * e];
  F al = 0, b0 = 0;  // squared norms of the own row and of partner row `wave`
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    const Ct pv = prow[lane + 64 * e];
    al = fma_<F>(xr[e].x, xr[e].x, fma_<F>(xr[e].y, xr[e].y, al));
    b0 = fma_<F>(pv.x, pv.x, fma_<F>(pv.y, pv.y, b0));
  }
  wave_sum2_uniform(al, b0, lane);
  if (lane == 0) nrm[wave] = b0;
  __syncthreads();
  for (int k = 0; k < B; ++k) {
    const int j = (wave + k) % B;
    rot += rotate_pair_reg_exact<F, E, EQ>(xr, L + (B + j) * ld, al, nrm + j, lane, tol2 TCMI_TI_PASS);
    __syncthreads();
    TCMI_TI(5)
  }
#pragma unroll
  for (int e = 0; e < E; ++e) xrow[lane + 64 * e] = xr[e];
  __syncthreads();
  return rot;
}

// Blocked one-sided Jacobi.  The P2 (padded) rows are cut into NB = P2 / B blocks of B rows; workgroup g
// holds TWO blocks (2B rows of [W | Y]) in LDS and owns B waves.  A sweep is a round-robin tournament
// of the blocks (NB - 1 global rounds); inside a round the B x B cross pairs are done in B LDS rounds
// separated only by __syncthreads (round 0 also does the intra-block pairs), then the rows go back to
// global memory (write-through) and the workgroups of the matrix meet at one grid barrier.  p - 1 pair
// rounds per sweep as in the flat scheme, but only NB - 1 of them cross the chip.
// A barrier that timed out (workgroups of the matrix not co-resident: shared or partitioned GPU) must not leave
// plausible-looking factors behind: every workgroup that notices poisons the outputs callers look at -- the kept rank
// becomes -1 and every singular value NaN -- before it leaves.
template <typename F>
__device__ __forceinline__ void svd_poison(F* s, int p, int* keep_out, F* tw2_out, int bi) {
  const F nan = __builtin_nanf("");
  for (int k = threadIdx.x; k < p; k += blockDim.x) s[k] = nan;
  if (threadIdx.x == 0) {
    if (keep_out) keep_out[bi] = -1;
    if (tw2_out) tw2_out[bi] = nan;
  }
}
// Probe builds (-DTCMI_SVD_TIMING, scripts/gpu_svd_phases.py): workgroup 0 accumulates the shader-clock time of each
// phase of a chip-wide round into control words 48..53 (row_in, intra rounds, cross rounds, row_out, barrier, total).
#ifdef TCMI_SVD_TIMING
#define TCMI_T(k)                                                                  \
  {                                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
    tacc[k] += now_ - tlast;                                                       \
    tlast = now_;                                                                  \
  }
#else
#define TCMI_T(k)
#endif
#define TCMI_SVD_SYNC()                                    \
  if (!grid_barrier(ctl, nwg, epoch, &s_dead)) {           \
    svd_poison<F>(s, p, keep_out, tw2_out, b + batch0);    \
    return;                                                \
  }

template <typename F, int B>
__global__ __launch_bounds__(64 * B) void svd_block_kernel(
    const typename Cx<F>::type* __restrict__ a, long long a_stride, typename Cx<F>::type* __restrict__ u, F* s,
    typename Cx<F>::type* __restrict__ vh, int* __restrict__ keep_out, F* __restrict__ tw2_out, int p, int q,
    int kmax, int P2, typename Cx<F>::type* work, long long work_stride, unsigned* ctl_base, int max_sweeps,
    int max_sv, F max_err, int relative, int absorb, int batch0) {
  using Ct = typename Cx<F>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Ct* L = reinterpret_cast<Ct*>(smem_raw);  // no static LDS in this kernel: the dynamic base stays 16-B aligned
  constexpr int T = 64 * B;
  const int b = blockIdx.y;
  const int ld = q + P2;
  const int NB = P2 / B;
  const unsigned nwg = gridDim.x;  // NB / 2
  const int g = blockIdx.x;
  a += (long long)(b + batch0) * a_stride;
  u += (long long)(b + batch0) * p * kmax;
  s += (long long)(b + batch0) * p;
  vh += (long long)(b + batch0) * (long long)kmax * q;
  Ct* W = work + (long long)b * work_stride;
  Ct* Y = W + (long long)P2 * q;
  F* sq = reinterpret_cast<F*>(Y + (long long)P2 * P2);  // [P2] squared singular values, [P2] 1/|Y_i|
  F* yn = sq + P2;
  unsigned* ctl = ctl_base + (long long)b * SVD_CTL_WORDS;
  unsigned epoch = 0;
  unsigned& s_dead = *reinterpret_cast<unsigned*>(smem_raw + 2ll * B * ld * (long long)sizeof(Ct));
  F* nrm = reinterpret_cast<F*>(smem_raw + 2ll * B * ld * (long long)sizeof(Ct) + 16);  // [B] tracked row norms

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  {  // W = a (zero pad rows), Y = I
    const long long t0 = (long long)g * T + threadIdx.x, step = (long long)nwg * T;
    const long long nW = (long long)P2 * q, nA = (long long)p * q, nY = (long long)P2 * P2;
    Ct zero;
    zero.x = 0;
    zero.y = 0;
    for (long long i = t0; i < nW; i += step) {
      Ct v = zero;
      if (i < nA) v = a[i];
      st_sc1(W + i, v);
    }
    for (long long i = t0; i < nY; i += step) {
      Ct v = zero;
      if (i / P2 == i % P2) v.x = 1;
      st_sc1(Y + i, v);
    }
  }
  TCMI_SVD_SYNC();

  const F tol2 = Eps<F>::v * Eps<F>::v * (F)q * (F)(TCMI_SVD_TOL_SCALE * TCMI_SVD_TOL_SCALE);
  const int M = NB - 1;
#ifdef TCMI_SVD_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  unsigned long long tin[8] = {0, 0, 0, 0, 0, 0, 0, tlast};  // inside a cross round: load, gram, sum, rotation, apply, barrier
  const unsigned long long tstart = tlast;
#endif
  if (max_sweeps > SVD_MAX_SWEEPS) max_sweeps = SVD_MAX_SWEEPS;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    int rot = 0;
    for (int R = 0; R < M; ++R) {
      const int bA = g == 0 ? R : (R + g) % M;
      const int bB = g == 0 ? M : (R - g + M) % M;
      // rows of the two blocks -> LDS (wave w carries local rows w and B + w)
      for (int h = 0; h < 2; ++h) {
        const int lr = h * B + wave;
        const long long gr = (long long)(h == 0 ? bA : bB) * B + wave;
        Ct* dst = L + (long long)lr * ld;
        row_in<Ct>(dst, W + gr * q, q, lane);
        row_in<Ct>(dst + q, Y + gr * P2, P2, lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      TCMI_T(0)
      if (R == 0 && B > 1) {  // intra-block pairs, both blocks at once (B/2 waves each)
        constexpr int H = B > 1 ? B / 2 : 1;
        const int blk = wave / H, idx = wave % H, Mb = B - 1;
        for (int k = 0; k < Mb; ++k) {
          const int i = idx == 0 ? k : (k + idx) % Mb;
          const int j = idx == 0 ? Mb : (k - idx + Mb) % Mb;
          rot += rotate_pair<F>(L + (long long)(blk * B + i) * ld, L + (long long)(blk * B + j) * ld, q, ld, lane, tol2);
          __syncthreads();
        }
      }
      TCMI_T(1)
      if (q == P2 && (q == 256 || q == 128 || q == 64)) {  // square, rows of whole 64-element chunks
        if (q == 256) rot += cross_rounds_exact<F, B, 8, 4>(L, nrm, wave, lane, tol2 TCMI_TI_PASS);
        else if (q == 128) rot += cross_rounds_exact<F, B, 4, 2>(L, nrm, wave, lane, tol2 TCMI_TI_PASS);
        else rot += cross_rounds_exact<F, B, 2, 1>(L, nrm, wave, lane, tol2 TCMI_TI_PASS);
      } else if (ld <= 512) {  // own row in registers for the B cross rounds
        constexpr int E = 8;
        Ct* xrow = L + (long long)wave * ld;
        Ct xr[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int c = lane + 64 * e;
          if (c < ld) {
            xr[e] = xrow[c];
          } else {
            xr[e].x = 0;
            xr[e].y = 0;
          }
        }
        for (int k = 0; k < B; ++k) {
          rot += rotate_pair_reg<F, E>(xr, L + (long long)(B + (wave + k) % B) * ld, q, ld, lane, tol2);
          __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int c = lane + 64 * e;
          if (c < ld) xrow[c] = xr[e];
        }
        __syncthreads();
      } else {
        for (int k = 0; k < B; ++k) {
          rot += rotate_pair<F>(L + (long long)wave * ld, L + (long long)(B + (wave + k) % B) * ld, q, ld, lane, tol2);
          __syncthreads();
        }
      }
      TCMI_T(2)
      for (int h = 0; h < 2; ++h) {
        const int lr = h * B + wave;
        const long long gr = (long long)(h == 0 ? bA : bB) * B + wave;
        const Ct* src = L + (long long)lr * ld;
        row_out<Ct>(W + gr * q, src, q, lane);
        row_out<Ct>(Y + gr * P2, src + q, P2, lane);
      }
      if (R == M - 1 && rot > 0 && lane == 0) __hip_atomic_fetch_add((gu32*)&ctl[2 + sweep], 1u, TCMI_RLX);
      TCMI_T(3)
      TCMI_SVD_SYNC();
      TCMI_T(4)
    }
    if (__hip_atomic_load((gu32*)&ctl[2 + sweep], TCMI_RLX) == 0) break;
  }

#ifdef TCMI_SVD_TIMING
  if (g == 0 && lane == 0 && wave < 8) ctl[40 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  if (g == 0 && threadIdx.x == 0) {
    tacc[5] = __builtin_amdgcn_s_memtime() - tstart;
    for (int k = 0; k < 6; ++k) ctl[48 + k] = (unsigned)(tacc[k] >> 4);
    for (int k = 0; k < 7; ++k) ctl[54 + k] = (unsigned)(tin[k] >> 4);
  }
#endif
  const int gw = g * B + wave;  // P2 / 2 waves in total, two rows each
  // squared row norms of W and Y: sigma_i = |W_i| / |Y_i| (W = Y a holds to rounding whatever the
  // accumulated non-unitarity of the rotations; dividing it out removes the common drift)
  for (int k = 0; k < 2; ++k) {
    const int row = 2 * gw + k;
    const Ct* w = W + (long long)row * q;
    const Ct* yr = Y + (long long)row * P2;
    F acc = 0, accy = 0;
    for (int c = lane; c < q; c += 64) {
      const Ct v = ld_sc1(w + c);
      acc = fma_<F>(v.x, v.x, fma_<F>(v.y, v.y, acc));
    }
    for (int c = lane; c < P2; c += 64) {
      const Ct v = ld_sc1(yr + c);
      accy = fma_<F>(v.x, v.x, fma_<F>(v.y, v.y, accy));
    }
    acc = wave_sum<F>(acc);
    accy = wave_sum<F>(accy);
    if (lane == 0) {
      st_sc1(sq + row, acc / accy);
      st_sc1(yn + row, 1 / sqrt(accy));
    }
  }
  TCMI_SVD_SYNC();

  // rank (descending, ties by row index; the zero pad rows sort last), then write s / vh / u sorted
  for (int k = 0; k < 2; ++k) {
    const int row = 2 * gw + k;
    if (row >= p) continue;
    const F v = ld_sc1(sq + row);
    const F iy = ld_sc1(yn + row);
    int cnt = 0;
    for (int j = lane; j < P2; j += 64) {
      const F o = ld_sc1(sq + j);
      cnt += (o > v || (o == v && j < row)) ? 1 : 0;
    }
    const int rank = wave_sum<int>(cnt);
    const F sig = sqrt(v);
    if (lane == 0) st_sc1(s + rank, sig);
    if (rank < kmax) {
      // W_row = |Y_row| sigma vh: vh = W_row iy / sigma
      const F inv = sig > 0 ? iy / sig : 0;
      const F sv = absorb == 2 ? iy : inv;
      const F su = (absorb == 1 ? sig : (F)1) * iy;
      const Ct* w = W + (long long)row * q;
      Ct* o = vh + (long long)rank * q;
      for (int c = lane; c < q; c += 64) {
        Ct t = ld_sc1(w + c);
        t.x *= sv;
        t.y *= sv;
        o[c] = t;
      }
      const Ct* yr = Y + (long long)row * P2;
      for (int c = lane; c < p; c += 64) {
        Ct t = ld_sc1(yr + c);
        t.x *= su;
        t.y *= -su;
        u[(long long)c * kmax + rank] = t;
      }
    }
  }
  TCMI_SVD_SYNC();

  if (g == 0 && threadIdx.x == 0) {
    int keep = (max_sv > 0 && max_sv < p) ? max_sv : p;
    if (max_err >= 0) {
      const F abs_err = relative ? max_err * ld_sc1(s) : max_err;
      F acc = 0;
      int nerr = 0;
      for (int k = p - 1; k >= 0; --k) {
        const F sk = ld_sc1(s + k);
        acc += sk * sk;
        if (sqrt(acc) > abs_err) ++nerr;
      }
      if (nerr < keep) keep = nerr;
    }
    F tw2 = 0;
    for (int k = p - 1; k >= keep; --k) {
      const F sk = ld_sc1(s + k);
      tw2 += sk * sk;
    }
    // a workgroup that timed out at an EARLIER barrier has poisoned s / keep / tw2 and left; this workgroup may still
    // have seen the last barrier complete -- do not overwrite the poison with valid-looking numbers
    if (nwg > 1 && __hip_atomic_load((gu32*)ctl + 1, TCMI_RLX)) {
      keep = -1;
      tw2 = __builtin_nanf("");
    }
    if (keep_out) keep_out[b + batch0] = keep;
    if (tw2_out) tw2_out[b + batch0] = tw2;
  }
}

// ---------------------------------------------------------------------------------------------- QR
constexpr int QR_TX = 64, QR_TY = 16, QR_THREADS = QR_TX * QR_TY;

template <typename F>
__device__ __forceinline__ F block_sum(F v, F* red) {
  v = wave_sum<F>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  F t = 0;
#pragma unroll
  for (int w = 0; w < QR_THREADS / 64; ++w) t += red[w];
  return t;
}

// X[k:, c_begin:c_end] -= u (2/un2) (u^H X[k:, c_begin:c_end]),  u = Aw[k:, k]
template <typename F>
__device__ __forceinline__ void reflect(typename Cx<F>::type* X, int ldx, const typename Cx<F>::type* Aw, int lda,
                                        int k, int m, int c_begin, int c_end, F scale,
                                        typename Cx<F>::type (*wbuf)[QR_TX]) {
  using Ct = typename Cx<F>::type;
  const int tx = threadIdx.x & (QR_TX - 1), ty = threadIdx.x / QR_TX;
  for (int c0 = c_begin; c0 < c_end; c0 += QR_TX) {
    const int j = c0 + tx;
    F ar = 0, ai = 0;
    if (j < c_end)
      for (int i = k + ty; i < m; i += QR_TY) {
        const Ct uu = Aw[(long long)i * lda + k], xx = X[(long long)i * ldx + j];
        // conj(u) * x
        ar = fma_<F>(uu.x, xx.x, fma_<F>(uu.y, xx.y, ar));
        ai = fma_<F>(uu.x, xx.y, fma_<F>(-uu.y, xx.x, ai));
      }
    wbuf[ty][tx].x = ar;
    wbuf[ty][tx].y = ai;
    __syncthreads();
    if (ty == 0) {
      F sr = 0, si = 0;
#pragma unroll
      for (int t = 0; t < QR_TY; ++t) {
        sr += wbuf[t][tx].x;
        si += wbuf[t][tx].y;
      }
      wbuf[0][tx].x = sr * scale;
      wbuf[0][tx].y = si * scale;
    }
    __syncthreads();
    const Ct w = wbuf[0][tx];
    if (j < c_end)
      for (int i = k + ty; i < m; i += QR_TY) {
        const Ct uu = Aw[(long long)i * lda + k];
        Ct xx = X[(long long)i * ldx + j];
        xx.x -= uu.x * w.x - uu.y * w.y;
        xx.y -= uu.x * w.y + uu.y * w.x;
        X[(long long)i * ldx + j] = xx;
      }
    __syncthreads();
  }
}

template <typename F>
__global__ __launch_bounds__(QR_THREADS) void qr_householder_kernel(const typename Cx<F>::type* __restrict__ a,
                                                                    typename Cx<F>::type* __restrict__ qout,
                                                                    typename Cx<F>::type* __restrict__ rout, int m,
                                                                    int n, typename Cx<F>::type* work,
                                                                    long long work_stride) {
  using Ct = typename Cx<F>::type;
  const int K = m < n ? m : n;
  const int b = blockIdx.x;
  a += (long long)b * m * n;
  qout += (long long)b * m * K;
  rout += (long long)b * K * n;
  Ct* Aw = work + (long long)b * work_stride;      // [m][n]
  Ct* diag = Aw + (long long)m * n;                // [K]
  F* un2 = reinterpret_cast<F*>(diag + K);         // [K]
  __shared__ F red[QR_THREADS / 64];
  __shared__ Ct wbuf[QR_TY][QR_TX];
  __shared__ F s_scale;
  const int tid = threadIdx.x;
  for (long long i = tid; i < (long long)m * n; i += QR_THREADS) Aw[i] = a[i];
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    F part = 0;
    for (int i = k + tid; i < m; i += QR_THREADS) {
      const Ct v = Aw[(long long)i * n + k];
      part = fma_<F>(v.x, v.x, fma_<F>(v.y, v.y, part));
    }
    const F nx2 = block_sum<F>(part, red);
    if (tid == 0) {
      const Ct alpha = Aw[(long long)k * n + k];
      const F nx = sqrt(nx2);
      Ct d;
      d.x = d.y = 0;
      F u2 = 0;
      if (nx > 0) {
        const F aa = sqrt(alpha.x * alpha.x + alpha.y * alpha.y);
        const F pr = aa > 0 ? alpha.x / aa : (F)1, pi = aa > 0 ? alpha.y / aa : (F)0;
        d.x = -pr * nx;
        d.y = -pi * nx;
        Ct u0;
        u0.x = alpha.x - d.x;
        u0.y = alpha.y - d.y;
        Aw[(long long)k * n + k] = u0;
        u2 = 2 * nx * (nx + aa);
      }
      diag[k] = d;
      un2[k] = u2;
      s_scale = u2 > 0 ? 2 / u2 : 0;
    }
    __syncthreads();
    const F scale = s_scale;
    if (scale > 0) reflect<F>(Aw, n, Aw, n, k, m, k + 1, n, scale, wbuf);
    __syncthreads();
  }
  // R
  for (long long i = tid; i < (long long)K * n; i += QR_THREADS) {
    const int r = (int)(i / n), c = (int)(i % n);
    Ct v;
    v.x = v.y = 0;
    if (c == r)
      v = diag[r];
    else if (c > r)
      v = Aw[(long long)r * n + c];
    rout[i] = v;
  }
  // Q = H_0 ... H_{K-1} [I]
  for (long long i = tid; i < (long long)m * K; i += QR_THREADS) {
    Ct v;
    v.x = (i / K == i % K) ? 1 : 0;
    v.y = 0;
    qout[i] = v;
  }
  __syncthreads();
  for (int k = K - 1; k >= 0; --k) {
    const F u2 = un2[k];
    if (u2 > 0) reflect<F>(qout, K, Aw, n, k, m, k, K, 2 / u2, wbuf);
    __syncthreads();
  }
}

// Register-resident Householder QR for m <= 256, n <= 64 * C (C = 2 for complex64, 1 for complex128):
// thread (ty = tid / 64, tx = tid % 64) owns rows ty + 16 t (t < 16) of columns tx + 64 c.  The matrix never
// leaves the register file during the factorisation: per step only the reflector goes through LDS
// (norm -> u -> per-column dot -> update, 4 __syncthreads), no global traffic inside the loop.  The
// reflectors are parked in the workspace and re-read (one 2 KB column per step) to form Q the same way.
template <typename F, int C>
__global__ __launch_bounds__(QR_THREADS) void qr_reg_kernel(const typename Cx<F>::type* __restrict__ a,
                                                            typename Cx<F>::type* __restrict__ qout,
                                                            typename Cx<F>::type* __restrict__ rout, int m, int n,
                                                            typename Cx<F>::type* work, long long work_stride) {
  using Ct = typename Cx<F>::type;
  constexpr int TR = 16;  // rows per thread
  const int K = m < n ? m : n;
  const int b = blockIdx.x;
  a += (long long)b * m * n;
  qout += (long long)b * m * K;
  rout += (long long)b * K * n;
  Ct* U = work + (long long)b * work_stride;  // [K][m] reflectors (row k = u_k, zero above k)
  __shared__ Ct ubuf[256];
  __shared__ Ct wbuf[QR_TY][QR_TX * C];
  __shared__ F red[QR_TY];
  __shared__ F s_scale[256];
  __shared__ Ct s_diag[256];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  Ct A[TR][C];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = ty + 16 * t, j = tx + 64 * c;
      Ct v;
      v.x = v.y = 0;
      if (i < m && j < n) v = a[(long long)i * n + j];
      A[t][c] = v;
    }
  for (int k = 0; k < K; ++k) {
    const int kc = k >> 6, kx = k & 63;
    // column k below the diagonal -> ubuf, partial norms
    if (tx == kx) {
      F part = 0;
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (c == kc) {
            const int i = ty + 16 * t;
            Ct v = A[t][c];
            if (i < k || i >= m) v.x = v.y = 0;
            if (i < 256) ubuf[i] = v;
            part = fma_<F>(v.x, v.x, fma_<F>(v.y, v.y, part));
          }
      red[ty] = part;
    }
    __syncthreads();
    if (tid == 0) {
      F nx2 = 0;
#pragma unroll
      for (int w = 0; w < QR_TY; ++w) nx2 += red[w];
      const Ct alpha = ubuf[k];
      const F nx = sqrt(nx2);
      Ct d;
      d.x = d.y = 0;
      F sc = 0;
      if (nx > 0) {
        const F aa = sqrt(alpha.x * alpha.x + alpha.y * alpha.y);
        const F pr = aa > 0 ? alpha.x / aa : (F)1, pi = aa > 0 ? alpha.y / aa : (F)0;
        d.x = -pr * nx;
        d.y = -pi * nx;
        Ct u0;
        u0.x = alpha.x - d.x;
        u0.y = alpha.y - d.y;
        ubuf[k] = u0;
        sc = 1 / (nx * (nx + aa));  // 2 / |u|^2
      }
      s_diag[k] = d;
      s_scale[k] = sc;
    }
    __syncthreads();
    const F scale = s_scale[k];
    // park the reflector (coalesced row of U)
    if (tid < m) U[(long long)k * m + tid] = ubuf[tid];
    if (scale > 0) {
      // w_j = scale * sum_i conj(u_i) A[i][j] for the columns j > k this thread touches
#pragma unroll
      for (int c = 0; c < C; ++c) {
        F ar = 0, ai = 0;
#pragma unroll
        for (int t = 0; t < TR; ++t) {
          const int i = ty + 16 * t;
          if (i >= k && i < m) {
            const Ct uu = ubuf[i], xx = A[t][c];
            ar = fma_<F>(uu.x, xx.x, fma_<F>(uu.y, xx.y, ar));
            ai = fma_<F>(uu.x, xx.y, fma_<F>(-uu.y, xx.x, ai));
          }
        }
        wbuf[ty][tx + 64 * c].x = ar;
        wbuf[ty][tx + 64 * c].y = ai;
      }
      __syncthreads();
      if (ty < C) {  // wave ty reduces column chunk ty
        F sr = 0, si = 0;
#pragma unroll
        for (int t = 0; t < QR_TY; ++t) {
          sr += wbuf[t][tx + 64 * ty].x;
          si += wbuf[t][tx + 64 * ty].y;
        }
        wbuf[0][tx + 64 * ty].x = sr * scale;
        wbuf[0][tx + 64 * ty].y = si * scale;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = tx + 64 * c;
        if (j > k) {
          const Ct w = wbuf[0][j];
#pragma unroll
          for (int t = 0; t < TR; ++t) {
            const int i = ty + 16 * t;
            if (i >= k && i < m) {
              const Ct uu = ubuf[i];
              A[t][c].x -= uu.x * w.x - uu.y * w.y;
              A[t][c].y -= uu.x * w.y + uu.y * w.x;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  // R: upper triangle from the registers, diagonal from s_diag
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = ty + 16 * t, j = tx + 64 * c;
      if (i < K && j < n) {
        Ct v;
        v.x = v.y = 0;
        if (j == i)
          v = s_diag[i];
        else if (j > i)
          v = A[t][c];
        rout[(long long)i * n + j] = v;
      }
    }
  // Q = H_0 ... H_{K-1} [I] in the same register layout (columns < K)
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = ty + 16 * t, j = tx + 64 * c;
      A[t][c].x = (i == j && j < K) ? 1 : 0;
      A[t][c].y = 0;
    }
  __syncthreads();
  for (int k = K - 1; k >= 0; --k) {
    const F scale = s_scale[k];
    if (scale > 0) {
      if (tid < m) ubuf[tid] = U[(long long)k * m + tid];
      __syncthreads();
#pragma unroll
      for (int c = 0; c < C; ++c) {
        F ar = 0, ai = 0;
#pragma unroll
        for (int t = 0; t < TR; ++t) {
          const int i = ty + 16 * t;
          if (i >= k && i < m) {
            const Ct uu = ubuf[i], xx = A[t][c];
            ar = fma_<F>(uu.x, xx.x, fma_<F>(uu.y, xx.y, ar));
            ai = fma_<F>(uu.x, xx.y, fma_<F>(-uu.y, xx.x, ai));
          }
        }
        wbuf[ty][tx + 64 * c].x = ar;
        wbuf[ty][tx + 64 * c].y = ai;
      }
      __syncthreads();
      if (ty < C) {
        F sr = 0, si = 0;
#pragma unroll
        for (int t = 0; t < QR_TY; ++t) {
          sr += wbuf[t][tx + 64 * ty].x;
          si += wbuf[t][tx + 64 * ty].y;
        }
        wbuf[0][tx + 64 * ty].x = sr * scale;
        wbuf[0][tx + 64 * ty].y = si * scale;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = tx + 64 * c;
        if (j >= k && j < K) {
          const Ct w = wbuf[0][j];
#pragma unroll
          for (int t = 0; t < TR; ++t) {
            const int i = ty + 16 * t;
            if (i >= k && i < m) {
              const Ct uu = ubuf[i];
              A[t][c].x -= uu.x * w.x - uu.y * w.y;
              A[t][c].y -= uu.x * w.y + uu.y * w.x;
            }
          }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = ty + 16 * t, j = tx + 64 * c;
      if (i < m && j < K) qout[(long long)i * K + j] = A[t][c];
    }
}

// theta[l,a',b',r] = sum_{a,b} G[a',b',a,b] T[l,a,b,r]
template <typename F>
__global__ void mps_gate_mix_kernel(const typename Cx<F>::type* __restrict__ T,
                                    const typename Cx<F>::type* __restrict__ G,
                                    typename Cx<F>::type* __restrict__ out, int L, int R, long long batch_stride,
                                    long long gate_stride) {
  using Ct = typename Cx<F>::type;
  const long long b = blockIdx.y;
  T += b * batch_stride;
  out += b * batch_stride;
  G += b * gate_stride;
  __shared__ Ct g[16];
  if (threadIdx.x < 16) g[threadIdx.x] = G[threadIdx.x];
  __syncthreads();
  const long long total = (long long)L * R;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long l = t / R, r = t - l * R;
    Ct in[4];
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) in[ab] = T[(l * 4 + ab) * R + r];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      F re = 0, im = 0;
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const Ct c = g[o * 4 + ab];
        re = fma_<F>(c.x, in[ab].x, fma_<F>(-c.y, in[ab].y, re));
        im = fma_<F>(c.x, in[ab].y, fma_<F>(c.y, in[ab].x, im));
      }
      Ct v;
      v.x = re;
      v.y = im;
      out[(l * 4 + o) * R + r] = v;
    }
  }
}

// Block geometry: B rows per block (= waves per workgroup), P2 = p rounded up to a multiple of 2B.
// 2B rows of [W | Y] must fit the 64 KiB of LDS a workgroup gets without opting in to more.
struct SvdGeom {
  int B, P2, wgs;
  long long lds_bytes, work_elems;
};

template <typename F>
static SvdGeom svd_geom(int p, int q) {
  using Ct = typename Cx<F>::type;
  int B = p >= 16 ? 8 : p >= 8 ? 4 : p >= 4 ? 2 : 1;
  SvdGeom gm;
  for (;; B /= 2) {
    const int P2 = (p + 2 * B - 1) / (2 * B) * (2 * B);
    gm.B = B;
    gm.P2 = P2;
    gm.lds_bytes = 2ll * B * (q + P2) * (long long)sizeof(Ct);
    if (gm.lds_bytes <= 65536 || B == 1) break;
  }
  gm.wgs = gm.P2 / gm.B / 2;
  const long long f_as_c = ((long long)gm.P2 * sizeof(F) + sizeof(Ct) - 1) / sizeof(Ct);
  gm.work_elems = (long long)gm.P2 * q + (long long)gm.P2 * gm.P2 + 2 * f_as_c + 8;
  return gm;
}

template <typename F, int B>
static void launch_svd(const SvdGeom& gm, int nb, hipStream_t st, const void* a, long long a_stride, void* u, void* s,
                       void* vh, int* keep, void* tw2, int p, int q, int kmax, void* work, unsigned* ctl,
                       int max_sweeps, int max_sv, double max_err, int relative, int absorb, int batch0) {
  using Ct = typename Cx<F>::type;
  hipLaunchKernelGGL((svd_block_kernel<F, B>), dim3(gm.wgs, nb, 1), dim3(64 * B), (size_t)gm.lds_bytes + 16 + 128, st,
                     reinterpret_cast<const Ct*>(a), a_stride, reinterpret_cast<Ct*>(u), reinterpret_cast<F*>(s),
                     reinterpret_cast<Ct*>(vh), keep, reinterpret_cast<F*>(tw2), p, q, kmax, gm.P2,
                     reinterpret_cast<Ct*>(work), gm.work_elems, ctl, max_sweeps, max_sv, (F)max_err, relative,
                     absorb, batch0);
}

template <typename F>
static int dispatch_svd(const SvdGeom& gm, int nb, hipStream_t st, const void* a, long long a_stride, void* u,
                        void* s, void* vh, int* keep, void* tw2, int p, int q, int kmax, void* work, unsigned* ctl,
                        int max_sweeps, int max_sv, double max_err, int relative, int absorb, int batch0) {
#define TCMI_SVD_CASE(BB)                                                                                         \
  case BB:                                                                                                        \
    launch_svd<F, BB>(gm, nb, st, a, a_stride, u, s, vh, keep, tw2, p, q, kmax, work, ctl, max_sweeps, max_sv,    \
                      max_err, relative, absorb, batch0);                                                         \
    return 0;
  switch (gm.B) {
    TCMI_SVD_CASE(1)
    TCMI_SVD_CASE(2)
    TCMI_SVD_CASE(4)
    TCMI_SVD_CASE(8)
  }
#undef TCMI_SVD_CASE
  return -1;
}

// Workgroups of the SVD kernel the device keeps resident at once: the spin barrier needs ALL workgroups of a matrix
// resident.  Bounded by LDS (160 KiB per CU), by the 32-wave limit of a CU and by the register file (the kernel is
// built for at most 128 VGPRs: 4 waves per SIMD), times the CU count; one workgroup per CU is taken off as a margin
// when more than one fits (the occupancy figures can be one high, MI355X_MICROARCH.md "Residency").
static int svd_resident_wgs(const SvdGeom& gm, bool f64) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      v = 256;
    cus = v;
  }
  long long per_cu = (160ll * 1024) / (gm.lds_bytes + 16 + 255);
  const long long by_waves = (f64 ? 12 : 16) / gm.B;   // waves per SIMD by registers (<= 96 / <= 152 VGPRs) x 4 SIMDs
  if (by_waves < per_cu) per_cu = by_waves;
  if (per_cu > 1) per_cu -= 1;
  if (per_cu < 1) per_cu = 1;
  return (int)(per_cu * cus);
}

// batch elements per launch so that every workgroup of the launch is resident; 0: one matrix alone does not fit
static int svd_chunk(const SvdGeom& gm, int batch, bool f64) {
  int chunk = svd_resident_wgs(gm, f64) / gm.wgs;
  if (chunk > batch) chunk = batch;
  return chunk;
}

}  // namespace tcmi

extern "C" int tcmi_set_error_(int code, const char* msg);

extern "C" {

long long tcmi_svd_work_bytes(int m, int n, int batch, int dtype) {
  if (m < 1 || n < m || batch < 1) return -1;
  if (dtype == TCMI_C64) {
    const tcmi::SvdGeom gm = tcmi::svd_geom<float>(m, n);
    const int chunk = tcmi::svd_chunk(gm, batch, false) > 0 ? tcmi::svd_chunk(gm, batch, false) : 1;
    return (long long)chunk * tcmi::SVD_CTL_WORDS * 4 + chunk * gm.work_elems * 8;
  }
  if (dtype == TCMI_C128) {
    const tcmi::SvdGeom gm = tcmi::svd_geom<double>(m, n);
    const int chunk = tcmi::svd_chunk(gm, batch, true) > 0 ? tcmi::svd_chunk(gm, batch, true) : 1;
    return (long long)chunk * tcmi::SVD_CTL_WORDS * 4 + chunk * gm.work_elems * 16;
  }
  return -1;
}

int tcmi_svd_trunc_batched(const void* a, void* u, void* s, void* vh, int* keep_out, void* tw2_out, int m, int n,
                           int kmax, int batch, int max_singular_values, double max_truncation_err, int relative,
                           int absorb, int max_sweeps, void* work, long long work_bytes, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !u || !s || !vh || !work || m < 1 || n < m || kmax < 1 || kmax > m || batch < 1 || absorb < 0 ||
      absorb > 2)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: bad argument (needs m <= n, 1 <= kmax <= m)");
  if (dtype != TCMI_C64 && dtype != TCMI_C128)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: bad dtype");
  const long long need = tcmi_svd_work_bytes(m, n, batch, dtype);
  if (need < 0 || work_bytes < need) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: workspace too small");
  if (max_sweeps <= 0) max_sweeps = 30;
  const tcmi::SvdGeom gm = dtype == TCMI_C64 ? tcmi::svd_geom<float>(m, n) : tcmi::svd_geom<double>(m, n);
  if (gm.lds_bytes > 65536)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: matrix too wide for the LDS-resident kernel");
  const int chunk = tcmi::svd_chunk(gm, batch, dtype == TCMI_C128);
  if (chunk < 1)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: the matrix needs more co-resident workgroups than the "
                                         "device holds (one-launch Jacobi SVD; split the problem)");
  unsigned* ctl = reinterpret_cast<unsigned*>(work);
  char* wbase = reinterpret_cast<char*>(work) + (long long)chunk * tcmi::SVD_CTL_WORDS * 4;
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int nb = batch - b0 < chunk ? batch - b0 : chunk;
    hipError_t e = hipMemsetAsync(ctl, 0, (size_t)chunk * tcmi::SVD_CTL_WORDS * 4, st);
    if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
    int rc;
    if (dtype == TCMI_C64)
      rc = tcmi::dispatch_svd<float>(gm, nb, st, a, (long long)m * n, u, s, vh, keep_out, tw2_out, m, n, kmax, wbase,
                                     ctl, max_sweeps, max_singular_values, max_truncation_err, relative, absorb, b0);
    else
      rc = tcmi::dispatch_svd<double>(gm, nb, st, a, (long long)m * n, u, s, vh, keep_out, tw2_out, m, n, kmax, wbase,
                                      ctl, max_sweeps, max_singular_values, max_truncation_err, relative, absorb, b0);
    if (rc != 0) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_svd_trunc_batched: unsupported size");
    e = hipGetLastError();
    if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  }
  return TCMI_OK;
}

long long tcmi_qr_work_bytes(int m, int n, int batch, int dtype) {
  if (m < 1 || n < 1 || batch < 1) return -1;
  const long long K = m < n ? m : n;
  const long long elems = (long long)m * n + 2 * K + 8;  // >= K * m, the reflector store of the register kernel
  if (dtype == TCMI_C64) return batch * elems * 8;
  if (dtype == TCMI_C128) return batch * elems * 16;
  return -1;
}

int tcmi_qr_batched(const void* a, void* q, void* r, int m, int n, int batch, void* work, long long work_bytes,
                    int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !q || !r || !work || m < 1 || n < 1 || batch < 1)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_qr_batched: bad argument");
  const long long need = tcmi_qr_work_bytes(m, n, batch, dtype);
  if (need < 0 || work_bytes < need) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_qr_batched: workspace too small");
  const long long K = m < n ? m : n;
  const long long stride = (long long)m * n + 2 * K + 8;
  if (dtype == TCMI_C64 && m <= 256 && n <= 128)
    hipLaunchKernelGGL((tcmi::qr_reg_kernel<float, 2>), dim3(batch), dim3(tcmi::QR_THREADS), 0, st,
                       reinterpret_cast<const float2*>(a), reinterpret_cast<float2*>(q),
                       reinterpret_cast<float2*>(r), m, n, reinterpret_cast<float2*>(work), stride);
  else if (dtype == TCMI_C128 && m <= 256 && n <= 64)
    hipLaunchKernelGGL((tcmi::qr_reg_kernel<double, 1>), dim3(batch), dim3(tcmi::QR_THREADS), 0, st,
                       reinterpret_cast<const double2*>(a), reinterpret_cast<double2*>(q),
                       reinterpret_cast<double2*>(r), m, n, reinterpret_cast<double2*>(work), stride);
  else if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::qr_householder_kernel<float>, dim3(batch), dim3(tcmi::QR_THREADS), 0, st,
                       reinterpret_cast<const float2*>(a), reinterpret_cast<float2*>(q),
                       reinterpret_cast<float2*>(r), m, n, reinterpret_cast<float2*>(work), stride);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::qr_householder_kernel<double>, dim3(batch), dim3(tcmi::QR_THREADS), 0, st,
                       reinterpret_cast<const double2*>(a), reinterpret_cast<double2*>(q),
                       reinterpret_cast<double2*>(r), m, n, reinterpret_cast<double2*>(work), stride);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_qr_batched: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_mps_gate_mix(const void* t, const void* gate, void* out, int L, int R, int batch, long long gate_stride,
                      int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!t || !gate || !out || L < 1 || R < 1 || batch < 1 || batch > 65535)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_mps_gate_mix: bad argument");
  const long long total = (long long)L * R;
  unsigned gx = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  dim3 grid(gx, batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::mps_gate_mix_kernel<float>, grid, block, 0, st, reinterpret_cast<const float2*>(t),
                       reinterpret_cast<const float2*>(gate), reinterpret_cast<float2*>(out), L, R, total * 4,
                       gate_stride);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::mps_gate_mix_kernel<double>, grid, block, 0, st, reinterpret_cast<const double2*>(t),
                       reinterpret_cast<const double2*>(gate), reinterpret_cast<double2*>(out), L, R, total * 4,
                       gate_stride);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_mps_gate_mix: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

}  // extern "C"
