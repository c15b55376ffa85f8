// tcmi tile-VM: the state-vector pass kernel for gfx950 (MI355X).
//
// One launch = one "pass" of the compiled plan (tcmi/plan.py): every workgroup loads a tile of
// 2^T amplitudes straight into registers (2^R per thread, 16-byte coalesced accesses), runs the
// pass program -- dense 1-/2-qubit gates on register bits, diagonal phase polynomials evaluated
// from the global index, LDS exchanges that re-map which tile bits are register bits -- and
// writes the tile back in place.  It replaces the reference's per-gate
// tn.contract_between -> backend.tensordot chain and the final reorder_edges transpose
// (reference tensorcircuit/cons.py:937-960, tensorcircuit/circuit.py:701-721).
//
// Written for CDNA4 only: wave64, 160 KiB LDS (64 KiB per 8192-amplitude tile -> 2 workgroups
// per CU), scalar (SGPR) gate coefficients, bank-conflict-free LDS exchange maps chosen on the
// host per exchange.  Descriptor layout: see tcmi_vm.h / tcmi/plan.py.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "tcmi_vm.h"
#include "tcmi_dev.h"

namespace tcmi {

template <typename F, int NR, int R, int J>
__device__ __forceinline__ void apply_g1_kind(typename Cx<F>::type (&a)[NR], int kind, const F (&m)[8]) {
  if (kind == 1) apply_g1<F, NR, J, 1>(a, m);
  else if (kind == 2) apply_g1<F, NR, J, 2>(a, m);
  else apply_g1<F, NR, J, 0>(a, m);
}

template <typename F, int NR, int R>
__device__ __forceinline__ void dispatch_g1m(typename Cx<F>::type (&a)[NR], int j, int kind, const F (&mm)[R][8]) {
  switch (j) {
    case 0: apply_g1_kind<F, NR, R, 0>(a, kind, mm[0]); break;
    case 1: if constexpr (R > 1) apply_g1_kind<F, NR, R, 1>(a, kind, mm[1]); break;
    case 2: if constexpr (R > 2) apply_g1_kind<F, NR, R, 2>(a, kind, mm[2]); break;
    case 3: if constexpr (R > 3) apply_g1_kind<F, NR, R, 3>(a, kind, mm[3]); break;
    case 4: if constexpr (R > 4) apply_g1_kind<F, NR, R, 4>(a, kind, mm[4]); break;
    case 5: if constexpr (R > 5) apply_g1_kind<F, NR, R, 5>(a, kind, mm[5]); break;
    default: break;
  }
}

// a[r] *= cs + i * z_J(r) * ys   (z_J = +1 / -1 for bit J of r clear / set)
template <typename F, int NR, int J>
__device__ __forceinline__ void apply_diagb(typename Cx<F>::type (&a)[NR], F cs, F ys) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const auto v = a[r];
    if ((r >> J) & 1) {
      a[r].x = fma_<F>(v.y, ys, v.x * cs);
      a[r].y = fma_<F>(-v.x, ys, v.y * cs);
    } else {
      a[r].x = fma_<F>(-v.y, ys, v.x * cs);
      a[r].y = fma_<F>(v.x, ys, v.y * cs);
    }
  }
}

template <typename F, int NR, int R>
__device__ __forceinline__ void dispatch_diagb(typename Cx<F>::type (&a)[NR], int j, F cs, F ys) {
  switch (j) {
    case 0: apply_diagb<F, NR, 0>(a, cs, ys); break;
    case 1: if constexpr (R > 1) apply_diagb<F, NR, 1>(a, cs, ys); break;
    case 2: if constexpr (R > 2) apply_diagb<F, NR, 2>(a, cs, ys); break;
    case 3: if constexpr (R > 3) apply_diagb<F, NR, 3>(a, cs, ys); break;
    case 4: if constexpr (R > 4) apply_diagb<F, NR, 4>(a, cs, ys); break;
    case 5: if constexpr (R > 5) apply_diagb<F, NR, 5>(a, cs, ys); break;
    default: break;
  }
}

#define TCMI_G2_CASE(A, B)                                         \
  case (A * 8 + B):                                                \
    if constexpr (R > B) {                                         \
      if (kind == 0) apply_g2<F, NR, A, B>(a, m);                  \
      else if (kind == 1) apply_perm2<F, NR, A, B, 1>(a);          \
      else if (kind == 2) apply_perm2<F, NR, A, B, 2>(a);          \
      else apply_perm2<F, NR, A, B, 3>(a);                         \
    }                                                              \
    break;

template <typename F, int NR, int R>
__device__ __forceinline__ void dispatch_g2(typename Cx<F>::type (&a)[NR], int jak, int jb, const F (&m)[32]) {
  const int ja = jak & 0xff, kind = jak >> 8;
  switch (ja * 8 + jb) {
    TCMI_G2_CASE(0, 1) TCMI_G2_CASE(0, 2) TCMI_G2_CASE(0, 3) TCMI_G2_CASE(0, 4) TCMI_G2_CASE(0, 5)
    TCMI_G2_CASE(1, 2) TCMI_G2_CASE(1, 3) TCMI_G2_CASE(1, 4) TCMI_G2_CASE(1, 5)
    TCMI_G2_CASE(2, 3) TCMI_G2_CASE(2, 4) TCMI_G2_CASE(2, 5)
    TCMI_G2_CASE(3, 4) TCMI_G2_CASE(3, 5)
    TCMI_G2_CASE(4, 5)
    default: break;
  }
}

// ---- fused Pauli-sum expectation (K4) ------------------------------------------------------------
// sum_r sgn(r & zr) * conj(a[r ^ XR]) * a[r]   (XR = compile-time register-bit mask of the X/Y bits)
template <typename F, int NR, int XR>
__device__ __forceinline__ void expect_x(const typename Cx<F>::type (&a)[NR], uint32_t zr, F& re, F& im) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const auto b = a[r ^ XR];
    const auto v = a[r];
    const F sr = (__popc((uint32_t)r & zr) & 1) ? (F)-1 : (F)1;
    const F tr = fma_<F>(b.y, v.y, b.x * v.x);
    const F ti = fma_<F>(-b.y, v.x, b.x * v.y);
    re = fma_<F>(sr, tr, re);
    im = fma_<F>(sr, ti, im);
  }
}

#define TCMI_EX_CASE(X) \
  case X:               \
    if constexpr (X < NR) expect_x<F, NR, X>(a, zr, re, im); \
    break;

template <typename F, int NR>
__device__ __forceinline__ void dispatch_expect_x(const typename Cx<F>::type (&a)[NR], int xr, uint32_t zr, F& re, F& im) {
  switch (xr) {  // one or two X/Y bits among the register bits
    TCMI_EX_CASE(1) TCMI_EX_CASE(2) TCMI_EX_CASE(4) TCMI_EX_CASE(8) TCMI_EX_CASE(16) TCMI_EX_CASE(32)
    TCMI_EX_CASE(3) TCMI_EX_CASE(5) TCMI_EX_CASE(6) TCMI_EX_CASE(9) TCMI_EX_CASE(10) TCMI_EX_CASE(12)
    TCMI_EX_CASE(17) TCMI_EX_CASE(18) TCMI_EX_CASE(20) TCMI_EX_CASE(24)
    TCMI_EX_CASE(33) TCMI_EX_CASE(34) TCMI_EX_CASE(36) TCMI_EX_CASE(40) TCMI_EX_CASE(48)
    default: break;
  }
}

// MODE 0: gate passes (G1M / G2 / DIAG).  MODE 1: measurement passes (EXPECT only).  Separate
// instantiations keep each kernel's control-flow graph (and register allocation) small.
template <typename F, int R, int LT, int MODE>
__global__ __launch_bounds__(1 << LT, (MODE == 1 && LT == 8) ? 2 : 1) void pass_kernel(typename Cx<F>::type* __restrict__ state,
                                                        long long state_stride,
                                                        const int* __restrict__ desc_g,
                                                        const F* __restrict__ ctab_g,
                                                        const F* __restrict__ ptab_g,
                                                        long long ptab_stride,
                                                        double* __restrict__ eout,
                                                        long long eout_stride, int ecopies,
                                                        long long ecopy_stride) {
  using C = typename Cx<F>::type;
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  constexpr int VEC = (sizeof(F) == 4) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C* lds = reinterpret_cast<C*>(smem);

  const uint32_t tid = threadIdx.x;
  state += (long long)blockIdx.y * state_stride;
  const KInt desc = (KInt)desc_g;
  const KPtr<F> ctab = (KPtr<F>)ctab_g;
  const KPtr<F> ptab = (KPtr<F>)(ptab_g + (long long)blockIdx.y * ptab_stride);

  const int nrounds = desc[5];
  const int flags = desc[6];
  if (eout) eout += (long long)blockIdx.y * eout_stride + (long long)(blockIdx.x % (unsigned)ecopies) * ecopy_stride;
  // workgroup base index: deposit blockIdx.x into the non-tile bit positions
  unsigned long long x = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < T; ++i) {
    const int p = desc[8 + i];
    const unsigned long long low = (1ull << p) - 1ull;
    x = ((x & ~low) << 1) | (x & low);
  }
  const uint32_t wg_base = (uint32_t)x;

  // measurement passes: the waves' partial sums meet in LDS and leave as one f64 atomic per term, workgroup and
  // pass (one atomic per WAVE and term serialised 2^(n-T+2) same-address atomics per term at the memory side:
  // the n = 28 TFIM measurement ran at 1.25 TB/s).  eacc[i] / eidx[i]: sum and output index of the i-th partial
  // sum of the pass, numbered in program order (identical in every wave: the op loop is descriptor-driven).
  constexpr int EACC = 512;
  F* const eacc = reinterpret_cast<F*>(lds + (1 << T));  // accumulated in the state's real type
  int* const eidx = reinterpret_cast<int*>(eacc + EACC);
  int eev = 0;
  if (MODE == 1) {
    for (int i = tid; i < EACC; i += (1 << LT)) eacc[i] = (F)0;
    __syncthreads();
  }
#define TCMI_EFLUSH()                                             \
  {                                                               \
    __syncthreads();                                              \
    for (int i = tid; i < eev; i += (1 << LT)) {                  \
      atomicAdd(eout + eidx[i], (double)eacc[i]);                 \
      eacc[i] = (F)0;                                             \
    }                                                             \
    eev = 0;                                                      \
    __syncthreads();                                              \
  }
#define TCMI_EADD(IDX, VAL)                                       \
  {                                                               \
    if (eev == EACC) TCMI_EFLUSH() /* workgroup-uniform */        \
    if ((tid & 63) == 0) {                                        \
      atomicAdd(eacc + eev, (F)(VAL));                            \
      eidx[eev] = IDX;                                            \
    }                                                             \
    ++eev;                                                        \
  }

  C a[NR];
  int pc = TCMI_HDR_WORDS;
#pragma unroll 1
  for (int k = 0; k < nrounds; ++k) {
    const KInt rr = desc + pc;
    const int nops = rr[0];
    const uint32_t tphys = xor_masks<LT>(tid, rr + 8);
    uint32_t rpm[R];
#pragma unroll
    for (int j = 0; j < R; ++j) rpm[j] = (uint32_t)rr[2 + j];

    if (k == 0) {
      const C* __restrict__ src = state + (wg_base | tphys);
#pragma unroll
      for (int r = 0; r < NR; r += VEC) {
        const uint32_t off = reg_mask<R>(r, rpm);
        if constexpr (VEC == 2) {
          const float4 v = *reinterpret_cast<const float4*>(src + off);
          a[r].x = v.x; a[r].y = v.y; a[r + 1].x = v.z; a[r + 1].y = v.w;
        } else {
          a[r] = src[off];
        }
      }
    } else {
      const uint32_t tslot = xor_masks<LT>(tid, rr + 24);
      uint32_t rsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) rsm[j] = (uint32_t)rr[18 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) a[r] = lds[tslot ^ reg_mask<R>(r, rsm)];
      __syncthreads();  // all reads done before the next exchange overwrites the tile
    }

    // ---- ops of this round ----
    int q = pc + TCMI_RR_WORDS;
#pragma unroll 1
    for (int o = 0; o < nops; ++o) {
      const int op = desc[q];
      if (MODE == 0 && op == TCMI_OP_G2) {
        const KPtr<F> mp = tab_ptr<F>(desc[q + 3], ctab, ptab);
        F m[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) m[i] = mp[i];
        dispatch_g2<F, NR, R>(a, desc[q + 1], desc[q + 2], m);
        q += 4;
      } else if (MODE == 0 && op == TCMI_OP_G1M) {
        // one-qubit gates on the register bits in mask; the R matrices sit contiguously in the
        // per-batch table and are fetched with one burst of scalar loads
        const int mk = desc[q + 1];
        const KPtr<F> mp = ptab + desc[q + 2];
        F mm[R][8];
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) mm[j][i] = mp[8 * j + i];
        }
#pragma unroll 1
        for (int j = 0; j < R; ++j) {
          if (!((mk >> j) & 1)) continue;
          dispatch_g1m<F, NR, R>(a, j, (mk >> (8 + 2 * j)) & 3, mm);
        }
        q += 3;
      } else if (MODE == 0 && op == TCMI_OP_DIAG) {
        const int nA = desc[q + 1], nB = desc[q + 2], nC = desc[q + 3];
        const KPtr<F> cf = ptab + desc[q + 4];
        q += 5;
        const uint32_t tidx = wg_base | tphys;
        double phi = 0.0;
#pragma unroll 1
        for (int e = 0; e < nA; e += TCMI_DIAG_CHUNK) {
          uint32_t mk[TCMI_DIAG_CHUNK];
          F cc[TCMI_DIAG_CHUNK];
#pragma unroll
          for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
            mk[i] = (uint32_t)desc[q + e + i];
            cc[i] = cf[e + i];
          }
#pragma unroll
          for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
            const double c = (double)cc[i];
            phi += (__popc(tidx & mk[i]) & 1) ? -c : c;
          }
        }
        q += nA;
        double cj[R];
#pragma unroll
        for (int j = 0; j < R; ++j) cj[j] = 0.0;
#pragma unroll 1
        for (int e = 0; e < nB; e += TCMI_DIAG_CHUNK) {
          uint32_t mk[TCMI_DIAG_CHUNK];
          int jj[TCMI_DIAG_CHUNK];
          F cc[TCMI_DIAG_CHUNK];
#pragma unroll
          for (int i = 0; i < TCMI_DIAG_CHUNK; ++i) {
            mk[i] = (uint32_t)desc[q + e + i];
            jj[i] = desc[q + nB + e + i];
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
        q += 2 * nB;
        F ph[NR];
        ph[0] = (F)(phi - rint(phi));
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const F c = (F)(cj[j] - rint(cj[j]));
#pragma unroll
          for (int r = 0; r < (1 << j); ++r) {
            ph[r | (1 << j)] = ph[r] - c;
            ph[r] += c;
          }
        }
#pragma unroll 1
        for (int e = 0; e < nC; ++e) {
          const uint32_t rmask = (uint32_t)desc[q + e];
          const F c = cf[nA + nB + e];
#pragma unroll
          for (int r = 0; r < NR; ++r) ph[r] += (__popc((uint32_t)r & rmask) & 1) ? -c : c;
        }
        q += nC;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          F s, c;
          sincos_turns<F>(ph[r], &s, &c);
          const C v = a[r];
          a[r].x = v.x * c - v.y * s;
          a[r].y = v.x * s + v.y * c;
        }
      } else if (MODE == 0 && op == TCMI_OP_DIAGB) {
        // one term on register bit j and thread bits: exp(i phi z_j(r) s(thread)), {cos, sin} from the builder
        const int j = desc[q + 1];
        const uint32_t m = (uint32_t)desc[q + 2];
        const KPtr<F> tp = ptab + desc[q + 3];
        q += 4;
        const F cs = tp[0], sn = tp[1];
        const F ys = (__popc((wg_base | tphys) & m) & 1) ? -sn : sn;
        dispatch_diagb<F, NR, R>(a, j, cs, ys);
      } else if (MODE == 0 && op == TCMI_OP_DIAGC) {
        // diagonal terms on register bits only: the 2^R phase factors are wave-uniform and come from the
        // builder (scalar loads), 4 lane-instructions per amplitude
        const KPtr<F> tp = ptab + desc[q + 1];
        q += 2;
        F tb[2 * NR];
#pragma unroll
        for (int i = 0; i < 2 * NR; ++i) tb[i] = tp[i];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const C v = a[r];
          a[r].x = v.x * tb[2 * r] - v.y * tb[2 * r + 1];
          a[r].y = v.x * tb[2 * r + 1] + v.y * tb[2 * r];
        }
      } else if (MODE == 1 && op == TCMI_OP_EXPECT) {
        // <psi|P_t|psi> partial sums for Pauli strings whose X/Y bits are register bits of this round
        const int nZ = desc[q + 1], nX = desc[q + 2];
        q += 3;
        const uint32_t tidx = wg_base | tphys;
        if (nZ > 0) {
          F p[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) p[r] = fma_<F>(a[r].x, a[r].x, a[r].y * a[r].y);
#pragma unroll 1
          for (int e = 0; e < nZ; ++e) {
            const uint32_t zr = (uint32_t)desc[q], zm = (uint32_t)desc[q + 1];
            const int oi = desc[q + 2];
            F acc = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) acc += (__popc((uint32_t)r & zr) & 1) ? -p[r] : p[r];
            if (__popc(tidx & zm) & 1) acc = -acc;
            acc = wave_sum_uniform(acc);
            TCMI_EADD(2 * oi, acc)
            q += 3;
          }
        }
#pragma unroll 1
        for (int e = 0; e < nX; ++e) {
          const int xr = desc[q];
          const uint32_t zr = (uint32_t)desc[q + 1], zm = (uint32_t)desc[q + 2];
          const int oi = desc[q + 3];
          F re = 0, im = 0;
          dispatch_expect_x<F, NR>(a, xr, zr, re, im);
          if (__popc(tidx & zm) & 1) { re = -re; im = -im; }
          re = wave_sum_uniform(re);
          im = wave_sum_uniform(im);
          TCMI_EADD(2 * oi, re)
          TCMI_EADD(2 * oi + 1, im)
          q += 4;
        }
      } else {
        break;  // unknown opcode: host validates descriptors, never reached
      }
    }
    pc += TCMI_RR_WORDS + rr[1];

    if (k < nrounds - 1) {
      const uint32_t tslot = xor_masks<LT>(tid, rr + 40);
      uint32_t wsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) wsm[j] = (uint32_t)rr[34 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) lds[tslot ^ reg_mask<R>(r, wsm)] = a[r];
      __syncthreads();
    } else if (!(flags & TCMI_FLAG_NOSTORE)) {
      C* __restrict__ dst = state + (wg_base | tphys);
#pragma unroll
      for (int r = 0; r < NR; r += VEC) {
        const uint32_t off = reg_mask<R>(r, rpm);
        if constexpr (VEC == 2) {
          float4 v;
          v.x = a[r].x; v.y = a[r].y; v.z = a[r + 1].x; v.w = a[r + 1].y;
          *reinterpret_cast<float4*>(dst + off) = v;
        } else {
          dst[off] = a[r];
        }
      }
    }
  }
  if (MODE == 1) TCMI_EFLUSH()
#undef TCMI_EADD
#undef TCMI_EFLUSH
}

// ---- builder: parameters -> per-batch gate tables (reference gates.py:692-743, 920-953) -------
template <typename F>
__global__ void build_kernel(const int* __restrict__ ginfo, int nrec, const double* __restrict__ cpool,
                             const F* __restrict__ params, long long pstride, F* __restrict__ ptab,
                             long long tstride, int batch) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (g >= nrec || b >= batch) return;
  const int* rec = ginfo + 8 * g;
  const int kind = rec[0], slot = rec[1], pidx = rec[2], dim = rec[3], off = rec[4];
  const double theta = (double)params[(long long)b * pstride + pidx];
  const double ang = fma(cpool[off], theta, cpool[off + 1]);
  F* out = ptab + (long long)b * tstride + slot;
  if (kind == TCMI_BK_TRIG) {
    double s, c;
    sincos(ang, &s, &c);
    const int nn = 2 * dim * dim;
    const double* c0 = cpool + off + 2;
    const double* c1 = c0 + nn;
    const double* c2 = c1 + nn;
    if (rec[5]) {
      // three-shear form of a rotation (plan.g1_shear_flavor): sign * M = S(u) L(v) S(u) on (x, y) [flavor 1]
      // or on (x, i y) [flavor 2]; out = {u, v, sign}
      double a = c0[0] + c * c1[0] + s * c2[0];
      double cc = (rec[5] == 1) ? (c0[4] + c * c1[4] + s * c2[4]) : (c0[5] + c * c1[5] + s * c2[5]);
      for (int i = 3; i < 8; ++i) out[i] = (F)0;
      if (rec[6] && fabs(a) >= TCMI_SHEAR2_CMIN) {
        // two-shear form M = diag(a, 1 / a) L(v) S(u) (the diagonal factor is a pending scale term of the plan)
        out[0] = (F)((rec[5] == 1 ? -cc : cc) / a);
        out[1] = (F)(cc * a);
        out[2] = (F)1;
        out[3] = (F)2;  // bit 30 of the float: the kernels' form flag
        return;
      }
      const double sg = a < 0 ? -1.0 : 1.0;
      a *= sg;
      cc *= sg;
      const double num = (rec[5] == 1) ? (a - 1.0) : (1.0 - a);
      out[0] = (F)(fabs(cc) > 1e-30 ? num / cc : 0.0);
      out[1] = (F)cc;
      out[2] = (F)sg;
      return;
    }
    for (int i = 0; i < nn; ++i) out[i] = (F)(c0[i] + c * c1[i] + s * c2[i]);
  } else if (kind == TCMI_BK_PHASE) {
    // one entry of a DIAGC table: exp(2 pi i sum_t s_t(r) (k_t theta_t + o_t)), s_t = parity of r & mask_t
    phase_entry<F>(cpool + off, dim, rec[5], params + (long long)b * pstride, false, out);
  } else if (kind == TCMI_BK_COEF) {
    out[0] = (F)(ang - rint(ang));  // phase coefficient in turns, reduced to [-0.5, 0.5]
  } else if (kind == TCMI_BK_SELECT) {
    // one of cpool[off] constant matrices, chosen by the (integer-valued) parameter
    const int nsel = (int)cpool[off];
    int idx = (int)rint(theta);
    idx = idx < 0 ? 0 : (idx >= nsel ? nsel - 1 : idx);
    const int nn = 2 * dim * dim;
    const double* tbl = cpool + off + 2 + (long long)idx * nn;
    for (int i = 0; i < nn; ++i) out[i] = (F)tbl[i];
  }
}

template <typename F>
__global__ void zero_state_kernel(typename Cx<F>::type* __restrict__ state, long long stride,
                                  unsigned long long nelem) {
  using C = typename Cx<F>::type;
  C* s = state + (long long)blockIdx.y * stride;
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = i0; i < nelem; i += step) {
    C v;
    v.x = (i == 0) ? (F)1 : (F)0;
    v.y = (F)0;
    s[i] = v;
  }
}

// Weights of the cut contraction: w[b][k] = prod_j coef_j(digit_j(k), theta_b), one thread per (b, k).  An entry is a
// constant or cos / sin of (scale * theta[param] + offset); the product runs in float64 whatever the state's precision.
template <typename F>
__global__ void cut_weights_kernel(const F* __restrict__ params, long long pstride, const int* __restrict__ tab_i,
                                   const double* __restrict__ tab_f, const unsigned char* __restrict__ digits, int K,
                                   int nb, int rmax, typename Cx<F>::type* __restrict__ w) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (k >= K) return;
  double ar = 1.0, ai = 0.0;
  for (int j = 0; j < nb; ++j) {
    const int e = j * rmax + digits[k * nb + j];
    const int kind = tab_i[e] & 3;
    double vr, vi = 0.0;
    if (kind == 0) {
      vr = tab_f[4 * e + 2];
      vi = tab_f[4 * e + 3];
    } else {
      const double a = (double)params[(long long)b * pstride + (tab_i[e] >> 2)] * tab_f[4 * e] + tab_f[4 * e + 1];
      vr = kind == 1 ? cos(a) : sin(a);
    }
    const double nr = ar * vr - ai * vi;
    ai = ar * vi + ai * vr;
    ar = nr;
  }
  typename Cx<F>::type o;
  o.x = (F)ar;
  o.y = (F)ai;
  w[(long long)b * K + k] = o;
}

// The 4 x 4 epilogue of a cut contraction with a deferred last crossing gate (tcmi/cut.py Epilogue): X[b] = prod_g (c0_g +
// cos(a) c1_g + sin(a) c2_g), later factors on the left, a = scale * theta_b[param] + offset; in float64, stored as
// complex64 row-major [out][in].  Factor g: tab_i[g] = parameter index (-1: constant), tab_f[g * 98] = {scale, offset,
// c0[16] (re, im), c1[16], c2[16]}.  Sixteen threads per circuit (one per matrix element, four circuits per workgroup):
// per factor every thread forms its element of the factor, the 4 x 4 product goes through LDS.
template <typename F>
__global__ __launch_bounds__(64) void cut_epilogue_kernel(const F* __restrict__ params, long long pstride, int batch,
                                                          const int* __restrict__ tab_i, const double* __restrict__ tab_f,
                                                          int nfac, float2* __restrict__ X) {
  __shared__ double xs[4][2][16], ms[4][2][16];
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int b = blockIdx.x * 4 + q;
  const int bb = b < batch ? b : batch - 1;
  const int o = e >> 2, c = e & 3;
  double xr = o == c ? 1.0 : 0.0, xi = 0.0;
  for (int g = 0; g < nfac; ++g) {
    const double* t = tab_f + (long long)g * 98;
    double cs = 0.0, sn = 0.0;
    const int pi = tab_i[g];
    if (pi >= 0) {
      const double a = (double)params[(long long)bb * pstride + pi] * t[0] + t[1];
      cs = cos(a);
      sn = sin(a);
    }
    __syncthreads();
    xs[q][0][e] = xr;
    xs[q][1][e] = xi;
    ms[q][0][e] = t[2 + 2 * e] + cs * t[34 + 2 * e] + sn * t[66 + 2 * e];
    ms[q][1][e] = t[3 + 2 * e] + cs * t[35 + 2 * e] + sn * t[67 + 2 * e];
    __syncthreads();
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double mr = ms[q][0][4 * o + k], mi = ms[q][1][4 * o + k];
      const double yr = xs[q][0][4 * k + c], yi = xs[q][1][4 * k + c];
      ar += mr * yr - mi * yi;
      ai += mr * yi + mi * yr;
    }
    xr = ar;
    xi = ai;
  }
  if (b < batch) {
    float2 w;
    w.x = (float)xr;
    w.y = (float)xi;
    X[(long long)b * 16 + e] = w;
  }
}

// second-generation complex64 measurement pass (tcmi_measure2.hip); -1 = no variant for this (R, LT)
int run_measure2_c64(const void* state, long long state_stride, int batch, int n, int R, int LT, const int* desc,
                     double* eout, long long eout_stride, int ecopies, long long ecopy_stride, hipStream_t st);

// second-generation complex64 gate pass (tcmi_vm2.hip); -1 = no variant for this (R, LT)
int run_pass2_c64(void* state, long long state_stride, int batch, int n, int R, int LT, const int* desc,
                  const void* ctab, const void* ptab, long long ptab_stride, hipStream_t st);

}  // namespace tcmi

// ---- C ABI ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return TCMI_ERR_HIP;
}
static int set_msg(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

template <typename F, int R, int LT, int MODE>
static int launch_pass_mode(void* state, long long state_stride, int batch, int n, const int* desc,
                       const void* ctab, const void* ptab, long long ptab_stride, double* eout,
                       long long eout_stride, int ecopies, long long ecopy_stride, hipStream_t st) {
  using C = typename tcmi::Cx<F>::type;
  constexpr int T = R + LT;
  if (n < T) return set_msg(TCMI_ERR_ARG, "tcmi_run_pass: n smaller than the tile");
  const size_t lds = (sizeof(C) << T) + (MODE == 1 ? 512 * (sizeof(F) + sizeof(int)) : 0);
  auto kern = tcmi::pass_kernel<F, R, LT, MODE>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_err("hipFuncSetAttribute", e);
  }
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(1u << LT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<C*>(state), state_stride, desc,
                     reinterpret_cast<const F*>(ctab), reinterpret_cast<const F*>(ptab), ptab_stride,
                     eout, eout_stride, ecopies < 1 ? 1 : ecopies, ecopy_stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err("pass_kernel launch", e);
  return TCMI_OK;
}

template <typename F, int R, int LT>
static int launch_pass(void* state, long long state_stride, int batch, int n, const int* desc,
                       const void* ctab, const void* ptab, long long ptab_stride, double* eout,
                       long long eout_stride, int ecopies, long long ecopy_stride, hipStream_t st) {
  if (eout)
    return launch_pass_mode<F, R, LT, 1>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, eout, eout_stride, ecopies, ecopy_stride, st);
  return launch_pass_mode<F, R, LT, 0>(state, state_stride, batch, n, desc, ctab, ptab, ptab_stride, eout, eout_stride, ecopies, ecopy_stride, st);
}

extern "C" int tcmi_set_error_(int code, const char* msg) { return set_msg(code, msg); }

extern "C" {

int tcmi_version(void) { return TCMI_VERSION; }

const char* tcmi_last_error(void) { return g_err; }

int tcmi_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int tcmi_run_pass(void* state, long long state_stride, int batch, int n, int R, int LT,
                  const int* desc_dev, const void* ctab_dev, const void* ptab_dev,
                  long long ptab_stride, double* eout_dev, long long eout_stride, int ecopies,
                  long long ecopy_stride, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!state || !desc_dev || batch < 1) return set_msg(TCMI_ERR_ARG, "tcmi_run_pass: bad argument");
  if (n > 32) return set_msg(TCMI_ERR_ARG, "tcmi_run_pass: n > 32 unsupported");
  if (dtype == TCMI_C64 && eout_dev && R == 5 && LT == 8 && n >= R + LT) {
    // measurement passes compiled for the second-generation kernel (TCMI_OP_EXPECT2 descriptors: the host emits them
    // for exactly this tile, tcmi/plan.py encode_measure_pass)
    const int rc = tcmi::run_measure2_c64(state, state_stride, batch, n, R, LT, desc_dev, eout_dev, eout_stride, ecopies,
                                          ecopy_stride, st);
    if (rc == TCMI_OK) return rc;
    return set_err("measure2_kernel launch", hipGetLastError());
  }
  if (dtype == TCMI_C64 && !eout_dev && n >= R + LT) {
    {
      const int rc = tcmi::run_pass2_c64(state, state_stride, batch, n, R, LT, desc_dev, ctab_dev, ptab_dev, ptab_stride, st);
      if (rc == TCMI_OK) return rc;
      if (rc != -1) return set_err("pass2_kernel launch", hipGetLastError());
    }
  }
#define TCMI_CASE(FT, RR, LL) \
  if (R == RR && LT == LL)    \
    return launch_pass<FT, RR, LL>(state, state_stride, batch, n, desc_dev, ctab_dev, ptab_dev, ptab_stride, eout_dev, eout_stride, ecopies, ecopy_stride, st);
  if (dtype == TCMI_C64) {
    TCMI_CASE(float, 5, 8)
    TCMI_CASE(float, 4, 8)
    TCMI_CASE(float, 5, 9)
    TCMI_CASE(float, 4, 9)
    TCMI_CASE(float, 2, 6)
  } else if (dtype == TCMI_C128) {
    TCMI_CASE(double, 4, 8)
    TCMI_CASE(double, 3, 8)
    TCMI_CASE(double, 2, 6)
  }
#undef TCMI_CASE
  return set_msg(TCMI_ERR_ARG, "tcmi_run_pass: unsupported (dtype, R, LT) variant");
}

int tcmi_build_tables(const int* ginfo_dev, int nrec, const double* cpool_dev, const void* params_dev,
                      long long params_stride, void* ptab_dev, long long ptab_stride, int batch,
                      int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (nrec == 0) return TCMI_OK;
  if (!ginfo_dev || !cpool_dev || !params_dev || !ptab_dev || batch < 1)
    return set_msg(TCMI_ERR_ARG, "tcmi_build_tables: bad argument");
  dim3 block(128, 1, 1), grid((nrec + 127) / 128, batch, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::build_kernel<float>, grid, block, 0, st, ginfo_dev, nrec, cpool_dev,
                       reinterpret_cast<const float*>(params_dev), params_stride,
                       reinterpret_cast<float*>(ptab_dev), ptab_stride, batch);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::build_kernel<double>, grid, block, 0, st, ginfo_dev, nrec, cpool_dev,
                       reinterpret_cast<const double*>(params_dev), params_stride,
                       reinterpret_cast<double*>(ptab_dev), ptab_stride, batch);
  else
    return set_msg(TCMI_ERR_ARG, "tcmi_build_tables: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err("build_kernel launch", e);
  return TCMI_OK;
}

int tcmi_cut_weights(const void* params_dev, long long params_stride, int batch, const int* tab_i_dev,
                     const double* tab_f_dev, const unsigned char* digits_dev, int K, int nb, int rmax, void* w_dev,
                     int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!params_dev || !tab_i_dev || !tab_f_dev || !digits_dev || !w_dev || batch < 1 || K < 1 || nb < 0 || rmax < 1)
    return set_msg(TCMI_ERR_ARG, "tcmi_cut_weights: bad argument");
  dim3 block(128, 1, 1), grid((K + 127) / 128, batch, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::cut_weights_kernel<float>, grid, block, 0, st, reinterpret_cast<const float*>(params_dev),
                       params_stride, tab_i_dev, tab_f_dev, digits_dev, K, nb, rmax, reinterpret_cast<float2*>(w_dev));
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::cut_weights_kernel<double>, grid, block, 0, st, reinterpret_cast<const double*>(params_dev),
                       params_stride, tab_i_dev, tab_f_dev, digits_dev, K, nb, rmax, reinterpret_cast<double2*>(w_dev));
  else
    return set_msg(TCMI_ERR_ARG, "tcmi_cut_weights: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err("cut_weights_kernel launch", e);
  return TCMI_OK;
}

int tcmi_cut_epilogue(const void* params_dev, long long params_stride, int batch, const int* tab_i_dev,
                      const double* tab_f_dev, int nfac, void* x_dev, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!params_dev || !tab_i_dev || !tab_f_dev || !x_dev || batch < 1 || nfac < 0)
    return set_msg(TCMI_ERR_ARG, "tcmi_cut_epilogue: bad argument");
  dim3 block(64, 1, 1), grid((batch + 3) / 4, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::cut_epilogue_kernel<float>, grid, block, 0, st, reinterpret_cast<const float*>(params_dev),
                       params_stride, batch, tab_i_dev, tab_f_dev, nfac, reinterpret_cast<float2*>(x_dev));
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::cut_epilogue_kernel<double>, grid, block, 0, st, reinterpret_cast<const double*>(params_dev),
                       params_stride, batch, tab_i_dev, tab_f_dev, nfac, reinterpret_cast<float2*>(x_dev));
  else
    return set_msg(TCMI_ERR_ARG, "tcmi_cut_epilogue: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err("cut_epilogue_kernel launch", e);
  return TCMI_OK;
}

int tcmi_init_zero_state(void* state, long long state_stride, int batch, int n, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!state || batch < 1 || n < 0 || n > 34) return set_msg(TCMI_ERR_ARG, "tcmi_init_zero_state: bad argument");
  const unsigned long long nelem = 1ull << n;
  unsigned gx = (unsigned)((nelem + 255) / 256 > 4096 ? 4096 : (nelem + 255) / 256);
  dim3 grid(gx, batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::zero_state_kernel<float>, grid, block, 0, st,
                       reinterpret_cast<float2*>(state), state_stride, nelem);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::zero_state_kernel<double>, grid, block, 0, st,
                       reinterpret_cast<double2*>(state), state_stride, nelem);
  else
    return set_msg(TCMI_ERR_ARG, "tcmi_init_zero_state: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err("zero_state_kernel launch", e);
  return TCMI_OK;
}

}  // extern "C"
