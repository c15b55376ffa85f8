// tcmi adjoint sweep: the reverse-mode (VJP) kernel of the state-vector path for gfx950.
//
// The reference differentiates through the stored intermediates of every tensordot
// (framework AD: jax.value_and_grad / torch.func.grad_and_value, reference
// tensorcircuit/backends/jax_backend.py:854-952, pytorch_backend.py:775-786); at 28 qubits that
// means hundreds of 2 GiB states.  Here the backward pass is an *adjoint sweep*: the forward plan
// is replayed in reverse on two vectors at once -- psi (un-computed gate by gate with U^dagger) and
// the cotangent lambda -- and every parametrised gate contributes
//        dL/dtheta += Re < lambda_after | K | psi_after >,   K = (dU/dtheta) U^dagger
// before both vectors are multiplied by U^dagger.  Memory: two states, independent of depth.
//
// Same tile-VM structure as tcmi_vm.hip (tile in registers, LDS exchanges between rounds), with both
// vectors resident: 2 x 2^R amplitudes per thread.  Descriptor layout: tcmi_vm.h ("backward ops").

#include "tcmi_dev.h"

namespace tcmi {

// Re sum conj(l) * (K a) over a pair on register bit J, then nothing is modified
template <typename F, int NR, int J>
__device__ __forceinline__ F grad_g1(const typename Cx<F>::type (&a)[NR], const typename Cx<F>::type (&l)[NR],
                                     const F (&k)[8]) {
  F acc = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    F t0r = k[0] * a[r].x, t0i = k[0] * a[r].y, t1r = k[4] * a[r].x, t1i = k[4] * a[r].y;
    t0r = fma_<F>(-k[1], a[r].y, t0r);
    t0i = fma_<F>(k[1], a[r].x, t0i);
    t1r = fma_<F>(-k[5], a[r].y, t1r);
    t1i = fma_<F>(k[5], a[r].x, t1i);
    cfma<F>(k[2], k[3], a[r1], t0r, t0i);
    cfma<F>(k[6], k[7], a[r1], t1r, t1i);
    acc = fma_<F>(l[r].x, t0r, acc);
    acc = fma_<F>(l[r].y, t0i, acc);
    acc = fma_<F>(l[r1].x, t1r, acc);
    acc = fma_<F>(l[r1].y, t1i, acc);
  }
  return acc;
}

// Structured generators.  For a unitary 1-qubit gate whose matrix is "real diagonal + imaginary
// off-diagonal" for every parameter value (plan.g1_kind == 2: rx and every exp(i phi(theta) X)),
// K = (dU/dtheta) U^dagger is anti-Hermitian of the same class: K = i kappa X, so
// Re<l|K|a> = -kappa Im(conj(l0) a1 + conj(l1) a0) — 4 FMAs per pair instead of 20.  Real matrices
// (kind 1: ry) give the real antisymmetric K = [[0, k01], [k10, 0]].
template <typename F, int NR, int J>
__device__ __forceinline__ F grad_g1_k2(const typename Cx<F>::type (&a)[NR], const typename Cx<F>::type (&l)[NR],
                                        const F (&k)[8]) {
  F acc = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    acc = fma_<F>(l[r].x, a[r1].y, acc);
    acc = fma_<F>(-l[r].y, a[r1].x, acc);
    acc = fma_<F>(l[r1].x, a[r].y, acc);
    acc = fma_<F>(-l[r1].y, a[r].x, acc);
  }
  return -k[3] * acc;
}

template <typename F, int NR, int J>
__device__ __forceinline__ F grad_g1_k1(const typename Cx<F>::type (&a)[NR], const typename Cx<F>::type (&l)[NR],
                                        const F (&k)[8]) {
  F a01 = 0, a10 = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    a01 = fma_<F>(l[r].x, a[r1].x, a01);
    a01 = fma_<F>(l[r].y, a[r1].y, a01);
    a10 = fma_<F>(l[r1].x, a[r].x, a10);
    a10 = fma_<F>(l[r1].y, a[r].y, a10);
  }
  return k[2] * a01 + k[4] * a10;
}

template <typename F, int NR, int JA, int JB>
__device__ __forceinline__ F grad_g2(const typename Cx<F>::type (&a)[NR], const typename Cx<F>::type (&l)[NR],
                                     const F (&k)[32]) {
  F acc = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (((r >> JA) & 1) || ((r >> JB) & 1)) continue;
    const int idx[4] = {r, r | (1 << JB), r | (1 << JA), r | (1 << JA) | (1 << JB)};
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      F re = 0, im = 0;
#pragma unroll
      for (int col = 0; col < 4; ++col) cfma<F>(k[2 * (4 * row + col)], k[2 * (4 * row + col) + 1], a[idx[col]], re, im);
      acc = fma_<F>(l[idx[row]].x, re, acc);
      acc = fma_<F>(l[idx[row]].y, im, acc);
    }
  }
  return acc;
}

template <typename F, int NR, int R, int J>
__device__ __forceinline__ void bw_g1_bit(typename Cx<F>::type (&a)[NR], typename Cx<F>::type (&l)[NR], int kind,
                                          bool has_k, const F (&ud)[8], const F (&kk)[8], double* gslot,
                                          uint32_t tid) {
  if (has_k) {
    F g = kind == 2 ? grad_g1_k2<F, NR, J>(a, l, kk) : kind == 1 ? grad_g1_k1<F, NR, J>(a, l, kk) : grad_g1<F, NR, J>(a, l, kk);
    g = wave_sum_uniform(g);
    if ((tid & 63) == 0) atomicAdd(gslot, (double)g);
  }
  if (kind == 1) { apply_g1<F, NR, J, 1>(a, ud); apply_g1<F, NR, J, 1>(l, ud); }
  else if (kind == 2) { apply_g1<F, NR, J, 2>(a, ud); apply_g1<F, NR, J, 2>(l, ud); }
  else { apply_g1<F, NR, J, 0>(a, ud); apply_g1<F, NR, J, 0>(l, ud); }
}

#define TCMI_BG2_CASE(A, B)                                                           \
  case (A * 8 + B):                                                                   \
    if constexpr (R > B) {                                                            \
      if (has_k) {                                                                    \
        F g = grad_g2<F, NR, A, B>(a, l, kk);                                         \
        g = wave_sum_uniform(g);                                                      \
        if ((tid & 63) == 0) atomicAdd(gslot, (double)g);                             \
      }                                                                               \
      if (kind == 0) { apply_g2<F, NR, A, B>(a, ud); apply_g2<F, NR, A, B>(l, ud); }  \
      else if (kind == 1) { apply_perm2<F, NR, A, B, 1>(a); apply_perm2<F, NR, A, B, 1>(l); } \
      else if (kind == 2) { apply_perm2<F, NR, A, B, 2>(a); apply_perm2<F, NR, A, B, 2>(l); } \
      else { apply_perm2<F, NR, A, B, 3>(a); apply_perm2<F, NR, A, B, 3>(l); }        \
    }                                                                                 \
    break;

template <typename F, int R, int LT>
__global__ __launch_bounds__(1 << LT) void adjoint_kernel(typename Cx<F>::type* __restrict__ psi,
                                                           typename Cx<F>::type* __restrict__ lam,
                                                           long long state_stride,
                                                           const int* __restrict__ desc_g,
                                                           const F* __restrict__ ctab_g,
                                                           const F* __restrict__ ptab_g,
                                                           long long ptab_stride,
                                                           double* __restrict__ gout,
                                                           long long gout_stride, int gcopies,
                                                           long long gcopy_stride) {
  using C = typename Cx<F>::type;
  constexpr int NR = 1 << R;
  constexpr int T = R + LT;
  constexpr int VEC = (sizeof(F) == 4) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  C* lds = reinterpret_cast<C*>(smem);
  C* lds2 = lds + (1 << T);

  const uint32_t tid = threadIdx.x;
  psi += (long long)blockIdx.y * state_stride;
  lam += (long long)blockIdx.y * state_stride;
  // gradient slots are replicated gcopies times to spread same-address atomics over L2 channels
  gout += (long long)blockIdx.y * gout_stride + (long long)(blockIdx.x % (unsigned)gcopies) * gcopy_stride;
  const KInt desc = (KInt)desc_g;
  const KPtr<F> ctab = (KPtr<F>)ctab_g;
  const KPtr<F> ptab = (KPtr<F>)(ptab_g + (long long)blockIdx.y * ptab_stride);

  const int nrounds = desc[5];
  unsigned long long x = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < T; ++i) {
    const int p = desc[8 + i];
    const unsigned long long low = (1ull << p) - 1ull;
    x = ((x & ~low) << 1) | (x & low);
  }
  const uint32_t wg_base = (uint32_t)x;

  C a[NR], l[NR];
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
      const C* __restrict__ sa = psi + (wg_base | tphys);
      const C* __restrict__ sl = lam + (wg_base | tphys);
#pragma unroll
      for (int r = 0; r < NR; r += VEC) {
        const uint32_t off = reg_mask<R>(r, rpm);
        if constexpr (VEC == 2) {
          const float4 v = *reinterpret_cast<const float4*>(sa + off);
          const float4 w = *reinterpret_cast<const float4*>(sl + off);
          a[r].x = v.x; a[r].y = v.y; a[r + 1].x = v.z; a[r + 1].y = v.w;
          l[r].x = w.x; l[r].y = w.y; l[r + 1].x = w.z; l[r + 1].y = w.w;
        } else {
          a[r] = sa[off];
          l[r] = sl[off];
        }
      }
    } else {
      const uint32_t tslot = xor_masks<LT>(tid, rr + 24);
      uint32_t rsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) rsm[j] = (uint32_t)rr[18 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t s = tslot ^ reg_mask<R>(r, rsm);
        a[r] = lds[s];
        l[r] = lds2[s];
      }
      __syncthreads();
    }

    int q = pc + TCMI_RR_WORDS;
#pragma unroll 1
    for (int o = 0; o < nops; ++o) {
      const int op = desc[q];
      if (op == TCMI_OP_G1M) {
        // {4, mask|kinds<<8, ubase, kmask, kbase, gslot[R]}
        const int mk = desc[q + 1], kmask = desc[q + 3];
        const KPtr<F> up = ptab + desc[q + 2];
        const KPtr<F> kp = ptab + desc[q + 4];
#pragma unroll 1
        for (int j = 0; j < R; ++j) {
          if (!((mk >> j) & 1)) continue;
          F ud[8], kk[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) { ud[i] = up[8 * j + i]; kk[i] = kp[8 * j + i]; }
          const int kind = (mk >> (8 + 2 * j)) & 3;
          const bool has_k = (kmask >> j) & 1;
          double* gs = gout + desc[q + 5 + j];
          switch (j) {
            case 0: bw_g1_bit<F, NR, R, 0>(a, l, kind, has_k, ud, kk, gs, tid); break;
            case 1: if constexpr (R > 1) bw_g1_bit<F, NR, R, 1>(a, l, kind, has_k, ud, kk, gs, tid); break;
            case 2: if constexpr (R > 2) bw_g1_bit<F, NR, R, 2>(a, l, kind, has_k, ud, kk, gs, tid); break;
            case 3: if constexpr (R > 3) bw_g1_bit<F, NR, R, 3>(a, l, kind, has_k, ud, kk, gs, tid); break;
            default: break;
          }
        }
        q += 5 + R;
      } else if (op == TCMI_OP_G2) {
        // {2, ja|kind<<8, jb, uslot, kslot(-1 = none), gslot}
        const int ja = desc[q + 1] & 0xff, kind = desc[q + 1] >> 8, jb = desc[q + 2];
        const KPtr<F> up = tab_ptr<F>(desc[q + 3], ctab, ptab);
        const bool has_k = desc[q + 4] >= 0;
        const KPtr<F> kp = ptab + (has_k ? desc[q + 4] : 0);
        double* gslot = gout + desc[q + 5];
        F ud[32], kk[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { ud[i] = up[i]; kk[i] = has_k ? kp[i] : (F)0; }
        switch (ja * 8 + jb) {
          TCMI_BG2_CASE(0, 1) TCMI_BG2_CASE(0, 2) TCMI_BG2_CASE(0, 3)
          TCMI_BG2_CASE(1, 2) TCMI_BG2_CASE(1, 3) TCMI_BG2_CASE(2, 3)
          default: break;
        }
        q += 6;
      } else if (op == TCMI_OP_DIAG) {
        // {3, nA, nB, nC, base, maskA[nA], maskB[nB], jB[nB], rmaskC[nC], gsA[nA], gsB[nB], gsC[nC]}
        // The table holds the FORWARD coefficients; the inverse phase is applied (conjugate).
        const int nA = desc[q + 1], nB = desc[q + 2], nC = desc[q + 3];
        const KPtr<F> cf = ptab + desc[q + 4];
        q += 5;
        const KInt mA = desc + q, mB = desc + q + nA, jB = desc + q + nA + nB, mC = desc + q + nA + 2 * nB;
        const KInt gA = mC + nC, gB = gA + nA, gC = gB + nB;
        q += 2 * nA + 3 * nB + 2 * nC;
        const uint32_t tidx = wg_base | tphys;
        // w[r] = Im(conj(lambda) psi): every term's gradient is a signed sum of w
        F w[NR];
        F w0 = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          w[r] = fma_<F>(l[r].x, a[r].y, -l[r].y * a[r].x);
          w0 += w[r];
        }
        F wj[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          F s = 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) s += ((r >> j) & 1) ? -w[r] : w[r];
          wj[j] = s;
        }
        double phi = 0.0;
#pragma unroll 1
        for (int e = 0; e < nA; ++e) {
          const uint32_t m = (uint32_t)mA[e];
          const bool neg = __popc(tidx & m) & 1;
          const double c = (double)cf[e];
          phi += neg ? -c : c;
          const int gs = gA[e];
          if (gs >= 0) {  // wave-uniform
            F v = wave_sum_uniform(neg ? -w0 : w0);
            if ((tid & 63) == 0) atomicAdd(gout + gs, (double)v);
          }
        }
        double cj[R];
#pragma unroll
        for (int j = 0; j < R; ++j) cj[j] = 0.0;
#pragma unroll 1
        for (int e = 0; e < nB; ++e) {
          const uint32_t m = (uint32_t)mB[e];
          const int jj = jB[e];
          const bool neg = __popc(tidx & m) & 1;
          const double c = (double)cf[nA + e];
          const double sgn = neg ? -c : c;
          F wsel = 0;
#pragma unroll
          for (int j = 0; j < R; ++j) {
            cj[j] += (j == jj) ? sgn : 0.0;
            wsel = (j == jj) ? wj[j] : wsel;
          }
          const int gs = gB[e];
          if (gs >= 0) {
            F v = wave_sum_uniform(neg ? -wsel : wsel);
            if ((tid & 63) == 0) atomicAdd(gout + gs, (double)v);
          }
        }
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
          const uint32_t rmask = (uint32_t)mC[e];
          const F c = cf[nA + nB + e];
          F s = 0;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const bool neg = __popc((uint32_t)r & rmask) & 1;
            ph[r] += neg ? -c : c;
            s += neg ? -w[r] : w[r];
          }
          const int gs = gC[e];
          if (gs >= 0) {
            F v = wave_sum_uniform(s);
            if ((tid & 63) == 0) atomicAdd(gout + gs, (double)v);
          }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          F s, c;
          sincos_turns<F>(ph[r], &s, &c);
          const C v = a[r], u = l[r];
          a[r].x = v.x * c + v.y * s;   // multiply by exp(-i phi)
          a[r].y = v.y * c - v.x * s;
          l[r].x = u.x * c + u.y * s;
          l[r].y = u.y * c - u.x * s;
        }
      } else {
        break;
      }
    }
    pc += TCMI_RR_WORDS + rr[1];

    if (k < nrounds - 1) {
      const uint32_t tslot = xor_masks<LT>(tid, rr + 40);
      uint32_t wsm[R];
#pragma unroll
      for (int j = 0; j < R; ++j) wsm[j] = (uint32_t)rr[34 + j];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t s = tslot ^ reg_mask<R>(r, wsm);
        lds[s] = a[r];
        lds2[s] = l[r];
      }
      __syncthreads();
    } else {
      C* __restrict__ da = psi + (wg_base | tphys);
      C* __restrict__ dl = lam + (wg_base | tphys);
#pragma unroll
      for (int r = 0; r < NR; r += VEC) {
        const uint32_t off = reg_mask<R>(r, rpm);
        if constexpr (VEC == 2) {
          float4 v, w;
          v.x = a[r].x; v.y = a[r].y; v.z = a[r + 1].x; v.w = a[r + 1].y;
          w.x = l[r].x; w.y = l[r].y; w.z = l[r + 1].x; w.w = l[r + 1].y;
          *reinterpret_cast<float4*>(da + off) = v;
          *reinterpret_cast<float4*>(dl + off) = w;
        } else {
          da[off] = a[r];
          dl[off] = l[r];
        }
      }
    }
  }
}

// out[idx] = sum_t w_t * i^{ny_t} * (-1)^{popc((idx ^ xm_t) & zm_t)} * in[idx ^ xm_t]
// (= (sum_t w_t P_t) |in>): the cotangent of <psi|H|psi> w.r.t. psi, reference circuit.py:899-902.
// Tiled form: a workgroup stages TILE consecutive amplitudes in LDS; every term whose X mask stays inside
// the tile (the low log2(TILE) qubits) gathers its partner from LDS, only the other terms touch global
// memory again (the flat version re-read the state once per distinct X mask through L2: 27 x for the
// 55-term TFIM cotangent).  Terms outer (mask / weight are scalar loads), the thread's EPT elements inner.
template <typename F, int TILE>
__global__ __launch_bounds__(256) void pauli_sum_kernel(const typename Cx<F>::type* __restrict__ in,
                                                        typename Cx<F>::type* __restrict__ out, long long stride,
                                                        unsigned long long nelem, const int* __restrict__ terms,
                                                        int nterms, const double* __restrict__ w, long long wstride) {
  using C = typename Cx<F>::type;
  constexpr int EPT = TILE / 256;
  __shared__ C tile[TILE];
  in += (long long)blockIdx.y * stride;
  out += (long long)blockIdx.y * stride;
  const KPtr<double> wk = (KPtr<double>)(w + (long long)blockIdx.y * wstride);
  const KInt tm = (KInt)terms;
  const unsigned long long ntiles = (nelem + TILE - 1) / TILE;
  for (unsigned long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const unsigned long long base = tl * TILE;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const unsigned loc = threadIdx.x + 256 * k;
      C v;
      v.x = 0;
      v.y = 0;
      if (base + loc < nelem) v = in[base + loc];
      tile[loc] = v;
    }
    __syncthreads();
    F re[EPT], im[EPT];
    C v[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      re[k] = 0;
      im[k] = 0;
      v[k].x = 0;
      v[k].y = 0;
    }
    uint32_t last_xm = 0xffffffffu;
    for (int t = 0; t < nterms; ++t) {
      const uint32_t xm = (uint32_t)tm[3 * t], zm = (uint32_t)tm[3 * t + 1];
      const int ny = tm[3 * t + 2] & 3;
      const F c0 = (F)wk[t];
      if (xm != last_xm) {  // terms are sorted by X mask
        last_xm = xm;
        if (xm < (uint32_t)TILE && base + TILE <= nelem) {
#pragma unroll
          for (int k = 0; k < EPT; ++k) v[k] = tile[(threadIdx.x + 256 * k) ^ xm];
        } else {
#pragma unroll
          for (int k = 0; k < EPT; ++k) {
            const unsigned long long idx = base + threadIdx.x + 256 * k;
            if (idx < nelem) v[k] = in[idx ^ xm];
          }
        }
      }
      // sign = parity((idx ^ xm) & zm), idx = base + tid + 256 k: the thread's part once per term, the rest is
      // wave-uniform (scalar code); i^ny folded into which component feeds which sum -> two FMAs per element and term
      // (the per-element popcount / select / swizzle form made this kernel VALU bound: 7.0 -> 5.6 ms per n = 28 state)
      constexpr uint32_t TM = (uint32_t)TILE - 1u;   // index bits inside the tile
      const F ct = (__popc((uint32_t)threadIdx.x & zm) & 1) ? -c0 : c0;
      const uint32_t ubase = (uint32_t)base;
      const int par0 = __popc(((ubase ^ xm) & ~TM) & zm) + __popc((xm & TM) & zm);
      if (ny == 0 || ny == 2) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const bool neg = ((par0 + __popc(((uint32_t)(256 * k)) & zm)) & 1) != (ny == 2);   // uniform
          const F c = neg ? -ct : ct;
          re[k] = fma_<F>(c, v[k].x, re[k]);
          im[k] = fma_<F>(c, v[k].y, im[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const bool neg = ((par0 + __popc(((uint32_t)(256 * k)) & zm)) & 1) != (ny == 3);   // uniform
          const F c = neg ? -ct : ct;   // i v = (-v.y, v.x)
          re[k] = fma_<F>(-c, v[k].y, re[k]);
          im[k] = fma_<F>(c, v[k].x, im[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const unsigned long long idx = base + threadIdx.x + 256 * k;
      if (idx < nelem) {
        C o;
        o.x = re[k];
        o.y = im[k];
        out[idx] = o;
      }
    }
  }
}

// adjoint builder: U^dagger tables and K = (dU/dtheta) U^dagger for parametrised gates
template <typename F>
__global__ void build_adjoint_kernel(const int* __restrict__ ginfo, int nrec, const double* __restrict__ cpool,
                                     const F* __restrict__ params, long long pstride, F* __restrict__ ptab,
                                     long long tstride, int batch) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (g >= nrec || b >= batch) return;
  const int* rec = ginfo + 8 * g;
  const int kind = rec[0], slot = rec[1], pidx = rec[2], dim = rec[3], off = rec[4];
  const double theta = (double)params[(long long)b * pstride + pidx];
  const double kap = cpool[off];
  const double ang = fma(kap, theta, cpool[off + 1]);
  F* out = ptab + (long long)b * tstride + slot;
  if (kind == TCMI_BK_COEF) {
    out[0] = (F)(ang - rint(ang));
    return;
  }
  if (kind == TCMI_BK_PHASE) {
    // one entry of a phase table of OP_DIAGF (same record as the forward builder, tcmi_vm.hip)
    phase_entry<F>(cpool + off, dim, rec[5], params + (long long)b * pstride, rec[6] != 0, out);
    return;
  }
  double s, c;
  sincos(ang, &s, &c);
  const int nn = dim * dim;
  const double* c0 = cpool + off + 2;
  const double* c1 = c0 + 2 * nn;
  const double* c2 = c1 + 2 * nn;
  // U and dU (dim <= 4)
  double ur[16], ui[16], dr[16], di[16];
  for (int i = 0; i < nn; ++i) {
    ur[i] = c0[2 * i] + c * c1[2 * i] + s * c2[2 * i];
    ui[i] = c0[2 * i + 1] + c * c1[2 * i + 1] + s * c2[2 * i + 1];
    dr[i] = kap * (-s * c1[2 * i] + c * c2[2 * i]);
    di[i] = kap * (-s * c1[2 * i + 1] + c * c2[2 * i + 1]);
  }
  if (kind == TCMI_BK_UDAG && rec[5]) {
    // three-shear form of U^dagger (see build_kernel in tcmi_vm.hip): U^dagger[0][0] = conj(U[0][0]),
    // U^dagger[1][0] = conj(U[0][1]); the sign is common to psi and lambda and is not needed
    double a = ur[0];
    double cc = (rec[5] == 1) ? ur[1] : -ui[1];
    for (int i = 3; i < 8; ++i) out[i] = (F)0;
    if (rec[6] && fabs(a) >= TCMI_SHEAR2_CMIN) {
      // two-shear form U^dagger = diag(a, 1 / a) L(v) S(u), as in build_kernel; the kernel shears lambda in the
      // other order and the plan's scale term carries diag(a, 1 / a) for psi, its reciprocal for lambda
      out[0] = (F)((rec[5] == 1 ? -cc : cc) / a);
      out[1] = (F)(cc * a);
      out[2] = (F)1;
      out[3] = (F)2;
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
  if (kind == TCMI_BK_UDAG) {
    for (int r = 0; r < dim; ++r)
      for (int q = 0; q < dim; ++q) {
        out[2 * (r * dim + q)] = (F)ur[q * dim + r];
        out[2 * (r * dim + q) + 1] = (F)(-ui[q * dim + r]);
      }
  } else if (kind == TCMI_BK_KMAT) {
    // K = dU * U^dagger : K[r][q] = sum_m dU[r][m] * conj(U[q][m])
    for (int r = 0; r < dim; ++r)
      for (int q = 0; q < dim; ++q) {
        double kr = 0, ki = 0;
        for (int m = 0; m < dim; ++m) {
          const double ar = dr[r * dim + m], ai = di[r * dim + m];
          const double br = ur[q * dim + m], bi = -ui[q * dim + m];
          kr += ar * br - ai * bi;
          ki += ar * bi + ai * br;
        }
        out[2 * (r * dim + q)] = (F)kr;
        out[2 * (r * dim + q) + 1] = (F)ki;
      }
  }
}

// <a|b> per batch element: double accumulation, wave reduction, one f64 atomic pair per wave into one of
// `copies` replicated accumulators (same-address atomics from every wave would serialise in one L2 channel).
template <typename F>
__global__ void vdot_kernel(const typename Cx<F>::type* __restrict__ a, const typename Cx<F>::type* __restrict__ b,
                            double* __restrict__ out, long long stride, unsigned long long nelem, int copies,
                            long long out_batch_stride) {
  using Ct = typename Cx<F>::type;
  const long long bi = blockIdx.y;
  a += bi * stride;
  b += bi * stride;
  double re = 0, im = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nelem;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const Ct x = a[i], y = b[i];
    re += (double)x.x * y.x + (double)x.y * y.y;
    im += (double)x.x * y.y - (double)x.y * y.x;
  }
  re = wave_sum<double>(re);
  im = wave_sum<double>(im);
  if ((threadIdx.x & 63) == 0) {
    const int c = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % (unsigned)copies);
    double* o = out + bi * out_batch_stride + 2 * c;
    atomicAdd(o, re);
    atomicAdd(o + 1, im);
  }
}

// second-generation complex64 adjoint pass (tcmi_adjoint2.hip); -1 = no variant for this (R, LT)
int run_adjoint2_c64(void* psi, void* lam, long long stride, int batch, int n, int R, int LT, const int* desc,
                     const void* ctab, const void* ptab, long long ptab_stride, double* gout, long long gout_stride,
                     int gcopies, long long gcopy_stride, hipStream_t st);

}  // namespace tcmi

// ---- C ABI ------------------------------------------------------------------------------------
extern "C" const char* tcmi_last_error(void);
extern "C" int tcmi_set_error_(int code, const char* msg);

template <typename F, int R, int LT>
static int launch_adjoint(void* psi, void* lam, long long stride, int batch, int n, const int* desc,
                          const void* ctab, const void* ptab, long long ptab_stride, double* gout,
                          long long gout_stride, int gcopies, long long gcopy_stride, hipStream_t st) {
  using C = typename tcmi::Cx<F>::type;
  constexpr int T = R + LT;
  if (n < T) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_run_adjoint_pass: n smaller than the tile");
  const size_t lds = 2 * (sizeof(C) << T);
  auto kern = tcmi::adjoint_kernel<F, R, LT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  }
  dim3 grid(1u << (n - T), (unsigned)batch, 1), block(1u << LT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<C*>(psi), reinterpret_cast<C*>(lam), stride,
                     desc, reinterpret_cast<const F*>(ctab), reinterpret_cast<const F*>(ptab), ptab_stride,
                     gout, gout_stride, gcopies, gcopy_stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

extern "C" {

int tcmi_run_adjoint_pass(void* psi, void* lam, long long state_stride, int batch, int n, int R, int LT,
                          const int* desc_dev, const void* ctab_dev, const void* ptab_dev,
                          long long ptab_stride, double* gout_dev, long long gout_stride, int gcopies,
                          long long gcopy_stride, int dtype, int opset, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!psi || !lam || !desc_dev || !gout_dev || batch < 1 || n > 32 || gcopies < 1 ||
      (opset != TCMI_OPSET_GENERIC && opset != TCMI_OPSET_PACKED))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_run_adjoint_pass: bad argument");
  if (opset == TCMI_OPSET_PACKED) {
    // the packed-f32 kernel; its plans hold one-qubit gate ops and table-form diagonal flushes only (host-checked)
    if (dtype != TCMI_C64) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_run_adjoint_pass: the packed op set is complex64 only");
    const int rc = tcmi::run_adjoint2_c64(psi, lam, state_stride, batch, n, R, LT, desc_dev, ctab_dev, ptab_dev, ptab_stride,
                                          gout_dev, gout_stride, gcopies, gcopy_stride, st);
    if (rc == TCMI_OK) return rc;
    if (rc != -1) return tcmi_set_error_(TCMI_ERR_HIP, "adjoint2_kernel launch failed");
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_run_adjoint_pass: unsupported packed (R, LT) variant");
  }
#define TCMI_CASE(FT, RR, LL) \
  if (R == RR && LT == LL)    \
    return launch_adjoint<FT, RR, LL>(psi, lam, state_stride, batch, n, desc_dev, ctab_dev, ptab_dev, ptab_stride, gout_dev, gout_stride, gcopies, gcopy_stride, st);
  if (dtype == TCMI_C64) {
    TCMI_CASE(float, 4, 8)
    TCMI_CASE(float, 2, 6)
  } else if (dtype == TCMI_C128) {
    TCMI_CASE(double, 3, 8)
    TCMI_CASE(double, 2, 6)
  }
#undef TCMI_CASE
  return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_run_adjoint_pass: unsupported (dtype, R, LT) variant");
}

int tcmi_build_adjoint_tables(const int* ginfo_dev, int nrec, const double* cpool_dev, const void* params_dev,
                              long long params_stride, void* ptab_dev, long long ptab_stride, int batch,
                              int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (nrec == 0) return TCMI_OK;
  if (!ginfo_dev || !cpool_dev || !params_dev || !ptab_dev || batch < 1)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_build_adjoint_tables: bad argument");
  dim3 block(128, 1, 1), grid((nrec + 127) / 128, batch, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::build_adjoint_kernel<float>, grid, block, 0, st, ginfo_dev, nrec, cpool_dev,
                       reinterpret_cast<const float*>(params_dev), params_stride,
                       reinterpret_cast<float*>(ptab_dev), ptab_stride, batch);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::build_adjoint_kernel<double>, grid, block, 0, st, ginfo_dev, nrec, cpool_dev,
                       reinterpret_cast<const double*>(params_dev), params_stride,
                       reinterpret_cast<double*>(ptab_dev), ptab_stride, batch);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_build_adjoint_tables: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_apply_pauli_sum(const void* in, void* out, long long state_stride, int batch, int n,
                         const int* terms_dev, int nterms, const double* weights_dev,
                         long long weights_stride, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || !terms_dev || !weights_dev || batch < 1 || n < 0 || n > 32 || nterms < 0)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_apply_pauli_sum: bad argument");
  const unsigned long long nelem = 1ull << n;
  const unsigned long long tiles64 = (nelem + 4095) / 4096, tiles128 = (nelem + 2047) / 2048;
  const unsigned long long tiles = dtype == TCMI_C64 ? tiles64 : tiles128;
  unsigned gx = (unsigned)(tiles > 4096 ? 4096 : tiles);
  dim3 grid(gx, batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL((tcmi::pauli_sum_kernel<float, 4096>), grid, block, 0, st, reinterpret_cast<const float2*>(in),
                       reinterpret_cast<float2*>(out), state_stride, nelem, terms_dev, nterms, weights_dev,
                       weights_stride);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL((tcmi::pauli_sum_kernel<double, 2048>), grid, block, 0, st, reinterpret_cast<const double2*>(in),
                       reinterpret_cast<double2*>(out), state_stride, nelem, terms_dev, nterms, weights_dev,
                       weights_stride);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_apply_pauli_sum: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_vdot(const void* a, const void* b, double* out, long long state_stride, int batch, int n, int copies,
              long long out_batch_stride, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!a || !b || !out || batch < 1 || n < 0 || n > 34 || copies < 1)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_vdot: bad argument");
  const unsigned long long nelem = 1ull << n;
  unsigned gx = (unsigned)((nelem + 1023) / 1024 > 4096 ? 4096 : (nelem + 1023) / 1024);
  dim3 grid(gx, batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::vdot_kernel<float>, grid, block, 0, st, reinterpret_cast<const float2*>(a),
                       reinterpret_cast<const float2*>(b), out, state_stride, nelem, copies, out_batch_stride);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::vdot_kernel<double>, grid, block, 0, st, reinterpret_cast<const double2*>(a),
                       reinterpret_cast<const double2*>(b), out, state_stride, nelem, copies, out_batch_stride);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_vdot: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

}  // extern "C"
