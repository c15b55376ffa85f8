// tcmi pairwise contraction engine (K1b / K2 of SURVEY.md section 2.2) for gfx950.
//
// tn.contract_between(a, b) == backend.tensordot(a, b, axes) (reference tensorcircuit/cons.py:948 via
// tensornetwork) is lowered on the host (tcmi/tn.py) to
//     permute(A) -> [free_A..., contracted...]   = M x K row-major
//     permute(B) -> [contracted..., free_B...]   = K x N row-major
//     C = A . B   (complex GEMM)                 = axes free_A + free_B, no output permute
// This file holds the two device kernels: a bit-permutation copy for tensors whose axes all have
// dimension 2 (every circuit tensor network) and a complex GEMM on the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32: exact f32 FMA, 3 real MFMAs per complex k-pair, Gauss's 3-product form).

#include "tcmi_dev.h"

extern "C" int tcmi_set_error_(int code, const char* msg);

namespace tcmi {

// out[o] = in[src(o)], src(o) = OR_b ((o >> b) & 1) << srcbit[b]   (axis permutation of a [2]^rank tensor)
template <typename C>
__global__ void permute_bits_kernel(const C* __restrict__ in, C* __restrict__ out, int rank,
                                    const int* __restrict__ srcbit, long long batch_stride) {
  const KInt sb = (KInt)srcbit;
  in += (long long)blockIdx.y * batch_stride;
  out += (long long)blockIdx.y * batch_stride;
  const unsigned long long nelem = 1ull << rank;
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long o = i0; o < nelem; o += step) {
    unsigned long long s = 0;
    for (int b = 0; b < rank; ++b) s |= ((o >> b) & 1ull) << sb[b];
    out[o] = in[s];
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[M x N] = A[M x K] . B[K x N], row-major interleaved complex64, one 64x64 tile per workgroup,
// 4 waves x (32x32 MFMA tile), K step 8.  LDS holds planar (re / im) operand tiles, k-major.
//  * 3 real products per complex product (Gauss): P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi);
//    C_re = P1 - P2, C_im = P3 - P1 - P2 — 3 MFMAs per k-pair instead of 4 on the f32 matrix pipe,
//    which is what bounds this kernel; the two operand sums are VALU adds on the fragments.
//  * row pitch 68 floats: 8-byte aligned rows for the vector (ds_write_b64) loaders of k-major
//    operands, and 2*68 = 8 (mod 32) makes the transposing loader of a row-major A hit 32 distinct
//    banks per half-wave (r01c profile: 25 % of LDS cycles were bank conflicts with pitch 65).
#define TCMI_BM 64
#define TCMI_BN 64
#define TCMI_BK 8
#define TCMI_CBK 16  // K step of the complex64 kernel: one barrier pair per 8 MFMA k-pairs
#define TCMI_LDP 68

template <bool TRANS_A>
__global__ __launch_bounds__(256, 6) void cgemm_mfma_kernel(const float2* __restrict__ A,
                                                          const float2* __restrict__ B,
                                                          float2* __restrict__ C, int M, int N, int K,
                                                          long long sA, long long sB, long long sC, int ksplit,
                                                          int kchunk) {
  __shared__ __attribute__((aligned(16))) float As_re[TCMI_CBK][TCMI_LDP], As_im[TCMI_CBK][TCMI_LDP];
  __shared__ __attribute__((aligned(16))) float Bs_re[TCMI_CBK][TCMI_LDP], Bs_im[TCMI_CBK][TCMI_LDP];
  // split-K (ksplit > 1): blockIdx.z = batch * ksplit + chunk, every chunk adds its partial product into a
  // zeroed C with f32 atomics -- few-tile products with a long K (the closing steps of a contraction tree:
  // 32 x 32 outputs over K = 2^20) would otherwise run on a handful of CUs
  const int bz = (int)blockIdx.z / ksplit, ks = (int)blockIdx.z - bz * ksplit;
  const int kbeg = ks * kchunk, kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  A += (long long)bz * sA;
  B += (long long)bz * sB;
  C += (long long)bz * sC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long long m0 = (long long)blockIdx.y * TCMI_BM, n0 = (long long)blockIdx.x * TCMI_BN;
  f32x16 p1 = {0}, p2 = {0}, p3 = {0};
  // loader coordinates: A tile 64 rows x 8 k (two complex per thread along k);
  //                     B tile 8 k x 64 cols (two complex per thread along n)
  // TRANS_A: A is stored [K][M] (k-major), loaded like B (coalesced along m)
  const int ai = TRANS_A ? (tid & 31) * 2 : tid >> 2, ak0 = TRANS_A ? tid >> 5 : (tid & 3) * 2;
  const int bk0 = tid >> 5, bj = (tid & 31) * 2;
  // 16-byte global loads need even leading dimensions (and 16-byte aligned batch bases)
  const bool vecA = TRANS_A ? ((M & 1) == 0 && (sA & 1) == 0) : ((K & 1) == 0 && (sA & 1) == 0);
  const bool vecB = (N & 1) == 0 && (sB & 1) == 0;
  for (int k0 = kbeg; k0 < kend; k0 += TCMI_CBK) {
#pragma unroll
    for (int h = 0; h < TCMI_CBK / 8; ++h) {
      const int ak = ak0 + 8 * h, bk = bk0 + 8 * h;
      float2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
      if constexpr (TRANS_A) {
        const long long r = m0 + ai;
        const int kk = k0 + ak;
        if (kk < kend) {
          if (vecA && r + 1 < M) {
            const float4 t = *reinterpret_cast<const float4*>(A + (long long)kk * M + r);
            v0.x = t.x; v0.y = t.y; v1.x = t.z; v1.y = t.w;
          } else {
            if (r < M) v0 = A[(long long)kk * M + r];
            if (r + 1 < M) v1 = A[(long long)kk * M + r + 1];
          }
        }
        *reinterpret_cast<float2*>(&As_re[ak][ai]) = make_float2(v0.x, v1.x);
        *reinterpret_cast<float2*>(&As_im[ak][ai]) = make_float2(v0.y, v1.y);
      } else {
        const long long r = m0 + ai;
        const int kk = k0 + ak;
        if (r < M) {
          if (vecA && kk + 1 < kend) {
            const float4 t = *reinterpret_cast<const float4*>(A + r * K + kk);
            v0.x = t.x; v0.y = t.y; v1.x = t.z; v1.y = t.w;
          } else {
            if (kk < kend) v0 = A[r * K + kk];
            if (kk + 1 < kend) v1 = A[r * K + kk + 1];
          }
        }
        As_re[ak][ai] = v0.x;
        As_im[ak][ai] = v0.y;
        As_re[ak + 1][ai] = v1.x;
        As_im[ak + 1][ai] = v1.y;
      }
      float2 w0 = {0.f, 0.f}, w1 = {0.f, 0.f};
      const long long c = n0 + bj;
      const int kb = k0 + bk;
      if (kb < kend) {
        if (vecB && c + 1 < N) {
          const float4 t = *reinterpret_cast<const float4*>(B + (long long)kb * N + c);
          w0.x = t.x; w0.y = t.y; w1.x = t.z; w1.y = t.w;
        } else {
          if (c < N) w0 = B[(long long)kb * N + c];
          if (c + 1 < N) w1 = B[(long long)kb * N + c + 1];
        }
      }
      *reinterpret_cast<float2*>(&Bs_re[bk][bj]) = make_float2(w0.x, w1.x);
      *reinterpret_cast<float2*>(&Bs_im[bk][bj]) = make_float2(w0.y, w1.y);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TCMI_CBK; kk += 2) {
      const int kr = kk + (lane >> 5);
      const float are = As_re[kr][wr * 32 + (lane & 31)], aim = As_im[kr][wr * 32 + (lane & 31)];
      const float bre = Bs_re[kr][wc * 32 + (lane & 31)], bim = Bs_im[kr][wc * 32 + (lane & 31)];
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(are, bre, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, bim, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(are + aim, bre + bim, p3, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const long long col = n0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const long long row = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) {
      float2 o;
      o.x = p1[reg] - p2[reg];
      o.y = p3[reg] - p1[reg] - p2[reg];
      if (ksplit > 1) {
        atomicAdd(&C[row * N + col].x, o.x);
        atomicAdd(&C[row * N + col].y, o.y);
      } else {
        C[row * N + col] = o;
      }
    }
  }
}

// ---- the same product for a k-major A ([K][M], the cut join's L^T), software-pipelined with LDS-DMA -------------------
// cgemm_mfma_kernel stages its operand tiles through VGPRs and waits for every load (single buffer, two barriers per
// K step): a workgroup's MFMA phase (24 x 64 cycles per wave and K step) never overlaps its own loads, and the other
// resident workgroups fill in only partly -- 0.71-0.74 of the f32 MFMA peak on the join GEMM (profiles/r02o_*).  Here
// the tiles go global -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers, no
// ds_write pass) into a ring of DMA_STAGES buffers; the loads of K step i + DMA_STAGES - 1 are issued before the MFMAs
// of step i, there is ONE raw s_barrier per K step and the only wait on the memory counter is a counted vmcnt that
// leaves the younger stages in flight across the barrier.
//   * LDS tile = the global layout: [16 k][64 m] interleaved (re, im), 512 B per k row; an MFMA operand fragment is ONE
//     ds_read_b64 per lane (lanes 0-31: 256 contiguous bytes of row k, lanes 32-63: row k + 1 -- conflict-free).
//   * Gauss's 3-product form as in cgemm_mfma_kernel (P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi)).
//   * blockIdx -> tile mapping is XCD-aware: consecutive workgroups go to the 8 XCDs round-robin, so workgroup id L
//     belongs to XCD L % 8 and is that XCD's (L / 8)-th tile; every XCD walks its own 16-column strip of C row by
//     row, i.e. its L2 holds one A row-tile and 16 B column-tiles (2.1 MB of 4 MB) instead of the whole operands.
// Requirements (host checks): M, N multiples of 64, K a multiple of 16, 16-byte aligned batch bases.
#define TCMI_DMA_BK 16
#define TCMI_DMA_STAGE_BYTES (2 * TCMI_DMA_BK * 64 * 8)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int STAGES>
__global__ __launch_bounds__(256, 3) void cgemm_dma_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                        float2* __restrict__ C, int M, int N, int K, long long sA,
                                                        long long sB, long long sC, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile walk (see above); falls back to the plain order when the grid does not split into 8 strips
  int tx, ty;
  {
    const unsigned L = blockIdx.x, ntile = (unsigned)tiles_x * (unsigned)tiles_y;
    if ((tiles_x & 7) == 0 && (ntile & 7) == 0) {
      const unsigned xcd = L & 7u, j = L >> 3, sw = (unsigned)tiles_x >> 3;   // strip width in tiles
      tx = (int)(xcd * sw + j % sw);
      ty = (int)(j / sw);
    } else {
      tx = (int)(L % (unsigned)tiles_x);
      ty = (int)(L / (unsigned)tiles_x);
    }
  }
  const long long m0 = (long long)ty * 64, n0 = (long long)tx * 64;
  A += (long long)blockIdx.y * sA + m0;
  B += (long long)blockIdx.y * sB + n0;
  C += (long long)blockIdx.y * sC;
  // DMA source of this lane: k row (4 wave + 2 h + lane / 32) of the stage, 16 bytes = two complex at column 2 (lane % 32)
  const float2* ag = A + (long long)(4 * wave + (lane >> 5)) * M + 2 * (lane & 31);
  const float2* bg = B + (long long)(4 * wave + (lane >> 5)) * N + 2 * (lane & 31);
  const long long a2 = 2ll * M, b2 = 2ll * N, a16 = 16ll * M, b16 = 16ll * N;
  const int nk = K / TCMI_DMA_BK;
#define TCMI_DMA_ISSUE(KT, ST)                                                                                         \
  {                                                                                                                    \
    char* sb_ = dsm + (ST) * TCMI_DMA_STAGE_BYTES + wave * 2048;                                                       \
    const float2* ap_ = ag + (long long)(KT) * a16;                                                                    \
    const float2* bp_ = bg + (long long)(KT) * b16;                                                                    \
    __builtin_amdgcn_global_load_lds((gptr_t)ap_, (lptr_t)sb_, 16, 0, 0);                                              \
    __builtin_amdgcn_global_load_lds((gptr_t)(ap_ + a2), (lptr_t)(sb_ + 1024), 16, 0, 0);                             \
    __builtin_amdgcn_global_load_lds((gptr_t)bp_, (lptr_t)(sb_ + 8192), 16, 0, 0);                                    \
    __builtin_amdgcn_global_load_lds((gptr_t)(bp_ + b2), (lptr_t)(sb_ + 8192 + 1024), 16, 0, 0);                      \
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) TCMI_DMA_ISSUE(s, s)
  f32x16 p1 = {0}, p2 = {0}, p3 = {0};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)dsm;
  // fragment addresses inside a stage: row (lane / 32) of the k pair, element wr * 32 + lane % 32 (A) / wc * 32 + ... (B)
  const int fa = (lane >> 5) * 512 + (wr * 32 + (lane & 31)) * 8;
  const int fb = 8192 + (lane >> 5) * 512 + (wc * 32 + (lane & 31)) * 8;
  int st = 0;
  for (int i = 0; i < nk; ++i) {
    // this wave's DMA pieces of stage i have landed (younger stages stay in flight), then everybody's have
    if (i + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (STAGES - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the buffer of stage i - 1 is free now (every wave finished its MFMAs on it before the barrier): refill it
    if (i + STAGES - 1 < nk) {
      const int sn = (st == 0) ? STAGES - 1 : st - 1;
      TCMI_DMA_ISSUE(i + STAGES - 1, sn)
    }
    // operand fragments of the whole K step: 16 ds_read_b64 issued together from inline asm (the compiler would put a
    // vmcnt(0) in front of any LDS read it can see while an LDS-DMA is outstanding -- that wait would drain the
    // prefetch just issued), then an uninterrupted MFMA stream (anything issued between the MFMAs of a wave costs
    // matrix-pipe time: reads interleaved with the MFMAs measured 7 % slower)
    const uint32_t sa_ = lds0 + (uint32_t)(st * TCMI_DMA_STAGE_BYTES) + (uint32_t)fa;
    const uint32_t sb_ = lds0 + (uint32_t)(st * TCMI_DMA_STAGE_BYTES) + (uint32_t)fb;
    v2f fa_[TCMI_DMA_BK / 2], fb_[TCMI_DMA_BK / 2];
#pragma unroll
    for (int q = 0; q < TCMI_DMA_BK / 2; ++q) {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa_[q]) : "v"(sa_), "n"(q * 1024));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb_[q]) : "v"(sb_), "n"(q * 1024));
    }
#pragma unroll
    for (int q = 0; q < TCMI_DMA_BK / 2; ++q) {
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa_[q]), "+v"(fb_[q]) : "n"(TCMI_DMA_BK - 2 - 2 * q));
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[q].x, fb_[q].x, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[q].y, fb_[q].y, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[q].x + fa_[q].y, fb_[q].x + fb_[q].y, p3, 0, 0, 0);
    }
    st = (st + 1 == STAGES) ? 0 : st + 1;
  }
#undef TCMI_DMA_ISSUE
  const long long col = n0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const long long row = m0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    float2 o;
    o.x = p1[reg] - p2[reg];
    o.y = p3[reg] - p1[reg] - p2[reg];
    C[row * N + col] = o;
  }
}

// ---- 128 x 128 workgroup tile, 64 x 64 per wave (2 x 2 MFMA tiles, three Gauss products each: 12 accumulators) --------
// scripts/ubench/mfma_gemm_steps.hip: the f32 matrix pipe issues a bare MFMA chain at 65 cycles per v_mfma_f32_32x32x2,
// but every other instruction of the issuing wave costs it 4-6 cycles -- the 64 x 64 tile's K step (24 MFMAs, 16 LDS
// reads, 16 adds, 4 DMA pieces, a barrier) runs at 72-77 cycles per MFMA in isolation and 80 in the kernel.  The
// register tile here needs ONE ds_read_b128 per operand and k pair for 12 MFMAs (a lane's 16 bytes hold two
// neighbouring rows / columns of the same k, which become the two A / B fragments), 4 adds and 2/3 of a DMA piece:
// 0.6 instead of 1.5 other instructions per MFMA.  Stage = A [16 k][128 m] + B [16 k][128 n] interleaved complex =
// 32 KiB, two stages, two workgroups per CU (accumulators in AGPRs: 192 + 48 fragment registers per lane).
#define TCMI_T128_STAGE_BYTES (2 * TCMI_DMA_BK * 128 * 8)
typedef float v4f_ __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void cgemm_dma128_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                              float2* __restrict__ C, int M, int N, int K, long long sA,
                                                              long long sB, long long sC, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tx, ty;
  {
    const unsigned L = blockIdx.x, ntile = (unsigned)tiles_x * (unsigned)tiles_y;
    if ((tiles_x & 7) == 0 && (ntile & 7) == 0) {
      const unsigned xcd = L & 7u, j = L >> 3, sw = (unsigned)tiles_x >> 3;
      tx = (int)(xcd * sw + j % sw);
      ty = (int)(j / sw);
    } else {
      tx = (int)(L % (unsigned)tiles_x);
      ty = (int)(L / (unsigned)tiles_x);
    }
  }
  const long long m0 = (long long)ty * 128, n0 = (long long)tx * 128;
  A += (long long)blockIdx.y * sA + m0;
  B += (long long)blockIdx.y * sB + n0;
  C += (long long)blockIdx.y * sC;
  // DMA: piece = one k row of a tile (128 complex = 1 KiB = 64 lanes x 16 bytes); wave w moves rows 4w .. 4w + 3 of A and B
  const float2* ag = A + (long long)(4 * wave) * M + 2 * lane;
  const float2* bg = B + (long long)(4 * wave) * N + 2 * lane;
  const int nk = K / TCMI_DMA_BK;
#define TCMI_DMA_ISSUE(KT, ST)                                                                                   \
  {                                                                                                              \
    char* sb_ = dsm + (ST) * TCMI_T128_STAGE_BYTES + wave * 4096;                                                \
    const float2* ap_ = ag + (long long)(KT) * 16 * M;                                                           \
    const float2* bp_ = bg + (long long)(KT) * 16 * N;                                                           \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                              \
      __builtin_amdgcn_global_load_lds((gptr_t)(ap_ + (long long)r * M), (lptr_t)(sb_ + r * 1024), 16, 0, 0);    \
      __builtin_amdgcn_global_load_lds((gptr_t)(bp_ + (long long)r * N), (lptr_t)(sb_ + 16384 + r * 1024), 16, 0, 0); \
    }                                                                                                            \
  }
  TCMI_DMA_ISSUE(0, 0)
  f32x16 acc[2][2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.f;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)dsm;
  // fragment address: k row (lane / 32) of the pair, 16 bytes = elements 2 (lane % 32), + 1 of the wave's 64 rows / columns
  const uint32_t fa = lds0 + (uint32_t)((lane >> 5) * 1024 + (wr * 64 + 2 * (lane & 31)) * 8);
  const uint32_t fb = lds0 + (uint32_t)(16384 + (lane >> 5) * 1024 + (wc * 64 + 2 * (lane & 31)) * 8);
  int st = 0;
  for (int i = 0; i < nk; ++i) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of stage i have landed ...
    __builtin_amdgcn_s_barrier();                          // ... everybody's have, and the other buffer is free
    if (i + 1 < nk) TCMI_DMA_ISSUE(i + 1, st ^ 1)
    const uint32_t sa_ = fa + (uint32_t)(st * TCMI_T128_STAGE_BYTES), sb_ = fb + (uint32_t)(st * TCMI_T128_STAGE_BYTES);
    // fragments of two k pairs at a time, double-buffered: the reads of quarter t + 1 are issued before the 24 MFMAs of
    // quarter t, so only the first quarter's LDS latency of a K step is exposed
    v4f_ xa[2][2], xb[2][2];
#define TCMI_QREAD(BUF, QT)                                                                                          \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                                    \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[BUF][q]) : "v"(sa_), "n"(((QT) * 2 + q) * 2048));         \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[BUF][q]) : "v"(sb_), "n"(((QT) * 2 + q) * 2048));         \
  }
#define TCMI_QWAIT(BUF) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[BUF][0]), "+v"(xa[BUF][1]), "+v"(xb[BUF][0]), "+v"(xb[BUF][1]));
    TCMI_QREAD(0, 0)
    TCMI_QWAIT(0)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      if (qt < 3) {
        if (qt & 1) { TCMI_QREAD(0, qt + 1) } else { TCMI_QREAD(1, qt + 1) }
      }
      __builtin_amdgcn_sched_barrier(0);     // the MFMAs below stay between the reads above and the wait below
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const v4f_ va = xa[qt & 1][q], vb = xb[qt & 1][q];
        // va = (re, im) of rows 2j and 2j + 1; vb likewise for two columns
        const float ar[2] = {va.x, va.z}, ai[2] = {va.y, va.w};
        const float br[2] = {vb.x, vb.z}, bi[2] = {vb.y, vb.w};
        const float as[2] = {ar[0] + ai[0], ar[1] + ai[1]}, bs[2] = {br[0] + bi[0], br[1] + bi[1]};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            acc[u][v][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[u], br[v], acc[u][v][0], 0, 0, 0);
            acc[u][v][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[u], bi[v], acc[u][v][1], 0, 0, 0);
            acc[u][v][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[u], bs[v], acc[u][v][2], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (qt < 3) {
        if (qt & 1) { TCMI_QWAIT(0) } else { TCMI_QWAIT(1) }
      }
    }
#undef TCMI_QREAD
#undef TCMI_QWAIT
    st ^= 1;
  }
#undef TCMI_DMA_ISSUE
  // MFMA result element (row i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), column j = lane & 31) of the (u, v) product
  // is C[wr 64 + 2 i + u][wc 64 + 2 j + v]: the two v of a lane are neighbours -> one 16-byte store per (u, reg)
  const long long colb = n0 + wc * 64 + 2 * (lane & 31);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const long long row = m0 + wr * 64 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) + u;
      v4f_ o;
      o.x = acc[u][0][0][reg] - acc[u][0][1][reg];
      o.y = acc[u][0][2][reg] - acc[u][0][0][reg] - acc[u][0][1][reg];
      o.z = acc[u][1][0][reg] - acc[u][1][1][reg];
      o.w = acc[u][1][2][reg] - acc[u][1][0][reg] - acc[u][1][1][reg];
      *reinterpret_cast<v4f_*>(C + row * N + colb) = o;
    }
}

// ---- tensordot of two [2]^rank tensors straight from their stored layouts (complex64) --------------------------------
// C[m][n] = sum_k A[rowA(m) | kA(k)] * B[kB(k) | colB(n)]: every index bit of M, N and K sits at its own address bit of
// the operand (row bit j of A at pa.free[j], k bit j at pa.k[j] in A and pb.k[j] in B, column bit j of B at
// pb.free[j]); C is written row-major [M x N], i.e. with the axes (free axes of A in stored order, free axes of B in
// stored order) -- exactly what tensordot returns, whatever axes are contracted.  The permute + GEMM route moves a
// big operand three times (read, write permuted, read again); this one reads it once.  Same 64 x 64 tile, planar
// k-major LDS tiles and 3-product MFMA core as cgemm_mfma_kernel; the loaders gather 8-byte elements.
//  * which index runs across the lanes of a loader (KFA / KFB: k-fast, else row / column fast) is chosen by the host
//    from the operand's lowest address bits, so that a wave touches whole 128-byte lines: the 4 lowest address bits
//    are always among the 6 lowest row bits and the 4 lowest k bits, which one tile iteration covers completely.
//  * offsets: the tile part of a row / column (6 low bits) and the 4 low k bits are deposited once per thread; the
//    tile origin and the running k block are wave-uniform and deposited with scalar loops.
struct BitPos {
  int free_[32];  // address bit of row (A) / column (B) bit j
  int k[32];      // address bit of k bit j
};

struct OutPos {
  int pos[32];  // address bit (in C) of bit j of the natural output index (row bits above column bits); identity: j
};

__device__ __forceinline__ uint32_t deposit_bits(uint32_t v, const int* pos, int nb, int from) {
  uint32_t off = 0;
  for (int j = from; j < nb; ++j) off |= ((v >> j) & 1u) << pos[j];
  return off;
}

// Loader of one operand tile (64 rows / columns x 16 k) into the planar k-major LDS tile.  MODE 0: row-fast pairs
// (address bit 0 = row bit 0: 16-byte loads of two rows), MODE 1: k-fast pairs (address bit 0 = k bit 0), MODE 2:
// single 8-byte elements, row-fast (address bit 0 is some other k bit).
template <int MODE>
struct BitsLoader {
  uint32_t off[MODE == 2 ? 4 : 2];
  int r, k;
  bool pair;
  __device__ __forceinline__ void init(int tid, uint32_t base, const int* pfree, const int* pk, int lr, int lk) {
    r = MODE == 1 ? (tid >> 3) : (MODE == 0 ? 2 * (tid & 31) : (tid & 63));
    k = MODE == 1 ? 2 * (tid & 7) : (MODE == 0 ? (tid >> 5) : (tid >> 6));
    pair = MODE == 1 ? lk > 0 : lr > 0;
#pragma unroll
    for (int h = 0; h < (MODE == 2 ? 4 : 2); ++h)
      off[h] = base | deposit_bits((uint32_t)row(h), pfree, lr < 6 ? lr : 6, 0) |
               deposit_bits((uint32_t)kk(h), pk, lk < 4 ? lk : 4, 0);
  }
  __device__ __forceinline__ int row(int h) const { return MODE == 1 ? r + 32 * h : r; }
  __device__ __forceinline__ int kk(int h) const { return MODE == 1 ? k : (MODE == 0 ? k + 8 * h : k + 4 * h); }
  __device__ __forceinline__ void load(const float2* __restrict__ P, uint32_t kofs, uint32_t r0, uint32_t R,
                                       uint32_t k0, uint32_t kend, float (*re)[TCMI_LDP], float (*im)[TCMI_LDP]) const {
#pragma unroll
    for (int h = 0; h < (MODE == 2 ? 4 : 2); ++h) {
      const int rr = row(h), kq = kk(h);
      float2 v0 = {0.f, 0.f}, v1 = {0.f, 0.f};
      if (r0 + rr < R && k0 + kq < kend) {
        const float2* src = P + (off[h] | kofs);
        if (MODE != 2 && pair) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          v0.x = t.x; v0.y = t.y; v1.x = t.z; v1.y = t.w;
        } else {
          v0 = *src;
        }
      }
      if constexpr (MODE == 1) {
        re[kq][rr] = v0.x; im[kq][rr] = v0.y; re[kq + 1][rr] = v1.x; im[kq + 1][rr] = v1.y;
      } else if constexpr (MODE == 0) {
        *reinterpret_cast<float2*>(&re[kq][rr]) = make_float2(v0.x, v1.x);
        *reinterpret_cast<float2*>(&im[kq][rr]) = make_float2(v0.y, v1.y);
      } else {
        re[kq][rr] = v0.x; im[kq][rr] = v0.y;
      }
    }
  }
};

template <int MA, int MB>
__global__ __launch_bounds__(256, 5) void cgemm_bits_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                            float2* __restrict__ C, int lm, int ln, int lk, BitPos pa,
                                                            BitPos pb, int ksplit, long long kchunk, OutPos po,
                                                            int permuted) {
  __shared__ __attribute__((aligned(16))) float As_re[TCMI_CBK][TCMI_LDP], As_im[TCMI_CBK][TCMI_LDP];
  __shared__ __attribute__((aligned(16))) float Bs_re[TCMI_CBK][TCMI_LDP], Bs_im[TCMI_CBK][TCMI_LDP];
  const uint32_t M = 1u << lm, N = 1u << ln, K = 1u << lk;   // rank <= 31: every extent and element index fits 32 bits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // blockIdx.x: column tile, blockIdx.y + 65535 * (blockIdx.z / ksplit): row tile, blockIdx.z % ksplit: k chunk
  const int zz = (int)blockIdx.z / ksplit, ks = (int)blockIdx.z - zz * ksplit;
  const unsigned long long m0l = ((unsigned long long)zz * 65535 + blockIdx.y) * TCMI_BM;
  if (m0l >= M) return;
  const uint32_t m0 = (uint32_t)m0l, n0 = blockIdx.x * TCMI_BN;
  const uint32_t kbeg = (uint32_t)ks * (uint32_t)kchunk;
  const uint32_t kend = ((unsigned long long)kbeg + (unsigned long long)kchunk < K) ? kbeg + (uint32_t)kchunk : K;
  f32x16 p1 = {0}, p2 = {0}, p3 = {0};

  BitsLoader<MA> la;
  BitsLoader<MB> lb;
  la.init(tid, deposit_bits(m0, pa.free_, lm, 6), pa.free_, pa.k, lm, lk);
  lb.init(tid, deposit_bits(n0, pb.free_, ln, 6), pb.free_, pb.k, ln, lk);

  // offsets of the running k block (wave-uniform): deposited once, then only the bits that toggle when the block
  // counter advances are flipped (two on average)
  uint32_t ka = deposit_bits(kbeg, pa.k, lk, 4);
  uint32_t kb = deposit_bits(kbeg, pb.k, lk, 4);
  for (uint32_t k0 = kbeg; k0 < kend; k0 += TCMI_CBK) {
    la.load(A, ka, m0, M, k0, kend, As_re, As_im);
    lb.load(B, kb, n0, N, k0, kend, Bs_re, Bs_im);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TCMI_CBK; kk += 2) {
      const int kr = kk + (lane >> 5);
      const float are = As_re[kr][wr * 32 + (lane & 31)], aim = As_im[kr][wr * 32 + (lane & 31)];
      const float bre = Bs_re[kr][wc * 32 + (lane & 31)], bim = Bs_im[kr][wc * 32 + (lane & 31)];
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(are, bre, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, bim, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(are + aim, bre + bim, p3, 0, 0, 0);
    }
    __syncthreads();
    const uint32_t blk = k0 >> 4;
    uint32_t tog = blk ^ (blk + 1);
    while (tog) {
      const int j = __builtin_ctz(tog) + 4;
      tog &= tog - 1;
      if (j < lk) {
        ka ^= 1u << pa.k[j];
        kb ^= 1u << pb.k[j];
      }
    }
  }
  const uint32_t col = n0 + wc * 32 + (lane & 31);
  const uint32_t rbase = m0 + wr * 32 + 4 * (lane >> 5);
  float2* Cb = C + ((unsigned long long)rbase << ln) + col;
  // result stored with permuted axes (tcmi_tensordot_bits_ex): bit j of the natural index (row << ln | col) goes to
  // address bit po.pos[j].  The 16 rows of a lane differ in row bits 0, 1, 3, 4 only (rbase has them clear), so the
  // address is the deposit of (rbase, col) OR'ed with the deposits of those four bits.
  uint32_t d0 = 1u << ln, d1 = 2u << ln, d3 = 8u << ln, d4 = 16u << ln;
  if (permuted) {
    Cb = C + (deposit_bits(col, po.pos, ln, 0) | deposit_bits(rbase, po.pos + ln, lm, 0));
    d0 = lm > 0 ? 1u << po.pos[ln] : 0u;
    d1 = lm > 1 ? 1u << po.pos[ln + 1] : 0u;
    d3 = lm > 3 ? 1u << po.pos[ln + 3] : 0u;
    d4 = lm > 4 ? 1u << po.pos[ln + 4] : 0u;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const uint32_t dr = (reg & 3) + 8 * (reg >> 2);
    if (rbase + dr < M && col < N) {
      float2 o;
      o.x = p1[reg] - p2[reg];
      o.y = p3[reg] - p1[reg] - p2[reg];
      float2* dst = Cb + (((reg & 1) ? d0 : 0u) | ((reg & 2) ? d1 : 0u) | ((reg & 4) ? d3 : 0u) | ((reg & 8) ? d4 : 0u));
      if (ksplit > 1) {
        atomicAdd(&dst->x, o.x);
        atomicAdd(&dst->y, o.y);
      } else {
        *dst = o;
      }
    }
  }
}

// tensordot from stored layouts for SMALL tensors (operands and result of at most 4096 elements, at most 8 contracted
// axes): the gate-absorbs-gate steps at the bottom of a circuit network's tree, hundreds per contraction.  One thread per
// output element, the k offsets of both operands tabulated in LDS once per workgroup; no tiles, no MFMA -- the 64 x 64
// tile kernel spends more time depositing its tile origin than these steps have arithmetic.
// flags: 1 = conjugate A's elements, 2 = conjugate B's; po: the result is stored with its axes permuted (the two VJPs
// of a tensordot are tensordot(g, conj b) and tensordot(conj a, g) followed by a transposition into the operand's own
// axis order: fused here they are one launch instead of three, and the reverse sweep of a circuit network is
// thousands of such steps of a few microseconds each).
__global__ __launch_bounds__(256) void tensordot_bits_small_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                                   float2* __restrict__ C, int lm, int ln, int lk, BitPos pa,
                                                                   BitPos pb, OutPos po, int flags) {
  __shared__ uint32_t kA[256], kB[256];
  const int tid = threadIdx.x;
  const int K = 1 << lk;
  if (tid < K) {
    kA[tid] = deposit_bits((uint32_t)tid, pa.k, lk, 0);
    kB[tid] = deposit_bits((uint32_t)tid, pb.k, lk, 0);
  }
  __syncthreads();
  const uint32_t o = blockIdx.x * 256u + (uint32_t)tid;
  if (o >= (1u << (lm + ln))) return;
  const uint32_t ra = deposit_bits(o >> ln, pa.free_, lm, 0), cb = deposit_bits(o & ((1u << ln) - 1u), pb.free_, ln, 0);
  const float sa = (flags & 1) ? -1.f : 1.f, sb = (flags & 2) ? -1.f : 1.f;
  float re = 0.f, im = 0.f;
  for (int k = 0; k < K; ++k) {
    float2 a = A[ra | kA[k]], b = B[cb | kB[k]];
    a.y *= sa;
    b.y *= sb;
    re = __builtin_fmaf(a.x, b.x, __builtin_fmaf(-a.y, b.y, re));
    im = __builtin_fmaf(a.x, b.y, __builtin_fmaf(a.y, b.x, im));
  }
  C[deposit_bits(o, po.pos, lm + ln, 0)] = make_float2(re, im);
}

// MANY gate-sized tensordots in one launch: blockIdx.y = job, each job a TCMI_SMALL_DESC_WORDS-word descriptor in device
// memory {a, b, c (64-bit addresses, lo / hi), lm, ln, lk, flags, row bits of a [12], k bits of a [8], column bits of b
// [12], k bits of b [8], output bits [12]}.  The steps of one LEVEL of a contraction tree are independent of each other;
// a 30-qubit depth-8 ladder has ~950 slice-invariant steps in ~40 levels, and its reverse sweep twice as many: one
// launch per level instead of one per step (tn.SmallBatch).
__global__ __launch_bounds__(256) void tensordot_small_batch_kernel(const int* __restrict__ desc_all) {
  __shared__ uint32_t kA[256], kB[256];
  __shared__ int d[TCMI_SMALL_DESC_WORDS];
  const int tid = threadIdx.x;
  if (tid < TCMI_SMALL_DESC_WORDS) d[tid] = desc_all[(long long)blockIdx.y * TCMI_SMALL_DESC_WORDS + tid];
  __syncthreads();
  const int lm = d[6], ln = d[7], lk = d[8], flags = d[9];
  if (blockIdx.x * 256u >= (1u << (lm + ln))) return;   // workgroup-uniform
  const float2* A = reinterpret_cast<const float2*>(((unsigned long long)(unsigned)d[1] << 32) | (unsigned)d[0]);
  const float2* B = reinterpret_cast<const float2*>(((unsigned long long)(unsigned)d[3] << 32) | (unsigned)d[2]);
  float2* C = reinterpret_cast<float2*>(((unsigned long long)(unsigned)d[5] << 32) | (unsigned)d[4]);
  const int* pfa = d + 10;
  const int* pka = d + 22;
  const int* pfb = d + 30;
  const int* pkb = d + 42;
  const int* pc = d + 50;
  const int K = 1 << lk;
  if (tid < K) {
    kA[tid] = deposit_bits((uint32_t)tid, pka, lk, 0);
    kB[tid] = deposit_bits((uint32_t)tid, pkb, lk, 0);
  }
  __syncthreads();
  const uint32_t o = blockIdx.x * 256u + (uint32_t)tid;
  if (o >= (1u << (lm + ln))) return;
  const uint32_t ra = deposit_bits(o >> ln, pfa, lm, 0), cb = deposit_bits(o & ((1u << ln) - 1u), pfb, ln, 0);
  const float sa = (flags & 1) ? -1.f : 1.f, sb = (flags & 2) ? -1.f : 1.f;
  float re = 0.f, im = 0.f;
  for (int k = 0; k < K; ++k) {
    float2 a = A[ra | kA[k]], b = B[cb | kB[k]];
    a.y *= sa;
    b.y *= sb;
    re = __builtin_fmaf(a.x, b.x, __builtin_fmaf(-a.y, b.y, re));
    im = __builtin_fmaf(a.x, b.y, __builtin_fmaf(a.y, b.x, im));
  }
  C[deposit_bits(o, pc, lm + ln, 0)] = make_float2(re, im);
}

// tensordot from stored layouts whose result has at most 8 x 8 elements and a long contraction (two big tensors closing
// to a few numbers: 2^25 x 2^25 -> 8 x 8 over K = 2^22 in a reconfigured RQC tree): one thread per k, MT x NT accumulators
// in registers, k offsets = the thread's low 8 k bits (deposited once) | the block counter's bits (scalar code), DPP wave
// sums -> LDS -> one f32 atomic per workgroup and output component into the zeroed C.  The MFMA tile kernel with split-K
// ran this shape at 0.4 TB/s (an 8 x 8 corner of a 64 x 64 tile, 2 M atomics).
template <int MT, int NT>
__global__ __launch_bounds__(256) void tensordot_bits_tiny_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                                  float2* __restrict__ C, int lm, int ln, int lk, BitPos pa,
                                                                  BitPos pb, OutPos po) {
  const int tid = threadIdx.x;
  const uint32_t K = 1u << lk;
  uint32_t ra[MT], cb[NT];
#pragma unroll
  for (int r = 0; r < MT; ++r) ra[r] = deposit_bits((uint32_t)r, pa.free_, lm, 0);      // wave-uniform
#pragma unroll
  for (int c = 0; c < NT; ++c) cb[c] = deposit_bits((uint32_t)c, pb.free_, ln, 0);
  const uint32_t ka_t = deposit_bits((uint32_t)tid, pa.k, lk < 8 ? lk : 8, 0);
  const uint32_t kb_t = deposit_bits((uint32_t)tid, pb.k, lk < 8 ? lk : 8, 0);
  float re[MT][NT], im[MT][NT];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = 0; c < NT; ++c) { re[r][c] = 0.f; im[r][c] = 0.f; }
  const uint32_t nblk = (K + 255u) >> 8;
  for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    if ((blk << 8) + (uint32_t)tid >= K) break;
    uint32_t ka = ka_t, kb = kb_t;
    for (int j = 8; j < lk; ++j) {           // the block counter's bits: scalar
      const uint32_t bit = (blk >> (j - 8)) & 1u;
      ka |= bit << pa.k[j];
      kb |= bit << pb.k[j];
    }
    float2 a[MT], b[NT];
#pragma unroll
    for (int r = 0; r < MT; ++r) a[r] = A[ra[r] | ka];
#pragma unroll
    for (int c = 0; c < NT; ++c) b[c] = B[cb[c] | kb];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        re[r][c] = __builtin_fmaf(a[r].x, b[c].x, __builtin_fmaf(-a[r].y, b[c].y, re[r][c]));
        im[r][c] = __builtin_fmaf(a[r].x, b[c].y, __builtin_fmaf(a[r].y, b[c].x, im[r][c]));
      }
  }
  __shared__ float red[4][2 * MT * NT];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const float sr = wave_sum_uniform(re[r][c]), si = wave_sum_uniform(im[r][c]);
      if ((tid & 63) == 0) {
        red[tid >> 6][2 * (r * NT + c)] = sr;
        red[tid >> 6][2 * (r * NT + c) + 1] = si;
      }
    }
  __syncthreads();
  if (tid < 2 * MT * NT)   // (po: identity unless the result is stored with permuted axes)
    atomicAdd(reinterpret_cast<float*>(C) + 2 * deposit_bits((uint32_t)tid >> 1, po.pos, lm + ln, 0) + (tid & 1),
              red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

// complex128 GEMM on the f64 matrix pipe (v_mfma_f64_16x16x4_f64): 64x64 tile per workgroup, 4 waves x
// (32x32 = 2x2 MFMA tiles), K step 8, the same 3-product (Gauss) form and planar k-major LDS tiles as
// the complex64 kernel.  Row pitch 66 doubles: 16-byte aligned rows for the vector loaders and
// 2 * 66 = 4 (mod 16) double-banks for the transposing loader.  C/D layout of the f64 MFMA:
// col = lane & 15, row = (lane >> 4) + 4 * reg.
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define TCMI_ZLDP 66

template <bool TRANS_A>
__global__ __launch_bounds__(256, 4) void zgemm_mfma_kernel(const double2* __restrict__ A,
                                                          const double2* __restrict__ B,
                                                          double2* __restrict__ C, int M, int N, int K,
                                                          long long sA, long long sB, long long sC) {
  __shared__ __attribute__((aligned(16))) double As_re[TCMI_BK][TCMI_ZLDP], As_im[TCMI_BK][TCMI_ZLDP];
  __shared__ __attribute__((aligned(16))) double Bs_re[TCMI_BK][TCMI_ZLDP], Bs_im[TCMI_BK][TCMI_ZLDP];
  A += (long long)blockIdx.z * sA;
  B += (long long)blockIdx.z * sB;
  C += (long long)blockIdx.z * sC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const long long m0 = (long long)blockIdx.y * TCMI_BM, n0 = (long long)blockIdx.x * TCMI_BN;
  f64x4 p1[2][2], p2[2][2], p3[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      p1[i][j] = f64x4{0, 0, 0, 0};
      p2[i][j] = f64x4{0, 0, 0, 0};
      p3[i][j] = f64x4{0, 0, 0, 0};
    }
  const int ai = TRANS_A ? (tid & 31) * 2 : tid >> 2, ak = TRANS_A ? tid >> 5 : (tid & 3) * 2;
  const int bk = tid >> 5, bj = (tid & 31) * 2;
  for (int k0 = 0; k0 < K; k0 += TCMI_BK) {
    {
      double2 v0 = {0., 0.}, v1 = {0., 0.};
      if constexpr (TRANS_A) {
        const long long r = m0 + ai;
        const int kk = k0 + ak;
        if (kk < K) {
          if (r < M) v0 = A[(long long)kk * M + r];
          if (r + 1 < M) v1 = A[(long long)kk * M + r + 1];
        }
        *reinterpret_cast<double2*>(&As_re[ak][ai]) = make_double2(v0.x, v1.x);
        *reinterpret_cast<double2*>(&As_im[ak][ai]) = make_double2(v0.y, v1.y);
      } else {
        const long long r = m0 + ai;
        const int kk = k0 + ak;
        if (r < M) {
          if (kk < K) v0 = A[r * K + kk];
          if (kk + 1 < K) v1 = A[r * K + kk + 1];
        }
        As_re[ak][ai] = v0.x;
        As_im[ak][ai] = v0.y;
        As_re[ak + 1][ai] = v1.x;
        As_im[ak + 1][ai] = v1.y;
      }
      double2 w0 = {0., 0.}, w1 = {0., 0.};
      const long long c = n0 + bj;
      const int kb = k0 + bk;
      if (kb < K) {
        if (c < N) w0 = B[(long long)kb * N + c];
        if (c + 1 < N) w1 = B[(long long)kb * N + c + 1];
      }
      *reinterpret_cast<double2*>(&Bs_re[bk][bj]) = make_double2(w0.x, w1.x);
      *reinterpret_cast<double2*>(&Bs_im[bk][bj]) = make_double2(w0.y, w1.y);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TCMI_BK; kk += 4) {
      const int kr = kk + (lane >> 4);
      double are[2], aim[2], asum[2], bre[2], bim[2], bsum[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        are[i] = As_re[kr][wr * 32 + i * 16 + (lane & 15)];
        aim[i] = As_im[kr][wr * 32 + i * 16 + (lane & 15)];
        asum[i] = are[i] + aim[i];
        bre[i] = Bs_re[kr][wc * 32 + i * 16 + (lane & 15)];
        bim[i] = Bs_im[kr][wc * 32 + i * 16 + (lane & 15)];
        bsum[i] = bre[i] + bim[i];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          p1[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[i], bre[j], p1[i][j], 0, 0, 0);
          p2[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[i], bim[j], p2[i][j], 0, 0, 0);
          p3[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(asum[i], bsum[j], p3[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long long col = n0 + wc * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long long row = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * reg;
        if (row < M && col < N) {
          double2 o;
          o.x = p1[i][j][reg] - p2[i][j][reg];
          o.y = p3[i][j][reg] - p1[i][j][reg] - p2[i][j][reg];
          C[row * N + col] = o;
        }
      }
    }
}

// complex128 (and tiny shapes): one output element per thread, fp64 FMA chain
template <typename F>
__global__ void cgemm_simple_kernel(const typename Cx<F>::type* __restrict__ A,
                                    const typename Cx<F>::type* __restrict__ B,
                                    typename Cx<F>::type* __restrict__ C, long long M, long long N, int K,
                                    long long sA, long long sB, long long sC, int trans_a) {
  using Ct = typename Cx<F>::type;
  A += (long long)blockIdx.z * sA;
  B += (long long)blockIdx.z * sB;
  C += (long long)blockIdx.z * sC;
  const long long total = M * N;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long o = i0; o < total; o += step) {
    const long long r = o / N, c = o - r * N;
    F re = 0, im = 0;
    for (int k = 0; k < K; ++k) {
      const Ct a = trans_a ? A[(long long)k * M + r] : A[r * K + k];
      const Ct b = B[(long long)k * N + c];
      re = fma_<F>(a.x, b.x, re);
      re = fma_<F>(-a.y, b.y, re);
      im = fma_<F>(a.x, b.y, im);
      im = fma_<F>(a.y, b.x, im);
    }
    Ct v;
    v.x = re;
    v.y = im;
    C[o] = v;
  }
}

// Small output, long contraction (M N < 1024, K large: the closing steps of a sliced network reduce two
// big tensors to a few numbers).  Split-K: a workgroup owns a K chunk, its 256 threads are
// (k-lane, output slot) pairs, partial sums are reduced through LDS and added to the zero-initialised
// C with float atomics.
template <typename F>
__global__ __launch_bounds__(256) void cgemm_splitk_kernel(const typename Cx<F>::type* __restrict__ A,
                                                           const typename Cx<F>::type* __restrict__ B,
                                                           typename Cx<F>::type* __restrict__ C, long long M,
                                                           long long N, long long K, long long sA, long long sB,
                                                           long long sC, int trans_a, long long kchunk, int to_log2) {
  using Ct = typename Cx<F>::type;
  __shared__ F red_re[256], red_im[256];
  A += (long long)blockIdx.z * sA;
  B += (long long)blockIdx.z * sB;
  C += (long long)blockIdx.z * sC;
  const int TO = 1 << to_log2, KL = 256 >> to_log2;
  const int o = threadIdx.x & (TO - 1), kl = threadIdx.x >> to_log2;
  const long long k0 = (long long)blockIdx.x * kchunk;
  const long long k1 = k0 + kchunk < K ? k0 + kchunk : K;
  const long long total = M * N;
  for (long long base = 0; base < total; base += TO) {
    const long long oo = base + o;
    F re = 0, im = 0;
    if (oo < total) {
      const long long r = oo / N, c = oo - r * N;
      for (long long k = k0 + kl; k < k1; k += KL) {
        const Ct a = trans_a ? A[k * M + r] : A[r * K + k];
        const Ct b = B[k * N + c];
        re = fma_<F>(a.x, b.x, re);
        re = fma_<F>(-a.y, b.y, re);
        im = fma_<F>(a.x, b.y, im);
        im = fma_<F>(a.y, b.x, im);
      }
    }
    red_re[threadIdx.x] = re;
    red_im[threadIdx.x] = im;
    __syncthreads();
    if (kl == 0 && oo < total) {
      F sr = 0, si = 0;
      for (int j = 0; j < KL; ++j) {
        sr += red_re[j * TO + o];
        si += red_im[j * TO + o];
      }
      atomicAdd(&C[oo].x, sr);
      atomicAdd(&C[oo].y, si);
    }
    __syncthreads();
  }
}


// A few outputs (M, N <= 8) and a long contraction: one thread per k (grid-stride), the M x N accumulators in
// registers, coalesced loads (A: one element per row and thread, B: the N contiguous elements of row k), DPP wave
// reduction and one atomic per wave and output into the zero-initialised C.  Reads both operands once at HBM speed
// (the (k-lane, output) pairing of cgemm_splitk_kernel fetches 8-byte fragments: 0.5 TB/s on the closing step of
// config 4).
template <typename F, int MT, int NT>
__global__ __launch_bounds__(256) void cgemm_tinyout_kernel(const typename Cx<F>::type* __restrict__ A,
                                                            const typename Cx<F>::type* __restrict__ B,
                                                            typename Cx<F>::type* __restrict__ C, long long K,
                                                            long long sA, long long sB, long long sC, int trans_a) {
  using Ct = typename Cx<F>::type;
  A += (long long)blockIdx.z * sA;
  B += (long long)blockIdx.z * sB;
  C += (long long)blockIdx.z * sC;
  F re[MT][NT], im[MT][NT];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = 0; c < NT; ++c) { re[r][c] = 0; im[r][c] = 0; }
  const long long step = (long long)gridDim.x * 256;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < K; k += step) {
    Ct a[MT], b[NT];
#pragma unroll
    for (int r = 0; r < MT; ++r) a[r] = trans_a ? A[k * MT + r] : A[(long long)r * K + k];
#pragma unroll
    for (int c = 0; c < NT; ++c) b[c] = B[k * NT + c];
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        re[r][c] = fma_<F>(a[r].x, b[c].x, re[r][c]);
        re[r][c] = fma_<F>(-a[r].y, b[c].y, re[r][c]);
        im[r][c] = fma_<F>(a[r].x, b[c].y, im[r][c]);
        im[r][c] = fma_<F>(a[r].y, b[c].x, im[r][c]);
      }
  }
  // wave sums -> LDS -> one atomic per workgroup and output component (same-address atomics serialise in L2)
  __shared__ F red[4][2 * MT * NT];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const F sr = wave_sum_uniform(re[r][c]), si = wave_sum_uniform(im[r][c]);
      if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][2 * (r * NT + c)] = sr;
        red[threadIdx.x >> 6][2 * (r * NT + c) + 1] = si;
      }
    }
  __syncthreads();
  if (threadIdx.x < 2 * MT * NT) {
    const F v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(reinterpret_cast<F*>(C) + threadIdx.x, v);
  }
}

template <typename F, int MT>
static void launch_tinyout_n(const void* A, const void* B, void* C, long long N, long long K, int batch, long long sA,
                             long long sB, long long sC, int trans_a, hipStream_t st) {
  using Ct = typename Cx<F>::type;
  long long nb = (K + 255) / 256;
  if (nb > 1024) nb = 1024;
  dim3 grid((unsigned)nb, 1, (unsigned)batch), block(256, 1, 1);
#define TCMI_TO(NTV)                                                                                                  \
  hipLaunchKernelGGL((cgemm_tinyout_kernel<F, MT, NTV>), grid, block, 0, st, reinterpret_cast<const Ct*>(A),            \
                     reinterpret_cast<const Ct*>(B), reinterpret_cast<Ct*>(C), K, sA, sB, sC, trans_a);
  if (N == 1) { TCMI_TO(1) } else if (N == 2) { TCMI_TO(2) } else if (N == 4) { TCMI_TO(4) } else { TCMI_TO(8) }
#undef TCMI_TO
}

template <typename F>
static void launch_tinyout(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                           long long sA, long long sB, long long sC, int trans_a, hipStream_t st) {
  if (M == 1) launch_tinyout_n<F, 1>(A, B, C, N, K, batch, sA, sB, sC, trans_a, st);
  else if (M == 2) launch_tinyout_n<F, 2>(A, B, C, N, K, batch, sA, sB, sC, trans_a, st);
  else if (M == 4) launch_tinyout_n<F, 4>(A, B, C, N, K, batch, sA, sB, sC, trans_a, st);
  else launch_tinyout_n<F, 8>(A, B, C, N, K, batch, sA, sB, sC, trans_a, st);
}

// ---- big tensor x small tensor over scattered bit positions -------------------------------------------------
// out[f][n] (big_first) or out[n][f] = sum_k big[deposit(f) | koff(k)] * small[k][n]: the contracted axes of the big
// [2]^rank tensor sit at arbitrary bit positions pos[0] < pos[1] < ... (bit j of k <-> pos[j]); f runs over the free
// bits in their stored order.  One thread per f: K gathered loads (coalesced across threads when the low bits are
// free), the small operand broadcast from LDS, NT accumulators at a time, contiguous stores.  Reads the big
// tensor once and writes the result once — the permute + skinny-GEMM route moves the big tensor three times.
struct ScatPos { int p[8]; };

template <typename F, int LK, int NT>
__global__ __launch_bounds__(256) void contract_scattered_kernel(const typename Cx<F>::type* __restrict__ big,
                                                                 const typename Cx<F>::type* __restrict__ small_,
                                                                 typename Cx<F>::type* __restrict__ out, int rank,
                                                                 ScatPos pos, int N, int big_first) {
  using C = typename Cx<F>::type;
  constexpr int K = 1 << LK;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  C* sB = reinterpret_cast<C*>(smem_);
  __shared__ unsigned long long koff[K];
  const int tid = threadIdx.x;
  for (int i = tid; i < K * N; i += 256) sB[i] = small_[i];
  if (tid < K) {
    unsigned long long off = 0;
#pragma unroll
    for (int j = 0; j < LK; ++j)
      if ((tid >> j) & 1) off |= 1ull << pos.p[j];
    koff[tid] = off;
  }
  __syncthreads();
  const unsigned long long nf = 1ull << (rank - LK);
  const unsigned long long step = (unsigned long long)gridDim.x * 256;
  for (unsigned long long f = (unsigned long long)blockIdx.x * 256 + tid; f < nf; f += step) {
    unsigned long long x = f;
#pragma unroll
    for (int j = 0; j < LK; ++j) {
      const unsigned long long low = (1ull << pos.p[j]) - 1ull;
      x = ((x & ~low) << 1) | (x & low);
    }
    C a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = big[x | koff[k]];
    for (int n0 = 0; n0 < N; n0 += NT) {
      C acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) { acc[n].x = 0; acc[n].y = 0; }
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const C b = sB[k * N + n0 + n];
          acc[n].x = fma_<F>(a[k].x, b.x, acc[n].x);
          acc[n].x = fma_<F>(-a[k].y, b.y, acc[n].x);
          acc[n].y = fma_<F>(a[k].x, b.y, acc[n].y);
          acc[n].y = fma_<F>(a[k].y, b.x, acc[n].y);
        }
      }
      if (big_first) {
        C* dst = out + f * (unsigned long long)N + n0;
#pragma unroll
        for (int n = 0; n < NT; ++n) dst[n] = acc[n];
      } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) out[(unsigned long long)(n0 + n) * nf + f] = acc[n];
      }
    }
  }
}

template <typename F, int LK>
static int launch_scattered(const void* big, const void* small_, void* out, int rank, const int* pos_host, long long N,
                            int big_first, hipStream_t st) {
  using C = typename Cx<F>::type;
  ScatPos pos;
  for (int j = 0; j < 8; ++j) pos.p[j] = j < LK ? pos_host[j] : 0;
  const unsigned long long nf = 1ull << (rank - LK);
  unsigned gx = (unsigned)((nf + 255) / 256 > 65536 ? 65536 : (nf + 255) / 256);
  const size_t lds = sizeof(C) * (size_t)(1 << LK) * (size_t)N;
  dim3 grid(gx, 1, 1), block(256, 1, 1);
#define TCMI_SCAT(NTV)                                                                                              \
  {                                                                                                                  \
    auto kern = contract_scattered_kernel<F, LK, NTV>;                                                               \
    if (lds > 48 * 1024) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const C*>(big), reinterpret_cast<const C*>(small_), \
                       reinterpret_cast<C*>(out), rank, pos, (int)N, big_first);                                    \
  }
  if (N >= 8) TCMI_SCAT(8)
  else if (N == 4) TCMI_SCAT(4)
  else if (N == 2) TCMI_SCAT(2)
  else TCMI_SCAT(1)
#undef TCMI_SCAT
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}


// Same contraction for few outputs per free index (N <= 16) and up to 2^8 contracted configurations: the N
// accumulators stay in registers and the big operand is streamed over k (each element is read once).
template <typename F, int NT>
__global__ __launch_bounds__(256) void contract_scattered_stream_kernel(const typename Cx<F>::type* __restrict__ big,
                                                                        const typename Cx<F>::type* __restrict__ small_,
                                                                        typename Cx<F>::type* __restrict__ out, int rank,
                                                                        ScatPos pos, int lk, int big_first) {
  using C = typename Cx<F>::type;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  C* sB = reinterpret_cast<C*>(smem_);
  __shared__ unsigned long long koff[256];
  const int tid = threadIdx.x;
  const int K = 1 << lk;
  for (int i = tid; i < K * NT; i += 256) sB[i] = small_[i];
  if (tid < K) {
    unsigned long long off = 0;
    for (int j = 0; j < lk; ++j)
      if ((tid >> j) & 1) off |= 1ull << pos.p[j];
    koff[tid] = off;
  }
  __syncthreads();
  const unsigned long long nf = 1ull << (rank - lk);
  const unsigned long long step = (unsigned long long)gridDim.x * 256;
  for (unsigned long long f = (unsigned long long)blockIdx.x * 256 + tid; f < nf; f += step) {
    unsigned long long x = f;
    for (int j = 0; j < lk; ++j) {
      const unsigned long long low = (1ull << pos.p[j]) - 1ull;
      x = ((x & ~low) << 1) | (x & low);
    }
    C acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { acc[n].x = 0; acc[n].y = 0; }
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const C a = big[x | koff[k]];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const C b = sB[k * NT + n];
        acc[n].x = fma_<F>(a.x, b.x, acc[n].x);
        acc[n].x = fma_<F>(-a.y, b.y, acc[n].x);
        acc[n].y = fma_<F>(a.x, b.y, acc[n].y);
        acc[n].y = fma_<F>(a.y, b.x, acc[n].y);
      }
    }
    if (big_first) {
      C* dst = out + f * (unsigned long long)NT;
#pragma unroll
      for (int n = 0; n < NT; ++n) dst[n] = acc[n];
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) out[(unsigned long long)n * nf + f] = acc[n];
    }
  }
}

template <typename F>
static int launch_scattered_stream(const void* big, const void* small_, void* out, int rank, const int* pos_host, int lk,
                                   long long N, int big_first, hipStream_t st) {
  using C = typename Cx<F>::type;
  ScatPos pos;
  for (int j = 0; j < 8; ++j) pos.p[j] = j < lk ? pos_host[j] : 0;
  const unsigned long long nf = 1ull << (rank - lk);
  unsigned gx = (unsigned)((nf + 255) / 256 > 65536 ? 65536 : (nf + 255) / 256);
  const size_t lds = sizeof(C) * ((size_t)N << lk);
  dim3 grid(gx, 1, 1), block(256, 1, 1);
#define TCMI_SCS(NTV)                                                                                                \
  {                                                                                                                  \
    auto kern = contract_scattered_stream_kernel<F, NTV>;                                                            \
    if (lds > 48 * 1024) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, lds, st, reinterpret_cast<const C*>(big), reinterpret_cast<const C*>(small_), \
                       reinterpret_cast<C*>(out), rank, pos, lk, big_first);                                        \
  }
  if (N == 16) TCMI_SCS(16)
  else if (N == 8) TCMI_SCS(8)
  else if (N == 4) TCMI_SCS(4)
  else if (N == 2) TCMI_SCS(2)
  else TCMI_SCS(1)
#undef TCMI_SCS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

// Zero fill as an ordinary kernel launch.  The split-K paths clear their output before accumulating with atomics; as a
// hipMemsetAsync that clear becomes a memset NODE when the call is captured into a HIP graph, and two graphs replayed
// side by side then produced garbage sums (scripts/experiments/README.md); a kernel node is ordered like its
// neighbours.
__global__ void zero_fill_kernel(unsigned long long* __restrict__ p, size_t n8) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += step) p[i] = 0ull;
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if ((bytes & 7) || (reinterpret_cast<size_t>(p) & 7)) return hipMemsetAsync(p, 0, bytes, st);
  const size_t n8 = bytes / 8;
  unsigned gx = (unsigned)((n8 + 255) / 256 > 1024 ? 1024 : (n8 + 255) / 256);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(gx), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(p), n8);
  return hipGetLastError();
}

}  // namespace tcmi


extern "C" {

int tcmi_permute_bits(const void* in, void* out, int rank, const int* srcbit_dev, int batch,
                      long long batch_stride, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!in || !out || !srcbit_dev || rank < 0 || rank > 34 || batch < 1 || in == out)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_permute_bits: bad argument");
  const unsigned long long nelem = 1ull << rank;
  unsigned gx = (unsigned)((nelem + 255) / 256 > 16384 ? 16384 : (nelem + 255) / 256);
  dim3 grid(gx, batch, 1), block(256, 1, 1);
  if (dtype == TCMI_C64)
    hipLaunchKernelGGL(tcmi::permute_bits_kernel<float2>, grid, block, 0, st, reinterpret_cast<const float2*>(in),
                       reinterpret_cast<float2*>(out), rank, srcbit_dev, batch_stride);
  else if (dtype == TCMI_C128)
    hipLaunchKernelGGL(tcmi::permute_bits_kernel<double2>, grid, block, 0, st, reinterpret_cast<const double2*>(in),
                       reinterpret_cast<double2*>(out), rank, srcbit_dev, batch_stride);
  else
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_permute_bits: bad dtype");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_contract_scattered(const void* big, int rank, const int* pos_host, int nk, const void* small_, long long n,
                            void* out, int big_first, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!big || !small_ || !out || !pos_host || nk < 1 || nk > 8 || rank < nk || rank > 34 || n < 1 || (n & (n - 1)) ||
      ((long long)(1 << nk) * n) > 4096 || (nk > 5 && n > 16))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_contract_scattered: bad argument");
  for (int j = 0; j < nk; ++j)
    if (pos_host[j] < 0 || pos_host[j] >= rank || (j > 0 && pos_host[j] <= pos_host[j - 1]))
      return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_contract_scattered: positions must be ascending bit indices");
  if (nk > 5) {   // many contracted configurations, few outputs: stream over k
    if (dtype == TCMI_C64) return tcmi::launch_scattered_stream<float>(big, small_, out, rank, pos_host, nk, n, big_first, st);
    if (dtype == TCMI_C128) return tcmi::launch_scattered_stream<double>(big, small_, out, rank, pos_host, nk, n, big_first, st);
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_contract_scattered: bad dtype");
  }
#define TCMI_SC(FT, LKV) \
  if (nk == LKV) return tcmi::launch_scattered<FT, LKV>(big, small_, out, rank, pos_host, n, big_first, st);
  if (dtype == TCMI_C64) {
    TCMI_SC(float, 1) TCMI_SC(float, 2) TCMI_SC(float, 3) TCMI_SC(float, 4) TCMI_SC(float, 5)
  } else if (dtype == TCMI_C128) {
    TCMI_SC(double, 1) TCMI_SC(double, 2) TCMI_SC(double, 3) TCMI_SC(double, 4) TCMI_SC(double, 5)
  }
#undef TCMI_SC
  return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_contract_scattered: bad dtype");
}

// Bit geometry of tensordot(a, b, axes): address bits of the k / row / column bits of both operands, and of the bits
// of the natural output index when the result is stored permuted (out_axes; NULL: natural).  0 or an error code.
static int bits_geometry(int rank_a, int rank_b, const int* axes_a, const int* axes_b, int nk, const int* out_axes,
                         tcmi::BitPos& pa, tcmi::BitPos& pb, tcmi::OutPos& po) {
  if (rank_a < 0 || rank_b < 0 || rank_a > 31 || rank_b > 31 || nk < 0 || nk > rank_a || nk > rank_b ||
      (nk > 0 && (!axes_a || !axes_b)))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: bad argument");
  unsigned usedA = 0, usedB = 0;
  for (int j = 0; j < nk; ++j) {
    if (axes_a[j] < 0 || axes_a[j] >= rank_a || axes_b[j] < 0 || axes_b[j] >= rank_b)
      return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: axis out of range");
    if ((usedA >> axes_a[j]) & 1u || (usedB >> axes_b[j]) & 1u)
      return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: repeated axis");
    usedA |= 1u << axes_a[j];
    usedB |= 1u << axes_b[j];
  }
  // k bit j <-> j-th contracted pair, pairs ordered by ascending address bit in the LARGER operand (its k offsets
  // then grow monotonically with k: consecutive k blocks walk its lines in order)
  int order[32];
  for (int j = 0; j < nk; ++j) order[j] = j;
  const bool a_big = rank_a >= rank_b;
  for (int i = 1; i < nk; ++i)
    for (int j = i; j > 0; --j) {
      const int pj = a_big ? rank_a - 1 - axes_a[order[j]] : rank_b - 1 - axes_b[order[j]];
      const int pi = a_big ? rank_a - 1 - axes_a[order[j - 1]] : rank_b - 1 - axes_b[order[j - 1]];
      if (pj < pi) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    }
  for (int j = 0; j < 32; ++j) pa.k[j] = pb.k[j] = pa.free_[j] = pb.free_[j] = 0;
  for (int j = 0; j < nk; ++j) {
    pa.k[j] = rank_a - 1 - axes_a[order[j]];
    pb.k[j] = rank_b - 1 - axes_b[order[j]];
  }
  // free axes in stored order: the LAST free axis is row / column bit 0
  for (int i = rank_a - 1, j = 0; i >= 0; --i)
    if (!((usedA >> i) & 1u)) pa.free_[j++] = rank_a - 1 - i;
  for (int i = rank_b - 1, j = 0; i >= 0; --i)
    if (!((usedB >> i) & 1u)) pb.free_[j++] = rank_b - 1 - i;
  // stored axis i of the result = natural axis out_axes[i] (natural order: free(a), free(b)); natural axis j is
  // bit rc - 1 - j of the natural output index
  const int rc = rank_a + rank_b - 2 * nk;
  for (int j = 0; j < 32; ++j) po.pos[j] = j;
  if (out_axes) {
    if (rc > 31) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits_ex: result rank above 31");
    unsigned seen = 0;
    for (int i = 0; i < rc; ++i) {
      if (out_axes[i] < 0 || out_axes[i] >= rc || ((seen >> out_axes[i]) & 1u))
        return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits_ex: out_axes is not a permutation");
      seen |= 1u << out_axes[i];
      po.pos[rc - 1 - out_axes[i]] = rc - 1 - i;
    }
  }
  return TCMI_OK;
}

int tcmi_tensordot_small_desc(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                              int nk, const int* out_axes, int flags, void* c, int* desc) {
  if (!desc || !a || !b || !c) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_small_desc: null argument");
  if (!tcmi_tensordot_bits_small_ok(rank_a, rank_b, nk))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_small_desc: not a small-kernel shape (tcmi_tensordot_bits_small_ok)");
  tcmi::BitPos pa, pb;
  tcmi::OutPos po;
  const int e = bits_geometry(rank_a, rank_b, axes_a, axes_b, nk, out_axes, pa, pb, po);
  if (e != TCMI_OK) return e;
  const unsigned long long pv[3] = {(unsigned long long)a, (unsigned long long)b, (unsigned long long)c};
  for (int i = 0; i < 3; ++i) {
    desc[2 * i] = (int)(unsigned)(pv[i] & 0xffffffffull);
    desc[2 * i + 1] = (int)(unsigned)(pv[i] >> 32);
  }
  desc[6] = rank_a - nk;
  desc[7] = rank_b - nk;
  desc[8] = nk;
  desc[9] = flags;
  for (int j = 0; j < 12; ++j) desc[10 + j] = pa.free_[j];
  for (int j = 0; j < 8; ++j) desc[22 + j] = pa.k[j];
  for (int j = 0; j < 12; ++j) desc[30 + j] = pb.free_[j];
  for (int j = 0; j < 8; ++j) desc[42 + j] = pb.k[j];
  for (int j = 0; j < 12; ++j) desc[50 + j] = po.pos[j];
  desc[62] = desc[63] = 0;
  return TCMI_OK;
}

int tcmi_tensordot_small_batch(const int* desc_dev, int count, int max_log2_out, void* stream) {
  if (!desc_dev || count < 0 || max_log2_out < 0 || max_log2_out > 12)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_small_batch: bad argument");
  if (count == 0) return TCMI_OK;
  if (count > 65535) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_small_batch: more than 65535 jobs");
  const unsigned gx = (unsigned)(((1u << max_log2_out) + 255u) / 256u);
  hipLaunchKernelGGL(tcmi::tensordot_small_batch_kernel, dim3(gx, (unsigned)count, 1), dim3(256, 1, 1), 0,
                     reinterpret_cast<hipStream_t>(stream), desc_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_tensordot_bits_small_ok(int rank_a, int rank_b, int nk) {
  const int lm = rank_a - nk, ln = rank_b - nk;
  return rank_a <= 12 && rank_b <= 12 && lm + ln <= 12 && nk <= 8 && nk >= 0 && lm >= 0 && ln >= 0;
}

int tcmi_tensordot_bits(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                        int nk, void* c, int dtype, void* stream) {
  return tcmi_tensordot_bits_ex(a, rank_a, b, rank_b, axes_a, axes_b, nk, nullptr, 0, c, dtype, stream);
}

int tcmi_tensordot_bits_ex(const void* a, int rank_a, const void* b, int rank_b, const int* axes_a, const int* axes_b,
                           int nk, const int* out_axes, int flags, void* c, int dtype, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (flags && !tcmi_tensordot_bits_small_ok(rank_a, rank_b, nk))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits_ex: conjugation only for the small-tensor kernel (tcmi_tensordot_bits_small_ok)");
  if (!a || !b || !c) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: bad argument");
  if (dtype != TCMI_C64) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: complex64 only");
  // axis i of a [2]^rank tensor is address bit rank - 1 - i
  tcmi::BitPos pa, pb;
  tcmi::OutPos po;
  {
    const int ge = bits_geometry(rank_a, rank_b, axes_a, axes_b, nk, out_axes, pa, pb, po);
    if (ge != TCMI_OK) return ge;
  }
  const int lm = rank_a - nk, ln = rank_b - nk;
  // loader mode of each operand by what its address bit 0 is: row / column bit 0 (0), k bit 0 (1), another k bit (2)
  const int ma = (nk > 0 && pa.k[0] == 0) ? 1 : ((lm > 0 && pa.free_[0] == 0) || rank_a == 0 ? 0 : 2);
  const int mb = (nk > 0 && pb.k[0] == 0) ? 1 : ((ln > 0 && pb.free_[0] == 0) || rank_b == 0 ? 0 : 2);
  const long long M = 1ll << lm, N = 1ll << ln, K = 1ll << nk;
  if (rank_a <= 12 && rank_b <= 12 && lm + ln <= 12 && nk <= 8) {
    // small tensors: one thread per output element
    dim3 grid((unsigned)(((1ll << (lm + ln)) + 255) / 256), 1, 1), block(256, 1, 1);
    hipLaunchKernelGGL(tcmi::tensordot_bits_small_kernel, grid, block, 0, st, reinterpret_cast<const float2*>(a),
                       reinterpret_cast<const float2*>(b), reinterpret_cast<float2*>(c), lm, ln, nk, pa, pb, po, flags);
    hipError_t se = hipGetLastError();
    if (se != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(se));
    return TCMI_OK;
  }
  if (lm <= 3 && ln <= 3 && K >= 2048) {
    // a few numbers out of a long contraction: the register-accumulator kernel
    hipError_t me = tcmi::zero_async(c, (size_t)(M * N) * sizeof(float2), st);
    if (me != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(me));
    long long nb = (K + 255) / 256;
    if (nb > 2048) nb = 2048;
    dim3 grid((unsigned)nb, 1, 1), block(256, 1, 1);
#define TCMI_TT(MV, NV)                                                                                                 \
  if (lm == MV && ln == NV)                                                                                             \
    hipLaunchKernelGGL((tcmi::tensordot_bits_tiny_kernel<(1 << MV), (1 << NV)>), grid, block, 0, st,                      \
                       reinterpret_cast<const float2*>(a), reinterpret_cast<const float2*>(b), reinterpret_cast<float2*>(c), \
                       lm, ln, nk, pa, pb, po);
    TCMI_TT(0, 0) TCMI_TT(0, 1) TCMI_TT(0, 2) TCMI_TT(0, 3) TCMI_TT(1, 0) TCMI_TT(1, 1) TCMI_TT(1, 2) TCMI_TT(1, 3)
    TCMI_TT(2, 0) TCMI_TT(2, 1) TCMI_TT(2, 2) TCMI_TT(2, 3) TCMI_TT(3, 0) TCMI_TT(3, 1) TCMI_TT(3, 2) TCMI_TT(3, 3)
#undef TCMI_TT
    hipError_t te = hipGetLastError();
    if (te != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(te));
    return TCMI_OK;
  }
  const long long gx = (N + TCMI_BN - 1) / TCMI_BN, gy_all = (M + TCMI_BM - 1) / TCMI_BM;
  const long long gy = gy_all < 65535 ? gy_all : 65535, gz = (gy_all + 65534) / 65535;
  if (gx > 2147483647ll || gz > 65535) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_tensordot_bits: grid too large");
  int ksplit = 1;
  long long kchunk = K;
  const long long tiles = gx * gy_all;
  // few output tiles, long contraction (the closing steps of a sliced network and most steps of its reverse sweep: a
  // rank-20 x rank-18 step over 11 axes is 16 tiles): split K until about a thousand workgroups stream the operands,
  // chunks of at least 128 k values (8 K steps; the first policy -- K >= 4096, chunks >= 1024 --
  // left such steps on 8 ... 64 workgroups, 100 ... 470 us each)
  const long long kmin = 128 > TCMI_CBK ? 128 : TCMI_CBK;
  if (tiles < 512 && K >= 2 * kmin) {
    long long want = (1024 + tiles - 1) / tiles;
    if (want > K / kmin) want = K / kmin;
    if (want * gz > 65535) want = 65535 / gz;
    if (want > 1) {
      kchunk = ((K + want - 1) / want + TCMI_CBK - 1) / TCMI_CBK * TCMI_CBK;
      ksplit = (int)((K + kchunk - 1) / kchunk);
      hipError_t me = tcmi::zero_async(c, (size_t)(M * N) * sizeof(float2), st);
      if (me != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(me));
    }
  }
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)(gz * ksplit)), block(256, 1, 1);
#define TCMI_TB(KA, KB)                                                                                               \
  if (ma == KA && mb == KB)                                                                                           \
    hipLaunchKernelGGL((tcmi::cgemm_bits_kernel<KA, KB>), grid, block, 0, st, reinterpret_cast<const float2*>(a),       \
                       reinterpret_cast<const float2*>(b), reinterpret_cast<float2*>(c), lm, ln, nk, pa, pb, ksplit, kchunk, \
                       po, out_axes ? 1 : 0);
  TCMI_TB(0, 0) TCMI_TB(0, 1) TCMI_TB(0, 2) TCMI_TB(1, 0) TCMI_TB(1, 1) TCMI_TB(1, 2) TCMI_TB(2, 0) TCMI_TB(2, 1) TCMI_TB(2, 2)
#undef TCMI_TB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

int tcmi_cgemm(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
               long long strideA, long long strideB, long long strideC, int trans_a, int dtype,
               void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || K > (1ll << 30))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: bad argument");
  if (dtype == TCMI_C64 && M * N >= 1024) {
    const long long gx = (N + TCMI_BN - 1) / TCMI_BN;
    if (batch > 65535) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: grid too large");
    // gridDim.y holds the row tiles: more than 65535 of them (skinny products of a contraction tree, M up to
    // 2^27) are issued as row chunks — plain pointer offsets for a row-major A
    const long long mchunk = 65535ll * TCMI_BM / 2;
    if (M > mchunk && trans_a) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: grid too large (transposed A)");
    // few output tiles and a long K: split K over blockIdx.z, partial products meet in C through atomics
    const long long tiles = gx * ((M + TCMI_BM - 1) / TCMI_BM) * batch;
    int ksplit = 1, kchunk = (int)K;
    if (tiles < 256 && K >= 4096) {
      long long want = (1024 + tiles - 1) / tiles;
      if (want > K / 1024) want = K / 1024;
      if (want * batch > 65535) want = 65535 / batch;
      if (want > 1) {
        kchunk = (int)(((K + want - 1) / want + TCMI_CBK - 1) / TCMI_CBK * TCMI_CBK);
        ksplit = (int)((K + kchunk - 1) / kchunk);
        for (int b = 0; b < batch; ++b) {
          hipError_t me = tcmi::zero_async(reinterpret_cast<float2*>(C) + (size_t)b * strideC, (size_t)(M * N) * sizeof(float2), st);
          if (me != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(me));
        }
      }
    }
    // k-major A, whole tiles, no split-K (the cut join): the LDS-DMA pipelined kernel
    if (trans_a && ksplit == 1 && (M % 64) == 0 && (N % 64) == 0 && (K % TCMI_DMA_BK) == 0 &&
        (M / 64) * (N / 64) < (1ll << 31) && (strideA & 1) == 0 && (strideB & 1) == 0 &&
        (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0) {
      if ((M % 128) == 0 && (N % 128) == 0 && (strideC & 1) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        const int txn = (int)(N / 128), tyn = (int)(M / 128);
        dim3 grid((unsigned)((long long)txn * tyn), (unsigned)batch, 1), block(256, 1, 1);
        static bool attr_set = false;
        if (!attr_set) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(tcmi::cgemm_dma128_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TCMI_T128_STAGE_BYTES) != hipSuccess)
            return tcmi_set_error_(TCMI_ERR_HIP, "tcmi_cgemm: cannot raise the dynamic LDS limit");
          attr_set = true;
        }
        hipLaunchKernelGGL(tcmi::cgemm_dma128_kernel, grid, block, 2 * TCMI_T128_STAGE_BYTES, st,
                           reinterpret_cast<const float2*>(A), reinterpret_cast<const float2*>(B),
                           reinterpret_cast<float2*>(C), (int)M, (int)N, (int)K, strideA, strideB, strideC, txn, tyn);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
        return TCMI_OK;
      }
      const int txn = (int)(N / 64), tyn = (int)(M / 64);
      dim3 grid((unsigned)((long long)txn * tyn), (unsigned)batch, 1), block(256, 1, 1);
#define TCMI_DMA_LAUNCH(S)                                                                                           \
  {                                                                                                                  \
    if ((S) * TCMI_DMA_STAGE_BYTES > 48 * 1024)                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tcmi::cgemm_dma_kernel<S>),                               \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (S) * TCMI_DMA_STAGE_BYTES);                  \
    hipLaunchKernelGGL(tcmi::cgemm_dma_kernel<S>, grid, block, (S) * TCMI_DMA_STAGE_BYTES, st,                      \
                       reinterpret_cast<const float2*>(A), reinterpret_cast<const float2*>(B),                      \
                       reinterpret_cast<float2*>(C), (int)M, (int)N, (int)K, strideA, strideB, strideC, txn, tyn);  \
  }
      TCMI_DMA_LAUNCH(3)      // ring depth 3 (2 and 4 were measured slower, round 3)
#undef TCMI_DMA_LAUNCH
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
      return TCMI_OK;
    }
    for (long long m0 = 0; m0 < M; m0 += mchunk) {
      const long long mm = (M - m0 < mchunk) ? (M - m0) : mchunk;
      const float2* Ap = reinterpret_cast<const float2*>(A) + (trans_a ? 0 : m0 * K);
      float2* Cp = reinterpret_cast<float2*>(C) + m0 * N;
      dim3 grid((unsigned)gx, (unsigned)((mm + TCMI_BM - 1) / TCMI_BM), (unsigned)(batch * ksplit)), block(256, 1, 1);
      if (trans_a)
        hipLaunchKernelGGL(tcmi::cgemm_mfma_kernel<true>, grid, block, 0, st, Ap, reinterpret_cast<const float2*>(B), Cp,
                           (int)mm, (int)N, (int)K, strideA, strideB, strideC, ksplit, kchunk);
      else
        hipLaunchKernelGGL(tcmi::cgemm_mfma_kernel<false>, grid, block, 0, st, Ap, reinterpret_cast<const float2*>(B), Cp,
                           (int)mm, (int)N, (int)K, strideA, strideB, strideC, ksplit, kchunk);
    }
  } else if (dtype == TCMI_C128 && M * N >= 1024) {
    const long long gx = (N + TCMI_BN - 1) / TCMI_BN;
    if (batch > 65535) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: grid too large");
    const long long mchunk = 65535ll * TCMI_BM / 2;
    if (M > mchunk && trans_a) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: grid too large (transposed A)");
    for (long long m0 = 0; m0 < M; m0 += mchunk) {
      const long long mm = (M - m0 < mchunk) ? (M - m0) : mchunk;
      const double2* Ap = reinterpret_cast<const double2*>(A) + (trans_a ? 0 : m0 * K);
      double2* Cp = reinterpret_cast<double2*>(C) + m0 * N;
      dim3 grid((unsigned)gx, (unsigned)((mm + TCMI_BM - 1) / TCMI_BM), (unsigned)batch), block(256, 1, 1);
      if (trans_a)
        hipLaunchKernelGGL(tcmi::zgemm_mfma_kernel<true>, grid, block, 0, st, Ap, reinterpret_cast<const double2*>(B), Cp,
                           (int)mm, (int)N, (int)K, strideA, strideB, strideC);
      else
        hipLaunchKernelGGL(tcmi::zgemm_mfma_kernel<false>, grid, block, 0, st, Ap, reinterpret_cast<const double2*>(B), Cp,
                           (int)mm, (int)N, (int)K, strideA, strideB, strideC);
    }
  } else if (M * N < 1024 && K >= 2048 && (dtype == TCMI_C64 || dtype == TCMI_C128)) {
    const long long total = M * N;
    int to_log2 = 0;
    while ((1 << to_log2) < total && to_log2 < 8) ++to_log2;
    long long kchunk = (K + 4095) / 4096;
    if (kchunk < 1024) kchunk = 1024;
    const long long chunks = (K + kchunk - 1) / kchunk;
    const size_t esz = dtype == TCMI_C64 ? 8 : 16;
    for (int b = 0; b < batch; ++b) {
      hipError_t me = tcmi::zero_async(reinterpret_cast<char*>(C) + (size_t)b * strideC * esz, (size_t)total * esz, st);
      if (me != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(me));
    }
    const bool pow2 = (M & (M - 1)) == 0 && (N & (N - 1)) == 0;
    if (pow2 && M <= 8 && N <= 8) {
      if (dtype == TCMI_C64) tcmi::launch_tinyout<float>(A, B, C, M, N, K, batch, strideA, strideB, strideC, trans_a, st);
      else tcmi::launch_tinyout<double>(A, B, C, M, N, K, batch, strideA, strideB, strideC, trans_a, st);
      hipError_t te = hipGetLastError();
      if (te != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(te));
      return TCMI_OK;
    }
    dim3 grid((unsigned)chunks, 1, (unsigned)batch), block(256, 1, 1);
    if (dtype == TCMI_C64)
      hipLaunchKernelGGL(tcmi::cgemm_splitk_kernel<float>, grid, block, 0, st, reinterpret_cast<const float2*>(A),
                         reinterpret_cast<const float2*>(B), reinterpret_cast<float2*>(C), M, N, K, strideA, strideB,
                         strideC, trans_a, kchunk, to_log2);
    else
      hipLaunchKernelGGL(tcmi::cgemm_splitk_kernel<double>, grid, block, 0, st, reinterpret_cast<const double2*>(A),
                         reinterpret_cast<const double2*>(B), reinterpret_cast<double2*>(C), M, N, K, strideA,
                         strideB, strideC, trans_a, kchunk, to_log2);
  } else {
    const long long total = M * N;
    unsigned gx = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    dim3 grid(gx, 1, (unsigned)batch), block(256, 1, 1);
    if (dtype == TCMI_C64)
      hipLaunchKernelGGL(tcmi::cgemm_simple_kernel<float>, grid, block, 0, st, reinterpret_cast<const float2*>(A),
                         reinterpret_cast<const float2*>(B), reinterpret_cast<float2*>(C), M, N, (int)K, strideA,
                         strideB, strideC, trans_a);
    else if (dtype == TCMI_C128)
      hipLaunchKernelGGL(tcmi::cgemm_simple_kernel<double>, grid, block, 0, st, reinterpret_cast<const double2*>(A),
                         reinterpret_cast<const double2*>(B), reinterpret_cast<double2*>(C), M, N, (int)K, strideA,
                         strideB, strideC, trans_a);
    else
      return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm: bad dtype");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

}  // extern "C"
