// complex64 GEMM on the bf16 / f16 matrix pipes with f32 accuracy (the cut-contraction join, reference circuit.py:701-721 ->
// cons.py:948 backend.tensordot of complex64 operands).
//
// gfx950 has no xf32 MFMA and the exact-f32 MFMA runs at the vector rate (157 TFLOP/s); the bf16 MFMA is 16 x faster.
// Every f32 operand value is cut into THREE bf16 pieces, x = x0 + x1 + x2 exactly (8 + 8 + 8 significand bits, each
// piece the round-to-nearest bf16 of what the previous pieces left; bf16 has the exponent range of f32), and a real
// product is the six piece products of order <= 2^-16:
//     x y ~= x0 y0 + (x0 y1 + x1 y0) + (x0 y2 + x1 y1 + x2 y0),
// each exact in the f32 accumulator of v_mfma_f32_32x32x16_bf16.  Dropped: x1 y2 + x2 y1 + x2 y2, at most
// (2 * 2^-24 + 2^-32) |x y| -- the size of ONE f32 rounding of the product, which the f32 MFMA path pays as well.
// tests/test_gpu_gemm_split.py and scripts/gpu_gemm_split.py measure both paths against a float64 product: mean error
// relative to sum |a||b| 2.2e-8 here, 2.6e-8 for the f32 MFMA kernel (max 2.2e-7 / 3.2e-7), also for operands graded
// over twelve decades.  Six bf16 MFMAs of K = 16 replace eight f32 MFMAs of K = 2: 16 / 6 = 2.7 x the exact-f32 rate.
// Gauss's three real products per complex product as in tcmi_cgemm ({re, im, re + im} of each operand).
//
// cgemm_split_kernel: persistent workgroups (one per CU: 144 KiB of LDS), 128 x 128 tile per workgroup, 64 x 64 per
// wave (2 x 2 MFMA tiles x 3 products = 12 f32x16 accumulators in AGPRs), k in steps of 16 (one MFMA).  Operands are
// read as f32: every thread loads 16 elements of the block after next straight into registers (8 x 16 bytes: two
// neighbouring rows of eight k), cuts the block that arrived a step earlier into its nine planes ({re, im, re + im} x
// three pieces) with ~300 VALU instructions placed five behind each MFMA of the step
// (sched_group_barrier), and writes them to the idle plane stage (18 ds_write_b128, conflict-free).  The MFMA
// fragments are 36 ds_read_b128 per wave and step (64 lanes read 1 KiB contiguous: conflict-free).  The stage layout
// is [operand][product t][piece s][k half][row position][8 k]; rows and columns of a tile sit in it interleaved
// (position p of a 64-group = row 2 (p % 32) + p / 32): the two rows a thread converts are 32 positions apart, so
// its 16-byte writes are contiguous across lanes, and a lane's results for the column tiles v = 0, 1 are neighbours
// in C (16-byte stores).  The last step of a tile already cuts block 0 of the next tile: only the result stores
// stand between two tiles.
//
// What was measured on the way (M = N = 4096, K = 256, batch 8; scripts/gpu_gemm_split_modes.py):
//   * operands pre-cut by a separate pass into bf16 planes and streamed by LDS-DMA (18 bytes per element): the DMA path
//     (72 KiB per step and workgroup), not the MFMA pipe, set the pace -- 1.35 us per step with the MFMAs removed, 1.31
//     with the DMA removed, 1.68 together; 1.26 ms + 0.10 ms pre-pass.
//   * one tile per workgroup: 6.4 us of every 38 us workgroup life were prologue and epilogue; persistent: 2.3 us.
//   * the shader clock under this load is 2.0 GHz (s_memtime / s_memrealtime), the MFMA floor of a step 1.15 us; the
//     step takes 1.75 us: MFMAs and fragment reads alone 1.36, the conversion VALU adds 0.33 (2.2 cycles each; an
//     independent VALU instruction beside an MFMA costs 1.1, a dependent one 2.5, scripts/ubench/mfma_valu_shadow.hip),
//     the plane writes 0.11.
//   Result 1.13 ms against 1.74 ms of cgemm_dma128_kernel on the same random operands (1.54x).
//
// EPI = 1 (tcmi_cgemm_split_epi): a 4 x 4 complex matrix X[b] per batch member is applied to the product before it is
// stored -- the deferred last crossing gate of the cut contraction (tcmi/cut.py), acting on the lowest row bit u and
// the lowest column bit v of the product.  The MFMA result layout puts the four (u, v) elements of every (row pair,
// column pair) into ONE thread (the 2 x 2 MFMA tiles of a wave interleave rows and columns), so the epilogue is 16
// complex multiply-adds per four results in registers (32 v_pk_fma_f32) and no data moves.  The product's column index
// is the caller's column index rotated left by one bit (B's columns come from a half-circuit whose first qubit was
// labelled last): column c of the product is stored at (c >> 1) | ((c & 1) * N / 2), as 8-byte stores that are 256
// bytes contiguous per row over 32 lanes.
//
// NP = 2 (tcmi_cgemm_split_f16, round 6): the same kernel on the f16 matrix pipe with TWO pieces per operand value, for
// operands of known magnitude (the comment in front of the kernel template has the arithmetic): six plane blocks per operand
// instead of nine, 36 MFMAs per step instead of 72, each product's MFMAs one phase behind its fragment reads.  2.51 -> 1.8 ms
// per 32-circuit launch of the headline's join; what bounds it then -- the issue of a step's ~240 instructions by one wave per
// SIMD, the store issue of the tile transition -- and the rebuilds that did not move it: DESIGN.md section 2b.
//
// (EPI = 2, round 5: the tail of TWO deferred crossing gates as a gate program over four index bits of the product, K = 64
// on config 2 -- 21.6 us per tile against 20.2 with one deferred gate: the program's ~3500 vector instructions per tile cost
// what the four k-steps saved.  Measured, lost, removed in round 6; DESIGN.md section 2b keeps the numbers.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/tcmi.h"
#include "tcmi_dev.h"

extern "C" int tcmi_set_error_(int code, const char* msg);

namespace tcmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* slptr_t;

constexpr int SPLIT_BK = 16;                               // k per step = one MFMA
constexpr int SPLIT_BLOCK_BYTES = 128 * SPLIT_BK * 2;      // one (product, piece) block of a 128-row tile: 4 KiB
constexpr int SPLIT_STAGE_BYTES = 18 * SPLIT_BLOCK_BYTES;  // three bf16 pieces: A 9 blocks, B 9 blocks
constexpr int split_stage_bytes(int np) { return 6 * np * SPLIT_BLOCK_BYTES; }   // NP pieces: 3 products x NP x {A, B}

#ifndef TCMI_S2_VPM
#define TCMI_S2_VPM 5
#endif
#ifndef TCMI_S2_PIPE
#define TCMI_S2_PIPE 1      // NP = 2: MFMAs one phase behind their fragment reads (0: the three-piece kernel's step)
#endif
#ifndef TCMI_S2_VPM2
#define TCMI_S2_VPM2 7      // NP = 2: vector instructions of the cut behind each of a phase's 12 MFMAs
#endif
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ uint32_t cvt_pk_f16(float lo, float hi) {
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_));
}
template <int NP>
__device__ __forceinline__ f32x16 split_mfma(const f32x4_ a, const f32x4_ b, const f32x16 c) {
  if (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// NP = 3: three bf16 pieces per operand value (header).  NP = 2: TWO f16 pieces of the value times a power of two the
// caller chose (scale_a for A, scale_b for B: |scale * x| and |scale * (re + im)| must stay below 65504), x scale = h + l
// with h the round-to-nearest f16 and l the round-to-nearest f16 of what h left (11 + 11 significand bits and the sign of l:
// |x scale - h - l| <= 2^-22 |x scale|), and a real product is h h' + h l' + l h' -- THREE f16 MFMAs, each exact in the f32
// accumulator, where the bf16 cut needs six; dropped: l l' and the residues, <= 3 * 2^-22 |x y| in the worst case and
// measured (tests/test_gpu_gemm_split.py) at the error of the f32 MFMA kernel.  The results leave scaled by
// so = 1 / (scale_a scale_b), exact for powers of two.
template <int MODE, int EPI, int NP>
__global__ __launch_bounds__(256, 1) void cgemm_split_kernel(const float2* __restrict__ A, const float2* __restrict__ B,
                                                              float2* __restrict__ C, int M, int N, int K, long long sA,
                                                              long long sB, long long sC, int tiles_x, int tiles_y, int batch,
                                                              const float2* __restrict__ X, float scale_a, float scale_b,
                                                              float so) {
  constexpr int NPL = 3 * NP;                                     // plane blocks per operand
  constexpr int STAGE_B = 2 * NPL * SPLIT_BLOCK_BYTES;            // one stage: A planes, then B planes
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = K / SPLIT_BK;
  const unsigned ntile = (unsigned)tiles_x * (unsigned)tiles_y, nwork = ntile * (unsigned)batch;
  unsigned long long tc0 = 0, tr0 = 0;
  if (MODE == 4) {
    tc0 = __builtin_amdgcn_s_memtime();
    tr0 = __builtin_amdgcn_s_memrealtime();
  }
  // persistent workgroups: work item w = (batch member, tile); tiles in the XCD-aware order of cgemm_dma128_kernel
  // (consecutive workgroup ids sit on different XCDs: item w of a batch member is the (w / 8)-th tile of strip w % 8)
#define TCMI_S2_TILE(W, M0, N0, BI)                                                  \
  {                                                                                  \
    const unsigned w_ = (W) < nwork ? (W) : nwork - 1;                               \
    const unsigned L_ = w_ % ntile;                                                  \
    BI = (int)(w_ / ntile);                                                          \
    int tx_, ty_;                                                                    \
    if ((tiles_x & 7) == 0 && (ntile & 7) == 0) {                                    \
      const unsigned xcd_ = L_ & 7u, j_ = L_ >> 3, sw_ = (unsigned)tiles_x >> 3;     \
      tx_ = (int)(xcd_ * sw_ + j_ % sw_);                                            \
      ty_ = (int)(j_ / sw_);                                                         \
    } else {                                                                         \
      tx_ = (int)(L_ % (unsigned)tiles_x);                                           \
      ty_ = (int)(L_ / (unsigned)tiles_x);                                           \
    }                                                                                \
    M0 = (long long)ty_ * 128;                                                       \
    N0 = (long long)tx_ * 128;                                                       \
  }
  unsigned work = blockIdx.x;
  long long m0, n0, m1, n1;
  int bi, bi1;
  TCMI_S2_TILE(work, m0, n0, bi)
  TCMI_S2_TILE(work + gridDim.x, m1, n1, bi1)
  // loader role of the wave: operand (waves 0, 1: A; 2, 3: B) and k half; lane = row pair
  const int lop = wave >> 1, lkg = wave & 1;
  const long long R = lop ? N : M;
  const int lsig = lane;
  const long long lofs = 2 * lsig + (long long)(lkg * 8) * R;
  const float2* src = (lop ? B + (long long)bi * sB + n0 : A + (long long)bi * sA + m0) + lofs;       // this tile
  const float2* src1 = (lop ? B + (long long)bi1 * sB + n1 : A + (long long)bi1 * sA + m1) + lofs;   // the next one
  // this thread's write slot inside a plane block: k half, then position of row 2 lane (+ 32 positions for row 2 lane + 1)
  const uint32_t wofs = (uint32_t)(lop * NPL * SPLIT_BLOCK_BYTES + lkg * 2048 + ((lane >> 5) * 64 + (lane & 31)) * 16);
  const float lsc = lop ? scale_b : scale_a;
  // -1 for the powers of two the launcher admits, but not a constant the compiler could fold: the residue x - h then is
  // ONE mixed-precision fma reading the f16 half where it is, instead of a conversion and a subtraction
  const float fm1 = -(so * scale_a * scale_b);
  f32x4_ g0[8], g1[8];
#define TCMI_S2_LOAD(G, KT)                                                                       \
  {                                                                                               \
    const float2* s_ = (KT) < nk ? src + (long long)((KT) * SPLIT_BK) * R : src1 + (long long)(((KT) - nk) * SPLIT_BK) * R; \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) G[j] = *reinterpret_cast<const f32x4_*>(s_ + (long long)j * R);          \
  }
  // product T (0: re, 1: im, 2: re + im) of the block in G -> its three piece blocks of stage ST
#define TCMI_S2_CONVERT(G, ST, T)                                                                                     \
  {                                                                                                                   \
    /* both rows' eight k pairs move through the cut in lock step (8 independent chains: a dependent VALU     */      \
    /* instruction issued right behind its producer costs 2.5 cycles beside an MFMA, an independent one 1.1,   */      \
    /* scripts/ubench/mfma_valu_shadow.hip) */                                                                       \
    float x_[8], y_[8];                                                                                               \
    uint32_t p0_[8], p1_[8], p2_[8];                                                                                  \
    if (NP == 2 && (T) == 0) {   /* the block is scaled once, in place, before its first product is cut */           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) G[j] *= lsc;                                                      \
    }                                                                                                                 \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                   \
      const int h = e >> 2, jp = e & 3;                                                                               \
      x_[e] = (T) == 0 ? G[2 * jp][2 * h] : ((T) == 1 ? G[2 * jp][2 * h + 1] : G[2 * jp][2 * h] + G[2 * jp][2 * h + 1]); \
      y_[e] = (T) == 0 ? G[2 * jp + 1][2 * h]                                                                         \
                       : ((T) == 1 ? G[2 * jp + 1][2 * h + 1] : G[2 * jp + 1][2 * h] + G[2 * jp + 1][2 * h + 1]);     \
    }                                                                                                                 \
    if (MODE == 6) {                                                                                                  \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        p0_[e] = __float_as_uint(x_[e]);                                                                              \
        p1_[e] = __float_as_uint(y_[e]);                                                                              \
        p2_[e] = p0_[e];                                                                                              \
      }                                                                                                               \
    } else if (NP == 2) {                                                                                             \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p0_[e] = cvt_pk_f16(x_[e], y_[e]);                                \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        const f16x2_ h_ = __builtin_bit_cast(f16x2_, p0_[e]);                                                         \
        x_[e] = __builtin_fmaf((float)h_[0], fm1, x_[e]);      /* v_fma_mix_f32: the f16 half read in place */        \
        y_[e] = __builtin_fmaf((float)h_[1], fm1, y_[e]);                                                             \
      }                                                                                                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p1_[e] = cvt_pk_f16(x_[e], y_[e]);                                \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p2_[e] = 0;                                                       \
    } else {                                                                                                          \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p0_[e] = cvt_pk_bf16(x_[e], y_[e]);                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        x_[e] -= __uint_as_float(p0_[e] << 16);                                                                       \
        y_[e] -= __uint_as_float(p0_[e] & 0xffff0000u);                                                               \
      }                                                                                                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p1_[e] = cvt_pk_bf16(x_[e], y_[e]);                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                 \
        x_[e] -= __uint_as_float(p1_[e] << 16);                                                                       \
        y_[e] -= __uint_as_float(p1_[e] & 0xffff0000u);                                                               \
      }                                                                                                               \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) p2_[e] = cvt_pk_bf16(x_[e], y_[e]);                               \
    }                                                                                                                 \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                   \
      const u32x4 q0_ = {p0_[4 * h], p0_[4 * h + 1], p0_[4 * h + 2], p0_[4 * h + 3]};                                 \
      const u32x4 q1_ = {p1_[4 * h], p1_[4 * h + 1], p1_[4 * h + 2], p1_[4 * h + 3]};                                 \
      const u32x4 q2_ = {p2_[4 * h], p2_[4 * h + 1], p2_[4 * h + 2], p2_[4 * h + 3]};                                 \
      char* w_ = dsm + (ST) * STAGE_B + wofs + h * 512 + (T) * NP * SPLIT_BLOCK_BYTES;                                \
      if (MODE == 5) {                                                                                                \
        asm volatile("" ::"v"(q0_), "v"(q1_), "v"(q2_));                                                              \
      } else {                                                                                                        \
        *reinterpret_cast<u32x4*>(w_) = q0_;                                                                          \
        *reinterpret_cast<u32x4*>(w_ + SPLIT_BLOCK_BYTES) = q1_;                                                      \
        if (NP == 3) *reinterpret_cast<u32x4*>(w_ + 2 * SPLIT_BLOCK_BYTES) = q2_;                                     \
      }                                                                                                               \
    }                                                                                                                 \
  }
  TCMI_S2_LOAD(g0, 0)
  TCMI_S2_LOAD(g1, 1)
  f32x16 acc[2][2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][t][e] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  TCMI_S2_CONVERT(g0, 0, 0)
  TCMI_S2_CONVERT(g0, 0, 1)
  TCMI_S2_CONVERT(g0, 0, 2)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(slptr_t)dsm;
  const uint32_t fa = lds0 + (uint32_t)((lane >> 5) * 2048 + (wr * 64 + (lane & 31)) * 16);
  const uint32_t fb = lds0 + (uint32_t)(NPL * SPLIT_BLOCK_BYTES + (lane >> 5) * 2048 + (wc * 64 + (lane & 31)) * 16);
  // one 16-k step on stage P: MFMAs of block I, the block after it (in GN) cut into stage 1 - P, block I + 2 requested
  // into GC (whose block is in stage P already)
#define TCMI_S2_SREAD(BUF, T)                                                                                           \
  _Pragma("unroll") for (int s = 0; s < NP; ++s) _Pragma("unroll") for (int u = 0; u < 2; ++u) {                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[BUF][u][s]) : "v"(sa_), "n"(((T) * NP + s) * 4096 + u * 512)); \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[BUF][u][s]) : "v"(sb_), "n"(((T) * NP + s) * 4096 + u * 512)); \
  }
#define TCMI_S2_SWAIT(BUF, NW)                                                                                    \
  if (NP == 3) {                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(%12)"                                                                         \
                 : "+v"(xa[BUF][0][0]), "+v"(xa[BUF][0][1]), "+v"(xa[BUF][0][2]), "+v"(xa[BUF][1][0]),            \
                   "+v"(xa[BUF][1][1]), "+v"(xa[BUF][1][2]), "+v"(xb[BUF][0][0]), "+v"(xb[BUF][0][1]),            \
                   "+v"(xb[BUF][0][2]), "+v"(xb[BUF][1][0]), "+v"(xb[BUF][1][1]), "+v"(xb[BUF][1][2])             \
                 : "n"(NW)                                                                                        \
                 : "memory");                                                                                     \
  } else {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                          \
                 : "+v"(xa[BUF][0][0]), "+v"(xa[BUF][0][1]), "+v"(xa[BUF][1][0]), "+v"(xa[BUF][1][1]),            \
                   "+v"(xb[BUF][0][0]), "+v"(xb[BUF][0][1]), "+v"(xb[BUF][1][0]), "+v"(xb[BUF][1][1])             \
                 : "n"(NW)                                                                                        \
                 : "memory");                                                                                     \
  }
#define TCMI_S2_MF(T, SA, SB)                                                                                       \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
    acc[u][v][T] = split_mfma<NP>(xa[(T) & 1][u][SA], xb[(T) & 1][v][SB], acc[u][v][T]);
  // the first MFMA of a tile into an accumulator starts from the constant 0 (nothing zeroes the accumulators between tiles)
#define TCMI_S2_MF0(T, SA, SB)                                                                                      \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
    acc[u][v][T] = split_mfma<NP>(xa[(T) & 1][u][SA], xb[(T) & 1][v][SB], zero16);
#define TCMI_S2_PHASE(T, GN, P, FIRST)                                                                               \
  {                                                                                                                  \
    if ((T) < 2) {                                                                                                   \
      if ((T) & 1) { TCMI_S2_SREAD(0, (T) + 1) } else { TCMI_S2_SREAD(1, (T) + 1) }                                  \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (MODE != 1) TCMI_S2_CONVERT(GN, 1 - (P), T)                                                                   \
    if (MODE != 2 && NP == 3) {                                                                                      \
      if (FIRST) { TCMI_S2_MF0(T, 0, 2) } else { TCMI_S2_MF(T, 0, 2) }                                                \
      TCMI_S2_MF(T, 1, 1) TCMI_S2_MF(T, 2, 0) TCMI_S2_MF(T, 0, 1) TCMI_S2_MF(T, 1, 0) TCMI_S2_MF(T, 0, 0)             \
    }                                                                                                                \
    if (MODE != 2 && NP == 2) {                                                                                      \
      if (FIRST) { TCMI_S2_MF0(T, 0, 1) } else { TCMI_S2_MF(T, 0, 1) }                                                \
      TCMI_S2_MF(T, 1, 0) TCMI_S2_MF(T, 0, 0)                                                                         \
    }                                                                                                                \
    if (MODE == 0 || MODE >= 3) {                                                                                    \
      /* spread the ~100 VALU instructions of the conversion evenly between the 24 MFMAs (4 ride in each MFMA's */   \
      /* shadow), the six plane writes behind them */                                                               \
      _Pragma("unroll") for (int z = 0; z < 8 * NP; ++z) {                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? TCMI_S2_VPM : TCMI_S2_VPM2, 0);                        \
      }                                                                                                              \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if ((T) < 2) {                                                                                                   \
      /* the 12 fragment reads of the next product are done; this phase's 6 plane writes (issued after them, LDS  */ \
      /* operations complete in order) stay in flight */                                                            \
      if ((T) & 1) { TCMI_S2_SWAIT(0, MODE == 1 ? 0 : 2 * NP) } else { TCMI_S2_SWAIT(1, MODE == 1 ? 0 : 2 * NP) }    \
    }                                                                                                                \
  }
#define TCMI_S2_STEP(I, GC, GN, P, FIRST)                                                                            \
  {                                                                                                                  \
    const uint32_t sa_ = fa + (uint32_t)((P) * STAGE_B), sb_ = fb + (uint32_t)((P) * STAGE_B);                         \
    f32x4_ xa[2][2][3], xb[2][2][3];                                                                                 \
    TCMI_S2_SREAD(0, 0)                                                                                              \
    if (MODE != 1) TCMI_S2_LOAD(GC, (I) + 2)                                                                         \
    TCMI_S2_SWAIT(0, 0)                                                                                              \
    TCMI_S2_PHASE(0, GN, P, FIRST)                                                                                   \
    TCMI_S2_PHASE(1, GN, P, FIRST)                                                                                   \
    TCMI_S2_PHASE(2, GN, P, FIRST)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
  }
  // NP = 2: the same step with the MFMAs of a product running ONE PHASE BEHIND its fragment reads.  With half the matrix
  // instructions per step (36 x 32 cycles) the latencies that the three-piece kernel's 72 MFMAs covered lay bare: the
  // fragment reads of a step's first product behind the barrier, the drain of the plane writes before it (stamps:
  // 2300 cycles per step against an MFMA floor of 1152).  Here phase f of a step cuts product f of the next block (vector
  // pipe, plane writes) beside the MFMAs of product f - 1 -- phase 0 beside product 2 of the PREVIOUS step, whose
  // fragments wait in registers -- and the reads of product f, issued at the start of phase f, have that whole phase to
  // land: nothing of the matrix pipe's work waits for the LDS or for the barrier.  Three fragment buffers (96 VGPRs, what the
  // two three-piece buffers took).
  f32x4_ qa[3][2][2], qb[3][2][2];
#define TCMI_Q_SREAD(T, ST)                                                                                             \
  {                                                                                                                     \
    const uint32_t sa_ = fa + (uint32_t)((ST) * STAGE_B), sb_ = fb + (uint32_t)((ST) * STAGE_B);                         \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int u = 0; u < 2; ++u) {                       \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[T][u][s]) : "v"(sa_), "n"(((T) * 2 + s) * 4096 + u * 512)); \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qb[T][u][s]) : "v"(sb_), "n"(((T) * 2 + s) * 4096 + u * 512)); \
    }                                                                                                                   \
  }
#define TCMI_Q_WAIT(T, NW)                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                            \
               : "+v"(qa[T][0][0]), "+v"(qa[T][0][1]), "+v"(qa[T][1][0]), "+v"(qa[T][1][1]), "+v"(qb[T][0][0]),   \
                 "+v"(qb[T][0][1]), "+v"(qb[T][1][0]), "+v"(qb[T][1][1])                                          \
               : "n"(NW)                                                                                          \
               : "memory");
#define TCMI_Q_MF(T, SA, SB, Z)                                                                                     \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v)                       \
    acc[u][v][T] = split_mfma<2>(qa[T][u][SA], qb[T][v][SB], (Z) ? zero16 : acc[u][v][T]);
  // product TC of block GN cut into stage 1 - P beside the MFMAs of product TM (MZ: 0 none, 1 the first of a tile into its
  // accumulators, 2 accumulating)
#define TCMI_Q_PHASE(TC, TM, MZ, GN, P)                                                                              \
  {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (MODE != 1) TCMI_S2_CONVERT(GN, 1 - (P), TC)                                                                  \
    if (MODE != 2 && (MZ) != 0) {                                                                                      \
      TCMI_Q_MF(TM, 0, 1, (MZ) == 1) TCMI_Q_MF(TM, 1, 0, 0) TCMI_Q_MF(TM, 0, 0, 0)                                    \
      _Pragma("unroll") for (int z = 0; z < 12; ++z) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x002, TCMI_S2_VPM2, 0);                                                \
      }                                                                                                              \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define TCMI_Q_STEP(I, GC, GN, P, MZP, MZ)                                                                        \
  {                                                                                                                  \
    TCMI_Q_SREAD(0, P)                                                                                               \
    if (MODE != 1) TCMI_S2_LOAD(GC, (I) + 2)                                                                         \
    TCMI_Q_PHASE(0, 2, MZP, GN, P)                                                                                   \
    TCMI_Q_WAIT(0, MODE == 1 ? 0 : 4)       /* product 0's reads; this phase's 4 plane writes stay in flight */       \
    TCMI_Q_SREAD(1, P)                                                                                               \
    TCMI_Q_PHASE(1, 0, MZ, GN, P)                                                                       \
    TCMI_Q_WAIT(1, MODE == 1 ? 0 : 4)                                                                                \
    TCMI_Q_SREAD(2, P)                                                                                               \
    TCMI_Q_PHASE(2, 1, MZ, GN, P)                                                                       \
    TCMI_Q_WAIT(2, 0)                                                                                                \
    __builtin_amdgcn_s_barrier();                                                                                    \
  }
  // MODE 10 (probe): cycles of a tile's step 0, step 1, remaining steps and epilogue, summed over the tiles of a workgroup
  unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, ps = 0;
#define TCMI_S2_STAMP(ACC)                                             \
  if (MODE == 10) {                                                    \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    ACC += now_ - ps;                                                  \
    ps = now_;                                                         \
  }
  if (MODE == 10) ps = __builtin_amdgcn_s_memtime();
  for (;;) {
    if (NP == 2 && TCMI_S2_PIPE) {
      TCMI_Q_STEP(0, g0, g1, 0, 0, 1)
      TCMI_S2_STAMP(pt0)
      TCMI_Q_STEP(1, g1, g0, 1, 1, 2)
      TCMI_S2_STAMP(pt1)
      for (int i = 2; i < nk; i += 2) {
        TCMI_Q_STEP(i, g0, g1, 0, 2, 2)
        TCMI_Q_STEP(i + 1, g1, g0, 1, 2, 2)
      }
      if (MODE != 2) { TCMI_Q_MF(2, 0, 1, 0) TCMI_Q_MF(2, 1, 0, 0) TCMI_Q_MF(2, 0, 0, 0) }      // the last step's third product
      TCMI_S2_STAMP(pt2)
    } else {
    TCMI_S2_STEP(0, g0, g1, 0, MODE != 2)
    TCMI_S2_STAMP(pt0)
    TCMI_S2_STEP(1, g1, g0, 1, 0)
    TCMI_S2_STAMP(pt1)
    for (int i = 2; i < nk; i += 2) {
      TCMI_S2_STEP(i, g0, g1, 0, 0)
      TCMI_S2_STEP(i + 1, g1, g0, 1, 0)
    }
    TCMI_S2_STAMP(pt2)
    }
    // the last step cut block 0 of the next tile into stage 0 and blocks 0, 1 of it are (being) loaded: only the
    // results stand between the tiles.  MFMA result element (i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), j = lane & 31)
    // of tile (u, v) is C[wr 64 + 2 i + u][wc 64 + 2 j + v]
    if (EPI) {
      // y[2 u' + v'] = sum_{u, v} X[b][2 u' + v'][2 u + v] c[u][v]; column 2 j + v of the product is column j + v N / 2 of C
      float2* Cb = C + (long long)bi * sC;
      const float2* Xb = X + (long long)__builtin_amdgcn_readfirstlane(bi) * 16;
      f32x2_ xr[16], xi[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float2 x = Xb[e];
        if (NP == 2) {      // the accumulators hold the product times scale_a scale_b
          x.x *= so;
          x.y *= so;
        }
        xr[e] = f32x2_{x.x, x.x};
        xi[e] = f32x2_{x.y, x.y};
      }
      const long long colh = (n0 >> 1) + wc * 32 + (lane & 31);
      const long long half = (long long)N >> 1;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const long long row0 = m0 + wr * 64 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5));
        f32x2_ c[4], cs[4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const float re = acc[u][v][0][reg] - acc[u][v][1][reg];
            const float im = acc[u][v][2][reg] - acc[u][v][0][reg] - acc[u][v][1][reg];
            c[2 * u + v] = f32x2_{re, im};
            cs[2 * u + v] = f32x2_{-im, re};
            if (MODE == 2) {
              acc[u][v][0][reg] = 0.f;
              acc[u][v][1][reg] = 0.f;
              acc[u][v][2][reg] = 0.f;
            }
          }
        f32x2_ yo[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          f32x2_ y = {0.f, 0.f};
          if (MODE == 7) {
            y = c[o] + xr[o];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              y = __builtin_elementwise_fma(xr[4 * o + i], c[i], y);
              y = __builtin_elementwise_fma(xi[4 * o + i], cs[i], y);
            }
          }
          yo[o] = y;
          float2 w;
          w.x = y.x;
          w.y = y.y;
          if (MODE != 8 && (MODE != 9 || y.x == 123.456f)) Cb[(row0 + (o >> 1)) * N + colh + (o & 1) * half] = w;
        }
        if (MODE == 8) {      // probe: the same bytes as 16-byte stores of neighbouring columns (wrong places)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x4_ o4 = {yo[2 * u].x, yo[2 * u].y, yo[2 * u + 1].x, yo[2 * u + 1].y};
            *reinterpret_cast<f32x4_*>(Cb + (row0 + u) * N + n0 + wc * 64 + 2 * (lane & 31)) = o4;
          }
        }
      }
    } else {
      float2* Cb = C + (long long)bi * sC;
      const long long colb = n0 + wc * 64 + 2 * (lane & 31);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const long long row = m0 + wr * 64 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) + u;
          f32x4_ o;
          o.x = acc[u][0][0][reg] - acc[u][0][1][reg];
          o.y = acc[u][0][2][reg] - acc[u][0][0][reg] - acc[u][0][1][reg];
          o.z = acc[u][1][0][reg] - acc[u][1][1][reg];
          o.w = acc[u][1][2][reg] - acc[u][1][0][reg] - acc[u][1][1][reg];
          if (NP == 2) o *= so;
          if (MODE != 3 || o.x == 123.456f) *reinterpret_cast<f32x4_*>(Cb + row * N + colb) = o;
#pragma unroll
          for (int v = 0; v < 2; ++v)
            if (MODE == 2) {
              acc[u][v][0][reg] = 0.f;
              acc[u][v][1][reg] = 0.f;
              acc[u][v][2][reg] = 0.f;
            }
        }
    }
    if (MODE == 4 && blockIdx.x == 100 && tid == 0) {      // probe: a stamp per finished tile into C[1 + tile count]
      float2 o;
      o.x = (float)(__builtin_amdgcn_s_memtime() - tc0);
      o.y = (float)(__builtin_amdgcn_s_memrealtime() - tr0);
      C[1 + work / gridDim.x] = o;
    }
    TCMI_S2_STAMP(pt3)
    work += gridDim.x;
    if (work >= nwork) break;
    m0 = m1; n0 = n1; bi = bi1;
    src = src1;
    TCMI_S2_TILE(work + gridDim.x, m1, n1, bi1)
    src1 = (lop ? B + (long long)bi1 * sB + n1 : A + (long long)bi1 * sA + m1) + lofs;
  }
  if (MODE == 10 && blockIdx.x == 100 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float2 o;
    o.x = (float)pt0; o.y = (float)pt1;
    C[0] = o;
    o.x = (float)pt2; o.y = (float)pt3;
    C[1] = o;
  }
#undef TCMI_Q_STEP
#undef TCMI_Q_PHASE
#undef TCMI_Q_MF
#undef TCMI_Q_WAIT
#undef TCMI_Q_SREAD
#undef TCMI_S2_STAMP
#undef TCMI_S2_STEP
#undef TCMI_S2_PHASE
#undef TCMI_S2_MF0
#undef TCMI_S2_MF
#undef TCMI_S2_SWAIT
#undef TCMI_S2_SREAD
#undef TCMI_S2_CONVERT
#undef TCMI_S2_LOAD
#undef TCMI_S2_TILE
  if (MODE == 4 && blockIdx.x == 100 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long dc = __builtin_amdgcn_s_memtime() - tc0, dr = __builtin_amdgcn_s_memrealtime() - tr0;
    float2 o;
    o.x = (float)dc;
    o.y = (float)dr;
    C[0] = o;
  }
}

}  // namespace tcmi

namespace {

int split_launch(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch, long long strideA,
                 long long strideB, long long strideC, const void* X, void* stream, const char* who, int np = 3,
                 float scale_a = 1.f, float scale_b = 1.f) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef TCMI_SPLIT_PROBE
  // probe builds of the kernel (libtcmi_probe.so only, scripts/gpu_gemm_split_modes.py): 1 no conversion, 2 no MFMA,
  // 3 no result stores, 4 cycle / wall-clock stamps into C[0], 5 conversion without plane writes, 6 plane writes without
  // conversion.  The production library has no such switch: it instantiates cgemm_split_kernel<0, *> alone.
  static const int mode = getenv("TCMI_SPLIT_MODE") ? atoi(getenv("TCMI_SPLIT_MODE")) : 0;
#endif
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || (M % 128) || (N % 128) || (K % 32) || batch > 65535 ||
      M > (1ll << 30) || N > (1ll << 30) || ((strideA | strideB | strideC) & 1) ||
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15) ||
      (reinterpret_cast<uintptr_t>(X) & 7))
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm_split: bad argument (k-major A, M and N multiples of 128, K of 32, "
                                         "16-byte aligned operands)");
  (void)who;
  const int txn = (int)(N / 128), tyn = (int)(M / 128);
  const long long nwork = (long long)txn * tyn * batch;
  if (nwork >= (1ll << 31)) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm_split: too many tiles");
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return tcmi_set_error_(TCMI_ERR_HIP, "tcmi_cgemm_split: cannot query the device");
    ncu = prop.multiProcessorCount;
  }
  const float so = 1.f / (scale_a * scale_b);
#define TCMI_SPLIT_LAUNCH_NP(MODE, EPI, NP)                                                                            \
  {                                                                                                                    \
    static bool attr_set_ = false;                                                                                     \
    if (!attr_set_) {                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(tcmi::cgemm_split_kernel<MODE, EPI, NP>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * tcmi::split_stage_bytes(NP)) != hipSuccess) \
        return tcmi_set_error_(TCMI_ERR_HIP, "tcmi_cgemm_split: cannot raise the dynamic LDS limit");                  \
      attr_set_ = true;                                                                                                \
    }                                                                                                                  \
    hipLaunchKernelGGL((tcmi::cgemm_split_kernel<MODE, EPI, NP>), dim3((unsigned)(nwork < ncu ? nwork : ncu), 1, 1),   \
                       dim3(256), 2 * tcmi::split_stage_bytes(NP), st, reinterpret_cast<const float2*>(A),             \
                       reinterpret_cast<const float2*>(B), reinterpret_cast<float2*>(C), (int)M, (int)N, (int)K, strideA, \
                       strideB, strideC, txn, tyn, batch, reinterpret_cast<const float2*>(X), scale_a, scale_b, so);   \
  }
#define TCMI_SPLIT_LAUNCH(MODE, EPI) TCMI_SPLIT_LAUNCH_NP(MODE, EPI, 3)
  if (np == 2) {
#ifdef TCMI_SPLIT_PROBE
    if (mode == 1 && !X) TCMI_SPLIT_LAUNCH_NP(1, 0, 2)
    else if (mode == 2 && !X) TCMI_SPLIT_LAUNCH_NP(2, 0, 2)
    else if (mode == 3 && !X) TCMI_SPLIT_LAUNCH_NP(3, 0, 2)
    else if (mode == 4 && !X) TCMI_SPLIT_LAUNCH_NP(4, 0, 2)
    else if (mode == 5 && !X) TCMI_SPLIT_LAUNCH_NP(5, 0, 2)
    else if (mode == 10 && !X) TCMI_SPLIT_LAUNCH_NP(10, 0, 2)
    else if (mode == 10 && X) TCMI_SPLIT_LAUNCH_NP(10, 1, 2)
    else if (mode == 7 && X) TCMI_SPLIT_LAUNCH_NP(7, 1, 2)
    else if (mode == 8 && X) TCMI_SPLIT_LAUNCH_NP(8, 1, 2)
    else if (mode == 9 && X) TCMI_SPLIT_LAUNCH_NP(9, 1, 2)
    else
#endif
    if (X) TCMI_SPLIT_LAUNCH_NP(0, 1, 2)
    else TCMI_SPLIT_LAUNCH_NP(0, 0, 2)
  } else if (X) {
#ifdef TCMI_SPLIT_PROBE
    // 7: epilogue without its multiply-adds, 8: its results as 16-byte stores (wrong places), 9: no result stores
    if (mode == 7) TCMI_SPLIT_LAUNCH(7, 1)
    else if (mode == 8) TCMI_SPLIT_LAUNCH(8, 1)
    else if (mode == 9) TCMI_SPLIT_LAUNCH(9, 1)
    else
#endif
    TCMI_SPLIT_LAUNCH(0, 1)
  } else {
#ifdef TCMI_SPLIT_PROBE
    if (mode == 1) TCMI_SPLIT_LAUNCH(1, 0)
    else if (mode == 2) TCMI_SPLIT_LAUNCH(2, 0)
    else if (mode == 3) TCMI_SPLIT_LAUNCH(3, 0)
    else if (mode == 4) TCMI_SPLIT_LAUNCH(4, 0)
    else if (mode == 5) TCMI_SPLIT_LAUNCH(5, 0)
    else if (mode == 6) TCMI_SPLIT_LAUNCH(6, 0)
    else
#endif
    TCMI_SPLIT_LAUNCH(0, 0)
  }
#undef TCMI_SPLIT_LAUNCH
#undef TCMI_SPLIT_LAUNCH_NP
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tcmi_set_error_(TCMI_ERR_HIP, hipGetErrorString(e));
  return TCMI_OK;
}

}  // namespace

extern "C" {

int tcmi_cgemm_split(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                     long long strideA, long long strideB, long long strideC, void* stream) {
  return split_launch(A, B, C, M, N, K, batch, strideA, strideB, strideC, nullptr, stream, "tcmi_cgemm_split");
}

int tcmi_cgemm_split_epi(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                         long long strideA, long long strideB, long long strideC, const void* X, void* stream) {
  if (!X) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm_split_epi: X is null");
  return split_launch(A, B, C, M, N, K, batch, strideA, strideB, strideC, X, stream, "tcmi_cgemm_split_epi");
}

int tcmi_cgemm_split_f16(const void* A, const void* B, void* C, long long M, long long N, long long K, int batch,
                         long long strideA, long long strideB, long long strideC, const void* X, float scale_a, float scale_b,
                         void* stream) {
  // powers of two only: the scaling and the un-scaling of the result are exact then
  int ea = 0, eb = 0;
  if (!(scale_a > 0.f) || !(scale_b > 0.f) || frexpf(scale_a, &ea) != 0.5f || frexpf(scale_b, &eb) != 0.5f || ea + eb < -120 ||
      ea + eb > 120)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_cgemm_split_f16: scale_a and scale_b must be powers of two");
  return split_launch(A, B, C, M, N, K, batch, strideA, strideB, strideC, X, stream, "tcmi_cgemm_split_f16", 2, scale_a,
                      scale_b);
}

}  // extern "C"
