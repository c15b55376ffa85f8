// Which formulation of the complex 2x2 gate is fastest on gfx950?  (a) scalar v_fma_f32 with SGPR
// coefficients (current kernel), (b) packed v_pk_fma_f32 on (re,im) pairs with op_sel swizzles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define NR 32
#define ITERS 512

// ---- (a) scalar -----------------------------------------------------------------------------
template <int J, int KIND>
__device__ __forceinline__ void g1_scalar(float2 (&a)[NR], const float* m) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    const float2 x = a[r], y = a[r1];
    if (KIND == 2) {
      a[r].x = __builtin_fmaf(-m[3], y.y, m[0] * x.x);
      a[r].y = __builtin_fmaf(m[3], y.x, m[0] * x.y);
      a[r1].x = __builtin_fmaf(-m[5], x.y, m[6] * y.x);
      a[r1].y = __builtin_fmaf(m[5], x.x, m[6] * y.y);
    } else {
      float re0 = m[0] * x.x, im0 = m[0] * x.y, re1 = m[4] * x.x, im1 = m[4] * x.y;
      re0 = __builtin_fmaf(-m[1], x.y, re0); im0 = __builtin_fmaf(m[1], x.x, im0);
      re1 = __builtin_fmaf(-m[5], x.y, re1); im1 = __builtin_fmaf(m[5], x.x, im1);
      re0 = __builtin_fmaf(m[2], y.x, re0); re0 = __builtin_fmaf(-m[3], y.y, re0);
      im0 = __builtin_fmaf(m[2], y.y, im0); im0 = __builtin_fmaf(m[3], y.x, im0);
      re1 = __builtin_fmaf(m[6], y.x, re1); re1 = __builtin_fmaf(-m[7], y.y, re1);
      im1 = __builtin_fmaf(m[6], y.y, im1); im1 = __builtin_fmaf(m[7], y.x, im1);
      a[r].x = re0; a[r].y = im0; a[r1].x = re1; a[r1].y = im1;
    }
  }
}
// ---- (b) packed: c = mr*(x.re,x.im) + mi*(-x.im, x.re) ---------------------------------------
__device__ __forceinline__ f2 swapneg(f2 v) { f2 r; r.x = -v.y; r.y = v.x; return r; }  // i*v
template <int J, int KIND>
__device__ __forceinline__ void g1_packed(f2 (&a)[NR], const float* m) {
  const f2 m00r = {m[0], m[0]}, m00i = {m[1], m[1]}, m01r = {m[2], m[2]}, m01i = {m[3], m[3]};
  const f2 m10r = {m[4], m[4]}, m10i = {m[5], m[5]}, m11r = {m[6], m[6]}, m11i = {m[7], m[7]};
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    const f2 x = a[r], y = a[r1];
    if (KIND == 2) {
      a[r] = __builtin_elementwise_fma(m01i, swapneg(y), m00r * x);
      a[r1] = __builtin_elementwise_fma(m10i, swapneg(x), m11r * y);
    } else {
      f2 o0 = m00r * x, o1 = m10r * x;
      o0 = __builtin_elementwise_fma(m00i, swapneg(x), o0);
      o1 = __builtin_elementwise_fma(m10i, swapneg(x), o1);
      o0 = __builtin_elementwise_fma(m01r, y, o0);
      o1 = __builtin_elementwise_fma(m11r, y, o1);
      o0 = __builtin_elementwise_fma(m01i, swapneg(y), o0);
      o1 = __builtin_elementwise_fma(m11i, swapneg(y), o1);
      a[r] = o0; a[r1] = o1;
    }
  }
}

template <int MODE, int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ mtab) {
  const float __attribute__((address_space(4)))* mt = (const float __attribute__((address_space(4)))*)mtab;
  float m[8];
  for (int i = 0; i < 8; ++i) m[i] = mt[i];
  if (MODE == 0) {
    float2 a[NR];
    for (int r = 0; r < NR; ++r) { a[r].x = threadIdx.x * 1e-3f + r; a[r].y = r * 0.5f; }
    for (int it = 0; it < ITERS; ++it) {
      g1_scalar<0, KIND>(a, m); g1_scalar<1, KIND>(a, m); g1_scalar<2, KIND>(a, m); g1_scalar<3, KIND>(a, m); g1_scalar<4, KIND>(a, m);
    }
    float s = 0; for (int r = 0; r < NR; ++r) s += a[r].x + a[r].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f2 a[NR];
    for (int r = 0; r < NR; ++r) { a[r].x = threadIdx.x * 1e-3f + r; a[r].y = r * 0.5f; }
    for (int it = 0; it < ITERS; ++it) {
      g1_packed<0, KIND>(a, m); g1_packed<1, KIND>(a, m); g1_packed<2, KIND>(a, m); g1_packed<3, KIND>(a, m); g1_packed<4, KIND>(a, m);
    }
    float s = 0; for (int r = 0; r < NR; ++r) s += a[r].x + a[r].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
template <int MODE, int KIND> void run(const char* name) {
  float *d, *m; hipMalloc(&d, 256 * 2048 * 4); hipMalloc(&m, 64);
  float hm[8] = {0.9f, 0.f, 0.f, -0.43f, 0.f, -0.43f, 0.9f, 0.f};
  if (KIND == 0) { float g[8] = {0.6f, 0.2f, -0.3f, 0.7f, 0.3f, 0.7f, 0.6f, -0.2f}; for (int i = 0; i < 8; ++i) hm[i] = g[i]; }
  hipMemcpy(m, hm, 32, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, KIND><<<2048, 256>>>(d, m);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<MODE, KIND><<<2048, 256>>>(d, m);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  // gates applied: ITERS*5 per thread-array; amplitude-gates = 2048*256*32 * ITERS*5
  double ag = 2048.0 * 256 * 32 * ITERS * 5;
  // cycles per (wave, gate): 2048 blocks*4 waves over 1024 SIMDs => 8 waves per SIMD sequentially
  double cyc = ms * 1e-3 * 2.4e9 / (8.0 * ITERS * 5);
  printf("%-34s %.3f ms  %.2f T amp-gates/s   %.0f cyc per wave-gate (16 pairs) at 2.4 GHz\n", name, ms, ag / ms / 1e9, cyc);
  hipFree(d); hipFree(m);
}
int main() {
  run<0, 2>("scalar fma, rx-like (8 ops/pair)");
  run<1, 2>("packed fma, rx-like (4 pk/pair)");
  run<0, 0>("scalar fma, general (16 ops/pair)");
  run<1, 0>("packed fma, general (8 pk/pair)");
  return 0;
}
