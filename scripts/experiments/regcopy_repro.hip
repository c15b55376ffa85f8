#include <hip/hip_runtime.h>
#include <stdint.h>
#define NR 32
#define R 5
typedef const int __attribute__((address_space(4)))* KInt;
typedef const float __attribute__((address_space(4)))* KF;
template <int J> __device__ __forceinline__ void shear4(float2 (&a)[NR], float p, float w) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    a[r].x = __builtin_fmaf(-p, a[r1].y, a[r].x);
    a[r].y = __builtin_fmaf(p, a[r1].x, a[r].y);
    a[r1].x = __builtin_fmaf(-w, a[r].y, a[r1].x);
    a[r1].y = __builtin_fmaf(w, a[r].x, a[r1].y);
    a[r].x = __builtin_fmaf(-p, a[r1].y, a[r].x);
    a[r].y = __builtin_fmaf(p, a[r1].x, a[r].y);
  }
}
__device__ __forceinline__ void disp(float2 (&a)[NR], int j, float p, float w) {
  switch (j) {
    case 0: shear4<0>(a, p, w); break;
    case 1: shear4<1>(a, p, w); break;
    case 2: shear4<2>(a, p, w); break;
    case 3: shear4<3>(a, p, w); break;
    case 4: shear4<4>(a, p, w); break;
    default: break;
  }
}
__global__ __launch_bounds__(256, 1) void k(float2* __restrict__ st, const int* __restrict__ dg, const float* __restrict__ tg) {
  const KInt desc = (KInt)dg; const KF tab = (KF)tg;
  float2 a[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) a[r] = st[(blockIdx.x * 256 + threadIdx.x) * NR + r];
  const int nops = desc[0];
  int q = 1;
#pragma unroll 1
  for (int o = 0; o < nops; ++o) {
#if VARIANT == 0
    const int j = desc[q]; const float p = tab[2 * o], w = tab[2 * o + 1];
    disp(a, j, p, w);
    q += 1;
#else
    const int mk = desc[q];
    float mm[R][2];
#pragma unroll
    for (int j = 0; j < R; ++j) { mm[j][0] = tab[10 * o + 2 * j]; mm[j][1] = tab[10 * o + 2 * j + 1]; }
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
      if (!((mk >> j) & 1)) continue;
      float p = 0, w = 0;
#pragma unroll
      for (int jj = 0; jj < R; ++jj) if (jj == j) { p = mm[jj][0]; w = mm[jj][1]; }
      disp(a, j, p, w);
    }
    q += 1;
#endif
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) st[(blockIdx.x * 256 + threadIdx.x) * NR + r] = a[r];
}
