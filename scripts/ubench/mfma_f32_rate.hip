// f32 MFMA issue rate on MI355X by occupancy and instruction mix: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
// variants: 0 = 3 accumulators of 32x32x2 only; 1 = + 2 v_add per 3 MFMAs (the Gauss sums of the complex GEMM);
//           2 = 16x16x4 (6 accumulators); 3 = 32x32x2 with 6 accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 p1 = {0}, p2 = {0}, p3 = {0}, p4 = {0}, p5 = {0}, p6 = {0};
  f32x4 q[6] = {};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x, c = a0, d = b0;
  for (int i = 0; i < iters; ++i) {
    if constexpr (V == 0) {
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, p3, 0, 0, 0);
    } else if constexpr (V == 1) {
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, p2, 0, 0, 0);
      float s = a + c, t = b + d;
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, t, p3, 0, 0, 0);
      a += 1.f; c = t;
    } else if constexpr (V == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) q[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q[j], 0, 0, 0);
    } else {
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, d, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d, p3, 0, 0, 0);
      p4 = __builtin_amdgcn_mfma_f32_32x32x2f32(c, b, p4, 0, 0, 0);
      p5 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, p5, 0, 0, 0);
      p6 = __builtin_amdgcn_mfma_f32_32x32x2f32(d, d, p6, 0, 0, 0);
    }
  }
  float r = 0;
  for (int j = 0; j < 16; ++j) r += p1[j] + p2[j] + p3[j] + p4[j] + p5[j] + p6[j];
  for (int j = 0; j < 6; ++j) r += q[j][0];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V>
void run(int wg_per_cu, int lds_pad) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // dynamic LDS limits the occupancy to wg_per_cu workgroups (4 waves = 1 per SIMD each)
  size_t lds = 160 * 1024 / wg_per_cu - 512; if (lds > 64 * 1024) { hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, out, 100, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_iter = V == 2 ? 6 : (V == 3 ? 6 : 3);
  const double flops_per = V == 2 ? 16 * 16 * 4 * 2.0 : 32 * 32 * 2 * 2.0;
  const double nm = (double)grid * 4 * iters * per_iter;
  // cycles per MFMA per SIMD at 2.4 GHz: time * 2.4e9 / (MFMAs per SIMD)
  printf("variant %d  %d waves/SIMD: %7.3f ms  %6.1f TF  %5.1f cycles(2.4GHz)/MFMA/SIMD\n", V, wg_per_cu, ms,
         nm * flops_per / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm / 1024));
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 3, 4}) run<0>(w, 0);
  for (int w : {1, 2, 3, 4}) run<1>(w, 0);
  for (int w : {1, 2, 3}) run<2>(w, 0);
  for (int w : {1, 2, 3}) run<3>(w, 0);
  return 0;
}
