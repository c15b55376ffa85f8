// What takes the join GEMM's inner loop from 65 to 80 cycles per f32 MFMA?  The K step of cgemm_dma_kernel rebuilt
// piece by piece: hipcc --offload-arch=gfx950 -O3 mfma_gemm_steps.hip -o mfma_gemm_steps
//   V0 24 MFMAs + 16 v_add per step            V1 + 16 ds_read_b64 (inline asm, all up front)
//   V2 + s_barrier per step                    V3 + 4 global_load_lds per wave and step (L2-resident source)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int V, int S>
__global__ __launch_bounds__(256, 3) void k(float* out, const float4* src, int iters) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 p1 = {0}, p2 = {0}, p3 = {0};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)dsm;
  const uint32_t fa = lds0 + (lane >> 5) * 512 + ((wave >> 1) * 32 + (lane & 31)) * 8;
  const uint32_t fb = lds0 + 8192 + (lane >> 5) * 512 + ((wave & 1) * 32 + (lane & 31)) * 8;
  for (int i = threadIdx.x; i < S * 4096; i += 256) reinterpret_cast<float*>(dsm)[i] = 1e-3f * (i & 255);
  __syncthreads();
  const float4* g = src + (blockIdx.x % 64) * 4096 + threadIdx.x;
  int st = 0;
  for (int i = 0; i < iters; ++i) {
    if (V >= 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (S - 2)) : "memory"); }
    if (V >= 2) __builtin_amdgcn_s_barrier();
    if (V >= 3) {
      char* sb_ = dsm + ((st + S - 1) % S) * 16384 + wave * 2048;
      const float4* gp = g + (i & 7) * 256;
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)sb_, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + 64), (lptr_t)(sb_ + 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + 2048), (lptr_t)(sb_ + 8192), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + 2112), (lptr_t)(sb_ + 8192 + 1024), 16, 0, 0);
    }
    v2f a[8], b[8];
    if (V >= 1) {
      const uint32_t sa = fa + st * 16384, sb = fb + st * 16384;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[q]) : "v"(sa), "n"(q * 1024));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b[q]) : "v"(sb), "n"(q * 1024));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[q].x = p1[q] * 1e-9f; a[q].y = 1.f + q; b[q].x = 2.f + lane; b[q].y = p2[q] * 1e-9f; }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (V >= 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[q]), "+v"(b[q]) : "n"(14 - 2 * q));
      p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x + a[q].y, b[q].x + b[q].y, p3, 0, 0, 0);
    }
    st = (st + 1 == S) ? 0 : st + 1;
  }
  float r = 0;
  for (int j = 0; j < 16; ++j) r += p1[j] + p2[j] + p3[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V, int S>
void run(int wgs) {
  float* out; hipMalloc(&out, 256 * 16 * 256 * 4);
  float4* src; hipMalloc(&src, 64 * 4096 * 16 + 65536); hipMemset(src, 0, 64 * 4096 * 16 + 65536);
  const int iters = 2000, grid = 256 * wgs;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  size_t lds = S * 16384;
  if (wgs * lds + wgs * 1024 > 160 * 1024) { printf("skip\n"); return; }
  // pad the LDS request so that exactly `wgs` workgroups fit a CU
  size_t want = 160 * 1024 / wgs - 1024; if (want > lds) lds = want;
  hipFuncSetAttribute((const void*)k<V, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<V, S>), dim3(grid), dim3(256), lds, 0, out, src, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, S>), dim3(grid), dim3(256), lds, 0, out, src, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)grid * 4 * iters * 24;
  printf("V%d stages %d, %d waves/SIMD: %7.3f ms  %6.1f TF  %5.1f cycles(2.4GHz)/MFMA/SIMD\n", V, S, wgs, ms,
         nm * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm / 1024));
  hipFree(out); hipFree(src);
}
int main() {
  for (int w : {1, 3}) run<0, 3>(w);
  for (int w : {1, 3}) run<1, 3>(w);
  for (int w : {1, 3}) run<2, 3>(w);
  for (int w : {1, 2, 3}) run<3, 3>(w);
  for (int w : {3, 5}) run<3, 2>(w);
  return 0;
}
