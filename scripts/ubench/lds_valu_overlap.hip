// Do LDS traffic and packed-FMA issue overlap on a gfx950 CU?  256-thread workgroups, 4 per CU (4 waves per SIMD), 32 KiB
// of LDS each -- the shape of the tile-VM passes.  Per iteration a wave runs NV packed FMAs (SGPR coefficient) and NL
// planar LDS round trips (ds_write_b32 + ds_read_b32 of 4 bytes per lane, conflict free), with or without barriers.
//   mode 0: VALU only    mode 1: LDS only    mode 2: both in every wave    mode 3: both + barriers (exchange-like)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 400
template <int MODE, int NV, int NL>
__global__ __launch_bounds__(256, 4) void k(float* out, float a) {
  extern __shared__ float lds[];
  v2f p[16];
  float x[32];
  for (int i = 0; i < 16; ++i) { p[i].x = threadIdx.x * 0.001f + i; p[i].y = i * 0.5f; }
  for (int i = 0; i < 32; ++i) x[i] = i + threadIdx.x * 0.01f;
  v2f c; c.x = a; c.y = 0.25f;
  const int t = threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int r = 0; r < NV / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(c));
    }
    if (MODE != 0) {
#pragma unroll
      for (int r = 0; r < NL / 32; ++r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) lds[i * 256 + t] = x[i];
        if (MODE == 3) __syncthreads();
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = ((volatile float*)lds)[i * 256 + (t ^ 1)];
        if (MODE == 3) __syncthreads();
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
  for (int i = 0; i < 32; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NV, int NL> float run(const char* name) {
  float* d; (void)hipMalloc(&d, 256 * 8192 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, NV, NL><<<8192, 256, 32768>>>(d, 1.0001f);
  (void)hipEventRecord(e0);
  k<MODE, NV, NL><<<8192, 256, 32768>>>(d, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s NV=%4d NL=%3d  %.3f ms\n", name, NV, NL, ms);
  (void)hipFree(d);
  return ms;
}
int main() {
  float a = run<0, 512, 64>("VALU only");
  float b = run<1, 512, 64>("LDS only (64 writes + 64 reads per iter)");
  float c = run<2, 512, 64>("both, same waves, no barrier");
  float d = run<3, 512, 64>("both + 4 barriers per iter (exchange-like)");
  printf("sum %.3f  max %.3f  both %.3f  both+barriers %.3f\n", a + b, a > b ? a : b, c, d);
  a = run<0, 256, 64>("VALU only");
  c = run<2, 256, 64>("both, same waves, no barrier");
  d = run<3, 256, 64>("both + 4 barriers per iter");
  printf("sum %.3f  both %.3f  both+barriers %.3f\n", a + b, c, d);
  return 0;
}
