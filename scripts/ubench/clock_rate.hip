// What does one VALU instruction cost in CYCLES, and what is the shader clock under a VALU-bound load?
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz: their ratio over a long packed-FMA chain is the clock
// the chip actually sustains (the spec sheet's 2.4 GHz is a peak), and cycles / instructions the true issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 20000
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(unsigned long long* out, float a) {
  v2f p[8];
  float x[8];
  for (int i = 0; i < 8; ++i) { p[i].x = threadIdx.x * 0.001f + i; p[i].y = i * 0.5f; x[i] = i + threadIdx.x * 0.01f; }
  v2f c; c.x = a; c.y = 0.25f;
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(c));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(c));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + x[i];
  if (threadIdx.x == 0 && blockIdx.x < 64) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; }
  if (s == 12345.678f) out[200] = 1;
}
template <int MODE> void run(const char* name, int grid) {
  unsigned long long* d; hipMalloc(&d, 4096 * 8); hipMemset(d, 0, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(d, 1.0001f);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int i = 0; i < 64; ++i) { cyc += h[3 * i]; rt += h[3 * i + 1]; }
  cyc /= 64; rt /= 64;
  double insts_per_simd = (double)grid * 4 / 1024.0 * ITERS * 8;   // wave-instructions a SIMD issued (4 waves per block)
  printf("%-34s grid %5d  %.3f ms  kernel-wide: %.2f ns/inst/SIMD | per wave: %.0f shader-clk, %.0f x 10ns => clk %.2f GHz, %.2f clk/inst (x waves sharing the SIMD)\n",
         name, grid, ms, ms * 1e6 / insts_per_simd, cyc, rt, cyc / (rt * 10.0), cyc / (ITERS * 8.0));
  hipFree(d);
}
int main() {
  for (int g : {256, 1024, 4096}) {
    run<0>("v_pk_fma_f32 (sgpr coef)", g);
    run<1>("v_add_f32 (vgpr only)", g);
    run<2>("mix pk_fma + add", g);
  }
  return 0;
}
