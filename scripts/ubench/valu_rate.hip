// VALU issue-rate microbenchmark for gfx950: plain v_fma_f32 vs v_pk_fma_f32 vs v_sin/v_cos vs f64.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
#define N_ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x[16];
  float2v p[8];
  double dd[8];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) { p[i].x = threadIdx.x * 0.001f + i; p[i].y = i * 0.5f; dd[i] = i + threadIdx.x; }
  float2v av = {a, a}, bv = {b, b};
  for (int it = 0; it < N_ITERS; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], av, bv);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_sinf(x[i]);
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dd[i] = dd[i] + (double)a;
    } else if (MODE == 4) {   // fma with two SGPR-ish operands distinct regs: x = x*y + z pattern
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], x[(i + 1) & 15], x[(i + 2) & 15]);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)dd[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double ops_per_iter_per_lane) {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<2048, 256>>>(d, 1.0001f, 0.5f);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<2048, 256>>>(d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double lane_ops = 2048.0 * 256 * N_ITERS * ops_per_iter_per_lane;
  // wave-instructions per second per SIMD
  double winst = 2048.0 * 4 * N_ITERS * (MODE == 1 || MODE == 3 ? 8 : 16);
  double cyc_per_inst = (ms * 1e-3 * 2.4e9) / (winst / 1024.0);
  printf("%-28s %.3f ms  %.1f Gop/s lane-ops  ~%.2f cyc/wave-inst/SIMD (at 2.4GHz)\n", name, ms, lane_ops / ms / 1e6, cyc_per_inst);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32 (imm operands)", 16);
  run<4>("v_fma_f32 (3 vgpr)", 16);
  run<1>("v_pk_fma_f32", 16);
  run<2>("v_sin_f32", 16);
  run<3>("v_add_f64", 8);
  return 0;
}
