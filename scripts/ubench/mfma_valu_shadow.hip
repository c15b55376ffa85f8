// How many VALU instructions ride in the shadow of a v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
// Variants: V0 bare MFMAs; V1 + 4 independent v_fma_f32 on registers the MFMAs never touch; V2 + 4 VALU forming a dependent
// chain; V3 + 4 VALU whose destination registers are source registers of the MFMA issued just before (WAR);
// V4 the conversion mix (v_cvt_pk_bf16_f32, v_lshlrev, v_and, v_sub); V5 as V1 with 6 VALU; V6 as V1 with 2 VALU.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_shadow.hip -o mfma_valu_shadow && ./mfma_valu_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f32x4 a0 = {1.f, 2.f, 3.f, (float)threadIdx.x}, a1 = a0 * 2.f, b0 = a0 + 1.f, b1 = a0 - 1.f;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 * 3, x5 = x0 * 5;
  const float c = out[0];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define MF(I, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[I]) : "v"(A), "v"(B));
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(X) : "v"(c));
    if (V == 0) {
      MF(0, a0, b0) MF(1, a0, b1) MF(2, a1, b0) MF(3, a1, b1)
    } else if (V == 1) {
      MF(0, a0, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF(1, a0, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3)
      MF(2, a1, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF(3, a1, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3)
    } else if (V == 2) {
      MF(0, a0, b0) FMA(x0) FMA(x0) FMA(x0) FMA(x0) MF(1, a0, b1) FMA(x0) FMA(x0) FMA(x0) FMA(x0)
      MF(2, a1, b0) FMA(x0) FMA(x0) FMA(x0) FMA(x0) MF(3, a1, b1) FMA(x0) FMA(x0) FMA(x0) FMA(x0)
    } else if (V == 3) {
      // the VALU writes a source register of the MFMA just issued
#define WAR(R) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(R) : "v"(c));
      MF(0, a0, b0) WAR(a0.x) WAR(a0.y) WAR(b0.x) WAR(b0.y) MF(1, a0, b1) WAR(a0.z) WAR(a0.w) WAR(b1.x) WAR(b1.y)
      MF(2, a1, b0) WAR(a1.x) WAR(a1.y) WAR(b0.z) WAR(b0.w) MF(3, a1, b1) WAR(a1.z) WAR(a1.w) WAR(b1.z) WAR(b1.w)
    } else if (V == 4) {
#define CV(P, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(P) : "v"(X), "v"(Y));
#define SH(D, P) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(D) : "v"(P));
#define AN(D, P) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(D) : "v"(P));
#define SU(D, X, Y) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(Y));
      float p, q, r;
      MF(0, a0, b0) CV(p, x0, x1) SH(q, p) AN(r, p) SU(x2, x0, q) MF(1, a0, b1) SU(x3, x1, r) CV(p, x2, x3) SH(q, p) AN(r, p)
      MF(2, a1, b0) SU(x4, x2, q) SU(x5, x3, r) CV(p, x4, x5) SH(q, p) MF(3, a1, b1) AN(r, p) SU(x0, x4, q) SU(x1, x5, r) CV(p, x0, x1)
    } else if (V == 5) {
      MF(0, a0, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) MF(1, a0, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5)
      MF(2, a1, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) MF(3, a1, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5)
    } else if (V == 6) {
      MF(0, a0, b0) FMA(x0) FMA(x1) MF(1, a0, b1) FMA(x2) FMA(x3) MF(2, a1, b0) FMA(x0) FMA(x1) MF(3, a1, b1) FMA(x2) FMA(x3)
    } else if (V == 7) {   // V1 with the accumulators in VGPRs
#define MFV(I, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[I]) : "v"(A), "v"(B));
      MFV(0, a0, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MFV(1, a0, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3)
      MFV(2, a1, b0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MFV(3, a1, b1) FMA(x0) FMA(x1) FMA(x2) FMA(x3)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = x0 + x1 + x2 + x3 + x4 + x5 + a0.x + a1.y + b0.z + b1.w;
  for (int i = 0; i < 4; ++i) s += acc[i][3];
  out[1 + blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[V] = t1 - t0;
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (1 + 256 * 256) * sizeof(float));
  hipMemset(out, 0, (1 + 256 * 256) * sizeof(float));
  hipMallocManaged(&cyc, 8 * sizeof(unsigned long long));
  const int iters = 2000;
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  hipDeviceSynchronize();
  const char* names[8] = {"bare MFMAs", "+4 independent v_fma", "+4 dependent v_fma", "+4 VALU writing the MFMA's sources",
                          "+4 conversion-mix VALU", "+6 independent v_fma", "+2 independent v_fma", "+4 independent, acc in VGPRs"};
  for (int v = 0; v < 8; ++v) printf("%-40s %.1f cycles per MFMA\n", names[v], (double)cyc[v] / (4.0 * iters));
  return 0;
}
