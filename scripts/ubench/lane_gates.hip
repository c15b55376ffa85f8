// How much do the dispatch merges cost, and would gates on LANE bits avoid them?  32 complex amplitudes per lane in
// registers, G rx-like gates read from a descriptor, three formulations:
//   A  register-bit gate behind `switch (j)` (the tile-VM's G1M dispatch): 4 FMA-class ops per amplitude + the
//      whole-array register copies the compiler inserts at the merges;
//   B  lane-bit gate with a runtime xor mask: partner = __shfl_xor(a, mask) (ds_bpermute), new = c*self - i s*partner,
//      the same code for every gate: no switch, no merges;
//   C  lane-bit gate with a compile-time DPP pattern (xor 1 / 2 / 8) behind a switch (3 arms).
// Build: hipcc --offload-arch=gfx950 -O3 -o lane_gates lane_gates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define NR 32
typedef const int __attribute__((address_space(4)))* KInt;
typedef const float __attribute__((address_space(4)))* KF;

template <int J> __device__ __forceinline__ void rx_reg(float2 (&a)[NR], float c, float s) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if ((r >> J) & 1) continue;
    const int r1 = r | (1 << J);
    const float2 x = a[r], y = a[r1];
    a[r].x = __builtin_fmaf(s, y.y, c * x.x);
    a[r].y = __builtin_fmaf(-s, y.x, c * x.y);
    a[r1].x = __builtin_fmaf(s, x.y, c * y.x);
    a[r1].y = __builtin_fmaf(-s, x.x, c * y.y);
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float2* __restrict__ st, const int* __restrict__ dg, const float* __restrict__ tg, int ngates) {
  const KInt desc = (KInt)dg;
  const KF tab = (KF)tg;
  float2 a[NR];
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * NR;
#pragma unroll
  for (int r = 0; r < NR; ++r) a[r] = st[base + r];
#pragma unroll 1
  for (int g = 0; g < ngates; ++g) {
    const int j = desc[g];
    const float c = tab[2 * g], s = tab[2 * g + 1];
    if (MODE == 0) {
      switch (j % 5) {
        case 0: rx_reg<0>(a, c, s); break;
        case 1: rx_reg<1>(a, c, s); break;
        case 2: rx_reg<2>(a, c, s); break;
        case 3: rx_reg<3>(a, c, s); break;
        default: rx_reg<4>(a, c, s); break;
      }
    } else if (MODE == 1) {
      const int mask = 1 << (j % 6);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float pr = __shfl_xor(a[r].x, mask), pi = __shfl_xor(a[r].y, mask);
        const float2 v = a[r];
        a[r].x = __builtin_fmaf(s, pi, c * v.x);
        a[r].y = __builtin_fmaf(-s, pr, c * v.y);
      }
    } else {
      switch (j % 3) {
#define DPPGATE(CTRL)                                                                                      \
  _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                                         \
    const float pr = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[r].x), CTRL, 0xf, 0xf, false)); \
    const float pi = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a[r].y), CTRL, 0xf, 0xf, false)); \
    const float2 v = a[r];                                                                                 \
    a[r].x = __builtin_fmaf(s, pi, c * v.x);                                                               \
    a[r].y = __builtin_fmaf(-s, pr, c * v.y);                                                              \
  }
        case 0: DPPGATE(0xB1) break;    // quad_perm [1,0,3,2]: lane ^ 1
        case 1: DPPGATE(0x4E) break;    // quad_perm [2,3,0,1]: lane ^ 2
        default: DPPGATE(0x128) break;  // row_ror:8: lane ^ 8
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) st[base + r] = a[r];
}

int main() {
  const int nwg = 256 * 2 * 8, ngates = 96;
  const size_t nel = (size_t)nwg * 256 * NR;
  float2* st;
  hipMalloc(&st, nel * sizeof(float2));
  hipMemset(st, 0, nel * sizeof(float2));
  std::vector<int> d(ngates);
  std::vector<float> t(2 * ngates);
  for (int g = 0; g < ngates; ++g) { d[g] = (g * 7 + 3) % 30; t[2 * g] = 0.8f; t[2 * g + 1] = 0.6f; }
  int* dd; float* dt;
  hipMalloc(&dd, ngates * 4); hipMalloc(&dt, 2 * ngates * 4);
  hipMemcpy(dd, d.data(), ngates * 4, hipMemcpyHostToDevice);
  hipMemcpy(dt, t.data(), 2 * ngates * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"A register-bit gates, switch(j)", "B lane-bit gates, runtime shfl_xor", "C lane-bit gates, DPP behind switch"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, st, dd, dt, ngates);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, st, dd, dt, ngates);
      else hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, st, dd, dt, ngates);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)nwg * 4, simds = 256.0 * 4;
    const double cyc = ms * 1e-3 * 2.4e9 / (waves / simds / 2.0) / ngates / 2.0;   // per wave-gate at 2 waves per SIMD
    printf("%-40s %.3f ms  %.2f T amp-gates/s  ~%.0f SIMD cycles per wave-gate (32 amps/lane)\n", names[mode], ms,
           (double)nel * ngates / (ms * 1e-3) / 1e12, cyc);
  }
  return 0;
}
