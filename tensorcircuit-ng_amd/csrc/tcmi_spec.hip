// Plan-specialised pass kernels: loader and launchers (host code; the kernels themselves are generated per plan by
// tcmi/specialize.py and compiled to gfx950 code objects -- see the header of that file).
//
// What this stands in for: the reference jits the circuit function once per structure and replays the compiled
// executable for every parameter value (backend.jit -> jax.jit, tensorcircuit/backends/jax_backend.py; used by the
// harness around the VQE step, benchmarks/scripts/vqe_tc.py:136-141).  Here the compiled artefact is one code object per
// pass; a handle is the hipFunction_t of its kernel.  Arguments mirror tcmi_run_pass / tcmi_run_adjoint_pass minus the
// descriptor (it is baked into the code).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "tcmi_vm.h"

extern "C" int tcmi_set_error_(int code, const char* msg);

namespace {
struct SpecKernel {
  hipModule_t mod;
  hipFunction_t fn;
  int lds_bytes;
  int flags;            // TCMI_SPEC_FLAG_SRC: generated with the "src" option (argument buffer of tcmi_spec_run_pass_from)
};
// live_mask (over the n - T bits of the tile index): 0xffffffff = one workgroup per tile; else one workgroup per tile whose
// index is zero outside the mask (the kernel spreads blockIdx.x over the mask's bits)
unsigned grid_x(int n, int T, unsigned live_mask) {
  unsigned tiles = 1u << (n - T);
  if (live_mask != 0xffffffffu) tiles = 1u << __builtin_popcount(live_mask & (tiles - 1u));
  return tiles;
}
int hip_fail(const char* what, hipError_t e) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  return tcmi_set_error_(TCMI_ERR_HIP, buf);
}
}  // namespace

extern "C" {

int tcmi_spec_load(const char* path_host, const char* kernel_name_host, int lds_bytes, void** handle_out_host) {
  if (!path_host || !kernel_name_host || !handle_out_host || lds_bytes < 0)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_load: bad argument");
  *handle_out_host = nullptr;
  SpecKernel* k = new SpecKernel{nullptr, nullptr, lds_bytes, 0};
  hipError_t e = hipModuleLoad(&k->mod, path_host);
  if (e != hipSuccess) {
    delete k;
    return hip_fail("tcmi_spec_load: hipModuleLoad", e);
  }
  e = hipModuleGetFunction(&k->fn, k->mod, kernel_name_host);
  if (e != hipSuccess) {
    hipModuleUnload(k->mod);
    delete k;
    return hip_fail("tcmi_spec_load: hipModuleGetFunction", e);
  }
  if (lds_bytes > 48 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) (void)hipGetLastError();  // module functions take the size from the launch on ROCm
  }
  *handle_out_host = k;
  return TCMI_OK;
}

int tcmi_spec_set_flags(void* handle, int flags) {
  if (!handle || (flags & ~TCMI_SPEC_FLAG_SRC)) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_set_flags: bad argument");
  reinterpret_cast<SpecKernel*>(handle)->flags = flags;
  return TCMI_OK;
}

int tcmi_spec_unload(void* handle) {
  if (!handle) return TCMI_OK;
  SpecKernel* k = reinterpret_cast<SpecKernel*>(handle);
  hipError_t e = hipModuleUnload(k->mod);
  delete k;
  return e == hipSuccess ? TCMI_OK : hip_fail("tcmi_spec_unload", e);
}

int tcmi_spec_run_pass(void* handle, void* state, long long state_stride, int batch, int n, int T, int LT,
                       const void* ctab, const void* ptab, long long ptab_stride, unsigned live_mask, unsigned zero_bits,
                       void* stream) {
  if (!handle || !state || batch < 1 || n < T || T <= LT || LT < 6 || LT > 10)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_run_pass: bad argument");
  SpecKernel* k = reinterpret_cast<SpecKernel*>(handle);
  if (k->flags & TCMI_SPEC_FLAG_SRC)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_run_pass: the kernel reads its input from another batch (tcmi_spec_run_pass_from)");
  struct {
    void* state;
    long long state_stride;
    const void* ctab;
    const void* ptab;
    long long ptab_stride;
    unsigned live_mask;
    unsigned zero_bits;
  } args = {state, state_stride, ctab, ptab, ptab_stride, live_mask, zero_bits};
  size_t sz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(k->fn, grid_x(n, T, live_mask), (unsigned)batch, 1, 1u << LT, 1, 1, (unsigned)k->lds_bytes,
                                       reinterpret_cast<hipStream_t>(stream), nullptr, cfg);
  return e == hipSuccess ? TCMI_OK : hip_fail("tcmi_spec_run_pass", e);
}

int tcmi_spec_run_pass_from(void* handle, void* state, long long state_stride, int batch, int n, int T, int LT,
                            const void* ctab, const void* ptab, long long ptab_stride, const void* src,
                            long long src_stride, int src_shift, const void* scale, void* stream) {
  if (!handle || !state || !src || batch < 1 || n < T || T <= LT || LT < 6 || LT > 10 || src_shift < 0 || src_shift > 30)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_run_pass_from: bad argument");
  SpecKernel* k = reinterpret_cast<SpecKernel*>(handle);
  if (!(k->flags & TCMI_SPEC_FLAG_SRC))   // a kernel without the "src" arguments would be launched with a mismatched buffer
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_run_pass_from: the kernel was not generated with the src option (tcmi_spec_set_flags)");
  struct {
    void* state;
    long long state_stride;
    const void* ctab;
    const void* ptab;
    long long ptab_stride;
    unsigned live_mask;
    unsigned zero_bits;
    const void* src;
    long long src_stride;
    unsigned src_shift;
    unsigned has_scale;
    const void* scale;
  } args = {state, state_stride, ctab, ptab, ptab_stride, 0xffffffffu, 0u, src, src_stride, (unsigned)src_shift,
            scale ? 1u : 0u, scale ? scale : src};
  size_t sz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(k->fn, grid_x(n, T, 0xffffffffu), (unsigned)batch, 1, 1u << LT, 1, 1,
                                       (unsigned)k->lds_bytes, reinterpret_cast<hipStream_t>(stream), nullptr, cfg);
  return e == hipSuccess ? TCMI_OK : hip_fail("tcmi_spec_run_pass_from", e);
}

int tcmi_spec_run_adjoint_pass(void* handle, void* psi, void* lam, long long state_stride, int batch, int n, int T, int LT,
                               const void* ctab, const void* ptab, long long ptab_stride, double* gout,
                               long long gout_stride, int gcopies, long long gcopy_stride, unsigned live_mask,
                               void* stream) {
  if (!handle || !psi || !lam || !gout || batch < 1 || n < T || T <= LT || LT < 6 || LT > 10 || gcopies < 1)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_spec_run_adjoint_pass: bad argument");
  SpecKernel* k = reinterpret_cast<SpecKernel*>(handle);
  struct {
    void* psi;
    void* lam;
    long long state_stride;
    const void* ctab;
    const void* ptab;
    long long ptab_stride;
    double* gout;
    long long gout_stride;
    int gcopies;
    unsigned live_mask;
    long long gcopy_stride;
  } args = {psi, lam, state_stride, ctab, ptab, ptab_stride, gout, gout_stride, gcopies, live_mask, gcopy_stride};
  size_t sz = sizeof(args);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(k->fn, grid_x(n, T, live_mask), (unsigned)batch, 1, 1u << LT, 1, 1, (unsigned)k->lds_bytes,
                                       reinterpret_cast<hipStream_t>(stream), nullptr, cfg);
  return e == hipSuccess ? TCMI_OK : hip_fail("tcmi_spec_run_adjoint_pass", e);
}

}  // extern "C"
