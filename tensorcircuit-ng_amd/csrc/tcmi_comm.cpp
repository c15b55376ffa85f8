// RCCL collective of the C ABI: tcmi_allreduce_sum and the communicator it runs on (SURVEY.md section 8(b) lists the
// entry; a host that is NOT torch -- the reference's own JAX / numpy processes bound through ctypes -- has no process
// group to borrow).  What it replaces on the path: the sum over devices of the per-slice / per-sample partial
// [value || gradients] (reference tensorcircuit/experimental.py:1145-1152 jnp.sum(device_values, axis=0) after the pmap,
// examples/slicing_auto_pmap_vqa.py:60-72), one small packed all-reduce per step, latency-bound over xGMI.
//
// RCCL is not linked: librccl is opened on first use (the library of the host framework when it already has one loaded --
// dlopen of a loaded soname returns it -- else the path given to tcmi_comm_load, else the ROCm installation's), so
// libtcmi.so loads on hosts without it and one-rank users never touch it.  Host code only; no kernels.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and enums only
#include <stdio.h>
#include <string.h>

#include "../../include/tcmi.h"

extern "C" int tcmi_set_error_(int code, const char* msg);

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g;

int load(const char* path) {
  if (g.lib) return TCMI_OK;
  const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  char tried[400] = "";
  for (const char* c : cands) {
    if (!c || !*c) continue;
    void* h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      strncat(tried, c, sizeof(tried) - strlen(tried) - 2);
      strncat(tried, " ", sizeof(tried) - strlen(tried) - 1);
      continue;
    }
    Rccl r;
    r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GetErrorString) {
      g = r;
      return TCMI_OK;
    }
    dlclose(h);
  }
  char buf[512];
  snprintf(buf, sizeof buf, "tcmi_comm: no usable librccl (tried: %s)", tried);
  return tcmi_set_error_(TCMI_ERR_HIP, buf);
}

int fail(const char* what, ncclResult_t r) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "rccl error");
  return tcmi_set_error_(TCMI_ERR_HIP, buf);
}
}  // namespace

extern "C" {

int tcmi_comm_load(const char* librccl_path_host) { return load(librccl_path_host); }

int tcmi_comm_unique_id(void* id_out_host) {
  if (!id_out_host) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_comm_unique_id: null output");
  int rc = load(nullptr);
  if (rc != TCMI_OK) return rc;
  ncclUniqueId id;
  ncclResult_t r = g.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  memcpy(id_out_host, &id, TCMI_COMM_ID_BYTES);
  return TCMI_OK;
}

int tcmi_comm_init(const void* id_host, int rank, int world, void** comm_out_host) {
  if (!id_host || !comm_out_host || world < 1 || rank < 0 || rank >= world)
    return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_comm_init: bad argument");
  *comm_out_host = nullptr;
  int rc = load(nullptr);
  if (rc != TCMI_OK) return rc;
  ncclUniqueId id;
  memcpy(&id, id_host, TCMI_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  ncclResult_t r = g.CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail("ncclCommInitRank", r);
  *comm_out_host = comm;
  return TCMI_OK;
}

int tcmi_allreduce_sum(void* comm, void* buf, long long count, int dtype, void* stream) {
  if (!comm || !buf || count < 0 || !g.AllReduce) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_allreduce_sum: bad argument");
  if (count == 0) return TCMI_OK;
  ncclDataType_t dt;
  size_t n = (size_t)count;
  switch (dtype) {   // complex sums are sums of their (re, im) floats
    case TCMI_F32: dt = ncclFloat32; break;
    case TCMI_F64: dt = ncclFloat64; break;
    case TCMI_C64: dt = ncclFloat32; n *= 2; break;
    case TCMI_C128: dt = ncclFloat64; n *= 2; break;
    default: return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_allreduce_sum: bad dtype");
  }
  ncclResult_t r = g.AllReduce(buf, buf, n, dt, ncclSum, reinterpret_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream));
  if (r != ncclSuccess) return fail("ncclAllReduce", r);
  return TCMI_OK;
}

int tcmi_comm_destroy(void* comm) {
  if (!comm) return TCMI_OK;
  if (!g.CommDestroy) return tcmi_set_error_(TCMI_ERR_ARG, "tcmi_comm_destroy: no communicator was ever created");
  ncclResult_t r = g.CommDestroy(reinterpret_cast<ncclComm_t>(comm));
  return r == ncclSuccess ? TCMI_OK : fail("ncclCommDestroy", r);
}

}  // extern "C"
