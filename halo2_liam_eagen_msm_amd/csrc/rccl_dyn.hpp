// RCCL bound at run time (dlopen): the single-GPU entries must not depend on librccl being loadable, and in a
// process that already carries an RCCL (PyTorch bundles one under the same soname) the loader hands back that
// copy, so the process ends up with one RCCL, not two.  Only the multi-GPU entries (lemsm_comm_*, lemsm_*_sharded_*,
// lemsm_node_*) touch this; a missing library surfaces as LEMSM_ERR_RCCL with the loader's message.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include <string>

namespace lemsm {

struct Rccl {
  void* handle = nullptr;
  std::string error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;    // optional: absent -> CommDestroy
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }

  static Rccl& get() {
    static Rccl r = load();   // thread-safe one-time initialisation
    return r;
  }

 private:
  template <class Fn> bool sym(Fn& fn, const char* name) {
    fn = reinterpret_cast<Fn>(dlsym(handle, name));
    if (!fn) { error = std::string("librccl: missing symbol ") + name; return false; }
    return true;
  }
  static Rccl load() {
    Rccl r;
    const char* env = getenv("LEMSM_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      const char* e = dlerror();
      r.error = std::string("dlopen(") + nm + "): " + (e ? e : "?");
    }
    if (!r.handle) return r;
    if (!(r.sym(r.GetUniqueId, "ncclGetUniqueId") && r.sym(r.CommInitRank, "ncclCommInitRank") &&
          r.sym(r.CommDestroy, "ncclCommDestroy") && r.sym(r.AllGather, "ncclAllGather") &&
          r.sym(r.Broadcast, "ncclBroadcast") && r.sym(r.GetErrorString, "ncclGetErrorString"))) {
      dlclose(r.handle); r.handle = nullptr;
    } else {
      r.CommAbort = reinterpret_cast<ncclResult_t (*)(ncclComm_t)>(dlsym(r.handle, "ncclCommAbort"));
      r.error.clear();
    }
    return r;
  }
};

}  // namespace lemsm
