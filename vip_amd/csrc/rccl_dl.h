// rccl_dl.h -- RCCL entry points resolved at run time (dlopen): libvipmi.so carries no link-time dependency on RCCL, and a
// process that already holds an RCCL (PyTorch's) shares that copy -- a communicator must come from the same library that
// executes the collectives on it.  Only the types of <rccl/rccl.h> are used at build time.
#pragma once
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace vipmi {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }
};

inline RcclApi& rccl_api() {
  static RcclApi api;
  return api;
}

// path == nullptr: the RCCL already in the process, else librccl.so.1 / librccl.so from the loader's search path
inline bool rccl_load(const char* path, const char** why) {
  static std::mutex mu;                      // first use may come from several host threads (one ctx per thread)
  std::lock_guard<std::mutex> lock(mu);
  RcclApi& a = rccl_api();
  if (a.ok()) return true;
  void* h = nullptr;
  if (path && *path) {
    h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  } else {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    *why = dlerror();
    return false;
  }
  RcclApi t;
  t.handle = h;
#define VIPMI_RCCL_SYM(field, name)                                   \
  t.field = reinterpret_cast<decltype(t.field)>(dlsym(h, name));      \
  if (!t.field) {                                                     \
    *why = "missing RCCL symbol " name;                               \
    return false;                                                     \
  }
  VIPMI_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  VIPMI_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  VIPMI_RCCL_SYM(CommDestroy, "ncclCommDestroy");
  VIPMI_RCCL_SYM(AllReduce, "ncclAllReduce");
  VIPMI_RCCL_SYM(Send, "ncclSend");
  VIPMI_RCCL_SYM(Recv, "ncclRecv");
  VIPMI_RCCL_SYM(GroupStart, "ncclGroupStart");
  VIPMI_RCCL_SYM(GroupEnd, "ncclGroupEnd");
  VIPMI_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef VIPMI_RCCL_SYM
  a = t;
  return true;
}

}  // namespace vipmi
