// RCCL bound at run time (dlopen/dlsym): libmfcd_hip.so has no link-time dependency on RCCL, so it loads on hosts
// without it, and inside a PyTorch process it binds to the very RCCL copy torch has already loaded (same soname)
// instead of bringing a second one into the process.  Only the types of <rccl/rccl.h> are used.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace mfcd_detail {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

inline const RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {   // first a copy that is already in the process, then a fresh load
            a.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (a.handle) break;
        }
        for (int k = 0; !a.handle && k < 4; ++k) a.handle = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather;
        return a;
    }();
    return api;
}

}  // namespace mfcd_detail
