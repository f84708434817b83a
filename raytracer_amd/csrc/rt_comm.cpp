// rt_comm.cpp — RCCL, bound at run time so that the library has no link dependency on it: only a caller
// that shards over several GPUs from C/C++ (no torch.distributed around) ever loads it.  Plain C++ (host
// API types only), built by g++.

#include "rt_comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>

#include "rt_hip.h"

namespace rtcomm {

static_assert(sizeof(ncclUniqueId) == kUniqueIdBytes, "rt_comm_unique_id hands out ncclUniqueId bytes");

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*errorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (lib == nullptr) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (lib != nullptr) {
            api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
            api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(lib, "ncclCommInitRank"));
            api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(lib, "ncclCommDestroy"));
            api.reduce = reinterpret_cast<decltype(api.reduce)>(dlsym(lib, "ncclReduce"));
            api.allReduce = reinterpret_cast<decltype(api.allReduce)>(dlsym(lib, "ncclAllReduce"));
            api.errorString = reinterpret_cast<decltype(api.errorString)>(dlsym(lib, "ncclGetErrorString"));
            if (api.getUniqueId && api.commInitRank && api.commDestroy && api.reduce && api.allReduce && api.errorString) api.lib = lib;
        }
    }
    return api.lib != nullptr ? &api : nullptr;
}

bool failed(Rccl *api, ncclResult_t r, const char *call, std::string &error) {
    if (r == ncclSuccess) return false;
    error = std::string(call) + ": " + api->errorString(r);
    return true;
}

}  // namespace

int uniqueId(void *id, size_t bytes) {
    if (id == nullptr || bytes != sizeof(ncclUniqueId)) return RT_ERR_INVALID;
    Rccl *api = rccl();
    if (api == nullptr) return RT_ERR_STATE;
    ncclUniqueId uid;
    if (api->getUniqueId(&uid) != ncclSuccess) return RT_ERR_HIP;
    std::memcpy(id, &uid, sizeof uid);
    return RT_OK;
}

int init(void **comm, int rank, int world, const void *id, size_t bytes, std::string &error) {
    if (comm == nullptr || id == nullptr || bytes != sizeof(ncclUniqueId)) return RT_ERR_INVALID;
    Rccl *api = rccl();
    if (api == nullptr) {
        error = "rt_comm_init: librccl.so.1 could not be loaded";
        return RT_ERR_STATE;
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t c = nullptr;
    if (failed(api, api->commInitRank(&c, world, uid, rank), "ncclCommInitRank", error)) return RT_ERR_HIP;
    *comm = c;
    return RT_OK;
}

void destroy(void *comm) {
    if (comm == nullptr) return;
    if (Rccl *api = rccl()) api->commDestroy(static_cast<ncclComm_t>(comm));
}

int reduceSum(void *comm, float *accum, size_t floats, unsigned long long *counter, int root, void *stream, std::string &error) {
    Rccl *api = rccl();
    if (api == nullptr || comm == nullptr) return RT_ERR_STATE;
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (failed(api, api->reduce(accum, accum, floats, ncclFloat, ncclSum, root, c, s), "ncclReduce", error)) return RT_ERR_HIP;
    if (failed(api, api->allReduce(counter, counter, 1, ncclUint64, ncclSum, c, s), "ncclAllReduce", error)) return RT_ERR_HIP;
    return RT_OK;
}

}  // namespace rtcomm
