// Data-parallel exchange step of the training path: ONE in-place SUM all-reduce of the flat CNN gradient over RCCL
// (xGMI on an 8-GPU MI355X node).  New capability: the reference is single device
// (/root/reference/karman-2d/karman_train.py:22,49 `--gpu` only sets CUDA_VISIBLE_DEVICES); the gradients of
// independent simulations add because the loss is a batch SUM (tf.nn.l2_loss, karman_train.py:430).
//
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy the host process already has, e.g. PyTorch's, is reused
// by soname), so libsol_hip.so itself has no link-time dependency on it and single-GPU users never load it.
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// ROCm install without the RCCL development headers: the five entry points this file binds by dlsym, declared as RCCL 2.x does
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclCommDestroy(ncclComm_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char* err = nullptr;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            q.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (q.h) break;
        }
        if (!q.h) { q.err = "librccl.so.1 not found (dlopen)"; return q; }
        q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(q.h, "ncclGetUniqueId"));
        q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(q.h, "ncclCommInitRank"));
        q.AllReduce = reinterpret_cast<decltype(q.AllReduce)>(dlsym(q.h, "ncclAllReduce"));
        q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.h, "ncclCommDestroy"));
        q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.h, "ncclGetErrorString"));
        if (!q.GetUniqueId || !q.CommInitRank || !q.AllReduce || !q.CommDestroy || !q.GetErrorString) q.err = "librccl.so.1 lacks an expected symbol";
        return q;
    }();
    return r;
}

#define SOL_RCCL_CHECK(expr)                                                                                        \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return sol_set_error(SOL_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

}  // namespace

struct sol_comm {
    ncclComm_t comm;
    int nranks, rank;
};

extern "C" int sol_comm_unique_id(char* id) {
    SOL_REQUIRE(id != nullptr, "sol_comm_unique_id: NULL id");
    static_assert(SOL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (rccl().err) return sol_set_error(SOL_ERR_HIP, "RCCL unavailable: %s", rccl().err);
    ncclUniqueId u;
    SOL_RCCL_CHECK(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return SOL_OK;
}

extern "C" int sol_comm_init(const char* id, int32_t nranks, int32_t rank, sol_comm** out) {
    SOL_REQUIRE(id && out && nranks >= 1 && rank >= 0 && rank < nranks, "sol_comm_init: bad arguments");
    if (rccl().err) return sol_set_error(SOL_ERR_HIP, "RCCL unavailable: %s", rccl().err);
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    int dev = -1, ndev = 0;
    SOL_HIP_CHECK(hipGetDeviceCount(&ndev));
    SOL_HIP_CHECK(hipGetDevice(&dev));
    SOL_REQUIRE(ndev >= 1 && dev >= 0 && dev < ndev, "sol_comm_init: no usable HIP device is current (device %d of %d)", dev, ndev);
    ncclComm_t c = nullptr;
    SOL_RCCL_CHECK(rccl().CommInitRank(&c, nranks, u, rank));      // binds to the calling thread's current HIP device
    *out = new sol_comm{c, nranks, rank};
    return SOL_OK;
}

extern "C" int sol_allreduce_grads(sol_comm* comm, void* stream, float* flat_grad, int64_t count) {
    SOL_REQUIRE(comm && comm->comm && flat_grad && count > 0, "sol_allreduce_grads: bad arguments");
    SOL_RCCL_CHECK(rccl().AllReduce(flat_grad, flat_grad, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream));
    return SOL_OK;
}

extern "C" int sol_comm_destroy(sol_comm* comm) {
    if (!comm) return SOL_OK;
    if (comm->comm && !rccl().err) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return SOL_OK;
}
