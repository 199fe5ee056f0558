// exchange.hip -- the training step's one exchange behind the C ABI (straps_allreduce_grads, "RCCL comm handle passed in").
//
// The reference trains on one GPU (run_train.py:23-26); data parallel is this build's addition: replicated weights,
// one sum all-reduce of the flat fp32 gradient buffer per step, 1/world folded into straps_adam_step.  A torch host can issue that
// all-reduce through torch.distributed; a host WITHOUT torch (the C-ABI boundary's whole point) uses the four entry points below:
//
//     rank 0:  straps_comm_unique_id(id)           -> 128 opaque bytes, shipped to the other ranks by the host's own means
//     all:     straps_comm_init_rank(id, n, r, &c) -> communicator over RCCL (xGMI inside a node)
//     step:    straps_allreduce_grads(g, n, c, st) -> in-place fp32 sum all-reduce, enqueued on the HIP stream `st`
//     end:     straps_comm_destroy(c)
//
// RCCL is resolved at RUN time, not linked: the communicator a host passes in was created by the RCCL that host loaded (torch ships its own
// librccl.so next to libtorch_hip.so), and a handle must go back into the SAME library -- so the copy already in the process is preferred
// (dlopen RTLD_NOLOAD), and only a process without one loads the system's librccl.so.1.  The library itself therefore has no RCCL
// dependency: single-GPU users never touch it.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

// The slice of the NCCL / RCCL C ABI used here, declared locally: the library must build on hosts without the RCCL development headers
// (single-GPU users, cross-compile boxes -- ADVICE round 4), and RCCL stays a RUN-time dependency only.  These five types are the
// stable public ABI of nccl.h (NCCL 2.x and every RCCL release): an opaque communicator pointer, a 128-byte id passed BY VALUE, and three
// int-sized enums of which this file names four values.
struct ncclComm;
typedef ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat32 = 7;      // nccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7
constexpr ncclRedOp_t ncclSum = 0;

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char origin[96] = "";
};

RcclApi g_rccl;

// 0 on success; fills g_rccl once per process (thread-safe through the function-local static's initialisation)
int rccl_resolve() {
    static const int rc = [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        void* h = nullptr;
        for (const char* n : names) {          // the copy this process already carries (e.g. torch's)
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (h) { snprintf(g_rccl.origin, sizeof(g_rccl.origin), "%s (already loaded in the process)", n); break; }
        }
        if (!h) {
            const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : fresh) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (h) { snprintf(g_rccl.origin, sizeof(g_rccl.origin), "%s (loaded by libstraps_hip)", n); break; }
            }
        }
        if (!h) return 1;
        g_rccl.handle = h;
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
        g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        return (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommCount && g_rccl.AllReduce && g_rccl.GetErrorString) ? 0 : 2;
    }();
    return rc;
}

#define STRAPS_NEED_RCCL(who)                                                                                                    \
    do {                                                                                                                         \
        const int r__ = rccl_resolve();                                                                                          \
        if (r__ != 0) {                                                                                                          \
            straps_set_error("%s: %s", who, r__ == 1 ? "no RCCL library in this process and none could be loaded (librccl.so.1)" \
                                                     : "the RCCL library lacks a required nccl* symbol");                        \
            return STRAPS_EUNSUPPORTED;                                                                                          \
        }                                                                                                                        \
    } while (0)

#define STRAPS_CHECK_RCCL(call, who)                                                              \
    do {                                                                                          \
        const ncclResult_t n__ = (call);                                                          \
        if (n__ != ncclSuccess) {                                                                 \
            straps_set_error("%s: RCCL error %d: %s", who, (int)n__, g_rccl.GetErrorString(n__)); \
            return STRAPS_EHIP;                                                                   \
        }                                                                                         \
    } while (0)

}  // namespace

extern "C" int straps_comm_unique_id(void* id128) {
    STRAPS_REQUIRE(id128, "straps_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == STRAPS_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes");
    STRAPS_NEED_RCCL("straps_comm_unique_id");
    ncclUniqueId id;
    STRAPS_CHECK_RCCL(g_rccl.GetUniqueId(&id), "straps_comm_unique_id");
    memcpy(id128, &id, sizeof(id));
    return STRAPS_OK;
}

extern "C" int straps_comm_init_rank(const void* id128, int nranks, int rank, void** comm) {
    STRAPS_REQUIRE(id128 && comm, "straps_comm_init_rank: null pointer");
    STRAPS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "straps_comm_init_rank: rank %d outside [0, %d)", rank, nranks);
    STRAPS_NEED_RCCL("straps_comm_init_rank");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    STRAPS_CHECK_RCCL(g_rccl.CommInitRank(&c, nranks, id, rank), "straps_comm_init_rank");
    *comm = (void*)c;
    return STRAPS_OK;
}

extern "C" int straps_comm_destroy(void* comm) {
    if (!comm) return STRAPS_OK;
    STRAPS_NEED_RCCL("straps_comm_destroy");
    STRAPS_CHECK_RCCL(g_rccl.CommDestroy((ncclComm_t)comm), "straps_comm_destroy");
    return STRAPS_OK;
}

extern "C" int straps_comm_size(void* comm) {
    if (!comm || rccl_resolve() != 0) return 0;
    int n = 0;
    if (g_rccl.CommCount((ncclComm_t)comm, &n) != ncclSuccess) return 0;
    return n;
}

extern "C" const char* straps_comm_library(void) {
    return rccl_resolve() == 0 ? g_rccl.origin : "";
}

extern "C" int straps_allreduce_grads(float* flat_g, long long n, void* comm, void* stream) {
    STRAPS_REQUIRE(flat_g && comm, "straps_allreduce_grads: null pointer");
    STRAPS_REQUIRE(n >= 0, "straps_allreduce_grads: negative count %lld", n);
    if (n == 0) return STRAPS_OK;
    STRAPS_NEED_RCCL("straps_allreduce_grads");
    STRAPS_CHECK_RCCL(g_rccl.AllReduce(flat_g, flat_g, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream),
                      "straps_allreduce_grads");
    return STRAPS_OK;
}
