// The one exchange step of the partitioned scan-to-map registration behind the C ABI: an RCCL
// communicator (one rank per GPU) and the sum all-reduce of the 8x8 float32 least-squares tile
// (256 bytes) on the grid's stream, between pcp_grid_ls_dev and pcp_grid_pose_dev.
//
// RCCL is bound at run time (dlopen) so that libpcp.so loads on machines and in processes that never
// use more than one GPU, and so that a process which already holds an RCCL (PyTorch bundles one) shares
// it instead of loading a second copy.  Search order: $PCP_RCCL_PATH, a copy already in the process,
// librccl.so.1 / librccl.so on the loader path, /opt/rocm/lib/librccl.so.1.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/pcp.h"
#include "pcp_host.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int bind_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    if (const char* path = getenv("PCP_RCCL_PATH")) {
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return pcp_fail("PCP_RCCL_PATH=%s: %s", path, dlerror());
    }
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // a copy the process already holds
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return pcp_fail("RCCL not found (set PCP_RCCL_PATH): %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
        return pcp_fail("the RCCL library found lacks a symbol this library needs");
    g_rccl = r;
    return 0;
}

#define PCP_NCCL(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t pcp_r_ = (expr);                                                                    \
        if (pcp_r_ != ncclSuccess) return pcp_fail("%s failed: %s", #expr, g_rccl.GetErrorString(pcp_r_)); \
    } while (0)

}  // namespace

struct pcp_comm {
    int device = 0, world = 1, rank = 0;
    ncclComm_t comm = nullptr;
};

extern "C" {

int pcp_comm_unique_id(char* id128) {
    if (!id128) return pcp_fail("null id buffer");
    if (int e = bind_rccl()) return e;
    static_assert(NCCL_UNIQUE_ID_BYTES == PCP_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    PCP_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int pcp_comm_init(int device, int world, int rank, const char* id128, pcp_comm** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    if (!id128) return pcp_fail("null id");
    if (world < 1 || rank < 0 || rank >= world) return pcp_fail("rank %d outside a world of %d", rank, world);
    if (int e = bind_rccl()) return e;
    int ndev = 0;
    PCP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return pcp_fail("device %d not present (%d visible)", device, ndev);
    PCP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    pcp_comm* c = new pcp_comm();
    c->device = device; c->world = world; c->rank = rank;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return pcp_fail("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, g_rccl.GetErrorString(r));
    }
    *out = c;
    return 0;
}

int pcp_comm_destroy(pcp_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

int pcp_comm_abort(pcp_comm* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)(g_rccl.CommAbort ? g_rccl.CommAbort(c->comm) : g_rccl.CommDestroy(c->comm));
    delete c;
    return 0;
}

int pcp_comm_info(pcp_comm* c, int* world, int* rank) {
    if (!c) return pcp_fail("null communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return 0;
}

int pcp_comm_allreduce_tile(pcp_comm* c, float* tile_dev, void* hip_stream) {
    if (!c) return pcp_fail("null communicator");
    if (!tile_dev) return pcp_fail("null tile");
    PCP_CHECK(hipSetDevice(c->device));
    PCP_NCCL(g_rccl.AllReduce(tile_dev, tile_dev, 64, ncclFloat32, ncclSum, c->comm, (hipStream_t)hip_stream));
    return 0;
}

}  // extern "C"
