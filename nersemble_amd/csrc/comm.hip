// comm.hip -- collectives issued by the library itself: an RCCL communicator owned by libnsx.so, and the level-parallel
// exchange (level_parallel.hip) with its collectives enqueued from C between the kernels they connect.
//
// The reference trains on one GPU (scripts/train/train_nersemble.py:272-274); the exchange is this package's contract
// (include/nsx.h, "level-parallel exchange").  Round 6 measured the emulated rank of an 8-rank job at ~1.9 ms of HOST time per
// step against ~2.0 ms of device time: five torch.distributed calls (~50 us each) and the Python between them were what
// the step waited for.  Here one call per direction enqueues pack -> collective -> kernels -> collective -> unpack on the
// caller's stream; the communicator is made once per job from a unique id the binding carries between the processes.
//
// RCCL is resolved at RUN time from the librccl.so.1 the process already holds (PyTorch ships and loads its own; a second
// copy of the library in one process would be a second runtime): libnsx.so has no link-time dependency on it, and a
// process that never makes a communicator never touches it.
#include "nsx_common.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>

struct nsx_comm {
    ncclComm_t comm;
    int world, rank;
};

namespace nsx {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllToAll) AllToAll = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static Rccl g_rccl;
static std::mutex g_rccl_mutex;
static std::string g_rccl_path;

static int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return NSX_OK;
    void* h = nullptr;
    if (!g_rccl_path.empty()) h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);       // the copy the process holds already
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_error("nsx_comm: librccl.so.1 cannot be loaded (%s); name it with nsx_comm_library", dlerror());
        return NSX_ERR_UNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
#define NSX_RCCL_SYM(name)                                                                          \
    r.name = reinterpret_cast<decltype(r.name)>(dlsym(h, "nccl" #name));                            \
    if (!r.name) {                                                                                  \
        set_error("nsx_comm: the loaded librccl has no nccl" #name);                                \
        return NSX_ERR_UNSUPPORTED;                                                                 \
    }
    NSX_RCCL_SYM(GetUniqueId) NSX_RCCL_SYM(CommInitRank) NSX_RCCL_SYM(CommDestroy) NSX_RCCL_SYM(AllGather)
    NSX_RCCL_SYM(AllToAll) NSX_RCCL_SYM(AllReduce) NSX_RCCL_SYM(GetErrorString)
#undef NSX_RCCL_SYM
    g_rccl = r;
    return NSX_OK;
}

static int rccl_fail(ncclResult_t e, const char* what) {
    set_error("%s: RCCL: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error");
    return NSX_ERR_HIP;
}

#define NSX_RCCL(call, what)                                        \
    do {                                                            \
        ncclResult_t e__ = (call);                                  \
        if (e__ != ncclSuccess) return rccl_fail(e__, what);        \
    } while (0)

// an emulated rank (one process stands for rank r of W, the others are replicas of it): every replica's payload
__global__ __launch_bounds__(256) void lp_replicate_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                           int W, int skip) {
    const int j = blockIdx.y;
    if (j == skip) return;
    uint4* out = dst + (int64_t)j * n16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = src[i];
}

static int comm_check(const nsx_comm* c, const nsx_lp_layout* lay, int emulate_rank, const char* who) {
    NSX_REQUIRE(c && c->comm && lay, "%s: NULL communicator / layout", who);
    if (emulate_rank >= 0) {
        NSX_REQUIRE(c->world == 1 && emulate_rank < lay->W, "%s: an emulated rank %d of %d needs a one-rank communicator (has %d)",
                    who, emulate_rank, lay->W, c->world);
    } else {
        NSX_REQUIRE(c->world == lay->W, "%s: the layout is for %d ranks, the communicator has %d", who, lay->W, c->world);
    }
    return NSX_OK;
}

// out block j = rank j's block [this rank] (W equal blocks of `bytes`); an emulated rank's one-rank collective hands the
// W blocks back as they are -- block j stands for what replica j would have sent
static int all_to_all(const nsx_comm* c, const nsx_lp_layout* lay, int emulate_rank, const uint8_t* send, uint8_t* recv,
                      int64_t bytes, hipStream_t st, const char* who) {
    const size_t count = (size_t)(emulate_rank >= 0 ? bytes * lay->W : bytes);
    NSX_RCCL(g_rccl.AllToAll(send, recv, count, ncclUint8, c->comm, st), who);
    return NSX_OK;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_comm_library(const char* path) {
    NSX_REQUIRE(path != nullptr, "nsx_comm_library: NULL path");
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    NSX_REQUIRE(!g_rccl.handle, "nsx_comm_library: RCCL is loaded already");
    g_rccl_path = path;
    return NSX_OK;
}

int nsx_comm_unique_id(uint8_t* id128) {
    NSX_REQUIRE(id128 != nullptr, "nsx_comm_unique_id: NULL argument");
    if (int rc = rccl_load()) return rc;
    static_assert(sizeof(ncclUniqueId) == NSX_COMM_ID_BYTES, "ncclUniqueId");
    ncclUniqueId id;
    NSX_RCCL(g_rccl.GetUniqueId(&id), "nsx_comm_unique_id");
    memcpy(id128, &id, sizeof(id));
    return NSX_OK;
}

int nsx_comm_create(const uint8_t* id128, int world_size, int rank, nsx_comm** out) {
    NSX_REQUIRE(id128 && out, "nsx_comm_create: NULL argument");
    NSX_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "nsx_comm_create: rank %d of %d", rank, world_size);
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    NSX_RCCL(g_rccl.CommInitRank(&comm, world_size, id, rank), "nsx_comm_create");
    nsx_comm* c = new nsx_comm;
    c->comm = comm; c->world = world_size; c->rank = rank;
    *out = c;
    return NSX_OK;
}

int nsx_comm_destroy(nsx_comm* comm) {
    if (!comm) return NSX_OK;
    if (comm->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm->comm);
    delete comm;
    return NSX_OK;
}

int nsx_comm_world_size(const nsx_comm* comm) { return comm ? comm->world : NSX_ERR_INVALID; }
int nsx_comm_rank(const nsx_comm* comm) { return comm ? comm->rank : NSX_ERR_INVALID; }

int nsx_comm_all_reduce_sum(nsx_comm* comm, float* data, int64_t count, void* stream) {
    NSX_REQUIRE(comm && comm->comm && (count == 0 || data) && count >= 0, "nsx_comm_all_reduce_sum: bad argument");
    if (count == 0) return NSX_OK;
    NSX_RCCL(g_rccl.AllReduce(data, data, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream),
             "nsx_comm_all_reduce_sum");
    return NSX_OK;
}

int nsx_lp_forward(const nsx_lp_layout* lay, nsx_comm* comm, int emulate_rank, const float* pn, const int32_t* slot, int64_t S,
                   const int64_t* n_device, const float* codes, int64_t code_stride, int rows, uint8_t* payload,
                   uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host, const nsx_half* tables,
                   const nsx_grid_geom* sub_geom, const float* window, uint8_t* send, float* codes_packed, uint8_t* recv,
                   nsx_half* feats, void* stream) {
    if (int rc = comm_check(comm, lay, emulate_rank, "nsx_lp_forward")) return rc;
    NSX_REQUIRE(payload && gathered && send && recv, "nsx_lp_forward: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = nsx_lp_fwd_pack(lay, pn, slot, S, n_device, codes, code_stride, rows, payload, stream)) return rc;
    if (emulate_rank >= 0) {
        NSX_RCCL(g_rccl.AllGather(payload, gathered + (int64_t)emulate_rank * lay->fwd_bytes, (size_t)lay->fwd_bytes, ncclUint8,
                                  comm->comm, st), "nsx_lp_forward (all-gather)");
        const int64_t n16 = lay->fwd_bytes / 16;                    // (fwd_bytes is a multiple of 256)
        int64_t bx = (n16 + 255) / 256;
        if (bx > 256) bx = 256;
        hipLaunchKernelGGL(lp_replicate_kernel, dim3((unsigned)bx, (unsigned)lay->W), dim3(256), 0, st,
                           reinterpret_cast<const uint4*>(payload), reinterpret_cast<uint4*>(gathered), n16, lay->W,
                           emulate_rank);
        NSX_LAUNCH_CHECK("nsx_lp_forward (replicas)");
    } else {
        NSX_RCCL(g_rccl.AllGather(payload, gathered, (size_t)lay->fwd_bytes, ncclUint8, comm->comm, st),
                 "nsx_lp_forward (all-gather)");
    }
    if (int rc = nsx_lp_fwd_run(lay, gathered, sizes_host, rows_host, tables, sub_geom, window, send, codes_packed, stream))
        return rc;
    if (int rc = all_to_all(comm, lay, emulate_rank, send, recv, lay->feat_bytes, st, "nsx_lp_forward (all-to-all)")) return rc;
    return nsx_lp_fwd_unpack(lay, recv, S, n_device, feats, stream);
}

int nsx_lp_backward(const nsx_lp_layout* lay, nsx_comm* comm, int emulate_rank, const float* dout, const float* pn,
                    const int32_t* slot, int64_t S, const int64_t* n_device, uint8_t* send, uint8_t* recv,
                    const uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host, const nsx_half* tables,
                    const nsx_grid_geom* sub_geom, const float* window, float* G, float* dz_scratch, float* csum_scratch,
                    uint8_t* ret, uint8_t* ret_recv, float* nonfinite, int rows, float* dx, float* dcode, void* stream) {
    if (int rc = comm_check(comm, lay, emulate_rank, "nsx_lp_backward")) return rc;
    NSX_REQUIRE(send && recv && ret && ret_recv, "nsx_lp_backward: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = nsx_lp_bwd_pack(lay, dout, pn, slot, S, n_device, send, stream)) return rc;
    if (int rc = all_to_all(comm, lay, emulate_rank, send, recv, lay->bwd_bytes, st, "nsx_lp_backward (all-to-all)")) return rc;
    if (int rc = nsx_lp_bwd_run(lay, recv, gathered, sizes_host, rows_host, tables, sub_geom, window, G, dz_scratch,
                                csum_scratch, ret, nonfinite, stream))
        return rc;
    if (int rc = all_to_all(comm, lay, emulate_rank, ret, ret_recv, lay->ret_bytes, st, "nsx_lp_backward (return all-to-all)"))
        return rc;
    return nsx_lp_bwd_unpack(lay, ret_recv, S, n_device, rows, dx, dcode, stream);
}

}  // extern "C"
