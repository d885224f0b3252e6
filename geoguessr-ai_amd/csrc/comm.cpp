// Data-parallel exchange behind the C-ABI: an opaque gg_comm holding one RCCL communicator (xGMI inside a node),
// with exactly the three collectives the path needs -- gradient sum all-reduce, parameter / BatchNorm-buffer broadcast, barrier.
// Replaces what the reference gets from Accelerate -> torch DistributedDataParallel -> NCCL (training/train_eval_loop.py:184-187,200-202,234).
//
// RCCL is bound lazily (dlopen "librccl.so.1" + dlsym): libgg.so keeps no link-time dependency on it and, inside a process that has
// already loaded RCCL (PyTorch-ROCm), the loader hands back that same library.  One communicator per process / device, one process
// per GPU.  Collectives are enqueued on the caller's HIP stream; nothing here synchronises the host except gg_comm_barrier(sync != 0).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include "../../include/gg.h"
#include "prof.h"

namespace {
typedef void* nccl_comm_t;
struct nccl_uid { char internal[128]; };                       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40-43)
enum { NCCL_SUM = 0, NCCL_CHAR = 0, NCCL_FLOAT = 7 };          // ncclRedOp_t / ncclDataType_t values (rccl.h:448-466)
struct Api {
    int (*GetUniqueId)(nccl_uid*);
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid, int);
    int (*CommDestroy)(nccl_comm_t);
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
    int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
    const char* (*GetErrorString)(int);
    bool ok = false;
};
Api g_api;
int load_api() {
    if (g_api.ok) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { gg_set_error("gg_comm: cannot load RCCL (%s)", dlerror()); return -1; }
#define GG_SYM(field, name)                                                                    \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, name));                     \
    if (!g_api.field) { gg_set_error("gg_comm: RCCL lacks %s", name); return -1; }
    GG_SYM(GetUniqueId, "ncclGetUniqueId") GG_SYM(CommInitRank, "ncclCommInitRank") GG_SYM(CommDestroy, "ncclCommDestroy")
    GG_SYM(AllReduce, "ncclAllReduce") GG_SYM(Broadcast, "ncclBroadcast") GG_SYM(GetErrorString, "ncclGetErrorString")
#undef GG_SYM
    // the enum values above are those of the NCCL 2.x ABI (ncclFloat = 7 since 2.0; RCCL keeps NCCL's numbering): refuse anything else
    int (*get_version)(int*) = reinterpret_cast<int (*)(int*)>(dlsym(h, "ncclGetVersion"));
    int ver = 0;
    if (!get_version || get_version(&ver) != 0 || ver < 2000 || ver >= 30000) {
        gg_set_error("gg_comm: RCCL reports NCCL API version %d; this binding is written against the 2.x ABI", ver);
        return -1;
    }
    g_api.ok = true;
    return 0;
}
}  // namespace

struct gg_comm { nccl_comm_t comm; int rank, world, device; float* scratch; };

#define GG_NCCL(call)                                                                          \
    do {                                                                                       \
        int r_ = (call);                                                                       \
        if (r_ != 0) { gg_set_error("%s failed: %s", #call, g_api.GetErrorString(r_)); return -4; } \
    } while (0)

extern "C" int gg_comm_unique_id(void* out128) {
    if (!out128) { gg_set_error("gg_comm_unique_id: null"); return -1; }
    if (load_api()) return -1;
    nccl_uid id;
    GG_NCCL(g_api.GetUniqueId(&id));
    memcpy(out128, id.internal, 128);
    return 0;
}
extern "C" int gg_comm_create(gg_comm** out, const void* unique_id128, int rank, int world, int device) {
    if (!out || !unique_id128 || world < 1 || rank < 0 || rank >= world) { gg_set_error("gg_comm_create: bad args"); return -1; }
    if (load_api()) return -1;
    if (hipSetDevice(device) != hipSuccess) { gg_set_error("gg_comm_create: hipSetDevice(%d) failed", device); return -2; }
    nccl_uid id;
    memcpy(id.internal, unique_id128, 128);
    gg_comm* c = new gg_comm{nullptr, rank, world, device, nullptr};
    int r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { gg_set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(r)); delete c; return -4; }
    if (hipMalloc(reinterpret_cast<void**>(&c->scratch), 256) != hipSuccess || hipMemset(c->scratch, 0, 256) != hipSuccess) {   // the barrier sums this word: zeros, not garbage
        gg_set_error("gg_comm_create: hipMalloc / hipMemset failed"); g_api.CommDestroy(c->comm); delete c; return -2;
    }
    *out = c;
    return 0;
}
extern "C" int gg_comm_destroy(gg_comm* c) {
    if (!c) return 0;
    if (c->scratch) (void)hipFree(c->scratch);
    if (g_api.ok && c->comm) g_api.CommDestroy(c->comm);
    delete c;
    return 0;
}
extern "C" int gg_comm_rank(const gg_comm* c) { return c ? c->rank : -1; }
extern "C" int gg_comm_world(const gg_comm* c) { return c ? c->world : -1; }
// in-place sum over the ranks of n floats (the flat trainable-gradient ranges; the 1/world average is folded into gg_adamw_step)
extern "C" int gg_comm_allreduce_sum_f32(gg_comm* c, float* buf, int64_t n, void* stream) {
    if (!c || !buf || n <= 0) { gg_set_error("gg_comm_allreduce_sum_f32: bad args"); return -1; }
    GG_NCCL(g_api.AllReduce(buf, buf, (size_t)n, NCCL_FLOAT, NCCL_SUM, c->comm, (hipStream_t)stream));
    return 0;
}
// rank `root`'s bytes overwrite everybody's (parameters at start-up, BatchNorm running statistics before a training forward)
extern "C" int gg_comm_broadcast(gg_comm* c, void* buf, int64_t bytes, int root, void* stream) {
    if (!c || !buf || bytes <= 0 || root < 0 || root >= c->world) { gg_set_error("gg_comm_broadcast: bad args"); return -1; }
    GG_NCCL(g_api.Broadcast(buf, buf, (size_t)bytes, NCCL_CHAR, root, c->comm, (hipStream_t)stream));
    return 0;
}
// a 1-element all-reduce on `stream`; sync != 0 additionally waits for it on the host (rank rendezvous)
extern "C" int gg_comm_barrier(gg_comm* c, void* stream, int sync) {
    if (!c) { gg_set_error("gg_comm_barrier: null comm"); return -1; }
    GG_NCCL(g_api.AllReduce(c->scratch, c->scratch, 1, NCCL_FLOAT, NCCL_SUM, c->comm, (hipStream_t)stream));
    if (sync && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { gg_set_error("gg_comm_barrier: stream sync failed"); return -2; }
    return 0;
}
