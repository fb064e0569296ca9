// Data-parallel exchanges on RCCL communicators owned by the library (SURVEY.md section 8b "comm entry points"): what
// the reference gets from DistributedDataParallel + SyncBatchNorm over NCCL (train.py:80-102), without torch.distributed on
// the data path.  One cris_comm per process (= per GPU) holds TWO RCCL communicators over the same ranks:
//   * `sync`  - the small SyncBN statistics all-reduces, issued inline on the caller's compute stream (they are on the
//               critical path by construction);
//   * `grad`  - the gradient-arena stage all-reduces (24-254 MB each), issued on the communicator's own side stream behind
//               an event of the compute stream, so that they overlap the rest of backward.  A separate communicator because
//               collectives of one communicator execute in issue order: on a shared one the tiny SyncBN exchanges of the
//               layers still in backward would queue behind 100+ MB gradient messages.
// RCCL is resolved at run time (dlopen of the librccl.so.1 the process already has, else the system one): the library has
// no link-time dependency on it and single-GPU users never load it.  xGMI is point-to-point (7 links per GPU): message
// sizes are the caller's business (few large stage messages, see cris/pytorch_amd/dist.py), not this file's.
#include "common.h"
#include "../../../include/cris_hip.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct rccl_api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char where[256] = {0};
};
rccl_api g_rccl;

template <typename F>
bool sym(F& fn, const char* name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    return fn != nullptr;
}

// 0 on success; the error string names what was tried
int rccl_load() {
    if (g_rccl.AllReduce) return 0;
    const char* tries[] = {getenv("CRIS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // the copy the process already holds (torch.distributed's, when torch is imported) wins: one RCCL per process
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (h) snprintf(g_rccl.where, sizeof(g_rccl.where), "librccl.so.1 (already loaded)");
    for (size_t i = 0; !h && i < sizeof(tries) / sizeof(tries[0]); ++i) {
        if (!tries[i] || !*tries[i]) continue;
        h = dlopen(tries[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) snprintf(g_rccl.where, sizeof(g_rccl.where), "%s", tries[i]);
    }
    if (!h) {
        cris_set_error("cris_comm: RCCL not found (tried CRIS_RCCL_LIB, librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1): %s", dlerror());
        return -2;
    }
    g_rccl.handle = h;
    if (!(sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && sym(g_rccl.CommInitRank, "ncclCommInitRank") &&
          sym(g_rccl.CommDestroy, "ncclCommDestroy") && sym(g_rccl.AllReduce, "ncclAllReduce") &&
          sym(g_rccl.Broadcast, "ncclBroadcast") && sym(g_rccl.GetErrorString, "ncclGetErrorString"))) {
        cris_set_error("cris_comm: %s lacks an expected nccl* symbol", g_rccl.where);
        g_rccl.AllReduce = nullptr;
        return -3;
    }
    return 0;
}
}  // namespace

struct cris_comm {
    int rank, world, device;
    ncclComm_t sync, grad;
    hipStream_t side;               // the gradient exchange's stream
    hipEvent_t ready, done;         // compute stream -> side ; side -> compute stream
    long buckets;                   // gradient messages issued since the last cris_comm_wait
};

#define RCCL_CHECK(call, what)                                                                              \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            cris_set_error("%s: %s failed: %s", __func__, what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return 100 + (int)r_;                                                                           \
        }                                                                                                   \
    } while (0)
#define HIP_CHECK(call, what)                                                                    \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            cris_set_error("%s: %s failed: %s", __func__, what, hipGetErrorString(e_));          \
            return (int)e_;                                                                      \
        }                                                                                        \
    } while (0)

extern "C" const char* cris_comm_rccl_path(void) { return rccl_load() == 0 ? g_rccl.where : nullptr; }

extern "C" int cris_comm_unique_id(void* id_bytes) {
    CRIS_CHECK_ARG(id_bytes, "null id buffer (CRIS_COMM_ID_BYTES bytes)");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId ids[2];
    static_assert(sizeof(ids) == CRIS_COMM_ID_BYTES, "CRIS_COMM_ID_BYTES = two ncclUniqueId");
    RCCL_CHECK(g_rccl.GetUniqueId(&ids[0]), "ncclGetUniqueId");
    RCCL_CHECK(g_rccl.GetUniqueId(&ids[1]), "ncclGetUniqueId");
    memcpy(id_bytes, ids, sizeof(ids));
    return 0;
}

extern "C" int cris_comm_init(int rank, int world, const void* id_bytes, cris_comm** out) {
    CRIS_CHECK_ARG(out && id_bytes && world >= 1 && rank >= 0 && rank < world, "bad rank / world / id");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId ids[2];
    memcpy(ids, id_bytes, sizeof(ids));
    cris_comm* c = new cris_comm();
    c->rank = rank; c->world = world; c->buckets = 0;
    c->sync = c->grad = nullptr; c->side = nullptr; c->ready = c->done = nullptr;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        cris_set_error("%s: stream / event creation failed: %s", __func__, hipGetErrorString(e));
        cris_comm_destroy(c);
        return (int)e;
    }
    // every rank makes the two calls in the same order (ncclCommInitRank is collective over the ranks of its id)
    ncclResult_t r = g_rccl.CommInitRank(&c->sync, world, ids[0], rank);
    if (r == ncclSuccess) r = g_rccl.CommInitRank(&c->grad, world, ids[1], rank);
    if (r != ncclSuccess) {
        cris_set_error("%s: ncclCommInitRank failed: %s", __func__, g_rccl.GetErrorString(r));
        cris_comm_destroy(c);
        return 100 + (int)r;
    }
    *out = c;
    return 0;
}

extern "C" int cris_comm_destroy(cris_comm* c) {
    if (!c) return 0;
    if (c->side) hipStreamSynchronize(c->side);
    if (c->sync) g_rccl.CommDestroy(c->sync);
    if (c->grad) g_rccl.CommDestroy(c->grad);
    if (c->ready) hipEventDestroy(c->ready);
    if (c->done) hipEventDestroy(c->done);
    if (c->side) hipStreamDestroy(c->side);
    delete c;
    return 0;
}

extern "C" int cris_comm_rank(const cris_comm* c) { return c ? c->rank : -1; }
extern "C" int cris_comm_world(const cris_comm* c) { return c ? c->world : -1; }

// SyncBN exchange ([S1 | S2] forward, [sum g | sum g*xhat] backward): in place, on the caller's stream
extern "C" int cris_comm_syncbn_exchange(cris_comm* c, float* stats, size_t n, void* stream) {
    CRIS_CHECK_ARG(c && stats && n > 0, "bad args");
    RCCL_CHECK(g_rccl.AllReduce(stats, stats, n, ncclFloat32, ncclSum, c->sync, (hipStream_t)stream), "ncclAllReduce (sync)");
    return 0;
}

// one stage of the gradient arena: the side stream waits for what `ready_stream` has queued so far (the stage's backward),
// then all-reduces the range in place; cris_comm_wait makes the optimizer's stream wait for all stages issued
extern "C" int cris_comm_allreduce_bucket(cris_comm* c, float* buf, size_t n, void* ready_stream) {
    CRIS_CHECK_ARG(c && buf && n > 0, "bad args");
    HIP_CHECK(hipEventRecord(c->ready, (hipStream_t)ready_stream), "hipEventRecord");
    HIP_CHECK(hipStreamWaitEvent(c->side, c->ready, 0), "hipStreamWaitEvent");
    RCCL_CHECK(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, c->grad, c->side), "ncclAllReduce (grad)");
    c->buckets++;
    return 0;
}

extern "C" int cris_comm_wait(cris_comm* c, void* stream) {
    CRIS_CHECK_ARG(c, "null communicator");
    if (c->buckets == 0) return 0;
    HIP_CHECK(hipEventRecord(c->done, c->side), "hipEventRecord");
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, c->done, 0), "hipStreamWaitEvent");
    c->buckets = 0;
    return 0;
}

// rank `root`'s values to every rank (parameters and BatchNorm buffers at construction: DistributedDataParallel's behaviour,
// and what the single-exchange SyncBN needs - identical running means)
extern "C" int cris_comm_broadcast(cris_comm* c, void* buf, size_t nbytes, int root, void* stream) {
    CRIS_CHECK_ARG(c && buf && nbytes > 0 && root >= 0 && root < c->world, "bad args");
    RCCL_CHECK(g_rccl.Broadcast(buf, buf, nbytes, ncclUint8, root, c->sync, (hipStream_t)stream), "ncclBroadcast");
    return 0;
}
