// skg_comm.cpp -- the library's own RCCL communicator for the gradient exchange of the data-parallel training step.
//
// Reference: utils.py:202-205 (DistributedDataParallel: per-parameter hooks, NCCL buckets) and main:26-31 (one process per
// GPU, backend "nccl").  Here the backward's worker thread (skg_train_plan.hip, skg_context) all-reduces the gradient
// arena chunk by chunk itself: behind every stage that completes a chunk it orders THIS communicator's stream behind the
// stage's device-scope event and calls ncclAllReduce -- no Python between the stage and its collective, no per-collective
// event pair of a framework's process group on the step's queues (measured at world size 1: the same exchange through
// torch.distributed costs the batch-4 bf16 step +0.2 ... 0.4 ms of 1.39).
//
// RCCL is bound at RUN time (dlopen of the librccl the process already has, normally PyTorch's): the library keeps no
// link-time dependency on it, and a process without RCCL simply gets SKG_E_UNSUPPORTED from skg_comm_load().
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "skghoi.h"

namespace {

struct Api {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommAbort) comm_abort = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
Api g_api;
std::mutex g_api_m;
thread_local char g_err[256] = "";

int fail(const char* what, const char* detail) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, detail ? detail : "?");
    return SKG_E_UNSUPPORTED;
}
int nccl_fail(const char* what, ncclResult_t r) {
    snprintf(g_err, sizeof(g_err), "%s: %s (ncclResult %d)", what,
             g_api.error_string ? g_api.error_string(r) : "?", (int)r);
    return SKG_E_COMM;
}
int hip_fail(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace

struct skg_comm {
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;       // the collectives' stream: ordered behind one stage event per chunk
    hipEvent_t done = nullptr;          // behind the last collective of a step (timing enabled: exposed-wait measurement)
    hipEvent_t aux = nullptr;           // behind a collective outside the step's chunk sequence (begin / end pair)
    hipEvent_t tail = nullptr;          // scratch: the tail of a caller's stream the exchange stream is ordered behind
    hipEvent_t joined = nullptr;        // scratch: the exchange stream's tail a caller's stream is ordered behind (in-stream chunk)
    int rank = 0, world = 1, device = 0;
    long collectives = 0;               // issued since creation
    bool dead = false;                  // aborted (a collective or its ordering failed on this rank): nothing further is issued
};

extern "C" {

const char* skg_comm_last_error(void) { return g_err; }

int skg_comm_load(const char* path) {
    std::lock_guard<std::mutex> g(g_api_m);
    if (g_api.lib) return 0;
    void* h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);       // whatever the process already loaded
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("dlopen(librccl)", dlerror());
    Api a;
    a.lib = h;
#define SKG_SYM(field, name)                                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                  \
    if (!a.field) { dlclose(h); return fail("dlsym", name); }
    SKG_SYM(get_unique_id, "ncclGetUniqueId")
    SKG_SYM(comm_init_rank, "ncclCommInitRank")
    SKG_SYM(comm_destroy, "ncclCommDestroy")
    SKG_SYM(comm_abort, "ncclCommAbort")
    SKG_SYM(all_reduce, "ncclAllReduce")
    SKG_SYM(error_string, "ncclGetErrorString")
#undef SKG_SYM
    g_api = a;
    return 0;
}

int skg_comm_unique_id(void* id_out) {
    if (!id_out) return SKG_E_ARG;
    if (!g_api.lib) return fail("skg_comm_unique_id", "skg_comm_load() first");
    static_assert(sizeof(ncclUniqueId) == SKG_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = g_api.get_unique_id(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int skg_comm_create(const void* id_in, int rank, int world, skg_comm** out) {
    if (!id_in || !out || world < 1 || rank < 0 || rank >= world) return SKG_E_ARG;
    *out = nullptr;
    if (!g_api.lib) return fail("skg_comm_create", "skg_comm_load() first");
    skg_comm* c = new (std::nothrow) skg_comm;
    if (!c) return SKG_E_ARG;
    c->rank = rank; c->world = world;
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->done);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->aux, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->tail, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->joined, hipEventDisableTiming);
    if (e != hipSuccess) { int rc = hip_fail("skg_comm_create", e); skg_comm_destroy(c); return rc; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclResult_t r = g_api.comm_init_rank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { c->nccl = nullptr; int rc = nccl_fail("ncclCommInitRank", r); skg_comm_destroy(c); return rc; }
    *out = c;
    return 0;
}

/* Tears the communicator down without waiting for its peers (ncclCommAbort): a rank whose step failed between two chunk
 * collectives calls it so that the peers -- already inside the next ncclAllReduce, on a communicator no watchdog looks after
 * -- get an error instead of waiting for ever.  The object stays valid (dead: every later collective returns SKG_E_COMM)
 * until skg_comm_destroy. */
int skg_comm_abort(skg_comm* c) {
    if (!c) return SKG_E_ARG;
    if (c->dead) return 0;
    c->dead = true;
    if (c->nccl && g_api.comm_abort) {
        ncclResult_t r = g_api.comm_abort(c->nccl);
        c->nccl = nullptr;
        if (r != ncclSuccess) return nccl_fail("ncclCommAbort", r);
    }
    return 0;
}
int skg_comm_dead(const skg_comm* c) { return c && c->dead ? 1 : 0; }

void skg_comm_destroy(skg_comm* c) {
    if (!c) return;
    if (c->stream && !c->dead) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && g_api.comm_destroy) (void)g_api.comm_destroy(c->nccl);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->aux) (void)hipEventDestroy(c->aux);
    if (c->tail) (void)hipEventDestroy(c->tail);
    if (c->joined) (void)hipEventDestroy(c->joined);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int skg_comm_world(const skg_comm* c) { return c ? c->world : 0; }
int skg_comm_rank(const skg_comm* c) { return c ? c->rank : -1; }
int64_t skg_comm_collectives(const skg_comm* c) { return c ? (int64_t)c->collectives : 0; }

}  // extern "C"

// A collective that fails on this rank leaves the peers inside it: whoever issued it (the backward's worker, the preparation's
// normaliser all-reduce on the caller's thread), the communicator is aborted on the spot so that they get an error too.
static int collective_failed(skg_comm* c, ncclResult_t r) {
    const int rc = nccl_fail("ncclAllReduce", r);
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    (void)skg_comm_abort(c);
    memcpy(g_err, keep, sizeof(keep));                 // (the text of the failure, not of the abort)
    return rc;
}

// ---- what the backward's worker thread calls (skg_train_plan.hip): the collective of ONE arena chunk behind `after`
// (a recorded event; NULL: behind nothing), and the close of a step -- `stream` ordered behind every chunk issued so far
int skg_comm_chunk(skg_comm* c, hipEvent_t after, float* p, int64_t n) {
    if (!c || !p || n < 0) return SKG_E_ARG;
    if (c->dead) { snprintf(g_err, sizeof(g_err), "the communicator was aborted after an earlier failure"); return SKG_E_COMM; }
    if (n == 0) return 0;
    if (after) {
        hipError_t e = hipStreamWaitEvent(c->stream, after, 0);
        if (e != hipSuccess) return hip_fail("hipStreamWaitEvent(exchange stream)", e);
    }
    ncclResult_t r = g_api.all_reduce(p, p, (size_t)n, ncclFloat32, ncclSum, c->nccl, c->stream);
    if (r != ncclSuccess) return collective_failed(c, r);
    ++c->collectives;
    return 0;
}

// The LAST chunk of a step, on the step's own stream: nothing is left to overlap it with (the backward has ended), so the
// collective goes where the optimizer waits anyway -- no event on the backward's queue in front of it, no cross-queue hop
// behind it (measured at world size 1: the two hops at the tail were most of what the data-parallel route cost).  `stream` is
// first ordered behind the exchange stream's tail (the earlier chunks: normally long finished); `done` is recorded behind
// the collective for skg_comm_exposed_ms.
int skg_comm_chunk_in_stream(skg_comm* c, hipStream_t stream, float* p, int64_t n) {
    if (!c || !p || n < 0) return SKG_E_ARG;
    if (c->dead) { snprintf(g_err, sizeof(g_err), "the communicator was aborted after an earlier failure"); return SKG_E_COMM; }
    hipError_t e = hipEventRecord(c->joined, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, c->joined, 0);
    if (e != hipSuccess) return hip_fail("skg_comm_chunk_in_stream", e);
    if (n > 0) {
        ncclResult_t r = g_api.all_reduce(p, p, (size_t)n, ncclFloat32, ncclSum, c->nccl, stream);
        if (r != ncclSuccess) return collective_failed(c, r);
        ++c->collectives;
    }
    e = hipEventRecord(c->done, stream);
    return e == hipSuccess ? 0 : hip_fail("skg_comm_chunk_in_stream", e);
}

hipStream_t skg_comm_stream(skg_comm* c) { return c ? c->stream : nullptr; }

int skg_comm_close_step(skg_comm* c, hipStream_t stream) {
    if (!c) return SKG_E_ARG;
    hipError_t e = hipEventRecord(c->done, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, c->done, 0);
    return e == hipSuccess ? 0 : hip_fail("skg_comm_close_step", e);
}

extern "C" {

int skg_comm_all_reduce_chunks_f32(skg_comm* c, float* arena, const int64_t* ends_host, int n_chunks, void* stream) {
    if (!c || !arena || !ends_host || n_chunks < 1 || n_chunks > SKG_TRAIN_BWD_STAGES) return SKG_E_ARG;
    int64_t done = 0;
    for (int i = 0; i < n_chunks; ++i) {
        if (ends_host[i] < done) return SKG_E_ARG;
        done = ends_host[i];
    }
    // the exchange stream behind everything `stream` holds now (the gradients' producers)
    hipError_t e = hipEventRecord(c->tail, (hipStream_t)stream);
    int rc = e == hipSuccess ? 0 : hip_fail("hipEventRecord", e);
    done = 0;
    for (int i = 0; i < n_chunks && !rc; ++i) {
        if (i == n_chunks - 1) rc = skg_comm_chunk_in_stream(c, (hipStream_t)stream, arena + done, ends_host[i] - done);
        else rc = skg_comm_chunk(c, i == 0 ? c->tail : nullptr, arena + done, ends_host[i] - done);
        done = ends_host[i];
    }
    return rc;
}

int skg_comm_all_reduce_begin_f32(skg_comm* c, float* p, int64_t n, void* stream) {
    if (!c || !p || n < 0) return SKG_E_ARG;
    hipError_t e = hipEventRecord(c->tail, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("hipEventRecord", e);
    int rc = skg_comm_chunk(c, c->tail, p, n);
    if (rc) return rc;
    e = hipEventRecord(c->aux, c->stream);
    return e == hipSuccess ? 0 : hip_fail("hipEventRecord", e);
}

int skg_comm_all_reduce_end(skg_comm* c, void* stream) {
    if (!c) return SKG_E_ARG;
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, c->aux, 0);
    return e == hipSuccess ? 0 : hip_fail("hipStreamWaitEvent", e);
}

int skg_comm_exposed_ms(skg_comm* c, void* after_event, float* ms_out) {
    if (!c || !after_event || !ms_out) return SKG_E_ARG;
    hipError_t e = hipEventSynchronize(c->done);
    if (e == hipSuccess) e = hipEventSynchronize((hipEvent_t)after_event);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, (hipEvent_t)after_event, c->done);
    if (e != hipSuccess) return hip_fail("skg_comm_exposed_ms", e);
    *ms_out = ms > 0.f ? ms : 0.f;
    return 0;
}

}  // extern "C"
