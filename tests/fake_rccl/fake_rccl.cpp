// fake_rccl.cpp -- TEST TRANSPORT, not product code.  A stand-in for librccl that lets two ranks which share ONE GPU (all a
// one-GPU test box can hold: RCCL itself refuses two ranks on one device) run the library's worker-issued gradient exchange
// end to end: skg_comm_load(<this library>) binds the six entry points below instead of RCCL's.  ncclAllReduce is a blocking
// host-side sum through a POSIX shared-memory file, in rank order (deterministic); every call first compares (sequence
// number, element count) across the ranks, so a rank that issues its collectives in another order fails the test with an
// error instead of hanging it.  Built by tests/test_trainer.py with g++ against the ROCm headers.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
constexpr int MAX_RANKS = 8;
constexpr size_t PIECE = 1u << 20;                 // floats per exchange piece
struct Shared {
    std::atomic<int> arrived;                      // sense-reversing barrier
    std::atomic<int> sense;
    std::atomic<int> attached;
    std::atomic<int> aborted;                      // a rank aborted its communicator: nobody waits for it any more
    long seq[MAX_RANKS];
    long count[MAX_RANKS];
    float slot[MAX_RANKS][PIECE];
};
struct Comm {
    Shared* sh = nullptr;
    int rank = 0, n = 1, local_sense = 0;
    long seq = 0;
    char name[64];
};
bool barrier(Comm* c) {
    if (c->sh->aborted.load()) return false;
    c->local_sense ^= 1;
    if (c->sh->arrived.fetch_add(1) + 1 == c->n) {
        c->sh->arrived.store(0);
        c->sh->sense.store(c->local_sense);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->sense.load() != c->local_sense) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (c->sh->aborted.load()) return false;                                                   // a peer aborted (ncclCommAbort)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;      // a peer never arrived
    }
    return true;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/skgfake_%d_%ld", (int)getpid(),
             (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS) return ncclInvalidArgument;
    Comm* c = new Comm;
    c->rank = rank; c->n = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); delete c; return ncclSystemError; }      // (zero-filled on creation)
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared*>(p);
    c->sh->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < nranks) {                                                      // collective, like the real one
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return ncclSystemError;
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->sh) { munmap(c->sh, sizeof(Shared)); shm_unlink(c->name); }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c && c->sh) c->sh->aborted.store(1);               // like the real one: the peers' pending collectives end with an error
    return ncclCommDestroy(comm);
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake transport error (order / size mismatch or a missing peer)"; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || dt != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    // every rank must be in the SAME collective: sequence number and size are compared first
    ++c->seq;
    {   // failure injection (tests): SKG_FAKE_RCCL_FAIL = "<rank>:<sequence number>" -- that collective fails on that rank
        // before it meets its peers (what a transport error in the middle of a backward looks like to the caller)
        const char* f = getenv("SKG_FAKE_RCCL_FAIL");
        int fr = -1; long fs = -1;
        if (f && sscanf(f, "%d:%ld", &fr, &fs) == 2 && fr == c->rank && fs == c->seq) return ncclInternalError;
    }
    c->sh->seq[c->rank] = c->seq; c->sh->count[c->rank] = (long)count;
    if (!barrier(c)) return ncclSystemError;
    bool same = true;
    for (int r = 0; r < c->n; ++r) same = same && c->sh->seq[r] == c->seq && c->sh->count[r] == (long)count;
    if (!barrier(c)) return ncclSystemError;
    if (!same) return ncclInvalidUsage;
    std::vector<float> host(count);
    if (hipMemcpy(host.data(), send, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t o = 0; o < count; o += PIECE) {
        const size_t n = count - o < PIECE ? count - o : PIECE;
        memcpy(c->sh->slot[c->rank], host.data() + o, n * sizeof(float));
        if (!barrier(c)) return ncclSystemError;
        for (size_t i = 0; i < n; ++i) {
            float s = c->sh->slot[0][i];
            for (int r = 1; r < c->n; ++r) s += c->sh->slot[r][i];                                 // rank order: every rank gets the same bits
            host[o + i] = s;
        }
        if (!barrier(c)) return ncclSystemError;
    }
    if (hipMemcpy(recv, host.data(), count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
