// fake_rccl.cpp -- TEST INFRASTRUCTURE: the ten RCCL entry points libvokselis_hip.so binds (vk_comm.hip: RcclApi),
// implemented inside ONE process with stream-ordered hipMemcpyAsync, so that the N > 1 branches of vk_gather_tiles and
// vk_group_render execute on a box with a single GPU.  Loaded only when the environment
// names it (VK_RCCL_LIB=tests/_build/libfake_rccl.so); the product never links it.
//
// Semantics kept from RCCL: a send and its matching receive (same communicator world, src -> dst, FIFO per pair) move
// `count` elements; the transfer is ordered after everything enqueued on the sender's stream before ncclSend and after
// everything on the receiver's stream before ncclRecv; work enqueued later on either stream waits for the transfer.
// Not kept: ncclCommInitRank does not block for the other ranks (one thread creates every rank in turn), several ranks
// may sit on one device, and a send posted before its receive only orders the sender's LATER work from the moment the
// receive arrives.  Single-threaded use only (the tests call from one thread).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct fakeComm *ncclComm_t;
typedef int ncclDataType_t;
}

namespace {

struct Pending { const void *src = nullptr; void *dst = nullptr; size_t bytes = 0; hipStream_t stream = nullptr; hipEvent_t posted = nullptr; int device = 0; };

struct World {
    int nranks = 0;
    std::map<std::pair<int, int>, std::deque<Pending>> sends, recvs;  // (src, dst) -> FIFO
    unsigned long long transfers = 0, bytes = 0;
};

std::map<std::string, std::shared_ptr<World>> g_worlds;  // by unique id
std::vector<std::weak_ptr<World>> g_all;                  // every world ever made (ncclCommInitAll's have no id)
unsigned long long g_next_id = 1;
unsigned long long g_total_transfers = 0, g_total_bytes = 0;

size_t type_bytes(int t) {
    switch (t) { case 0: case 1: case 10: case 11: return 1; case 6: case 9: return 2; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; default: return 0; }
}

// the transfer itself: dst stream waits for the sender's data, copies, and the sender's stream waits for the copy
ncclResult_t transfer(World &w, const Pending &s, const Pending &r) {
    if (s.bytes != r.bytes) return ncclInvalidArgument;
    hipEvent_t done = nullptr;
    if (hipSetDevice(r.device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamWaitEvent(r.stream, s.posted, 0) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(done, r.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamWaitEvent(s.stream, done, 0) != hipSuccess) return ncclUnhandledCudaError;
    (void)hipEventDestroy(done);      // (released by the runtime once it has fired)
    (void)hipEventDestroy(s.posted);
    if (r.posted) (void)hipEventDestroy(r.posted);
    w.transfers++; w.bytes += s.bytes;
    g_total_transfers++; g_total_bytes += s.bytes;
    return ncclSuccess;
}

}  // namespace

struct fakeComm { std::shared_ptr<World> world; int rank = 0, device = 0; };

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    std::memcpy(id->internal, "FAKERCCL", 8);
    std::memcpy(id->internal + 8, &g_next_id, sizeof(g_next_id));
    g_next_id++;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks <= 0 || rank < 0 || rank >= nranks || std::memcmp(id.internal, "FAKERCCL", 8) != 0) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof(id.internal));
    auto &w = g_worlds[key];
    if (!w) { w = std::make_shared<World>(); w->nranks = nranks; g_all.push_back(w); }
    if (w->nranks != nranks) return ncclInvalidArgument;
    fakeComm *c = new fakeComm();
    c->world = w; c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    auto w = std::make_shared<World>();
    w->nranks = ndev;
    g_all.push_back(w);
    for (int i = 0; i < ndev; i++) {
        fakeComm *c = new fakeComm();
        c->world = w; c->rank = i; c->device = devlist ? devlist[i] : i;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    delete comm;
    return ncclSuccess;
}

// ncclCommAbort: the communicator goes at once; sends / receives it had posted and that never met their partner are dropped
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    World &w = *comm->world;
    for (auto *qs : {&w.sends, &w.recvs})
        for (auto &q : *qs) {
            const bool mine = (qs == &w.sends) ? q.first.first == comm->rank : q.first.second == comm->rank;
            if (!mine) continue;
            for (auto &p : q.second) if (p.posted) (void)hipEventDestroy(p.posted);
            q.second.clear();
        }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !buf || peer < 0 || peer >= comm->world->nranks || peer == comm->rank || type_bytes(type) == 0) return ncclInvalidArgument;
    World &w = *comm->world;
    Pending s;
    s.src = buf; s.bytes = count * type_bytes(type); s.stream = stream; s.device = comm->device;
    if (hipSetDevice(comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&s.posted, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(s.posted, stream) != hipSuccess) return ncclUnhandledCudaError;
    auto key = std::make_pair(comm->rank, peer);
    auto &rq = w.recvs[key];
    if (!rq.empty()) { Pending r = rq.front(); rq.pop_front(); return transfer(w, s, r); }
    w.sends[key].push_back(s);
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !buf || peer < 0 || peer >= comm->world->nranks || peer == comm->rank || type_bytes(type) == 0) return ncclInvalidArgument;
    World &w = *comm->world;
    Pending r;
    r.dst = buf; r.bytes = count * type_bytes(type); r.stream = stream; r.device = comm->device;
    auto key = std::make_pair(peer, comm->rank);
    auto &sq = w.sends[key];
    if (!sq.empty()) { Pending s = sq.front(); sq.pop_front(); return transfer(w, s, r); }
    w.recvs[key].push_back(r);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
        case ncclInvalidArgument: return "invalid argument (fake RCCL)";
        default: return "error (fake RCCL)";
    }
}

// test hook: transfers and bytes moved so far by this shim (proves the n > 1 branches ran)
void fake_rccl_stats(unsigned long long *transfers, unsigned long long *bytes) {
    if (transfers) *transfers = g_total_transfers;
    if (bytes) *bytes = g_total_bytes;
}
// sends / receives still waiting for their partner, over all worlds (must be 0 once a collective step is complete)
unsigned long long fake_rccl_unmatched() {
    unsigned long long n = 0;
    for (auto &wp : g_all)
        if (auto w = wp.lock()) { for (auto &q : w->sends) n += q.second.size(); for (auto &q : w->recvs) n += q.second.size(); }
    return n;
}

}  // extern "C"
