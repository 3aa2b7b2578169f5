// vk_comm.hip -- multi-GPU: RCCL over xGMI behind the C-ABI.  vk_comm_* / vk_gather_tiles (one process per
// GPU) and vk_group_* (one process per node).  librccl is loaded on first use.
#include "vk_ctx.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

// ---- RCCL, loaded on first use ---------------------------------------------------------------------
// librccl is 570 MB; a single-GPU user never pays for it.  dlopen by SONAME finds the copy a host process
// already carries (PyTorch-ROCm bundles one), so a process never holds two.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static RcclApi g_rccl;

static bool rccl_load() {
    if (g_rccl.lib) return true;
    // VK_RCCL_LIB names the library to bind instead (a particular RCCL build; the test suite's single-process stand-in,
    // tests/fake_rccl.cpp, through which the N > 1 branches below run on a one-GPU box).  It is taken or refused -- no
    // silent return to the system's copy.
    const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void *h = nullptr;
    const char *forced = std::getenv("VK_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { g_rccl.err = std::string("VK_RCCL_LIB=") + forced + ": " + dlerror(); return false; }
    }
    for (const char *n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!h) { g_rccl.err = std::string("RCCL not found: ") + dlerror(); return false; }
    auto sym = [&](const char *n) -> void * { void *p = dlsym(h, n); if (!p) g_rccl.err = std::string("RCCL symbol missing: ") + n; return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(sym("ncclCommAbort"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.CommAbort || !g_rccl.GroupStart || !g_rccl.GroupEnd ||
        !g_rccl.Send || !g_rccl.Recv || !g_rccl.GetErrorString) { dlclose(h); return false; }
    g_rccl.lib = h;
    return true;
}

#define NCCL_TRY(ctx, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(ctx, VK_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

void comm_release(vk_ctx *ctx) {
    if (ctx->comm && ctx->comm_owned && g_rccl.lib) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_size = 0; ctx->comm_rank = 0; ctx->comm_owned = false;
}

extern "C" {

int vk_comm_available(void) {
    if (!rccl_load()) return fail(nullptr, VK_ERR_UNSUPPORTED, g_rccl.err);
    return VK_OK;
}

int vk_comm_unique_id(void *id128) {
    if (!id128) return VK_ERR_INVALID;
    if (!rccl_load()) return fail(nullptr, VK_ERR_UNSUPPORTED, g_rccl.err);
    static_assert(sizeof(ncclUniqueId) == VK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(nullptr, g_rccl.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return VK_OK;
}

int vk_comm_init_rank(vk_ctx *ctx, const void *id128, int rank, int nranks) {
    if (!ctx || !id128) return fail(ctx, VK_ERR_INVALID, "vk_comm_init_rank: NULL argument");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(ctx, VK_ERR_INVALID, "vk_comm_init_rank: rank/nranks");
    if (ctx->in_group) return fail(ctx, VK_ERR_INVALID, "vk_comm_init_rank: this context belongs to a vk_group (its communicator is the group's)");
    if (!rccl_load()) return fail(ctx, VK_ERR_UNSUPPORTED, g_rccl.err);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NCCL_TRY(ctx, g_rccl.CommInitRank(&ctx->comm, nranks, id, rank));
    ctx->comm_rank = rank; ctx->comm_size = nranks; ctx->comm_owned = true;
    return VK_OK;
}

int vk_comm_destroy(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->in_group) return fail(ctx, VK_ERR_INVALID, "vk_comm_destroy: this context belongs to a vk_group (vk_group_destroy releases it)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    comm_release(ctx);
    return VK_OK;
}

// A rank that has found a peer gone (a gather that does not complete within the caller's time limit) cannot use vk_comm_destroy: that
// waits for the stream the dead transfer sits on.  ncclCommAbort cancels what is in flight and frees the communicator.
int vk_comm_abort(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->in_group) return fail(ctx, VK_ERR_INVALID, "vk_comm_abort: this context belongs to a vk_group");
    if (!ctx->comm) return VK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclComm_t c = ctx->comm;
    const bool owned = ctx->comm_owned;
    ctx->comm = nullptr; ctx->comm_size = 0; ctx->comm_rank = 0; ctx->comm_owned = false;
    if (owned && g_rccl.lib) NCCL_TRY(ctx, g_rccl.CommAbort(c));
    return VK_OK;
}

int vk_comm_info(vk_ctx *ctx, int *rank, int *nranks) {
    if (!ctx) return VK_ERR_INVALID;
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm_size;
    return VK_OK;
}

// Every rank contributes n_pixels pixels of the backbuffer's format; the root receives [nranks][n_pixels].
// Every peer has its own xGMI link to the root, so the gather is one send per peer inside one group.
int vk_gather_tiles(vk_ctx *ctx, const void *send, void *recv, size_t n_pixels, int root, void *hip_stream) {
    if (!ctx || !send) return fail(ctx, VK_ERR_INVALID, "vk_gather_tiles: NULL argument");
    if (!ctx->comm) return fail(ctx, VK_ERR_INVALID, "vk_gather_tiles: no communicator (vk_comm_init_rank)");
    if (root < 0 || root >= ctx->comm_size) return fail(ctx, VK_ERR_INVALID, "vk_gather_tiles: root out of range");
    if (ctx->comm_rank == root && !recv) return fail(ctx, VK_ERR_INVALID, "vk_gather_tiles: the root needs a receive buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = n_pixels * wire_px_bytes(ctx->out_format, ctx->wire);
    if (bytes == 0) return VK_OK;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    if (ctx->comm_rank == root) {
        unsigned char *r = static_cast<unsigned char *>(recv);
        if (r + (size_t)root * bytes != send) HIP_TRY(ctx, hipMemcpyAsync(r + (size_t)root * bytes, send, bytes, hipMemcpyDeviceToDevice, st));
        if (ctx->comm_size > 1) {
            NCCL_TRY(ctx, g_rccl.GroupStart());
            ncclResult_t bad = ncclSuccess;  // (an error inside the group still closes it: a group left open would swallow every later call)
            for (int p = 0; p < ctx->comm_size && bad == ncclSuccess; p++)
                if (p != root) bad = g_rccl.Recv(r + (size_t)p * bytes, bytes, ncclUint8, p, ctx->comm, st);
            const ncclResult_t end = g_rccl.GroupEnd();
            if (bad != ncclSuccess) return fail(ctx, VK_ERR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(bad));
            if (end != ncclSuccess) return fail(ctx, VK_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(end));
        }
    } else {
        NCCL_TRY(ctx, g_rccl.Send(send, bytes, ncclUint8, root, ctx->comm, st));
    }
    return VK_OK;
}

}  // extern "C"

// Single-process group: one context per GPU, communicators from ncclCommInitAll ("single-process
// ncclCommInitAll is sufficient intra-node").  Generalises the reference's tile loop (examples/xor/main.rs:235-254):
// the tiles of a frame go to N GPUs instead of N dispatches.
struct vk_group {
    std::vector<vk_ctx *> ctx;
    std::vector<void *> send;     // per rank: compact tiles of the current batch
    void *recv = nullptr;         // root: [n][slots][frames][ts][ts]
    size_t send_bytes = 0, recv_bytes = 0;
    std::vector<int> ordinals;
    bool have_comm = false;             // the communicators are made by the first gathered render: a peer-direct group never loads RCCL
    bool peer_direct = false;           // vk_group_peer_direct: the members store into the root's frames themselves
    std::vector<hipEvent_t> marched;    // ... and the root's stream waits for these
    std::string err;
};

extern "C" {

int vk_group_create(int n, const int *ordinals, vk_group **out) {
    if (!out) return fail(nullptr, VK_ERR_INVALID, "vk_group_create: out is NULL");
    *out = nullptr;
    if (n <= 0 || n > 64 || !ordinals) return fail(nullptr, VK_ERR_INVALID, "vk_group_create: 1..64 device ordinals");
    vk_group *g = new (std::nothrow) vk_group();
    if (!g) return fail(nullptr, VK_ERR_OOM, "vk_group_create: host allocation failed");
    auto bail = [&](int code, const std::string &msg) { for (vk_ctx *c : g->ctx) (void)vk_ctx_destroy(c); delete g; return fail(nullptr, code, msg); };
    for (int i = 0; i < n; i++) {
        vk_ctx *c = nullptr;
        int rc = vk_ctx_create(ordinals[i], &c);
        if (rc) return bail(rc, std::string("vk_group_create: ") + vk_last_error(nullptr));
        c->in_group = true;
        g->ctx.push_back(c);
    }
    g->ordinals.assign(ordinals, ordinals + n);
    g->send.assign(n, nullptr);
    *out = g;
    return VK_OK;
}

// ncclCommInitAll over the group's GPUs, on first use (the gathered path of vk_group_render)
static int group_ensure_comm(vk_group *g) {
    const int n = (int)g->ctx.size();
    if (g->have_comm || n <= 1) return VK_OK;
    if (!rccl_load()) { g->err = g_rccl.err; return VK_ERR_UNSUPPORTED; }
    std::vector<ncclComm_t> comms(n);
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, g->ordinals.data());
    if (r != ncclSuccess) { g->err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); return VK_ERR_HIP; }
    for (int i = 0; i < n; i++) { g->ctx[i]->comm = comms[i]; g->ctx[i]->comm_rank = i; g->ctx[i]->comm_size = n; g->ctx[i]->comm_owned = true; }
    g->have_comm = true;
    return VK_OK;
}

int vk_group_destroy(vk_group *g) {
    if (!g) return VK_ERR_INVALID;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (g->send[i]) (void)hipFree(g->send[i]);
    }
    if (g->recv) { (void)hipSetDevice(g->ctx[0]->device); (void)hipFree(g->recv); }
    for (size_t i = 0; i < g->marched.size(); i++) if (g->marched[i]) { (void)hipSetDevice(g->ctx[i]->device); (void)hipEventDestroy(g->marched[i]); }
    for (vk_ctx *c : g->ctx) (void)vk_ctx_destroy(c);
    delete g;
    return VK_OK;
}

int vk_group_peer_direct(vk_group *g, int enable) {
    if (!g) return VK_ERR_INVALID;
    if (!enable) { g->peer_direct = false; return VK_OK; }
    const int root_dev = g->ctx[0]->device;
    for (size_t i = 1; i < g->ctx.size(); i++) {
        const int dev = g->ctx[i]->device;
        if (dev == root_dev) continue;  // (several members on one GPU: the rehearsal of the tests)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev, root_dev) != hipSuccess || !can) { g->err = "vk_group_peer_direct: GPU " + std::to_string(dev) + " cannot access GPU " + std::to_string(root_dev); return VK_ERR_UNSUPPORTED; }
        if (hipSetDevice(dev) != hipSuccess) { g->err = "hipSetDevice failed"; return VK_ERR_HIP; }
        const hipError_t e = hipDeviceEnablePeerAccess(root_dev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { g->err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return VK_ERR_HIP; }
        (void)hipGetLastError();
    }
    if (g->marched.size() != g->ctx.size()) {
        g->marched.assign(g->ctx.size(), nullptr);
        for (size_t i = 0; i < g->ctx.size(); i++) {
            if (hipSetDevice(g->ctx[i]->device) != hipSuccess || hipEventCreateWithFlags(&g->marched[i], hipEventDisableTiming) != hipSuccess) { g->err = "vk_group_peer_direct: event creation failed"; return VK_ERR_HIP; }
        }
    }
    g->peer_direct = true;
    return VK_OK;
}

int vk_group_size(vk_group *g) { return g ? (int)g->ctx.size() : 0; }
vk_ctx *vk_group_ctx(vk_group *g, int i) { return (g && i >= 0 && i < (int)g->ctx.size()) ? g->ctx[i] : nullptr; }
const char *vk_group_last_error(vk_group *g) { return g ? g->err.c_str() : ""; }

// n_frames frames (cameras: n_frames x 144 bytes), each cut into tile_size^2 tiles dealt heaviest-first over the
// group's GPUs: every GPU marches its tiles of every frame in one launch, one grouped send/recv brings them to GPU 0
// over xGMI, GPU 0 un-tiles into `out_frames` ([n_frames][H][W] on GPU 0's device).  Asynchronous.
int vk_group_render(vk_group *g, int mode, uint32_t n_frames, const void *cameras, uint32_t tile_size, float dt_scale, uint32_t flags, void *out_frames) {
    if (!g || !cameras || !out_frames) return VK_ERR_INVALID;
    const int n = (int)g->ctx.size();
    vk_ctx *root = g->ctx[0];
    auto gfail = [&](vk_ctx *c, int rc) { g->err = c ? c->err : g_create_err; return rc; };
    if (n == 1) {
        int rc = vk_render_batch(root, mode, n_frames, cameras, tile_size, 0, 1, dt_scale, flags, out_frames, 0, 0, nullptr, nullptr);
        return rc ? gfail(root, rc) : VK_OK;
    }
    if (g->peer_direct) {
        // Every member marches its share of every frame straight into the root's frames (whole-frame addressing, peer memory); the root, which
        // also clears the tiles the silhouette cannot reach, waits for the others' launches on its stream.  One deal for the whole group.
        for (int i = 1; i < n; i++) g->ctx[i]->root_skip = root->root_skip;
        for (int i = 0; i < n; i++) {
            int rc = vk_render_batch(g->ctx[i], mode, n_frames, cameras, tile_size, (uint32_t)i, (uint32_t)n, dt_scale, flags, out_frames, 0, 0, nullptr, nullptr);
            if (rc) return gfail(g->ctx[i], rc);
            if (i > 0) {
                if (hipSetDevice(g->ctx[i]->device) != hipSuccess || hipEventRecord(g->marched[i], g->ctx[i]->stream) != hipSuccess) { g->err = "vk_group_render: event record failed"; return VK_ERR_HIP; }
            }
        }
        if (hipSetDevice(root->device) != hipSuccess) { g->err = "hipSetDevice failed"; return VK_ERR_HIP; }
        for (int i = 1; i < n; i++)
            if (hipStreamWaitEvent(root->stream, g->marched[i], 0) != hipSuccess) { g->err = "vk_group_render: stream wait failed"; return VK_ERR_HIP; }
        return VK_OK;
    }
    { int rc = group_ensure_comm(g); if (rc) return rc; }
    uint32_t cap = 0;
    if (vk_partition_slots_weighted(root->width, root->height, tile_size, (uint32_t)n, root->root_skip, &cap) != VK_OK) { g->err = "vk_group_render: bad tile size"; return VK_ERR_INVALID; }
    for (int i = 1; i < n; i++) { g->ctx[i]->root_skip = root->root_skip; g->ctx[i]->wire = root->wire; }  // one deal, one wire format for the whole group
    const size_t tile_bytes = (size_t)tile_size * tile_size * wire_px_bytes(root->out_format, root->wire);
    const size_t need = (size_t)cap * n_frames * tile_bytes;
    if (g->send_bytes < need) {
        for (int i = 0; i < n; i++) {
            if (hipSetDevice(g->ctx[i]->device) != hipSuccess) { g->err = "hipSetDevice failed"; return VK_ERR_HIP; }
            (void)hipStreamSynchronize(g->ctx[i]->stream);
            if (g->send[i]) (void)hipFree(g->send[i]);
            g->send[i] = nullptr;
            if (hipMalloc(&g->send[i], need) != hipSuccess) { g->send_bytes = 0; g->err = "vk_group_render: tile buffer allocation failed"; return VK_ERR_OOM; }
        }
        (void)hipSetDevice(root->device);
        if (g->recv) (void)hipFree(g->recv);
        g->recv = nullptr;
        if (hipMalloc(&g->recv, need * n) != hipSuccess) { g->send_bytes = 0; g->err = "vk_group_render: gather buffer allocation failed"; return VK_ERR_OOM; }
        g->send_bytes = need; g->recv_bytes = need * n;
    }
    uint32_t bid0 = 0, act = 0;
    for (int i = 0; i < n; i++) {
        uint32_t bid = 0, a = 0;
        int rc = vk_render_batch(g->ctx[i], mode, n_frames, cameras, tile_size, (uint32_t)i, (uint32_t)n, dt_scale, flags, g->send[i], 1, cap, &bid, &a);
        if (rc) return gfail(g->ctx[i], rc);
        if (i == 0) { bid0 = bid; act = a; }
    }
    const size_t bytes = (size_t)act * n_frames * tile_bytes;  // what each rank sends: its leading active slots
    if (bytes) {
        if (hipSetDevice(root->device) != hipSuccess) { g->err = "hipSetDevice failed"; return VK_ERR_HIP; }
        if (hipMemcpyAsync(g->recv, g->send[0], bytes, hipMemcpyDeviceToDevice, root->stream) != hipSuccess) { g->err = "tile copy failed"; return VK_ERR_HIP; }
        ncclResult_t r = g_rccl.GroupStart();
        for (int p = 1; p < n && r == ncclSuccess; p++) {
            r = g_rccl.Recv(static_cast<unsigned char *>(g->recv) + (size_t)p * bytes, bytes, ncclUint8, p, root->comm, root->stream);
            if (r == ncclSuccess) r = g_rccl.Send(g->send[p], bytes, ncclUint8, 0, g->ctx[p]->comm, g->ctx[p]->stream);
        }
        ncclResult_t e = g_rccl.GroupEnd();
        if (r != ncclSuccess || e != ncclSuccess) { g->err = std::string("vk_group_render: RCCL: ") + g_rccl.GetErrorString(r != ncclSuccess ? r : e); return VK_ERR_HIP; }
    }
    int rc = vk_untile_batch(root, bid0, g->recv, act, out_frames);
    return rc ? gfail(root, rc) : VK_OK;
}

int vk_group_sync(vk_group *g) {
    if (!g) return VK_ERR_INVALID;
    for (vk_ctx *c : g->ctx) { int rc = vk_ctx_sync(c); if (rc) { g->err = c->err; return rc; } }
    return VK_OK;
}

}  // extern "C"
