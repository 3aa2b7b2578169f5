// vk_api.hip -- C-ABI (include/vokselis_hip.h) over the kernels in vk_kernels.hpp.
// gfx950 only; no CPU fallback: every entry point fails with VK_ERR_HIP / VK_ERR_NO_DEVICE
// when the HIP runtime or the device is unavailable.
#include "../../include/vokselis_hip.h"
#include "vk_kernels.hpp"
#include "vk_staged.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use (vk_comm_*, vk_group_*)

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace vk;

struct vk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop{};

    // volume
    void *vol = nullptr, *vol2 = nullptr;  // cells / dense voxels / pair
    void *scopy[3] = {nullptr, nullptr, nullptr};  // VK_LAYOUT_STAGED: one brick copy per slow axis
    StagedDesc sdesc{};
    uint32_t stage_cap_bytes = 0, stage_slab_cells = 0, stage_copies_mask = 7;  // tunables (vk_debug_set_param)
    uint32_t stage_group = 2;    // staged march: one LDS window per 256-thread group of four waves (vk_staged.hpp: raymarch_staged_group_kernel): 0 never, 1 always, 2 where it pays (launch_staged)
    uint32_t frame_runs = 1;     // batched launches: every XCD marches a run of consecutive frames of a tile position (0: frames x, x + 8, ... as in round 2)
    uint32_t stage_grow_every = 0;  // slab search: try one cell above the last fit every n-th round (0: 4 for u8, 1 for f16; tools/staged_grow.py)
    uint32_t stage_row_pad = 2;  // odd LDS row pitch of the staged window: 0 never, 1 always, 2 (default) with group windows on u8 volumes (launch_staged)
    uint8_t *dist = nullptr;
    uint32_t *lut = nullptr;  // per-axis cell-index tables (cell units | byte offsets), cell layouts only
    size_t vol_bytes = 0;
    uint32_t nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0;
    int format = -1, layout = 0;
    int vol_kind = -1;  // vk::VolKind
    double empty_fraction = 0.0;  // share of cells that are exactly transparent (packed layouts)
    VolumeDesc vdesc{};

    // uniforms (host copies; passed to kernels by value)
    unsigned char uniform[48] = {0};
    float camera[36] = {0};  // 144-byte CameraUniform
    bool have_camera = false;

    // output
    void *backbuffer = nullptr;
    uint32_t width = 0, height = 0;
    int out_format = VK_OUT_RGBA32F;
    uint32_t *steps = nullptr;
    unsigned long long *counters = nullptr;

    // heaviest-first tile order (launch-order heuristic; see tile_order_update)
    std::vector<uint32_t> order, order_pos;
    uint32_t order_active = 0;  // leading positions of `order` whose tiles can contain non-clear pixels
    std::vector<unsigned char> order_key;
    unsigned long long *trace = nullptr;
    size_t trace_blocks = 0;
    int wire = VK_WIRE_RGBA;     // compact tiles of a partition: whole pixels, or colour only (vk_partition_wire)
    bool want_trace = false;
    uint32_t trip_log_cap = 0;   // > 0: the trace buffer holds per-trip logs of that many u32 entries per wave instead of stamps (tools/repack_census.py)
    // The device copies of (order, order_pos) live in a ring of kOrderRing slots fed from pinned staging: a new
    // camera takes the next slot with one stream-ordered copy -- no host or device synchronisation -- while
    // launches still in flight (other streams: frames in flight) keep reading the slots they were given.
    uint32_t *d_order = nullptr, *d_order_pos = nullptr;  // the current slot
    uint32_t *d_ring = nullptr, *h_ring = nullptr;
    size_t d_order_cap = 0;      // entries per table in every slot
    int ring_slot = -1;
    hipEvent_t ring_ev[16] = {};     // slot uploaded
    hipStream_t ring_stream[16] = {};
    bool ring_done[16] = {};
    uint32_t ring_active[16] = {};   // order_active of the slot
    uint32_t order_seq = 0;          // order changes so far; change e lives in slot e % 16

    // batched launches (vk_render_batch): per-batch tables {FrameDesc[B], order[B][n_tiles], pos[B][n_tiles]} in a small
    // ring of device slots fed from pinned staging; a slot is rewritten only after the last kernel that read it
    struct BatchSlot {
        unsigned char *d = nullptr, *h = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        uint32_t id = 0, n_frames = 0, n_tiles = 0, ts = 0, nranks = 0, max_active = 0, root_skip = 0;
        uint32_t width = 0, height = 0;  // the frame shape and pixel format the batch was dealt for: an un-tile under another
        int out_format = -1;             // shape (vk_backbuffer_resize in between) would scatter tiles out of bounds
        int wire = 0;                    // ... and the wire format its compact tiles were written in
    } batch[4];
    uint32_t batch_seq = 0;
    // table blocks a growing batch has outgrown: hipFree / hipHostFree synchronise the device, so they wait here for a
    // call that synchronises anyway (vk_backbuffer_resize, vk_ctx_destroy) instead of stalling four launches in flight
    std::vector<std::pair<void *, void *>> batch_retired;
    // the order of the last camera a batch computed one for (a still camera costs no host work from batch to batch)
    std::vector<unsigned char> batch_key;
    std::vector<uint32_t> batch_order, batch_pos;
    uint32_t batch_n_active = 0;
    // Skip kernels: steps a walk may take in a trip in which other lanes sample / in which every lane walks (0: no cap).
    // tools/walk_cap_sweep.py: 8 / 12 -- C2 0.0806 -> 0.0728 ms per frame in batches, 0.1625 -> 0.1555 single; a fog with
    // 80 % of its 16^3 blocks knocked out 0.161 -> 0.135.  Multiples of the walk loop's four steps do best.
    uint32_t walk_cap = 8, walk_cap_all = 12;
    uint32_t order_rays = 3;     // estimate rays per tile edge of the heaviest-first order (single-frame launches)
    uint32_t order_rays_batch = 1;  // ... of launches spanning >= 4 frames
    uint32_t wave_prio = 1;      // issue priority by ray length (set_wave_priority); 0 for A/B measurements
    uint32_t naive_lds_pad = 0;  // debug: extra dynamic LDS per workgroup of the cell kernels (caps the waves per SIMD)
    uint32_t root_skip = 0;  // dealing: rank 0 sits out every root_skip-th round (vk_partition_root_skip)

    // present targets (next row N1/N2)
    uint32_t *rgba8 = nullptr, *bgra8 = nullptr;
    uint32_t present_w = 0, present_h = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_open = false, timing_done = false;

    // multi-GPU: this context's RCCL communicator (vk_comm_init_rank / vk_group_create)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    bool comm_owned = false;
    bool in_group = false;  // a member of a vk_group: its communicator belongs to the group's world
};

static thread_local std::string g_create_err;

static int fail(vk_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
        }                                                                                      \
    } while (0)

static size_t px_bytes(int fmt) { return fmt == VK_OUT_RGBA16F ? 8 : 16; }
// a pixel of a partition's compact tiles: the backbuffer's pixel, or its three colour channels (VK_WIRE_RGB)
static size_t wire_px_bytes(int fmt, int wire) { return wire == VK_WIRE_RGB ? px_bytes(fmt) / 4 * 3 : px_bytes(fmt); }

// ---- RCCL, loaded on first use ---------------------------------------------------------------------
// librccl is 570 MB; a single-GPU user never pays for it.  dlopen by SONAME finds the copy a host process
// already carries (PyTorch-ROCm bundles one), so a process never holds two.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd ||
        !g_rccl.Send || !g_rccl.Recv || !g_rccl.GetErrorString) { dlclose(h); return false; }
    g_rccl.lib = h;
    return true;
}

#define NCCL_TRY(ctx, expr)                                                                                        \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(ctx, VK_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

static void comm_release(vk_ctx *ctx);

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

uint32_t vk_dispatch_optimal(uint32_t len, uint32_t subgroup_size) {
    if (subgroup_size == 0) return 0;
    uint32_t padded = (subgroup_size - len % subgroup_size) % subgroup_size;
    return (len + padded) / subgroup_size;
}

const char *vk_last_error(vk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int vk_ctx_create(int device_ordinal, vk_ctx **out) {
    if (!out) return fail(nullptr, VK_ERR_INVALID, "vk_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, VK_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, VK_ERR_INVALID, "vk_ctx_create: device ordinal out of range");
    vk_ctx *ctx = new (std::nothrow) vk_ctx();
    if (!ctx) return fail(nullptr, VK_ERR_OOM, "vk_ctx_create: host allocation failed");
    ctx->device = device_ordinal;
    auto bail = [&](hipError_t err, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        delete ctx;
        return fail(nullptr, VK_ERR_HIP, msg);
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipGetDeviceProperties(&ctx->prop, device_ordinal)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string msg = std::string("device is ") + ctx->prop.gcnArchName + ", this library carries gfx950 code only";
        delete ctx;
        return fail(nullptr, VK_ERR_NO_DEVICE, msg);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&ctx->counters, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemset(ctx->counters, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(counters)");
    *out = ctx;
    return VK_OK;
}

static void comm_release(vk_ctx *ctx) {
    if (ctx->comm && ctx->comm_owned && g_rccl.lib) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_size = 0; ctx->comm_rank = 0; ctx->comm_owned = false;
}

static void free_volume(vk_ctx *ctx) {
    if (ctx->vol) (void)hipFree(ctx->vol);
    if (ctx->vol2) (void)hipFree(ctx->vol2);
    if (ctx->dist) (void)hipFree(ctx->dist);
    if (ctx->lut) (void)hipFree(ctx->lut);
    for (void *&c : ctx->scopy) { if (c) (void)hipFree(c); c = nullptr; }
    ctx->vol = ctx->vol2 = nullptr;
    ctx->dist = nullptr;
    ctx->lut = nullptr;
    ctx->vol_bytes = 0;
    ctx->format = -1;
    ctx->vol_kind = -1;
    ctx->empty_fraction = 0.0;
}

int vk_ctx_destroy(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    free_volume(ctx);
    if (ctx->backbuffer) (void)hipFree(ctx->backbuffer);
    if (ctx->steps) (void)hipFree(ctx->steps);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->rgba8) (void)hipFree(ctx->rgba8);
    if (ctx->bgra8) (void)hipFree(ctx->bgra8);
    if (ctx->trace) (void)hipFree(ctx->trace);
    if (ctx->d_ring) (void)hipFree(ctx->d_ring);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    for (auto &b : ctx->batch) { if (b.d) (void)hipFree(b.d); if (b.h) (void)hipHostFree(b.h); if (b.ev) (void)hipEventDestroy(b.ev); }
    for (auto &r : ctx->batch_retired) { (void)hipFree(r.first); (void)hipHostFree(r.second); }
    for (hipEvent_t e : ctx->ring_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VK_OK;
}

int vk_ctx_set_stream(vk_ctx *ctx, void *hip_stream) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return VK_OK;
}

int vk_ctx_sync(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_device_info(vk_ctx *ctx, char *name, size_t name_cap, int *compute_units, int *arch_is_gfx950,
                   size_t *total_mem_bytes) {
    if (!ctx) return VK_ERR_INVALID;
    if (name && name_cap) std::snprintf(name, name_cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (arch_is_gfx950) *arch_is_gfx950 = std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) == 0;
    if (total_mem_bytes) *total_mem_bytes = ctx->prop.totalGlobalMem;
    return VK_OK;
}

// ---- volume ------------------------------------------------------------------------------------

// Everything a volume owns on the device.  build_from_dense() fills a local one and the context adopts it only
// when every allocation and kernel has succeeded, so a failed upload leaves the previous volume (or "no volume")
// intact and never a half-built one for the next vk_render to dereference.
struct VolBuild {
    void *vol = nullptr, *vol2 = nullptr;
    uint8_t *dist = nullptr;
    uint32_t *lut = nullptr;
    void *scopy[3] = {nullptr, nullptr, nullptr};
    size_t vol_bytes = 0;
    uint32_t nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0;
    int format = -1, layout = 0, vol_kind = -1;
    double empty_fraction = 0.0;
    VolumeDesc vdesc{};
    StagedDesc sdesc{};
    bool committed = false;
    ~VolBuild() {
        if (committed) return;
        (void)hipFree(vol); (void)hipFree(vol2); (void)hipFree(dist); (void)hipFree(lut);
        for (void *c : scopy) (void)hipFree(c);
    }
};

// The adopted dense source (vk_volume_upload / vk_volume_generate): freed on every exit unless a layout keeps it.
struct DenseSource {
    const void *p = nullptr, *p2 = nullptr;
    bool owned = false;
    ~DenseSource() { if (owned) { (void)hipFree(const_cast<void *>(p)); (void)hipFree(const_cast<void *>(p2)); } }
    void release() { owned = false; }
};

static void free_volume(vk_ctx *ctx);

static int commit_volume(vk_ctx *ctx, VolBuild &nb) {
    free_volume(ctx);
    for (auto &b : ctx->batch) b.id = 0;  // batches dealt for the previous volume are no longer un-tiled
    ctx->batch_key.clear();
    ctx->vol = nb.vol; ctx->vol2 = nb.vol2; ctx->dist = nb.dist; ctx->lut = nb.lut;
    for (int k = 0; k < 3; k++) ctx->scopy[k] = nb.scopy[k];
    ctx->vol_bytes = nb.vol_bytes;
    ctx->nx = nb.nx; ctx->ny = nb.ny; ctx->nz = nb.nz; ctx->nbx = nb.nbx; ctx->nby = nb.nby; ctx->nbz = nb.nbz;
    ctx->format = nb.format; ctx->layout = nb.layout; ctx->vol_kind = nb.vol_kind;
    ctx->empty_fraction = nb.empty_fraction;
    ctx->vdesc = nb.vdesc;
    ctx->sdesc = nb.sdesc;
    nb.committed = true;
    return VK_OK;
}

static int build_from_dense(vk_ctx *ctx, const void *d_src, const void *d_src2, bool own_src, uint32_t nx,
                            uint32_t ny, uint32_t nz, int format, int layout) {
    // d_src is dense device memory; with own_src the LINEAR layout adopts it, every other layout frees it.
    DenseSource src;
    src.p = d_src; src.p2 = d_src2; src.owned = own_src;
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : (format == VK_FMT_R16_FLOAT ? 2 : 8);
    if (layout == VK_LAYOUT_AUTO) {
        const double cells = ((double)((nx - 1) / 4 + 2)) * ((ny - 1) / 4 + 2) * ((nz - 1) / 4 + 2) * 64.0;
        const double cell_bytes = cells * 16.0;
        if (format == VK_FMT_RGBA16F_PAIR) {
            // 16-byte (density, normals) records in 4^3 bricks while the array and its index tables stay small
            const double rec_bytes = ((double)((nx + 3) / 4)) * ((ny + 3) / 4) * ((nz + 3) / 4) * 64.0 * 16.0;
            layout = (rec_bytes <= (double)kPairOob && pair_lut_entries(nx, ny, nz) * 4u <= 16384u) ? VK_LAYOUT_PACKED : VK_LAYOUT_LINEAR;
        }
        // beyond ~4 GiB of cells the march stops being cache-resident and the 8-16x inflation of the cell layouts
        // turns into HBM traffic: dense 8^3 bricks staged through LDS win there
        else if (cell_bytes > 4.0 * 1024 * 1024 * 1024) layout = VK_LAYOUT_STAGED;
        // u8: the (tap, delta) pair cells cost 2x the bytes and ~20 % fewer VALU ops per sample
        else layout = format == VK_FMT_R8_UNORM ? VK_LAYOUT_PACKED_PAIRS : VK_LAYOUT_PACKED;
    }
    if (format == VK_FMT_RGBA16F_PAIR && layout != VK_LAYOUT_LINEAR && layout != VK_LAYOUT_PACKED)
        return fail(ctx, VK_ERR_UNSUPPORTED, "RGBA16F_PAIR volumes use VK_LAYOUT_LINEAR or VK_LAYOUT_PACKED (bricked 16-byte records)");
    if (format == VK_FMT_RGBA16F_PAIR && layout == VK_LAYOUT_PACKED) {
        const double rec_bytes = ((double)((nx + 3) / 4)) * ((ny + 3) / 4) * ((nz + 3) / 4) * 64.0 * 16.0;
        if (rec_bytes > (double)kPairOob || pair_lut_entries(nx, ny, nz) * 4u > 16384u)
            return fail(ctx, VK_ERR_UNSUPPORTED, "RGBA16F_PAIR record layout holds <= 1.25 GiB of records: use VK_LAYOUT_LINEAR");
    }
    VolBuild nb;
    nb.nx = nx; nb.ny = ny; nb.nz = nz;
    nb.format = format;
    nb.layout = layout;
    // kernels launched below, then one synchronisation; the message names the stage that failed
    auto finish = [&](const char *what) -> int {
        hipError_t le = hipGetLastError();
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (le != hipSuccess || se != hipSuccess) return fail(ctx, VK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(le != hipSuccess ? le : se));
        return VK_OK;
    };
    auto alloc = [&](void **p, size_t bytes, const char *what) -> int {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) { *p = nullptr; return fail(ctx, e == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP, std::string(what) + " allocation failed: " + hipGetErrorString(e)); }
        return VK_OK;
    };
    int rc;
    if (layout == VK_LAYOUT_LINEAR) {
        if (own_src) {
            nb.vol = const_cast<void *>(d_src);
            nb.vol2 = const_cast<void *>(d_src2);
            src.release();
        } else {
            if ((rc = alloc(&nb.vol, n_vox * bpv, "dense volume"))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(nb.vol, d_src, n_vox * bpv, hipMemcpyDeviceToDevice, ctx->stream));
            if (d_src2) {
                if ((rc = alloc(&nb.vol2, n_vox * bpv, "dense volume (normals)"))) return rc;
                HIP_TRY(ctx, hipMemcpyAsync(nb.vol2, d_src2, n_vox * bpv, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        nb.vol_bytes = n_vox * bpv * (d_src2 ? 2 : 1);
        nb.vol_kind = format == VK_FMT_R16_FLOAT ? VOL_LINEAR_F16 : VOL_LINEAR_U8;
        if ((rc = finish("dense copy"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (format == VK_FMT_RGBA16F_PAIR) {  // layout == VK_LAYOUT_PACKED: interleaved records, 4^3 bricks
        nb.nbx = (nx + 3) / 4; nb.nby = (ny + 3) / 4; nb.nbz = (nz + 3) / 4;
        const uint64_t n_rec = (uint64_t)nb.nbx * nb.nby * nb.nbz * 64u;
        if ((rc = alloc(&nb.vol, n_rec * 16, "record array"))) return rc;
        const uint32_t padded = pair_lut_entries(nx, ny, nz);
        if ((rc = alloc((void **)&nb.lut, (size_t)padded * sizeof(uint32_t), "index table"))) return rc;
        nb.vol_bytes = n_rec * 16;
        nb.vol_kind = VOL_PAIRB;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_rec + 255) / 256, 1ull << 22);
        hipLaunchKernelGGL(pack_pairs_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint2 *)d_src, (const uint2 *)d_src2, (uint4 *)nb.vol, nx, ny, nz,
                           nb.nbx, nb.nby, n_rec);
        hipLaunchKernelGGL(build_pair_luts_kernel, dim3((padded + 255) / 256), dim3(256), 0, ctx->stream, nb.lut, nx, ny, nz, nb.nbx, nb.nby);
        if ((rc = finish("record re-layout"))) return rc;
        nb.vdesc.max_off = (int64_t)(n_rec - 1) * 16;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_STAGED) {
        // dense 8^3 bricks without apron behind a replicated border of kStagePad voxels, one copy per SLOW axis
        // (vk_staged.hpp): copy k = (slow k, fast (k+1)%3, mid (k+2)%3)
        const bool u8 = format == VK_FMT_R8_UNORM;
        const uint32_t n[3] = {nx, ny, nz};
        StagedDesc &D = nb.sdesc;
        for (int a = 0; a < 3; a++) D.nv[a] = 8u * ((n[a] + 7u) / 8u + 2u);
        uint32_t mask = ctx->stage_copies_mask & 7u;
        if (mask == 0) mask = 7u;
        const uint32_t vpp = u8 ? 16u : 8u;
        for (int k = 0; k < 3; k++) {
            if (!(mask & (1u << k))) continue;
            const int F = (k + 1) % 3, M = (k + 2) % 3, S = k;
            D.npf[k] = (D.nv[F] + vpp - 1) / vpp;
            D.nbm[k] = D.nv[M] / 8u;
            const uint64_t n_bricks = (uint64_t)D.npf[k] * D.nbm[k] * (D.nv[S] / 8u);
            if (n_bricks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for 32-bit brick indices");
            const uint64_t n_pieces = n_bricks * 64u;
            if ((rc = alloc(&nb.scopy[k], n_pieces * 16, "staged brick copy"))) return rc;
            D.copy[k] = (const unsigned char *)nb.scopy[k];
            nb.vol_bytes += n_pieces * 16;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_pieces + 255) / 256, 1ull << 22);  // grid-stride kernel
            if (u8) hipLaunchKernelGGL(pack_staged_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, d_src, (uint4 *)nb.scopy[k], nx, ny, nz, k, D.npf[k], D.nbm[k], n_pieces);
            else hipLaunchKernelGGL(pack_staged_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_src, (uint4 *)nb.scopy[k], nx, ny, nz, k, D.npf[k], D.nbm[k], n_pieces);
        }
        // a wave takes the copy whose SLOW axis is its rays' major axis; without it, the copy whose MID axis is
        // (lines are then partly used), else whatever exists
        for (int a = 0; a < 3; a++) {
            const int pref[3] = {a, (a + 1) % 3, (a + 2) % 3};  // slow == a; mid == a; fast == a
            for (int j = 2; j >= 0; j--) if (mask & (1u << pref[j])) D.copy_of_major[a] = (uint32_t)pref[j];
        }
        nb.vol_kind = u8 ? VOL_S8U8 : VOL_S8F16;
        if ((rc = finish("staged brick re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_QUADS) {
        // quad elements in 9x8x8 bricks: padded coordinate c = i + 1 in [0, n] -> (n >> 3) + 1 bricks per axis
        const bool f16q = format == VK_FMT_R16_FLOAT;
        nb.nbx = (nx >> 3) + 1; nb.nby = (ny >> 3) + 1; nb.nbz = (nz >> 3) + 1;
        const uint64_t n_bricksq = (uint64_t)nb.nbx * nb.nby * nb.nbz;
        const uint64_t n_elems = n_bricksq * 576u;
        const size_t ebytes = f16q ? 8 : 4;
        if (n_bricksq >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
        if ((rc = alloc(&nb.vol, n_elems * ebytes + 16, "quad layout (4.5x the dense bytes; VK_LAYOUT_STAGED is 1x per copy)"))) return rc;  // + slack: a sample reads two elements
        nb.vol_bytes = n_elems * ebytes;
        nb.vol_kind = f16q ? VOL_QF16 : VOL_Q8;
        const uint32_t blocksq = (uint32_t)std::min<uint64_t>((n_elems + 255) / 256, 1ull << 22);  // grid-stride kernel
        if (f16q) hipLaunchKernelGGL(pack_quads_kernel<true>, dim3(blocksq), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        else hipLaunchKernelGGL(pack_quads_kernel<false>, dim3(blocksq), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        if ((rc = finish("quad re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_BRICKED) {
        // dense 9^3 bricks: brick b holds voxels [8b-1, 8b+7]; cell coords go up to n -> (n >> 3) + 1 bricks
        const bool f16b = format == VK_FMT_R16_FLOAT;
        nb.nbx = (nx >> 3) + 1; nb.nby = (ny >> 3) + 1; nb.nbz = (nz >> 3) + 1;
        const uint64_t n_bricks9 = (uint64_t)nb.nbx * nb.nby * nb.nbz;
        const uint64_t n_elems = n_bricks9 * 729u;
        if (n_bricks9 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
        if ((rc = alloc(&nb.vol, n_elems * bpv + 16, "9^3 brick array"))) return rc;  // + slack: the last tap pair reads 2 elements
        nb.vol_bytes = n_elems * bpv;
        nb.vol_kind = f16b ? VOL_B9F16 : VOL_B9U8;
        const uint32_t blocks9 = (uint32_t)std::min<uint64_t>((n_elems + 255) / 256, 1ull << 22);  // grid-stride kernel
        if (f16b) hipLaunchKernelGGL(pack_bricks9_kernel<true>, dim3(blocks9), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        else hipLaunchKernelGGL(pack_bricks9_kernel<false>, dim3(blocks9), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        if ((rc = finish("brick re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    // PACKED: cells for low-corner voxels i in [-1, n-1]; physical brick (i >> 2) + 1
    const bool f16 = format == VK_FMT_R16_FLOAT;
    if (f16 && layout == VK_LAYOUT_PACKED_PAIRS)
        return fail(ctx, VK_ERR_UNSUPPORTED, "PACKED_PAIRS stores exact u8 differences; f16 volumes use PACKED");
    const int kind = f16 ? VOL_PF16 : (layout == VK_LAYOUT_PACKED_PAIRS ? VOL_P16 : VOL_P8);
    nb.nbx = ((nx - 1) >> 2) + 2;
    nb.nby = ((ny - 1) >> 2) + 2;
    nb.nbz = ((nz - 1) >> 2) + 2;
    const uint64_t n_bricks = (uint64_t)nb.nbx * nb.nby * nb.nbz;
    const uint64_t n_cells = n_bricks * kBrickCells;
    const size_t cell_bytes = kind == VOL_P8 ? 8 : 16;
    if (n_bricks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for 32-bit brick indices");
    if (n_cells >= (1ull << 32)) return fail(ctx, VK_ERR_UNSUPPORTED, "cell layouts hold < 2^32 cells (about 1600^3): use VK_LAYOUT_STAGED or VK_LAYOUT_AUTO");
    const uint64_t pack_blocks64 = (n_cells + 255) / 256;
    if (pack_blocks64 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for one launch");
    const uint32_t pack_blocks = (uint32_t)pack_blocks64;
    if ((rc = alloc(&nb.vol, n_cells * cell_bytes, "cell array"))) return rc;
    // scratch: occupancy map + two pass buffers
    struct Scratch {
        uint8_t *occ = nullptr, *tx = nullptr, *txy = nullptr;
        ~Scratch() { (void)hipFree(occ); (void)hipFree(tx); (void)hipFree(txy); }
    } sc;
    if ((rc = alloc((void **)&sc.occ, n_cells, "re-layout scratch")) || (rc = alloc((void **)&sc.tx, n_cells, "re-layout scratch")) ||
        (rc = alloc((void **)&sc.txy, n_cells, "re-layout scratch")))
        return rc;
    nb.vol_kind = kind;
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters + 7, 0, sizeof(unsigned long long), ctx->stream));
    if (kind == VOL_PF16)
        hipLaunchKernelGGL(pack_cells_kernel<VOL_PF16>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    else if (kind == VOL_P16)
        hipLaunchKernelGGL(pack_cells_kernel<VOL_P16>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    else
        hipLaunchKernelGGL(pack_cells_kernel<VOL_P8>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    {
        unsigned long long ne = 0;
        hipError_t le = hipGetLastError();
        hipError_t ce = hipMemcpyAsync(&ne, ctx->counters + 7, sizeof(ne), hipMemcpyDeviceToHost, ctx->stream);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (le != hipSuccess || ce != hipSuccess || se != hipSuccess)
            return fail(ctx, VK_ERR_HIP, std::string("volume re-layout: ") + hipGetErrorString(le != hipSuccess ? le : (ce != hipSuccess ? ce : se)));
        nb.empty_fraction = (double)ne / (double)n_cells;
    }
    // Distance maps.  Eight one-sided maps (one per ray octant) when skipping will be on by default
    // or may well be forced on (>= 30 % empty cells) and they stay <= 2 GiB; otherwise one isotropic map serves every octant.
    const bool octants = nb.empty_fraction >= 0.30 && n_cells <= (1ull << 28);  // (the default policy skips from 45 %)
    const uint64_t dist_bytes = octants ? 8 * n_cells : n_cells;
    if ((rc = alloc((void **)&nb.dist, dist_bytes, "distance map"))) return rc;
    nb.vol_bytes = n_cells * cell_bytes + dist_bytes;
    nb.vdesc.dist_oct_stride = octants ? (uint32_t)n_cells : 0u;
    auto pass = [&](const uint8_t *in, uint8_t *out, int axis, int dir, int last) {
        hipLaunchKernelGGL(dist_pass_kernel, dim3(pack_blocks), dim3(256), 0, ctx->stream, in, out, nb.nbx, nb.nby, nb.nbz, axis, dir, last);
    };
    if (octants) {
        for (int ux = 0; ux < 2; ux++) {
            pass(sc.occ, sc.tx, 0, ux ? 1 : -1, 0);
            for (int uy = 0; uy < 2; uy++) {
                pass(sc.tx, sc.txy, 1, uy ? 1 : -1, 0);
                for (int uz = 0; uz < 2; uz++) pass(sc.txy, nb.dist + (size_t)(ux | (uy << 1) | (uz << 2)) * n_cells, 2, uz ? 1 : -1, 1);
            }
        }
    } else {
        pass(sc.occ, sc.tx, 0, 0, 0);
        pass(sc.tx, sc.txy, 1, 0, 0);
        pass(sc.txy, nb.dist, 2, 0, 1);
    }
    if ((rc = finish("distance maps"))) return rc;
    {
        // per-axis cell-index tables of the fast path, two copies: cell units, byte offsets
        const uint32_t padded = cell_lut_entries(nx, ny, nz);
        if ((rc = alloc((void **)&nb.lut, (size_t)padded * 2 * sizeof(uint32_t), "cell-index table"))) return rc;
        hipLaunchKernelGGL(build_cell_luts_kernel, dim3((padded + 255) / 256), dim3(256), 0, ctx->stream, nb.lut, nx, ny, nz, nb.nbx, nb.nby,
                           (uint32_t)(cell_bytes == 8 ? 3 : 4));
        if ((rc = finish("cell-index tables"))) return rc;
    }
    // addressing constants (vk_kernels.hpp: VolumeDesc)
    VolumeDesc &V = nb.vdesc;
    const int64_t cb = (int64_t)cell_bytes, bxn = nb.nbx, bxyn = (int64_t)nb.nbx * nb.nby;
    V.sh_x = cell_bytes == 8 ? 3 : 4;
    V.sh_y = V.sh_x + 2;
    V.sh_z = V.sh_x + 4;
    V.kx = (int32_t)(60 * cb);
    V.ky = (int32_t)((64 * bxn - 16) * cb);
    V.kz = (64 * bxyn - 64) * cb;
    V.c0 = (64 * bxyn + 64 * bxn + 64) * cb;
    V.max_off = (int64_t)(n_cells - 1) * cb;
    return commit_volume(ctx, nb);
}

static int check_volume_args(vk_ctx *ctx, const void *p, const void *p2, uint32_t nx, uint32_t ny, uint32_t nz, int format,
                             int layout) {
    if (!ctx) return VK_ERR_INVALID;
    if (!p) return fail(ctx, VK_ERR_INVALID, "volume pointer is NULL");
    if (nx == 0 || ny == 0 || nz == 0 || nx > 8192 || ny > 8192 || nz > 8192)
        return fail(ctx, VK_ERR_INVALID, "volume dims must be in [1, 8192]");
    if (format < VK_FMT_R8_UNORM || format > VK_FMT_RGBA16F_PAIR) return fail(ctx, VK_ERR_INVALID, "unknown volume format");
    if (format == VK_FMT_RGBA16F_PAIR && !p2) return fail(ctx, VK_ERR_INVALID, "RGBA16F_PAIR needs the normals volume");
    if (layout < VK_LAYOUT_AUTO || layout > VK_LAYOUT_STAGED) return fail(ctx, VK_ERR_INVALID, "unknown layout");
    return VK_OK;
}

int vk_volume_upload(vk_ctx *ctx, const void *host, const void *host2, uint32_t nx, uint32_t ny, uint32_t nz,
                     int format, int layout) {
    int rc = check_volume_args(ctx, host, host2, nx, ny, nz, format, layout);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : (format == VK_FMT_R16_FLOAT ? 2 : 8);
    void *d = nullptr, *d2 = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * bpv));
    hipError_t e = hipMemcpy(d, host, n_vox * bpv, hipMemcpyHostToDevice);
    if (e == hipSuccess && format == VK_FMT_RGBA16F_PAIR) {
        e = hipMalloc(&d2, n_vox * bpv);
        if (e == hipSuccess) e = hipMemcpy(d2, host2, n_vox * bpv, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(d);
        if (d2) (void)hipFree(d2);
        return fail(ctx, VK_ERR_HIP, std::string("volume upload: ") + hipGetErrorString(e));
    }
    return build_from_dense(ctx, d, d2, true, nx, ny, nz, format, layout);
}

int vk_volume_upload_device(vk_ctx *ctx, const void *dev, const void *dev2, uint32_t nx, uint32_t ny, uint32_t nz,
                            int format, int layout) {
    int rc = check_volume_args(ctx, dev, dev2, nx, ny, nz, format, layout);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return build_from_dense(ctx, dev, dev2, false, nx, ny, nz, format, layout);
}

int vk_volume_generate(vk_ctx *ctx, int kind, uint32_t nx, uint32_t ny, uint32_t nz, int format, uint32_t seed,
                       uint32_t lo, uint32_t span, int layout) {
    int dummy = 0;
    int rc = check_volume_args(ctx, &dummy, nullptr, nx, ny, nz, format, layout);
    if (rc) return rc;
    if (kind != VK_GEN_FOG && kind != VK_GEN_BONSAI_STANDIN && kind != VK_GEN_FOG_DENSE_CORE) return fail(ctx, VK_ERR_INVALID, "unknown generator kind");
    const uint32_t core = kind == VK_GEN_FOG_DENSE_CORE ? 1u : 0u;
    if (core) kind = VK_GEN_FOG;
    if (format == VK_FMT_RGBA16F_PAIR) return fail(ctx, VK_ERR_UNSUPPORTED, "generators make scalar volumes");
    if (kind == VK_GEN_BONSAI_STANDIN && format != VK_FMT_R8_UNORM) return fail(ctx, VK_ERR_UNSUPPORTED, "the bonsai stand-in is a u8 volume");
    if (kind == VK_GEN_FOG && format == VK_FMT_R8_UNORM && (span == 0 || lo + span > 256)) return fail(ctx, VK_ERR_INVALID, "fog range outside u8");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : 2;
    void *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * bpv));
    // grid-stride kernels: a launch may not exceed 2^32 threads
    const uint64_t blocks = std::min<uint64_t>((n_vox + 255) / 256, 1ull << 22);
    if (kind == VK_GEN_BONSAI_STANDIN)
        hipLaunchKernelGGL(generate_kernel<2>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    else if (format == VK_FMT_R16_FLOAT)
        hipLaunchKernelGGL(generate_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    else
        hipLaunchKernelGGL(generate_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, VK_ERR_HIP, std::string("generator launch: ") + hipGetErrorString(e)); }
    return build_from_dense(ctx, d, nullptr, true, nx, ny, nz, format, layout);
}

int vk_volume_generate_xor(vk_ctx *ctx, uint32_t nx, uint32_t ny, uint32_t nz, float time) {
    int dummy = 0;
    int rc = check_volume_args(ctx, &dummy, &dummy, nx, ny, nz, VK_FMT_RGBA16F_PAIR, VK_LAYOUT_LINEAR);
    if (rc) return rc;
    if (!std::isfinite(time)) return fail(ctx, VK_ERR_INVALID, "time must be finite");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    if ((n_vox + 255) / 256 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
    void *d = nullptr, *d2 = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * 8));
    hipError_t e = hipMalloc(&d2, n_vox * 8);
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, VK_ERR_OOM, std::string("xor normals: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(xor_generate_kernel, dim3((uint32_t)((n_vox + 255) / 256)), dim3(256), 0, ctx->stream, (uint2 *)d, (uint2 *)d2, nx, ny, nz, time);
    e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); (void)hipFree(d2); return fail(ctx, VK_ERR_HIP, std::string("xor generator launch: ") + hipGetErrorString(e)); }
    return build_from_dense(ctx, d, d2, true, nx, ny, nz, VK_FMT_RGBA16F_PAIR, VK_LAYOUT_AUTO);
}

int vk_volume_empty_fraction(vk_ctx *ctx, double *fraction) {
    if (!ctx || !fraction) return VK_ERR_INVALID;
    if (ctx->format < 0) return fail(ctx, VK_ERR_INVALID, "no volume uploaded");
    *fraction = ctx->empty_fraction;
    return VK_OK;
}

int vk_volume_info(vk_ctx *ctx, uint32_t dims[3], int *format, int *layout, size_t *device_bytes) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->format < 0) return fail(ctx, VK_ERR_INVALID, "no volume uploaded");
    if (dims) { dims[0] = ctx->nx; dims[1] = ctx->ny; dims[2] = ctx->nz; }
    if (format) *format = ctx->format;
    if (layout) *layout = ctx->layout;
    if (device_bytes) *device_bytes = ctx->vol_bytes;
    return VK_OK;
}

// ---- uniforms ------------------------------------------------------------------------------------

int vk_set_uniform(vk_ctx *ctx, const void *blob48) {
    if (!ctx || !blob48) return fail(ctx, VK_ERR_INVALID, "vk_set_uniform: NULL argument");
    std::memcpy(ctx->uniform, blob48, 48);  // read by neither fs_main nor get_col2
    return VK_OK;
}

int vk_set_camera(vk_ctx *ctx, const void *blob144) {
    if (!ctx || !blob144) return fail(ctx, VK_ERR_INVALID, "vk_set_camera: NULL argument");
    std::memcpy(ctx->camera, blob144, 144);
    for (int i = 0; i < 36; i++)
        if (!std::isfinite(ctx->camera[i])) { ctx->have_camera = false; return fail(ctx, VK_ERR_INVALID, "camera blob has non-finite entries"); }
    ctx->have_camera = true;
    return VK_OK;
}

// ---- backbuffer ------------------------------------------------------------------------------------

int vk_backbuffer_resize(vk_ctx *ctx, uint32_t width, uint32_t height, int out_format) {
    if (!ctx) return VK_ERR_INVALID;
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, VK_ERR_INVALID, "backbuffer size must be in [1, 32768]");
    if (out_format != VK_OUT_RGBA32F && out_format != VK_OUT_RGBA16F) return fail(ctx, VK_ERR_INVALID, "unknown output format");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &b : ctx->batch) b.id = 0;  // batch ids held by the caller named tiles of the old shape
    ctx->batch_key.clear();
    for (auto &r : ctx->batch_retired) { (void)hipFree(r.first); (void)hipHostFree(r.second); }
    ctx->batch_retired.clear();
    if (ctx->backbuffer) (void)hipFree(ctx->backbuffer);
    if (ctx->steps) (void)hipFree(ctx->steps);
    ctx->backbuffer = nullptr;
    ctx->steps = nullptr;
    ctx->width = ctx->height = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->backbuffer, (size_t)width * height * px_bytes(out_format)));
    ctx->width = width;
    ctx->height = height;
    ctx->out_format = out_format;
    return vk_backbuffer_clear(ctx);
}

int vk_backbuffer_info(vk_ctx *ctx, uint32_t *width, uint32_t *height, int *out_format, void **device_ptr) {
    if (!ctx) return VK_ERR_INVALID;
    if (width) *width = ctx->width;
    if (height) *height = ctx->height;
    if (out_format) *out_format = ctx->out_format;
    if (device_ptr) *device_ptr = ctx->backbuffer;
    return VK_OK;
}

int vk_backbuffer_clear(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "no backbuffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t)ctx->width * ctx->height;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (ctx->out_format == VK_OUT_RGBA16F)
        hipLaunchKernelGGL(clear_kernel<OUT_RGBA16F>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->backbuffer, n);
    else
        hipLaunchKernelGGL(clear_kernel<OUT_RGBA32F>, dim3(blocks), dim3(256), 0, ctx->stream, ctx->backbuffer, n);
    HIP_TRY(ctx, hipGetLastError());
    return VK_OK;
}

// ---- render ------------------------------------------------------------------------------------

}  // extern "C"

template <int VOL, bool SKIP, bool SAFE>
static void launch_naive(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V_in, uint32_t grid, bool count) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    VolumeDesc V = V_in;
    if (!SKIP && V.lut) V.lut += cell_lut_entries(V.nx, V.ny, V.nz);  // byte-offset copy of the tables
    // the fast path of the cell layouts keeps its per-axis index tables in LDS (vk_kernels.hpp: load_cell_luts)
    constexpr bool lut = (VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16) && !SAFE;
    const uint32_t lds = (lut ? cell_lut_bytes(V.nx, V.ny, V.nz) : 0u) + ctx->naive_lds_pad;  // (pad: occupancy experiments, vk_debug_set_param)
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, OUT_RGBA16F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, OUT_RGBA16F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, OUT_RGBA32F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, OUT_RGBA32F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    }
}

template <int VOL>
static void launch_packed(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, bool skip, bool safe) {
    if (skip) { if (safe) launch_naive<VOL, true, true>(ctx, L, V, grid, count); else launch_naive<VOL, true, false>(ctx, L, V, grid, count); }
    else { if (safe) launch_naive<VOL, false, true>(ctx, L, V, grid, count); else launch_naive<VOL, false, false>(ctx, L, V, grid, count); }
}

// LDS window per wave of the staged march: more LDS = thicker slabs (fewer rounds, each with its slab search, bounds
// and fill) but fewer waves per CU.  The best budget depends on the view: what counts is the box an 8 x 8 pixel wave
// sweeps per slab -- its footprint in cells (distance x pixel angle x n) plus the lateral drift of oblique rays -- and
// the measured optimum (tools/staged_cameras.py: six cameras, C4 and C5) follows the bytes of that box for a slab of
// T* = 4 cells (u8: at the VALU issue limit, occupancy first) or 6 cells (f16), in whole waves per CU between 5 KiB (8 waves
// per SIMD) and 20 KiB (2): C5 from far away 5.83 -> 3.5 ms, from close by 38.4 -> 34.1 ms against a fixed 8 KiB.
static uint32_t staged_cap_auto(const vk_ctx *ctx, const float *cam, bool u8, uint32_t *slab_cells) {
    const uint32_t fallback = u8 ? 6144u : 10240u;
    // the slab search's upper limit: the budget decides the thickness, this only bounds the search (6 / 8 cells until round 3 cut slabs short that
    // would have fitted: C4 1.855 -> 1.78 ms at 16-24, C5 10.0 -> 9.76 at 16; tools/staged_group.py, profiles/r03_staged_group.txt)
    *slab_cells = u8 ? 16u : 24u;
    if (!cam) return fallback;
    const float *m = cam + 20;
    const double W = ctx->width, H = ctx->height;
    auto dir_of = [&](double px, double py, double d[3]) {
        const double X = 2.0 * px / W - 1.0, Y = 1.0 - 2.0 * py / H;
        const double qw = m[3] * X + m[7] * Y + m[11] + m[15];
        double len = 0.0;
        for (int k = 0; k < 3; k++) { d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) / qw - cam[k]; len += d[k] * d[k]; }
        len = std::sqrt(len);
        for (int k = 0; k < 3; k++) d[k] /= len;
        return len;
    };
    double a[3], b[3];
    if (!(dir_of(0.5 * W, 0.5 * H, a) > 0.0) || !(dir_of(0.5 * W + 8.0, 0.5 * H + 8.0, b) > 0.0)) return fallback;
    const double cosang = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    if (!std::isfinite(cosang)) return fallback;
    const double ang = std::acos(std::min(1.0, std::max(-1.0, cosang))) / std::sqrt(2.0);  // 8 pixels along one image axis
    const double to_c[3] = {0.5 - cam[0], 0.5 - cam[1], 0.5 - cam[2]};
    const double dist = std::sqrt(to_c[0] * to_c[0] + to_c[1] * to_c[1] + to_c[2] * to_c[2]);
    const double n[3] = {(double)ctx->nx, (double)ctx->ny, (double)ctx->nz};
    // cells per step along each axis for the central ray; the largest is the slab axis S, then F = S+1, M = S+2 (vk_staged.hpp)
    const double c[3] = {std::fabs(a[0]) * n[0], std::fabs(a[1]) * n[1], std::fabs(a[2]) * n[2]};
    const int S = (c[0] >= c[1] && c[0] >= c[2]) ? 0 : (c[1] >= c[2] ? 1 : 2), F = (S + 1) % 3, M = (S + 2) % 3;
    if (!(c[S] > 0.0) || !std::isfinite(dist)) return fallback;
    const double Tstar = u8 ? 4.0 : 6.0;
    const double rows = std::ceil(ang * dist * n[M] + c[M] / c[S] * Tstar + 3.0), cols = std::ceil(ang * dist * n[F] + c[F] / c[S] * Tstar + 3.0);
    const double per_piece = u8 ? 16.0 : 8.0;
    const double bytes = (Tstar + 1.0) * rows * std::ceil((cols + per_piece - 1.0) / per_piece) * 16.0;
    if (!std::isfinite(bytes)) return fallback;
    // a budget buys whole waves per CU (160 KiB of LDS, 4 SIMDs): the smallest of 32, 28, ... 8 waves' shares that holds the box
    for (uint32_t waves = 32u; waves > 8u; waves -= 4u) {
        const uint32_t cap = (163840u / waves) & ~15u;
        if ((double)cap >= 0.95 * bytes) return cap;
    }
    return 20480u;
}

template <int VOL>
static void launch_staged(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, const float *cam) {
    StagedDesc D = ctx->sdesc;
    uint32_t slab_auto = 8u;
    const uint32_t cap_auto = staged_cap_auto(ctx, cam, VOL == VOL_S8U8, &slab_auto);
    D.cap_bytes = std::min(std::max((ctx->stage_cap_bytes ? ctx->stage_cap_bytes : cap_auto) & ~15u, 1024u), 65536u);
    D.slab_cells = std::min(std::max(ctx->stage_slab_cells ? ctx->stage_slab_cells : slab_auto, 1u), 32u);
    D.row_pad = ctx->stage_row_pad == 1u ? 1u : 0u;
    // u8 (at the issue-slot limit): every 4th round -- C5 10.40 -> 10.12 ms, other views +-1 %; f16 (waiting on fills, not on slots): every round
    D.grow_every = ctx->stage_grow_every ? ctx->stage_grow_every : (VOL == VOL_S8U8 ? 4u : 1u);
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    // One window for the four waves of a 256-thread group (2 x 2 neighbouring 8x8 blocks) instead of one per wave: the rays of 16 x 16 pixels
    // sweep far less than four 8 x 8 boxes, so the slab is ~twice as thick and a ray meets half as many rounds -- against two barriers
    // per round.  Measured (tools/staged_group.py, staged_group_sweep.py; frames bitwise equal): it pays where the wave is starved of LDS --
    // u8 volumes seen from close by (C5: 10.05 -> 9.79 ms; four orbit frames per launch 9.25 -> 9.17; 6.25 KiB per wave and slabs up to 24 cells
    // read 9.83 single but 9.30 in the 4-frame launch: the per-wave budget stays) -- and costs where the budget is large
    // already (far views: a workgroup's 64 KiB is less than four waves' 80) or the launch is one partial round of waves (C4: single frame
    // +2.5 %, four orbit frames per launch 1.67 -> 1.58 ms: f16 takes it in launches of four frames or more).  stage_group: 0 never, 1 always,
    // 2 (default) by these rules.
    const bool group = (L.ts & 15u) == 0u && (ctx->stage_group == 1u || (ctx->stage_group == 2u && !ctx->stage_cap_bytes &&
                                                                                 (VOL == VOL_S8U8 ? cap_auto <= 6400u : (L.frames != nullptr && L.n_frames >= 4u && cap_auto <= 16384u))));
    if (group) {
        // An odd row pitch (in 16-byte pieces) spreads the window rows of neighbouring pixel rows over the LDS banks.  Per wave it cost C5 4 % (a
        // padded piece per row out of 5 KiB); in a group's window the pad is a smaller share and four waves' taps collide more: C5 9.93 -> 9.53 ms,
        // the diagonal view -3.5 %, four orbit frames per launch 9.19 -> 8.91, axis-aligned +2 %, close-up +-0 (profiles/r03_staged_group.txt).
        if (ctx->stage_row_pad == 2u && VOL == VOL_S8U8) D.row_pad = 1u;
        if (!ctx->stage_slab_cells) D.slab_cells = std::max(D.slab_cells, 24u);  // four waves' LDS hold a slab about twice as thick
        const uint32_t lds = std::min(D.cap_bytes * kGroupWaves, 65536u) & ~15u;  // four waves' LDS, less the exchange block
        D.cap_bytes = lds - kGroupExchBytes;
        const uint32_t groups = (grid + kGroupWaves - 1u) / kGroupWaves;
        if (f16) {
            if (count) hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA16F, true>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
            else hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA16F, false>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
        } else {
            if (count) hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA32F, true>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
            else hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA32F, false>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
        }
        return;
    }
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA16F, true>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
        else hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA16F, false>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA32F, true>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
        else hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA32F, false>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
    }
}

// Screen-space bounding rectangle of the unit cube (NAIVE mode): the 8 corners projected with
// proj_view in double; any corner at or behind the eye plane disables the cull.  Padded by 2 px.
// Pixels outside [x0,x1) x [y0,y1) cannot hit the box.
static void cull_rect_wh(uint32_t W, uint32_t H, const float *cam, int mode, int32_t r[4]) {
    r[0] = 0; r[1] = 0; r[2] = (int32_t)W; r[3] = (int32_t)H;
    if (mode != VK_MODE_NAIVE_TRILINEAR) return;
    const float *pv = cam + 4;
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
    for (int c = 0; c < 8; c++) {
        const double X = c & 1, Y = (c >> 1) & 1, Z = (c >> 2) & 1;
        const double cx = pv[0] * X + pv[4] * Y + pv[8] * Z + pv[12], cy = pv[1] * X + pv[5] * Y + pv[9] * Z + pv[13];
        const double cw = pv[3] * X + pv[7] * Y + pv[11] * Z + pv[15];
        if (!(cw > 1e-6)) return;
        const double sx = (cx / cw * 0.5 + 0.5) * W, sy = (0.5 - cy / cw * 0.5) * H;
        x0 = std::min(x0, sx); x1 = std::max(x1, sx); y0 = std::min(y0, sy); y1 = std::max(y1, sy);
    }
    if (!(std::isfinite(x0) && std::isfinite(x1) && std::isfinite(y0) && std::isfinite(y1))) return;
    r[0] = (int32_t)std::max(0.0, std::floor(x0) - 2.0);
    r[1] = (int32_t)std::max(0.0, std::floor(y0) - 2.0);
    r[2] = (int32_t)std::min((double)W, std::ceil(x1) + 2.0);
    r[3] = (int32_t)std::min((double)H, std::ceil(y1) + 2.0);
}
static void cull_rect_cam(const vk_ctx *ctx, const float *cam, int mode, int32_t r[4]) { cull_rect_wh(ctx->width, ctx->height, cam, mode, r); }

static void cull_rect(const vk_ctx *ctx, int mode, int32_t r[4]) { cull_rect_cam(ctx, ctx->camera, mode, r); }

// The cube's silhouette on the screen: the convex hull of its 8 projected corners (counter-clockwise in screen
// coordinates, y down), in double.  A pixel's ray hits the box only if the pixel centre lies inside it, so a tile that a
// hull edge separates from it by more than 2 px holds only clear-colour pixels.  The bounding rectangle alone keeps
// 288 of C2's 510 tiles; the hull keeps the ones a ray can actually hit.  n = 0: no hull (a corner behind the eye
// plane, or another mode) -- the rectangle decides alone.
struct CullHull { int n = 0; double x[16], y[16]; };
static void cull_hull_wh(uint32_t W, uint32_t H, const float *cam, int mode, CullHull &h) {
    h.n = 0;
    if (mode != VK_MODE_NAIVE_TRILINEAR) return;
    const float *pv = cam + 4;
    std::pair<double, double> p[8];
    for (int c = 0; c < 8; c++) {
        const double X = c & 1, Y = (c >> 1) & 1, Z = (c >> 2) & 1;
        const double cx = pv[0] * X + pv[4] * Y + pv[8] * Z + pv[12], cy = pv[1] * X + pv[5] * Y + pv[9] * Z + pv[13];
        const double cw = pv[3] * X + pv[7] * Y + pv[11] * Z + pv[15];
        if (!(cw > 1e-6)) return;
        p[c] = {(cx / cw * 0.5 + 0.5) * W, (0.5 - cy / cw * 0.5) * H};
        if (!(std::isfinite(p[c].first) && std::isfinite(p[c].second))) return;
    }
    std::sort(p, p + 8);
    auto cross = [](const std::pair<double, double> &o, const std::pair<double, double> &a, const std::pair<double, double> &b) {
        return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first);
    };
    std::pair<double, double> hull[16];
    int k = 0;
    for (int i = 0; i < 8; i++) { while (k >= 2 && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    for (int i = 6, t = k + 1; i >= 0; i--) { while (k >= t && cross(hull[k - 2], hull[k - 1], p[i]) <= 0) k--; hull[k++] = p[i]; }
    k--;  // (the last point repeats the first)
    if (k < 3) return;  // degenerate (edge-on): the rectangle decides
    h.n = k;
    for (int i = 0; i < k; i++) { h.x[i] = hull[i].first; h.y[i] = hull[i].second; }
}
// (forward) the tile-level decision, shared by the tile order and vk_tiles_active
static bool tile_is_inactive(const int32_t cr[4], const CullHull &hull, int64_t x0, int64_t y0, uint32_t ts);

// true when some hull edge has the whole rectangle [x0,x1] x [y0,y1] more than `pad` pixels on its outer side
static bool hull_separates(const CullHull &h, double x0, double y0, double x1, double y1, double pad) {
    for (int i = 0; i < h.n; i++) {
        const int j = i + 1 == h.n ? 0 : i + 1;
        const double ex = h.x[j] - h.x[i], ey = h.y[j] - h.y[i];
        const double len = std::sqrt(ex * ex + ey * ey);
        if (!(len > 0)) continue;
        // monotone chain with this cross-product sign walks the hull with its interior on the left: d < 0 is outside
        const double nx = -ey, ny = ex;  // left normal
        const double d0 = nx * (x0 - h.x[i]) + ny * (y0 - h.y[i]), d1 = nx * (x1 - h.x[i]) + ny * (y0 - h.y[i]);
        const double d2 = nx * (x0 - h.x[i]) + ny * (y1 - h.y[i]), d3 = nx * (x1 - h.x[i]) + ny * (y1 - h.y[i]);
        if (std::max(std::max(d0, d1), std::max(d2, d3)) < -pad * len) return true;
    }
    return false;
}

static bool tile_is_inactive(const int32_t cr[4], const CullHull &hull, int64_t x0, int64_t y0, uint32_t ts) {
    return x0 + ts <= cr[0] || x0 >= cr[2] || y0 + ts <= cr[1] || y0 >= cr[3] ||
           (hull.n && hull_separates(hull, (double)x0, (double)y0, (double)(x0 + ts), (double)(y0 + ts), 2.0));
}

// Tiles are dealt to the launch (and, at N > 1, to the ranks) heaviest first.  The frame is ~70 %
// empty and a dense ray ends after 2 steps while a grazing one takes 513, so with ~10 working waves
// per SIMD the kernel's tail is set by whichever heavy tiles start last; starting them first (and
// round-robining them over ranks) shortens it.  The cost estimate is the nominal step count of a
// 3x3 grid of rays per tile, from the same camera maths as the kernel, in double precision on the
// host.  It is only a launch order: every tile is rendered by the same kernel whatever its rank.
static void compute_tile_order_raw(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                                   uint32_t *order, uint32_t *order_pos, uint32_t &order_active, int G) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    const double W = ctx->width, H = ctx->height;
    // tiles that do not touch the cube's screen rectangle hold only clear-colour pixels: they sort last (in index
    // order) and are "inactive" -- never marched, never gathered (the root clears them in vk_untile); no rays for them
    int32_t cr[4];
    cull_rect_cam(ctx, cam, mode, cr);
    CullHull hull;
    cull_hull_wh(ctx->width, ctx->height, cam, mode, hull);
    struct Key { double cost; uint32_t tile; };
    std::vector<Key> act;
    act.reserve(n);
    const float *m = cam + 20;
    const double dims[3] = {(double)std::max(ctx->nx, 1u), (double)std::max(ctx->ny, 1u), (double)std::max(ctx->nz, 1u)};
    uint32_t n_inactive = 0;
    for (uint32_t j = 0; j < ty; j++)
        for (uint32_t i = 0; i < tx; i++) {
            const int64_t x0 = (int64_t)ox + (int64_t)i * ts, y0 = (int64_t)oy + (int64_t)j * ts;
            const uint32_t tile = j * tx + i;
            if (tile_is_inactive(cr, hull, x0, y0, ts)) { order[n - 1 - n_inactive++] = tile; continue; }  // (reversed below)
            double c = 0.0;
            for (int sy = 0; sy < G; sy++)
                for (int sx = 0; sx < G; sx++) {
                    const double px = (double)x0 + (2 * sx + 1) * ts / (2.0 * G), py = (double)y0 + (2 * sy + 1) * ts / (2.0 * G);
                    if (px < 0 || py < 0 || px >= W || py >= H) continue;
                    double e[3], d[3], lo, hi;
                    if (mode == VK_MODE_NAIVE_TRILINEAR) {
                        const double X = 2.0 * px / W - 1.0, Y = 1.0 - 2.0 * py / H;
                        const double qw = 1.0 / (m[3] * X + m[7] * Y + m[11] + m[15]);
                        for (int k = 0; k < 3; k++) { e[k] = cam[k]; d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) * qw - e[k]; }
                        lo = 0.0; hi = 1.0;
                    } else {
                        const double X = 2.0 * px / W - 1.0, Y = (2.0 * py / H - 1.0) * -(H / W);
                        const double aw = 1.0 / (m[3] * X + m[7] * Y + m[15]), bw = 1.0 / (m[3] * X + m[7] * Y + m[11] + m[15]);
                        for (int k = 0; k < 3; k++) {
                            e[k] = (m[k] * X + m[4 + k] * Y + m[12 + k]) * aw;
                            d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) * bw - e[k];
                        }
                        lo = -1.0; hi = 1.0;
                    }
                    const double len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                    if (!(len2 > 0)) continue;
                    // steps = (t1 - t0) / dt with t in units of |d| (the normalisation cancels): dt = min_k 1 / (dims_k |d_k|)
                    double t0 = -1e300, t1 = 1e300, inv_dt = 0.0;
                    for (int k = 0; k < 3; k++) {
                        const double inv = 1.0 / d[k], ta = (lo - e[k]) * inv, tb = (hi - e[k]) * inv;
                        t0 = std::max(t0, std::min(ta, tb));
                        t1 = std::min(t1, std::max(ta, tb));
                        inv_dt = std::max(inv_dt, dims[k] * std::fabs(d[k]));
                    }
                    t0 = std::max(t0, 0.0);
                    if (t1 > t0 && inv_dt > 0) c += (t1 - t0) * inv_dt;
                }
            act.push_back({c, tile});
        }
    const uint32_t n_active = (uint32_t)act.size();
    std::stable_sort(act.begin(), act.end(), [](const Key &a, const Key &b) { return a.cost > b.cost; });
    for (uint32_t q = 0; q < n_active; q++) order[q] = act[q].tile;
    std::reverse(order + n_active, order + n);  // inactive tiles in index order
    // Position q goes to XCD q % 8 (rank q % N first, when the frame is partitioned): dealt straight, bin 0
    // would receive the heaviest tile of every round of 8.  Reverse every other round (snake) so the bins'
    // sums even out; the active tiles stay in front.
    for (size_t g = 8; g + 8 <= n_active; g += 16) std::reverse(order + g, order + g + 8);
    order_active = n_active;
    for (size_t q = 0; q < n; q++) order_pos[order[q]] = (uint32_t)q;
}

static void compute_tile_order(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                               std::vector<uint32_t> &order, std::vector<uint32_t> &order_pos, uint32_t &order_active) {
    const size_t n = (size_t)((rw + ts - 1) / ts) * ((rh + ts - 1) / ts);
    order.resize(n); order_pos.resize(n);
    compute_tile_order_raw(ctx, cam, mode, ox, oy, rw, rh, ts, order.data(), order_pos.data(), order_active, (int)ctx->order_rays);
}

static int tile_order_update(vk_ctx *ctx, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                             float dt_scale) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    std::vector<unsigned char> key(144 + 40);
    std::memcpy(key.data(), ctx->camera, 144);
    const uint32_t kk[10] = {(uint32_t)mode, (uint32_t)ox, (uint32_t)oy, rw, rh, ts, ctx->width, ctx->height, ctx->nx ^ (ctx->ny << 10) ^ (ctx->nz << 20), 0};
    std::memcpy(key.data() + 144, kk, 40);
    (void)dt_scale;
    if (key == ctx->order_key && ctx->order.size() == n) return VK_OK;
    compute_tile_order(ctx, ctx->camera, mode, ox, oy, rw, rh, ts, ctx->order, ctx->order_pos, ctx->order_active);
    const uint32_t n_active = ctx->order_active;
    constexpr int kOrderRing = 16;
    if (ctx->d_order_cap < n) {
        HIP_TRY(ctx, hipDeviceSynchronize());  // a larger frame shape: rebuild the ring (rare)
        if (ctx->d_ring) (void)hipFree(ctx->d_ring);
        if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
        ctx->d_ring = ctx->h_ring = nullptr; ctx->d_order = ctx->d_order_pos = nullptr;
        ctx->d_order_cap = 0; ctx->ring_slot = -1;
        HIP_TRY(ctx, hipMalloc(&ctx->d_ring, (size_t)kOrderRing * 2 * n * sizeof(uint32_t)));
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_ring, (size_t)kOrderRing * 2 * n * sizeof(uint32_t)));
        ctx->d_order_cap = n;
    }
    const size_t cap = ctx->d_order_cap;
    const int slot = (int)(++ctx->order_seq % (uint32_t)kOrderRing);
    // Waiting on the slot's previous upload (kOrderRing cameras ago) makes the pinned staging safe to rewrite.  The
    // device slot itself is safe to overwrite because a context works on ONE stream: the kernels that read the slot
    // kOrderRing cameras ago were enqueued on ctx->stream before this copy (vk_ctx_set_stream drains the old stream
    // first), so the copy is stream-ordered after them.  (Frames in flight are batched launches now, vk_render_batch,
    // which carry their own tables.)
    if (ctx->ring_ev[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->ring_ev[slot]));
    else HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_ev[slot], hipEventDisableTiming));
    uint32_t *hs = ctx->h_ring + (size_t)slot * 2 * cap, *ds = ctx->d_ring + (size_t)slot * 2 * cap;
    std::memcpy(hs, ctx->order.data(), n * sizeof(uint32_t));
    std::memcpy(hs + cap, ctx->order_pos.data(), n * sizeof(uint32_t));
    HIP_TRY(ctx, hipMemcpyAsync(ds, hs, (cap + n) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ring_ev[slot], ctx->stream));
    ctx->ring_stream[slot] = ctx->stream;
    ctx->ring_done[slot] = false;
    ctx->ring_active[slot] = n_active;
    ctx->ring_slot = slot;
    ctx->d_order = ds;
    ctx->d_order_pos = ds + cap;
    ctx->order_key = key;
    return VK_OK;
}

// A launch on a stream other than the one that uploaded the current order slot waits for that upload.
static int order_wait(vk_ctx *ctx) {
    const int s = ctx->ring_slot;
    if (s < 0 || ctx->ring_done[s] || ctx->stream == ctx->ring_stream[s]) return VK_OK;
    if (hipEventQuery(ctx->ring_ev[s]) == hipSuccess) { ctx->ring_done[s] = true; return VK_OK; }
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_ev[s], 0));
    return VK_OK;
}

// Argument and state checks shared by every render entry point; `cam` is the 144-byte camera the frame uses.
static int check_render(vk_ctx *ctx, int mode, const float *cam, float dt_scale, uint32_t ts, uint32_t rank, uint32_t nranks) {
    if (!ctx) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (ctx->format < 0 && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "render: no volume uploaded");
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "render: no backbuffer (vk_backbuffer_resize)");
    if (!cam) return fail(ctx, VK_ERR_INVALID, "render: no camera (vk_set_camera)");
    // PROCEDURAL shares the compute twin's ray, box and step: geometry helpers treat it as that mode
    const int geo_mode = mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode;
    if (mode == VK_MODE_NAIVE_TRILINEAR && ctx->format == VK_FMT_RGBA16F_PAIR)
        return fail(ctx, VK_ERR_INVALID, "NAIVE_TRILINEAR needs a scalar volume (R8_UNORM / R16_FLOAT)");
    if (mode == VK_MODE_COMPUTE_NEAREST && ctx->format != VK_FMT_RGBA16F_PAIR)
        return fail(ctx, VK_ERR_INVALID, "COMPUTE_NEAREST needs an RGBA16F_PAIR volume");
    if (!(dt_scale > 0.0f) || !std::isfinite(dt_scale)) return fail(ctx, VK_ERR_INVALID, "dt_scale must be finite and > 0");
    if (ts == 0 || (ts & 7u) || ts > 1024) return fail(ctx, VK_ERR_INVALID, "tile size must be a multiple of 8 in [8, 1024]");
    if (nranks == 0 || rank >= nranks) return fail(ctx, VK_ERR_INVALID, "rank/nranks");
    // Loop-termination guard (the reference would hang the GPU on a dt that no longer advances t):
    // t <= |eye - box| + box diagonal; require dt >= 8 ulp(t_max).
    {
        const float *e = cam;
        float reach = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + 4.0f;
        if (geo_mode == VK_MODE_COMPUTE_NEAREST) reach += 200.0f;  // near-plane point of a far=100 frustum
        float nmax = (float)std::max(ctx->nx, std::max(ctx->ny, ctx->nz));
        float dt_min = mode == VK_MODE_NAIVE_TRILINEAR ? dt_scale / nmax : dt_scale * 0.01f;
        float ulp = std::nextafter(reach, 2.0f * reach) - reach;
        if (!(dt_min >= 8.0f * ulp)) return fail(ctx, VK_ERR_UNSUPPORTED, "dt too small against the camera distance: the march would not advance");
    }
    return VK_OK;
}

// Launch the march kernel of the context's volume for a filled LaunchDesc (one frame or a batch).
// `reach_cam`: the camera whose distance decides whether the unclamped fast path is safe (the farthest of a batch).
static int dispatch_march(vk_ctx *ctx, int mode, const LaunchDesc &L_in, uint32_t flags, const float *reach_cam) {
    LaunchDesc L = L_in;
    const bool count = (flags & VK_RENDER_COUNT) != 0;
    VolumeDesc V = ctx->vdesc;
    V.data = ctx->vol; V.data2 = ctx->vol2; V.dist = ctx->dist;
    V.lut = ctx->lut;  // the no-skip variants take the byte-offset copy (set where the variant is chosen)
    V.nx = ctx->nx; V.ny = ctx->ny; V.nz = ctx->nz;
    V.nbx = ctx->nbx; V.nby = ctx->nby; V.nbz = ctx->nbz;
    const uint64_t n_blocks = L.n_blocks;
    uint32_t grid = (uint32_t)((n_blocks + 511) / 512 * 512);
    L.grid_march = grid;
    if (mode != VK_MODE_NAIVE_TRILINEAR) L.clear_max_inactive = 0;  // (their kernels have no clearing blocks: every tile is active)
    // whole-frame batches: the strips that clear the inactive tiles ride behind the march blocks (clear_inactive_strip)
    const uint64_t clear_blocks = (uint64_t)L.clear_max_inactive * L.n_frames * ((L.ts * L.ts + 511u) / 512u);
    if ((uint64_t)grid + clear_blocks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "launch too large: fewer frames per batch");
    grid += (uint32_t)clear_blocks;
    if (grid == 0) return VK_OK;
    if (mode == VK_MODE_PROCEDURAL) {
        float time = 0.0f;
        std::memcpy(&time, ctx->uniform + 36, sizeof(float));  // Uniform.time (global_ubo.rs:52-65), what xor.wgsl reads as un.time
        if (!std::isfinite(time)) return fail(ctx, VK_ERR_INVALID, "Uniform.time must be finite");
        const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
        if (f16) {
            if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
            else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        } else {
            if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
            else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        }
    } else if (mode == VK_MODE_COMPUTE_NEAREST && ctx->vol_kind == VOL_PAIRB) {
        const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
        const uint32_t lds = pair_lut_entries(V.nx, V.ny, V.nz) * 4u;
        if (f16) {
            if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
            else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        } else {
            if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
            else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        }
    } else if (mode == VK_MODE_COMPUTE_NEAREST) {
        const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
        if (f16) {
            if (count) hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
            else hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
        } else {
            if (count) hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
            else hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
        }
    } else {
        // Skipping costs a distance lookup per probing trip; it only pays when there is something to skip
        // (tools/skip_crossover.py, DESIGN.md section 4: on 256^3 volumes with a share e of exactly-transparent cells
        // the skip kernel overtakes the dense one between e = 0.36 and e = 0.56: 0.335 / 0.348 / 0.407 ms for dense /
        // adaptive / probing always at e = 0.36, 0.332 / 0.302 / 0.292 at e = 0.56).  Default policy by the census taken
        // at upload:  e < 0.45: the dense kernel;  0.45 <= e < 0.55: the skip kernel with adaptive probing (dense
        // stretches where nothing is being skipped);  e >= 0.55: the skip kernel probing on every trip (emptier
        // volumes -- the bonsai stand-in is at 0.77 -- spend their time in the skip walks, and the stretches only cost).
        // VK_RENDER_FORCE_SKIP takes the skip kernel whatever the census, adaptive unless VK_RENDER_PROBE_ALWAYS.
        const bool forced = (flags & VK_RENDER_FORCE_SKIP) != 0;
        const bool skip = !(flags & VK_RENDER_NO_SKIP) && (forced || ctx->empty_fraction >= 0.45);
        if (skip && !forced && ctx->empty_fraction >= 0.55) L.debug_flags &= ~4u;
        if (skip && ctx->empty_fraction < 0.05) L.debug_flags |= 8u;  // forced on (almost) solid material: long dense stretches from the start
        // SAFE=false (no per-axis clamps, 32-bit offsets, index tables in LDS) only when provably
        // harmless: the cell array is < 4 GiB, the tables fit a modest LDS budget, and the camera is
        // near enough that the accumulated position stays within 0.5/n of the box
        // (|p error| <= ~64 ulp(reach) << 0.5/n).
        bool safe = true;
        {
            const float *e = reach_cam;
            float reach = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + 4.0f;
            float nmax = (float)std::max(ctx->nx, std::max(ctx->ny, ctx->nz));
            float ulp = std::nextafter(reach, 2.0f * reach) - reach;
            if (V.max_off + 16 < (1ll << 32) && cell_lut_bytes(ctx->nx, ctx->ny, ctx->nz) <= 16384u && 64.0f * ulp < 0.25f / nmax && !(flags & VK_RENDER_SAFE)) safe = false;
        }
        switch (ctx->vol_kind) {
            case VOL_P8: launch_packed<VOL_P8>(ctx, L, V, grid, count, skip, safe); break;
            case VOL_P16: launch_packed<VOL_P16>(ctx, L, V, grid, count, skip, safe); break;
            case VOL_PF16: launch_packed<VOL_PF16>(ctx, L, V, grid, count, skip, safe); break;
            case VOL_B9U8: launch_naive<VOL_B9U8, false, true>(ctx, L, V, grid, count); break;
            case VOL_B9F16: launch_naive<VOL_B9F16, false, true>(ctx, L, V, grid, count); break;
            case VOL_S8U8: launch_staged<VOL_S8U8>(ctx, L, V, grid, count, reach_cam); break;
            case VOL_S8F16: launch_staged<VOL_S8F16>(ctx, L, V, grid, count, reach_cam); break;
            case VOL_Q8: launch_naive<VOL_Q8, false, true>(ctx, L, V, grid, count); break;
            case VOL_QF16: launch_naive<VOL_QF16, false, true>(ctx, L, V, grid, count); break;
            case VOL_LINEAR_F16: launch_naive<VOL_LINEAR_F16, false, true>(ctx, L, V, grid, count); break;
            default: launch_naive<VOL_LINEAR_U8, false, true>(ctx, L, V, grid, count); break;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return VK_OK;
}

static int render_common(vk_ctx *ctx, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                         uint32_t rank, uint32_t nranks, float dt_scale, uint32_t flags, void *compact_out) {
    if (!ctx) return VK_ERR_INVALID;
    int crc = check_render(ctx, mode, ctx->have_camera ? ctx->camera : nullptr, dt_scale, ts, rank, nranks);
    if (crc) return crc;
    const int geo_mode = mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode;
    if (rw == 0 || rh == 0) return VK_OK;  // empty tile
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool count = (flags & VK_RENDER_COUNT) != 0;
    if (count && !ctx->steps) {
        HIP_TRY(ctx, hipMalloc(&ctx->steps, (size_t)ctx->width * ctx->height * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->steps, 0, (size_t)ctx->width * ctx->height * sizeof(uint32_t), ctx->stream));
    }
    LaunchDesc L{};
    L.eye[0] = ctx->camera[0]; L.eye[1] = ctx->camera[1]; L.eye[2] = ctx->camera[2]; L.eye[3] = ctx->camera[3];
    std::memcpy(L.inv_proj, ctx->camera + 20, 64);
    L.W = ctx->width; L.H = ctx->height;
    L.ox = ox; L.oy = oy; L.rw = rw; L.rh = rh;
    L.ts = ts;
    L.tiles_x = (rw + ts - 1) / ts;
    L.tiles_y = (rh + ts - 1) / ts;
    {
        int32_t cr[4];
        cull_rect(ctx, geo_mode, cr);
        L.cull_x0 = cr[0]; L.cull_y0 = cr[1]; L.cull_x1 = cr[2]; L.cull_y1 = cr[3];
    }
    L.rank = rank; L.nranks = nranks;
    {
        int orc = tile_order_update(ctx, geo_mode, ox, oy, rw, rh, ts, dt_scale);
        if (orc) return orc;
        orc = order_wait(ctx);
        if (orc) return orc;
        L.tile_order = ctx->d_order;
    }
    // a partition (compact output) covers only the active tiles; a plain render covers the whole region
    const uint64_t tiles = compact_out ? (uint64_t)ctx->order_active : (uint64_t)L.tiles_x * L.tiles_y;
    L.root_skip = nranks > 1 ? ctx->root_skip : 0u;
    const uint64_t slots = deal_rounds((uint32_t)tiles, nranks, L.root_skip);
    L.n_tiles_launch = (uint32_t)tiles;
    // (PROCEDURAL / COMPUTE_NEAREST: every tile is active; the naive and staged kernels are the ones that test it)
    L.n_active_tiles = ctx->order_active;
    if (tiles == 0) return VK_OK;
    const uint64_t per_tile = (uint64_t)(ts / 8) * (ts / 8);
    const uint64_t n_blocks = slots * per_tile;
    if (n_blocks >= (1ull << 31) - 512) return fail(ctx, VK_ERR_UNSUPPORTED, "launch too large");
    L.n_blocks = (uint32_t)n_blocks;
    L.compact = compact_out ? 1u + (uint32_t)ctx->wire : 0u;
    L.dt_scale = dt_scale;
    L.out = compact_out ? compact_out : ctx->backbuffer;
    L.steps = count ? ctx->steps : nullptr;
    L.counters = count ? ctx->counters : nullptr;
    L.trace = nullptr;
    L.frames = nullptr;
    L.n_frames = 1;
    L.debug_flags = ((flags & VK_RENDER_DEBUG_TRIPS) ? 1u : 0u) | ((flags & VK_RENDER_DEBUG_FALLBACK) ? 2u : 0u) | ((flags & VK_RENDER_PROBE_ALWAYS) ? 0u : 4u) | (ctx->wave_prio ? 16u : 0u);
    L.walk_cap = ctx->walk_cap ? (float)ctx->walk_cap : HUGE_VALF;
    L.walk_cap_all = ctx->walk_cap_all ? (float)ctx->walk_cap_all : HUGE_VALF;
    if (count && ctx->want_trace) {
        // (a per-trip log of trip_log_cap u32 per wave = trip_log_cap / 8 records of the stamps' size)
        const uint64_t recs = ctx->trip_log_cap ? n_blocks * (ctx->trip_log_cap / 8u) : n_blocks;
        if (ctx->trace_blocks < recs) {
            if (ctx->trace) (void)hipFree(ctx->trace);
            ctx->trace = nullptr; ctx->trace_blocks = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->trace, recs * 4 * sizeof(unsigned long long)));
            ctx->trace_blocks = recs;
        }
        // start = +inf (atomicMin), end = 0 (atomicMax): fill {0xff.., 0} pairs
        std::vector<unsigned long long> init(recs * 4, 0ull);
        if (ctx->trip_log_cap) L.debug_flags |= 64u | (ctx->trip_log_cap << 16);
        else for (uint64_t i = 0; i < n_blocks; i++) init[4 * i] = ~0ull;
        HIP_TRY(ctx, hipMemcpy(ctx->trace, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        L.trace = ctx->trace;
    }
    return dispatch_march(ctx, mode, L, flags, ctx->camera);
}

extern "C" {

int vk_render(vk_ctx *ctx, int mode, int32_t tile_x, int32_t tile_y, uint32_t tile_w, uint32_t tile_h, float dt_scale,
              uint32_t flags) {
    return render_common(ctx, mode, tile_x, tile_y, tile_w, tile_h, 64, 0, 1, dt_scale, flags, nullptr);
}

int vk_partition_wire(vk_ctx *ctx, int wire) {
    if (!ctx) return VK_ERR_INVALID;
    if (wire != VK_WIRE_RGBA && wire != VK_WIRE_RGB) return fail(ctx, VK_ERR_INVALID, "vk_partition_wire: VK_WIRE_RGBA or VK_WIRE_RGB");
    ctx->wire = wire;
    return VK_OK;
}

int vk_wire_pixel_bytes(vk_ctx *ctx, uint32_t *bytes) {
    if (!ctx || !bytes) return fail(ctx, VK_ERR_INVALID, "vk_wire_pixel_bytes: NULL argument");
    *bytes = (uint32_t)wire_px_bytes(ctx->out_format, ctx->wire);
    return VK_OK;
}

int vk_partition_slots(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t *n_slots) {
    return vk_partition_slots_weighted(width, height, tile_size, nranks, 0, n_slots);
}

int vk_partition_slots_weighted(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint32_t root_skip, uint32_t *n_slots) {
    if (!n_slots || tile_size == 0 || (tile_size & 7u) || nranks == 0 || width == 0 || height == 0 || root_skip == 1) return VK_ERR_INVALID;
    uint64_t tiles = (uint64_t)((width + tile_size - 1) / tile_size) * ((height + tile_size - 1) / tile_size);
    *n_slots = deal_rounds((uint32_t)tiles, nranks, nranks > 1 ? root_skip : 0u);
    return VK_OK;
}

// Which tiles of a width x height frame can hold a pixel whose ray hits the volume's box under this camera: the decision
// every partition makes (inactive tiles are never marched nor gathered; the root clears them).  Pure host arithmetic, no
// context: active[tile] (row-major, tiles_x * tiles_y bytes) is 1 or 0.
int vk_tiles_active(const void *camera144, int mode, uint32_t width, uint32_t height, uint32_t tile_size, unsigned char *active, uint32_t *n_active) {
    if (!camera144 || !active || tile_size == 0 || (tile_size & 7u) || width == 0 || height == 0) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return VK_ERR_INVALID;
    float cam[36];
    std::memcpy(cam, camera144, 144);
    for (float v : cam) if (!std::isfinite(v)) return VK_ERR_INVALID;
    const int geo_mode = mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode;
    int32_t cr[4];
    cull_rect_wh(width, height, cam, geo_mode, cr);
    CullHull hull;
    cull_hull_wh(width, height, cam, geo_mode, hull);
    const uint32_t tx = (width + tile_size - 1) / tile_size, ty = (height + tile_size - 1) / tile_size;
    uint32_t n = 0;
    for (uint32_t j = 0; j < ty; j++)
        for (uint32_t i = 0; i < tx; i++) {
            const bool on = !tile_is_inactive(cr, hull, (int64_t)i * tile_size, (int64_t)j * tile_size, tile_size);
            active[(size_t)j * tx + i] = on ? 1 : 0;
            n += on;
        }
    if (n_active) *n_active = n;
    return VK_OK;
}

int vk_partition_root_skip(vk_ctx *ctx, uint32_t root_skip) {
    if (!ctx) return VK_ERR_INVALID;
    if (root_skip == 1) return fail(ctx, VK_ERR_INVALID, "vk_partition_root_skip: 0 (never) or >= 2 (rank 0 sits out every k-th round)");
    ctx->root_skip = root_skip;
    return VK_OK;
}

int vk_render_partition(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t rank, uint32_t nranks, float dt_scale,
                        uint32_t flags, void *compact_out) {
    if (!ctx) return VK_ERR_INVALID;
    if (!compact_out) return fail(ctx, VK_ERR_INVALID, "vk_render_partition: compact_out is NULL");
    return render_common(ctx, mode, 0, 0, ctx->width, ctx->height, tile_size, rank, nranks, dt_scale, flags, compact_out);
}

int vk_partition_active(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t nranks, uint32_t *n_active_tiles, uint32_t *n_active_slots) {
    if (!ctx) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_active: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u) || nranks == 0) return fail(ctx, VK_ERR_INVALID, "bad tile size / nranks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the same geometry key as the render calls use (PROCEDURAL marches the compute twin's rays)
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size, 1.0f);
    if (orc) return orc;
    if (n_active_tiles) *n_active_tiles = ctx->order_active;
    if (n_active_slots) *n_active_slots = deal_rounds(ctx->order_active, nranks, nranks > 1 ? ctx->root_skip : 0u);
    return VK_OK;
}

int vk_debug_set_tile_order(vk_ctx *ctx, const uint32_t *order, uint32_t n) {
    // experiment hook: replace the current (already computed) order table; stays until the key changes
    if (!ctx || !order) return VK_ERR_INVALID;
    if (n != ctx->order.size() || !ctx->d_order) return fail(ctx, VK_ERR_INVALID, "vk_debug_set_tile_order: no order of that size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (uint32_t q = 0; q < n; q++) { ctx->order[q] = order[q]; ctx->order_pos[order[q]] = q; }
    HIP_TRY(ctx, hipMemcpy(ctx->d_order, ctx->order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_order_pos, ctx->order_pos.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    return VK_OK;
}

int vk_partition_order(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t *order_out, uint32_t n_tiles) {
    if (!ctx || !order_out) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: NULL argument");
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_order: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u)) return fail(ctx, VK_ERR_INVALID, "tile size must be a multiple of 8");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size, 1.0f);
    if (orc) return orc;
    if (n_tiles != ctx->order.size()) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: n_tiles does not match the partition");
    std::memcpy(order_out, ctx->order.data(), n_tiles * sizeof(uint32_t));
    return VK_OK;
}

static int untile_common(vk_ctx *ctx, const void *gathered, uint32_t tile_size, uint32_t nranks, uint32_t slot_stride) {
    if (!ctx || !gathered) return fail(ctx, VK_ERR_INVALID, "vk_untile: NULL argument");
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "no backbuffer");
    if (tile_size == 0 || (tile_size & 7u) || nranks == 0) return fail(ctx, VK_ERR_INVALID, "vk_untile: bad tile size / nranks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->have_camera) return fail(ctx, VK_ERR_INVALID, "vk_untile: no camera (the tile order follows the camera)");
    const uint32_t tiles_x = (ctx->width + tile_size - 1) / tile_size;
    // The un-tile follows the order the partitions were marched with: the tables of the LAST partition call,
    // not those of a camera uploaded since.  Only a context that has never partitioned this frame shape derives them here.
    const size_t n_tiles = (size_t)tiles_x * ((ctx->height + tile_size - 1) / tile_size);
    if (!ctx->d_order_pos || ctx->order.size() != n_tiles) {
        const int m = ctx->format == VK_FMT_RGBA16F_PAIR || ctx->format < 0 ? VK_MODE_COMPUTE_NEAREST : VK_MODE_NAIVE_TRILINEAR;
        int orc = tile_order_update(ctx, m, 0, 0, ctx->width, ctx->height, tile_size, 1.0f);
        if (orc) return orc;
    }
    int owc = order_wait(ctx);
    if (owc) return owc;
    const uint32_t *d_pos = ctx->d_order_pos;
    const uint32_t n_active = ctx->order_active;
    const uint32_t n_slots = slot_stride;  // slots per rank in `gathered`
    const uint32_t chunks = (tile_size * tile_size + 511u) / 512u;
    const uint64_t blocks = (uint64_t)n_tiles * chunks;
    if (blocks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "vk_untile: too many tiles for one launch");
    const uint32_t rs = nranks > 1 ? ctx->root_skip : 0u;
    if (ctx->out_format == VK_OUT_RGBA16F)
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA16F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, ctx->backbuffer, ctx->width, ctx->height, tile_size, tiles_x, (uint32_t)n_tiles, nranks, n_slots, 1u, d_pos, (const FrameDesc *)nullptr, rs, n_active, (const uint32_t *)nullptr, (const FrameDesc *)nullptr, (uint32_t)ctx->wire);
    else
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA32F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, ctx->backbuffer, ctx->width, ctx->height, tile_size, tiles_x, (uint32_t)n_tiles, nranks, n_slots, 1u, d_pos, (const FrameDesc *)nullptr, rs, n_active, (const uint32_t *)nullptr, (const FrameDesc *)nullptr, (uint32_t)ctx->wire);
    HIP_TRY(ctx, hipGetLastError());
    return VK_OK;
}

int vk_untile(vk_ctx *ctx, const void *gathered, uint32_t tile_size, uint32_t nranks, uint32_t slot_stride) {
    return untile_common(ctx, gathered, tile_size, nranks, slot_stride);
}

// ---- batched launches -----------------------------------------------------------------------------

int vk_render_batch(vk_ctx *ctx, int mode, uint32_t n_frames, const void *cameras, uint32_t tile_size, uint32_t rank, uint32_t nranks,
                    float dt_scale, uint32_t flags, void *out, int compact, uint32_t slot_capacity, uint32_t *batch_id, uint32_t *n_active_slots) {
    if (!ctx) return VK_ERR_INVALID;
    if (!cameras || !out) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: NULL argument");
    if (n_frames == 0 || n_frames > VK_MAX_BATCH_FRAMES) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: 1..1024 frames per batch");
    if (flags & VK_RENDER_COUNT) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: the step counters describe one frame; count with vk_render");
    if (!compact && nranks != 1) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: whole frames need nranks == 1; a rank's share is compact");
    const float *cams = reinterpret_cast<const float *>(cameras);
    for (uint32_t i = 0; i < n_frames * 36u; i++)
        if (!std::isfinite(cams[i])) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: camera blob has non-finite entries");
    const int geo_mode = mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode;
    const float *far_cam = cams;  // the camera farthest from the volume decides the safe-path test
    for (uint32_t f = 0; f < n_frames; f++) {
        const float *c = cams + 36 * f;
        int crc = check_render(ctx, mode, c, dt_scale, tile_size, rank, nranks);
        if (crc) return crc;
        if (c[0] * c[0] + c[1] * c[1] + c[2] * c[2] > far_cam[0] * far_cam[0] + far_cam[1] * far_cam[1] + far_cam[2] * far_cam[2]) far_cam = c;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t ts = tile_size, tx = (ctx->width + ts - 1) / ts, ty = (ctx->height + ts - 1) / ts;
    const size_t n_tiles = (size_t)tx * ty;
    const size_t bytes = (size_t)n_frames * (sizeof(FrameDesc) + 2 * n_tiles * sizeof(uint32_t));
    vk_ctx::BatchSlot &B = ctx->batch[ctx->batch_seq % 4u];
    if (B.ev) HIP_TRY(ctx, hipEventSynchronize(B.ev));  // the launches of four batches ago have long finished
    else HIP_TRY(ctx, hipEventCreateWithFlags(&B.ev, hipEventDisableTiming));
    B.id = 0;  // claimed: whatever fails below, the slot no longer answers to its old id
    if (B.cap < bytes) {
        // sized for 256 frames from the start and doubled from there; the outgrown blocks are retired, not freed (hipFree and
        // hipHostFree synchronise the device: a driver whose batches grow -- 20, then 32 frames -- stalled four launches each time)
        if (B.d || B.h) ctx->batch_retired.emplace_back(B.d, B.h);
        B.d = B.h = nullptr;
        const size_t want = std::max({bytes, 2 * B.cap, (size_t)256 * (sizeof(FrameDesc) + 2 * n_tiles * sizeof(uint32_t))});
        B.cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&B.d, want));
        HIP_TRY(ctx, hipHostMalloc((void **)&B.h, want));
        B.cap = want;
    }
    FrameDesc *fd = reinterpret_cast<FrameDesc *>(B.h);
    uint32_t *h_order = reinterpret_cast<uint32_t *>(B.h + (size_t)n_frames * sizeof(FrameDesc));
    uint32_t *h_pos = h_order + (size_t)n_frames * n_tiles;
    uint32_t max_active = 0, min_active = 0xffffffffu;
    // The tile order depends on the camera (and the frame / volume shape).  It is written straight into the pinned
    // staging block; a frame with the camera of the frame before it (or of the last frame of the previous batch) copies
    // that frame's tables instead of casting the estimate rays again.
    std::vector<uint32_t> &order = ctx->batch_order, &pos = ctx->batch_pos;
    // Estimate rays per tile: the 3 x 3 grid of the single-frame launches, or just the tile's centre ray when the grid
    // spans >= 4 frames -- position-major over many frames the launch time no longer depends on the finer estimate
    // (tools/order_rays.py) and the host's share drops from 40 to 10 us per camera, which is what an orbiting camera
    // at N = 8 (13.6 us of march per frame and rank) needs.
    const int G = n_frames >= 4 ? (int)ctx->order_rays_batch : (int)ctx->order_rays;
    const uint32_t kk[8] = {(uint32_t)geo_mode, ts, ctx->width, ctx->height, ctx->nx, ctx->ny, ctx->nz, (uint32_t)G};
    std::vector<unsigned char> key(144 + 32);
    std::memcpy(key.data() + 144, kk, 32);
    std::vector<uint32_t> actives(n_frames, 0u);
    // Frames whose camera differs from the frame before them each need an order of their own: ~10 us of host arithmetic apiece (hull test of 510
    // tiles, estimate rays, sort).  One GPU hides that behind its 69 us per frame; a rank of 8 marches its share of a frame in ~9 us, and a
    // stream of distinct cameras would leave it waiting for its own host.  The frames are independent, so they are cut over a few threads
    // (nothing below writes shared state: every frame owns its slice of the staging block).
    std::vector<uint32_t> own;  // frames that compute their order (the others copy the frame before them)
    for (uint32_t f = 0; f < n_frames; f++) {
        const float *c = cams + 36 * f;
        if (f > 0 && std::memcmp(c, c - 36, 144) == 0) continue;
        if (f == 0) {
            std::memcpy(key.data(), c, 144);
            if (key == ctx->batch_key && order.size() == n_tiles) {
                std::memcpy(h_order, order.data(), n_tiles * sizeof(uint32_t));
                std::memcpy(h_pos, pos.data(), n_tiles * sizeof(uint32_t));
                actives[0] = ctx->batch_n_active;
                continue;
            }
        }
        own.push_back(f);
    }
    {
        auto work = [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                const uint32_t f = own[k];
                compute_tile_order_raw(ctx, cams + 36 * f, geo_mode, 0, 0, ctx->width, ctx->height, ts, h_order + (size_t)f * n_tiles, h_pos + (size_t)f * n_tiles, actives[f], G);
            }
        };
        const size_t n_own = own.size();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t n_thr = n_own >= 24 ? std::min<size_t>({4, hw, n_own / 8}) : 1;
        if (n_thr <= 1) work(0, n_own);
        else {
            std::vector<std::thread> pool;
            for (size_t i = 1; i < n_thr; i++) pool.emplace_back(work, n_own * i / n_thr, n_own * (i + 1) / n_thr);
            work(0, n_own / n_thr);
            for (auto &t : pool) t.join();
        }
    }
    for (uint32_t f = 0; f < n_frames; f++) {
        const float *c = cams + 36 * f;
        uint32_t *fo = h_order + (size_t)f * n_tiles, *fp = h_pos + (size_t)f * n_tiles;
        if (f > 0 && std::memcmp(c, c - 36, 144) == 0) {
            std::memcpy(fo, fo - n_tiles, n_tiles * sizeof(uint32_t));
            std::memcpy(fp, fp - n_tiles, n_tiles * sizeof(uint32_t));
            actives[f] = actives[f - 1];
        }
        const uint32_t n_active = actives[f];
        std::memcpy(fd[f].eye, c, 16);
        std::memcpy(fd[f].inv_proj, c + 20, 64);
        int32_t cr[4];
        cull_rect_cam(ctx, c, geo_mode, cr);
        fd[f].cull_x0 = cr[0]; fd[f].cull_y0 = cr[1]; fd[f].cull_x1 = cr[2]; fd[f].cull_y1 = cr[3];
        fd[f].order_off = (uint32_t)(f * n_tiles);
        // the march covers the active tiles only; whole frames get their inactive ones from the clearing strips at the end of the grid
        fd[f].n_active = n_active;
        fd[f].pad[0] = n_active; fd[f].pad[1] = 0;
        max_active = std::max(max_active, n_active);
        min_active = std::min(min_active, n_active);
    }
    {   // remember the last frame's tables for the next batch
        const uint32_t l = n_frames - 1;
        std::memcpy(key.data(), cams + 36 * l, 144);
        ctx->batch_key = key;
        order.assign(h_order + (size_t)l * n_tiles, h_order + (size_t)(l + 1) * n_tiles);
        pos.assign(h_pos + (size_t)l * n_tiles, h_pos + (size_t)(l + 1) * n_tiles);
        ctx->batch_n_active = actives[l];
    }
    const uint32_t root_skip = nranks > 1 ? ctx->root_skip : 0u;
    const uint32_t slots_active = deal_rounds(max_active, nranks, root_skip);
    if (n_active_slots) *n_active_slots = slots_active;
    const uint64_t slots = (uint64_t)slots_active;  // (whole frames: nranks == 1, so slots_active == max_active)
    if (compact && slots_active > slot_capacity) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: slot_capacity smaller than the active slots of this batch");
    HIP_TRY(ctx, hipMemcpyAsync(B.d, B.h, bytes, hipMemcpyHostToDevice, ctx->stream));
    B.id = ++ctx->batch_seq;
    B.n_frames = n_frames; B.n_tiles = (uint32_t)n_tiles; B.ts = ts; B.nranks = nranks; B.max_active = max_active; B.root_skip = root_skip;
    B.width = ctx->width; B.height = ctx->height; B.out_format = ctx->out_format; B.wire = ctx->wire;
    if (batch_id) *batch_id = B.id;
    // whole frames: the tiles behind a frame's active positions get their clear colour from strips at the end of the grid
    const uint32_t clear_max_inactive = (!compact && geo_mode == VK_MODE_NAIVE_TRILINEAR && min_active < n_tiles) ? (uint32_t)n_tiles - min_active : 0u;
    if (slots == 0 && clear_max_inactive == 0) { HIP_TRY(ctx, hipEventRecord(B.ev, ctx->stream)); return VK_OK; }
    LaunchDesc L{};
    L.clear_max_inactive = clear_max_inactive;
    L.W = ctx->width; L.H = ctx->height;
    L.ox = 0; L.oy = 0; L.rw = ctx->width; L.rh = ctx->height;
    L.ts = ts; L.tiles_x = tx; L.tiles_y = ty;
    L.rank = rank; L.nranks = nranks; L.root_skip = root_skip;
    L.tile_order = reinterpret_cast<const uint32_t *>(B.d + (size_t)n_frames * sizeof(FrameDesc));
    L.n_tiles_launch = 0;
    const uint64_t per_tile = (uint64_t)(ts / 8) * (ts / 8);
    const uint64_t n_blocks = slots * n_frames * per_tile;
    if (n_blocks >= (1ull << 31) - 512) return fail(ctx, VK_ERR_UNSUPPORTED, "launch too large: fewer frames per batch");
    L.n_blocks = (uint32_t)n_blocks;
    L.compact = compact ? 1u + (uint32_t)ctx->wire : 0u;
    L.dt_scale = dt_scale;
    L.out = out;
    L.steps = nullptr; L.counters = nullptr; L.trace = nullptr;
    L.debug_flags = ((flags & VK_RENDER_PROBE_ALWAYS) ? 0u : 4u) | (ctx->wave_prio ? 16u : 0u) | (ctx->frame_runs ? 32u : 0u);
    L.walk_cap = ctx->walk_cap ? (float)ctx->walk_cap : HUGE_VALF;
    L.walk_cap_all = ctx->walk_cap_all ? (float)ctx->walk_cap_all : HUGE_VALF;
    L.frames = reinterpret_cast<const FrameDesc *>(B.d);
    L.n_frames = n_frames;
    const int rc = dispatch_march(ctx, mode, L, flags, far_cam);
    HIP_TRY(ctx, hipEventRecord(B.ev, ctx->stream));
    return rc;
}

int vk_untile_batch(vk_ctx *ctx, uint32_t batch_id, const void *gathered, uint32_t n_slots, void *out_frames) {
    return vk_untile_batch_over(ctx, batch_id, gathered, n_slots, out_frames, 0u);
}

int vk_untile_batch_over(vk_ctx *ctx, uint32_t batch_id, const void *gathered, uint32_t n_slots, void *out_frames, uint32_t prev_batch_id) {
    if (!ctx || !gathered || !out_frames) return fail(ctx, VK_ERR_INVALID, "vk_untile_batch: NULL argument");
    vk_ctx::BatchSlot *B = nullptr, *P = nullptr;
    for (auto &b : ctx->batch) if (b.id == batch_id && b.id != 0) B = &b;
    if (!B) return fail(ctx, VK_ERR_INVALID, "vk_untile_batch: that batch is no longer held (more than 3 batches ago)");
    // out_frames still holds the result of un-tiling `prev_batch_id` (the caller's word): tiles inactive then and now are not
    // written again.  A batch of another shape, or one no longer held, is simply not used.
    if (prev_batch_id != 0)
        for (auto &b : ctx->batch) if (b.id == prev_batch_id) P = &b;
    if (B->width != ctx->width || B->height != ctx->height || B->out_format != ctx->out_format)
        return fail(ctx, VK_ERR_INVALID, "vk_untile_batch: the backbuffer changed shape or format since that batch was dealt");
    if (B->wire != ctx->wire) return fail(ctx, VK_ERR_INVALID, "vk_untile_batch: the wire format changed since that batch was dealt (vk_partition_wire)");
    if (P && (P->n_frames != B->n_frames || P->n_tiles != B->n_tiles || P->ts != B->ts || P->width != B->width || P->height != B->height ||
              P->out_format != B->out_format || !P->d)) P = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t tx = (ctx->width + B->ts - 1) / B->ts;
    const uint32_t chunks = (B->ts * B->ts + 511u) / 512u;
    const uint64_t blocks = (uint64_t)B->n_frames * B->n_tiles * chunks;
    if (blocks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "vk_untile_batch: too many tiles for one launch");
    const FrameDesc *frames = reinterpret_cast<const FrameDesc *>(B->d);
    const uint32_t *pos = reinterpret_cast<const uint32_t *>(B->d + (size_t)B->n_frames * sizeof(FrameDesc)) + (size_t)B->n_frames * B->n_tiles;
    // FrameDesc::n_active of a compact batch is the frame's active tile count (what the gather carried)
    const FrameDesc *pframes = P ? reinterpret_cast<const FrameDesc *>(P->d) : nullptr;
    const uint32_t *ppos = P ? reinterpret_cast<const uint32_t *>(P->d + (size_t)P->n_frames * sizeof(FrameDesc)) + (size_t)P->n_frames * P->n_tiles : nullptr;
    if (ctx->out_format == VK_OUT_RGBA16F)
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA16F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, out_frames, ctx->width, ctx->height, B->ts, tx, B->n_tiles, B->nranks, n_slots, B->n_frames, pos, frames, B->root_skip, 0u, ppos, pframes, (uint32_t)B->wire);
    else
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA32F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, out_frames, ctx->width, ctx->height, B->ts, tx, B->n_tiles, B->nranks, n_slots, B->n_frames, pos, frames, B->root_skip, 0u, ppos, pframes, (uint32_t)B->wire);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(B->ev, ctx->stream));
    if (P) HIP_TRY(ctx, hipEventRecord(P->ev, ctx->stream));  // (its tables were read too: the slot is reused after this launch)
    return VK_OK;
}

// Device buffers for hosts without another allocator (the C++ host, a plain C consumer): frame batches, gather buffers.
int vk_device_alloc(vk_ctx *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr || bytes == 0) return fail(ctx, VK_ERR_INVALID, "vk_device_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *ptr = nullptr;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP, std::string("vk_device_alloc: ") + hipGetErrorString(e));
    return VK_OK;
}

int vk_device_free(vk_ctx *ctx, void *ptr) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ptr));
    return VK_OK;
}

int vk_device_download(vk_ctx *ctx, void *dst_host, const void *src_device, size_t bytes) {
    if (!ctx || !dst_host || !src_device) return fail(ctx, VK_ERR_INVALID, "vk_device_download: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}


// ---- multi-GPU: RCCL over xGMI behind the C-ABI ------------------------------------

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
    std::string err;
};

extern "C" {

int vk_group_create(int n, const int *ordinals, vk_group **out) {
    if (!out) return fail(nullptr, VK_ERR_INVALID, "vk_group_create: out is NULL");
    *out = nullptr;
    if (n <= 0 || n > 64 || !ordinals) return fail(nullptr, VK_ERR_INVALID, "vk_group_create: 1..64 device ordinals");
    if (n > 1 && !rccl_load()) return fail(nullptr, VK_ERR_UNSUPPORTED, g_rccl.err);
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
    if (n > 1) {
        std::vector<ncclComm_t> comms(n);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, ordinals);
        if (r != ncclSuccess) return bail(VK_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
        for (int i = 0; i < n; i++) { g->ctx[i]->comm = comms[i]; g->ctx[i]->comm_rank = i; g->ctx[i]->comm_size = n; g->ctx[i]->comm_owned = true; }
    }
    g->send.assign(n, nullptr);
    *out = g;
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
    for (vk_ctx *c : g->ctx) (void)vk_ctx_destroy(c);
    delete g;
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

// ---- present + capture (next rows N1, N2) ------------------------------------------------------

int vk_present(vk_ctx *ctx, uint32_t width, uint32_t height, int also_bgra) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "vk_present: no backbuffer");
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, VK_ERR_INVALID, "vk_present: size must be in [1, 32768]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (width != ctx->present_w || height != ctx->present_h || (also_bgra && !ctx->bgra8)) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->rgba8) (void)hipFree(ctx->rgba8);
        if (ctx->bgra8) (void)hipFree(ctx->bgra8);
        ctx->rgba8 = ctx->bgra8 = nullptr;
        ctx->present_w = ctx->present_h = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->rgba8, (size_t)width * height * 4));
        if (also_bgra) HIP_TRY(ctx, hipMalloc(&ctx->bgra8, (size_t)width * height * 4));
        ctx->present_w = width; ctx->present_h = height;
    }
    const uint64_t n = (uint64_t)width * height;
    hipLaunchKernelGGL(present_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->backbuffer,
                       ctx->out_format == VK_OUT_RGBA16F ? OUT_RGBA16F : OUT_RGBA32F, ctx->width, ctx->height, width, height,
                       ctx->rgba8, also_bgra ? ctx->bgra8 : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return VK_OK;
}

int vk_capture_frame(vk_ctx *ctx, void *dst, size_t dst_bytes, uint32_t *out_width, uint32_t *out_height,
                     uint32_t *out_padded_bytes_per_row) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->rgba8) return fail(ctx, VK_ERR_INVALID, "vk_capture_frame: nothing presented yet (vk_present)");
    // ImageDimentions::new(w, h, 256): even-rounded size, rows padded to 256 B (src/utils/mod.rs:99-113)
    const uint32_t w = ctx->present_w - (ctx->present_w % 2), h = ctx->present_h - (ctx->present_h % 2);
    const uint32_t unpadded = w * 4, padded = unpadded + (256 - unpadded % 256) % 256;
    if (out_width) *out_width = w;
    if (out_height) *out_height = h;
    if (out_padded_bytes_per_row) *out_padded_bytes_per_row = padded;
    if (!dst) return VK_OK;  // size query
    if (dst_bytes < (size_t)padded * h) return fail(ctx, VK_ERR_INVALID, "vk_capture_frame: destination smaller than padded_bytes_per_row * height");
    if (w == 0 || h == 0) return VK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::memset(dst, 0, (size_t)padded * h);
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, padded, ctx->rgba8, (size_t)ctx->present_w * 4, unpadded, h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

// ---- results ------------------------------------------------------------------------------------

int vk_readback(vk_ctx *ctx, void *dst, size_t row_pitch_bytes) {
    if (!ctx || !dst) return fail(ctx, VK_ERR_INVALID, "vk_readback: NULL argument");
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "no backbuffer");
    const size_t row = (size_t)ctx->width * px_bytes(ctx->out_format);
    if (row_pitch_bytes < row) return fail(ctx, VK_ERR_INVALID, "row pitch smaller than a row");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, row_pitch_bytes, ctx->backbuffer, row, row, ctx->height, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_step_counts(vk_ctx *ctx, uint64_t *s_ref, uint64_t *s_sampled) {
    if (!ctx) return VK_ERR_INVALID;
    unsigned long long h[2] = {0, 0};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (s_ref) *s_ref = h[0];
    if (s_sampled) *s_sampled = h[1];
    return VK_OK;
}

int vk_simt_census(vk_ctx *ctx, uint64_t out[4]) {
    if (!ctx || !out) return fail(ctx, VK_ERR_INVALID, "vk_simt_census: NULL argument");
    unsigned long long h[8] = {0};
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) out[i] = h[2 + i];
    return VK_OK;
}

int vk_debug_wave_trace(vk_ctx *ctx, int enable, uint64_t *out, size_t n_blocks) {
    if (!ctx) return VK_ERR_INVALID;
    ctx->want_trace = enable != 0;
    if (!out) return VK_OK;
    if (!ctx->trace || n_blocks > ctx->trace_blocks) return fail(ctx, VK_ERR_INVALID, "vk_debug_wave_trace: no trace of that size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->trace, n_blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VK_OK;
}

int vk_debug_set_param(vk_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return VK_ERR_INVALID;
    const std::string n(name);
    if (n == "stage_cap_bytes") ctx->stage_cap_bytes = (uint32_t)value;          // LDS window of the staged march (next render)
    else if (n == "stage_slab_cells") ctx->stage_slab_cells = (uint32_t)value;   // cells per slab along the major axis (next render)
    else if (n == "trip_log_cap") { if (value < 0 || value > 4096 || ((uint32_t)value & 7u)) return fail(ctx, VK_ERR_INVALID, "trip_log_cap: a multiple of 8 up to 4096"); ctx->trip_log_cap = (uint32_t)value; }
    else if (n == "stage_group") ctx->stage_group = (uint32_t)value;             // staged march: windows shared by the four waves of a group (tools/staged_group.py)
    else if (n == "frame_runs") ctx->frame_runs = (uint32_t)value;               // batched launches: runs of consecutive frames per XCD (tools/frame_runs.py)
    else if (n == "stage_grow_every") ctx->stage_grow_every = (uint32_t)value;   // slab search growth period (next render)
    else if (n == "stage_row_pad") ctx->stage_row_pad = (uint32_t)value;          // odd row pitch of the staged window (next render)
    else if (n == "wave_prio") ctx->wave_prio = (uint32_t)value;
    else if (n == "walk_cap") ctx->walk_cap = (uint32_t)value;
    else if (n == "walk_cap_all") ctx->walk_cap_all = (uint32_t)value;
    else if (n == "order_rays") { ctx->order_rays = ctx->order_rays_batch = (uint32_t)std::min<double>(std::max<double>(value, 1), 8); ctx->batch_key.clear(); ctx->order_key.clear(); }
    else if (n == "naive_lds_pad") ctx->naive_lds_pad = (uint32_t)value;          // experiments: caps the cell kernels' waves per SIMD
    else if (n == "stage_copies_mask") ctx->stage_copies_mask = (uint32_t)value; // which brick copies to build (next upload)
    else return fail(ctx, VK_ERR_INVALID, "vk_debug_set_param: unknown parameter " + n);
    return VK_OK;
}

int vk_step_counts_reset(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return VK_OK;
}

int vk_readback_steps(vk_ctx *ctx, uint32_t *dst) {
    if (!ctx || !dst) return fail(ctx, VK_ERR_INVALID, "vk_readback_steps: NULL argument");
    if (!ctx->steps) return fail(ctx, VK_ERR_INVALID, "no VK_RENDER_COUNT launch since the last resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->steps, (size_t)ctx->width * ctx->height * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_timer_begin(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    ctx->timing_open = true;
    ctx->timing_done = false;
    return VK_OK;
}

int vk_timer_end(vk_ctx *ctx) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->timing_open) return fail(ctx, VK_ERR_INVALID, "vk_timer_end without vk_timer_begin");
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timing_open = false;
    ctx->timing_done = true;
    return VK_OK;
}

int vk_timer_elapsed_ms(vk_ctx *ctx, float *ms) {
    if (!ctx || !ms) return fail(ctx, VK_ERR_INVALID, "vk_timer_elapsed_ms: NULL argument");
    if (!ctx->timing_done) return fail(ctx, VK_ERR_INVALID, "no completed timer bracket");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VK_OK;
}

}  // extern "C"
