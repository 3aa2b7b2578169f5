// vk_order.hip -- host-side geometry of a frame: the cube's screen rectangle and silhouette, the heaviest-first tile order
// (launch order at N = 1, the deal over ranks at N > 1; the reference's tile loop is examples/xor/main.rs:77-95,235-254) and the
// device ring the order tables travel through.  vk_partition_* / vk_tiles_active.
#include "vk_ctx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

// The geometry itself -- cull rectangle, silhouette hull, tile_is_inactive, the heaviest-first order -- lives in vk_hostmath.hpp (pure C++:
// tests/hostmath_fuzz.cpp runs it under -fsanitize=address,undefined); these are the context-shaped entry points the other TUs call.
static_assert(vk::kModeNaive == VK_MODE_NAIVE_TRILINEAR, "vk_hostmath.hpp's mode constant");
void cull_rect_cam(const vk_ctx *ctx, const float *cam, int mode, int32_t r[4]) { vk::cull_rect_wh(ctx->width, ctx->height, cam, mode, r); }

void compute_tile_order_raw(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                            uint32_t *order, uint32_t *order_pos, uint32_t &order_active, int G) {
    const uint32_t dims[3] = {ctx->nx, ctx->ny, ctx->nz};
    vk::tile_order(ctx->width, ctx->height, dims, cam, mode, ox, oy, rw, rh, ts, order, order_pos, order_active, G);
}

static void compute_tile_order(const vk_ctx *ctx, const float *cam, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts,
                               std::vector<uint32_t> &order, std::vector<uint32_t> &order_pos, uint32_t &order_active) {
    const size_t n = (size_t)((rw + ts - 1) / ts) * ((rh + ts - 1) / ts);
    order.resize(n); order_pos.resize(n);
    compute_tile_order_raw(ctx, cam, mode, ox, oy, rw, rh, ts, order.data(), order_pos.data(), order_active, (int)ctx->order_rays);
}

// Every caller of tile_order_update goes on to launch a reader of the current slot on ctx->stream: who that is, for the day
// the ring comes round to the slot again.
static void order_note_use(vk_ctx *ctx) {
    if (ctx->ring_slot < 0) return;
    ctx->ring_use_stream[ctx->ring_slot] = ctx->stream;
    ctx->ring_use_frame[ctx->ring_slot] = ctx->fif_open ? ctx->fif[ctx->fif_cur].id : 0;
}

// The order of the context's camera for a region: computed on the host when the key (camera, region, frame and volume shape) changes, and
// -- need_device -- uploaded to the next slot of the device ring.  A whole-pixel single-frame launch of up to kOrderInline tiles carries the
// order in its kernel arguments instead (LaunchDesc::order_inline) and asks for no upload: the copy was a blit kernel of ~10 us in the
// frame's own stream, on a frame of 140.
int tile_order_update(vk_ctx *ctx, int mode, int32_t ox, int32_t oy, uint32_t rw, uint32_t rh, uint32_t ts, bool need_device) {
    const uint32_t tx = (rw + ts - 1) / ts, ty = (rh + ts - 1) / ts;
    const size_t n = (size_t)tx * ty;
    std::vector<unsigned char> key(144 + 40);
    std::memcpy(key.data(), ctx->camera, 144);
    const uint32_t kk[10] = {(uint32_t)mode, (uint32_t)ox, (uint32_t)oy, rw, rh, ts, ctx->width, ctx->height, ctx->nx ^ (ctx->ny << 10) ^ (ctx->nz << 20), 0};
    std::memcpy(key.data() + 144, kk, 40);
    if (!(key == ctx->order_key && ctx->order.size() == n)) {
        compute_tile_order(ctx, ctx->camera, mode, ox, oy, rw, rh, ts, ctx->order, ctx->order_pos, ctx->order_active);
        ctx->order_key = key;
        ctx->order_on_device = false;
    }
    if (!need_device) return VK_OK;
    return order_ensure_device(ctx);
}

// The host's current order on the device (a no-op when it is there already).
int order_ensure_device(vk_ctx *ctx) {
    if (ctx->order_on_device) { order_note_use(ctx); return VK_OK; }
    const size_t n = ctx->order.size();
    if (n == 0) return fail(ctx, VK_ERR_INVALID, "no tile order yet");
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
    // device slot itself is safe to overwrite: on ONE stream the kernels that read the slot kOrderRing cameras ago were
    // enqueued before this copy (vk_ctx_set_stream drains the old stream first), so the copy is stream-ordered after them;
    // with frames in flight (several streams) the slot's last reader may belong to a frame on ANOTHER stream -- a frame that
    // draws its tiles one launch at a time (the xor example's 3 x 6 tile loop) goes round the ring inside every frame: this
    // stream then waits for that frame's end (nothing to wait for when vk_frame_begin has since retaken its surface: it
    // waited on the host then).
    if (ctx->ring_ev[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->ring_ev[slot]));
    else HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_ev[slot], hipEventDisableTiming));
    if (ctx->fif_k > 1 && ctx->ring_use_stream[slot] && ctx->ring_use_stream[slot] != ctx->stream) {
        bool ordered = false;
        const uint64_t f = ctx->ring_use_frame[slot];
        if (f != 0) {
            const vk_ctx::FrameSlot *fs = nullptr;
            for (uint32_t i = 0; i < ctx->fif_k; i++) if (ctx->fif[i].id == f) fs = &ctx->fif[i];
            if (!fs) ordered = true;
            else if (fs->ended) {
                if (hipEventQuery(fs->done) != hipSuccess) { (void)hipGetLastError(); HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, fs->done, 0)); }
                ordered = true;
            }
        }
        if (!ordered) {  // a reader outside any frame: order this stream after everything its stream has been given so far
            if (!ctx->ring_guard_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ring_guard_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ctx->ring_guard_ev, ctx->ring_use_stream[slot]));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_guard_ev, 0));
        }
    }
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
    ctx->order_on_device = true;
    order_note_use(ctx);
    return VK_OK;
}

// A launch on a stream other than the one that uploaded the current order slot waits for that upload.
int order_wait(vk_ctx *ctx) {
    const int s = ctx->ring_slot;
    if (s < 0 || ctx->ring_done[s] || ctx->stream == ctx->ring_stream[s]) return VK_OK;
    if (hipEventQuery(ctx->ring_ev[s]) == hipSuccess) { ctx->ring_done[s] = true; return VK_OK; }
    (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error for the launch that follows to find)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ring_ev[s], 0));
    return VK_OK;
}

extern "C" {

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

int vk_partition_active(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t nranks, uint32_t *n_active_tiles, uint32_t *n_active_slots) {
    if (!ctx) return VK_ERR_INVALID;
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_active: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u) || nranks == 0) return fail(ctx, VK_ERR_INVALID, "bad tile size / nranks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the same geometry key as the render calls use (PROCEDURAL marches the compute twin's rays)
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size, false);
    if (orc) return orc;
    if (n_active_tiles) *n_active_tiles = ctx->order_active;
    if (n_active_slots) *n_active_slots = deal_rounds(ctx->order_active, nranks, nranks > 1 ? ctx->root_skip : 0u);
    return VK_OK;
}

int vk_debug_set_tile_order(vk_ctx *ctx, const uint32_t *order, uint32_t n) {
    // experiment hook: replace the current (already computed) order table; stays until the key changes
    if (!ctx || !order) return VK_ERR_INVALID;
    if (n != ctx->order.size()) return fail(ctx, VK_ERR_INVALID, "vk_debug_set_tile_order: no order of that size");
    {
        std::vector<char> seen(n, 0);
        for (uint32_t q = 0; q < n; q++) {
            if (order[q] >= n || seen[order[q]]) return fail(ctx, VK_ERR_INVALID, "vk_debug_set_tile_order: not a permutation of the tiles");
            seen[order[q]] = 1;
            // only the leading order_active positions are marched, the tiles behind them are cleared: the active tiles stay in front
            if ((q < ctx->order_active) != (ctx->order_pos[order[q]] < ctx->order_active))
                return fail(ctx, VK_ERR_INVALID, "vk_debug_set_tile_order: the active tiles (vk_partition_active) must stay the leading ones");
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (uint32_t q = 0; q < n; q++) { ctx->order[q] = order[q]; ctx->order_pos[order[q]] = q; }
    if (ctx->order_on_device) {
        HIP_TRY(ctx, hipMemcpy(ctx->d_order, ctx->order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->d_order_pos, ctx->order_pos.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return VK_OK;
}

int vk_partition_order(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t *order_out, uint32_t n_tiles) {
    if (!ctx || !order_out) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: NULL argument");
    if (mode != VK_MODE_NAIVE_TRILINEAR && mode != VK_MODE_COMPUTE_NEAREST && mode != VK_MODE_PROCEDURAL) return fail(ctx, VK_ERR_INVALID, "unknown mode");
    if (!ctx->backbuffer || !ctx->have_camera || (ctx->format < 0 && mode != VK_MODE_PROCEDURAL))
        return fail(ctx, VK_ERR_INVALID, "vk_partition_order: needs camera and backbuffer (and a volume, except PROCEDURAL)");
    if (tile_size == 0 || (tile_size & 7u)) return fail(ctx, VK_ERR_INVALID, "tile size must be a multiple of 8");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int orc = tile_order_update(ctx, mode == VK_MODE_PROCEDURAL ? VK_MODE_COMPUTE_NEAREST : mode, 0, 0, ctx->width, ctx->height, tile_size, false);
    if (orc) return orc;
    if (n_tiles != ctx->order.size()) return fail(ctx, VK_ERR_INVALID, "vk_partition_order: n_tiles does not match the partition");
    std::memcpy(order_out, ctx->order.data(), n_tiles * sizeof(uint32_t));
    return VK_OK;
}

}  // extern "C"
