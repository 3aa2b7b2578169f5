// vk_batch.hip -- vk_render_batch: B frames, each with its own camera and tile order, in ONE launch (the reference keeps frames
// in flight through its queue, src/lib.rs:178-194).
#include "vk_ctx.hpp"

#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

extern "C" {

// ---- batched launches -----------------------------------------------------------------------------

int vk_render_batch(vk_ctx *ctx, int mode, uint32_t n_frames, const void *cameras, uint32_t tile_size, uint32_t rank, uint32_t nranks,
                    float dt_scale, uint32_t flags, void *out, int compact, uint32_t slot_capacity, uint32_t *batch_id, uint32_t *n_active_slots) {
    if (!ctx) return VK_ERR_INVALID;
    if (!cameras || !out) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: NULL argument");
    if (n_frames == 0 || n_frames > VK_MAX_BATCH_FRAMES) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: 1..1024 frames per batch");
    if (flags & VK_RENDER_COUNT) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: the step counters describe one frame; count with vk_render");
    if (flags & (VK_RENDER_PRESENT | VK_RENDER_PRESENT_BGRA | VK_RENDER_PRESENT_ONLY)) return fail(ctx, VK_ERR_INVALID, "vk_render_batch: the fused present belongs to the one-frame pass (vk_render)");
    // (compact == 0 with nranks > 1: this rank's tiles at their place in whole frames that live elsewhere -- the root's, over xGMI: vk_group_peer_direct)
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
    const size_t bytes = batch_table_bytes(n_frames, n_tiles, sizeof(FrameDesc));  // (vk_hostmath.hpp: FrameDesc[B] | order[B][n_tiles] | pos[B][n_tiles])
    vk_ctx::BatchSlot &B = ctx->batch[ctx->batch_seq % 4u];
    if (B.ev) HIP_TRY(ctx, hipEventSynchronize(B.ev));  // the launches of four batches ago have long finished
    else HIP_TRY(ctx, hipEventCreateWithFlags(&B.ev, hipEventDisableTiming));
    if (!ctx->batch_retired.empty()) {
        // outgrown table blocks: free them once nothing in flight can still read them, i.e. when every slot's last launch has completed
        bool idle = true;
        for (auto &b : ctx->batch) if (b.ev && hipEventQuery(b.ev) != hipSuccess) idle = false;
        (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error to find after the launch below)
        if (idle) { for (auto &r : ctx->batch_retired) { (void)hipFree(r.first); (void)hipHostFree(r.second); } ctx->batch_retired.clear(); }
    }
    B.id = 0;  // claimed: whatever fails below, the slot no longer answers to its old id
    if (B.cap < bytes) {
        // sized for 256 frames from the start and doubled from there; the outgrown blocks are retired, not freed (hipFree and
        // hipHostFree synchronise the device: a driver whose batches grow -- 20, then 32 frames -- stalled four launches each time)
        if (B.d || B.h) ctx->batch_retired.emplace_back(B.d, B.h);
        B.d = B.h = nullptr;
        // (the 256-frame floor is for drivers whose batches grow; it is capped at 4 MiB so that a one-frame batch at a small tile size --
        // ts = 8 at 1080p is 32 400 tiles -- does not pin 66 MB per slot)
        const size_t floor256 = std::min<size_t>(batch_table_bytes(256u, n_tiles, sizeof(FrameDesc)), (size_t)4 << 20);
        const size_t want = std::max({bytes, 2 * B.cap, floor256});
        B.cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&B.d, want));
        HIP_TRY(ctx, hipHostMalloc((void **)&B.h, want));
        B.cap = want;
    }
    FrameDesc *fd = reinterpret_cast<FrameDesc *>(B.h);
    uint32_t *h_order = reinterpret_cast<uint32_t *>(B.h + batch_order_offset(n_frames, sizeof(FrameDesc)));
    uint32_t *h_pos = h_order + (size_t)n_frames * n_tiles;
    uint32_t max_active = 0, min_active = 0xffffffffu;
    // The tile order depends on the camera (and the frame / volume shape).  It is written straight into the pinned
    // staging block; a frame with the camera of the frame before it (or of the last frame of the previous batch) copies
    // that frame's tables instead of casting the estimate rays again.
    std::vector<uint32_t> &order = ctx->batch_order, &pos = ctx->batch_pos;
    // Estimate rays per tile: the 3 x 3 grid of the single-frame launches, or just the tile's centre ray when the grid
    // spans >= 4 frames -- position-major over many frames the launch time no longer depends on the finer estimate
    // (docs/archive/tools/order_rays.py) and the host's share drops from 40 to 10 us per camera, which is what an orbiting camera
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
            // (nothing may unwind through the C-ABI: a thread that cannot be created -- EAGAIN, a pid limit -- leaves its share to this one)
            std::vector<std::thread> pool;
            size_t started = 1;
            try {
                for (; started < n_thr; started++) pool.emplace_back(work, n_own * started / n_thr, n_own * (started + 1) / n_thr);
            } catch (...) {}
            work(0, n_own / n_thr);
            if (started < n_thr) work(n_own * started / n_thr, n_own);
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
    // (of a frame whose tiles several ranks write, rank 0 clears)
    const uint32_t clear_max_inactive = (!compact && rank == 0 && geo_mode == VK_MODE_NAIVE_TRILINEAR && min_active < n_tiles) ? (uint32_t)n_tiles - min_active : 0u;
    if (slots == 0 && clear_max_inactive == 0) { HIP_TRY(ctx, hipEventRecord(B.ev, ctx->stream)); return VK_OK; }
    LaunchDesc L{};
    L.clear_max_inactive = clear_max_inactive;
    L.W = ctx->width; L.H = ctx->height;
    L.ox = 0; L.oy = 0; L.rw = ctx->width; L.rh = ctx->height;
    L.ts = ts; L.tiles_x = tx; L.tiles_y = ty;
    L.rank = rank; L.nranks = nranks; L.root_skip = root_skip;
    L.tile_order = reinterpret_cast<const uint32_t *>(B.d + batch_order_offset(n_frames, sizeof(FrameDesc)));
    L.n_tiles_launch = 0;
    const uint64_t per_tile = (uint64_t)(ts / 8) * (ts / 8);
    const uint64_t n_blocks = slots * n_frames * per_tile;
    if (n_blocks >= (1ull << 31) - 512) return fail(ctx, VK_ERR_UNSUPPORTED, "launch too large: fewer frames per batch");
    L.n_blocks = (uint32_t)n_blocks;
    L.compact = compact ? 1u + (uint32_t)ctx->wire : 0u;
    L.dt_scale = dt_scale;
    L.out = out;
    L.steps = nullptr; L.counters = nullptr; L.trace = nullptr;
    L.flags = launch_flags(ctx, flags, true);
    L.walk_cap = ctx->walk_cap ? (float)ctx->walk_cap : HUGE_VALF;
    L.walk_cap_all = ctx->walk_cap_all ? (float)ctx->walk_cap_all : HUGE_VALF;
    L.pair_walk_min = ctx->pair_walk_min;
    L.frames = reinterpret_cast<const FrameDesc *>(B.d);
    L.n_frames = n_frames;
    const int rc = dispatch_march(ctx, mode, L, flags, far_cam);
    HIP_TRY(ctx, hipEventRecord(B.ev, ctx->stream));
    return rc;
}

}  // extern "C"
