// vk_render.hip -- vk_render / vk_render_partition: one raycast pass into the backbuffer (the reference's
// RaycastPipeline::record, examples/bonsai/raycast.rs / examples/xor/raycast.rs) -- argument checks, LaunchDesc, kernel choice.
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

// Argument and state checks shared by every render entry point; `cam` is the 144-byte camera the frame uses.
// The kernels' policy switches (vk_common.hpp: LaunchFlag) from the caller's VK_RENDER_* flags and the context's knobs.
uint32_t launch_flags(const vk_ctx *ctx, uint32_t render_flags, bool batch) {
    uint32_t f = 0;
    if (render_flags & VK_RENDER_DEBUG_TRIPS) f |= LF_STEPS_ARE_TRIPS;
    if (render_flags & VK_RENDER_DEBUG_FALLBACK) f |= LF_STEPS_ARE_FALLBACKS;
    if (!(render_flags & VK_RENDER_PROBE_ALWAYS)) f |= LF_ADAPTIVE_PROBING;
    if (ctx->wave_prio) f |= LF_WAVE_PRIORITY;
    if (batch && ctx->frame_runs) f |= LF_FRAME_RUNS;
    // probe ahead pays where a frame's waves run against their own dependent chains -- a lone single-frame launch (-11 %) -- and costs where the
    // machine is full (+6-8 % in batches).  A frame of a ring in which three frames execute at once (k = 4) is in the second case: C2 0.0768 ->
    // 0.0724 ms per frame with the leaner kernel there, 0.080 -> 0.088 at k = 3 where two execute (profiles/r06_frames_in_flight.txt)
    const bool crowded = frames_crowded(ctx);
    if (ctx->probe_ahead == 1u || (ctx->probe_ahead == 2u && !batch && !crowded)) f |= LF_PROBE_AHEAD;  // (dispatch_march drops it for dt_scale > 1.25)
    return f;
}

int check_render(vk_ctx *ctx, int mode, const float *cam, float dt_scale, uint32_t ts, uint32_t rank, uint32_t nranks) {
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
int dispatch_march(vk_ctx *ctx, int mode, const LaunchDesc &L_in, uint32_t flags, const float *reach_cam) {
    LaunchDesc L = L_in;
    // probe-ahead requests the distance byte of a position one step past a ray's end: inside the index tables' padding only while a step is
    // at most ~1.5 cells (vk_march.hpp: locate)
    if (!(L.dt_scale <= 1.25f)) L.flags &= ~(uint32_t)LF_PROBE_AHEAD;
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
        if ((flags & VK_RENDER_DEVICE_SINE) && count) return fail(ctx, VK_ERR_INVALID, "VK_RENDER_DEVICE_SINE is a tolerance mode: count steps with the specified sine");
        launch_procedural(ctx, L, grid, count, time, (flags & VK_RENDER_DEVICE_SINE) != 0);
    } else if (mode == VK_MODE_COMPUTE_NEAREST) {
        launch_compute(ctx, L, V, grid, count, ctx->vol_kind == VOL_PAIRB, !(flags & VK_RENDER_NO_SKIP));  // bricked 16-byte records, or the two dense volumes (the literal twin)
    } else {
        // Skipping costs a distance lookup per probing trip; it only pays when there is something to skip
        // (docs/archive/tools/skip_crossover.py, DESIGN.md section 4: on 256^3 volumes with a share e of exactly-transparent cells
        // the skip kernel overtakes the dense one between e = 0.36 and e = 0.56: 0.335 / 0.348 / 0.407 ms for dense /
        // adaptive / probing always at e = 0.36, 0.332 / 0.302 / 0.292 at e = 0.56).  Default policy by the census taken
        // at upload:  e < 0.45: the dense kernel;  0.45 <= e < 0.55: the skip kernel with adaptive probing (dense
        // stretches where nothing is being skipped);  e >= 0.55: the skip kernel probing on every trip (emptier
        // volumes -- the bonsai stand-in is at 0.77 -- spend their time in the skip walks, and the stretches only cost).
        // VK_RENDER_FORCE_SKIP takes the skip kernel whatever the census, adaptive unless VK_RENDER_PROBE_ALWAYS.
        const bool forced = (flags & VK_RENDER_FORCE_SKIP) != 0;
        const bool skip = !(flags & VK_RENDER_NO_SKIP) && (forced || ctx->empty_fraction >= 0.45);
        if (skip && !forced && ctx->empty_fraction >= 0.55) L.flags &= ~(uint32_t)LF_ADAPTIVE_PROBING;
        if (skip && ctx->empty_fraction < 0.05) L.flags |= LF_LONG_STRETCHES;  // forced on (almost) solid material: long dense stretches from the start
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
        if (ctx->vol_kind == VOL_S8U8 || ctx->vol_kind == VOL_S8F16) launch_staged(ctx, L, V, grid, count, reach_cam);
        else launch_cells(ctx, L, V, grid, count, skip, safe, (flags & VK_RENDER_FAST_WALK) ? 2 : 0);  // (vk_march.hpp: WalkKind)
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
    constexpr uint32_t kPresentFlags = VK_RENDER_PRESENT | VK_RENDER_PRESENT_BGRA | VK_RENDER_PRESENT_ONLY;
    if ((flags & kPresentFlags) && compact_out) return fail(ctx, VK_ERR_INVALID, "VK_RENDER_PRESENT*: whole-pixel passes into the backbuffer only (vk_render)");
    if (rw == 0 || rh == 0) return VK_OK;  // empty tile
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (flags & kPresentFlags) {  // the present target at the backbuffer's own size, as vk_present(width, height) would make it
        int prc = present_targets(ctx, ctx->width, ctx->height, (flags & VK_RENDER_PRESENT_BGRA) != 0);
        if (prc) return prc;
    }
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
        cull_rect_cam(ctx, ctx->camera, geo_mode, cr);
        L.cull_x0 = cr[0]; L.cull_y0 = cr[1]; L.cull_x1 = cr[2]; L.cull_y1 = cr[3];
    }
    L.rank = rank; L.nranks = nranks;
    // whole-pixel launches of few tiles (C2 at 64-pixel tiles: 510) carry the order in their arguments; partitions and large tile counts
    // read the device table (the partition's un-tile needs it there anyway)
    const bool order_inline = !compact_out && (uint64_t)L.tiles_x * L.tiles_y <= kOrderInline && !ctx->order_never_inline;
    {
        int orc = tile_order_update(ctx, geo_mode, ox, oy, rw, rh, ts, !order_inline);
        if (orc) return orc;
        if (order_inline) {
            for (size_t q = 0; q < ctx->order.size(); q++) L.order_inline[q] = (uint16_t)ctx->order[q];
            L.tile_order = nullptr;
        } else {
            orc = order_wait(ctx);
            if (orc) return orc;
            L.tile_order = ctx->d_order;
        }
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
    L.flags = launch_flags(ctx, flags, false) | (order_inline ? (uint32_t)LF_ORDER_INLINE : 0u);
    L.walk_cap = ctx->walk_cap ? (float)ctx->walk_cap : HUGE_VALF;
    L.walk_cap_all = ctx->walk_cap_all ? (float)ctx->walk_cap_all : HUGE_VALF;
    L.pair_walk_min = ctx->pair_walk_min;
    if (flags & kPresentFlags) {
        L.present_rgba8 = ctx->rgba8;
        L.present_bgra8 = (flags & VK_RENDER_PRESENT_BGRA) ? ctx->bgra8 : nullptr;
        L.present_only = (flags & VK_RENDER_PRESENT_ONLY) ? 1u : 0u;
    }
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
        if (ctx->trip_log_cap) { L.flags |= LF_TRIP_LOG; L.trip_log_cap = ctx->trip_log_cap; }
        else for (uint64_t i = 0; i < n_blocks; i++) init[4 * i] = ~0ull;
        HIP_TRY(ctx, hipMemcpy(ctx->trace, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        L.trace = ctx->trace;
    }
    return dispatch_march(ctx, mode, L, flags, ctx->camera);
}

extern "C" {

int vk_render(vk_ctx *ctx, int mode, int32_t tile_x, int32_t tile_y, uint32_t tile_w, uint32_t tile_h, float dt_scale,
              uint32_t flags) {
    // tiles of the launch order: 64 x 64 pixels; 32 x 32 for the staged march, whose waves are few and long (C4 1.80 -> 1.68 ms, C5 9.40 -> 9.02;
    // C2 and the compute twin lose 0.5 - 4 % at 32)
    const bool staged = ctx && (ctx->vol_kind == vk::VOL_S8U8 || ctx->vol_kind == vk::VOL_S8F16) && mode == VK_MODE_NAIVE_TRILINEAR;
    return render_common(ctx, mode, tile_x, tile_y, tile_w, tile_h, ctx && ctx->render_tile ? ctx->render_tile : (staged ? 32u : 64u), 0, 1, dt_scale, flags, nullptr);
}

int vk_render_partition(vk_ctx *ctx, int mode, uint32_t tile_size, uint32_t rank, uint32_t nranks, float dt_scale,
                        uint32_t flags, void *compact_out) {
    if (!ctx) return VK_ERR_INVALID;
    if (!compact_out) return fail(ctx, VK_ERR_INVALID, "vk_render_partition: compact_out is NULL");
    return render_common(ctx, mode, 0, 0, ctx->width, ctx->height, tile_size, rank, nranks, dt_scale, flags, compact_out);
}

}  // extern "C"
