// vk_common.hpp -- what every kernel of the vokselis raycast path shares: volume layouts, launch descriptors, the block -> pixel
// map, the output store and the arithmetic that mirrors oracle/vokselis_oracle.c.  Hand-written for gfx950 (CDNA4), wave64.
//
// The WGSL of shaders/raycast_naive.wgsl (fs_main) and shaders/raycast_compute.wgsl (render / get_col2 / single / tile) is
// re-authored as wave64 compute kernels: one lane = one ray, one wave = one 8x8 pixel block.  There is no dense contraction
// here, so no MFMA: the path is gather + VALU.  The opacity path (everything that feeds the loop trip count and the
// alpha >= 0.95 early-out) reproduces the arithmetic specification of the oracle operation for operation; every TU is compiled
// with -ffp-contract=off and fused operations appear only where fmaf is written.
//
// Kernel headers: vk_march.hpp (cell layouts), vk_staged.hpp (8^3 bricks through LDS), vk_compute.hpp (compute twin, C3),
// vk_volume_kernels.hpp (re-layout, generators), vk_post.hpp (clear, un-tile, present).  Each is included by exactly one TU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vk_hostmath.hpp"  // deal_*, block / pixel / record index maps (shared with the host and with tests/hostmath_fuzz.cpp)

namespace vk {

// ---- layouts ---------------------------------------------------------------------------------
// PACKED: the volume is re-laid out as "cells".  Cell (cx,cy,cz), cx in [0, n], stands for the
// trilinear footprint whose low corner is voxel cx-1 (clamp-to-edge applied at build time), and
// stores that footprint's 8 taps contiguously: 8 B (u8) or 16 B (f16), tap b = dx + 2*dy + 4*dz.
// One trilinear sample is therefore ONE aligned 8/16-byte load -- exactly the algorithmic
// B_step of SURVEY 8(d).  Cells are grouped in 4x4x4 bricks (512 B / 1 KiB = 4 / 8 cache lines)
// so a wave's 8x8 ray bundle touches a handful of lines whatever the ray direction.  A u8 map in
// the same cell order (one 64 B line per brick) holds each cell's Chebyshev distance, in cells,
// to the nearest cell that has any tap above the transfer function's zero threshold (0 = this
// cell contributes); it drives exact empty-space skipping.  There are eight such maps, one per
// octant of ray directions: map o only looks at cells AHEAD of a ray of that octant (offset >= 0
// on the axes it moves up, <= 0 on the others), so a ray leaving a surface skips at once instead of
// creeping away from what is behind it.
constexpr int kBrick = 4;
constexpr int kBrickCells = 64;
constexpr int kDistRadius = 24;  // distance map saturates at kDistRadius + 1

// P8: 8 u8 taps (8 B).  P16: u8 volume as 4 x (tap, delta = next_x_tap - tap) f16 pairs (16 B).
// PF16: f16 volume, 8 f16 taps (16 B).
// B9U8 / B9F16: dense voxels in 8^3 bricks stored with a one-voxel apron on the low side (9^3 = 729
// voxels, clamp-to-edge baked in): brick b holds voxels [8b-1, 8b+7] per axis, so the 8 taps of any
// sample come from ONE brick at fixed local offsets (+1, +9, +81).  1.42x the dense bytes instead
// of 8-16x: the layout for volumes far larger than the caches.
// Q8 / QF16 ("quads"): every stored element holds the 2x2 (y, z) neighbourhood of a voxel --
// [v(x,y,z), v(x,y+1,z), v(x,y,z+1), v(x,y+1,z+1)], clamp-to-edge baked in -- so the 8 taps of a sample are
// TWO CONSECUTIVE elements (x and x+1): one 8-byte (u8) or 16-byte (f16) load instead of the four scattered
// x-pair loads of the 9^3 bricks.  Elements live in 9x8x8 bricks (x carries a one-element apron so the pair
// never straddles a brick).  4.5x the dense bytes: for volumes far larger than the caches, where the
// four-load layouts are bound by the texture-address path rather than by HBM.
// PAIRB: the two rgba16f volumes of the compute mode interleaved as 16-byte (density, normals) records in
// 4^3 bricks: one nearest-neighbour step is ONE aligned 16-byte load.
// S8U8 / S8F16: dense 8^3 bricks without apron, staged through LDS by the wave (vk_staged.hpp).
enum VolKind : int { VOL_LINEAR_U8 = 0, VOL_LINEAR_F16 = 1, VOL_P8 = 2, VOL_P16 = 3, VOL_PF16 = 4, VOL_B9U8 = 5, VOL_B9F16 = 6, VOL_PAIRB = 7, VOL_Q8 = 8, VOL_QF16 = 9, VOL_S8U8 = 10, VOL_S8F16 = 11 };
enum OutKind : int { OUT_RGBA32F = 0, OUT_RGBA16F = 1 };

struct VolumeDesc {
    const void *data;     // cells (PACKED) or dense voxels (LINEAR); PAIR: density rgba16f
    const void *data2;    // PAIR: normals rgba16f
    const uint8_t *dist;  // PACKED: per-cell distance maps (same index as the cells), one per ray octant
    uint32_t dist_oct_stride;  // cells between consecutive octant maps; 0: one isotropic map serves all octants
    const uint32_t *lut;       // per-axis cell-index tables of the fast path (cell units with the skip map, byte offsets without)
    uint32_t nx, ny, nz;  // voxel dims
    uint32_t nbx, nby, nbz;  // brick grid dims
    // byte offset of the cell with low-corner voxel (ix,iy,iz), b = i >> 2:
    //   c0 + bx*kx + by*ky + bz*kz + (ix << sh_x) + (iy << sh_y) + (iz << sh_z)
    int64_t kz, c0, max_off;
    int32_t kx, ky;
    uint32_t sh_x, sh_y, sh_z;
};

// One frame of a batched launch (vk_render_batch): its camera, its cull rectangle and its own tile order.
struct FrameDesc {
    float eye[4];
    float inv_proj[16];  // column-major
    int32_t cull_x0, cull_y0, cull_x1, cull_y1;
    uint32_t order_off;  // offset of this frame's order table in LaunchDesc::tile_order
    uint32_t n_active;   // leading positions of that order this launch covers
    uint32_t pad[2];
};
static_assert(sizeof(FrameDesc) == 112, "FrameDesc is read with scalar loads: keep it a multiple of 16 bytes");

struct LaunchDesc {
    float eye[4];
    float inv_proj[16];  // column-major
    uint32_t W, H;       // full image
    int32_t ox, oy;      // region origin in image pixels
    uint32_t rw, rh;     // region size
    uint32_t ts;         // partition tile edge, multiple of 8
    uint32_t tiles_x, tiles_y;
    uint32_t n_tiles_launch;  // leading positions of the order this launch covers
    uint32_t rank, nranks;
    uint32_t root_skip;  // dealing: rank 0 sits out every root_skip-th round (< 2: never)
    uint32_t n_blocks;   // logical 8x8 blocks of this launch
    uint32_t compact;    // 0: output is [H][W] pixels; 1: [slot][ts][ts] pixels; 2: [slot] records of ts*ts (r, g) pairs followed by ts*ts b values (the lean wire format: alpha is 1 in every pixel this path writes)
    float dt_scale;
    const uint32_t *tile_order;  // position in the heaviest-first order -> tile id (row-major)
    int32_t cull_x0, cull_y0, cull_x1, cull_y1;  // pixels outside [x0,x1) x [y0,y1) cannot hit the box
    void *out;
    uint32_t *steps;               // optional per-pixel iteration counts [H][W]
    unsigned long long *counters;  // optional {S_ref, S_sampled, census...}
    uint32_t flags;                // LaunchFlag bits (below)
    uint32_t trip_log_cap;         // LF_TRIP_LOG: u32 entries per wave in `trace`
    unsigned long long *trace;     // optional per-block {start, end, where, work} records (COUNT builds)
    // batched launch: the grid spans n_frames frames, position-major (slot 0 of every frame, then slot 1, ...), so
    // the heaviest tiles of all frames start first.  frames == nullptr: one frame, described by the fields above.
    // Compact output of a batch is position-major too, [slot][frame][ts][ts], so that the tiles a rank has to send
    // (the leading active slots of every frame) are one contiguous prefix; full frames are [frame][H][W].
    const FrameDesc *frames;
    uint32_t n_frames;
    uint32_t n_active_tiles;       // single-frame launches: leading positions of the order that are active (FrameDesc::pad[0] in batches)
    uint32_t grid_march;           // blocks of the grid that march (a multiple of 512); the blocks behind them clear inactive tiles
    uint32_t clear_max_inactive;   // ... of whole-frame batches: inactive tiles per frame at most (0: no such blocks)
    float walk_cap;                // skip kernels: steps a walk may take in a trip in which other lanes sample (+inf: no cap)
    float walk_cap_all;            // ... and in a trip in which every lane walks
    uint32_t pair_walk_min;        // compute twin, record kernel: a run of records that cannot contribute is walked only if it is at least this many steps long
    // Present fused into the pass's epilogue (VK_RENDER_PRESENT; single-frame, whole-pixel launches only): the lane that holds a pixel also
    // writes its presented Rgba8 (and Bgra8) value at the same index -- the present pass at the backbuffer's own size (store_present).
    uint32_t *present_rgba8;       // nullptr: no fused present
    uint32_t *present_bgra8;       // optional surface copy
    uint32_t present_only;         // != 0: the HDR pixel itself is not stored
    // LF_ORDER_INLINE: the tile order rides in the kernel arguments (single-frame launches of up to kOrderInline tiles): no table upload
    // in the frame's stream.  Last member: launches that do not use it never touch these bytes.
    uint16_t order_inline[1024];
};
constexpr uint32_t kOrderInline = 1024;
static_assert(sizeof(LaunchDesc) <= 2560, "LaunchDesc + VolumeDesc + StagedDesc travel as kernel arguments (4 KiB)");

// LaunchDesc::flags.  Policy and instrumentation switches of one launch, set by the host (vk_render.hip) and read by the kernels.
enum LaunchFlag : uint32_t {
    LF_STEPS_ARE_TRIPS = 1u,      // COUNT builds: the per-pixel step image holds march-loop trips instead of reference iterations
    LF_STEPS_ARE_FALLBACKS = 2u,  // COUNT builds, staged kernels: ... holds the steps served from global memory
    LF_ADAPTIVE_PROBING = 4u,     // skip kernels: probe in windows, run dense stretches where nothing is skipped
    LF_LONG_STRETCHES = 8u,       // ... and start with long dense stretches (the census found almost nothing to skip)
    LF_WAVE_PRIORITY = 16u,       // s_setprio by the length of the wave's longest ray (set_wave_priority)
    LF_FRAME_RUNS = 32u,          // batched launches: every XCD marches a run of consecutive frames of a tile position
    LF_PROBE_AHEAD = 128u,        // skip kernels, fast path: the next position's distance byte is requested under this trip's sample (vk_march.hpp: AHEAD)
    LF_ORDER_INLINE = 256u,       // map_pixel reads the tile order from LaunchDesc::order_inline (kernel arguments) instead of the device table
    LF_TRIP_LOG = 64u,            // COUNT builds: `trace` holds per-trip logs of trip_log_cap entries per wave (docs/archive/tools/repack_census.py)
};

// ---- block -> pixels -------------------------------------------------------------------------
// logical_block / group_logical_block (vk_hostmath.hpp): the XCD-aware relabelling of workgroups.  Speed only -- nothing depends on the placement.
// (A persistent variant -- one wave per hardware slot pulling blocks from per-XCD atomic queues --
// balanced the per-SIMD work better (max/mean 1.62 -> 1.45) but lost 0.235 -> 0.35 ms to the
// dequeue round trips of ~32 k mostly trivial blocks: tools/experiments/persistent_workqueue.patch.)

// (Dealing the tiles out SIMD by SIMD -- every SIMD one block from each of 8 tiles -- left the per-SIMD work
// spread at max/mean 1.66: the spread is block-to-block variation inside tiles, not tile placement.
// tools/experiments/simd_interleaved_block_order.patch)

// Issue priority by the length of the wave's longest ray, in quarters of the longest possible march (n / dt_scale
// trips): a frame is one or two rounds of resident waves, and under even sharing of a SIMD's issue slots the longest
// waves -- started first, finished last -- set the frame time while the short ones leave early.  Speed only.
__device__ __forceinline__ void set_wave_priority(bool hit, uint32_t left, float full) {
    const float trips = hit ? (float)left : 0.0f;
    if (__ballot(trips > 0.75f * full)) __builtin_amdgcn_s_setprio(3);
    else if (__ballot(trips > 0.5f * full)) __builtin_amdgcn_s_setprio(2);
    else if (__ballot(trips > 0.25f * full)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

struct PixelMap {
    int32_t x, y;      // image coordinates
    bool valid;        // inside region and image
    size_t out_index;  // pixel index into the output
    uint32_t rec, rec_px;  // compact output: the (slot, frame) record and the pixel inside it (out_index = rec * ts * ts + rec_px)
    uint32_t pos;      // position of the wave's tile in the heaviest-first order (wave-uniform)
};

// The frame a wave belongs to: camera, cull rectangle, order table.  Wave-uniform (scalar loads).
struct FrameView {
    float eye[4];
    float inv_proj[16];
    int32_t cull_x0, cull_y0, cull_x1, cull_y1;
    const uint32_t *order;
    uint32_t n_tiles_launch;
    uint32_t n_active;  // leading positions whose tiles the box's silhouette can reach; the tiles behind them hold only clear colour
    uint32_t frame;
    uint32_t lb;  // the wave's logical block inside its frame
};

__device__ __forceinline__ FrameView frame_view(const LaunchDesc &L, uint32_t lb) {
    FrameView f;
    if (L.frames) {
        const uint32_t sps = L.ts >> 3, per_tile = sps * sps;
        // (slot, frame) pairs, frame fastest; with LF_FRAME_RUNS every XCD marches a run of CONSECUTIVE frames of a tile position: under a moving
        // camera neighbouring frames share almost all of their cells, frames eight apart far fewer (vk_hostmath.hpp: batch_block_split)
        const BlockSplit bs = batch_block_split(lb, per_tile, L.n_frames, (L.flags & LF_FRAME_RUNS) != 0u);
        f.frame = bs.frame;
        f.lb = bs.slot * per_tile + bs.sub;
        const FrameDesc &d = L.frames[f.frame];
#pragma unroll
        for (int i = 0; i < 4; i++) f.eye[i] = d.eye[i];
#pragma unroll
        for (int i = 0; i < 16; i++) f.inv_proj[i] = d.inv_proj[i];
        f.cull_x0 = d.cull_x0; f.cull_y0 = d.cull_y0; f.cull_x1 = d.cull_x1; f.cull_y1 = d.cull_y1;
        f.order = L.tile_order + d.order_off;
        f.n_tiles_launch = d.n_active;
        f.n_active = d.pad[0];
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) f.eye[i] = L.eye[i];
#pragma unroll
        for (int i = 0; i < 16; i++) f.inv_proj[i] = L.inv_proj[i];
        f.cull_x0 = L.cull_x0; f.cull_y0 = L.cull_y0; f.cull_x1 = L.cull_x1; f.cull_y1 = L.cull_y1;
        f.order = L.tile_order;
        f.n_tiles_launch = L.n_tiles_launch;
        f.n_active = L.n_active_tiles;
        f.frame = 0;
        f.lb = lb;
    }
    return f;
}

__device__ __forceinline__ PixelMap map_pixel(const LaunchDesc &L, const FrameView &fv, uint32_t lane) {
    PixelMap m;
    const uint32_t lb = fv.lb;
    uint32_t sps = L.ts >> 3;              // 8x8 blocks per tile edge
    uint32_t per_tile = sps * sps;
    uint32_t slot = lb / per_tile, sub = lb - slot * per_tile;
    uint32_t pos = deal_pos(L.rank, slot, L.nranks, L.root_skip);  // position in the heaviest-first order
    const uint32_t n_tiles = L.tiles_x * L.tiles_y;
    uint32_t tile = pos < fv.n_tiles_launch ? ((L.flags & LF_ORDER_INLINE) ? (uint32_t)L.order_inline[pos] : fv.order[pos]) : n_tiles;
    const TilePixel tp = tile_pixel(L.ts, L.tiles_x, tile, sub, lane);  // (vk_hostmath.hpp)
    const uint32_t lx = tp.lx, ly = tp.ly, rx = tp.rx, ry = tp.ry;
    m.x = L.ox + (int32_t)rx;
    m.y = L.oy + (int32_t)ry;
    m.valid = (tile < n_tiles) && rx < L.rw && ry < L.rh && m.x >= 0 && m.y >= 0 &&
              m.x < (int32_t)L.W && m.y < (int32_t)L.H;
    const uint32_t nf = L.frames ? L.n_frames : 1u;
    m.pos = pos;
    m.rec = compact_record(slot, nf, fv.frame);
    m.rec_px = ly * L.ts + lx;
    m.out_index = L.compact ? compact_pixel_index(m.rec, L.ts, lx, ly) : frame_pixel_index(fv.frame, L.W, L.H, (uint32_t)m.x, (uint32_t)m.y);
    return m;
}

template <int OUT>
__device__ __forceinline__ void store_pixel(void *out, size_t idx, float r, float g, float b, float a) {
    if (OUT == OUT_RGBA32F) {
        reinterpret_cast<float4 *>(out)[idx] = make_float4(r, g, b, a);
    } else {
        // v_cvt_f16_f32 in the default round-to-nearest-even mode (never the pkrtz form)
        union { _Float16 h[4]; uint2 u; } p;
        p.h[0] = (_Float16)r; p.h[1] = (_Float16)g; p.h[2] = (_Float16)b; p.h[3] = (_Float16)a;
        reinterpret_cast<uint2 *>(out)[idx] = p.u;
    }
}

// ---- present pass arithmetic (shaders/present.wgsl:23-35,111-119), shared by present_kernel (vk_post.hpp) and the fused epilogue ----
__device__ __forceinline__ float aces_film(float x) {
    float num = x * (2.51f * x + 0.03f), den = x * (2.43f * x + 0.59f) + 0.14f;
    return fminf(fmaxf(num / den, 0.0f), 1.0f);
}
__device__ __forceinline__ float present_srgb(float c) {
    float sel = ceilf(c - 0.0031308f);
    float under = 12.92f * c;
    float over = 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * 0.41666f) - 0.055f;
    return sel > 0.0f ? over : under;  // mix(under, over, sel) with sel in {0, 1}
}
// (r, g, b, a) sampled from the backbuffer -> Rgba8Unorm / Bgra8Unorm words: ACESFilm + linear_to_srgb on the colour, round to 8 bits
__device__ __forceinline__ void present_pack(const float px[4], uint32_t &rgba, uint32_t &bgra) {
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = px[k];
        if (k < 3) v = present_srgb(aces_film(v));
        c[k] = floorf(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f);
    }
    const uint32_t r = (uint32_t)c[0], g = (uint32_t)c[1], b = (uint32_t)c[2], al = (uint32_t)c[3];
    rgba = r | (g << 8) | (b << 16) | (al << 24);
    bgra = b | (g << 8) | (r << 16) | (al << 24);
}
// The fused present of one pixel: what present_kernel computes for a sample on the texel's centre -- weights (1, 0, 0, 0) -- from the value
// the backbuffer holds (an rgba16f surface holds the RNE-rounded half: present what is stored, not what was computed).
template <int OUT>
__device__ __forceinline__ void store_present(const LaunchDesc &L, size_t idx, float r, float g, float b) {
    float px[4] = {r, g, b, 1.0f};
    if (OUT == OUT_RGBA16F) {
#pragma unroll
        for (int k = 0; k < 3; k++) px[k] = (float)(_Float16)px[k];
    }
    uint32_t rgba, bgra;
    present_pack(px, rgba, bgra);
    L.present_rgba8[idx] = rgba;
    if (L.present_bgra8) L.present_bgra8[idx] = bgra;
}

// A pixel of a launch: the full (r, g, b, 1) pixel, or -- compact == 2, the lean wire format of a partition -- (r, g) into the
// record's first plane and b into its second.  Every pixel this path writes has alpha 1 (raycast_naive.wgsl:124,
// raycast_compute.wgsl:143), and the tiles of a partition exist to be moved over xGMI: 6 bytes instead of 8 (rgba16f).
template <int OUT>
__device__ __forceinline__ void store_out(const LaunchDesc &L, const PixelMap &pm, float r, float g, float b) {
    if (L.present_rgba8) {  // wave-uniform (the host sets it for compact == 0, one frame)
        store_present<OUT>(L, pm.out_index, r, g, b);
        if (L.present_only) return;
    }
    if (L.compact != 2u) { store_pixel<OUT>(L.out, pm.out_index, r, g, b, 1.0f); return; }  // wave-uniform
    const size_t tt = (size_t)L.ts * L.ts;
    if (OUT == OUT_RGBA32F) {
        float *rec = reinterpret_cast<float *>(L.out) + (size_t)pm.rec * tt * 3u;
        reinterpret_cast<float2 *>(rec)[pm.rec_px] = make_float2(r, g);
        rec[tt * 2u + pm.rec_px] = b;
    } else {
        _Float16 *rec = reinterpret_cast<_Float16 *>(L.out) + (size_t)pm.rec * tt * 3u;
        union { _Float16 h[2]; uint32_t u; } q;
        q.h[0] = (_Float16)r; q.h[1] = (_Float16)g;  // v_cvt_f16_f32, round to nearest even, as store_pixel
        reinterpret_cast<uint32_t *>(rec)[pm.rec_px] = q.u;
        rec[tt * 2u + pm.rec_px] = (_Float16)b;
    }
}

// ---- shared arithmetic (mirrors oracle/vokselis_oracle.c) ------------------------------------
__device__ __forceinline__ void mat4_mul_vec4(const float *m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float s = m[0 * 4 + r] * x;
        s = s + m[1 * 4 + r] * y;
        s = s + m[2 * 4 + r] * z;
        s = s + m[3 * 4 + r] * w;
        o[r] = s;
    }
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
    float len = sqrtf((x * x + y * y) + z * z);
    x = x / len; y = y / len; z = z / len;
}

// intersect_box: raycast_naive.wgsl:50-61 / raycast_compute.wgsl:42-53
__device__ __forceinline__ void intersect_box(const float o[3], const float d[3], float lo, float hi, float &t0, float &t1) {
    float tmin[3], tmax[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float inv = 1.0f / d[i];
        float a = (lo - o[i]) * inv, b = (hi - o[i]) * inv;
        tmin[i] = fminf(a, b);
        tmax[i] = fmaxf(a, b);
    }
    t0 = fmaxf(tmin[0], fmaxf(tmin[1], tmin[2]));
    t1 = fminf(tmax[0], fminf(tmax[1], tmax[2]));
}

// raycast_naive.wgsl:63-68.  pow(x, 1/2.4) = exp2(log2(x)/2.4) on the transcendental unit;
// colour only (never control flow), |err| ~ 1e-6.
__device__ __forceinline__ float linear_to_srgb(float x) {
    if (x <= 0.0031308f) return 12.92f * x;
    float p = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f));
    return 1.055f * p - 0.055f;
}

// raycast_naive.wgsl:106-107 -- bit-exact with vo_transfer_alpha: min(x, c), then smoothstep's affine map as ONE
// fused op whose constants carry the scale of x, then t*t*(3 - 2t).  SCALE 0: x is a value (f16 volumes); 1: filtered
// R8Unorm taps on their 0..255 scale; 2: the same times 2^-24 (the staged kernel's u8 taps enter the filter as f16
// subnormals; a power of two folds into c and k1 exactly).  5 VALU (min, fma+clamp, mul, fma, mul).
template <int SCALE>
__device__ __forceinline__ float transfer_alpha(float x) {
    constexpr float k2 = (float)(-0.1 / 1.1);
    constexpr float c = SCALE == 0 ? 0.9f : (SCALE == 1 ? 229.5f : 229.5f * 0x1p-24f);
    constexpr float k1 = SCALE == 0 ? (float)(1.0 / 1.1) : (SCALE == 1 ? (float)(1.0 / (255.0 * 1.1)) : (float)(1.0 / (255.0 * 1.1)) * 16777216.0f);
    float s = fmaf(fminf(x, c), k1, k2);
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    return (s * s) * fmaf(-2.0f, s, 3.0f);
}

// raycast_naive.wgsl:70-81: 0.5 + 0.5*cos(6.28318*(c*a + d)).  v_cos_f32 takes its argument in
// revolutions, so the phase is a single fma with constants pre-divided by 2*pi.
__device__ __forceinline__ void vertigo(float a, float &r, float &g, float &b) {
    constexpr double k = 6.28318 / 6.283185307179586476925;
    constexpr float c0 = (float)(1.0 * k), c1 = (float)(1.7 * k), c2 = (float)(0.4 * k);
    constexpr float d1 = (float)(0.15 * k), d2 = (float)(0.20 * k);
    r = fmaf(0.5f, __builtin_amdgcn_cosf(a * c0), 0.5f);
    g = fmaf(0.5f, __builtin_amdgcn_cosf(fmaf(a, c1, d1)), 0.5f);
    b = fmaf(0.5f, __builtin_amdgcn_cosf(fmaf(a, c2, d2)), 0.5f);
}

__device__ __forceinline__ float h2f(uint32_t bits16) {
    union { uint16_t u; _Float16 h; } c;
    c.u = (uint16_t)bits16;
    return (float)c.h;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ float trilerp(const float t[8], float fx, float fy, float fz) {
    float c00 = fmaf(fx, t[1] - t[0], t[0]), c10 = fmaf(fx, t[3] - t[2], t[2]);
    float c01 = fmaf(fx, t[5] - t[4], t[4]), c11 = fmaf(fx, t[7] - t[6], t[6]);
    float c0 = fmaf(fy, c10 - c00, c00), c1 = fmaf(fy, c11 - c01, c01);
    return fmaf(fz, c1 - c0, c0);
}

// ---- NAIVE_TRILINEAR: raycast_naive.wgsl:83-125 ----------------------------------------------
// The loop is VALU-issue bound on gfx950 (~4.4 cycles per wave64 VALU instruction at 8 waves/SIMD,
// tools/ubench/valu_rate.hip), so the kernel is written for instruction count:
//  * v_cvt_flr_i32_f32 + v_fract_f32 give the cell index and the lerp weight in 2 ops per axis;
//  * the cell index ((Bz*nby + By)*nbx + Bx)*64 + wz*16 + wy*4 + wx  (B = (i>>2)+1, w = i&3) splits per axis, so the
//    fast path (SAFE=false) reads it from three small LDS tables: 4 VALU + 3 ds_read_b32 instead of 14 VALU;
//  * P16 cells hold (tap, delta) f16 pairs so an x-lerp is one v_fma_mix_f32, no unpack;
//  * SAFE=false also drops the per-axis clamps and reads cells through a bounds-checked 32-bit-offset buffer
//    resource when the host has proved both are safe (vk_render.hip: dispatch_march); SAFE=true keeps the closed form.
__device__ __forceinline__ int cvt_floor_i32(float u) {
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(u));
    return i;
}
// float -> u32, saturating: negatives and NaN give 0 (C++ leaves that conversion undefined)
__device__ __forceinline__ uint32_t cvt_u32_sat(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
__device__ __forceinline__ int med3_i32(int v, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// true on exactly one active lane of the wave (instrumentation in COUNT builds only)
__device__ __forceinline__ bool wave_leader() {
    return (int)(threadIdx.x & 63u) == __ffsll((unsigned long long)__ballot(1)) - 1;
}
// ---- per-axis cell-index tables (LDS) -----------------------------------------------------------
// Cell with low-corner voxel i (i in [-1, n-1]) lives in physical brick (i >> 2) + 1 at in-brick
// position i & 3, so its index splits per axis:
//   Tx[i] = 64*((i>>2)+1) + (i&3),  Ty[i] = 64*nbx*((i>>2)+1) + 4*(i&3),  Tz[i] = 64*nbx*nby*((i>>2)+1) + 16*(i&3)
// Each table has n + 3 entries, entry e = i + 2 for i in [-2, n]; the two outer entries repeat their
// neighbours (clamp), so a position one step outside the box -- the prefetch of march_stream --
// still reads a real entry.  The three tables are stored back to back.  `shift` pre-scales the
// entries to byte offsets when no per-cell side table is read.
__host__ __device__ __forceinline__ uint32_t cell_lut_entries(uint32_t nx, uint32_t ny, uint32_t nz) { return (nx + ny + nz + 9u + 3u) & ~3u; }  // padded to whole uint4
__host__ __device__ __forceinline__ uint32_t cell_lut_bytes(uint32_t nx, uint32_t ny, uint32_t nz) { return cell_lut_entries(nx, ny, nz) * 4u; }
// Bounds-checked view of the cell array for the fast path (< 4 GiB): a raw buffer resource, so an
// offset outside the array reads zeros instead of faulting (memory-safety net; never hit by a valid ray).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cell_buffer(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}

// ---- COMPUTE_NEAREST: raycast_compute.wgsl:62-144 --------------------------------------------
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    const float inv = 1.0f / (e1 - e0);
    float s = (x - e0) * inv;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    return (s * s) * fmaf(-2.0f, s, 3.0f);
}

// ---- COMPUTE_NEAREST on the bricked record layout ---------------------------------------------
// Per-axis tables as for the cells, here for voxel i in [-kPairPad, n - 1 + kPairPad]: in-range entries
// are byte offsets of the record, out-of-range ones the marker kPairOob.  The records are read through
// a raw buffer resource sized to the array, so any sum that contains a marker is past the end and
// the load returns zeros -- exactly the zeros this build defines for out-of-range texel loads (A.2).
constexpr uint32_t kPairPad = 8;
constexpr int kPairDistRadius = 60;  // the records' skip map (embedded in the records, vk_volume_kernels.hpp) saturates at kPairDistRadius + 1 voxels
constexpr uint32_t kPairOob = 0x50000000u;  // > any record offset (array <= kPairOob bytes); 3 markers do not wrap
__host__ __device__ __forceinline__ uint32_t pair_lut_entries(uint32_t nx, uint32_t ny, uint32_t nz) { return (nx + ny + nz + 6u * kPairPad + 3u) & ~3u; }

// ---- 8^3 bricks staged through LDS (vk_staged.hpp): what the host and the re-layout kernel share ----------------------------
constexpr int kStagePad = 8;

struct StagedDesc {
    const unsigned char *copy[3];  // copy[k]: SLOW axis k, FAST axis (k+1)%3, MID axis (k+2)%3 (nullptr: not built)
    uint32_t copy_of_major[3];     // copy used by a wave whose rays' major axis is x / y / z
    uint32_t nv[3];                // padded voxel extent per axis (multiple of 8, >= n + kStagePad + 2)
    uint32_t npf[3];               // per copy: 16-byte pieces along its FAST axis
    uint32_t nbm[3];               // per copy: bricks along its MID axis
    uint32_t cap_bytes;            // LDS window capacity = dynamic LDS of the launch
    uint32_t slab_cells;           // a round is a slab of at most this many cells along the wave's major axis
    uint32_t grow_every;           // the slab search tries one cell above the last fit every grow_every-th round (>= 1)
    uint32_t row_pad;              // 1: window rows of an even number of pieces carry one more (an odd row pitch, in pieces, spreads the rows of a wave over the LDS banks)
};

}  // namespace vk
