// vk_kernels.hpp -- hand-written HIP (gfx950 / CDNA4) kernels of the vokselis raycast path.
//
// The WGSL of shaders/raycast_naive.wgsl (fs_main) and shaders/raycast_compute.wgsl
// (render/get_col2/single/tile) re-authored as wave64 compute kernels.  One lane = one ray,
// one wave = one 8x8 pixel block.  There is no dense contraction here, so no MFMA: the path is
// gather + VALU.  The opacity path (everything that feeds the loop trip count and the
// alpha >= 0.95 early-out) reproduces the arithmetic specification of oracle/vokselis_oracle.c
// operation for operation; this TU is compiled with -ffp-contract=off and fused ops appear
// only where fmaf is written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
    uint32_t debug_flags;          // bit 0: per-pixel march-loop trips instead of iterations (COUNT builds); bit 1 (staged): steps served from global memory; bit 2: adaptive probing (skip kernels), bit 3: start with long dense stretches, bit 4: issue priority by ray length
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
};

// ---- block -> pixels -------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD's L2).  A
// group of 512 consecutive physical blocks is mapped so that each XCD receives 64 consecutive
// logical blocks = the 64 waves of one 64x64 pixel tile: neighbouring rays share an L2, while
// successive tiles still spread over all XCDs (the frame is ~70 % empty, so contiguous bands
// per XCD would not balance).  Speed only -- nothing depends on the placement.
// (A persistent variant -- one wave per hardware slot pulling blocks from per-XCD atomic queues --
// balanced the per-SIMD work better (max/mean 1.62 -> 1.45) but lost 0.235 -> 0.35 ms to the
// dequeue round trips of ~32 k mostly trivial blocks: tools/experiments/persistent_workqueue.patch.)
__device__ __forceinline__ uint32_t logical_block(uint32_t b) {
    uint32_t group = b >> 9, r = b & 511u;
    return (group << 9) + ((r & 7u) << 6) + (r >> 3);
}

// (Dealing the tiles out SIMD by SIMD -- every SIMD one block from each of 8 tiles -- left the per-SIMD work
// spread at max/mean 1.66: the spread is block-to-block variation inside tiles, not tile placement.
// tools/experiments/simd_interleaved_block_order.patch)

// Issue priority by the length of the wave's longest ray, in quarters of the longest possible march (n / dt_scale
// trips): a frame is one or two rounds of resident waves, and under even sharing of a SIMD's issue slots the longest
// waves -- started first, finished last -- set the frame time while the short ones leave early.  Speed only.
__device__ __forceinline__ void set_wave_priority(bool hit, float t0, float t1, float dt, float full) {
    const float trips = hit ? (t1 - t0) / dt : 0.0f;
    if (__ballot(trips > 0.75f * full)) __builtin_amdgcn_s_setprio(3);
    else if (__ballot(trips > 0.5f * full)) __builtin_amdgcn_s_setprio(2);
    else if (__ballot(trips > 0.25f * full)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// ---- dealing positions of the heaviest-first order to ranks ------------------------------------
// Round j gives one position to every rank, in rank order.  With root_skip = k >= 2 the root (rank 0) sits out every
// k-th round (rounds k-1, 2k-1, ...): it also un-tiles every frame, and a lighter share of the march keeps it from
// being the rank everybody waits for.  k < 2: plain round robin, position q -> rank q % N, slot q / N.
__host__ __device__ __forceinline__ uint32_t deal_pos(uint32_t rank, uint32_t slot, uint32_t N, uint32_t k) {
    if (k < 2u) return rank + slot * N;
    const uint32_t round = rank ? slot : slot + slot / (k - 1u);  // the root's slot j is its j-th full round
    const uint32_t start = round * N - round / k;                 // one position less for every light round before
    return start + rank - ((round % k) == k - 1u ? 1u : 0u);      // (the root never sees a light round)
}
__host__ __device__ __forceinline__ void deal_owner(uint32_t pos, uint32_t N, uint32_t k, uint32_t &rank, uint32_t &slot) {
    if (k < 2u) { rank = pos % N; slot = pos / N; return; }
    const uint32_t G = k * N - 1u, g = pos / G, o = pos - g * G;  // a group: k - 1 full rounds and a light one
    uint32_t rj;
    if (o < (k - 1u) * N) { rj = o / N; rank = o - rj * N; }
    else { rj = k - 1u; rank = o - (k - 1u) * N + 1u; }
    slot = rank ? g * k + rj : g * (k - 1u) + rj;
}
// rounds needed to deal `tiles` positions = slots of a non-root rank
__host__ __device__ __forceinline__ uint32_t deal_rounds(uint32_t tiles, uint32_t N, uint32_t k) {
    if (k < 2u) return (tiles + N - 1u) / N;
    uint32_t r = tiles / N;
    while (r * N - r / k < tiles) r++;
    return r;
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
        const uint32_t g = lb / per_tile, sub = lb - g * per_tile;  // g: (slot, frame) pairs, frame fastest
        const uint32_t slot = g / L.n_frames;
        uint32_t fr = g - slot * L.n_frames;
        if (L.debug_flags & 32u) {
            // Consecutive g go to consecutive XCDs (logical_block), so with the frame index running fastest XCD x would march frames x, x + 8,
            // x + 16 ... of a tile position.  Under a moving camera neighbouring frames share almost all of their cells (a few pixels of shift),
            // frames eight apart far fewer: give every XCD a run of CONSECUTIVE frames instead -- residue x of the position takes frames
            // [start_x, start_x + count_x) -- so that a position's frames meet in one L2, one after the other.  A relabelling only.
            const uint32_t x = fr & 7u, j = fr >> 3, q = L.n_frames >> 3, rem = L.n_frames & 7u;
            fr = x * q + min(x, rem) + j;
        }
        f.frame = fr;
        f.lb = slot * per_tile + sub;
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
    uint32_t tile = pos < fv.n_tiles_launch ? fv.order[pos] : n_tiles;
    uint32_t tty = tile / L.tiles_x, ttx = tile - tty * L.tiles_x;
    uint32_t sy = sub / sps, sx = sub - sy * sps;
    uint32_t lx = sx * 8 + (lane & 7u), ly = sy * 8 + (lane >> 3);  // inside the tile
    uint32_t rx = ttx * L.ts + lx, ry = tty * L.ts + ly;             // inside the region
    m.x = L.ox + (int32_t)rx;
    m.y = L.oy + (int32_t)ry;
    m.valid = (tile < n_tiles) && rx < L.rw && ry < L.rh && m.x >= 0 && m.y >= 0 &&
              m.x < (int32_t)L.W && m.y < (int32_t)L.H;
    const uint32_t nf = L.frames ? L.n_frames : 1u;
    m.pos = pos;
    m.rec = slot * nf + fv.frame;
    m.rec_px = ly * L.ts + lx;
    m.out_index = L.compact ? (size_t)m.rec * (L.ts * L.ts) + m.rec_px
                            : ((size_t)fv.frame * L.H + (size_t)m.y) * L.W + (size_t)m.x;
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

// A pixel of a launch: the full (r, g, b, 1) pixel, or -- compact == 2, the lean wire format of a partition -- (r, g) into the
// record's first plane and b into its second.  Every pixel this path writes has alpha 1 (raycast_naive.wgsl:124,
// raycast_compute.wgsl:143), and the tiles of a partition exist to be moved over xGMI: 6 bytes instead of 8 (rgba16f).
template <int OUT>
__device__ __forceinline__ void store_out(const LaunchDesc &L, const PixelMap &pm, float r, float g, float b) {
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
//    resource when the host has proved both are safe (vk_api.hip: render_common); SAFE=true keeps the closed form.
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

// Built once per volume (two copies back to back: cell units, then byte offsets) ...
__global__ __launch_bounds__(256) void build_cell_luts_kernel(uint32_t *__restrict__ out, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t nbx,
                                                               uint32_t nby, uint32_t byte_shift) {
    const uint32_t n0 = nx + 3u, n1 = ny + 3u, n2 = nz + 3u, total = n0 + n1 + n2, padded = cell_lut_entries(nx, ny, nz);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < padded; e += gridDim.x * blockDim.x) {
        uint32_t v = 0;
        if (e < total) {
            uint32_t j = e, n = n0, brick_mul = 64u, cell_mul = 1u;
            if (e >= n0 + n1) { j = e - n0 - n1; n = n2; brick_mul = 64u * nbx * nby; cell_mul = 16u; }
            else if (e >= n0) { j = e - n0; n = n1; brick_mul = 64u * nbx; cell_mul = 4u; }
            const uint32_t c = min(max(j, 1u), n - 2u) - 1u;  // cell coordinate i + 1 in [0, n_vox]
            v = brick_mul * ((c + 3u) >> 2) + cell_mul * ((c + 3u) & 3u);
        }
        out[e] = v;
        out[padded + e] = v << byte_shift;
    }
}

// ... and copied into LDS by every wave of the march: a handful of 16-byte loads instead of ~200 VALU
// instructions of index arithmetic per wave.
__device__ __forceinline__ void load_cell_luts(const VolumeDesc &V, uint32_t *lut, uint32_t lane) {
    const uint32_t n4 = cell_lut_entries(V.nx, V.ny, V.nz) >> 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(V.lut);
    uint4 *dst = reinterpret_cast<uint4 *>(lut);
    for (uint32_t e = lane; e < n4; e += 64u) dst[e] = src[e];
}

// Bounds-checked view of the cell array for the fast path (< 4 GiB): a raw buffer resource, so an
// offset outside the array reads zeros instead of faulting (memory-safety net; never hit by a valid ray).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cell_buffer(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}

// x-lerps of one cell: c00, c10, c01, c11 (the four x edges of the footprint)
template <int VOL>
struct CellBits { u32x4_t v; };
template <>
struct CellBits<VOL_P8> { u32x2_t v; };

// PIN: mark the load volatile (aux bit 31: compiler-only, nothing changes in the encoding) so that it
// is issued where it is written -- a prefetch must not be sunk behind the loop's exit branch.
template <int VOL, bool PIN = false>
__device__ __forceinline__ CellBits<VOL> load_cell(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
    constexpr int aux = PIN ? (int)0x80000000u : 0;
    CellBits<VOL> c;
    if constexpr (VOL == VOL_P8) c.v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, aux);
    else c.v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, aux);
    return c;
}

template <int VOL>
__device__ __forceinline__ void xlerp_cell(const CellBits<VOL> &cb, float fx, float &c00, float &c10, float &c01, float &c11) {
    if constexpr (VOL == VOL_P8) {
        const uint32_t lo = cb.v.x, hi = cb.v.y;
        float t0_ = (float)(lo & 0xffu), t1_ = (float)((lo >> 8) & 0xffu), t2_ = (float)((lo >> 16) & 0xffu), t3_ = (float)(lo >> 24);
        float t4_ = (float)(hi & 0xffu), t5_ = (float)((hi >> 8) & 0xffu), t6_ = (float)((hi >> 16) & 0xffu), t7_ = (float)(hi >> 24);
        c00 = fmaf(fx, t1_ - t0_, t0_); c10 = fmaf(fx, t3_ - t2_, t2_);
        c01 = fmaf(fx, t5_ - t4_, t4_); c11 = fmaf(fx, t7_ - t6_, t6_);
    } else {
        union { u32x4_t u; half2_t h[4]; } c;
        c.u = cb.v;
        if constexpr (VOL == VOL_P16) {
            // (tap, delta) pairs: delta = t1 - t0 is exact in f16 for u8 data -> v_fma_mix_f32
            c00 = fmaf(fx, (float)c.h[0].y, (float)c.h[0].x); c10 = fmaf(fx, (float)c.h[1].y, (float)c.h[1].x);
            c01 = fmaf(fx, (float)c.h[2].y, (float)c.h[2].x); c11 = fmaf(fx, (float)c.h[3].y, (float)c.h[3].x);
        } else {
            float a0 = (float)c.h[0].x, a1 = (float)c.h[0].y, a2 = (float)c.h[1].x, a3 = (float)c.h[1].y;
            float a4 = (float)c.h[2].x, a5 = (float)c.h[2].y, a6 = (float)c.h[3].x, a7 = (float)c.h[3].y;
            c00 = fmaf(fx, a1 - a0, a0); c10 = fmaf(fx, a3 - a2, a2);
            c01 = fmaf(fx, a5 - a4, a4); c11 = fmaf(fx, a7 - a6, a6);
        }
    }
}

// ---- the march, resumable ---------------------------------------------------------------------
// Everything a ray needs to continue: the accumulators of the reference loop (t, p, alpha, colour
// sums), its per-ray constants and where its pixel goes.  64 bytes.
struct RayState {
    float t, t1, dt, px, py, pz, sx, sy, sz, A, Gr, Gg, Gb;
    uint32_t out;        // pixel index into the launch's output
    uint32_t pad[2];
};
static_assert(sizeof(RayState) == 64, "RayState is one 64-byte record");

struct Census {  // SIMT execution census + step counters (COUNT builds only)
    uint32_t n_iter = 0, n_samp = 0, w_outer = 0, w_inner = 0, w_sample = 0, n_look = 0, n_fb = 0;
    uint32_t skips = 0;  // trips that skipped (every build: drives the adaptive probing policy)
    // per-trip log of the wave (COUNT builds, debug bit 6; tools/repack_census.py): entry = live lanes | samplers << 7 | samplers whose alpha is
    // not 0 << 14 | wave-level walk iterations << 21
    uint32_t *log = nullptr;
    uint32_t log_cap = 0, trip_no = 0;
};

// Runs at most `budget` trips of the reference loop (raycast_naive.wgsl:101-119) on the state and
// returns whether the ray is still alive.  State in, state out: a ray marched in several pieces
// goes through exactly the same f32 operations as one marched in one go.
//
// SAFE=false (the fast path of the PACKED layouts) looks the cell index up instead of computing it:
// idx = Tx[ix] + Ty[iy] + Tz[iz] with three small per-axis tables in LDS (`lut`, filled by the kernel:
// see load_cell_luts).  Three ds_read_b32 (not VALU) + one v_add3 replace the 14 integer VALU
// instructions of the closed form -- the loop is VALU-issue bound.  Every table entry is a valid
// non-negative partial index and LDS reads outside the allocation return 0, so any combination stays
// inside the cell array: no clamp.
template <int VOL, bool SKIP, bool SAFE, bool COUNT, bool BOUNDED = false>
__device__ __forceinline__ bool march(const VolumeDesc &V, RayState &r, const uint32_t budget, Census &cs,
                                      const uint32_t *lut = nullptr, const float walk_cap = __builtin_inff(), const float walk_cap_all = __builtin_inff()) {
    constexpr bool PACKED = (VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16);
    constexpr bool BRICK9 = (VOL == VOL_B9U8 || VOL == VOL_B9F16);
    float t = r.t, px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    const float t1 = r.t1, dt = r.dt, sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    uint32_t &n_iter = cs.n_iter, &n_samp = cs.n_samp, &w_outer = cs.w_outer, &w_inner = cs.w_inner, &w_sample = cs.w_sample, &n_look = cs.n_look;
    const int mx = (int)V.nx - 1, my = (int)V.ny - 1, mz = (int)V.nz - 1;

    // per-ray constants of the skip bound (see below)
    // Skip bound per axis: room_i / |du_i| with room_i = d - f_i (moving up) or f_i + d - 1 (moving
    // down), minus a 0.02-cell margin that covers the rounding of the accumulated position
    // (<= 1e-3 cells); folded into two fmas: r_i = f_i * ska_i + (d * idu_i + skb_i).
    float idux = 0.f, iduy = 0.f, iduz = 0.f, skax = 0.f, skay = 0.f, skaz = 0.f, skbx = 0.f, skby = 0.f, skbz = 0.f;
    if (SKIP) {
        // rcp (1 ulp) is enough: these constants only bound a skip length, with the margins below.
        const float dux = fabsf(sx) * fnx, duy = fabsf(sy) * fny, duz = fabsf(sz) * fnz;  // cells per step
        idux = __builtin_amdgcn_rcpf(dux); iduy = __builtin_amdgcn_rcpf(duy); iduz = __builtin_amdgcn_rcpf(duz);
        // Position margin, in cells: a walk crosses at most kDistRadius + 1 cells of its fastest axis,
        // i.e. n <= 25 / max(du) steps, each adding <= 2^-25 of rounding to a coordinate in [0, 1]
        // (x n_i cells); doubled, plus 0.01 for the rounding of u itself.
        const float n_walk = (float)(kDistRadius + 1) * __builtin_amdgcn_rcpf(fmaxf(dux, fmaxf(duy, duz)));
        const float mg = fmaf(n_walk * 0x1p-24f, fmaxf(fnx, fmaxf(fny, fnz)), 0.01f);
        // Step margin: the walk below stops on the accumulated t; n additions drift by <= n * 2^-24 * t1,
        // i.e. a fraction e = 2^-24 * t1 / dt of the walk length (x4 for the fma and rcp roundings).
        const float e = fminf(0x1p-22f * t1 * __builtin_amdgcn_rcpf(dt), 1.0f);
        const float sc = 1.0f - e, cst = -(e + 0.01f);
        skax = (sx >= 0.0f ? -idux : idux) * sc; skay = (sy >= 0.0f ? -iduy : iduy) * sc; skaz = (sz >= 0.0f ? -iduz : iduz) * sc;
        skbx = fmaf((sx >= 0.0f ? -mg : -1.0f - mg) * idux, sc, cst);
        skby = fmaf((sy >= 0.0f ? -mg : -1.0f - mg) * iduy, sc, cst);
        skbz = fmaf((sz >= 0.0f ? -mg : -1.0f - mg) * iduz, sc, cst);
        idux *= sc; iduy *= sc; iduz *= sc;
    }
    // this ray's octant selects its distance map (bit i: moving up on axis i, as in ska/skb above)
    const uint32_t doff = SKIP ? ((sx >= 0.0f ? 1u : 0u) | (sy >= 0.0f ? 2u : 0u) | (sz >= 0.0f ? 4u : 0u)) * V.dist_oct_stride : 0u;
    const uint32_t *luty = lut + (V.nx + 3), *lutz = lut + (V.nx + V.ny + 6);
    const __amdgpu_buffer_rsrc_t cells = cell_buffer(V.data, SAFE ? 0u : (uint32_t)V.max_off + (1u << V.sh_x));
    const float t1q = __builtin_canonicalizef(t1);  // known-quiet copy: keeps a per-trip canonicalise out of the skip branch

    // One exit test per trip: `t < t1` (:101) and the alpha early-out (:115-117) are folded into the
    // loop condition; p and t are dead after the break, so advancing them unconditionally (:118)
    // changes nothing observable.
    uint32_t trip = 0;  // wave-uniform: the active lanes of a wave entered the loop together
    while (t < t1 && A < 0.95f && (!BOUNDED || trip < budget)) {
        if (BOUNDED) ++trip;
        if (COUNT) { n_look++; if (wave_leader()) w_outer++; }
        uint32_t *le = nullptr;
        if (COUNT && cs.log) {
            if (cs.trip_no < cs.log_cap) { le = cs.log + cs.trip_no; atomicAdd(le, 1u); }
            cs.trip_no++;
        }
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        int ix = cvt_floor_i32(ux), iy = cvt_floor_i32(uy), iz = cvt_floor_i32(uz);
        const float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy), fz = __builtin_amdgcn_fractf(uz);
        float c00, c10, c01, c11;  // x-lerped corners
        if (PACKED) {
            if (SAFE) { ix = med3_i32(ix, -1, mx); iy = med3_i32(iy, -1, my); iz = med3_i32(iz, -1, mz); }
            const int bx = ix >> 2, by = iy >> 2, bz = iz >> 2;
            const char *cptr = nullptr;
            uint32_t d = 0, coff = 0;
            if (SAFE) {
                int64_t off = (int64_t)bz * (int64_t)V.kz + (int64_t)(by * (int)V.ky + bx * (int)V.kx) +
                              (int64_t)((iz << V.sh_z) + (iy << V.sh_y) + (ix << V.sh_x)) + (int64_t)V.c0;
                off = off < 0 ? 0 : (off > (int64_t)V.max_off ? (int64_t)V.max_off : off);
                cptr = reinterpret_cast<const char *>(V.data) + off;
                if (SKIP) d = V.dist[(uint64_t)(off >> V.sh_x) + doff];
            } else {
                // cell index (SKIP) / cell byte offset (!SKIP) from the per-axis tables; entry i + 2 is voxel i
                const uint32_t idx = lut[ix + 2] + luty[iy + 2] + lutz[iz + 2];
                coff = SKIP ? (uint32_t)(idx << V.sh_x) : idx;
                if (SKIP) d = V.dist[idx + doff];
            }
            // A trip in which some lanes sample is paced by them: whatever a walker covers beyond a few steps it covers while
            // the samplers -- and every walker with a shorter walk -- wait for the longest walk of the wave (the walk loop ran
            // 3.9 iterations of four steps per trip with 19 of 64 lanes active).  In such a trip walks are capped; the walker
            // probes again next trip, which the wave makes anyway.  Any stop is exact: what is not skipped now is probed again.
            // (One compare serves the branch and the wave-level test; the cap is a scalar.)
            // (one compare serves the branch and the wave-level test; the cap is chosen on the scalar unit, as bits)
            const unsigned long long samplers = SKIP ? __ballot(d == 0) : 0ull;
            const uint32_t cap_now = samplers != 0ull ? __builtin_amdgcn_readfirstlane(__float_as_uint(walk_cap)) : __builtin_amdgcn_readfirstlane(__float_as_uint(walk_cap_all));
            if (SKIP && d != 0) {
                if (BOUNDED) cs.skips++;
                // Every cell within Chebyshev distance d-1 of this one is empty.  Sample j sits at
                // u + j*du; it is skipped iff its cell provably stays in that range on every axis:
                // j*|du| < d - f (moving up) or j*|du| <= f + d - 1 (moving down), minus the margins:
                // j < r = min_i r_i, r_i = f_i * ska_i + (d * idu_i + skb_i).
                const float fd = (float)d;
                const float rx = fmaf(fx, skax, fmaf(fd, idux, skbx));
                const float ry = fmaf(fy, skay, fmaf(fd, iduy, skby));
                const float rz = fmaf(fz, skaz, fmaf(fd, iduz, skbz));
                // The walk advances the reference loop's own accumulators (p += s, t += dt: the same
                // f32 additions in the same order) while t < tstop.  tstop <= t1, so every skipped
                // iteration passed the reference's `t < t1` test on the very same t; and t_j < tstop
                // means j < r.  The current sample (j = 0, its cell is empty) is always skipped.
                float rmin = fminf(fminf(rx, ry), rz);
                asm("v_min_f32 %0, %1, %2" : "=v"(rmin) : "s"(cap_now), "v"(rmin));  // (a known-quiet scalar: no canonicalise)
                float tstop = fmaf(rmin, dt, t);
                asm("v_min_f32 %0, %1, %2" : "=v"(tstop) : "v"(tstop), "v"(t1q));    // (t1 is finite: no canonicalise per trip)
                const float tstop2 = fmaf(-1.5f, dt, tstop);  // t < tstop2  =>  t + dt < tstop as well
                const float tstop4 = fmaf(-3.5f, dt, tstop);  // t < tstop4  =>  t + 3 dt < tstop as well
                px = px + sx; py = py + sy; pz = pz + sz;
                t = t + dt;
                if (COUNT) { n_iter++; if (wave_leader()) { w_inner++; if (le) atomicAdd(le, 1u << 21); } }
                while (t < tstop4) {  // four skipped iterations per trip of the walk
#pragma unroll
                    for (int j = 0; j < 4; j++) { px = px + sx; py = py + sy; pz = pz + sz; t = t + dt; }
                    if (COUNT) { n_iter += 4; if (wave_leader()) { w_inner++; if (le) atomicAdd(le, 1u << 21); } }
                }
                if (t < tstop2) {
                    // (the empty asm keeps this an exec-masked region: if-converted, the two steps are computed for every lane and
                    // then selected by FOUR v_cndmask through VCC in a row, ~16 issue cycles each -- profiles/r03_ubench_valu_issue_rate.txt --
                    // twice the cost of a whole four-step walk iteration)
                    asm volatile("" : "+v"(t));
                    px = px + sx; py = py + sy; pz = pz + sz;
                    t = t + dt;
                    px = px + sx; py = py + sy; pz = pz + sz;
                    t = t + dt;
                    if (COUNT) { n_iter += 2; }
                }
                if (t < tstop) {
                    px = px + sx; py = py + sy; pz = pz + sz;
                    t = t + dt;
                    if (COUNT) { n_iter++; }
                }
                continue;
            }
            CellBits<VOL> cb;
            if (SAFE) {
                if constexpr (VOL == VOL_P8) { const uint2 c = *reinterpret_cast<const uint2 *>(cptr); cb.v.x = c.x; cb.v.y = c.y; }
                else { const uint4 c = *reinterpret_cast<const uint4 *>(cptr); cb.v.x = c.x; cb.v.y = c.y; cb.v.z = c.z; cb.v.w = c.w; }
            } else {
                cb = load_cell<VOL>(cells, coff);
            }
            xlerp_cell<VOL>(cb, fx, c00, c10, c01, c11);
        } else if (BRICK9) {
            // cell coords c = i + 1 in [0, n]; brick c >> 3, local c & 7; the taps sit at local
            // (l, l+1) per axis of the 9^3 brick: offsets {0,1} + {0,9} + {0,81} from one base
            const int cx = med3_i32(ix, -1, mx) + 1, cy = med3_i32(iy, -1, my) + 1, cz = med3_i32(iz, -1, mz) + 1;
            const uint32_t brick = (uint32_t)(((cz >> 3) * (int)V.nby + (cy >> 3)) * (int)V.nbx + (cx >> 3));
            const uint32_t local = (uint32_t)((cz & 7) * 81 + (cy & 7) * 9 + (cx & 7));
            typedef uint16_t u16_unaligned __attribute__((aligned(1)));
            typedef uint32_t u32_unaligned __attribute__((aligned(2)));
            float tp[8];
            if (VOL == VOL_B9U8) {
                const uint8_t *b = reinterpret_cast<const uint8_t *>(V.data) + ((uint64_t)brick * 729u + local);
                const uint32_t p00 = *reinterpret_cast<const u16_unaligned *>(b), p10 = *reinterpret_cast<const u16_unaligned *>(b + 9);
                const uint32_t p01 = *reinterpret_cast<const u16_unaligned *>(b + 81), p11 = *reinterpret_cast<const u16_unaligned *>(b + 90);
                tp[0] = (float)(p00 & 0xffu); tp[1] = (float)(p00 >> 8); tp[2] = (float)(p10 & 0xffu); tp[3] = (float)(p10 >> 8);
                tp[4] = (float)(p01 & 0xffu); tp[5] = (float)(p01 >> 8); tp[6] = (float)(p11 & 0xffu); tp[7] = (float)(p11 >> 8);
            } else {
                const uint16_t *b = reinterpret_cast<const uint16_t *>(V.data) + ((uint64_t)brick * 729u + local);
                const uint32_t p00 = *reinterpret_cast<const u32_unaligned *>(b), p10 = *reinterpret_cast<const u32_unaligned *>(b + 9);
                const uint32_t p01 = *reinterpret_cast<const u32_unaligned *>(b + 81), p11 = *reinterpret_cast<const u32_unaligned *>(b + 90);
                tp[0] = h2f(p00 & 0xffffu); tp[1] = h2f(p00 >> 16); tp[2] = h2f(p10 & 0xffffu); tp[3] = h2f(p10 >> 16);
                tp[4] = h2f(p01 & 0xffffu); tp[5] = h2f(p01 >> 16); tp[6] = h2f(p11 & 0xffffu); tp[7] = h2f(p11 >> 16);
            }
            c00 = fmaf(fx, tp[1] - tp[0], tp[0]); c10 = fmaf(fx, tp[3] - tp[2], tp[2]);
            c01 = fmaf(fx, tp[5] - tp[4], tp[4]); c11 = fmaf(fx, tp[7] - tp[6], tp[6]);
        } else {
            int x0 = clampi(ix, 0, mx), x1 = clampi(ix + (ix < 0x7fffffff), 0, mx);
            int y0 = clampi(iy, 0, my), y1 = clampi(iy + (iy < 0x7fffffff), 0, my);
            int z0 = clampi(iz, 0, mz), z1 = clampi(iz + (iz < 0x7fffffff), 0, mz);
            size_t sy_ = V.nx, sz_ = (size_t)V.nx * V.ny;
            size_t r00 = y0 * sy_ + z0 * sz_, r10 = y1 * sy_ + z0 * sz_;
            size_t r01 = y0 * sy_ + z1 * sz_, r11 = y1 * sy_ + z1 * sz_;
            float tp[8];
            if (VOL == VOL_LINEAR_U8) {
                const uint8_t *v = reinterpret_cast<const uint8_t *>(V.data);
                tp[0] = (float)v[r00 + x0]; tp[1] = (float)v[r00 + x1]; tp[2] = (float)v[r10 + x0]; tp[3] = (float)v[r10 + x1];
                tp[4] = (float)v[r01 + x0]; tp[5] = (float)v[r01 + x1]; tp[6] = (float)v[r11 + x0]; tp[7] = (float)v[r11 + x1];
            } else {
                const uint16_t *v = reinterpret_cast<const uint16_t *>(V.data);
                tp[0] = h2f(v[r00 + x0]); tp[1] = h2f(v[r00 + x1]); tp[2] = h2f(v[r10 + x0]); tp[3] = h2f(v[r10 + x1]);
                tp[4] = h2f(v[r01 + x0]); tp[5] = h2f(v[r01 + x1]); tp[6] = h2f(v[r11 + x0]); tp[7] = h2f(v[r11 + x1]);
            }
            c00 = fmaf(fx, tp[1] - tp[0], tp[0]); c10 = fmaf(fx, tp[3] - tp[2], tp[2]);
            c01 = fmaf(fx, tp[5] - tp[4], tp[4]); c11 = fmaf(fx, tp[7] - tp[6], tp[6]);
        }
        float c0 = fmaf(fy, c10 - c00, c00), c1 = fmaf(fy, c11 - c01, c01);
        float r = fmaf(fz, c1 - c0, c0);
        const float a = transfer_alpha<(VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_LINEAR_U8 || VOL == VOL_B9U8) ? 1 : 0>(r);
        if (COUNT && le) atomicAdd(le, (1u << 7) + (a != 0.0f ? 1u << 14 : 0u));
        if (SKIP) {
            // A cell is non-empty as soon as one of its 8 taps is above the threshold; the FILTERED value of a sample inside it
            // often is not (a lone voxel just above it, a silhouette), and then alpha is exactly 0: w = 0, every accumulator
            // takes +0 (the cosines are finite).  When that holds for every lane sampling in this trip -- it mostly does in
            // the executions that serve one or two lanes -- the palette and the compositing are left out: 17 of the 45
            // instructions, no bit changes.
            if (__ballot(a != 0.0f) == 0ull) {
                if (COUNT) { n_iter++; n_samp++; if (wave_leader()) w_sample++; }
                px = px + sx; py = py + sy; pz = pz + sz;  // :118
                t = t + dt;
                continue;
            }
        }
        // vertigo(): cos(6.28318*(c*a + d)); v_cos_f32 takes revolutions
        constexpr double kk = 6.28318 / 6.283185307179586476925;
        constexpr float pc0 = (float)(1.0 * kk), pc1 = (float)(1.7 * kk), pc2 = (float)(0.4 * kk);
        constexpr float pd1 = (float)(0.15 * kk), pd2 = (float)(0.20 * kk);
        const float cr = __builtin_amdgcn_cosf(a * pc0);
        const float cg = __builtin_amdgcn_cosf(fmaf(a, pc1, pd1));
        const float cb = __builtin_amdgcn_cosf(fmaf(a, pc2, pd2));
        if (COUNT) { n_iter++; n_samp++; if (wave_leader()) w_sample++; }
        const float w = (1.0f - A) * a;  // :112-114
        Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
        A = A + w;
        px = px + sx; py = py + sy; pz = pz + sz;  // :118
        t = t + dt;
    }
    r.t = t; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
    return t < t1 && A < 0.95f;
}

// The same loop for the fast path without skipping (every trip samples), software-pipelined: the
// position is advanced first and the NEXT trip's cell is requested before this trip's sample is
// evaluated, so the fetch latency overlaps the ~40 VALU instructions of a sample instead of adding
// to them -- it is the lone heavy waves at the tail of a frame that set the frame time.  The f32
// operations on t, p, A and the colour sums are those of march(), in the same order per variable.
// The request one step past the ray's end reads a real (clamped) table entry and is never used.
// CELL_LUT: the tables hold cell indices (the skip kernels' copy) instead of byte offsets; `budget` bounds the trips
// (0xffffffff: none) so that the skip kernels can run stretches of it between probing windows.
template <int VOL, bool COUNT, bool CELL_LUT = false>
__device__ __forceinline__ bool march_stream(const VolumeDesc &V, RayState &r, Census &cs, const uint32_t *lut, uint32_t budget = 0xffffffffu) {
    float t = r.t, px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    const float t1 = r.t1, dt = r.dt, sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const uint32_t *luty = lut + (V.nx + 3), *lutz = lut + (V.nx + V.ny + 6);
    const __amdgpu_buffer_rsrc_t cells = cell_buffer(V.data, (uint32_t)V.max_off + (1u << V.sh_x));
    if (!(t < t1 && A < 0.95f)) return false;
    const uint32_t lsh = CELL_LUT ? V.sh_x : 0u;
    float fx, fy, fz;
    CellBits<VOL> c0, c1;  // two cell buffers, used alternately (no register copies between trips)
    {
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        c0 = load_cell<VOL>(cells, (lut[cvt_floor_i32(ux) + 2] + luty[cvt_floor_i32(uy) + 2] + lutz[cvt_floor_i32(uz) + 2]) << lsh);
    }
    // one trip: request `nxt` for the advanced position, evaluate `cur`; returns whether the ray goes on
    auto trip = [&](const CellBits<VOL> &cur, CellBits<VOL> &nxt) -> bool {
        if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; if (wave_leader()) { cs.w_outer++; cs.w_sample++; } }
        px = px + sx; py = py + sy; pz = pz + sz;  // :118
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        nxt = load_cell<VOL>(cells, (lut[cvt_floor_i32(ux) + 2] + luty[cvt_floor_i32(uy) + 2] + lutz[cvt_floor_i32(uz) + 2]) << lsh);
        float c00, c10, c01, c11;
        xlerp_cell<VOL>(cur, fx, c00, c10, c01, c11);
        float l0 = fmaf(fy, c10 - c00, c00), l1 = fmaf(fy, c11 - c01, c01);
        float v = fmaf(fz, l1 - l0, l0);
        const float a = transfer_alpha<(VOL == VOL_P8 || VOL == VOL_P16) ? 1 : 0>(v);
        constexpr double kk = 6.28318 / 6.283185307179586476925;
        constexpr float pc0 = (float)(1.0 * kk), pc1 = (float)(1.7 * kk), pc2 = (float)(0.4 * kk);
        constexpr float pd1 = (float)(0.15 * kk), pd2 = (float)(0.20 * kk);
        const float cr = __builtin_amdgcn_cosf(a * pc0);
        const float cg = __builtin_amdgcn_cosf(fmaf(a, pc1, pd1));
        const float cb = __builtin_amdgcn_cosf(fmaf(a, pc2, pd2));
        const float w = (1.0f - A) * a;  // :112-114
        Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
        A = A + w;
        t = t + dt;
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return t < t1 && A < 0.95f;
    };
    bool alive = true;
    for (;;) {
        if (!trip(c0, c1)) { alive = false; break; }
        if (budget != 0xffffffffu && --budget == 0u) break;
        if (!trip(c1, c0)) { alive = false; break; }
        if (budget != 0xffffffffu && --budget == 0u) break;
    }
    // The last requests are consumed here, on the exit path too: with a use on both sides of the exit
    // branch the compiler cannot sink a request behind it (which would undo the pipelining).
    asm volatile("" ::"v"(c0.v), "v"(c1.v));
    r.t = t; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
    return alive;
}

// The dense 9^3-brick layouts, software-pipelined the same way: these serve volumes far larger than
// the caches, where every step's four x-pair loads are HBM/fabric latency.  The next trip's loads are
// requested (clamped indices, so always inside the array) before this trip's sample is evaluated.
template <int VOL, bool COUNT>
__device__ __forceinline__ void march_b9_stream(const VolumeDesc &V, RayState &r, Census &cs) {
    static_assert(VOL == VOL_B9U8 || VOL == VOL_B9F16, "9^3 brick layouts");
    float t = r.t, px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    const float t1 = r.t1, dt = r.dt, sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const int mx = (int)V.nx - 1, my = (int)V.ny - 1, mz = (int)V.nz - 1;
    if (!(t < t1 && A < 0.95f)) return;
    typedef uint16_t u16_unaligned __attribute__((aligned(1)));
    typedef uint32_t u32_unaligned __attribute__((aligned(2)));
    struct Taps { uint32_t p00, p10, p01, p11; };  // x pairs at (y, z) = (0,0) (1,0) (0,1) (1,1)
    auto request = [&](float ux, float uy, float uz) -> Taps {
        const int cx = med3_i32(cvt_floor_i32(ux), -1, mx) + 1, cy = med3_i32(cvt_floor_i32(uy), -1, my) + 1, cz = med3_i32(cvt_floor_i32(uz), -1, mz) + 1;
        const uint32_t brick = (uint32_t)(((cz >> 3) * (int)V.nby + (cy >> 3)) * (int)V.nbx + (cx >> 3));
        const uint32_t local = (uint32_t)((cz & 7) * 81 + (cy & 7) * 9 + (cx & 7));
        Taps q;
        if (VOL == VOL_B9U8) {
            const uint8_t *b = reinterpret_cast<const uint8_t *>(V.data) + ((uint64_t)brick * 729u + local);
            q.p00 = *reinterpret_cast<const u16_unaligned *>(b); q.p10 = *reinterpret_cast<const u16_unaligned *>(b + 9);
            q.p01 = *reinterpret_cast<const u16_unaligned *>(b + 81); q.p11 = *reinterpret_cast<const u16_unaligned *>(b + 90);
        } else {
            const uint16_t *b = reinterpret_cast<const uint16_t *>(V.data) + ((uint64_t)brick * 729u + local);
            q.p00 = *reinterpret_cast<const u32_unaligned *>(b); q.p10 = *reinterpret_cast<const u32_unaligned *>(b + 9);
            q.p01 = *reinterpret_cast<const u32_unaligned *>(b + 81); q.p11 = *reinterpret_cast<const u32_unaligned *>(b + 90);
        }
        return q;
    };
    float fx, fy, fz;
    Taps c0, c1;
    {
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        c0 = request(ux, uy, uz);
    }
    auto trip = [&](const Taps &cur, Taps &nxt) -> bool {
        if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; if (wave_leader()) { cs.w_outer++; cs.w_sample++; } }
        px = px + sx; py = py + sy; pz = pz + sz;  // :118
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        nxt = request(ux, uy, uz);
        float tp[8];
        if (VOL == VOL_B9U8) {
            tp[0] = (float)(cur.p00 & 0xffu); tp[1] = (float)(cur.p00 >> 8); tp[2] = (float)(cur.p10 & 0xffu); tp[3] = (float)(cur.p10 >> 8);
            tp[4] = (float)(cur.p01 & 0xffu); tp[5] = (float)(cur.p01 >> 8); tp[6] = (float)(cur.p11 & 0xffu); tp[7] = (float)(cur.p11 >> 8);
        } else {
            tp[0] = h2f(cur.p00 & 0xffffu); tp[1] = h2f(cur.p00 >> 16); tp[2] = h2f(cur.p10 & 0xffffu); tp[3] = h2f(cur.p10 >> 16);
            tp[4] = h2f(cur.p01 & 0xffffu); tp[5] = h2f(cur.p01 >> 16); tp[6] = h2f(cur.p11 & 0xffffu); tp[7] = h2f(cur.p11 >> 16);
        }
        const float c00 = fmaf(fx, tp[1] - tp[0], tp[0]), c10 = fmaf(fx, tp[3] - tp[2], tp[2]);
        const float c01 = fmaf(fx, tp[5] - tp[4], tp[4]), c11 = fmaf(fx, tp[7] - tp[6], tp[6]);
        const float l0 = fmaf(fy, c10 - c00, c00), l1 = fmaf(fy, c11 - c01, c01);
        float v = fmaf(fz, l1 - l0, l0);
        const float a = transfer_alpha<VOL == VOL_B9U8 ? 1 : 0>(v);
        constexpr double kk = 6.28318 / 6.283185307179586476925;
        constexpr float pc0 = (float)(1.0 * kk), pc1 = (float)(1.7 * kk), pc2 = (float)(0.4 * kk);
        constexpr float pd1 = (float)(0.15 * kk), pd2 = (float)(0.20 * kk);
        const float cr = __builtin_amdgcn_cosf(a * pc0);
        const float cg = __builtin_amdgcn_cosf(fmaf(a, pc1, pd1));
        const float cb = __builtin_amdgcn_cosf(fmaf(a, pc2, pd2));
        const float w = (1.0f - A) * a;  // :112-114
        Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
        A = A + w;
        t = t + dt;
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return t < t1 && A < 0.95f;
    };
    for (;;) {
        if (!trip(c0, c1)) break;
        if (!trip(c1, c0)) break;
    }
    asm volatile("" ::"v"(c0.p00), "v"(c0.p10), "v"(c0.p01), "v"(c0.p11), "v"(c1.p00), "v"(c1.p10), "v"(c1.p01), "v"(c1.p11));
    r.t = t; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
}

// The quad layouts: one load per sample (two consecutive elements), software-pipelined like the others.
template <int VOL, bool COUNT>
__device__ __forceinline__ void march_quads_stream(const VolumeDesc &V, RayState &r, Census &cs) {
    static_assert(VOL == VOL_Q8 || VOL == VOL_QF16, "quad layouts");
    float t = r.t, px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    const float t1 = r.t1, dt = r.dt, sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const int mx = (int)V.nx - 1, my = (int)V.ny - 1, mz = (int)V.nz - 1;
    if (!(t < t1 && A < 0.95f)) return;
    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));  // elements are 4 / 8 bytes: the pair is under-aligned
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    struct Taps { uint32_t a, b, c, d; };  // u8: a = element(x), b = element(x+1); f16: (a, b) = element(x), (c, d) = element(x+1)
    auto request = [&](float ux, float uy, float uz) -> Taps {
        const int cx = med3_i32(cvt_floor_i32(ux), -1, mx) + 1, cy = med3_i32(cvt_floor_i32(uy), -1, my) + 1, cz = med3_i32(cvt_floor_i32(uz), -1, mz) + 1;
        const uint32_t brick = (uint32_t)(((cz >> 3) * (int)V.nby + (cy >> 3)) * (int)V.nbx + (cx >> 3));
        const uint32_t local = (uint32_t)(((cz & 7) * 8 + (cy & 7)) * 9 + (cx & 7));
        const uint64_t e = (uint64_t)brick * 576u + local;
        Taps q;
        if (VOL == VOL_Q8) {
            const u32x2_a4 v = *reinterpret_cast<const u32x2_a4 *>(reinterpret_cast<const uint32_t *>(V.data) + e);
            q.a = v.x; q.b = v.y; q.c = 0; q.d = 0;
        } else {
            const u32x4_a8 v = *reinterpret_cast<const u32x4_a8 *>(reinterpret_cast<const uint2 *>(V.data) + e);
            q.a = v.x; q.b = v.y; q.c = v.z; q.d = v.w;
        }
        return q;
    };
    float fx, fy, fz;
    Taps c0, c1;
    {
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        c0 = request(ux, uy, uz);
    }
    auto trip = [&](const Taps &cur, Taps &nxt) -> bool {
        if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; if (wave_leader()) { cs.w_outer++; cs.w_sample++; } }
        px = px + sx; py = py + sy; pz = pz + sz;  // :118
        const float ux = fmaf(px, fnx, -0.5f), uy = fmaf(py, fny, -0.5f), uz = fmaf(pz, fnz, -0.5f);
        nxt = request(ux, uy, uz);
        float tp[8];  // tap index dx + 2*dy + 4*dz
        if (VOL == VOL_Q8) {
            tp[0] = (float)(cur.a & 0xffu); tp[2] = (float)((cur.a >> 8) & 0xffu); tp[4] = (float)((cur.a >> 16) & 0xffu); tp[6] = (float)(cur.a >> 24);
            tp[1] = (float)(cur.b & 0xffu); tp[3] = (float)((cur.b >> 8) & 0xffu); tp[5] = (float)((cur.b >> 16) & 0xffu); tp[7] = (float)(cur.b >> 24);
        } else {
            tp[0] = h2f(cur.a & 0xffffu); tp[2] = h2f(cur.a >> 16); tp[4] = h2f(cur.b & 0xffffu); tp[6] = h2f(cur.b >> 16);
            tp[1] = h2f(cur.c & 0xffffu); tp[3] = h2f(cur.c >> 16); tp[5] = h2f(cur.d & 0xffffu); tp[7] = h2f(cur.d >> 16);
        }
        const float c00 = fmaf(fx, tp[1] - tp[0], tp[0]), c10 = fmaf(fx, tp[3] - tp[2], tp[2]);
        const float c01 = fmaf(fx, tp[5] - tp[4], tp[4]), c11 = fmaf(fx, tp[7] - tp[6], tp[6]);
        const float l0 = fmaf(fy, c10 - c00, c00), l1 = fmaf(fy, c11 - c01, c01);
        float v = fmaf(fz, l1 - l0, l0);
        const float a = transfer_alpha<VOL == VOL_Q8 ? 1 : 0>(v);
        constexpr double kk = 6.28318 / 6.283185307179586476925;
        constexpr float pc0 = (float)(1.0 * kk), pc1 = (float)(1.7 * kk), pc2 = (float)(0.4 * kk);
        constexpr float pd1 = (float)(0.15 * kk), pd2 = (float)(0.20 * kk);
        const float cr = __builtin_amdgcn_cosf(a * pc0);
        const float cg = __builtin_amdgcn_cosf(fmaf(a, pc1, pd1));
        const float cb = __builtin_amdgcn_cosf(fmaf(a, pc2, pd2));
        const float w = (1.0f - A) * a;  // :112-114
        Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
        A = A + w;
        t = t + dt;
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return t < t1 && A < 0.95f;
    };
    for (;;) {
        if (!trip(c0, c1)) break;
        if (!trip(c1, c0)) break;
    }
    asm volatile("" ::"v"(c0.a), "v"(c0.b), "v"(c0.c), "v"(c0.d), "v"(c1.a), "v"(c1.b), "v"(c1.c), "v"(c1.d));
    r.t = t; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
}

// Whole frames in a batched launch: the march covers the active tiles; the tiles behind a frame's active positions hold
// only the clear colour (examples/bonsai/main.rs:41).  They are written by extra blocks at the END of the same grid -- one
// wave clears 512 pixels, 64 consecutive ones per store -- which the dispatcher hands out when the last march waves are
// draining: the stores ride in the launch's tail.  (As blocks of the march proper they were 20 000 waves per C2 frame whose
// only work was a store behind a full wave set-up; as a kernel of their own they cost 2.7 us per frame in series.)
template <int OUT>
__device__ __forceinline__ void clear_inactive_strip(const LaunchDesc &L, uint32_t b, uint32_t lane) {
    const uint32_t strips = (L.ts * L.ts + 511u) / 512u;  // 512-pixel strips per tile
    const uint32_t strip = b % strips; b /= strips;
    const uint32_t j = b % L.clear_max_inactive;
    const uint32_t frame = b / L.clear_max_inactive;
    if (frame >= L.n_frames) return;
    const FrameDesc &d = L.frames[frame];
    const uint32_t n_tiles = L.tiles_x * L.tiles_y, pos = d.pad[0] + j;  // pad[0]: the frame's active tile count
    if (pos >= n_tiles) return;
    const uint32_t tile = L.tile_order[d.order_off + pos];
    const uint32_t tyi = tile / L.tiles_x, txi = tile - tyi * L.tiles_x;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) {  // store k of the wave covers 64 consecutive pixels of the tile's rows
        const uint32_t l = strip * 512u + k * 64u + lane;
        if (l >= L.ts * L.ts) return;
        const uint32_t ly = l / L.ts, lx = l - ly * L.ts;
        const uint32_t x = txi * L.ts + lx, y = tyi * L.ts + ly;
        if (x < L.W && y < L.H) store_pixel<OUT>(L.out, ((size_t)frame * L.H + y) * L.W + x, 0.0f, 0.0f, 0.0f, 1.0f);
    }
}

template <int VOL, bool SKIP, bool SAFE, int OUT, bool COUNT>
__global__ __launch_bounds__(64) void raymarch_naive_kernel(const LaunchDesc L, const VolumeDesc V) {
    static_assert(VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16 || (!SKIP && SAFE), "linear / bricked layouts: no skip map, clamped indices");
    if (blockIdx.x >= L.grid_march) { clear_inactive_strip<OUT>(L, blockIdx.x - L.grid_march, threadIdx.x); return; }  // wave-uniform
    const uint32_t lb = logical_block(blockIdx.x);
    if (lb >= L.n_blocks) return;  // wave-uniform
    const uint32_t lane = threadIdx.x;
    unsigned long long t_start = 0;
    if (COUNT) t_start = __builtin_amdgcn_s_memrealtime();
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, lane);
    {
        // Screen-space cull (wave-uniform): an 8x8 block wholly outside the projected cube's bounding
        // rectangle (host-computed, padded) holds only misses: clear colour, no ray set-up.
        const int bx0 = pm.x - (int)(lane & 7u), by0 = pm.y - (int)(lane >> 3);
        // ... and so does every block of a tile the box's silhouette cannot reach (the inactive tiles behind the order's
        // active positions: a whole-frame launch covers them too, a partition never launches them)
        if (pm.pos >= fv.n_active || bx0 + 8 <= fv.cull_x0 || bx0 >= fv.cull_x1 || by0 + 8 <= fv.cull_y0 || by0 >= fv.cull_y1) {
            if (!pm.valid) return;
            store_out<OUT>(L, pm, 0.0f, 0.0f, 0.0f);
            if (COUNT && L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = 0;
            return;
        }
    }
    constexpr bool USE_LUT = (VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16) && !SAFE;
    extern __shared__ uint32_t cell_lut[];
    if (USE_LUT) {  // all 64 lanes are still here
        load_cell_luts(V, cell_lut, lane);
        __syncthreads();
    }
    if (!pm.valid) return;

    // --- ray: SURVEY A.1 step 1 (replaces vs_main + rasteriser) ---
    float fxp = (float)pm.x + 0.5f, fyp = (float)pm.y + 0.5f;
    float ndcx = (2.0f * fxp) / (float)L.W - 1.0f;
    float ndcy = 1.0f - (2.0f * fyp) / (float)L.H;
    float q[4];
    mat4_mul_vec4(fv.inv_proj, ndcx, ndcy, 1.0f, 1.0f, q);
    const float eye[3] = {fv.eye[0], fv.eye[1], fv.eye[2]};
    float dir[3] = {q[0] / q[3] - eye[0], q[1] / q[3] - eye[1], q[2] / q[3] - eye[2]};
    normalize3(dir[0], dir[1], dir[2]);

    float t0, t1;
    intersect_box(eye, dir, 0.0f, 1.0f, t0, t1);
    Census cs;
    const bool trip_log = COUNT && L.trace && (L.debug_flags & 64u);
    if (trip_log) {
        cs.log_cap = L.debug_flags >> 16;
        cs.log = reinterpret_cast<uint32_t *>(L.trace) + (size_t)lb * cs.log_cap;
    }
    // colour is accumulated as G = sum w*cos(phase); C = 0.5*A + 0.5*G at the end (sum w == A)
    float Gr = 0.0f, Gg = 0.0f, Gb = 0.0f, A = 0.0f;
    float Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    if (!(t0 > t1)) {  // :91-93
        t0 = fmaxf(t0, 0.0f);  // :94
        const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
        float dtx = 1.0f / (fnx * fabsf(dir[0]));
        float dty = 1.0f / (fny * fabsf(dir[1]));
        float dtz = 1.0f / (fnz * fabsf(dir[2]));
        const float dt = L.dt_scale * fminf(dtx, fminf(dty, dtz));  // :97-99
        float px = eye[0] + t0 * dir[0], py = eye[1] + t0 * dir[1], pz = eye[2] + t0 * dir[2];  // :100
        const float sx = dir[0] * dt, sy = dir[1] * dt, sz = dir[2] * dt;  // :118
        RayState r;
        r.t = t0; r.t1 = t1; r.dt = dt;
        r.px = px; r.py = py; r.pz = pz; r.sx = sx; r.sy = sy; r.sz = sz;
        r.A = 0.0f; r.Gr = 0.0f; r.Gg = 0.0f; r.Gb = 0.0f;  // colour sums: G = sum w*cos(phase); C = A/2 + G/2 (sum w == A)
        r.out = (uint32_t)pm.out_index;
        // (not in the skip kernels: a ray's nominal length says little about its work there -- C2 at 64 orbit frames per launch 0.06509 -> 0.06467 ms without)
        if (!SKIP && (L.debug_flags & 16u)) set_wave_priority(true, t0, t1, dt, fmaxf(fnx, fmaxf(fny, fnz)) / L.dt_scale);
        if constexpr (USE_LUT && !SKIP) march_stream<VOL, COUNT>(V, r, cs, cell_lut);
        else if constexpr (SKIP) {
            if (L.debug_flags & 4u) {
                // Adaptive probing (wave-uniform policy, any policy is exact: a sampled empty cell adds +0).  Probe for a
                // window of 16 trips; if fewer than 1 in 8 of the wave's live rays skipped anything in it, the wave is in
                // material that cannot be skipped: run the dense loop -- no distance look-up, and on the fast path
                // software-pipelined -- for a stretch that doubles every time the next window confirms it (64 .. 512
                // trips), then probe again.  Fog pays ~9 % of its trips at the probing price instead of all of them.
                const uint32_t stretch0 = (L.debug_flags & 8u) ? 256u : 64u;  // bit 3: the census found (almost) nothing to skip
                uint32_t stretch = stretch0;
                for (;;) {
                    cs.skips = 0;
                    bool alive = march<VOL, true, SAFE, COUNT, true>(V, r, 16u, cs, USE_LUT ? cell_lut : nullptr, L.walk_cap, L.walk_cap_all);
                    const unsigned long long live = __ballot(alive);
                    if (live == 0ull) break;
                    if (__popcll(__ballot(alive && cs.skips != 0u)) * 8 >= __popcll(live)) { stretch = stretch0; continue; }
                    if constexpr (USE_LUT) alive = march_stream<VOL, COUNT, true>(V, r, cs, cell_lut, stretch);
                    else alive = march<VOL, false, SAFE, COUNT, true>(V, r, stretch, cs, nullptr);
                    if (__ballot(alive) == 0ull) break;
                    stretch = min(stretch * 2u, 512u);
                }
            } else {
                march<VOL, SKIP, SAFE, COUNT>(V, r, 0xffffffffu, cs, USE_LUT ? cell_lut : nullptr, L.walk_cap, L.walk_cap_all);
            }
        }
        else if constexpr (VOL == VOL_B9U8 || VOL == VOL_B9F16) march_b9_stream<VOL, COUNT>(V, r, cs);
        else if constexpr (VOL == VOL_Q8 || VOL == VOL_QF16) march_quads_stream<VOL, COUNT>(V, r, cs);
        else march<VOL, SKIP, SAFE, COUNT>(V, r, 0xffffffffu, cs, USE_LUT ? cell_lut : nullptr);
        A = r.A; Gr = r.Gr; Gg = r.Gg; Gb = r.Gb;
        Cr = linear_to_srgb(fmaf(0.5f, Gr, 0.5f * A));  // :121-123
        Cg = linear_to_srgb(fmaf(0.5f, Gg, 0.5f * A));
        Cb = linear_to_srgb(fmaf(0.5f, Gb, 0.5f * A));
    }
    store_out<OUT>(L, pm, Cr, Cg, Cb);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = (L.debug_flags & 1u) ? cs.n_look : cs.n_iter;
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)cs.n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)cs.n_samp);
            atomicAdd(&L.counters[2], (unsigned long long)cs.w_outer);
            atomicAdd(&L.counters[3], (unsigned long long)cs.w_inner);
            atomicAdd(&L.counters[4], (unsigned long long)cs.w_sample);
            atomicAdd(&L.counters[5], (unsigned long long)cs.n_look);
        }
        if (L.trace && !trip_log) {  // stamps leave only through this debug buffer
            unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            atomicMin(&L.trace[4 * (size_t)lb], t_start);
            atomicMax(&L.trace[4 * (size_t)lb + 1], t_end);
            // where the wave ran: HW_ID (wave/simd/cu/sh/se fields) and XCC_ID
            L.trace[4 * (size_t)lb + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            // wave-level work: march-loop trips | skip-walk trips << 20 | sample executions << 40
            atomicAdd(&L.trace[4 * (size_t)lb + 3], (unsigned long long)cs.w_outer | ((unsigned long long)cs.w_inner << 20) | ((unsigned long long)cs.w_sample << 40));
        }
    }
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
constexpr uint32_t kPairOob = 0x50000000u;  // > any record offset (array <= kPairOob bytes); 3 markers do not wrap
__host__ __device__ __forceinline__ uint32_t pair_lut_entries(uint32_t nx, uint32_t ny, uint32_t nz) { return (nx + ny + nz + 6u * kPairPad + 3u) & ~3u; }

__global__ __launch_bounds__(256) void build_pair_luts_kernel(uint32_t *__restrict__ out, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t nbx, uint32_t nby) {
    const uint32_t n0 = nx + 2u * kPairPad, n1 = ny + 2u * kPairPad, n2 = nz + 2u * kPairPad, total = n0 + n1 + n2, padded = pair_lut_entries(nx, ny, nz);
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < padded; e += gridDim.x * blockDim.x) {
        uint32_t v = kPairOob;
        if (e < total) {
            uint32_t j = e, n = nx, brick_mul = 64u, cell_mul = 1u;
            if (e >= n0 + n1) { j = e - n0 - n1; n = nz; brick_mul = 64u * nbx * nby; cell_mul = 16u; }
            else if (e >= n0) { j = e - n0; n = ny; brick_mul = 64u * nbx; cell_mul = 4u; }
            if (j >= kPairPad && j < n + kPairPad) { const uint32_t i = j - kPairPad; v = (brick_mul * (i >> 2) + cell_mul * (i & 3u)) << 4; }
        }
        out[e] = v;
    }
}

// dense x-fastest (density, normals) -> bricked 16-byte records
__global__ __launch_bounds__(256) void pack_pairs_kernel(const uint2 *__restrict__ den, const uint2 *__restrict__ nrm, uint4 *__restrict__ dst,
                                                         uint32_t nx, uint32_t ny, uint32_t nz, uint32_t nbx, uint32_t nby, uint64_t n_rec) {
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_rec; id += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t brick = id >> 6;
        const uint32_t w = (uint32_t)(id & 63u);
        const uint32_t bx = (uint32_t)(brick % nbx);
        const uint64_t rest = brick / nbx;
        const uint32_t by = (uint32_t)(rest % nby), bz = (uint32_t)(rest / nby);
        const uint32_t x = bx * 4 + (w & 3u), y = by * 4 + ((w >> 2) & 3u), z = bz * 4 + (w >> 4);
        uint4 r = make_uint4(0, 0, 0, 0);
        if (x < nx && y < ny && z < nz) {
            const size_t src = (size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)z);
            const uint2 d = den[src], n = nrm[src];
            r = make_uint4(d.x, d.y, n.x, n.y);
        }
        dst[id] = r;
    }
}

// Same arithmetic as raymarch_compute_kernel below (raycast_compute.wgsl:62-131), one 16-byte record
// per step, software-pipelined: p = eye + t*dir does not depend on the loads, so the next step's
// record is requested before this step is shaded.  A lone wave of this mode used to pay a full
// cache-miss latency per step (<= 346 dependent steps per ray).
template <int OUT, bool COUNT>
__global__ __launch_bounds__(64) void raymarch_compute_records_kernel(const LaunchDesc L, const VolumeDesc V) {
    const uint32_t lb = logical_block(blockIdx.x);
    if (lb >= L.n_blocks) return;
    const uint32_t lane = threadIdx.x;
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, lane);
    extern __shared__ uint32_t pair_lut[];
    {
        const uint32_t n4 = pair_lut_entries(V.nx, V.ny, V.nz) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(V.lut);
        uint4 *dst = reinterpret_cast<uint4 *>(pair_lut);
        for (uint32_t e = lane; e < n4; e += 64u) dst[e] = src[e];
        __syncthreads();
    }
    if (!pm.valid) return;

    float dimx = (float)L.W, dimy = (float)L.H;
    float aspect_ratio = dimy / dimx;
    float scx = 2.0f * (float)pm.x / dimx - 1.0f;
    float scy = 2.0f * (float)pm.y / dimy - 1.0f;
    scy = scy * -aspect_ratio;
    float vp[4], vt[4];
    mat4_mul_vec4(fv.inv_proj, scx, scy, 0.0f, 1.0f, vp);
    mat4_mul_vec4(fv.inv_proj, scx, scy, 1.0f, 1.0f, vt);
    const float eye[3] = {vp[0] / vp[3], vp[1] / vp[3], vp[2] / vp[3]};
    float dir[3] = {vt[0] / vt[3] - eye[0], vt[1] / vt[3] - eye[1], vt[2] / vt[3] - eye[2]};
    normalize3(dir[0], dir[1], dir[2]);

    const float clr[3] = {0.023f, 0.02f, 0.02f};  // :118, clear alpha 0
    float C[3] = {clr[0], clr[1], clr[2]};
    uint32_t n_iter = 0;
    float t0, t1;
    intersect_box(eye, dir, -1.0f, 1.0f, t0, t1);
    if (t0 < t1) {  // :123
        t0 = fmaxf(t0, 0.0f);
        float A = 0.1f;  // get_col2 :63
        const float bsx = (float)V.nx, bsy = (float)V.ny, bsz = (float)V.nz;
        float dtx = 1.0f / (bsx * fabsf(dir[0]));
        float dty = 1.0f / (bsy * fabsf(dir[1]));
        float dtz = 1.0f / (bsz * fabsf(dir[2]));
        const float dt = L.dt_scale * fmaxf(fminf(dtx, fminf(dty, dtz)), 0.01f);  // :66-68
        const float hbx = bsx / 2.0f, hby = bsy / 2.0f, hbz = bsz / 2.0f;
        float l1x = -2.0f, l1y = -2.0f, l1z = -1.0f, l2x = 1.0f, l2y = 1.0f, l2z = -1.0f;
        normalize3(l1x, l1y, l1z);
        normalize3(l2x, l2y, l2z);
        const uint32_t *luty = pair_lut + (V.nx + 2u * kPairPad), *lutz = pair_lut + (V.nx + V.ny + 4u * kPairPad);
        const __amdgpu_buffer_rsrc_t recs = cell_buffer(V.data, (uint32_t)V.max_off + 16u);
        struct Req { float px, py, pz; u32x4_t r; };
        auto request = [&](float t) -> Req {
            Req q;
            q.px = eye[0] + t * dir[0]; q.py = eye[1] + t * dir[1]; q.pz = eye[2] + t * dir[2];
            // ivec3 truncation (:75); |p| stays within a few steps of the box, so the conversions are in range
            const int ix = (int)((q.px + 1.0f) * hbx), iy = (int)((q.py + 1.0f) * hby), iz = (int)((q.pz + 1.0f) * hbz);
            const uint32_t off = pair_lut[ix + (int)kPairPad] + luty[iy + (int)kPairPad] + lutz[iz + (int)kPairPad];
            q.r = __builtin_amdgcn_raw_buffer_load_b128(recs, (int)off, 0, 0);
            return q;
        };
        float t = t0;
        Req cur = request(t), nxt = cur;
        for (;;) {  // :69, entered with t < t1
            const float tn = t + dt;
            nxt = request(tn);
            const float px = cur.px, py = cur.py, pz = cur.pz;
            const uint32_t d0 = cur.r.x, d1 = cur.r.y, m0 = cur.r.z, m1 = cur.r.w;
            float vc0 = h2f(d0 & 0xffffu), vc1 = h2f(d0 >> 16), vc2 = h2f(d1 & 0xffffu), vc3 = h2f(d1 >> 16);
            float n0 = h2f(m0 & 0xffffu), n1 = h2f(m0 >> 16), n2 = h2f(m1 & 0xffffu);
            n_iter++;
            // The shader's literal expressions (kept word for word in raymarch_compute_kernel below, which the tests hold
            // this kernel to bit for bit) carry terms that are zero for every finite record: dot((0,-1,0), n) is -n.y,
            // mix(shade, bl * (0,0,0.6), 0.2) has zero red and green contributions from bl, and clear.rgb * clear.a * (1 - a)
            // is 0 * (1 - a).  IEEE arithmetic forbids the compiler to drop them (0 * x is NaN for an infinite x); with finite
            // taps they only ever add a zero to a non-zero accumulator, so leaving them out changes no bit: 15 of the step's
            // 85 instructions.  A volume with infinities or NaNs renders differently from the literal form.
            float sh = fmaxf(0.0f, -n1);
            float va = (vc3 * vc3) * vc3;
            va = smoothstepf(0.0f, 0.7f, va);
            float dl = fmaxf((n0 * l1x + n1 * l1y) + n2 * l1z, 0.0f);
            float ss = smoothstepf(0.3f, 1.5f, (px * l2x + py * l2y) + pz * l2z);
            float col0 = vc0 + 3.0f * 1.0f * dl * ss, col1 = vc1 + 3.0f * 0.1f * dl * ss, col2 = vc2 + 3.0f * 0.13f * dl * ss;
            float bl = 0.9f * fminf(fmaxf(0.5f - 0.5f * n1, 0.0f), 1.0f);
            float sh0 = sh * (1.0f - 0.2f);
            float sh1 = sh0;
            float sh2 = sh * (1.0f - 0.2f) + (bl * 0.6f) * 0.2f;
            float w = (1.0f - A) * va;
            C[0] = C[0] + w * col0 * sh0;
            C[1] = C[1] + w * col1 * sh1;
            C[2] = C[2] + w * col2 * sh2;
            A = A + w;
            if (A >= 0.95f) break;
            t = tn;
            if (!(t < t1)) break;
            cur = nxt;
        }
        asm volatile("" ::"v"(nxt.r));  // the last request is consumed on the exit path too (keeps it ahead of the shading)
    }
    store_out<OUT>(L, pm, C[0], C[1], C[2]);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = n_iter;
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)n_iter);
        }
    }
}

template <int OUT, bool COUNT>
__global__ __launch_bounds__(64) void raymarch_compute_kernel(const LaunchDesc L, const VolumeDesc V) {
    const uint32_t lb = logical_block(blockIdx.x);
    if (lb >= L.n_blocks) return;
    const uint32_t lane = threadIdx.x;
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, lane);
    if (!pm.valid) return;

    // render(): raycast_compute.wgsl:99-116 -- no half-pixel offset, y scaled by -H/W
    float dimx = (float)L.W, dimy = (float)L.H;
    float aspect_ratio = dimy / dimx;
    float scx = 2.0f * (float)pm.x / dimx - 1.0f;
    float scy = 2.0f * (float)pm.y / dimy - 1.0f;
    scy = scy * -aspect_ratio;
    float vp[4], vt[4];
    mat4_mul_vec4(fv.inv_proj, scx, scy, 0.0f, 1.0f, vp);
    mat4_mul_vec4(fv.inv_proj, scx, scy, 1.0f, 1.0f, vt);
    const float eye[3] = {vp[0] / vp[3], vp[1] / vp[3], vp[2] / vp[3]};
    float dir[3] = {vt[0] / vt[3] - eye[0], vt[1] / vt[3] - eye[1], vt[2] / vt[3] - eye[2]};
    normalize3(dir[0], dir[1], dir[2]);

    const float clr[3] = {0.023f, 0.02f, 0.02f};  // :118, clear alpha 0
    float C[3] = {clr[0], clr[1], clr[2]};
    uint32_t n_iter = 0;
    float t0, t1;
    intersect_box(eye, dir, -1.0f, 1.0f, t0, t1);
    if (t0 < t1) {  // :123
        t0 = fmaxf(t0, 0.0f);
        float A = 0.1f;  // get_col2 :63
        const float bsx = (float)V.nx, bsy = (float)V.ny, bsz = (float)V.nz;
        float dtx = 1.0f / (bsx * fabsf(dir[0]));
        float dty = 1.0f / (bsy * fabsf(dir[1]));
        float dtz = 1.0f / (bsz * fabsf(dir[2]));
        const float dt = L.dt_scale * fmaxf(fminf(dtx, fminf(dty, dtz)), 0.01f);  // :66-68
        const float hbx = bsx / 2.0f, hby = bsy / 2.0f, hbz = bsz / 2.0f;
        float l1x = -2.0f, l1y = -2.0f, l1z = -1.0f, l2x = 1.0f, l2y = 1.0f, l2z = -1.0f;
        normalize3(l1x, l1y, l1z);
        normalize3(l2x, l2y, l2z);
        const uint2 *den = reinterpret_cast<const uint2 *>(V.data);
        const uint2 *nrm = reinterpret_cast<const uint2 *>(V.data2);
        for (float t = t0; t < t1; t = t + dt) {  // :69
            float px = eye[0] + t * dir[0], py = eye[1] + t * dir[1], pz = eye[2] + t * dir[2];
            int ix = (int)((px + 1.0f) * hbx), iy = (int)((py + 1.0f) * hby), iz = (int)((pz + 1.0f) * hbz);
            // textureLoad with naga's Unchecked bounds policy: this build defines OOB as zeros (A.2)
            bool inb = ix >= 0 && iy >= 0 && iz >= 0 && ix < (int)V.nx && iy < (int)V.ny && iz < (int)V.nz;
            uint2 dv = make_uint2(0, 0), nv = make_uint2(0, 0);
            if (inb) {
                size_t idx = (size_t)ix + (size_t)V.nx * ((size_t)iy + (size_t)V.ny * (size_t)iz);
                dv = den[idx];
                nv = nrm[idx];
            }
            float vc0 = h2f(dv.x & 0xffffu), vc1 = h2f(dv.x >> 16), vc2 = h2f(dv.y & 0xffffu), vc3 = h2f(dv.y >> 16);
            float n0 = h2f(nv.x & 0xffffu), n1 = h2f(nv.x >> 16), n2 = h2f(nv.y & 0xffffu);
            n_iter++;
            float sh = fmaxf(0.0f, (0.0f * n0 + -1.0f * n1) + 0.0f * n2);
            float va = (vc3 * vc3) * vc3;
            va = smoothstepf(0.0f, 0.7f, va);
            float dl = fmaxf((n0 * l1x + n1 * l1y) + n2 * l1z, 0.0f);
            float ss = smoothstepf(0.3f, 1.5f, (px * l2x + py * l2y) + pz * l2z);
            float col0 = vc0 + 3.0f * 1.0f * dl * ss, col1 = vc1 + 3.0f * 0.1f * dl * ss, col2 = vc2 + 3.0f * 0.13f * dl * ss;
            float bl = 0.9f * fminf(fmaxf(0.5f - 0.5f * n1, 0.0f), 1.0f);
            float sh0 = sh * (1.0f - 0.2f) + (bl * 0.0f) * 0.2f;
            float sh1 = sh0;
            float sh2 = sh * (1.0f - 0.2f) + (bl * 0.6f) * 0.2f;
            float w = (1.0f - A) * va;
            C[0] = (C[0] + w * col0 * sh0) + clr[0] * 0.0f * (1.0f - va);
            C[1] = (C[1] + w * col1 * sh1) + clr[1] * 0.0f * (1.0f - va);
            C[2] = (C[2] + w * col2 * sh2) + clr[2] * 0.0f * (1.0f - va);
            A = A + w * (1.0f - 0.0f);
            if (A >= 0.95f) break;
        }
    }
    store_out<OUT>(L, pm, C[0], C[1], C[2]);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = n_iter;
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)n_iter);
        }
    }
}

// ---- volume re-layout ------------------------------------------------------------------------
// One thread per cell, cells enumerated in storage order (coalesced 8/16-byte stores).  Physical
// brick B = (i >> 2) + 1 and in-brick w = i & 3 per axis, i = low-corner voxel index in [-1, n-1].
template <int VOL>
__global__ __launch_bounds__(256) void pack_cells_kernel(const void *__restrict__ src, void *__restrict__ dst,
                                                          uint8_t *__restrict__ occ, uint32_t nx, uint32_t ny,
                                                          uint32_t nz, uint32_t nbx, uint32_t nby, uint64_t n_cells,
                                                          unsigned long long *__restrict__ n_empty) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_cells) return;
    uint64_t brick = id >> 6;
    uint32_t w = (uint32_t)(id & 63u);
    uint32_t bx = (uint32_t)(brick % nbx);
    uint64_t rest = brick / nbx;
    uint32_t by = (uint32_t)(rest % nby), bz = (uint32_t)(rest / nby);
    int ix = ((int)bx - 1) * 4 + (int)(w & 3u), iy = ((int)by - 1) * 4 + (int)((w >> 2) & 3u), iz = ((int)bz - 1) * 4 + (int)(w >> 4);
    int mx = (int)nx - 1, my = (int)ny - 1, mz = (int)nz - 1;
    int xs[2] = {clampi(ix, 0, mx), clampi(ix + 1, 0, mx)};
    int ys[2] = {clampi(iy, 0, my), clampi(iy + 1, 0, my)};
    int zs[2] = {clampi(iz, 0, mz), clampi(iz + 1, 0, mz)};
    uint32_t t[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        size_t idx = (size_t)xs[b & 1] + (size_t)nx * ((size_t)ys[(b >> 1) & 1] + (size_t)ny * (size_t)zs[b >> 2]);
        t[b] = (VOL == VOL_PF16) ? (uint32_t)reinterpret_cast<const uint16_t *>(src)[idx]
                                 : (uint32_t)reinterpret_cast<const uint8_t *>(src)[idx];
    }
    // occ = 0 if any tap is above the transfer function's zero threshold (u8 > 25: 25/255 < 0.1 <=
    // 26/255; f16 > 0.1f or NaN), else 255 ("no contributing cell seen yet")
    bool nonempty = false;
#pragma unroll
    for (int b = 0; b < 8; b++) nonempty |= (VOL == VOL_PF16) ? !(h2f(t[b]) <= 0.1f) : (t[b] > 25u);
    occ[id] = nonempty ? 0 : 255;
    {   // census of exactly-transparent cells (one atomic per wave): decides whether skipping can pay
        const unsigned long long m = __ballot(!nonempty);
        if ((threadIdx.x & 63u) == 0 && m) atomicAdd(n_empty, (unsigned long long)__popcll(m));
    }
    if (VOL == VOL_P8) {
        uint32_t lo = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
        uint32_t hi = t[4] | (t[5] << 8) | (t[6] << 16) | (t[7] << 24);
        reinterpret_cast<uint2 *>(dst)[id] = make_uint2(lo, hi);
    } else if (VOL == VOL_P16) {
        union { uint4 u; _Float16 h[8]; } c;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c.h[2 * k] = (_Float16)(float)t[2 * k];                                  // tap (dx = 0)
            c.h[2 * k + 1] = (_Float16)((float)t[2 * k + 1] - (float)t[2 * k]);      // delta, |.| <= 255: exact
        }
        reinterpret_cast<uint4 *>(dst)[id] = c.u;
    } else {
        reinterpret_cast<uint4 *>(dst)[id] = make_uint4(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16));
    }
}

// Dense voxels -> 9^3 bricks: one thread per stored voxel, brick b holds voxels [8b-1, 8b+7] per axis
// (clamped to the volume: clamp-to-edge is baked in).
template <bool F16>
__global__ __launch_bounds__(256) void pack_bricks9_kernel(const void *__restrict__ src, void *__restrict__ dst, uint32_t nx,
                                                           uint32_t ny, uint32_t nz, uint32_t nbx, uint32_t nby, uint64_t n_elems) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_elems; id += stride) {  // may exceed 2^32
        uint64_t brick = id / 729u;
        uint32_t l = (uint32_t)(id - brick * 729u);
        uint32_t lz = l / 81u, ly = (l - lz * 81u) / 9u, lx = l - lz * 81u - ly * 9u;
        uint32_t bx = (uint32_t)(brick % nbx);
        uint64_t rest = brick / nbx;
        uint32_t by = (uint32_t)(rest % nby), bz = (uint32_t)(rest / nby);
        int x = clampi((int)(bx * 8 + lx) - 1, 0, (int)nx - 1), y = clampi((int)(by * 8 + ly) - 1, 0, (int)ny - 1);
        int z = clampi((int)(bz * 8 + lz) - 1, 0, (int)nz - 1);
        size_t idx = (size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)z);
        if (F16) reinterpret_cast<uint16_t *>(dst)[id] = reinterpret_cast<const uint16_t *>(src)[idx];
        else reinterpret_cast<uint8_t *>(dst)[id] = reinterpret_cast<const uint8_t *>(src)[idx];
    }
}

// Dense voxels -> quad elements in 9x8x8 bricks.  Padded coordinate c = i + 1 (i = low-corner voxel of a
// footprint, i in [-1, n-1]) maps to voxel clamp(c - 1); brick (cx>>3, cy>>3, cz>>3), local (cx&7 .. with the
// x apron lx = 8 repeating the next brick's lx = 0).  One thread per element.
template <bool F16>
__global__ __launch_bounds__(256) void pack_quads_kernel(const void *__restrict__ src, void *__restrict__ dst, uint32_t nx, uint32_t ny,
                                                         uint32_t nz, uint32_t nbx, uint32_t nby, uint64_t n_elems) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_elems; id += stride) {
        const uint64_t brick = id / 576u;
        const uint32_t l = (uint32_t)(id - brick * 576u);
        const uint32_t lz = l / 72u, ly = (l - lz * 72u) / 9u, lx = l - lz * 72u - ly * 9u;
        const uint32_t bx = (uint32_t)(brick % nbx);
        const uint64_t rest = brick / nbx;
        const uint32_t by = (uint32_t)(rest % nby), bz = (uint32_t)(rest / nby);
        const int cx = (int)(bx * 8 + lx), cy = (int)(by * 8 + ly), cz = (int)(bz * 8 + lz);
        const int x = clampi(cx - 1, 0, (int)nx - 1);
        const int y0 = clampi(cy - 1, 0, (int)ny - 1), y1 = clampi(cy, 0, (int)ny - 1);
        const int z0 = clampi(cz - 1, 0, (int)nz - 1), z1 = clampi(cz, 0, (int)nz - 1);
        const size_t sy_ = nx, sz_ = (size_t)nx * ny;
        const size_t i00 = x + y0 * sy_ + z0 * sz_, i10 = x + y1 * sy_ + z0 * sz_, i01 = x + y0 * sy_ + z1 * sz_, i11 = x + y1 * sy_ + z1 * sz_;
        if (F16) {
            const uint16_t *v = reinterpret_cast<const uint16_t *>(src);
            reinterpret_cast<uint2 *>(dst)[id] = make_uint2((uint32_t)v[i00] | ((uint32_t)v[i10] << 16), (uint32_t)v[i01] | ((uint32_t)v[i11] << 16));
        } else {
            const uint8_t *v = reinterpret_cast<const uint8_t *>(src);
            reinterpret_cast<uint32_t *>(dst)[id] = (uint32_t)v[i00] | ((uint32_t)v[i10] << 8) | ((uint32_t)v[i01] << 16) | ((uint32_t)v[i11] << 24);
        }
    }
}

// One separable pass of the Chebyshev (L-infinity) distance transform over the cells:
// out(c) = min_j max(in(c + j*axis), |j|), |j| <= kDistRadius, j restricted to j >= 0 (dir > 0),
// j <= 0 (dir < 0) or unrestricted (dir == 0).  Outside the grid counts as empty.
// Cells are addressed in their bricked storage order.
__device__ __forceinline__ uint64_t cell_index(uint32_t x, uint32_t y, uint32_t z, uint32_t nbx, uint32_t nby) {
    uint64_t brick = ((uint64_t)(z >> 2) * nby + (y >> 2)) * nbx + (x >> 2);
    return brick * 64 + (((z & 3u) << 4) | ((y & 3u) << 2) | (x & 3u));
}

__global__ __launch_bounds__(256) void dist_pass_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                        uint32_t nbx, uint32_t nby, uint32_t nbz, int axis, int dir, int last) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t n = (uint64_t)nbx * nby * nbz * 64;
    if (id >= n) return;
    uint64_t brick = id >> 6;
    uint32_t w = (uint32_t)(id & 63u);
    uint32_t bx = (uint32_t)(brick % nbx);
    uint64_t rest = brick / nbx;
    uint32_t by = (uint32_t)(rest % nby), bz = (uint32_t)(rest / nby);
    uint32_t c[3] = {bx * 4 + (w & 3u), by * 4 + ((w >> 2) & 3u), bz * 4 + (w >> 4)};
    const int dim = (int)(axis == 0 ? nbx : (axis == 1 ? nby : nbz)) * 4;
    const int c0 = (int)c[axis];
    int best = in[id];
    const int jlo = dir > 0 ? 0 : max(-kDistRadius, -c0), jhi = dir < 0 ? 0 : min(kDistRadius, dim - 1 - c0);
    for (int j = jlo; j <= jhi; j++) {
        const int aj = j < 0 ? -j : j;
        if (aj >= best) continue;  // cannot improve
        uint32_t q[3] = {c[0], c[1], c[2]};
        q[axis] = (uint32_t)(c0 + j);
        const int v = in[cell_index(q[0], q[1], q[2], nbx, nby)];
        best = min(best, max(v, aj));
    }
    if (last) best = min(best, kDistRadius + 1);
    out[id] = (uint8_t)best;
}

// ---- misc ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed) {
    return lowbias32(seed ^ (x * 0x9E3779B1U + y * 0x85EBCA77U + z * 0xC2B2AE3DU));
}

// Deterministic synthetic volumes, integer arithmetic only (bit-identical to the oracle's
// vo_volume_fog_u8 / vo_volume_fog_f16 / vo_volume_standin_u8 and to vokselis_amd/volumes.py).
__device__ __forceinline__ uint32_t vnoise(uint32_t X, uint32_t Y, uint32_t Z, uint32_t sh, uint32_t seed) {
    const uint32_t m = (1u << sh) - 1, S = 1u << sh;
    const uint32_t cx = X >> sh, cy = Y >> sh, cz = Z >> sh;
    const uint32_t fx = X & m, fy = Y & m, fz = Z & m;
    uint64_t acc = 0;
#pragma unroll
    for (uint32_t dz = 0; dz < 2; dz++)
#pragma unroll
        for (uint32_t dy = 0; dy < 2; dy++)
#pragma unroll
            for (uint32_t dx = 0; dx < 2; dx++) {
                uint64_t w = (uint64_t)(dx ? fx : S - fx) * (dy ? fy : S - fy) * (dz ? fz : S - fz);
                acc += w * (hash3(cx + dx, cy + dy, cz + dz, seed) >> 24);
            }
    return (uint32_t)(acc >> (3 * sh));
}

// "bonsai stand-in": dense pot (>= 232), mid-density bent trunk, smooth noise-thresholded canopy,
// air = white noise 0..20 (exactly transparent) with 0.2 % speckle 26..41.  SURVEY 8(d) C1.
__device__ __forceinline__ uint32_t standin_voxel(uint32_t x, uint32_t y, uint32_t z, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed) {
    const int32_t X = (int32_t)(((2 * (uint64_t)x + 1) * 2048) / nx);
    const int32_t Y = (int32_t)(((2 * (uint64_t)y + 1) * 2048) / ny);
    const int32_t Z = (int32_t)(((2 * (uint64_t)z + 1) * 2048) / nz);
    const uint32_t n_lo = vnoise((uint32_t)X, (uint32_t)Y, (uint32_t)Z, 9, seed ^ 0x1111u);
    const uint32_t n_hi = vnoise((uint32_t)X, (uint32_t)Y, (uint32_t)Z, 7, seed ^ 0x2222u);
    {
        int64_t dx = X - 2048, dy = Y - 600, dz = Z - 2048;
        if (dx * dx + 7 * dy * dy + dz * dz < 1400 * 1400) return 232 + (n_hi >> 4);
    }
    if (Y >= 900 && Y < 2600) {
        int32_t h = Y - 900;
        int64_t cx = 2048 + ((int64_t)h * h) / 8000, cz = 2048 - h / 6, rr = 230 - h / 12;
        int64_t dx = X - cx, dz = Z - cz;
        if (dx * dx + dz * dz < rr * rr) return 110 + (n_hi >> 2);
    }
    {
        int64_t dx = X - 2150, dy = Y - 2850, dz = Z - 1950;
        int64_t q = (dx * dx * 256) / (1750 * 1750) + (dy * dy * 256) / (1050 * 1050) + (dz * dz * 256) / (1750 * 1750);
        if (q < 256) {
            int32_t f = (int32_t)((2 * n_lo + n_hi) / 3);
            int32_t d = f - (int32_t)(q / 3) - 52;
            if (d > 0) { int32_t v = 28 + 2 * d; return (uint32_t)(v > 225 ? 225 : v); }
        }
    }
    const uint32_t h = hash3(x, y, z, seed ^ 0x3333u);
    if ((h & 0x1ffu) == 0) return 26 + ((h >> 9) & 15);
    return (h >> 16) % 21;
}

// kind 0: fog u8 in [lo, lo+span); 1: fog f16 bit patterns 0x2D1F + h % 656; 2: bonsai stand-in u8.
// core: the fog with a dense ball at the centre (SURVEY 8d, C4 / C5 "dense-core variant"): a voxel is in the core iff
// (2x+1-nx)^2 + (2y+1-ny)^2 + (2z+1-nz)^2 < (min(nx,ny,nz)/2)^2 (radius: a quarter of the smallest dimension); there
// u8 = 232 + h % 24 (alpha per step >= 0.8: a ray that enters leaves the loop within two steps), f16 = 0x3B9A + h % 64
// (0.95 .. 0.98).  Integer arithmetic only.
__host__ __device__ __forceinline__ bool in_dense_core(uint32_t x, uint32_t y, uint32_t z, uint32_t nx, uint32_t ny, uint32_t nz) {
    const int64_t dx = 2 * (int64_t)x + 1 - (int64_t)nx, dy = 2 * (int64_t)y + 1 - (int64_t)ny, dz = 2 * (int64_t)z + 1 - (int64_t)nz;
    const int64_t r = (int64_t)(nx < ny ? (nx < nz ? nx : nz) : (ny < nz ? ny : nz)) / 2;
    return dx * dx + dy * dy + dz * dz < r * r;
}
template <int KIND>
__global__ __launch_bounds__(256) void generate_kernel(void *__restrict__ dst, uint32_t nx, uint32_t ny, uint32_t nz,
                                                       uint32_t seed, uint32_t lo, uint32_t span, uint32_t core) {
    const uint64_t n = (uint64_t)nx * ny * nz, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += stride) {  // grid-stride: n may exceed 2^32
        uint32_t x = (uint32_t)(id % nx);
        uint64_t rest = id / nx;
        uint32_t y = (uint32_t)(rest % ny), z = (uint32_t)(rest / ny);
        if (KIND == 2) {
            reinterpret_cast<uint8_t *>(dst)[id] = (uint8_t)standin_voxel(x, y, z, nx, ny, nz, seed);
        } else {
            uint32_t h = hash3(x, y, z, seed) >> 8;
            const bool dense = core && in_dense_core(x, y, z, nx, ny, nz);
            if (KIND == 1) reinterpret_cast<uint16_t *>(dst)[id] = (uint16_t)(dense ? 0x3B9Au + h % 64u : 0x2D1Fu + h % 656u);
            else reinterpret_cast<uint8_t *>(dst)[id] = (uint8_t)(dense ? 232u + h % 24u : lo + h % span);
        }
    }
}

// ---- xor example's volume generator (next row N3): shaders/xor.wgsl:18-78 --------------------
// cs_main for every voxel: fbm value noise (3 octaves x 8 sin-hashes) and its finite-difference
// gradient (3 more evaluations).  ALU-bound: 96 hashes per voxel.  hash()'s sine is the specified
// one (f64 Cody-Waite + minimax polynomial, rounded once to f32), so the volume is reproducible.
// a * b + k with the (wave-uniform, loop-invariant) coefficient k read from a scalar register pair.  Written out
// because the compiler otherwise turns every Horner step into v_mov_b64 (copy the coefficient) + v_fmac_f64: 258 of
// the procedural loop's 1068 instructions were such copies.  Same single-rounding fma, bit for bit.
__device__ __forceinline__ double fma_k(double a, double b, double k) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

__device__ __forceinline__ float sin_spec(float h) {
    const double x = (double)h;
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double r2 = r * r;
    double sp = 1.58969099521155010221e-10;
    sp = fma_k(sp, r2, -2.50507602534068634195e-08);
    sp = fma_k(sp, r2, 2.75573137070700676789e-06);
    sp = fma_k(sp, r2, -1.98412698298579493134e-04);
    sp = fma_k(sp, r2, 8.33333333332248946124e-03);
    sp = fma_k(sp, r2, -1.66666666666666324348e-01);
    const double sn = fma(r * r2, sp, r);
    double cp = -1.13596475577881948265e-11;
    cp = fma_k(cp, r2, 2.08757232129817482790e-09);
    cp = fma_k(cp, r2, -2.75573143513906633035e-07);
    cp = fma_k(cp, r2, 2.48015872894767294178e-05);
    cp = fma_k(cp, r2, -1.38888888888741095749e-03);
    cp = fma_k(cp, r2, 4.16666666666666019037e-02);
    const double cs = fma(r2 * r2, cp, fma(-0.5, r2, 1.0));
    // Quadrant (|k| < 2^31 for every argument the hash makes): q = k & 3 -> sn, cs, -sn, -cs.  Written on the bit patterns -- odd q takes
    // the cosine series (v_bfi_b32 on both halves), bit 1 of q flips the sign bit -- because the compiler lowers the four-way select of
    // doubles to two nested exec-mask branches per sine (6 scalar instructions, 3 compares, 2 v_cndmask through VCC: about a third of a
    // sine's issue cycles by profiles/r03_ubench_valu_issue_rate.txt, 24 sines per step).  The same value bit for bit: a negated double
    // differs in its sign bit only.
    const uint32_t qi = (uint32_t)(int)k;
    union { double d; uint32_t u[2]; } S, C, R;
    S.d = sn; C.d = cs;
    const uint32_t odd = 0u - (qi & 1u);  // all ones: the cosine series
    uint32_t lo, hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(odd), "v"(C.u[0]), "v"(S.u[0]));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(odd), "v"(C.u[1]), "v"(S.u[1]));
    R.u[0] = lo;
    R.u[1] = hi ^ ((qi << 30) & 0x80000000u);
    return (float)R.d;
}
__device__ __forceinline__ float xor_fract(float x) { return x - floorf(x); }
__device__ __forceinline__ float xor_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float xor_hash(float h) { return xor_fract(sin_spec(h) * 43758.5453123f); }
__device__ __forceinline__ float xor_noise(float x0, float x1, float x2) {
    const float p0 = floorf(x0), p1 = floorf(x1), p2 = floorf(x2);
    float f0 = xor_fract(x0), f1 = xor_fract(x1), f2 = xor_fract(x2);
    f0 = f0 * f0 * (3.0f - 2.0f * f0); f1 = f1 * f1 * (3.0f - 2.0f * f1); f2 = f2 * f2 * (3.0f - 2.0f * f2);
    const float n = p0 + p1 * 157.0f + 113.0f * p2;
    return xor_mix(xor_mix(xor_mix(xor_hash(n + 0.0f), xor_hash(n + 1.0f), f0), xor_mix(xor_hash(n + 157.0f), xor_hash(n + 158.0f), f0), f1),
                   xor_mix(xor_mix(xor_hash(n + 113.0f), xor_hash(n + 114.0f), f0), xor_mix(xor_hash(n + 270.0f), xor_hash(n + 271.0f), f0), f1),
                   f2);
}
__device__ __forceinline__ float xor_fbm(float p0, float p1, float p2) {
    float f = 0.5000f * xor_noise(p0, p1, p2);
    p0 = p0 * 2.01f; p1 = p1 * 2.01f; p2 = p2 * 2.01f;
    f = f + 0.2500f * xor_noise(p0, p1, p2);
    p0 = p0 * 2.02f; p1 = p1 * 2.02f; p2 = p2 * 2.02f;
    f = f + 0.1250f * xor_noise(p0, p1, p2);
    return f;
}
__device__ __forceinline__ void xor_noise_volume(float c0, float c1, float c2, float off1, float &val, float &alpha) {
    val = xor_fbm((c0 + 1.0f) * 32.0f, (c1 + off1) * 32.0f, (c2 + 21.0f) * 32.0f);
    const float len = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
    alpha = val * smoothstepf(0.5f, 0.25f, len);
}
__global__ __launch_bounds__(256) void xor_generate_kernel(uint2 *__restrict__ density, uint2 *__restrict__ normals,
                                                           uint32_t nx, uint32_t ny, uint32_t nz, float time) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint64_t)nx * ny * nz) return;
    const uint32_t x = (uint32_t)(id % nx);
    const uint64_t rest = id / nx;
    const uint32_t y = (uint32_t)(rest % ny), z = (uint32_t)(rest / ny);
    const float d0 = (float)nx, d1 = (float)ny, d2 = (float)nz;
    const float c0 = ((float)x - d0 / 2.0f) / d0, c1 = ((float)y - d1 / 2.0f) / d1, c2 = ((float)z - d2 / 2.0f) / d2;
    const float off1 = sin_spec(time * 1.0f) * 0.1f;
    float val, alpha, v1, a0, a1, a2;
    xor_noise_volume(c0, c1, c2, off1, val, alpha);
    xor_noise_volume(c0 - 0.0001f, c1, c2, off1, v1, a0);
    xor_noise_volume(c0, c1 - 0.0001f, c2, off1, v1, a1);
    xor_noise_volume(c0, c1, c2 - 0.0001f, off1, v1, a2);
    const float g0 = alpha - a0, g1 = alpha - a1, g2 = alpha - a2;
    const float nl = sqrtf((g0 * g0 + g1 * g1) + g2 * g2);
    const float n0 = g0 / nl, n1 = g1 / nl, n2 = g2 / nl;  // normalize(0) = NaN, as on any GPU
    const float ln = sqrtf((n0 * n0 + n1 * n1) + n2 * n2);
    union { _Float16 h[4]; uint2 u; } dv, nv;
    dv.h[0] = (_Float16)(val / 2.0f); dv.h[1] = dv.h[0]; dv.h[2] = dv.h[0]; dv.h[3] = (_Float16)alpha;
    nv.h[0] = (_Float16)n0; nv.h[1] = (_Float16)n1; nv.h[2] = (_Float16)n2; nv.h[3] = (_Float16)ln;
    density[id] = dv.u;
    normals[id] = nv.u;
}

// ---- PROCEDURAL: the compute twin's ray and march with the texel loads replaced by the xor
// example's density function at the sample position, noise_volume(p / 2) (shaders/xor.wgsl:55-61), colour =
// density.rgb / 2, no normals -- mirrors pixel_procedural of the oracle operation for operation.  No volume,
// no loads: 24 specified sines (f64 Cody-Waite, ~45 f64 operations each) and ~200 f32 flops per step.
template <int OUT, bool COUNT>
__global__ __launch_bounds__(64) void raymarch_procedural_kernel(const LaunchDesc L, float time) {
    const uint32_t lb = logical_block(blockIdx.x);
    if (lb >= L.n_blocks) return;
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, threadIdx.x);
    if (!pm.valid) return;
    float dimx = (float)L.W, dimy = (float)L.H;
    float aspect_ratio = dimy / dimx;
    float scx = 2.0f * (float)pm.x / dimx - 1.0f;
    float scy = 2.0f * (float)pm.y / dimy - 1.0f;
    scy = scy * -aspect_ratio;
    float vp[4], vt[4];
    mat4_mul_vec4(fv.inv_proj, scx, scy, 0.0f, 1.0f, vp);
    mat4_mul_vec4(fv.inv_proj, scx, scy, 1.0f, 1.0f, vt);
    const float eye[3] = {vp[0] / vp[3], vp[1] / vp[3], vp[2] / vp[3]};
    float dir[3] = {vt[0] / vt[3] - eye[0], vt[1] / vt[3] - eye[1], vt[2] / vt[3] - eye[2]};
    normalize3(dir[0], dir[1], dir[2]);
    float C[3] = {0.023f, 0.02f, 0.02f};
    uint32_t n_iter = 0;
    float t0, t1;
    intersect_box(eye, dir, -1.0f, 1.0f, t0, t1);
    if (t0 < t1) {
        t0 = fmaxf(t0, 0.0f);
        float A = 0.1f;
        const float bs = 256.0f;
        float dtx = 1.0f / (bs * fabsf(dir[0])), dty = 1.0f / (bs * fabsf(dir[1])), dtz = 1.0f / (bs * fabsf(dir[2]));
        const float dt = L.dt_scale * fmaxf(fminf(dtx, fminf(dty, dtz)), 0.01f);
        const float off1 = sin_spec(time * 1.0f) * 0.1f;
        for (float t = t0; t < t1; t = t + dt) {
            const float px = eye[0] + t * dir[0], py = eye[1] + t * dir[1], pz = eye[2] + t * dir[2];
            float val, alpha;
            xor_noise_volume(px * 0.5f, py * 0.5f, pz * 0.5f, off1, val, alpha);
            n_iter++;
            const float vc = val / 2.0f;
            float va = (alpha * alpha) * alpha;
            va = smoothstepf(0.0f, 0.7f, va);
            const float w = (1.0f - A) * va;
            C[0] = C[0] + w * vc; C[1] = C[1] + w * vc; C[2] = C[2] + w * vc;
            A = A + w;
            if (A >= 0.95f) break;
            if (!(dt > 0.0f)) break;
        }
    }
    store_out<OUT>(L, pm, C[0], C[1], C[2]);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = n_iter;
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)n_iter);
        }
    }
}

template <int OUT>
__global__ __launch_bounds__(256) void clear_kernel(void *out, uint64_t n_px) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n_px) store_pixel<OUT>(out, id, 0.0f, 0.0f, 0.0f, 1.0f);
}

// ---- present pass (next row N1): shaders/present.wgsl:23-35,111-119 ---------------------------
// One fused pass: bilinear resample of the backbuffer (linear clamp-to-edge sampler,
// src/context/present_pipeline.rs:95-104) -> ACESFilm -> branch-free linear_to_srgb -> RGBA8 (the
// Rgba8Unorm copy capture_frame reads) and, optionally, BGRA8 (the surface format).  HBM-bound:
// 8-16 B read + 4-8 B written per pixel.
__device__ __forceinline__ float4 load_px(const void *bb, int fmt, size_t idx) {
    if (fmt == OUT_RGBA32F) return reinterpret_cast<const float4 *>(bb)[idx];
    uint2 v = reinterpret_cast<const uint2 *>(bb)[idx];
    return make_float4(h2f(v.x & 0xffffu), h2f(v.x >> 16), h2f(v.y & 0xffffu), h2f(v.y >> 16));
}
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
__global__ __launch_bounds__(256) void present_kernel(const void *__restrict__ bb, int fmt, uint32_t bw, uint32_t bh,
                                                      uint32_t w, uint32_t h, uint32_t *__restrict__ rgba8,
                                                      uint32_t *__restrict__ bgra8) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint64_t)w * h) return;
    uint32_t x = (uint32_t)(id % w), y = (uint32_t)(id / w);
    float uvx = ((float)x + 0.5f) / (float)w, uvy = ((float)y + 0.5f) / (float)h;
    float ux = fmaf(uvx, (float)bw, -0.5f), uy = fmaf(uvy, (float)bh, -0.5f);
    int ix = cvt_floor_i32(ux), iy = cvt_floor_i32(uy);
    float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy);
    int x0 = clampi(ix, 0, (int)bw - 1), x1 = clampi(ix + 1, 0, (int)bw - 1);
    int y0 = clampi(iy, 0, (int)bh - 1), y1 = clampi(iy + 1, 0, (int)bh - 1);
    const float4 t00 = load_px(bb, fmt, (size_t)y0 * bw + x0);
    float px[4] = {t00.x, t00.y, t00.z, t00.w};
    if (fx != 0.0f || fy != 0.0f) {
        // (a sample on a texel centre -- nearly every pixel when the backbuffer has the window's size -- has weights
        // exactly (1, 0, 0, 0): fma(0, b - a, a) is a for finite taps, so the other three are not fetched)
        const float4 t10 = load_px(bb, fmt, (size_t)y0 * bw + x1), t01 = load_px(bb, fmt, (size_t)y1 * bw + x0), t11 = load_px(bb, fmt, (size_t)y1 * bw + x1);
        const float a1[4] = {t10.x, t10.y, t10.z, t10.w}, b0[4] = {t01.x, t01.y, t01.z, t01.w}, b1[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float a = fmaf(fx, a1[k] - px[k], px[k]), b = fmaf(fx, b1[k] - b0[k], b0[k]);
            px[k] = fmaf(fy, b - a, a);
        }
    }
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = px[k];
        if (k < 3) v = present_srgb(aces_film(v));
        c[k] = floorf(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f);
    }
    uint32_t r = (uint32_t)c[0], g = (uint32_t)c[1], b = (uint32_t)c[2], al = (uint32_t)c[3];
    rgba8[id] = r | (g << 8) | (b << 16) | (al << 24);
    if (bgra8) bgra8[id] = b | (g << 8) | (r << 16) | (al << 24);
}

// Root side of a batched multi-GPU launch: gathered [nranks][slot][frame][ts][ts] (slot < n_slots) -> frames [B][H][W].
// tile_pos: per frame, the inverse order (tile id -> position) at tile_pos[frame * n_tiles + tile].
// One workgroup moves 512 pixels of one tile, two adjacent pixels (16 / 32 bytes) per lane: the tile's owner and
// source base are wave-uniform, loads and stores are whole 16-byte vectors (the per-pixel form of round 1 reached
// 2 TB/s; this is the copy the root pays for every frame).  frames == nullptr: one frame with n_active_one active tiles
// (vk_untile).
template <int OUT>
__global__ __launch_bounds__(256) void untile_batch_kernel(const void *__restrict__ gathered, void *__restrict__ out, uint32_t W, uint32_t H, uint32_t ts,
                                                           uint32_t tiles_x, uint32_t n_tiles, uint32_t nranks, uint32_t n_slots, uint32_t n_frames,
                                                           const uint32_t *__restrict__ tile_pos, const FrameDesc *__restrict__ frames, uint32_t root_skip,
                                                           uint32_t n_active_one, const uint32_t *__restrict__ prev_tile_pos = nullptr,
                                                           const FrameDesc *__restrict__ prev_frames = nullptr, uint32_t wire_rgb = 0u) {
    const uint32_t chunks = (ts * ts + 511u) / 512u;  // 512-pixel chunks per tile
    uint32_t b = blockIdx.x;
    const uint32_t chunk = b % chunks; b /= chunks;
    const uint32_t tile = b % n_tiles;
    const uint32_t frame = b / n_tiles;
    if (frame >= n_frames) return;
    const uint32_t l = chunk * 512u + threadIdx.x * 2u;  // first of this lane's two pixels inside the tile (ts is even)
    if (l >= ts * ts) return;
    const uint32_t ly = l / ts, lx = l - ly * ts;
    const uint32_t tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const uint32_t x = txi * ts + lx, y = tyi * ts + ly;
    if (x >= W || y >= H) return;
    const bool two = x + 1u < W;  // W odd: the last pixel of a row stands alone
    const size_t dst = ((size_t)frame * H + y) * W + x;
    const uint32_t pos = tile_pos[(size_t)frame * n_tiles + tile];
    const uint32_t n_active = frames ? frames[frame].n_active : n_active_one;
    if (pos >= n_active) {  // the box's silhouette cannot reach this tile: never marched, never gathered
        // `out` still holds the un-tiled frames of an earlier batch (prev_*): a tile that was inactive then as well is clear already
        if (prev_tile_pos && prev_tile_pos[(size_t)frame * n_tiles + tile] >= prev_frames[frame].n_active) return;
        store_pixel<OUT>(out, dst, 0.0f, 0.0f, 0.0f, 1.0f);
        if (two) store_pixel<OUT>(out, dst + 1, 0.0f, 0.0f, 0.0f, 1.0f);
        return;
    }
    uint32_t rank, slot;
    deal_owner(pos, nranks, root_skip, rank, slot);
    const size_t src = ((((size_t)rank * n_slots + slot) * n_frames + frame) * ts + ly) * ts + lx;
    if (wire_rgb) {  // records of ts*ts (r, g) pairs + ts*ts b values (store_out): alpha is 1
        const size_t tt = (size_t)ts * ts, rec = ((size_t)rank * n_slots + slot) * n_frames + frame;
        if (OUT == OUT_RGBA32F) {
            const float *g = reinterpret_cast<const float *>(gathered) + rec * tt * 3u;
            const float4 rg = *reinterpret_cast<const float4 *>(g + 2u * l);  // (l even: 16-byte aligned)
            const float2 bb = *reinterpret_cast<const float2 *>(g + 2u * tt + l);
            float4 *o = reinterpret_cast<float4 *>(out);
            o[dst] = make_float4(rg.x, rg.y, bb.x, 1.0f);
            if (two) o[dst + 1] = make_float4(rg.z, rg.w, bb.y, 1.0f);
        } else {
            const uint16_t *g = reinterpret_cast<const uint16_t *>(gathered) + rec * tt * 3u;
            const uint2 rg = *reinterpret_cast<const uint2 *>(g + 2u * l);
            const uint32_t bb = *reinterpret_cast<const uint32_t *>(g + 2u * tt + l);
            uint2 *o = reinterpret_cast<uint2 *>(out);
            const uint2 p0 = make_uint2(rg.x, (bb & 0xffffu) | 0x3c000000u), p1 = make_uint2(rg.y, (bb >> 16) | 0x3c000000u);  // 0x3c00: 1.0
            if (two && ((dst & 1u) == 0u)) *reinterpret_cast<uint4 *>(o + dst) = make_uint4(p0.x, p0.y, p1.x, p1.y);
            else { o[dst] = p0; if (two) o[dst + 1] = p1; }
        }
        return;
    }
    if (OUT == OUT_RGBA32F) {
        const float4 *g = reinterpret_cast<const float4 *>(gathered);
        float4 *o = reinterpret_cast<float4 *>(out);
        o[dst] = g[src];
        if (two) o[dst + 1] = g[src + 1];
    } else {
        const uint2 *g = reinterpret_cast<const uint2 *>(gathered);
        uint2 *o = reinterpret_cast<uint2 *>(out);
        if (two && ((dst & 1u) == 0u)) {  // src is even (ts, lx even): one 16-byte load, one 16-byte store
            *reinterpret_cast<uint4 *>(o + dst) = *reinterpret_cast<const uint4 *>(g + src);
        } else {
            o[dst] = g[src];
            if (two) o[dst + 1] = g[src + 1];
        }
    }
}

}  // namespace vk
