// vk_post.hpp -- what happens to a frame after the march: clear (LoadOp::Clear), the root's un-tile of gathered tiles
// (examples/xor/main.rs:77-95 offsets, generalised), the present pass (shaders/present.wgsl).  Included by vk_post.hip only.
#pragma once

#include "vk_common.hpp"

namespace vk {

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
    uint32_t rgba, bgra;
    present_pack(px, rgba, bgra);  // (vk_common.hpp: shared with the fused epilogue)
    rgba8[id] = rgba;
    if (bgra8) bgra8[id] = bgra;
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
    const UntileItem it = untile_item(blockIdx.x, threadIdx.x, ts, n_tiles, n_frames);  // (vk_hostmath.hpp: the map tests/hostmath_fuzz.cpp replays)
    if (!it.in_range) return;
    const uint32_t tile = it.tile, frame = it.frame, l = it.l;
    const uint32_t ly = l / ts, lx = l - ly * ts;
    const uint32_t tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const uint32_t x = txi * ts + lx, y = tyi * ts + ly;
    if (x >= W || y >= H) return;
    const bool two = x + 1u < W;  // W odd: the last pixel of a row stands alone
    const size_t dst = frame_pixel_index(frame, W, H, x, y);
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
    const size_t src = gathered_pixel_index(rank, n_slots, slot, n_frames, frame, ts, lx, ly);
    if (wire_rgb) {  // records of ts*ts (r, g) pairs + ts*ts b values (store_out): alpha is 1
        const size_t tt = (size_t)ts * ts, rec = gathered_record(rank, n_slots, slot, n_frames, frame);
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
