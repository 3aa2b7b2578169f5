// vk_post.hip -- after the march: backbuffer clear, un-tile of gathered tiles on the root, present pass (shaders/present.wgsl)
// and capture_frame's byte layout (src/utils/mod.rs:91-118).
#include "vk_ctx.hpp"
#include "vk_post.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

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
    if (ctx->order.size() != n_tiles) {
        const int m = ctx->format == VK_FMT_RGBA16F_PAIR || ctx->format < 0 ? VK_MODE_COMPUTE_NEAREST : VK_MODE_NAIVE_TRILINEAR;
        int orc = tile_order_update(ctx, m, 0, 0, ctx->width, ctx->height, tile_size, true);
        if (orc) return orc;
    }
    { int orc = order_ensure_device(ctx); if (orc) return orc; }  // (the tables as the partition left them)
    int owc = order_wait(ctx);
    if (owc) return owc;
    const uint32_t *d_pos = ctx->d_order_pos;
    const uint32_t n_active = ctx->order_active;
    const uint32_t n_slots = slot_stride;  // slots per rank in `gathered`
    const uint64_t blocks = untile_blocks(tile_size, (uint32_t)n_tiles, 1u);
    if (blocks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "vk_untile: too many tiles for one launch");
    const uint32_t rs = nranks > 1 ? ctx->root_skip : 0u;
    if (ctx->out_format == VK_OUT_RGBA16F)
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA16F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, ctx->backbuffer, ctx->width, ctx->height, tile_size, tiles_x, (uint32_t)n_tiles, nranks, n_slots, 1u, d_pos, (const FrameDesc *)nullptr, rs, n_active, (const uint32_t *)nullptr, (const FrameDesc *)nullptr, (uint32_t)ctx->wire);
    else
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA32F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, ctx->backbuffer, ctx->width, ctx->height, tile_size, tiles_x, (uint32_t)n_tiles, nranks, n_slots, 1u, d_pos, (const FrameDesc *)nullptr, rs, n_active, (const uint32_t *)nullptr, (const FrameDesc *)nullptr, (uint32_t)ctx->wire);
    HIP_TRY(ctx, hipGetLastError());
    return VK_OK;
}

// The current surface's Rgba8 (and Bgra8) present targets at width x height (the rgb_texture of src/context.rs:59-60,246).
int present_targets(vk_ctx *ctx, uint32_t width, uint32_t height, bool also_bgra) {
    if (width == ctx->present_w && height == ctx->present_h && ctx->rgba8 && (!also_bgra || ctx->bgra8)) return VK_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->rgba8) (void)hipFree(ctx->rgba8);
    if (ctx->bgra8) (void)hipFree(ctx->bgra8);
    ctx->rgba8 = ctx->bgra8 = nullptr;
    ctx->present_w = ctx->present_h = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->rgba8, (size_t)width * height * 4));
    if (also_bgra) HIP_TRY(ctx, hipMalloc(&ctx->bgra8, (size_t)width * height * 4));
    // (a tile-by-tile fused present fills it piecewise: what no tile covers reads as transparent black, never as stale memory)
    HIP_TRY(ctx, hipMemsetAsync(ctx->rgba8, 0, (size_t)width * height * 4, ctx->stream));
    if (also_bgra) HIP_TRY(ctx, hipMemsetAsync(ctx->bgra8, 0, (size_t)width * height * 4, ctx->stream));
    ctx->present_w = width; ctx->present_h = height;
    return VK_OK;
}

extern "C" {

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

int vk_untile(vk_ctx *ctx, const void *gathered, uint32_t tile_size, uint32_t nranks, uint32_t slot_stride) {
    return untile_common(ctx, gathered, tile_size, nranks, slot_stride);
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
    const uint64_t blocks = untile_blocks(B->ts, B->n_tiles, B->n_frames);
    if (blocks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "vk_untile_batch: too many tiles for one launch");
    const FrameDesc *frames = reinterpret_cast<const FrameDesc *>(B->d);
    const uint32_t *pos = reinterpret_cast<const uint32_t *>(B->d + batch_order_offset(B->n_frames, sizeof(FrameDesc))) + (size_t)B->n_frames * B->n_tiles;
    // FrameDesc::n_active of a compact batch is the frame's active tile count (what the gather carried)
    const FrameDesc *pframes = P ? reinterpret_cast<const FrameDesc *>(P->d) : nullptr;
    const uint32_t *ppos = P ? reinterpret_cast<const uint32_t *>(P->d + batch_order_offset(P->n_frames, sizeof(FrameDesc))) + (size_t)P->n_frames * P->n_tiles : nullptr;
    if (ctx->out_format == VK_OUT_RGBA16F)
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA16F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, out_frames, ctx->width, ctx->height, B->ts, tx, B->n_tiles, B->nranks, n_slots, B->n_frames, pos, frames, B->root_skip, 0u, ppos, pframes, (uint32_t)B->wire);
    else
        hipLaunchKernelGGL(untile_batch_kernel<OUT_RGBA32F>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, gathered, out_frames, ctx->width, ctx->height, B->ts, tx, B->n_tiles, B->nranks, n_slots, B->n_frames, pos, frames, B->root_skip, 0u, ppos, pframes, (uint32_t)B->wire);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(B->ev, ctx->stream));
    if (P) HIP_TRY(ctx, hipEventRecord(P->ev, ctx->stream));  // (its tables were read too: the slot is reused after this launch)
    return VK_OK;
}

// ---- present + capture (next rows N1, N2) ------------------------------------------------------

int vk_present(vk_ctx *ctx, uint32_t width, uint32_t height, int also_bgra) {
    if (!ctx) return VK_ERR_INVALID;
    if (!ctx->backbuffer) return fail(ctx, VK_ERR_INVALID, "vk_present: no backbuffer");
    if (width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, VK_ERR_INVALID, "vk_present: size must be in [1, 32768]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int prc = present_targets(ctx, width, height, also_bgra != 0);
    if (prc) return prc;
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
    if (padded != unpadded)  // the row padding reads as zeros (a 1920-wide frame has none: 7680 = 30 x 256)
        for (uint32_t y = 0; y < h; y++) std::memset(static_cast<unsigned char *>(dst) + (size_t)y * padded + unpadded, 0, padded - unpadded);
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, padded, ctx->rgba8, (size_t)ctx->present_w * 4, unpadded, h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

}  // extern "C"
