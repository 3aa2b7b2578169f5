// vk_launch_compute.hip -- instantiates the compute twin (raycast_compute.wgsl) and the C3 procedural march (vk_compute.hpp).
#include "vk_ctx.hpp"
#include "vk_compute.hpp"

using namespace vk;

void launch_procedural(vk_ctx *ctx, const LaunchDesc &L, uint32_t grid, bool count, float time, bool device_sine) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    if (device_sine) {  // (tolerance mode: no counting instantiation of its own -- the counters describe the specified march)
        if (f16) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, false, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, false, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        return;
    }
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
    }
}

template <bool SKIP, int RING, int REV = 1>
static void launch_records(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    const uint32_t lds = pair_lut_entries(V.nx, V.ny, V.nz) * 4u;
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, true, SKIP, RING, REV>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, false, SKIP, RING, REV>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, true, SKIP, RING, REV>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, false, SKIP, RING, REV>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    }
}

void launch_compute(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, bool records, bool skip) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    if (records) {
        // request buffers x revolutions of the ring per loop iteration (vk_compute.hpp; profiles/r04_compute_twin_skip_and_ring.txt): a launch of
        // one frame that does not fill the machine (720p: 14 400 waves) is its longest waves' chains and gains from the deeper ring; launches
        // that do (1080p: 32 400; several frames) from the longer loop body
        // ... and a frame of a crowded ring of frames in flight (three executing) from the smallest ring: the xor 720p frame 0.0500 -> 0.0456 ms
        // per frame at k = 4 with four buffers and one revolution (6: 0.0500, 4 x 2: 0.0502; on one stream 4 costs 1.5 %)
        const uint32_t ring = ctx->pair_ring ? ctx->pair_ring : (frames_crowded(ctx) ? 4u : ((L.n_frames > 1 || L.n_blocks >= 24000u) ? 42u : 6u));
        if (!skip) launch_records<false, 4, 1>(ctx, L, V, grid, count);
        else if (ring == 42) launch_records<true, 4, 2>(ctx, L, V, grid, count);
        else if (ring == 6) launch_records<true, 6, 1>(ctx, L, V, grid, count);
        else launch_records<true, 4, 1>(ctx, L, V, grid, count);
        return;
    }
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, V);
    }
}
