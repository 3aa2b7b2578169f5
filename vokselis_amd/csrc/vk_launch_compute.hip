// vk_launch_compute.hip -- instantiates the compute twin (raycast_compute.wgsl) and the C3 procedural march (vk_compute.hpp).
#include "vk_ctx.hpp"
#include "vk_compute.hpp"

using namespace vk;

void launch_procedural(vk_ctx *ctx, const LaunchDesc &L, uint32_t grid, bool count, float time) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA16F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, true>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
        else hipLaunchKernelGGL((raymarch_procedural_kernel<OUT_RGBA32F, false>), dim3(grid), dim3(64), 0, ctx->stream, L, time);
    }
}

template <bool SKIP, int RING>
static void launch_records(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    const uint32_t lds = pair_lut_entries(V.nx, V.ny, V.nz) * 4u;
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, true, SKIP, RING>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA16F, false, SKIP, RING>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, true, SKIP, RING>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_compute_records_kernel<OUT_RGBA32F, false, SKIP, RING>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    }
}

void launch_compute(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, bool records, bool skip) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    if (records) {
        const uint32_t ring = ctx->pair_ring;
        if (!skip) launch_records<false, 4>(ctx, L, V, grid, count);
        else if (ring >= 8) launch_records<true, 8>(ctx, L, V, grid, count);
        else if (ring >= 6) launch_records<true, 6>(ctx, L, V, grid, count);
        else launch_records<true, 4>(ctx, L, V, grid, count);
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
