// vk_launch_cells.hip -- instantiates raymarch_naive_kernel (vk_march.hpp) for every cell / comparison layout.
#include "vk_ctx.hpp"
#include "vk_march.hpp"

using namespace vk;

template <int VOL, bool SKIP, bool SAFE, int WALK = WALK_LOOP, bool AHEAD = false>
static void launch_naive(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V_in, uint32_t grid, bool count) {
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    VolumeDesc V = V_in;
    if (!SKIP && V.lut) V.lut += cell_lut_entries(V.nx, V.ny, V.nz);  // byte-offset copy of the tables
    // the fast path of the cell layouts keeps its per-axis index tables in LDS (vk_march.hpp: load_cell_luts)
    constexpr bool lut = (VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16) && !SAFE;
    const uint32_t lds = (lut ? cell_lut_bytes(V.nx, V.ny, V.nz) : 0u) + ctx->naive_lds_pad;  // (pad: occupancy experiments, vk_debug_set_param)
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, WALK, AHEAD, OUT_RGBA16F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, WALK, AHEAD, OUT_RGBA16F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, WALK, AHEAD, OUT_RGBA32F, true>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
        else hipLaunchKernelGGL((raymarch_naive_kernel<VOL, SKIP, SAFE, WALK, AHEAD, OUT_RGBA32F, false>), dim3(grid), dim3(64), lds, ctx->stream, L, V);
    }
}

template <int VOL>
static void launch_packed(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, bool skip, bool safe, int walk) {
    // the probe-ahead trip (LF_PROBE_AHEAD: single-frame launches) exists for the skip kernels' fast path, unbounded march only
    const bool ahead = skip && !safe && (L.flags & LF_PROBE_AHEAD) && !(L.flags & LF_ADAPTIVE_PROBING);
    if (skip && walk == WALK_FMA) {
        if (safe) launch_naive<VOL, true, true, WALK_FMA>(ctx, L, V, grid, count);
        else if (ahead) launch_naive<VOL, true, false, WALK_FMA, true>(ctx, L, V, grid, count);
        else launch_naive<VOL, true, false, WALK_FMA>(ctx, L, V, grid, count);
    } else if (skip) {
        if (safe) launch_naive<VOL, true, true>(ctx, L, V, grid, count);
        else if (ahead) launch_naive<VOL, true, false, WALK_LOOP, true>(ctx, L, V, grid, count);
        else launch_naive<VOL, true, false>(ctx, L, V, grid, count);
    } else { if (safe) launch_naive<VOL, false, true>(ctx, L, V, grid, count); else launch_naive<VOL, false, false>(ctx, L, V, grid, count); }
}

void launch_cells(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, bool skip, bool safe, int walk) {
    switch (ctx->vol_kind) {
        case VOL_P8: launch_packed<VOL_P8>(ctx, L, V, grid, count, skip, safe, walk); break;
        case VOL_P16: launch_packed<VOL_P16>(ctx, L, V, grid, count, skip, safe, walk); break;
        case VOL_PF16: launch_packed<VOL_PF16>(ctx, L, V, grid, count, skip, safe, walk); break;
        case VOL_B9U8: launch_naive<VOL_B9U8, false, true>(ctx, L, V, grid, count); break;
        case VOL_B9F16: launch_naive<VOL_B9F16, false, true>(ctx, L, V, grid, count); break;
        case VOL_Q8: launch_naive<VOL_Q8, false, true>(ctx, L, V, grid, count); break;
        case VOL_QF16: launch_naive<VOL_QF16, false, true>(ctx, L, V, grid, count); break;
        case VOL_LINEAR_F16: launch_naive<VOL_LINEAR_F16, false, true>(ctx, L, V, grid, count); break;
        default: launch_naive<VOL_LINEAR_U8, false, true>(ctx, L, V, grid, count); break;
    }
}
