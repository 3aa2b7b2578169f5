// vk_launch_staged.hip -- instantiates the LDS-staged march (vk_staged.hpp) and picks its LDS budget from the view.
#include "vk_ctx.hpp"
#include "vk_staged.hpp"

#include <algorithm>
#include <cmath>

using namespace vk;

// LDS window per wave of the staged march: more LDS = thicker slabs (fewer rounds, each with its slab search, bounds
// and fill) but fewer waves per CU.  The best budget depends on the view: what counts is the box an 8 x 8 pixel wave
// sweeps per slab -- its footprint in cells (distance x pixel angle x n) plus the lateral drift of oblique rays -- and
// the measured optimum (docs/archive/tools/staged_cameras.py: six cameras, C4 and C5) follows the bytes of that box for a slab of
// T* = 4 cells (u8: at the VALU issue limit, occupancy first) or 6 cells (f16), in whole waves per CU between 5 KiB (8 waves
// per SIMD) and 20 KiB (2): C5 from far away 5.83 -> 3.5 ms, from close by 38.4 -> 34.1 ms against a fixed 8 KiB.
static uint32_t staged_cap_auto(const vk_ctx *ctx, const float *cam, bool u8, uint32_t *slab_cells) {
    const uint32_t fallback = u8 ? 6144u : 10240u;
    // the slab search's upper limit: the budget decides the thickness, this only bounds the search (6 / 8 cells until round 3 cut slabs short that
    // would have fitted: C4 1.855 -> 1.78 ms at 16-24, C5 10.0 -> 9.76 at 16; docs/archive/tools/staged_group.py, profiles/r03_staged_group.txt)
    *slab_cells = u8 ? 16u : 24u;
    if (!cam) return fallback;
    const float *m = cam + 20;
    const double W = ctx->width, H = ctx->height;
    auto dir_of = [&](double px, double py, double d[3]) {
        const double X = 2.0 * px / W - 1.0, Y = 1.0 - 2.0 * py / H;
        const double qw = m[3] * X + m[7] * Y + m[11] + m[15];
        double len = 0.0;
        for (int k = 0; k < 3; k++) { d[k] = (m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) / qw - cam[k]; len += d[k] * d[k]; }
        len = std::sqrt(len);
        for (int k = 0; k < 3; k++) d[k] /= len;
        return len;
    };
    double a[3], b[3];
    if (!(dir_of(0.5 * W, 0.5 * H, a) > 0.0) || !(dir_of(0.5 * W + 8.0, 0.5 * H + 8.0, b) > 0.0)) return fallback;
    const double cosang = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    if (!std::isfinite(cosang)) return fallback;
    const double ang = std::acos(std::min(1.0, std::max(-1.0, cosang))) / std::sqrt(2.0);  // 8 pixels along one image axis
    const double to_c[3] = {0.5 - cam[0], 0.5 - cam[1], 0.5 - cam[2]};
    const double dist = std::sqrt(to_c[0] * to_c[0] + to_c[1] * to_c[1] + to_c[2] * to_c[2]);
    const double n[3] = {(double)ctx->nx, (double)ctx->ny, (double)ctx->nz};
    // cells per step along each axis for the central ray; the largest is the slab axis S, then F = S+1, M = S+2 (vk_staged.hpp)
    const double c[3] = {std::fabs(a[0]) * n[0], std::fabs(a[1]) * n[1], std::fabs(a[2]) * n[2]};
    const int S = (c[0] >= c[1] && c[0] >= c[2]) ? 0 : (c[1] >= c[2] ? 1 : 2), F = (S + 1) % 3, M = (S + 2) % 3;
    if (!(c[S] > 0.0) || !std::isfinite(dist)) return fallback;
    const double Tstar = u8 ? 4.0 : 6.0;
    const double rows = std::ceil(ang * dist * n[M] + c[M] / c[S] * Tstar + 3.0), cols = std::ceil(ang * dist * n[F] + c[F] / c[S] * Tstar + 3.0);
    const double per_piece = u8 ? 16.0 : 8.0;
    const double bytes = (Tstar + 1.0) * rows * std::ceil((cols + per_piece - 1.0) / per_piece) * 16.0;
    if (!std::isfinite(bytes)) return fallback;
    // a budget buys whole waves per CU (160 KiB of LDS, 4 SIMDs): the smallest of 32, 28, ... 8 waves' shares that holds the box
    for (uint32_t waves = 32u; waves > 8u; waves -= 4u) {
        const uint32_t cap = (163840u / waves) & ~15u;
        if ((double)cap >= 0.95 * bytes) return cap;
    }
    return 20480u;
}

template <int VOL>
static void launch_staged_t(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, const float *cam) {
    StagedDesc D = ctx->sdesc;
    uint32_t slab_auto = 8u;
    const uint32_t cap_auto = staged_cap_auto(ctx, cam, VOL == VOL_S8U8, &slab_auto);
    D.cap_bytes = std::min(std::max((ctx->stage_cap_bytes ? ctx->stage_cap_bytes : cap_auto) & ~15u, 1024u), 65536u);
    D.slab_cells = std::min(std::max(ctx->stage_slab_cells ? ctx->stage_slab_cells : slab_auto, 1u), 32u);
    D.row_pad = ctx->stage_row_pad == 1u ? 1u : 0u;
    // u8 (at the issue-slot limit): every 4th round -- C5 10.40 -> 10.12 ms, other views +-1 %; f16 (waiting on fills, not on slots): every round
    D.grow_every = ctx->stage_grow_every ? ctx->stage_grow_every : (VOL == VOL_S8U8 ? 4u : 1u);
    const bool f16 = ctx->out_format == VK_OUT_RGBA16F;
    // One window for the four waves of a 256-thread group (2 x 2 neighbouring 8x8 blocks) instead of one per wave: the rays of 16 x 16 pixels
    // sweep far less than four 8 x 8 boxes, so the slab is ~twice as thick and a ray meets half as many rounds -- against two barriers
    // per round.  Measured (docs/archive/tools/staged_group.py, staged_group_sweep.py; frames bitwise equal): it pays where the wave is starved of LDS --
    // u8 volumes seen from close by (C5: 10.05 -> 9.79 ms; four orbit frames per launch 9.25 -> 9.17; 6.25 KiB per wave and slabs up to 24 cells
    // read 9.83 single but 9.30 in the 4-frame launch: the per-wave budget stays) -- and costs where the budget is large
    // already (far views: a workgroup's 64 KiB is less than four waves' 80) or the launch is one partial round of waves (C4: single frame
    // +2.5 %, four orbit frames per launch 1.67 -> 1.58 ms: f16 takes it in launches of four frames or more).  stage_group: 0 never, 1 always,
    // 2 (default) by these rules.
    const bool group = (L.ts & 15u) == 0u && (ctx->stage_group == 1u || (ctx->stage_group == 2u && !ctx->stage_cap_bytes &&
                                                                                 (VOL == VOL_S8U8 ? cap_auto <= 6400u : (L.frames != nullptr && L.n_frames >= 4u && cap_auto <= 16384u))));
    if (group) {
        // An odd row pitch (in 16-byte pieces) spreads the window rows of neighbouring pixel rows over the LDS banks.  Per wave it cost C5 4 % (a
        // padded piece per row out of 5 KiB); in a group's window the pad is a smaller share and four waves' taps collide more: C5 9.93 -> 9.53 ms,
        // the diagonal view -3.5 %, four orbit frames per launch 9.19 -> 8.91, axis-aligned +2 %, close-up +-0 (profiles/r03_staged_group.txt).
        if (ctx->stage_row_pad == 2u && VOL == VOL_S8U8) D.row_pad = 1u;
        if (!ctx->stage_slab_cells) D.slab_cells = std::max(D.slab_cells, 24u);  // four waves' LDS hold a slab about twice as thick
        const uint32_t lds = std::min(D.cap_bytes * kGroupWaves, 65536u) & ~15u;  // four waves' LDS, less the exchange block
        D.cap_bytes = lds - kGroupExchBytes;
        const uint32_t groups = (grid + kGroupWaves - 1u) / kGroupWaves;
        if (f16) {
            if (count) hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA16F, true>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
            else hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA16F, false>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
        } else {
            if (count) hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA32F, true>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
            else hipLaunchKernelGGL((raymarch_staged_group_kernel<VOL, OUT_RGBA32F, false>), dim3(groups), dim3(256), lds, ctx->stream, L, V, D);
        }
        return;
    }
    if (f16) {
        if (count) hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA16F, true>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
        else hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA16F, false>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
    } else {
        if (count) hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA32F, true>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
        else hipLaunchKernelGGL((raymarch_staged_kernel<VOL, OUT_RGBA32F, false>), dim3(grid), dim3(64), D.cap_bytes, ctx->stream, L, V, D);
    }
}

void launch_staged(vk_ctx *ctx, const LaunchDesc &L, const VolumeDesc &V, uint32_t grid, bool count, const float *cam) {
    if (ctx->vol_kind == VOL_S8U8) launch_staged_t<VOL_S8U8>(ctx, L, V, grid, count, cam);
    else launch_staged_t<VOL_S8F16>(ctx, L, V, grid, count, cam);
}
