// vk_volume.hip -- VolumeTexture::new (src/context/volume_texture.rs:32-59) behind the C-ABI: the dense x-fastest volume is
// re-laid out on the device (cells + skip maps, 8^3 bricks for LDS staging, records for the compute twin ...).  A build commits
// to the context only when every allocation and kernel has succeeded.
#include "vk_ctx.hpp"
#include "vk_volume_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vk;

void free_volume(vk_ctx *ctx) {
    if (ctx->vol) (void)hipFree(ctx->vol);
    if (ctx->vol2) (void)hipFree(ctx->vol2);
    if (ctx->dist) (void)hipFree(ctx->dist);
    if (ctx->lut) (void)hipFree(ctx->lut);
    for (void *&c : ctx->scopy) { if (c) (void)hipFree(c); c = nullptr; }
    ctx->vol = ctx->vol2 = nullptr;
    ctx->dist = nullptr;
    ctx->lut = nullptr;
    ctx->vol_bytes = 0;
    ctx->format = -1;
    ctx->vol_kind = -1;
    ctx->empty_fraction = 0.0;
}

// ---- volume ------------------------------------------------------------------------------------

// Everything a volume owns on the device.  build_from_dense() fills a local one and the context adopts it only
// when every allocation and kernel has succeeded, so a failed upload leaves the previous volume (or "no volume")
// intact and never a half-built one for the next vk_render to dereference.
struct VolBuild {
    void *vol = nullptr, *vol2 = nullptr;
    uint8_t *dist = nullptr;
    uint32_t *lut = nullptr;
    void *scopy[3] = {nullptr, nullptr, nullptr};
    size_t vol_bytes = 0;
    uint32_t nx = 0, ny = 0, nz = 0, nbx = 0, nby = 0, nbz = 0;
    int format = -1, layout = 0, vol_kind = -1;
    double empty_fraction = 0.0;
    VolumeDesc vdesc{};
    StagedDesc sdesc{};
    bool committed = false;
    ~VolBuild() {
        if (committed) return;
        (void)hipFree(vol); (void)hipFree(vol2); (void)hipFree(dist); (void)hipFree(lut);
        for (void *c : scopy) (void)hipFree(c);
    }
};

// The adopted dense source (vk_volume_upload / vk_volume_generate): freed on every exit unless a layout keeps it.
struct DenseSource {
    const void *p = nullptr, *p2 = nullptr;
    bool owned = false;
    ~DenseSource() { if (owned) { (void)hipFree(const_cast<void *>(p)); (void)hipFree(const_cast<void *>(p2)); } }
    void release() { owned = false; }
};

static int commit_volume(vk_ctx *ctx, VolBuild &nb) {
    // frames in flight on other slots' streams still read the previous volume, and the next ones must find this one built
    // (the build ran on the current slot's stream): drain every slot.  One slot: stream order does it, as before.
    if (ctx->fif_k > 1) { int drc = frames_drain(ctx); if (drc) return drc; }
    free_volume(ctx);
    for (auto &b : ctx->batch) b.id = 0;  // batches dealt for the previous volume are no longer un-tiled
    ctx->batch_key.clear();
    ctx->vol = nb.vol; ctx->vol2 = nb.vol2; ctx->dist = nb.dist; ctx->lut = nb.lut;
    for (int k = 0; k < 3; k++) ctx->scopy[k] = nb.scopy[k];
    ctx->vol_bytes = nb.vol_bytes;
    ctx->nx = nb.nx; ctx->ny = nb.ny; ctx->nz = nb.nz; ctx->nbx = nb.nbx; ctx->nby = nb.nby; ctx->nbz = nb.nbz;
    ctx->format = nb.format; ctx->layout = nb.layout; ctx->vol_kind = nb.vol_kind;
    ctx->empty_fraction = nb.empty_fraction;
    ctx->vdesc = nb.vdesc;
    ctx->sdesc = nb.sdesc;
    nb.committed = true;
    return VK_OK;
}

static int build_from_dense(vk_ctx *ctx, const void *d_src, const void *d_src2, bool own_src, uint32_t nx,
                            uint32_t ny, uint32_t nz, int format, int layout) {
    // d_src is dense device memory; with own_src the LINEAR layout adopts it, every other layout frees it.
    DenseSource src;
    src.p = d_src; src.p2 = d_src2; src.owned = own_src;
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : (format == VK_FMT_R16_FLOAT ? 2 : 8);
    if (layout == VK_LAYOUT_AUTO) {
        const double cells = ((double)((nx - 1) / 4 + 2)) * ((ny - 1) / 4 + 2) * ((nz - 1) / 4 + 2) * 64.0;
        const double cell_bytes = cells * 16.0;
        if (format == VK_FMT_RGBA16F_PAIR) {
            // 16-byte (density, normals) records in 4^3 bricks while the array and its index tables stay small
            const double rec_bytes = ((double)((nx + 3) / 4)) * ((ny + 3) / 4) * ((nz + 3) / 4) * 64.0 * 16.0;
            layout = (rec_bytes <= (double)kPairOob && pair_lut_entries(nx, ny, nz) * 4u <= 16384u) ? VK_LAYOUT_PACKED : VK_LAYOUT_LINEAR;
        }
        // beyond ~4 GiB of cells the march stops being cache-resident and the 8-16x inflation of the cell layouts
        // turns into HBM traffic: dense 8^3 bricks staged through LDS win there
        else if (cell_bytes > 4.0 * 1024 * 1024 * 1024) layout = VK_LAYOUT_STAGED;
        // u8: the (tap, delta) pair cells cost 2x the bytes and ~20 % fewer VALU ops per sample
        else layout = format == VK_FMT_R8_UNORM ? VK_LAYOUT_PACKED_PAIRS : VK_LAYOUT_PACKED;
    }
    if (format == VK_FMT_RGBA16F_PAIR && layout != VK_LAYOUT_LINEAR && layout != VK_LAYOUT_PACKED)
        return fail(ctx, VK_ERR_UNSUPPORTED, "RGBA16F_PAIR volumes use VK_LAYOUT_LINEAR or VK_LAYOUT_PACKED (bricked 16-byte records)");
    if (format == VK_FMT_RGBA16F_PAIR && layout == VK_LAYOUT_PACKED) {
        const double rec_bytes = ((double)((nx + 3) / 4)) * ((ny + 3) / 4) * ((nz + 3) / 4) * 64.0 * 16.0;
        if (rec_bytes > (double)kPairOob || pair_lut_entries(nx, ny, nz) * 4u > 16384u)
            return fail(ctx, VK_ERR_UNSUPPORTED, "RGBA16F_PAIR record layout holds <= 1.25 GiB of records: use VK_LAYOUT_LINEAR");
    }
    VolBuild nb;
    nb.nx = nx; nb.ny = ny; nb.nz = nz;
    nb.format = format;
    nb.layout = layout;
    // kernels launched below, then one synchronisation; the message names the stage that failed
    auto finish = [&](const char *what) -> int {
        hipError_t le = hipGetLastError();
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (le != hipSuccess || se != hipSuccess) return fail(ctx, VK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(le != hipSuccess ? le : se));
        return VK_OK;
    };
    auto alloc = [&](void **p, size_t bytes, const char *what) -> int {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) { *p = nullptr; return fail(ctx, e == hipErrorOutOfMemory ? VK_ERR_OOM : VK_ERR_HIP, std::string(what) + " allocation failed: " + hipGetErrorString(e)); }
        return VK_OK;
    };
    int rc;
    if (layout == VK_LAYOUT_LINEAR) {
        if (own_src) {
            nb.vol = const_cast<void *>(d_src);
            nb.vol2 = const_cast<void *>(d_src2);
            src.release();
        } else {
            if ((rc = alloc(&nb.vol, n_vox * bpv, "dense volume"))) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(nb.vol, d_src, n_vox * bpv, hipMemcpyDeviceToDevice, ctx->stream));
            if (d_src2) {
                if ((rc = alloc(&nb.vol2, n_vox * bpv, "dense volume (normals)"))) return rc;
                HIP_TRY(ctx, hipMemcpyAsync(nb.vol2, d_src2, n_vox * bpv, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        nb.vol_bytes = n_vox * bpv * (d_src2 ? 2 : 1);
        nb.vol_kind = format == VK_FMT_R16_FLOAT ? VOL_LINEAR_F16 : VOL_LINEAR_U8;
        if ((rc = finish("dense copy"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (format == VK_FMT_RGBA16F_PAIR) {  // layout == VK_LAYOUT_PACKED: interleaved records, 4^3 bricks
        nb.nbx = (nx + 3) / 4; nb.nby = (ny + 3) / 4; nb.nbz = (nz + 3) / 4;
        const uint64_t n_rec = (uint64_t)nb.nbx * nb.nby * nb.nbz * 64u;
        if ((rc = alloc(&nb.vol, n_rec * 16, "record array"))) return rc;
        const uint32_t padded = pair_lut_entries(nx, ny, nz);
        if ((rc = alloc((void **)&nb.lut, (size_t)padded * sizeof(uint32_t), "index table"))) return rc;
        nb.vol_bytes = n_rec * 16;
        nb.vol_kind = VOL_PAIRB;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_rec + 255) / 256, 1ull << 22);
        // scratch of the skip map: occupancy + two pass buffers, one byte per record each
        struct PairScratch {
            uint8_t *occ = nullptr, *tx = nullptr, *txy = nullptr;
            ~PairScratch() { (void)hipFree(occ); (void)hipFree(tx); (void)hipFree(txy); }
        } ps;
        if ((rc = alloc((void **)&ps.occ, n_rec, "re-layout scratch")) || (rc = alloc((void **)&ps.tx, n_rec, "re-layout scratch")) ||
            (rc = alloc((void **)&ps.txy, n_rec, "re-layout scratch")))
            return rc;
        hipLaunchKernelGGL(pack_pairs_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint2 *)d_src, (const uint2 *)d_src2, (uint4 *)nb.vol, nx, ny, nz,
                           nb.nbx, nb.nby, n_rec, ps.occ);
        {   // isotropic Chebyshev distance to the nearest record that can contribute, written into the records' unused eighth half
            const uint64_t pb64 = (n_rec + 255) / 256;
            if (pb64 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for one launch");
            const uint32_t pb = (uint32_t)pb64;
            hipLaunchKernelGGL(dist_pass_kernel, dim3(pb), dim3(256), 0, ctx->stream, ps.occ, ps.tx, nb.nbx, nb.nby, nb.nbz, 0, 0, 0, kPairDistRadius);
            hipLaunchKernelGGL(dist_pass_kernel, dim3(pb), dim3(256), 0, ctx->stream, ps.tx, ps.txy, nb.nbx, nb.nby, nb.nbz, 1, 0, 0, kPairDistRadius);
            hipLaunchKernelGGL(dist_pass_kernel, dim3(pb), dim3(256), 0, ctx->stream, ps.txy, ps.occ, nb.nbx, nb.nby, nb.nbz, 2, 0, 1, kPairDistRadius);
            hipLaunchKernelGGL(embed_pair_dist_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (uint4 *)nb.vol, ps.occ, n_rec);
        }
        hipLaunchKernelGGL(build_pair_luts_kernel, dim3((padded + 255) / 256), dim3(256), 0, ctx->stream, nb.lut, nx, ny, nz, nb.nbx, nb.nby);
        if ((rc = finish("record re-layout"))) return rc;
        nb.vdesc.max_off = (int64_t)(n_rec - 1) * 16;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_STAGED) {
        // dense 8^3 bricks without apron behind a replicated border of kStagePad voxels, one copy per SLOW axis
        // (vk_staged.hpp): copy k = (slow k, fast (k+1)%3, mid (k+2)%3)
        const bool u8 = format == VK_FMT_R8_UNORM;
        const uint32_t n[3] = {nx, ny, nz};
        StagedDesc &D = nb.sdesc;
        for (int a = 0; a < 3; a++) D.nv[a] = 8u * ((n[a] + 7u) / 8u + 2u);
        uint32_t mask = ctx->stage_copies_mask & 7u;
        if (mask == 0) mask = 7u;
        const uint32_t vpp = u8 ? 16u : 8u;
        for (int k = 0; k < 3; k++) {
            if (!(mask & (1u << k))) continue;
            const int F = (k + 1) % 3, M = (k + 2) % 3, S = k;
            D.npf[k] = (D.nv[F] + vpp - 1) / vpp;
            D.nbm[k] = D.nv[M] / 8u;
            const uint64_t n_bricks = (uint64_t)D.npf[k] * D.nbm[k] * (D.nv[S] / 8u);
            if (n_bricks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for 32-bit brick indices");
            const uint64_t n_pieces = n_bricks * 64u;
            if ((rc = alloc(&nb.scopy[k], n_pieces * 16, "staged brick copy"))) return rc;
            D.copy[k] = (const unsigned char *)nb.scopy[k];
            nb.vol_bytes += n_pieces * 16;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_pieces + 255) / 256, 1ull << 22);  // grid-stride kernel
            if (u8) hipLaunchKernelGGL(pack_staged_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, d_src, (uint4 *)nb.scopy[k], nx, ny, nz, k, D.npf[k], D.nbm[k], n_pieces);
            else hipLaunchKernelGGL(pack_staged_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_src, (uint4 *)nb.scopy[k], nx, ny, nz, k, D.npf[k], D.nbm[k], n_pieces);
        }
        // a wave takes the copy whose SLOW axis is its rays' major axis; without it, the copy whose MID axis is
        // (lines are then partly used), else whatever exists
        for (int a = 0; a < 3; a++) {
            const int pref[3] = {a, (a + 1) % 3, (a + 2) % 3};  // slow == a; mid == a; fast == a
            for (int j = 2; j >= 0; j--) if (mask & (1u << pref[j])) D.copy_of_major[a] = (uint32_t)pref[j];
        }
        nb.vol_kind = u8 ? VOL_S8U8 : VOL_S8F16;
        if ((rc = finish("staged brick re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_QUADS) {
        // quad elements in 9x8x8 bricks: padded coordinate c = i + 1 in [0, n] -> (n >> 3) + 1 bricks per axis
        const bool f16q = format == VK_FMT_R16_FLOAT;
        nb.nbx = (nx >> 3) + 1; nb.nby = (ny >> 3) + 1; nb.nbz = (nz >> 3) + 1;
        const uint64_t n_bricksq = (uint64_t)nb.nbx * nb.nby * nb.nbz;
        const uint64_t n_elems = n_bricksq * 576u;
        const size_t ebytes = f16q ? 8 : 4;
        if (n_bricksq >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
        if ((rc = alloc(&nb.vol, n_elems * ebytes + 16, "quad layout (4.5x the dense bytes; VK_LAYOUT_STAGED is 1x per copy)"))) return rc;  // + slack: a sample reads two elements
        nb.vol_bytes = n_elems * ebytes;
        nb.vol_kind = f16q ? VOL_QF16 : VOL_Q8;
        const uint32_t blocksq = (uint32_t)std::min<uint64_t>((n_elems + 255) / 256, 1ull << 22);  // grid-stride kernel
        if (f16q) hipLaunchKernelGGL(pack_quads_kernel<true>, dim3(blocksq), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        else hipLaunchKernelGGL(pack_quads_kernel<false>, dim3(blocksq), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        if ((rc = finish("quad re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    if (layout == VK_LAYOUT_BRICKED) {
        // dense 9^3 bricks: brick b holds voxels [8b-1, 8b+7]; cell coords go up to n -> (n >> 3) + 1 bricks
        const bool f16b = format == VK_FMT_R16_FLOAT;
        nb.nbx = (nx >> 3) + 1; nb.nby = (ny >> 3) + 1; nb.nbz = (nz >> 3) + 1;
        const uint64_t n_bricks9 = (uint64_t)nb.nbx * nb.nby * nb.nbz;
        const uint64_t n_elems = n_bricks9 * 729u;
        if (n_bricks9 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
        if ((rc = alloc(&nb.vol, n_elems * bpv + 16, "9^3 brick array"))) return rc;  // + slack: the last tap pair reads 2 elements
        nb.vol_bytes = n_elems * bpv;
        nb.vol_kind = f16b ? VOL_B9F16 : VOL_B9U8;
        const uint32_t blocks9 = (uint32_t)std::min<uint64_t>((n_elems + 255) / 256, 1ull << 22);  // grid-stride kernel
        if (f16b) hipLaunchKernelGGL(pack_bricks9_kernel<true>, dim3(blocks9), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        else hipLaunchKernelGGL(pack_bricks9_kernel<false>, dim3(blocks9), dim3(256), 0, ctx->stream, d_src, nb.vol, nx, ny, nz, nb.nbx, nb.nby, n_elems);
        if ((rc = finish("brick re-layout"))) return rc;
        return commit_volume(ctx, nb);
    }
    // PACKED: cells for low-corner voxels i in [-1, n-1]; physical brick (i >> 2) + 1
    const bool f16 = format == VK_FMT_R16_FLOAT;
    if (f16 && layout == VK_LAYOUT_PACKED_PAIRS)
        return fail(ctx, VK_ERR_UNSUPPORTED, "PACKED_PAIRS stores exact u8 differences; f16 volumes use PACKED");
    const int kind = f16 ? VOL_PF16 : (layout == VK_LAYOUT_PACKED_PAIRS ? VOL_P16 : VOL_P8);
    nb.nbx = ((nx - 1) >> 2) + 2;
    nb.nby = ((ny - 1) >> 2) + 2;
    nb.nbz = ((nz - 1) >> 2) + 2;
    const uint64_t n_bricks = (uint64_t)nb.nbx * nb.nby * nb.nbz;
    const uint64_t n_cells = n_bricks * kBrickCells;
    const size_t cell_bytes = kind == VOL_P8 ? 8 : 16;
    if (n_bricks >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for 32-bit brick indices");
    if (n_cells >= (1ull << 32)) return fail(ctx, VK_ERR_UNSUPPORTED, "cell layouts hold < 2^32 cells (about 1600^3): use VK_LAYOUT_STAGED or VK_LAYOUT_AUTO");
    const uint64_t pack_blocks64 = (n_cells + 255) / 256;
    if (pack_blocks64 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large for one launch");
    const uint32_t pack_blocks = (uint32_t)pack_blocks64;
    if ((rc = alloc(&nb.vol, n_cells * cell_bytes, "cell array"))) return rc;
    // scratch: occupancy map + two pass buffers
    struct Scratch {
        uint8_t *occ = nullptr, *tx = nullptr, *txy = nullptr;
        ~Scratch() { (void)hipFree(occ); (void)hipFree(tx); (void)hipFree(txy); }
    } sc;
    if ((rc = alloc((void **)&sc.occ, n_cells, "re-layout scratch")) || (rc = alloc((void **)&sc.tx, n_cells, "re-layout scratch")) ||
        (rc = alloc((void **)&sc.txy, n_cells, "re-layout scratch")))
        return rc;
    nb.vol_kind = kind;
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters + 7, 0, sizeof(unsigned long long), ctx->stream));
    if (kind == VOL_PF16)
        hipLaunchKernelGGL(pack_cells_kernel<VOL_PF16>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    else if (kind == VOL_P16)
        hipLaunchKernelGGL(pack_cells_kernel<VOL_P16>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    else
        hipLaunchKernelGGL(pack_cells_kernel<VOL_P8>, dim3(pack_blocks), dim3(256), 0, ctx->stream, d_src, nb.vol, sc.occ, nx, ny, nz, nb.nbx, nb.nby, n_cells, ctx->counters + 7);
    {
        unsigned long long ne = 0;
        hipError_t le = hipGetLastError();
        hipError_t ce = hipMemcpyAsync(&ne, ctx->counters + 7, sizeof(ne), hipMemcpyDeviceToHost, ctx->stream);
        hipError_t se = hipStreamSynchronize(ctx->stream);
        if (le != hipSuccess || ce != hipSuccess || se != hipSuccess)
            return fail(ctx, VK_ERR_HIP, std::string("volume re-layout: ") + hipGetErrorString(le != hipSuccess ? le : (ce != hipSuccess ? ce : se)));
        nb.empty_fraction = (double)ne / (double)n_cells;
    }
    // Distance maps.  Eight one-sided maps (one per ray octant) when skipping will be on by default
    // or may well be forced on (>= 30 % empty cells) and they stay <= 2 GiB; otherwise one isotropic map serves every octant.
    const bool octants = nb.empty_fraction >= 0.30 && n_cells <= (1ull << 28);  // (the default policy skips from 45 %)
    const uint64_t dist_bytes = octants ? 8 * n_cells : n_cells;
    if ((rc = alloc((void **)&nb.dist, dist_bytes, "distance map"))) return rc;
    nb.vol_bytes = n_cells * cell_bytes + dist_bytes;
    nb.vdesc.dist_oct_stride = octants ? (uint32_t)n_cells : 0u;
    auto pass = [&](const uint8_t *in, uint8_t *out, int axis, int dir, int last) {
        hipLaunchKernelGGL(dist_pass_kernel, dim3(pack_blocks), dim3(256), 0, ctx->stream, in, out, nb.nbx, nb.nby, nb.nbz, axis, dir, last);
    };
    if (octants) {
        for (int ux = 0; ux < 2; ux++) {
            pass(sc.occ, sc.tx, 0, ux ? 1 : -1, 0);
            for (int uy = 0; uy < 2; uy++) {
                pass(sc.tx, sc.txy, 1, uy ? 1 : -1, 0);
                for (int uz = 0; uz < 2; uz++) pass(sc.txy, nb.dist + (size_t)(ux | (uy << 1) | (uz << 2)) * n_cells, 2, uz ? 1 : -1, 1);
            }
        }
    } else {
        pass(sc.occ, sc.tx, 0, 0, 0);
        pass(sc.tx, sc.txy, 1, 0, 0);
        pass(sc.txy, nb.dist, 2, 0, 1);
    }
    if ((rc = finish("distance maps"))) return rc;
    {
        // per-axis cell-index tables of the fast path, two copies: cell units, byte offsets
        const uint32_t padded = cell_lut_entries(nx, ny, nz);
        if ((rc = alloc((void **)&nb.lut, (size_t)padded * 2 * sizeof(uint32_t), "cell-index table"))) return rc;
        hipLaunchKernelGGL(build_cell_luts_kernel, dim3((padded + 255) / 256), dim3(256), 0, ctx->stream, nb.lut, nx, ny, nz, nb.nbx, nb.nby,
                           (uint32_t)(cell_bytes == 8 ? 3 : 4));
        if ((rc = finish("cell-index tables"))) return rc;
    }
    // addressing constants (vk_kernels.hpp: VolumeDesc)
    VolumeDesc &V = nb.vdesc;
    const int64_t cb = (int64_t)cell_bytes, bxn = nb.nbx, bxyn = (int64_t)nb.nbx * nb.nby;
    V.sh_x = cell_bytes == 8 ? 3 : 4;
    V.sh_y = V.sh_x + 2;
    V.sh_z = V.sh_x + 4;
    V.kx = (int32_t)(60 * cb);
    V.ky = (int32_t)((64 * bxn - 16) * cb);
    V.kz = (64 * bxyn - 64) * cb;
    V.c0 = (64 * bxyn + 64 * bxn + 64) * cb;
    V.max_off = (int64_t)(n_cells - 1) * cb;
    return commit_volume(ctx, nb);
}

static int check_volume_args(vk_ctx *ctx, const void *p, const void *p2, uint32_t nx, uint32_t ny, uint32_t nz, int format,
                             int layout) {
    if (!ctx) return VK_ERR_INVALID;
    if (!p) return fail(ctx, VK_ERR_INVALID, "volume pointer is NULL");
    if (nx == 0 || ny == 0 || nz == 0 || nx > 8192 || ny > 8192 || nz > 8192)
        return fail(ctx, VK_ERR_INVALID, "volume dims must be in [1, 8192]");
    if (format < VK_FMT_R8_UNORM || format > VK_FMT_RGBA16F_PAIR) return fail(ctx, VK_ERR_INVALID, "unknown volume format");
    if (format == VK_FMT_RGBA16F_PAIR && !p2) return fail(ctx, VK_ERR_INVALID, "RGBA16F_PAIR needs the normals volume");
    if (layout < VK_LAYOUT_AUTO || layout > VK_LAYOUT_STAGED) return fail(ctx, VK_ERR_INVALID, "unknown layout");
    return VK_OK;
}

extern "C" {

int vk_volume_upload(vk_ctx *ctx, const void *host, const void *host2, uint32_t nx, uint32_t ny, uint32_t nz,
                     int format, int layout) {
    int rc = check_volume_args(ctx, host, host2, nx, ny, nz, format, layout);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : (format == VK_FMT_R16_FLOAT ? 2 : 8);
    void *d = nullptr, *d2 = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * bpv));
    hipError_t e = hipMemcpy(d, host, n_vox * bpv, hipMemcpyHostToDevice);
    if (e == hipSuccess && format == VK_FMT_RGBA16F_PAIR) {
        e = hipMalloc(&d2, n_vox * bpv);
        if (e == hipSuccess) e = hipMemcpy(d2, host2, n_vox * bpv, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(d);
        if (d2) (void)hipFree(d2);
        return fail(ctx, VK_ERR_HIP, std::string("volume upload: ") + hipGetErrorString(e));
    }
    return build_from_dense(ctx, d, d2, true, nx, ny, nz, format, layout);
}

int vk_volume_upload_device(vk_ctx *ctx, const void *dev, const void *dev2, uint32_t nx, uint32_t ny, uint32_t nz,
                            int format, int layout) {
    int rc = check_volume_args(ctx, dev, dev2, nx, ny, nz, format, layout);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return build_from_dense(ctx, dev, dev2, false, nx, ny, nz, format, layout);
}

int vk_volume_generate(vk_ctx *ctx, int kind, uint32_t nx, uint32_t ny, uint32_t nz, int format, uint32_t seed,
                       uint32_t lo, uint32_t span, int layout) {
    int dummy = 0;
    int rc = check_volume_args(ctx, &dummy, nullptr, nx, ny, nz, format, layout);
    if (rc) return rc;
    if (kind != VK_GEN_FOG && kind != VK_GEN_BONSAI_STANDIN && kind != VK_GEN_FOG_DENSE_CORE) return fail(ctx, VK_ERR_INVALID, "unknown generator kind");
    const uint32_t core = kind == VK_GEN_FOG_DENSE_CORE ? 1u : 0u;
    if (core) kind = VK_GEN_FOG;
    if (format == VK_FMT_RGBA16F_PAIR) return fail(ctx, VK_ERR_UNSUPPORTED, "generators make scalar volumes");
    if (kind == VK_GEN_BONSAI_STANDIN && format != VK_FMT_R8_UNORM) return fail(ctx, VK_ERR_UNSUPPORTED, "the bonsai stand-in is a u8 volume");
    if (kind == VK_GEN_FOG && format == VK_FMT_R8_UNORM && (span == 0 || lo + span > 256)) return fail(ctx, VK_ERR_INVALID, "fog range outside u8");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    const size_t bpv = format == VK_FMT_R8_UNORM ? 1 : 2;
    void *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * bpv));
    // grid-stride kernels: a launch may not exceed 2^32 threads
    const uint64_t blocks = std::min<uint64_t>((n_vox + 255) / 256, 1ull << 22);
    if (kind == VK_GEN_BONSAI_STANDIN)
        hipLaunchKernelGGL(generate_kernel<2>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    else if (format == VK_FMT_R16_FLOAT)
        hipLaunchKernelGGL(generate_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    else
        hipLaunchKernelGGL(generate_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, d, nx, ny, nz, seed, lo, span, core);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, VK_ERR_HIP, std::string("generator launch: ") + hipGetErrorString(e)); }
    return build_from_dense(ctx, d, nullptr, true, nx, ny, nz, format, layout);
}

int vk_volume_generate_xor(vk_ctx *ctx, uint32_t nx, uint32_t ny, uint32_t nz, float time) {
    int dummy = 0;
    int rc = check_volume_args(ctx, &dummy, &dummy, nx, ny, nz, VK_FMT_RGBA16F_PAIR, VK_LAYOUT_LINEAR);
    if (rc) return rc;
    if (!std::isfinite(time)) return fail(ctx, VK_ERR_INVALID, "time must be finite");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_vox = (size_t)nx * ny * nz;
    if ((n_vox + 255) / 256 >= (1ull << 31)) return fail(ctx, VK_ERR_UNSUPPORTED, "volume too large");
    void *d = nullptr, *d2 = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, n_vox * 8));
    hipError_t e = hipMalloc(&d2, n_vox * 8);
    if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, VK_ERR_OOM, std::string("xor normals: ") + hipGetErrorString(e)); }
    hipLaunchKernelGGL(xor_generate_kernel, dim3((uint32_t)((n_vox + 255) / 256)), dim3(256), 0, ctx->stream, (uint2 *)d, (uint2 *)d2, nx, ny, nz, time);
    e = hipGetLastError();
    if (e != hipSuccess) { (void)hipFree(d); (void)hipFree(d2); return fail(ctx, VK_ERR_HIP, std::string("xor generator launch: ") + hipGetErrorString(e)); }
    return build_from_dense(ctx, d, d2, true, nx, ny, nz, VK_FMT_RGBA16F_PAIR, VK_LAYOUT_AUTO);
}

int vk_volume_empty_fraction(vk_ctx *ctx, double *fraction) {
    if (!ctx || !fraction) return VK_ERR_INVALID;
    if (ctx->format < 0) return fail(ctx, VK_ERR_INVALID, "no volume uploaded");
    *fraction = ctx->empty_fraction;
    return VK_OK;
}

int vk_volume_info(vk_ctx *ctx, uint32_t dims[3], int *format, int *layout, size_t *device_bytes) {
    if (!ctx) return VK_ERR_INVALID;
    if (ctx->format < 0) return fail(ctx, VK_ERR_INVALID, "no volume uploaded");
    if (dims) { dims[0] = ctx->nx; dims[1] = ctx->ny; dims[2] = ctx->nz; }
    if (format) *format = ctx->format;
    if (layout) *layout = ctx->layout;
    if (device_bytes) *device_bytes = ctx->vol_bytes;
    return VK_OK;
}

}  // extern "C"
