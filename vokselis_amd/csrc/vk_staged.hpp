// vk_staged.hpp -- NAIVE_TRILINEAR for volumes far larger than the caches: 8^3 bricks staged through LDS.
//
// raycast_naive.wgsl:96-119 issues one texture fetch per step; on CDNA4 that is 8 taps, and once the volume no
// longer fits the caches the four scattered x-pair loads per lane of the 9^3-brick kernel keep the texture-address
// path busy (49 cache-line look-ups per wave-level load, profiles/r01_c4_b9_pmc_summary.txt).  Here a wave stages
// the voxels its 64 rays are about to sample into LDS with coalesced 16-byte LDS-DMA loads and reads the taps with
// ds_read:
//
//  * HBM layout: dense voxels in 8^3 bricks (u8: 16 x 8 x 8), NO apron; clamp-to-edge is a replicated border of
//    kStagePad voxels baked in around the volume.  A brick is 64 "pieces" of 16 bytes (8 f16 / 16 u8 voxels along
//    the copy's FAST axis); the 8 pieces of a 128-byte line are 8 consecutive rows along the MID axis, the 8 lines
//    of a brick are its 8 slices along the SLOW axis.  There are up to three copies, one per SLOW axis: a wave
//    uses the copy whose SLOW axis is the major axis of its rays, so the window it needs is a few slices thick,
//    every fetched line is used, and the 8/16-voxel granularity of a piece falls on a lateral axis where the
//    window is wide anyway.  (288 GB of HBM: three copies of 2048^3 u8 are 26 GB.)
//  * LDS window: an axis-aligned box of voxels, dense, [slow][mid][fast], rows whole pieces -- so the window is
//    contiguous in piece order and one global_load_lds_dwordx4 fills 64 consecutive pieces from 64 arbitrary
//    source addresses.  No apron, no per-brick padding: the window holds what the rays need plus alignment.
//  * Rounds are slabs of T cells along the wave's major axis (see march_staged_perm): the window's range along
//    that axis is exact, its lateral range is a wave-reduced bounding box with a proven margin, so the step loop
//    carries no residency test and no clamp.  A box that exceeds the LDS budget takes a thinner slab; what even T = 1 cannot
//    hold is served from global memory (always correct, slow, never on the benchmark configurations).
//  * The arithmetic on the taps is that of march() (LINEAR / 9^3 layouts): frames are bitwise equal.
#pragma once

#include "vk_march.hpp"

namespace vk {

// floor(q / d) = umulhi(q, kMagic[d]) for q * d < 2^32, d in [2, 64]
__device__ const uint32_t kStageMagic[65] = {
    0u, 0u, 2147483649u, 1431655766u, 1073741825u, 858993460u, 715827883u, 613566757u, 536870913u, 477218589u, 429496730u, 390451573u, 357913942u,
    330382100u, 306783379u, 286331154u, 268435457u, 252645136u, 238609295u, 226050911u, 214748365u, 204522253u, 195225787u, 186737709u, 178956971u,
    171798692u, 165191050u, 159072863u, 153391690u, 148102321u, 143165577u, 138547333u, 134217729u, 130150525u, 126322568u, 122713352u, 119304648u,
    116080198u, 113025456u, 110127367u, 107374183u, 104755300u, 102261127u, 99882961u, 97612894u, 95443718u, 93368855u, 91382283u, 89478486u,
    87652394u, 85899346u, 84215046u, 82595525u, 81037119u, 79536432u, 78090315u, 76695845u, 75350304u, 74051161u, 72796056u, 71582789u, 70409300u,
    69273667u, 68174085u, 67108865u};

// Wave-wide minimum / maximum by DPP, all 64 lanes active; the results are wave-uniform (read from lane 63).
// Written as fused v_min/v_max_i32_dpp: from C++ the compiler emits v_mov + s_nop + v_mov_dpp + v_min per stage
// (25 issue slots per reduction; a round makes five).  A DPP operand written by the previous VALU instruction needs two
// wait states: the single form pays them as s_nop, the four-way form interleaves four independent reductions so that
// every dependent pair is four instructions apart.
#define VK_DPP_STAGES(OP, R)                                                             \
    OP " %" #R ", %" #R ", %" #R " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
    OP " %" #R ", %" #R ", %" #R " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
    OP " %" #R ", %" #R ", %" #R " row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
    OP " %" #R ", %" #R ", %" #R " row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"          \
    OP " %" #R ", %" #R ", %" #R " row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"        \
    OP " %" #R ", %" #R ", %" #R " row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1\n\t"

__device__ __forceinline__ int wave_min_i32(int v) {
    int r;
    asm volatile("s_nop 1\n\t" VK_DPP_STAGES("v_min_i32_dpp", 1) "v_readlane_b32 %0, %1, 63" : "=s"(r), "+v"(v));
    return __builtin_amdgcn_readfirstlane(r);  // (an asm with a vector output counts as divergent: re-assert that r is wave-uniform)
}
__device__ __forceinline__ int wave_max_i32(int v) {
    int r;
    asm volatile("s_nop 1\n\t" VK_DPP_STAGES("v_max_i32_dpp", 1) "v_readlane_b32 %0, %1, 63" : "=s"(r), "+v"(v));
    return __builtin_amdgcn_readfirstlane(r);
}
#define VK_DPP4(CTRL)                                              \
    "v_min_i32_dpp %4, %4, %4 " CTRL "\n\tv_max_i32_dpp %5, %5, %5 " CTRL "\n\t" \
    "v_min_i32_dpp %6, %6, %6 " CTRL "\n\tv_max_i32_dpp %7, %7, %7 " CTRL "\n\t"
// (min a, max b, min c, max d) over the wave, interleaved
__device__ __forceinline__ void wave_minmax4(int a, int b, int c, int d, int &ra, int &rb, int &rc, int &rd) {
    asm volatile("s_nop 1\n\t"
                 VK_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 VK_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 VK_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 VK_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                 VK_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 VK_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n\t"
                 "v_readlane_b32 %0, %4, 63\n\tv_readlane_b32 %1, %5, 63\n\tv_readlane_b32 %2, %6, 63\n\tv_readlane_b32 %3, %7, 63"
                 : "=s"(ra), "=s"(rb), "=s"(rc), "=s"(rd), "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    ra = __builtin_amdgcn_readfirstlane(ra); rb = __builtin_amdgcn_readfirstlane(rb);  // wave-uniform, see wave_min_i32
    rc = __builtin_amdgcn_readfirstlane(rc); rd = __builtin_amdgcn_readfirstlane(rd);
}

// mask bit set ? a : b, the lane mask in a scalar register pair (v_cndmask_b32_e64)
__device__ __forceinline__ int select_i32(unsigned long long mask, int a, int b) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// The 8 taps of a sample from the LDS window: pair k (two consecutive elements along the FAST axis) at byte
// address a[k].  Eight naturally aligned element reads, in inline assembly: written in C++ the compiler merges the
// two halves of a pair into one ds_read_b32 at 2-byte alignment, which is legal but slow on gfx950 (C4: 5.7 ms
// with such reads, 2.8 ms with every address forced to a multiple of 4).  The loads are followed by their own
// s_waitcnt, tied to the results, because the compiler does not count loads it cannot see.
template <bool U8>
__device__ __forceinline__ void lds_tap_pairs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t lo[4], uint32_t hi[4]) {
    uint32_t l0, l1, l2, l3, h0, h1, h2, h3;
    if constexpr (U8) {
        asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %8 offset:1\n\tds_read_u8 %2, %9\n\tds_read_u8 %3, %9 offset:1\n\t"
                     "ds_read_u8 %4, %10\n\tds_read_u8 %5, %10 offset:1\n\tds_read_u8 %6, %11\n\tds_read_u8 %7, %11 offset:1"
                     : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    } else {
        asm volatile("ds_read_u16 %0, %8\n\tds_read_u16 %1, %8 offset:2\n\tds_read_u16 %2, %9\n\tds_read_u16 %3, %9 offset:2\n\t"
                     "ds_read_u16 %4, %10\n\tds_read_u16 %5, %10 offset:2\n\tds_read_u16 %6, %11\n\tds_read_u16 %7, %11 offset:2"
                     : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2), "+v"(l3), "+v"(h3));
    lo[0] = l0; lo[1] = l1; lo[2] = l2; lo[3] = l3; hi[0] = h0; hi[1] = h1; hi[2] = h2; hi[3] = h3;
}

// f16 taps enter the filter straight from the low halves of their registers: v_fma_mix_f32 converts exactly and
// rounds once, so  mix_sub(b, a) == (float)b - (float)a  and  mix_lerp(f, d, a) == fmaf(f, d, (float)a)  bit for bit.
__device__ __forceinline__ float mix_sub(uint32_t b, uint32_t a) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
__device__ __forceinline__ float mix_lerp(float f, float d, uint32_t a) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(f), "v"(d), "v"(a));
    return r;
}
// a * b + c on 24-bit signed operands, b wave-uniform (window addressing: the row and slice pitches)
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// One sample's transfer function, colour and compositing: the statements of march() after the filter.
template <int SCALE>
__device__ __forceinline__ void composite_step(float v, float &A, float &Gr, float &Gg, float &Gb) {
    const float a = transfer_alpha<SCALE>(v);
    constexpr double kk = 6.28318 / 6.283185307179586476925;
    constexpr float pc0 = (float)(1.0 * kk), pc1 = (float)(1.7 * kk), pc2 = (float)(0.4 * kk);
    constexpr float pd1 = (float)(0.15 * kk), pd2 = (float)(0.20 * kk);
    const float cr = __builtin_amdgcn_cosf(a * pc0);
    const float cg = __builtin_amdgcn_cosf(fmaf(a, pc1, pd1));
    const float cb = __builtin_amdgcn_cosf(fmaf(a, pc2, pd2));
    const float w = (1.0f - A) * a;  // :112-114
    Gr = fmaf(w, cr, Gr); Gg = fmaf(w, cg, Gg); Gb = fmaf(w, cb, Gb);
    A = A + w;
}

// The filter on four pairs along the copy's FAST axis: pair k = (lo[k], hi[k]) sits at MID offset k & 1, SLOW
// offset k >> 1.  lerp x, then y, then z, as the oracle: c = fma(f, b - a, a).
template <bool U8, int PERM>
__device__ __forceinline__ float filter_pairs(const uint32_t lo[4], const uint32_t hi[4], float fx, float fy, float fz) {
    constexpr int S = PERM, F = (PERM + 1) % 3, M = (PERM + 2) % 3;
    uint32_t tp[8];  // tap index dx + 2*dy + 4*dz
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            int d[3];
            d[F] = e; d[M] = k & 1; d[S] = k >> 1;
            tp[d[0] + 2 * d[1] + 4 * d[2]] = e ? hi[k] : lo[k];
        }
    // u8 taps take the f16 path too: the byte b, read as an f16 bit pattern, is the subnormal b * 2^-24, which
    // v_fma_mix_f32 converts exactly.  The whole filter is linear, every intermediate stays in f32's normal range, and
    // a power-of-two scale commutes with rounding: each value below is 2^-24 times the specified one, bit for bit, and
    // the transfer function's constants fold the 2^24 back (exactly).  Eight conversions per sample saved.
    float dl[4], c[4];  // x edges at (dy, dz) = (0,0) (1,0) (0,1) (1,1); the four differences first: no back-to-back dependence
#pragma unroll
    for (int j = 0; j < 4; j++) dl[j] = mix_sub(tp[2 * j + 1], tp[2 * j]);
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = mix_lerp(fx, dl[j], tp[2 * j]);
    const float c0 = fmaf(fy, c[1] - c[0], c[0]), c1 = fmaf(fy, c[3] - c[2], c[2]);
    return fmaf(fz, c1 - c0, c0);  // u8: 2^-24 times the filtered taps (transfer_alpha<2> carries the scale)
}

// One step with the 8 taps read from the brick copy itself (global memory): the always-correct path for what the
// window cannot serve.
template <int VOL, int PERM>
__device__ __forceinline__ float sample_global(const StagedDesc &D, const float p[3], const float fn[3]) {
    constexpr int S = PERM, F = (PERM + 1) % 3, M = (PERM + 2) % 3;
    constexpr bool U8 = (VOL == VOL_S8U8);
    constexpr int BPV = U8 ? 1 : 2, VSH = U8 ? 4 : 3;
    const unsigned char *const base = D.copy[PERM];
    const uint32_t npf = D.npf[PERM], nbm = D.nbm[PERM];
    const float ux = fmaf(p[0], fn[0], -0.5f), uy = fmaf(p[1], fn[1], -0.5f), uz = fmaf(p[2], fn[2], -0.5f);
    const int i[3] = {cvt_floor_i32(ux), cvt_floor_i32(uy), cvt_floor_i32(uz)};
    const float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy), fz = __builtin_amdgcn_fractf(uz);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const uint32_t vf = (uint32_t)min(max(i[F] + kStagePad + e, 0), (int)D.nv[F] - 1);
            const uint32_t vm = (uint32_t)min(max(i[M] + kStagePad + (k & 1), 0), (int)D.nv[M] - 1);
            const uint32_t vs = (uint32_t)min(max(i[S] + kStagePad + (k >> 1), 0), (int)D.nv[S] - 1);
            const uint32_t brick = (vf >> VSH) + npf * ((vm >> 3) + nbm * (vs >> 3));
            const unsigned char *g = base + (((uint64_t)brick << 10) | (uint64_t)(((vs & 7u) << 7) | ((vm & 7u) << 4))) + (vf & ((1u << VSH) - 1u)) * BPV;
            const uint32_t v = U8 ? (uint32_t)*g : (uint32_t)*reinterpret_cast<const uint16_t *>(g);
            if (e) hi[k] = v; else lo[k] = v;
        }
    }
    return filter_pairs<U8, PERM>(lo, hi, fx, fy, fz);
}

// The march of one wave on copy PERM (SLOW axis S = the wave's major axis).
//
// Rounds are SLABS of T cells along S, taken in the wave's direction of travel: in a round every ray takes the
// steps whose sample cell lies in the slab (about 2T of them at dt_scale 0.5), rays that have not reached the slab
// wait, rays that have ended are gone.  Synchronising on position instead of on the step index keeps the box thin
// whatever the entry face: rays of an oblique bundle enter far apart along their major axis.
//  * The window's S range is exact (the slab test is made on the very cell index a step samples).
//  * Its lateral range is the bounding box, over the rays in the slab, of the segment from the ray's position now
//    to where it crosses the slab's far plane (positions are linear in S: two fmas per axis), reduced over the wave
//    (4 reductions) with 1/32 voxel of margin against the rounding of the accumulated position (<= 1e-3 voxel).
//  * A box larger than the LDS budget takes a thinner slab; at T = 1 the step is served from global memory.  Rays that do not
//    travel with the wave along S (opposite sign, or fewer than 0.2 cells per step: a bounded lateral slope is what
//    bounds the box) are marched from global memory after the others -- adjacent pixels do not produce such rays
//    at any sane field of view; the path exists for safety.
template <int VOL, int PERM, bool COUNT>
__device__ __forceinline__ void march_staged_perm(const VolumeDesc &V, const StagedDesc &D, RayState &r, bool alive, Census &cs, const uint32_t lane) {
    constexpr int S = PERM, F = (PERM + 1) % 3, M = (PERM + 2) % 3;
    constexpr bool U8 = (VOL == VOL_S8U8);
    constexpr int BPV = U8 ? 1 : 2, VSH = U8 ? 4 : 3;
    extern __shared__ unsigned char stage_win[];
    const uint32_t win_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)stage_win;  // LDS byte address of the window
    float A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    uint32_t left = r.left;  // iterations of :101 not made yet (RayState)
    float p[3] = {r.px, r.py, r.pz};
    const float s[3] = {r.sx, r.sy, r.sz};
    const float fn[3] = {(float)V.nx, (float)V.ny, (float)V.nz};
    const unsigned char *const base = D.copy[PERM];
    const uint32_t npf = D.npf[PERM], nbm = D.nbm[PERM], cap = D.cap_bytes;
    const int T0 = (int)D.slab_cells;
    const int nvm1[3] = {(int)D.nv[0] - 1, (int)D.nv[1] - 1, (int)D.nv[2] - 1};

    alive = alive && (left != 0u && A < 0.95f);
    const float duS = s[S] * fn[S];  // cells per step along S
    const bool dir_up = __popcll(__ballot(alive && duS > 0.0f)) >= __popcll(__ballot(alive && duS < 0.0f));  // wave-uniform
    const bool fit = alive && (dir_up ? duS >= 0.2f : duS <= -0.2f);
    const float inv = fit ? 1.0f / duS : 0.0f;
    const float ratM = (s[M] * fn[M]) * inv, ratF = (s[F] * fn[F]) * inv;  // lateral cells per cell of S

    int Tprev = T0, sig_next = 0;
    uint32_t round_no = 0;
    bool have_sig = false;
    const uint64_t layerB = ((uint64_t)npf * (uint64_t)nbm) << 10;  // bytes of one layer of bricks along S
    for (;;) {
        const bool live = fit && (left != 0u && A < 0.95f);
        if (__ballot(live) == 0ull) break;  // wave-uniform
        const float uS = fmaf(p[S], fn[S], -0.5f), uM = fmaf(p[M], fn[M], -0.5f), uF = fmaf(p[F], fn[F], -0.5f);
        const int iS = cvt_floor_i32(uS);
        // The slab starts at the rearmost live ray -- found by a reduction in the first round and after a fallback round;
        // after a marched slab every live ray has left it (or was ahead of it), so the cell behind its far plane is a valid
        // start without looking (if the rearmost ray is further on, the slab only holds a few unused cells).
        const int sig = have_sig ? sig_next : (dir_up ? wave_min_i32(live ? iS : 0x7fffffff) : wave_max_i32(live ? iS : (int)0x80000000));
        // The box changes slowly from round to round: the search starts at the last fit, and one above it only every `grow_every`-th round
        // (tried every round, the thicker slab fails most of the time -- ~25 scalar instructions of a round's ~240, on a kernel at its
        // issue-slot limit).  Any T is exact: a thinner slab only means more rounds.
        int T = min(T0, Tprev + ((round_no++ % D.grow_every) == 0u ? 1 : 0));
        int clo = dir_up ? sig : sig - T + 1;
        const bool inslab = live && (uint32_t)(iS - clo) < (uint32_t)T;
        // lateral bounds of the rays in the slab: from here to the slab's far plane
        const float e = (dir_up ? (float)(clo + T) : (float)clo) - uS;
        const float eM = fmaf(e, ratM, uM), eF = fmaf(e, ratF, uF);
        const int kMl = cvt_floor_i32(fminf(uM, eM) * 64.0f), kMh = cvt_floor_i32(fmaxf(uM, eM) * 64.0f);
        const int kFl = cvt_floor_i32(fminf(uF, eF) * 64.0f), kFh = cvt_floor_i32(fmaxf(uF, eF) * 64.0f);
        int bMl, bMh, bFl, bFh;
        // (the four selects through an SGPR-pair mask: four v_cndmask in a row through VCC, as the compiler writes them, cost ~16 issue
        // cycles each -- profiles/r03_ubench_valu_issue_rate.txt -- against 4 in this form)
        const unsigned long long in_mask = __ballot(inslab);
        wave_minmax4(select_i32(in_mask, kMl, 0x7fffffff), select_i32(in_mask, kMh, (int)0x80000000), select_i32(in_mask, kFl, 0x7fffffff), select_i32(in_mask, kFh, (int)0x80000000), bMl, bMh, bFl, bFh);
        bMl -= 2; bMh += 3; bFl -= 2; bFh += 3;
        const int ilM = min(max((bMl >> 6) + kStagePad, 0), nvm1[M]), ihM = min(max((bMh >> 6) + kStagePad + 1, 0), nvm1[M]);  // + 1: the upper tap
        const int ilF = min(max((bFl >> 6) + kStagePad, 0), nvm1[F]), ihF = min(max((bFh >> 6) + kStagePad + 1, 0), nvm1[F]);
        const uint32_t pf0 = (uint32_t)ilF >> VSH, Efn = ((uint32_t)ihF >> VSH) - pf0 + 1u, Em = (uint32_t)(ihM - ilM) + 1u;
        // row pitch in pieces: the pieces the rays need, plus one when that makes the pitch odd (row_pad): rows of 4 pieces (64 B) put every
        // fourth window row -- i.e. every second pixel row of the wave -- on the same LDS banks.  The extra piece is fetched like the others
        // (the DMA writes 64 consecutive pieces per instruction: a hole cannot be skipped) and never read.
        const uint32_t Efp = (D.row_pad && !(Efn & 1u) && Efn < 64u) ? Efn + 1u : Efn;
        // the thickest slab whose box fits the window (scalar)
        int ilS;
        uint32_t Es;
        bool fits;
        for (;;) {
            clo = dir_up ? sig : sig - T + 1;
            ilS = min(max(clo + kStagePad, 0), nvm1[S]);
            Es = (uint32_t)(min(max(clo + T + kStagePad, 0), nvm1[S]) - ilS) + 1u;  // cells clo .. clo+T-1 and the upper tap
            fits = Efp <= 64u && Em <= 64u && Es * Em * Efp * 16u <= cap;
            if (fits || T == 1) break;
            T--;
        }
        Tprev = T;
        have_sig = fits;
        sig_next = dir_up ? clo + T : clo - 1;
        if (fits) {
            // ---- fill: piece (slice, q) of the window [slow][mid][fast-piece] <- its 16 bytes in the copy.  A lane keeps its
            // (row, piece) of the slice, i.e. a 32-bit offset inside one layer of bricks, for every slice; the slice only
            // moves the wave-uniform base: no vector arithmetic per load.
            const uint32_t slicePieces = Em * Efp;
            const uint32_t mgF = kStageMagic[Efp];
            for (uint32_t j = 0; j < slicePieces; j += 64u) {
                const uint32_t q = j + lane;
                if (q < slicePieces) {
                    const uint32_t m = Efp == 1u ? q : __umulhi(q, mgF);
                    const uint32_t f = q - m * Efp;
                    const uint32_t mm = (uint32_t)ilM + m;
                    const uint32_t voff = ((pf0 + min(f, Efn - 1u) + npf * (mm >> 3)) << 10) | ((mm & 7u) << 4);  // (f == Efn: the pad piece repeats the row's last one)
                    // global_load_lds_dwordx4 with the slice's base in a scalar pair and the lane's 32-bit offset: no vector
                    // arithmetic per load (the builtin only takes a 64-bit per-lane address: one v_lshl_add_u64 each).  The
                    // base moves by 128 bytes per slice inside a layer of bricks and by the rest of the layer at a brick
                    // boundary; M0 (the LDS destination) is saved and restored once around the slices.
                    uint32_t sv = (uint32_t)ilS;
                    const unsigned char *sbase = base + layerB * (uint64_t)(sv >> 3) + ((sv & 7u) << 7);
                    uint32_t lds_dst = win_lds + j * 16u;
                    uint32_t keep_m0;
                    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
                    for (uint32_t si = 0; si < Es;) {
                        const uint32_t run = min(8u - (sv & 7u), Es - si);  // slices left in this layer of bricks
                        for (uint32_t k = 0; k < run; k++) {
                            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
                            sbase += 128;
                            lds_dst += slicePieces * 16u;
                        }
                        si += run; sv += run;
                        sbase += layerB - 1024u;  // (from the end of this layer's 8 slices to the next layer's first)
                    }
                    asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- the steps inside the slab, taps from the window
            const int rowB = (int)(Efp * 16u), sliceB = (int)(Em * Efp * 16u);
            const int cbase = (kStagePad - ilS) * sliceB + (kStagePad - ilM) * rowB + (kStagePad - (int)(pf0 << VSH)) * BPV + (int)win_lds;
            if (live) {
                for (;;) {
                    const float ux = fmaf(p[0], fn[0], -0.5f), uy = fmaf(p[1], fn[1], -0.5f), uz = fmaf(p[2], fn[2], -0.5f);
                    const int i[3] = {cvt_floor_i32(ux), cvt_floor_i32(uy), cvt_floor_i32(uz)};
                    // one exit test per step: the ray ended (:101, :115-117) or left the slab (or has not reached it)
                    if (!((int)(left != 0u) & (int)(A < 0.95f) & (int)((uint32_t)(i[S] - clo) < (uint32_t)T))) break;
                    const float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy), fz = __builtin_amdgcn_fractf(uz);
                    const int a0 = mad_i24(i[S], sliceB, mad_i24(i[M], rowB, i[F] * BPV + cbase));  // |operands| < 2^23
                    uint32_t lo[4], hi[4];
                    lds_tap_pairs<U8>((uint32_t)a0, (uint32_t)(a0 + rowB), (uint32_t)(a0 + sliceB), (uint32_t)(a0 + sliceB + rowB), lo, hi);
                    const float v = filter_pairs<U8, PERM>(lo, hi, fx, fy, fz);
                    composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
                    if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; }
                    p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];  // :118
                    left -= 1u;  // :101
                }
            }
        } else if (live && iS == sig) {
            // ---- even a one-cell slab exceeds the window: the rearmost rays take one step from global memory
            const float v = sample_global<VOL, PERM>(D, p, fn);
            composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
            if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; cs.n_fb++; }
            p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];
            left -= 1u;
        }
        if (COUNT && wave_leader()) { cs.w_outer++; if (!fits) cs.w_inner++; cs.w_sample += (uint32_t)T; }
    }
    // rays that do not travel with the wave along S
    if (__ballot(alive && !fit) != 0ull) {
        if (alive && !fit) {
            while (left != 0u && A < 0.95f) {
                const float v = sample_global<VOL, PERM>(D, p, fn);
                composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
                if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; cs.n_fb++; }
                p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];
                left -= 1u;
            }
        }
    }
    r.left = left; r.px = p[0]; r.py = p[1]; r.pz = p[2]; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
}

// ---- the same march with ONE window for the four waves of a 256-thread group (2 x 2 neighbouring 8x8 blocks) ------------------
// A round's fixed work -- slab start, lateral bounds, slab search, fill -- is paid per round; the rays of 16 x 16 pixels sweep a box that is
// far smaller than four boxes of 8 x 8 pixels (the lateral drift of oblique rays and the 16-byte piece granularity are paid once), so with
// four waves' LDS the slab is about twice as thick and a ray meets half as many rounds.  Everything that was wave-uniform in
// march_staged_perm is group-uniform here: the copy, the direction of travel, the slab, the box, `fits` -- exchanged through a few LDS
// accumulators (one lane per wave: ds_min / ds_max; a barrier; a broadcast read).  Two barriers per round (three when the slab start has
// to be found by a reduction): bounds -> fill -> steps.  No wave leaves the loop before the whole group is done (a wave without live
// rays still fills its share), so every wave executes the same barriers.  The arithmetic per ray is march_staged_perm's: frames are
// bitwise equal.
constexpr uint32_t kGroupWaves = 4u, kGroupExchBytes = 256u;
constexpr uint32_t kGroupAccI = 0u, kGroupAccS = 16u, kGroupAccB = 24u;  // int offsets: 9 set-up sums | 2 x {any, sig} | 2 x {any, bMl, bMh, bFl, bFh, -, -, -}

template <int VOL, int PERM, bool COUNT>
__device__ __forceinline__ void march_staged_group_perm(const VolumeDesc &V, const StagedDesc &D, RayState &r, bool alive, Census &cs, const uint32_t lane,
                                                        const uint32_t wave, const bool dir_up) {
    constexpr int S = PERM, F = (PERM + 1) % 3, M = (PERM + 2) % 3;
    constexpr bool U8 = (VOL == VOL_S8U8);
    constexpr int BPV = U8 ? 1 : 2, VSH = U8 ? 4 : 3;
    extern __shared__ unsigned char stage_win[];
    int *const ex = reinterpret_cast<int *>(stage_win);  // the group's exchange block (kGroupExchBytes), then the window
    const uint32_t win_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)stage_win + kGroupExchBytes;  // LDS byte address of the window
    float A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    uint32_t left = r.left;  // iterations of :101 not made yet (RayState)
    float p[3] = {r.px, r.py, r.pz};
    const float s[3] = {r.sx, r.sy, r.sz};
    const float fn[3] = {(float)V.nx, (float)V.ny, (float)V.nz};
    const unsigned char *const base = D.copy[PERM];
    const uint32_t npf = D.npf[PERM], nbm = D.nbm[PERM], cap = D.cap_bytes;
    const int T0 = (int)D.slab_cells;
    const int nvm1[3] = {(int)D.nv[0] - 1, (int)D.nv[1] - 1, (int)D.nv[2] - 1};

    alive = alive && (left != 0u && A < 0.95f);
    const float duS = s[S] * fn[S];  // cells per step along S
    // (dir_up: the GROUP's direction of travel along S, decided by the kernel over all four waves)
    const bool fit = alive && (dir_up ? duS >= 0.2f : duS <= -0.2f);
    const float inv = fit ? 1.0f / duS : 0.0f;
    const float ratM = (s[M] * fn[M]) * inv, ratF = (s[F] * fn[F]) * inv;  // lateral cells per cell of S

    int Tprev = T0, sig_next = 0;
    uint32_t round_no = 0;
    bool have_sig = false;
    const uint64_t layerB = ((uint64_t)npf * (uint64_t)nbm) << 10;  // bytes of one layer of bricks along S
    uint32_t par = 0;  // the exchange accumulators alternate between two sets (see group_exchange)
    for (;;) {
        const bool live = fit && (left != 0u && A < 0.95f);
        const int any_mine = __ballot(live) != 0ull ? 1 : 0;
        const float uS = fmaf(p[S], fn[S], -0.5f), uM = fmaf(p[M], fn[M], -0.5f), uF = fmaf(p[F], fn[F], -0.5f);
        const int iS = cvt_floor_i32(uS);
        // The slab starts at the rearmost live ray -- found by a reduction in the first round and after a fallback round;
        // after a marched slab every live ray has left it (or was ahead of it), so the cell behind its far plane is a valid
        // start without looking (if the rearmost ray is further on, the slab only holds a few unused cells).
        int sig = sig_next;
        if (!have_sig) {  // group-uniform: the rearmost live ray of the four waves
            const int mine = dir_up ? wave_min_i32(live ? iS : 0x7fffffff) : wave_max_i32(live ? iS : (int)0x80000000);
            int *const as = ex + kGroupAccS + par * 2;
            if (lane == 0u) { atomicMax(as, any_mine); if (dir_up) atomicMin(as + 1, mine); else atomicMax(as + 1, mine); }
            __syncthreads();
            const int any_g = __builtin_amdgcn_readfirstlane(as[0]);
            sig = __builtin_amdgcn_readfirstlane(as[1]);
            if (any_g == 0) break;  // nobody in the group has a live ray that travels with it
        }
        // The box changes slowly from round to round: the search starts at the last fit, and one above it only every `grow_every`-th round
        // (tried every round, the thicker slab fails most of the time -- ~25 scalar instructions of a round's ~240, on a kernel at its
        // issue-slot limit).  Any T is exact: a thinner slab only means more rounds.
        int T = min(T0, Tprev + ((round_no++ % D.grow_every) == 0u ? 1 : 0));
        int clo = dir_up ? sig : sig - T + 1;
        const bool inslab = live && (uint32_t)(iS - clo) < (uint32_t)T;
        // lateral bounds of the rays in the slab: from here to the slab's far plane
        const float e = (dir_up ? (float)(clo + T) : (float)clo) - uS;
        const float eM = fmaf(e, ratM, uM), eF = fmaf(e, ratF, uF);
        const int kMl = cvt_floor_i32(fminf(uM, eM) * 64.0f), kMh = cvt_floor_i32(fmaxf(uM, eM) * 64.0f);
        const int kFl = cvt_floor_i32(fminf(uF, eF) * 64.0f), kFh = cvt_floor_i32(fmaxf(uF, eF) * 64.0f);
        int bMl, bMh, bFl, bFh;
        // (the four selects through an SGPR-pair mask: four v_cndmask in a row through VCC, as the compiler writes them, cost ~16 issue
        // cycles each -- profiles/r03_ubench_valu_issue_rate.txt -- against 4 in this form)
        const unsigned long long in_mask = __ballot(inslab);
        wave_minmax4(select_i32(in_mask, kMl, 0x7fffffff), select_i32(in_mask, kMh, (int)0x80000000), select_i32(in_mask, kFl, 0x7fffffff), select_i32(in_mask, kFh, (int)0x80000000), bMl, bMh, bFl, bFh);
        {   // ... and over the group: LDS atomics by one lane per wave, a barrier, one broadcast read.  The barrier is also the one between the
            // steps of the round before (window reads) and this round's fill (window writes).
            int *const ab = ex + kGroupAccB + par * 8;
            if (lane == 0u) { atomicMax(ab, any_mine); atomicMin(ab + 1, bMl); atomicMax(ab + 2, bMh); atomicMin(ab + 3, bFl); atomicMax(ab + 4, bFh); }
            __syncthreads();
            const int any_g = __builtin_amdgcn_readfirstlane(ab[0]);
            bMl = __builtin_amdgcn_readfirstlane(ab[1]); bMh = __builtin_amdgcn_readfirstlane(ab[2]);
            bFl = __builtin_amdgcn_readfirstlane(ab[3]); bFh = __builtin_amdgcn_readfirstlane(ab[4]);
            // the other set is idle until the next round's atomics, which every wave issues after the barrier below: reset it now
            if (wave == 0u && lane < 8u) {
                const uint32_t o = par ^ 1u;
                ex[kGroupAccB + o * 8 + lane] = (lane == 0u || lane > 4u) ? 0 : ((lane & 1u) ? 0x7fffffff : (int)0x80000000);
                if (lane < 2u) ex[kGroupAccS + o * 2 + lane] = lane == 0u ? 0 : (dir_up ? 0x7fffffff : (int)0x80000000);
            }
            if (any_g == 0) break;  // group-uniform
        }
        bMl -= 2; bMh += 3; bFl -= 2; bFh += 3;
        const int ilM = min(max((bMl >> 6) + kStagePad, 0), nvm1[M]), ihM = min(max((bMh >> 6) + kStagePad + 1, 0), nvm1[M]);  // + 1: the upper tap
        const int ilF = min(max((bFl >> 6) + kStagePad, 0), nvm1[F]), ihF = min(max((bFh >> 6) + kStagePad + 1, 0), nvm1[F]);
        const uint32_t pf0 = (uint32_t)ilF >> VSH, Efn = ((uint32_t)ihF >> VSH) - pf0 + 1u, Em = (uint32_t)(ihM - ilM) + 1u;
        // row pitch in pieces: the pieces the rays need, plus one when that makes the pitch odd (row_pad): rows of 4 pieces (64 B) put every
        // fourth window row -- i.e. every second pixel row of the wave -- on the same LDS banks.  The extra piece is fetched like the others
        // (the DMA writes 64 consecutive pieces per instruction: a hole cannot be skipped) and never read.
        const uint32_t Efp = (D.row_pad && !(Efn & 1u) && Efn < 64u) ? Efn + 1u : Efn;
        // the thickest slab whose box fits the window (scalar)
        int ilS;
        uint32_t Es;
        bool fits;
        for (;;) {
            clo = dir_up ? sig : sig - T + 1;
            ilS = min(max(clo + kStagePad, 0), nvm1[S]);
            Es = (uint32_t)(min(max(clo + T + kStagePad, 0), nvm1[S]) - ilS) + 1u;  // cells clo .. clo+T-1 and the upper tap
            fits = Efp <= 64u && Em <= 256u && Es * Em * Efp * 16u <= cap;  // (a group's box is taller than a wave's)
            if (fits || T == 1) break;
            T--;
        }
        Tprev = T;
        have_sig = fits;
        sig_next = dir_up ? clo + T : clo - 1;
        if (fits) {
            // ---- fill: piece (slice, q) of the window [slow][mid][fast-piece] <- its 16 bytes in the copy.  A lane keeps its
            // (row, piece) of the slice, i.e. a 32-bit offset inside one layer of bricks, for every slice; the slice only
            // moves the wave-uniform base: no vector arithmetic per load.
            const uint32_t slicePieces = Em * Efp;
            const uint32_t mgF = kStageMagic[Efp];
            for (uint32_t j = wave * 64u; j < slicePieces; j += 64u * kGroupWaves) {  // chunks of 64 pieces, dealt over the group's waves
                const uint32_t q = j + lane;
                if (q < slicePieces) {
                    const uint32_t m = Efp == 1u ? q : __umulhi(q, mgF);
                    const uint32_t f = q - m * Efp;
                    const uint32_t mm = (uint32_t)ilM + m;
                    const uint32_t voff = ((pf0 + min(f, Efn - 1u) + npf * (mm >> 3)) << 10) | ((mm & 7u) << 4);  // (f == Efn: the pad piece repeats the row's last one)
                    // global_load_lds_dwordx4 with the slice's base in a scalar pair and the lane's 32-bit offset: no vector
                    // arithmetic per load (the builtin only takes a 64-bit per-lane address: one v_lshl_add_u64 each).  The
                    // base moves by 128 bytes per slice inside a layer of bricks and by the rest of the layer at a brick
                    // boundary; M0 (the LDS destination) is saved and restored once around the slices.
                    uint32_t sv = (uint32_t)ilS;
                    const unsigned char *sbase = base + layerB * (uint64_t)(sv >> 3) + ((sv & 7u) << 7);
                    uint32_t lds_dst = win_lds + j * 16u;
                    uint32_t keep_m0;
                    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
                    for (uint32_t si = 0; si < Es;) {
                        const uint32_t run = min(8u - (sv & 7u), Es - si);  // slices left in this layer of bricks
                        for (uint32_t k = 0; k < run; k++) {
                            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
                            sbase += 128;
                            lds_dst += slicePieces * 16u;
                        }
                        si += run; sv += run;
                        sbase += layerB - 1024u;  // (from the end of this layer's 8 slices to the next layer's first)
                    }
                    asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // every wave's share of the window has landed
            // ---- the steps inside the slab, taps from the window
            const int rowB = (int)(Efp * 16u), sliceB = (int)(Em * Efp * 16u);
            const int cbase = (kStagePad - ilS) * sliceB + (kStagePad - ilM) * rowB + (kStagePad - (int)(pf0 << VSH)) * BPV + (int)win_lds;
            if (live) {
                for (;;) {
                    const float ux = fmaf(p[0], fn[0], -0.5f), uy = fmaf(p[1], fn[1], -0.5f), uz = fmaf(p[2], fn[2], -0.5f);
                    const int i[3] = {cvt_floor_i32(ux), cvt_floor_i32(uy), cvt_floor_i32(uz)};
                    // one exit test per step: the ray ended (:101, :115-117) or left the slab (or has not reached it)
                    if (!((int)(left != 0u) & (int)(A < 0.95f) & (int)((uint32_t)(i[S] - clo) < (uint32_t)T))) break;
                    const float fx = __builtin_amdgcn_fractf(ux), fy = __builtin_amdgcn_fractf(uy), fz = __builtin_amdgcn_fractf(uz);
                    const int a0 = mad_i24(i[S], sliceB, mad_i24(i[M], rowB, i[F] * BPV + cbase));  // |operands| < 2^23
                    uint32_t lo[4], hi[4];
                    lds_tap_pairs<U8>((uint32_t)a0, (uint32_t)(a0 + rowB), (uint32_t)(a0 + sliceB), (uint32_t)(a0 + sliceB + rowB), lo, hi);
                    const float v = filter_pairs<U8, PERM>(lo, hi, fx, fy, fz);
                    composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
                    if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; }
                    p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];  // :118
                    left -= 1u;  // :101
                }
            }
        } else {
            __syncthreads();  // (keeps the rounds' barrier count uniform: the accumulator reset above relies on it)
          if (live && iS == sig) {
            // ---- even a one-cell slab exceeds the window: the rearmost rays take one step from global memory
            const float v = sample_global<VOL, PERM>(D, p, fn);
            composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
            if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; cs.n_fb++; }
            p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];
            left -= 1u;
          }
        }
        if (COUNT && wave_leader()) { cs.w_outer++; if (!fits) cs.w_inner++; cs.w_sample += (uint32_t)T; }
        par ^= 1u;
    }
    // rays that do not travel with the wave along S
    if (__ballot(alive && !fit) != 0ull) {
        if (alive && !fit) {
            while (left != 0u && A < 0.95f) {
                const float v = sample_global<VOL, PERM>(D, p, fn);
                composite_step<U8 ? 2 : 0>(v, A, Gr, Gg, Gb);
                if (COUNT) { cs.n_look++; cs.n_iter++; cs.n_samp++; cs.n_fb++; }
                p[0] = p[0] + s[0]; p[1] = p[1] + s[1]; p[2] = p[2] + s[2];
                left -= 1u;
            }
        }
    }
    r.left = left; r.px = p[0]; r.py = p[1]; r.pz = p[2]; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
}

// fs_main (raycast_naive.wgsl:83-125) on the staged layout.  Same ray set-up, output and counters as
// raymarch_naive_kernel; no lane leaves before the march: all 64 take part in the reductions and the fills.
template <int VOL, int OUT, bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void raymarch_staged_kernel(const LaunchDesc L, const VolumeDesc V, const StagedDesc D) {
    static_assert(VOL == VOL_S8U8 || VOL == VOL_S8F16, "staged layouts");
    if (blockIdx.x >= L.grid_march) { clear_inactive_strip<OUT>(L, blockIdx.x - L.grid_march, threadIdx.x); return; }  // wave-uniform
    const uint32_t lb = logical_block(blockIdx.x);
    if (lb >= L.n_blocks) return;  // wave-uniform
    const uint32_t lane = threadIdx.x;
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, lane);
    {
        const int bx0 = pm.x - (int)(lane & 7u), by0 = pm.y - (int)(lane >> 3);
        if (pm.pos >= fv.n_active || bx0 + 8 <= fv.cull_x0 || bx0 >= fv.cull_x1 || by0 + 8 <= fv.cull_y0 || by0 >= fv.cull_y1) {  // wave-uniform
            if (!pm.valid) return;
            store_out<OUT>(L, pm, 0.0f, 0.0f, 0.0f);
            if (COUNT && L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = 0;
            return;
        }
    }
    RayState r;
    r.left = 0u; r.px = r.py = r.pz = 0.0f; r.sx = r.sy = r.sz = 0.0f;
    r.A = 0.0f; r.Gr = r.Gg = r.Gb = 0.0f; r.out = 0;
    bool hit = false;
    if (pm.valid) {
        // --- ray: SURVEY A.1 step 1, as raymarch_naive_kernel ---
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
        if (!(t0 > t1)) {  // :91-93
            t0 = fmaxf(t0, 0.0f);  // :94
            const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
            float dtx = 1.0f / (fnx * fabsf(dir[0]));
            float dty = 1.0f / (fny * fabsf(dir[1]));
            float dtz = 1.0f / (fnz * fabsf(dir[2]));
            const float dt = L.dt_scale * fminf(dtx, fminf(dty, dtz));  // :97-99
            r.left = min(count_trips(t0, t1, dt), 0x7fffffffu);  // :101
            r.px = eye[0] + t0 * dir[0]; r.py = eye[1] + t0 * dir[1]; r.pz = eye[2] + t0 * dir[2];  // :100
            r.sx = dir[0] * dt; r.sy = dir[1] * dt; r.sz = dir[2] * dt;  // :118
            hit = true;
        }
    }
    // the wave's major axis (cells per step), by majority of its rays: selects the copy
    const float ax = fabsf(r.sx) * (float)V.nx, ay = fabsf(r.sy) * (float)V.ny, az = fabsf(r.sz) * (float)V.nz;
    const int mj = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const int c0 = __popcll(__ballot(hit && mj == 0)), c1 = __popcll(__ballot(hit && mj == 1)), c2 = __popcll(__ballot(hit && mj == 2));
    Census cs;
    if (c0 + c1 + c2 != 0) {  // wave-uniform
        if (L.flags & LF_WAVE_PRIORITY) set_wave_priority(hit, r.left, fmaxf((float)V.nx, fmaxf((float)V.ny, (float)V.nz)) / L.dt_scale);
        const int major = (c0 >= c1 && c0 >= c2) ? 0 : (c1 >= c2 ? 1 : 2);
        const uint32_t copy = D.copy_of_major[major];
        if (copy == 0u) march_staged_perm<VOL, 0, COUNT>(V, D, r, hit, cs, lane);
        else if (copy == 1u) march_staged_perm<VOL, 1, COUNT>(V, D, r, hit, cs, lane);
        else march_staged_perm<VOL, 2, COUNT>(V, D, r, hit, cs, lane);
    }
    if (!pm.valid) return;
    float Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    if (hit) {
        Cr = linear_to_srgb(fmaf(0.5f, r.Gr, 0.5f * r.A));  // :121-123
        Cg = linear_to_srgb(fmaf(0.5f, r.Gg, 0.5f * r.A));
        Cb = linear_to_srgb(fmaf(0.5f, r.Gb, 0.5f * r.A));
    }
    store_out<OUT>(L, pm, Cr, Cg, Cb);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = (L.flags & LF_STEPS_ARE_FALLBACKS) ? cs.n_fb : ((L.flags & LF_STEPS_ARE_TRIPS) ? cs.n_look : cs.n_iter);
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)cs.n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)cs.n_samp);
            atomicAdd(&L.counters[2], (unsigned long long)cs.w_outer);   // rounds
            atomicAdd(&L.counters[3], (unsigned long long)cs.w_inner);   // rounds served from global memory
            atomicAdd(&L.counters[4], (unsigned long long)cs.w_sample);  // sum of K over rounds
            atomicAdd(&L.counters[5], (unsigned long long)cs.n_look);
        }
    }
}

// The same pass with one window per 256-thread group: the four waves of a group are the 2 x 2 neighbouring 8x8 blocks of a tile (tile
// edge a multiple of 16).  Workgroups go round-robin over the XCDs, so a run of 128 consecutive groups gives every XCD the 16 groups of one
// 64 x 64 tile, as logical_block does for single waves.  No wave returns before the march unless the whole group does.
template <int VOL, int OUT, bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void raymarch_staged_group_kernel(const LaunchDesc L, const VolumeDesc V, const StagedDesc D) {
    static_assert(VOL == VOL_S8U8 || VOL == VOL_S8F16, "staged layouts");
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x * kGroupWaves >= L.grid_march) { clear_inactive_strip<OUT>(L, blockIdx.x * kGroupWaves + wave - L.grid_march, lane); return; }  // group-uniform
    const uint32_t lb = group_logical_block(blockIdx.x, wave, L.ts);  // (vk_hostmath.hpp; kGroupWaves == 4)
    if (lb >= L.n_blocks) return;  // group-uniform: n_blocks is a multiple of the blocks of a tile
    const FrameView fv = frame_view(L, lb);
    const PixelMap pm = map_pixel(L, fv, lane);
    if (pm.pos >= fv.n_active) {  // group-uniform: the whole tile is inactive
        if (!pm.valid) return;
        store_out<OUT>(L, pm, 0.0f, 0.0f, 0.0f);
        if (COUNT && L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = 0;
        return;
    }
    const int bx0 = pm.x - (int)(lane & 7u), by0 = pm.y - (int)(lane >> 3);
    const bool culled = bx0 + 8 <= fv.cull_x0 || bx0 >= fv.cull_x1 || by0 + 8 <= fv.cull_y0 || by0 >= fv.cull_y1;  // wave-uniform: this wave holds only misses
    RayState r;
    r.left = 0u; r.px = r.py = r.pz = 0.0f; r.sx = r.sy = r.sz = 0.0f;
    r.A = 0.0f; r.Gr = r.Gg = r.Gb = 0.0f; r.out = 0;
    bool hit = false;
    if (pm.valid && !culled) {
        // --- ray: SURVEY A.1 step 1, as raymarch_naive_kernel ---
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
        if (!(t0 > t1)) {  // :91-93
            t0 = fmaxf(t0, 0.0f);  // :94
            const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
            float dtx = 1.0f / (fnx * fabsf(dir[0]));
            float dty = 1.0f / (fny * fabsf(dir[1]));
            float dtz = 1.0f / (fnz * fabsf(dir[2]));
            const float dt = L.dt_scale * fminf(dtx, fminf(dty, dtz));  // :97-99
            r.left = min(count_trips(t0, t1, dt), 0x7fffffffu);  // :101
            r.px = eye[0] + t0 * dir[0]; r.py = eye[1] + t0 * dir[1]; r.pz = eye[2] + t0 * dir[2];  // :100
            r.sx = dir[0] * dt; r.sy = dir[1] * dt; r.sz = dir[2] * dt;  // :118
            hit = true;
        }
    }
    // The group's major axis and direction of travel, by majority of its rays: nine counts per wave, summed in LDS.
    extern __shared__ unsigned char stage_win[];
    int *const ex = reinterpret_cast<int *>(stage_win);
    if (threadIdx.x < 16u) ex[kGroupAccI + threadIdx.x] = 0;
    __syncthreads();
    {
        const float cs3[3] = {r.sx * (float)V.nx, r.sy * (float)V.ny, r.sz * (float)V.nz};  // cells per step, signed
        const float ax = fabsf(cs3[0]), ay = fabsf(cs3[1]), az = fabsf(cs3[2]);
        const int mj = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        const bool alive0 = hit && r.left != 0u;
        int cnt[9];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cnt[k] = __popcll(__ballot(hit && mj == k));
            cnt[3 + k] = __popcll(__ballot(alive0 && cs3[k] > 0.0f));
            cnt[6 + k] = __popcll(__ballot(alive0 && cs3[k] < 0.0f));
        }
        if (lane == 0u) {
#pragma unroll
            for (int k = 0; k < 9; k++) if (cnt[k]) atomicAdd(ex + kGroupAccI + k, cnt[k]);
        }
    }
    __syncthreads();
    const int c0 = __builtin_amdgcn_readfirstlane(ex[kGroupAccI + 0]), c1 = __builtin_amdgcn_readfirstlane(ex[kGroupAccI + 1]), c2 = __builtin_amdgcn_readfirstlane(ex[kGroupAccI + 2]);
    Census cs;
    if (c0 + c1 + c2 != 0) {  // group-uniform
        if (L.flags & LF_WAVE_PRIORITY) set_wave_priority(hit, r.left, fmaxf((float)V.nx, fmaxf((float)V.ny, (float)V.nz)) / L.dt_scale);
        const int major = (c0 >= c1 && c0 >= c2) ? 0 : (c1 >= c2 ? 1 : 2);
        const uint32_t copy = D.copy_of_major[major];
        // (the S axis of copy k is axis k: the direction of travel is counted on that axis)
        const bool dir_up = __builtin_amdgcn_readfirstlane(ex[kGroupAccI + 3 + copy]) >= __builtin_amdgcn_readfirstlane(ex[kGroupAccI + 6 + copy]);
        if (wave == 0u && lane < 16u) {  // both sets of the round accumulators at their neutral values
            const uint32_t k = lane & 7u;
            ex[kGroupAccB + lane] = (k == 0u || k > 4u) ? 0 : ((k & 1u) ? 0x7fffffff : (int)0x80000000);
            if (lane < 4u) ex[kGroupAccS + lane] = (lane & 1u) == 0u ? 0 : (dir_up ? 0x7fffffff : (int)0x80000000);
        }
        __syncthreads();
        if (copy == 0u) march_staged_group_perm<VOL, 0, COUNT>(V, D, r, hit, cs, lane, wave, dir_up);
        else if (copy == 1u) march_staged_group_perm<VOL, 1, COUNT>(V, D, r, hit, cs, lane, wave, dir_up);
        else march_staged_group_perm<VOL, 2, COUNT>(V, D, r, hit, cs, lane, wave, dir_up);
    }
    if (!pm.valid) return;
    float Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    if (hit) {
        Cr = linear_to_srgb(fmaf(0.5f, r.Gr, 0.5f * r.A));  // :121-123
        Cg = linear_to_srgb(fmaf(0.5f, r.Gg, 0.5f * r.A));
        Cb = linear_to_srgb(fmaf(0.5f, r.Gb, 0.5f * r.A));
    }
    store_out<OUT>(L, pm, Cr, Cg, Cb);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = (L.flags & LF_STEPS_ARE_FALLBACKS) ? cs.n_fb : ((L.flags & LF_STEPS_ARE_TRIPS) ? cs.n_look : cs.n_iter);
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)cs.n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)cs.n_samp);
            atomicAdd(&L.counters[2], (unsigned long long)cs.w_outer);   // rounds (per wave)
            atomicAdd(&L.counters[3], (unsigned long long)cs.w_inner);   // rounds served from global memory
            atomicAdd(&L.counters[4], (unsigned long long)cs.w_sample);  // sum of T over rounds
            atomicAdd(&L.counters[5], (unsigned long long)cs.n_look);
        }
    }
}

}  // namespace vk
