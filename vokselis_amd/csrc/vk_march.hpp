// vk_march.hpp -- NAIVE_TRILINEAR (raycast_naive.wgsl:83-125) on the cell layouts (and the LINEAR / 9^3 / quad comparison layouts).
// Included by vk_launch_cells.hip only (and, for RayState / Census / clear_inactive_strip, by vk_staged.hpp).
#pragma once

#include "vk_common.hpp"
#include "vk_trips.hpp"

namespace vk {

// ... and copied into LDS by every wave of the march: a handful of 16-byte loads instead of ~200 VALU
// instructions of index arithmetic per wave.
__device__ __forceinline__ void load_cell_luts(const VolumeDesc &V, uint32_t *lut, uint32_t lane) {
    const uint32_t n4 = cell_lut_entries(V.nx, V.ny, V.nz) >> 2;
    const uint4 *src = reinterpret_cast<const uint4 *>(V.lut);
    uint4 *dst = reinterpret_cast<uint4 *>(lut);
    for (uint32_t e = lane; e < n4; e += 64u) dst[e] = src[e];
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

// -k for a walk of k = ceil(r) steps, 1 <= k <= iterations left (nleft = -left < 0): floor(-r) in one conversion (it saturates), clamped in one v_med3
__device__ __forceinline__ int walk_steps_neg(float r, int nleft) {
    int k;
    asm("v_cvt_flr_i32_f32_e64 %0, -%1\n\tv_med3_i32 %0, %0, %2, -1" : "=&v"(k) : "v"(r), "v"(nleft));
    return k;
}

// ---- the march, resumable ---------------------------------------------------------------------
// Everything a ray needs to continue: the accumulators of the reference loop (p, alpha, colour sums), its per-ray constants and
// where its pixel goes.  64 bytes.
// The loop variable of raycast_naive.wgsl:101 is not among them.  t feeds nothing but its own test `t < t1`, so the march carries
// `left`, the number of iterations the loop has still to make -- count_trips() (vk_trips.hpp) computes the loop's trip count from
// (t0, t1, dt) exactly, rounding by rounding, once per ray (round 5).
struct RayState {
    uint32_t left;       // iterations of :101 not made yet
    float px, py, pz, sx, sy, sz, A, Gr, Gg, Gb;
    uint32_t out;        // pixel index into the launch's output
    uint32_t pad[4];
};
static_assert(sizeof(RayState) == 64, "RayState is one 64-byte record");

struct Census {  // SIMT execution census + step counters (COUNT builds only)
    uint32_t n_iter = 0, n_samp = 0, w_outer = 0, w_inner = 0, w_sample = 0, n_look = 0, n_fb = 0;
    uint32_t skips = 0;  // trips that skipped (every build: drives the adaptive probing policy)
    // per-trip log of the wave (COUNT builds, debug bit 6; docs/archive/tools/repack_census.py): entry = live lanes | samplers << 7 | samplers whose alpha is
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
//
// WALK: how a run of k exactly transparent steps advances the reference's position (p += s, k times: raycast_naive.wgsl:118; the loop
// variable of :101 is the integer `left`, see RayState: left -= k).
//   WALK_LOOP  the additions themselves, four steps per loop iteration, capped per trip: bit-exact, the default.
//   WALK_FMA   VK_RENDER_FAST_WALK, tolerance mode: one real addition, then k - 1 more steps as ONE fma with the rounded increment of
//              that addition, v_1 + m (v_1 - v_0).  Inside a binade every rounded addition of the same addend moves an accumulator
//              by the same amount, so this IS the reference's value unless the coordinate crosses a power of two during the walk;
//              there it parts from the reference by at most (steps after the crossing) x half an ulp (positions agree to ~2e-4 cell).
//              The number of iterations is the reference's whatever the walk does (`left` is exact), unless the alpha >= 0.95
//              early-out flips (profiles/r04_walk_modes.txt).  No cap: a walk of any length costs the same dozen instructions.
//   (Cutting the closed form at every binade boundary makes it exact again -- and 15 - 37 % slower than the loop: every crossing
//   costs the lane another probing trip.  docs/archive/experiments/skip_walk_binade_cut_closed_form.patch)
enum WalkKind : int { WALK_LOOP = 0, WALK_FMA = 2 };

//
// AHEAD (LF_PROBE_AHEAD; the fast path of the skip kernels): a trip needs two fetches one after the other -- the cell's distance byte, then, if it is 0,
// the cell -- and a heavy wave's chain of sampling trips pays both latencies per step.  With AHEAD the distance byte of the NEXT position is requested
// while this trip's sample is evaluated: a lane that samples advances p and t (the reference's additions, :118 -- neither depends on the sample)
// right after requesting its cell, locates the next position and requests its distance, then filters and composites.  One byte load is wasted
// when the ray ends with that sample.  Walkers locate their new position at the end of their walk, as they did at the top of the next trip before.
// Per ray the same operations in the same order on every variable.
template <int VOL, bool SKIP, bool SAFE, bool COUNT, bool BOUNDED = false, int WALK = WALK_LOOP, bool AHEAD = false>
__device__ __forceinline__ bool march(const VolumeDesc &V, RayState &r, const uint32_t budget, Census &cs,
                                      const uint32_t *lut = nullptr, const float walk_cap = __builtin_inff(), const float walk_cap_all = __builtin_inff()) {
    constexpr bool PACKED = (VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16);
    constexpr bool BRICK9 = (VOL == VOL_B9U8 || VOL == VOL_B9F16);
    static_assert(!AHEAD || (PACKED && SKIP && !SAFE && !BOUNDED), "probe-ahead: the skip kernels' fast path, unbounded");
    float px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    int nleft = -(int)r.left;  // minus the iterations left (counts up to 0: a walk's -k = floor(-r) comes out of one conversion)
    const float sx = r.sx, sy = r.sy, sz = r.sz;
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
        // (the walk counts its steps: no margin for a drifting loop variable; 2^-12 covers rcp and the fmas, 0.01 of a step on top)
        constexpr float sc = 1.0f - 0x1p-12f, cst = -0.01f;
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

    // One exit test per trip: `t < t1` (:101, here: iterations left) and the alpha early-out (:115-117) are folded into the
    // loop condition; p is dead after the break, so advancing it unconditionally (:118) changes nothing observable.
    uint32_t trip = 0;  // wave-uniform: the active lanes of a wave entered the loop together
    // AHEAD: where the ray stands, carried from trip to trip: lerp weights, cell index, and the cell's distance byte (possibly still in flight)
    float a_fx = 0.0f, a_fy = 0.0f, a_fz = 0.0f;
    uint32_t a_idx = 0, a_d = 0;
    auto locate = [&](float qx, float qy, float qz) {
        const float ux = fmaf(qx, fnx, -0.5f), uy = fmaf(qy, fny, -0.5f), uz = fmaf(qz, fnz, -0.5f);
        a_fx = __builtin_amdgcn_fractf(ux); a_fy = __builtin_amdgcn_fractf(uy); a_fz = __builtin_amdgcn_fractf(uz);
        a_idx = lut[cvt_floor_i32(ux) + 2] + luty[cvt_floor_i32(uy) + 2] + lutz[cvt_floor_i32(uz) + 2];
        // A request made for a ray that turns out to have ended looks at most one step past the box: <= dt_scale cells on any axis.  The tables carry
        // one clamped entry beyond i = -1 and i = n - 1, which covers 1.5 cells; the host sets LF_PROBE_AHEAD only for dt_scale <= 1.25
        // (vk_render.hip), so the index is always a real cell's and the map is read inside its allocation.  (No branch around the request, and a
        // plain global load: an exec-masked region, or a bounds-checked buffer load, gives most of the gain back.)
        a_d = V.dist[a_idx + doff];
    };
    if (AHEAD) locate(px, py, pz);
    while (nleft != 0 && A < 0.95f && (!BOUNDED || trip < budget)) {
        if (BOUNDED) ++trip;
        if (COUNT) { n_look++; if (wave_leader()) w_outer++; }
        uint32_t *le = nullptr;
        if (COUNT && cs.log) {
            if (cs.trip_no < cs.log_cap) { le = cs.log + cs.trip_no; atomicAdd(le, 1u); }
            cs.trip_no++;
        }
        const float ux = AHEAD ? 0.0f : fmaf(px, fnx, -0.5f), uy = AHEAD ? 0.0f : fmaf(py, fny, -0.5f), uz = AHEAD ? 0.0f : fmaf(pz, fnz, -0.5f);
        int ix = AHEAD ? 0 : cvt_floor_i32(ux), iy = AHEAD ? 0 : cvt_floor_i32(uy), iz = AHEAD ? 0 : cvt_floor_i32(uz);
        float fx = AHEAD ? a_fx : __builtin_amdgcn_fractf(ux), fy = AHEAD ? a_fy : __builtin_amdgcn_fractf(uy), fz = AHEAD ? a_fz : __builtin_amdgcn_fractf(uz);
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
            } else if (AHEAD) {
                coff = (uint32_t)(a_idx << V.sh_x);
                d = a_d;
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
            uint32_t cap_now;  // (s_cmp + s_cselect: written as a ternary the compiler builds two branches and five moves around it; C2 batches -0.6 %)
            asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(cap_now) : "s"(samplers), "s"(__builtin_amdgcn_readfirstlane(__float_as_uint(walk_cap))), "s"(__builtin_amdgcn_readfirstlane(__float_as_uint(walk_cap_all))) : "scc");
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
                // Samples j = 0 .. k - 1 are skipped, k = ceil(r) (j < r keeps a sample inside the empty range; the current sample, j = 0,
                // sits in an empty cell: k >= 1), at most the trip's cap and at most the iterations the loop has left -- every skipped
                // iteration is one the reference makes (it passes `t < t1` on the reference's own t: that is what `left` counts).
                // The walk advances the reference's position, p += s, k times: the same f32 additions in the same order.
                float rmin = fminf(fminf(rx, ry), rz);
                if constexpr (WALK == WALK_FMA) {
                    // one real addition, the other k - 1 as one fma with that addition's rounded increment (tolerance: see WalkKind)
                    const float x_1 = px + sx, y_1 = py + sy, z_1 = pz + sz;
                    const int kneg = walk_steps_neg(rmin, nleft);  // -k
                    const float m = (float)(-1 - kneg);
                    const float dx_q = x_1 - px, dy_q = y_1 - py, dz_q = z_1 - pz;
                    px = fmaf(m, dx_q, x_1); py = fmaf(m, dy_q, y_1); pz = fmaf(m, dz_q, z_1);
                    nleft -= kneg;
                    if (COUNT) { n_iter += (uint32_t)(-kneg); if (wave_leader()) { w_inner++; if (le) atomicAdd(le, 1u << 21); } }
                    if (AHEAD) locate(px, py, pz);
                    continue;
                }
                asm("v_min_f32 %0, %1, %2" : "=v"(rmin) : "s"(cap_now), "v"(rmin));  // (a known-quiet scalar: no canonicalise)
                // m = k - 1: the steps after the first.  Single-frame launches (AHEAD) round it down to even -- the odd step is left to the next
                // trip's probe (any stop is exact) and the walk loses a branch: the lone chains of a single frame pay per instruction
                // (C2 single frame -3.7 %; batches +2 %: they keep the odd step)
                const uint32_t m = ~(uint32_t)walk_steps_neg(rmin, nleft) & (AHEAD ? ~1u : ~0u);
                const int c = (int)~m;  // -k
                nleft -= c;
                if (COUNT) { n_iter += (uint32_t)(-c); if (wave_leader()) { w_inner++; if (le) atomicAdd(le, 1u << 21); } }
                px = px + sx; py = py + sy; pz = pz + sz;
                // four skipped iterations per trip of the walk (the counter's decrement is its own test: v_sub_co)
                for (uint32_t q = m >> 2; !__builtin_usub_overflow(q, 1u, &q);) {
#pragma unroll
                    for (int j = 0; j < 4; j++) { px = px + sx; py = py + sy; pz = pz + sz; }
                    if (COUNT) { if (wave_leader()) { w_inner++; if (le) atomicAdd(le, 1u << 21); } }
                }
                if (!AHEAD && (m & 1u)) {
                    asm volatile("" : "+v"(px));
                    px = px + sx; py = py + sy; pz = pz + sz;
                }
                if (m & 2u) {
                    // (the empty asm keeps this an exec-masked region: if-converted, the two steps are computed for every lane and
                    // then selected by v_cndmask through VCC in a row, ~16 issue cycles each -- profiles/r03_ubench_valu_issue_rate.txt --
                    // twice the cost of a whole four-step walk iteration)
                    asm volatile("" : "+v"(px));
                    px = px + sx; py = py + sy; pz = pz + sz;
                    px = px + sx; py = py + sy; pz = pz + sz;
                }
                if (AHEAD) locate(px, py, pz);
                continue;
            }
            CellBits<VOL> cb;
            if (SAFE) {
                if constexpr (VOL == VOL_P8) { const uint2 c = *reinterpret_cast<const uint2 *>(cptr); cb.v.x = c.x; cb.v.y = c.y; }
                else { const uint4 c = *reinterpret_cast<const uint4 *>(cptr); cb.v.x = c.x; cb.v.y = c.y; cb.v.z = c.z; cb.v.w = c.w; }
            } else {
                cb = load_cell<VOL>(cells, coff);
            }
            if (AHEAD) {
                // :118 and :101's increment (one iteration fewer left) now -- neither depends on the sample -- then the next position's distance byte is requested
                // under this sample's arithmetic (fx, fy, fz keep THIS position's weights)
                px = px + sx; py = py + sy; pz = pz + sz;
                nleft += 1;
                locate(px, py, pz);
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
                if (!AHEAD) {
                    px = px + sx; py = py + sy; pz = pz + sz;  // :118
                    nleft += 1;
                }
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
        if (!AHEAD) {
            px = px + sx; py = py + sy; pz = pz + sz;  // :118
            nleft += 1;
        }
    }
    if (AHEAD) asm volatile("" ::"v"(a_d));  // (the last request is consumed on the exit path too)
    r.left = (uint32_t)(-nleft); r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
    return nleft != 0 && A < 0.95f;
}

// The same loop for the fast path without skipping (every trip samples), software-pipelined: the
// position is advanced first and the NEXT trip's cell is requested before this trip's sample is
// evaluated, so the fetch latency overlaps the ~40 VALU instructions of a sample instead of adding
// to them -- it is the lone heavy waves at the tail of a frame that set the frame time.  The f32
// operations on p, A and the colour sums are those of march(), in the same order per variable.
// The request one step past the ray's end reads a real (clamped) table entry and is never used.
// CELL_LUT: the tables hold cell indices (the skip kernels' copy) instead of byte offsets; `budget` bounds the trips
// (0xffffffff: none) so that the skip kernels can run stretches of it between probing windows.
template <int VOL, bool COUNT, bool CELL_LUT = false>
__device__ __forceinline__ bool march_stream(const VolumeDesc &V, RayState &r, Census &cs, const uint32_t *lut, uint32_t budget = 0xffffffffu) {
    float px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    uint32_t left = r.left;
    const float sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const uint32_t *luty = lut + (V.nx + 3), *lutz = lut + (V.nx + V.ny + 6);
    const __amdgpu_buffer_rsrc_t cells = cell_buffer(V.data, (uint32_t)V.max_off + (1u << V.sh_x));
    if (!(left != 0u && A < 0.95f)) return false;
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
        left -= 1u;  // :101
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return left != 0u && A < 0.95f;
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
    r.left = left; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
    return alive;
}

// The dense 9^3-brick layouts, software-pipelined the same way: these serve volumes far larger than
// the caches, where every step's four x-pair loads are HBM/fabric latency.  The next trip's loads are
// requested (clamped indices, so always inside the array) before this trip's sample is evaluated.
template <int VOL, bool COUNT>
__device__ __forceinline__ void march_b9_stream(const VolumeDesc &V, RayState &r, Census &cs) {
    static_assert(VOL == VOL_B9U8 || VOL == VOL_B9F16, "9^3 brick layouts");
    float px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    uint32_t left = r.left;
    const float sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const int mx = (int)V.nx - 1, my = (int)V.ny - 1, mz = (int)V.nz - 1;
    if (!(left != 0u && A < 0.95f)) return;
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
        left -= 1u;  // :101
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return left != 0u && A < 0.95f;
    };
    for (;;) {
        if (!trip(c0, c1)) break;
        if (!trip(c1, c0)) break;
    }
    asm volatile("" ::"v"(c0.p00), "v"(c0.p10), "v"(c0.p01), "v"(c0.p11), "v"(c1.p00), "v"(c1.p10), "v"(c1.p01), "v"(c1.p11));
    r.left = left; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
}

// The quad layouts: one load per sample (two consecutive elements), software-pipelined like the others.
template <int VOL, bool COUNT>
__device__ __forceinline__ void march_quads_stream(const VolumeDesc &V, RayState &r, Census &cs) {
    static_assert(VOL == VOL_Q8 || VOL == VOL_QF16, "quad layouts");
    float px = r.px, py = r.py, pz = r.pz, A = r.A, Gr = r.Gr, Gg = r.Gg, Gb = r.Gb;
    uint32_t left = r.left;
    const float sx = r.sx, sy = r.sy, sz = r.sz;
    const float fnx = (float)V.nx, fny = (float)V.ny, fnz = (float)V.nz;
    const int mx = (int)V.nx - 1, my = (int)V.ny - 1, mz = (int)V.nz - 1;
    if (!(left != 0u && A < 0.95f)) return;
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
        left -= 1u;  // :101
        fx = __builtin_amdgcn_fractf(ux); fy = __builtin_amdgcn_fractf(uy); fz = __builtin_amdgcn_fractf(uz);
        return left != 0u && A < 0.95f;
    };
    for (;;) {
        if (!trip(c0, c1)) break;
        if (!trip(c1, c0)) break;
    }
    asm volatile("" ::"v"(c0.a), "v"(c0.b), "v"(c0.c), "v"(c0.d), "v"(c1.a), "v"(c1.b), "v"(c1.c), "v"(c1.d));
    r.left = left; r.px = px; r.py = py; r.pz = pz; r.A = A; r.Gr = Gr; r.Gg = Gg; r.Gb = Gb;
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

// AHEAD: the probe-ahead trip (march<..., AHEAD>), an instantiation of its own -- it needs six more registers, and the launches that fill the machine keep the leaner kernel
template <int VOL, bool SKIP, bool SAFE, int WALK, bool AHEAD, int OUT, bool COUNT>
__global__ __launch_bounds__(64) void raymarch_naive_kernel(const LaunchDesc L, const VolumeDesc V) {
    static_assert(VOL == VOL_P8 || VOL == VOL_P16 || VOL == VOL_PF16 || (!SKIP && SAFE), "linear / bricked layouts: no skip map, clamped indices");
    static_assert(SKIP || WALK == WALK_LOOP, "the closed-form walks are variants of the skip kernels");
    static_assert(!AHEAD || (SKIP && !SAFE), "probe ahead: the skip kernels' fast path");
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
    const bool trip_log = COUNT && L.trace && (L.flags & LF_TRIP_LOG);
    if (trip_log) {
        cs.log_cap = L.trip_log_cap;
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
        r.left = min(count_trips(t0, t1, dt), 0x7fffffffu);  // :101
        r.px = px; r.py = py; r.pz = pz; r.sx = sx; r.sy = sy; r.sz = sz;
        r.A = 0.0f; r.Gr = 0.0f; r.Gg = 0.0f; r.Gb = 0.0f;  // colour sums: G = sum w*cos(phase); C = A/2 + G/2 (sum w == A)
        r.out = (uint32_t)pm.out_index;
        // (not in the skip kernels: a ray's nominal length says little about its work there -- C2 at 64 orbit frames per launch 0.06509 -> 0.06467 ms without)
        if (!SKIP && (L.flags & LF_WAVE_PRIORITY)) set_wave_priority(true, r.left, fmaxf(fnx, fmaxf(fny, fnz)) / L.dt_scale);
        if constexpr (USE_LUT && !SKIP) march_stream<VOL, COUNT>(V, r, cs, cell_lut);
        else if constexpr (SKIP) {
            if (L.flags & LF_ADAPTIVE_PROBING) {
                // Adaptive probing (wave-uniform policy, any policy is exact: a sampled empty cell adds +0).  Probe for a
                // window of 16 trips; if fewer than 1 in 8 of the wave's live rays skipped anything in it, the wave is in
                // material that cannot be skipped: run the dense loop -- no distance look-up, and on the fast path
                // software-pipelined -- for a stretch that doubles every time the next window confirms it (64 .. 512
                // trips), then probe again.  Fog pays ~9 % of its trips at the probing price instead of all of them.
                const uint32_t stretch0 = (L.flags & LF_LONG_STRETCHES) ? 256u : 64u;  // the census found (almost) nothing to skip
                uint32_t stretch = stretch0;
                for (;;) {
                    cs.skips = 0;
                    bool alive = march<VOL, true, SAFE, COUNT, true, WALK>(V, r, 16u, cs, USE_LUT ? cell_lut : nullptr, L.walk_cap, L.walk_cap_all);
                    const unsigned long long live = __ballot(alive);
                    if (live == 0ull) break;
                    if (__popcll(__ballot(alive && cs.skips != 0u)) * 8 >= __popcll(live)) { stretch = stretch0; continue; }
                    if constexpr (USE_LUT) alive = march_stream<VOL, COUNT, true>(V, r, cs, cell_lut, stretch);
                    else alive = march<VOL, false, SAFE, COUNT, true>(V, r, stretch, cs, nullptr);
                    if (__ballot(alive) == 0ull) break;
                    stretch = min(stretch * 2u, 512u);
                }
            } else {
                march<VOL, SKIP, SAFE, COUNT, false, WALK, AHEAD>(V, r, 0xffffffffu, cs, USE_LUT ? cell_lut : nullptr, L.walk_cap, L.walk_cap_all);
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
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = (L.flags & LF_STEPS_ARE_TRIPS) ? cs.n_look : cs.n_iter;
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

}  // namespace vk
