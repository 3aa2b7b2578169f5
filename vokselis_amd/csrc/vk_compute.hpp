// vk_compute.hpp -- COMPUTE_NEAREST (raycast_compute.wgsl:62-144) on the bricked record layout and on the two dense volumes,
// and PROCEDURAL.  Included by vk_launch_compute.hip only.
#pragma once

#include "vk_common.hpp"
#include "vk_xor.hpp"

namespace vk {

// Same arithmetic as raymarch_compute_kernel below (raycast_compute.wgsl:62-131), one 16-byte record
// per step, software-pipelined: p = eye + t*dir does not depend on the loads, so the next step's
// record is requested before this step is shaded.  A lone wave of this mode used to pay a full
// cache-miss latency per step (<= 346 dependent steps per ray).
//
// SKIP (round 4): exact empty-space skipping, as the cell march has it.  A record whose opacity smoothstep(0, 0.7, a^3) is exactly 0 gives
// w = 0: every accumulator takes +0 (the max / min of the shading launder a NaN normal, so the product is 0 x finite), A does not move, and
// the loop's only other state is t.  Such a step need not be shaded -- provided t still takes the same sequence of rounded additions (:69) and
// every skipped iteration passed the reference's own `t < t1` on the very same t.  The record carries, in the half get_col2 never reads
// (normals.w), its Chebyshev distance d in voxels to the nearest record that can contribute: the samples of the next
// m = 1 + floor((d - 2) / (voxels per step)) iterations truncate to voxels inside that empty range, so they are walked (one addition each)
// instead of fetched and shaded.  The xor example's blob fills half of its cube and a ray leaves it by the opacity early-out or crosses
// empty space before and after it.  Frames and per-pixel iteration counts do not change by a bit (tests: SKIP == !SKIP == the literal twin).
template <int OUT, bool COUNT, bool SKIP = true, int RING = 4, int REV = 1>
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
    uint32_t n_iter = 0, n_samp = 0;
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
        // steps per voxel of Chebyshev distance: after i steps a sample's voxel differs from this one's by at most floor(i * u) + 1 on every axis
        // (u = voxels per step on the fastest axis; the + 1 is the truncation), which stays inside the empty range d - 1 for i <= (d - 2) / u;
        // 0.01 voxel covers the rounding of p = eye + t * dir (~1e-5 voxel).  rcp: 1 ulp, far inside that margin.
        const float inv_u = SKIP ? __builtin_amdgcn_rcpf(dt * fmaxf(fabsf(dir[0]) * hbx, fmaxf(fabsf(dir[1]) * hby, fabsf(dir[2]) * hbz))) : 0.0f;
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
        // The march alternates between two wave-level phases.
        //  SHADE: RING request buffers in a ring, RING - 1 steps in flight while one is shaded (round 4; one step ahead until then): a step's
        //   record is a miss to the Infinity Cache or HBM more often than not (268 MB of records, a new 4^3 brick every third step), ~1-2 us
        //   against the ~0.3 us a lone wave needs to shade a step -- and a frame of this mode lasts as long as its longest rays.  p = eye +
        //   t * dir does not depend on the loads and t takes the reference's own additions, so the steps ahead are known.  Unrolled over
        //   the ring (no register copies).  Nothing in this loop walks: a request that a walk might have rewritten would make the compiler
        //   wait for the newest load at every use, and the ring would hide nothing (measured: 123 us against 81 for the 720p frame).  A
        //   lane whose record cannot contribute shades it all the same -- it adds +0 -- until EVERY live lane of the wave is at least
        //   `walk_min` steps from anything that can contribute; then the wave leaves the loop.
        //  WALK: every live lane hops over its run of records that cannot contribute -- the reference's own additions on t, one exposed
        //   fetch per hop to learn the next distance -- until it meets a record that can (or is about to), or its ray ends; lanes that are
        //   there already wait.  A ray is [empty][blob][empty] more often than not: three phases per wave.
        // (>= 2: the wave leaves SHADE only when every live lane will take at least one hop in WALK, and WALK hands a lane back only when it
        // would not -- each phase makes progress, whatever the knob says)
        const float walk_min = fmaxf((float)L.pair_walk_min, 2.0f);
        float t = t0;
        bool alive = true;
        Req q = request(t);  // the record of t
        for (;;) {
            if (SKIP) {  // ---- WALK
                while (alive) {
                    const uint32_t dvox = q.r.w >> 16;
                    const float m = __builtin_floorf(fmaf((float)dvox, inv_u, -2.01f * inv_u)) + 1.0f;  // iterations, this one included, whose samples cannot contribute
                    if (dvox == 0u || m < 2.0f) break;
                    float c = 0.0f;
                    do { t = t + dt; c += 1.0f; } while (c < m && t < t1);
                    n_iter += (uint32_t)c;
                    if (!(t < t1)) { alive = false; break; }
                    q = request(t);
                }
            }
            if (__ballot(alive) == 0ull) break;  // wave-uniform
            // ---- SHADE
            Req ring[RING];
            ring[0] = q;
            float tl = t;  // the t of the last step requested
#pragma unroll
            for (int k = 1; k < RING - 1; k++) { tl = tl + dt; ring[k] = request(fminf(tl, t1)); }
            ring[RING - 1] = ring[RING - 2];
            // The loop body is RING iterations of :69 in a straight line -- no exit between them, no branch on `alive` (selects), every lane
            // requests, live or not (a request past the ray's end is never shaded: it looks at the exit point, inside the tables).  Only so can the
            // compiler count the loads in flight at each use: an exit from the middle of the body is routed through the loop's latch (the
            // structurizer's single-exit form), which puts a path from the NEWEST request to the top of the loop, and a branch around a request
            // makes their number unknown -- either way it waits for every load in flight before every other step (s_waitcnt vmcnt(0)).  As it
            // is, the compiler still drains the ring once per loop iteration (it copies the buffers at the top of the loop), hence REV: the
            // iteration is REV revolutions of the ring.  The wave decides once per iteration whether to leave; the up to RING * REV - 1 steps it
            // shades beyond that add +0.
            auto leave = [&](const Req &cur) -> bool {
                if (SKIP) {
                    const uint32_t dvox = cur.r.w >> 16;  // (an out-of-range load returns zeros: distance 0 -- it is shaded and adds +0)
                    const bool deep = fmaf((float)dvox, inv_u, -2.01f * inv_u) + 1.0f >= walk_min;
                    return __ballot(alive && !deep) == 0ull;  // every live lane may walk (or none is left)
                }
                return __ballot(alive) == 0ull;
            };
            // one iteration of :69 on `cur` (the record of t), requesting into `tgt`
            auto trip = [&](Req &cur, Req &tgt) {
                tl = tl + dt;
                tgt = request(fminf(tl, t1));
                {
                    const float px = cur.px, py = cur.py, pz = cur.pz;
                    const uint32_t d0 = cur.r.x, d1 = cur.r.y, m0 = cur.r.z, m1 = cur.r.w;
                    float vc0 = h2f(d0 & 0xffffu), vc1 = h2f(d0 >> 16), vc2 = h2f(d1 & 0xffffu), vc3 = h2f(d1 >> 16);
                    float n0 = h2f(m0 & 0xffffu), n1 = h2f(m0 >> 16), n2 = h2f(m1 & 0xffffu);
                    n_iter += alive ? 1u : 0u; n_samp += alive ? 1u : 0u;
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
                    const float c0n = C[0] + w * col0 * sh0, c1n = C[1] + w * col1 * sh1, c2n = C[2] + w * col2 * sh2, An = A + w, tn = t + dt;
                    C[0] = alive ? c0n : C[0]; C[1] = alive ? c1n : C[1]; C[2] = alive ? c2n : C[2];
                    A = alive ? An : A;
                    const bool on = alive && !(An >= 0.95f);  // :93-95, then :69's increment and test
                    t = on ? tn : t;
                    alive = on && tn < t1;
                }
            };
            while (!leave(ring[0])) {  // :69
#pragma unroll
                for (int k = 0; k < RING * REV; k++) trip(ring[k % RING], ring[(k + RING - 1) % RING]);
            }
            q = ring[0];
#pragma unroll
            for (int k = 0; k < RING; k++) asm volatile("" ::"v"(ring[k].r));  // the last requests are consumed on the exit path too (keeps them ahead of the shading)
            if (!SKIP || __ballot(alive) == 0ull) break;
        }
    }
    store_out<OUT>(L, pm, C[0], C[1], C[2]);
    if (COUNT) {
        if (L.steps) L.steps[(size_t)pm.y * L.W + (size_t)pm.x] = n_iter;
        if (L.counters) {
            atomicAdd(&L.counters[0], (unsigned long long)n_iter);
            atomicAdd(&L.counters[1], (unsigned long long)n_samp);
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

// ---- PROCEDURAL: the compute twin's ray and march with the texel loads replaced by the xor
// example's density function at the sample position, noise_volume(p / 2) (shaders/xor.wgsl:55-61), colour =
// density.rgb / 2, no normals -- mirrors pixel_procedural of the oracle operation for operation.  No volume,
// no loads: 24 specified sines (f64 Cody-Waite, ~45 f64 operations each) and ~200 f32 flops per step.
// DEVSIN: the hash's sine is the hardware's (VK_RENDER_DEVICE_SINE, a tolerance mode: vk_xor.hpp) instead of the specified one.
template <int OUT, bool COUNT, bool DEVSIN = false>
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
            xor_noise_volume<DEVSIN>(px * 0.5f, py * 0.5f, pz * 0.5f, off1, val, alpha);
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

}  // namespace vk
