// vk_xor.hpp -- shaders/xor.wgsl:18-61 (hash / noise / fbm / noise_volume) with the SPECIFIED sine, shared by the xor volume
// generator (vk_volume_kernels.hpp) and the C3 procedural march (vk_compute.hpp).  Device functions only.
#pragma once

#include "vk_common.hpp"

namespace vk {

// ---- xor example's volume generator (next row N3): shaders/xor.wgsl:18-78 --------------------
// cs_main for every voxel: fbm value noise (3 octaves x 8 sin-hashes) and its finite-difference
// gradient (3 more evaluations).  ALU-bound: 96 hashes per voxel.  hash()'s sine is the specified
// one (f64 Cody-Waite + minimax polynomial, rounded once to f32), so the volume is reproducible.
// a * b + k with the (wave-uniform, loop-invariant) coefficient k read from a scalar register pair.  Written out
// because the compiler otherwise turns every Horner step into v_mov_b64 (copy the coefficient) + v_fmac_f64: 258 of
// the procedural loop's 1068 instructions were such copies.  Same single-rounding fma, bit for bit.
__device__ __forceinline__ double fma_k(double a, double b, double k) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

__device__ __forceinline__ float sin_spec(float h) {
    const double x = (double)h;
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    const double r2 = r * r;
    double sp = 1.58969099521155010221e-10;
    sp = fma_k(sp, r2, -2.50507602534068634195e-08);
    sp = fma_k(sp, r2, 2.75573137070700676789e-06);
    sp = fma_k(sp, r2, -1.98412698298579493134e-04);
    sp = fma_k(sp, r2, 8.33333333332248946124e-03);
    sp = fma_k(sp, r2, -1.66666666666666324348e-01);
    const double sn = fma(r * r2, sp, r);
    double cp = -1.13596475577881948265e-11;
    cp = fma_k(cp, r2, 2.08757232129817482790e-09);
    cp = fma_k(cp, r2, -2.75573143513906633035e-07);
    cp = fma_k(cp, r2, 2.48015872894767294178e-05);
    cp = fma_k(cp, r2, -1.38888888888741095749e-03);
    cp = fma_k(cp, r2, 4.16666666666666019037e-02);
    const double cs = fma(r2 * r2, cp, fma(-0.5, r2, 1.0));
    // Quadrant (|k| < 2^31 for every argument the hash makes): q = k & 3 -> sn, cs, -sn, -cs.  Written on the bit patterns -- odd q takes
    // the cosine series (v_bfi_b32 on both halves), bit 1 of q flips the sign bit -- because the compiler lowers the four-way select of
    // doubles to two nested exec-mask branches per sine (6 scalar instructions, 3 compares, 2 v_cndmask through VCC: about a third of a
    // sine's issue cycles by profiles/r03_ubench_valu_issue_rate.txt, 24 sines per step).  The same value bit for bit: a negated double
    // differs in its sign bit only.
    const uint32_t qi = (uint32_t)(int)k;
    union { double d; uint32_t u[2]; } S, C, R;
    S.d = sn; C.d = cs;
    const uint32_t odd = 0u - (qi & 1u);  // all ones: the cosine series
    uint32_t lo, hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(odd), "v"(C.u[0]), "v"(S.u[0]));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(odd), "v"(C.u[1]), "v"(S.u[1]));
    R.u[0] = lo;
    R.u[1] = hi ^ ((qi << 30) & 0x80000000u);
    return (float)R.d;
}
__device__ __forceinline__ float xor_fract(float x) { return x - floorf(x); }
__device__ __forceinline__ float xor_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
// The sine a GPU running xor.wgsl as it stands would use: hardware v_sin_f32 (argument in revolutions) behind a multiply by 1 / 2 pi and
// v_fract -- how the AMD compilers lower sin().  For the hash's arguments (up to ~8e5, where 1 / 2 pi of an ulp is ~1e-2 revolutions) it
// returns a few correct bits, and the hash multiplies what is left by 43758: DEV = true renders a DIFFERENT noise field of the same
// statistics.  A tolerance mode (VK_RENDER_DEVICE_SINE) for the procedural march only; the volume generator always takes the specified sine.
__device__ __forceinline__ float sin_device(float h) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(h * 0.15915494309189535f)); }
template <bool DEV>
__device__ __forceinline__ float xor_hash(float h) { return xor_fract((DEV ? sin_device(h) : sin_spec(h)) * 43758.5453123f); }
template <bool DEV>
__device__ __forceinline__ float xor_noise(float x0, float x1, float x2) {
    const float p0 = floorf(x0), p1 = floorf(x1), p2 = floorf(x2);
    float f0 = xor_fract(x0), f1 = xor_fract(x1), f2 = xor_fract(x2);
    f0 = f0 * f0 * (3.0f - 2.0f * f0); f1 = f1 * f1 * (3.0f - 2.0f * f1); f2 = f2 * f2 * (3.0f - 2.0f * f2);
    const float n = p0 + p1 * 157.0f + 113.0f * p2;
    return xor_mix(xor_mix(xor_mix(xor_hash<DEV>(n + 0.0f), xor_hash<DEV>(n + 1.0f), f0), xor_mix(xor_hash<DEV>(n + 157.0f), xor_hash<DEV>(n + 158.0f), f0), f1),
                   xor_mix(xor_mix(xor_hash<DEV>(n + 113.0f), xor_hash<DEV>(n + 114.0f), f0), xor_mix(xor_hash<DEV>(n + 270.0f), xor_hash<DEV>(n + 271.0f), f0), f1),
                   f2);
}
template <bool DEV>
__device__ __forceinline__ float xor_fbm(float p0, float p1, float p2) {
    float f = 0.5000f * xor_noise<DEV>(p0, p1, p2);
    p0 = p0 * 2.01f; p1 = p1 * 2.01f; p2 = p2 * 2.01f;
    f = f + 0.2500f * xor_noise<DEV>(p0, p1, p2);
    p0 = p0 * 2.02f; p1 = p1 * 2.02f; p2 = p2 * 2.02f;
    f = f + 0.1250f * xor_noise<DEV>(p0, p1, p2);
    return f;
}
template <bool DEV = false>
__device__ __forceinline__ void xor_noise_volume(float c0, float c1, float c2, float off1, float &val, float &alpha) {
    val = xor_fbm<DEV>((c0 + 1.0f) * 32.0f, (c1 + off1) * 32.0f, (c2 + 21.0f) * 32.0f);
    const float len = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
    alpha = val * smoothstepf(0.5f, 0.25f, len);
}

}  // namespace vk
