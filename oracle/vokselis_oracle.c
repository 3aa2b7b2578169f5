/*
 * vokselis_oracle.c -- CPU restatement of the vokselis raycast hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see vokselis_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is part of the specification: the only fused operations are the
 * fmaf() calls written below.
 */
#include "vokselis_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small vector helpers (explicit op order; no contraction)                   */

static inline float smoothstepf(float e0, float e1, float x);
static inline float vmin(float a, float b) { return fminf(a, b); }
static inline float vmax(float a, float b) { return fmaxf(a, b); }

/* ------------------------------------------------------------------------- */
/* f16 <-> f32                                                                */

float vo_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t vo_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u | ((ax >> 13) & 0x3ffu) : 0));
    }
    if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) { /* subnormal or zero in f16 */
        if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
        uint32_t e = ax >> 23;
        uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126 - e; /* 14..24 */
        uint32_t half = 1u << (shift - 1);
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t e = (ax >> 23) - 112;
    uint32_t m = ax & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}

void vo_rgba32f_to_rgba16f(const float *src, uint16_t *dst, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = vo_f32_to_f16(src[i]);
}

/* ------------------------------------------------------------------------- */
/* Camera (src/camera.rs:87-171) + glam 0.20.5            */

void vo_camera_eye(float zoom, float pitch, float yaw, const float target[3], float eye[3]) {
    /* fix_eye, src/camera.rs:148-157 */
    float pitch_cos = cosf(pitch);
    float vx = sinf(yaw) * pitch_cos;
    float vy = sinf(pitch);
    float vz = cosf(yaw) * pitch_cos;
    eye[0] = target[0] - zoom * vx;
    eye[1] = target[1] - zoom * vy;
    eye[2] = target[2] - zoom * vz;
}

static void mat4_mul(const float a[16], const float b[16], float out[16]) {
    /* glam Mat4 * Mat4: each result column = ((a.c0*b.x + a.c1*b.y) + a.c2*b.z) + a.c3*b.w */
    for (int c = 0; c < 4; c++) {
        for (int r = 0; r < 4; r++) {
            float s = a[0 * 4 + r] * b[c * 4 + 0];
            s = s + a[1 * 4 + r] * b[c * 4 + 1];
            s = s + a[2 * 4 + r] * b[c * 4 + 2];
            s = s + a[3 * 4 + r] * b[c * 4 + 3];
            out[c * 4 + r] = s;
        }
    }
}

static void mat4_inverse(const float m[16], float inv[16]) {
    /* general cofactor inverse, f32 (glam's SIMD ordering is not reproducible offline;
     * fixtures store the blob). */
    float a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    float rdet = 1.0f / det;
    for (int i = 0; i < 16; i++) inv[i] = a[i] * rdet;
}

static void normalize3(float v[3]) {
    float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] = v[0] / len;
    v[1] = v[1] / len;
    v[2] = v[2] / len;
}

static void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void vo_camera_uniform_build(float zoom, float pitch, float yaw, const float target[3], float aspect,
                             vo_camera_uniform *out) {
    float eye[3];
    vo_camera_eye(zoom, pitch, yaw, target, eye);
    /* Mat4::look_at_rh(eye, target, +Y), src/camera.rs:110 */
    float f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
    normalize3(f);
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float s[3], u[3];
    cross3(f, up, s);
    normalize3(s);
    cross3(s, f, u);
    float view[16] = {s[0], u[0], -f[0], 0.0f, s[1], u[1], -f[1], 0.0f, s[2], u[2], -f[2], 0.0f,
                      -dot3(s, eye), -dot3(u, eye), dot3(f, eye), 1.0f};
    /* Mat4::perspective_rh(PI/2, aspect, 0.1, 100), src/camera.rs:88-90,111 */
    const float fovy = 3.14159265358979323846f / 2.0f, zn = 0.1f, zf = 100.0f;
    float sn = sinf(0.5f * fovy), cs = cosf(0.5f * fovy);
    float h = cs / sn;
    float w = h / aspect;
    float r = zf / (zn - zf);
    float proj[16] = {w, 0, 0, 0, 0, h, 0, 0, 0, 0, r, -1.0f, 0, 0, r * zn, 0};
    out->view_position[0] = eye[0];
    out->view_position[1] = eye[1];
    out->view_position[2] = eye[2];
    out->view_position[3] = 1.0f;
    mat4_mul(proj, view, out->proj_view); /* src/camera.rs:112 */
    mat4_inverse(out->proj_view, out->inv_proj); /* src/camera.rs:169 */
}

/* ------------------------------------------------------------------------- */
/* intersect_box                                                              */

void vo_intersect_box(const float orig[3], const float dir[3], float lo, float hi, float t01[2]) {
    float tmin[3], tmax[3];
    for (int i = 0; i < 3; i++) {
        float inv = 1.0f / dir[i];
        float a = (lo - orig[i]) * inv;
        float b = (hi - orig[i]) * inv;
        tmin[i] = vmin(a, b);
        tmax[i] = vmax(a, b);
    }
    t01[0] = vmax(tmin[0], vmax(tmin[1], tmin[2]));
    t01[1] = vmin(tmax[0], vmin(tmax[1], tmax[2]));
}

/* ------------------------------------------------------------------------- */
/* Trilinear sample                                                           */

static inline float lerp_fma(float a, float b, float f) { return fmaf(f, b - a, a); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

float vo_sample_trilinear(const void *vol, uint32_t nx, uint32_t ny, uint32_t nz, int format,
                          const float p[3], int flags, int *any_nonempty) {
    /* u = p*n - 0.5 as ONE fused op (texel-space coordinate of the sample); the literal reading multiplies, then subtracts */
    const int literal = (flags & VO_FLAG_LITERAL_WGSL) != 0;
    float ux = literal ? p[0] * (float)nx - 0.5f : fmaf(p[0], (float)nx, -0.5f);
    float uy = literal ? p[1] * (float)ny - 0.5f : fmaf(p[1], (float)ny, -0.5f);
    float uz = literal ? p[2] * (float)nz - 0.5f : fmaf(p[2], (float)nz, -0.5f);
    float flx = floorf(ux), fly = floorf(uy), flz = floorf(uz);
    /* fractional weights: fract(u) = min(u - floor(u), 1 - 2^-24), the GPU fract semantic
     * (bit-identical to gfx950 v_fract_f32; checked exhaustively by tools/ubench/semantics.hip) */
    float fx = ux - flx, fy = uy - fly, fz = uz - flz;
    if (fx >= 1.0f) fx = 0x1.fffffep-1f;
    if (fy >= 1.0f) fy = 0x1.fffffep-1f;
    if (fz >= 1.0f) fz = 0x1.fffffep-1f;
    /* the int conversion saturates like v_cvt_i32_f32; NaN -> 0 */
    int ix = (flx != flx) ? 0 : (flx < -2147483648.0f ? INT32_MIN : (flx >= 2147483648.0f ? INT32_MAX : (int)flx));
    int iy = (fly != fly) ? 0 : (fly < -2147483648.0f ? INT32_MIN : (fly >= 2147483648.0f ? INT32_MAX : (int)fly));
    int iz = (flz != flz) ? 0 : (flz < -2147483648.0f ? INT32_MIN : (flz >= 2147483648.0f ? INT32_MAX : (int)flz));
    /* ClampToEdge (wgpu SamplerDescriptor::default address modes, volume_texture.rs:61-66) */
    int x0 = clampi(ix, 0, (int)nx - 1), x1 = clampi(ix < INT32_MAX ? ix + 1 : ix, 0, (int)nx - 1);
    int y0 = clampi(iy, 0, (int)ny - 1), y1 = clampi(iy < INT32_MAX ? iy + 1 : iy, 0, (int)ny - 1);
    int z0 = clampi(iz, 0, (int)nz - 1), z1 = clampi(iz < INT32_MAX ? iz + 1 : iz, 0, (int)nz - 1);
    size_t sx = 1, sy = nx, sz = (size_t)nx * ny; /* x fastest: volume_texture.rs:50-59 */
    size_t idx[8] = {x0 * sx + y0 * sy + z0 * sz, x1 * sx + y0 * sy + z0 * sz, x0 * sx + y1 * sy + z0 * sz,
                     x1 * sx + y1 * sy + z0 * sz, x0 * sx + y0 * sy + z1 * sz, x1 * sx + y0 * sy + z1 * sz,
                     x0 * sx + y1 * sy + z1 * sz, x1 * sx + y1 * sy + z1 * sz};
    float t[8];
    int nonempty = 0;
    if (format == VO_FMT_R8_UNORM) {
        const uint8_t *v = (const uint8_t *)vol;
        for (int k = 0; k < 8; k++) {
            uint8_t c = v[idx[k]];
            nonempty |= (c > 25);
            t[k] = (flags & (VO_FLAG_TAPNORM_PER_TAP | VO_FLAG_LITERAL_WGSL)) ? (float)c / 255.0f : (float)c;
        }
    } else {
        const uint16_t *v = (const uint16_t *)vol;
        for (int k = 0; k < 8; k++) {
            t[k] = vo_f16_to_f32(v[idx[k]]);
            nonempty |= (t[k] > 0.1f);
        }
    }
    if (any_nonempty) *any_nonempty = nonempty;
    if (literal) {
        /* no fused operation anywhere: a + f * (b - a), product and sum rounded separately; R8Unorm taps were normalised
         * one by one above (c / 255 as a true divide), so nothing is left to scale */
        float c00 = t[0] + fx * (t[1] - t[0]), c10 = t[2] + fx * (t[3] - t[2]);
        float c01 = t[4] + fx * (t[5] - t[4]), c11 = t[6] + fx * (t[7] - t[6]);
        float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
        return c0 + fz * (c1 - c0);
    }
    /* lerp x, then y, then z; lerp(a,b,f) = fma(f, b-a, a) */
    float c00 = lerp_fma(t[0], t[1], fx), c10 = lerp_fma(t[2], t[3], fx);
    float c01 = lerp_fma(t[4], t[5], fx), c11 = lerp_fma(t[6], t[7], fx);
    float c0 = lerp_fma(c00, c10, fy), c1 = lerp_fma(c01, c11, fy);
    float r = lerp_fma(c0, c1, fz);
    if (format == VO_FMT_R8_UNORM && !(flags & VO_FLAG_TAPNORM_PER_TAP) && !(flags & VO_FLAG_RAW_UNORM8)) {
        /* R8Unorm normalisation applied once after filtering (linear => same value in exact
         * arithmetic as normalising each tap); 1/255 rounded to f32.  The march itself asks for the raw
         * (0..255 scale) value, VO_FLAG_RAW_UNORM8: its transfer function carries the 1/255 (vo_transfer_alpha). */
        r = r * (1.0f / 255.0f);
    }
    return r;
}

/* ------------------------------------------------------------------------- */
/* Transfer function + compositing pieces                                     */

float vo_transfer_alpha(float x, int raw_unorm8) {
    /* raycast_naive.wgsl:106: clamp(vec3(0.4), vec3(.9), val) == min(max(0.4,0.9), val) (F8)
     * raycast_naive.wgsl:107: smoothstep(0.10, 1.2, v) = t*t*(3 - 2t), t = clamp((v - 0.1) / 1.1).
     * Specification (round 2): the affine map of t is ONE fused op, t = clamp(fma(min(x, c), k1, k2)), with the scale
     * of the sample folded into its constants -- x is either a value (f16 volumes, per-tap-normalised taps: c = 0.9,
     * k1 = 1/1.1) or the filtered R8Unorm taps on their 0..255 scale (c = 0.9 * 255, k1 = 1/(255 * 1.1)); k2 = -0.1/1.1.
     * WGSL leaves both the R8Unorm conversion and the lowering of smoothstep's divide to the implementation; this
     * reading costs the GPU two instructions less per sample than "multiply by 1/255, subtract, multiply". */
    const float c = raw_unorm8 ? 229.5f : 0.9f;
    const float k1 = raw_unorm8 ? (float)(1.0 / (255.0 * 1.1)) : (float)(1.0 / 1.1);
    const float k2 = (float)(-0.1 / 1.1);
    float s = fmaf(vmin(x, c), k1, k2);
    s = vmin(vmax(s, 0.0f), 1.0f);
    return (s * s) * fmaf(-2.0f, s, 3.0f);
}

/* The round-1 text of the same function (oracle history: replaced by the fused form above in round 2, goldens regenerated
 * then): R8Unorm normalised by one multiply with f32(1/255) after the filter, v = min(0.9, r), s = (v - 0.1) * f32(1 / 1.1)
 * -- three roundings where the current text has one -- then the same clamp and cubic.  Kept as the yardstick that bounds how
 * far the re-specification moved alpha (tests/test_oracle_cpu.py::test_transfer_respecification_is_bounded). */
float vo_transfer_alpha_r1(float x, int raw_unorm8) {
    float r = raw_unorm8 ? x * (1.0f / 255.0f) : x;
    float v = vmin(0.9f, r);
    const float inv = 1.0f / (1.2f - 0.10f);
    float s = (v - 0.10f) * inv;
    s = vmin(vmax(s, 0.0f), 1.0f);
    return (s * s) * fmaf(-2.0f, s, 3.0f);
}

/* raycast_naive.wgsl:106-107 AS WRITTEN, no licence taken: x is a normalised value; clamp(0.4, 0.9, val) read as
 * min(0.9, val) (SURVEY F8 -- that one reading is not negotiable, bonsai.png rules the other out); smoothstep with its divide,
 * t = clamp((v - 0.10) / (1.2 - 0.10), 0, 1), t * t * (3 - 2 t), every operation rounded separately. */
float vo_transfer_alpha_literal(float x) {
    float v = vmin(0.9f, x);
    float s = (v - 0.10f) / (1.2f - 0.10f);
    s = vmin(vmax(s, 0.0f), 1.0f);
    return s * s * (3.0f - 2.0f * s);
}

void vo_vertigo(float a, float rgb[3]) {
    /* raycast_naive.wgsl:70-81: 0.5 + 0.5*cos(TAU*(c*t + d)) */
    const float TAU = 6.28318f;
    const float c[3] = {1.0f, 1.7f, 0.4f}, d[3] = {0.0f, 0.15f, 0.20f};
    for (int k = 0; k < 3; k++) rgb[k] = 0.5f + 0.5f * cosf(TAU * (c[k] * a + d[k]));
}

float vo_linear_to_srgb(float x) {
    if (x <= 0.0031308f) return 12.92f * x;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}

/* ------------------------------------------------------------------------- */
/* Ray generation                                                             */

static void mat4_mul_vec4(const float m[16], const float v[4], float o[4]) {
    for (int r = 0; r < 4; r++) {
        float s = m[0 * 4 + r] * v[0];
        s = s + m[1 * 4 + r] * v[1];
        s = s + m[2 * 4 + r] * v[2];
        s = s + m[3 * 4 + r] * v[3];
        o[r] = s;
    }
}

void vo_ray_naive(const vo_camera_uniform *cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float eye[3],
                  float dir[3]) {
    /* the pixel-centre view ray; any point on it gives the same direction
     * as the rasteriser's interpolated (pos - eye), raycast_naive.wgsl:45-46,85. */
    float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
    float ndc[4] = {(2.0f * fx) / (float)W - 1.0f, 1.0f - (2.0f * fy) / (float)H, 1.0f, 1.0f};
    float q[4];
    mat4_mul_vec4(cam->inv_proj, ndc, q);
    eye[0] = cam->view_position[0];
    eye[1] = cam->view_position[1];
    eye[2] = cam->view_position[2];
    dir[0] = q[0] / q[3] - eye[0];
    dir[1] = q[1] / q[3] - eye[1];
    dir[2] = q[2] / q[3] - eye[2];
    normalize3(dir);
}

void vo_ray_compute(const vo_camera_uniform *cam, uint32_t W, uint32_t H, float cx, float cy, float eye[3],
                    float dir[3]) {
    /* raycast_compute.wgsl:99-116: no half-pixel offset; y scaled by -H/W; eye is the near-plane point */
    float dx = (float)W, dy = (float)H;
    float aspect_ratio = dy / dx;
    float sx = 2.0f * cx / dx - 1.0f;
    float sy = 2.0f * cy / dy - 1.0f;
    sy = sy * -aspect_ratio;
    float sp[4] = {sx, sy, 0.0f, 1.0f}, st[4] = {sx, sy, 1.0f, 1.0f};
    float vp[4], vt[4];
    mat4_mul_vec4(cam->inv_proj, sp, vp);
    mat4_mul_vec4(cam->inv_proj, st, vt);
    eye[0] = vp[0] / vp[3];
    eye[1] = vp[1] / vp[3];
    eye[2] = vp[2] / vp[3];
    dir[0] = vt[0] / vt[3] - eye[0];
    dir[1] = vt[1] / vt[3] - eye[1];
    dir[2] = vt[2] / vt[3] - eye[2];
    normalize3(dir);
}

/* ------------------------------------------------------------------------- */
/* NAIVE_TRILINEAR pixel (raycast_naive.wgsl:83-125)                          */

static void pixel_naive(const vo_render_args *a, uint32_t x, uint32_t y, float out[4], uint32_t *steps,
                        uint32_t *sampled) {
    float eye[3], dir[3], th[2];
    vo_ray_naive(a->camera, a->width, a->height, x, y, eye, dir);
    vo_intersect_box(eye, dir, 0.0f, 1.0f, th);
    *steps = 0;
    *sampled = 0;
    if (th[0] > th[1]) { /* :91-93 */
        out[0] = out[1] = out[2] = 0.0f;
        out[3] = 1.0f;
        return;
    }
    th[0] = vmax(th[0], 0.0f); /* :94 */
    float C[3] = {0, 0, 0}, A = 0.0f;
    /* :97-99 with the literal 256 generalised to the volume dims */
    float dtx = 1.0f / ((float)a->nx * fabsf(dir[0]));
    float dty = 1.0f / ((float)a->ny * fabsf(dir[1]));
    float dtz = 1.0f / ((float)a->nz * fabsf(dir[2]));
    float dt = a->dt_scale * vmin(dtx, vmin(dty, dtz));
    float p[3] = {eye[0] + th[0] * dir[0], eye[1] + th[0] * dir[1], eye[2] + th[0] * dir[2]}; /* :100 */
    float step[3] = {dir[0] * dt, dir[1] * dt, dir[2] * dt}; /* :118, loop invariant */
    uint32_t n = 0, ns = 0;
    for (float t = th[0]; t < th[1]; t = t + dt) { /* :101 */
        int nonempty = 0;
        const int literal = (a->flags & VO_FLAG_LITERAL_WGSL) != 0;
        const int raw8 = a->format == VO_FMT_R8_UNORM && !(a->flags & (VO_FLAG_TAPNORM_PER_TAP | VO_FLAG_LITERAL_WGSL));
        float r = vo_sample_trilinear(a->volume, a->nx, a->ny, a->nz, a->format, p, a->flags | VO_FLAG_RAW_UNORM8, &nonempty);
        float al = literal ? vo_transfer_alpha_literal(r) : ((a->flags & VO_FLAG_TRANSFER_R1) ? vo_transfer_alpha_r1(r, raw8) : vo_transfer_alpha(r, raw8));
        float rgb[3];
        vo_vertigo(al, rgb);
        n++;
        ns += (uint32_t)nonempty;
        /* :112-114; background term is identically +0 (tex.a == 1 for a one-channel format) */
        float w = (1.0f - A) * al;
        if (literal) {
            /* :104,112 word for word: val_alpha = pow(tex.a, 2) with tex.a = 1 for a one-channel format, and the background term
             * background.rgb * background.a * (1 - val_alpha) added to every channel (it is +0: kept so that nothing is assumed) */
            const float bg[4] = {0.1f, 0.2f, 0.3f, 0.01f};
            const float val_alpha = powf(1.0f, 2.0f);
            C[0] = C[0] + w * rgb[0] + bg[0] * bg[3] * (1.0f - val_alpha);
            C[1] = C[1] + w * rgb[1] + bg[1] * bg[3] * (1.0f - val_alpha);
            C[2] = C[2] + w * rgb[2] + bg[2] * bg[3] * (1.0f - val_alpha);
        } else {
            C[0] = C[0] + w * rgb[0];
            C[1] = C[1] + w * rgb[1];
            C[2] = C[2] + w * rgb[2];
        }
        A = A + w;
        if (A >= 0.95f && !(a->flags & VO_FLAG_NO_EARLY_OUT)) break; /* :115-117 */
        p[0] = p[0] + step[0];
        p[1] = p[1] + step[1];
        p[2] = p[2] + step[2];
        /* a zero/NaN dt would never terminate; the reference would hang the GPU, we stop */
        if (!(dt > 0.0f)) break;
    }
    out[0] = vo_linear_to_srgb(C[0]);
    out[1] = vo_linear_to_srgb(C[1]);
    out[2] = vo_linear_to_srgb(C[2]);
    out[3] = 1.0f;
    *steps = n;
    *sampled = ns;
}

/* ------------------------------------------------------------------------- */
/* COMPUTE_NEAREST pixel (raycast_compute.wgsl:62-131)                        */

static inline float smoothstepf(float e0, float e1, float x) {
    const float inv = 1.0f / (e1 - e0);
    float s = (x - e0) * inv;
    s = vmin(vmax(s, 0.0f), 1.0f);
    return (s * s) * fmaf(-2.0f, s, 3.0f);
}

/* smoothstep AS WRITTEN in the WGSL specification -- t = clamp((x - e0) / (e1 - e0), 0, 1); t * t * (3 - 2 t) -- a true divide, every
 * operation rounded on its own: the literal reading of raycast_compute.wgsl:79,82 and shaders/xor.wgsl:59 (VO_FLAG_LITERAL_WGSL). */
static inline float smoothstep_literal(float e0, float e1, float x) {
    float s = (x - e0) / (e1 - e0);
    s = vmin(vmax(s, 0.0f), 1.0f);
    return s * s * (3.0f - 2.0f * s);
}

static void load_rgba16f(const uint16_t *vol, uint32_t nx, uint32_t ny, uint32_t nz, int ix, int iy, int iz,
                         float o[4]) {
    /* naga bounds policy is Unchecked (src/utils/shader_compiler.rs:89-94): out-of-range texel
     * loads are driver-defined; this build defines them as zeros. */
    if (ix < 0 || iy < 0 || iz < 0 || ix >= (int)nx || iy >= (int)ny || iz >= (int)nz) {
        o[0] = o[1] = o[2] = o[3] = 0.0f;
        return;
    }
    const uint16_t *t = vol + 4 * ((size_t)ix + (size_t)nx * ((size_t)iy + (size_t)ny * (size_t)iz));
    for (int k = 0; k < 4; k++) o[k] = vo_f16_to_f32(t[k]);
}

static inline int trunc_i32(float f) {
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int)f;
}

static void xor_noise_volume_ex(const float c[3], float time, float out[4], int literal);

/* C3: the "procedural, no volume texture" configuration.  The reference has no such example
 * (F4); this build defines it as the compute twin's ray and march (render/get_col2, raycast_compute.wgsl:62-131)
 * with the two texel loads replaced by the xor example's own density function evaluated at the sample
 * position, `noise_volume(p / 2)` (shaders/xor.wgsl:55-61; p/2 is the generator's coordinate of that point),
 * and no normals: colour = density.rgb / 2 unshaded.  24 specified sines + ~200 flops per step, 0 bytes. */
static void pixel_procedural(const vo_render_args *a, uint32_t gx, uint32_t gy, float out[4], uint32_t *steps) {
    float eye[3], dir[3], th[2];
    vo_ray_compute(a->camera, a->width, a->height, (float)gx, (float)gy, eye, dir);
    const float clear[3] = {0.023f, 0.02f, 0.02f};
    *steps = 0;
    vo_intersect_box(eye, dir, -1.0f, 1.0f, th);
    out[0] = clear[0]; out[1] = clear[1]; out[2] = clear[2]; out[3] = 1.0f;
    if (!(th[0] < th[1])) return;
    th[0] = vmax(th[0], 0.0f);
    float C[3] = {clear[0], clear[1], clear[2]}, A = 0.1f;
    const float bs = 256.0f; /* the literal grid of the xor example (examples/xor/main.rs) */
    float dtx = 1.0f / (bs * fabsf(dir[0])), dty = 1.0f / (bs * fabsf(dir[1])), dtz = 1.0f / (bs * fabsf(dir[2]));
    const float dt = a->dt_scale * vmax(vmin(dtx, vmin(dty, dtz)), 0.01f);
    uint32_t n = 0;
    for (float t = th[0]; t < th[1]; t = t + dt) {
        const float p[3] = {eye[0] + t * dir[0], eye[1] + t * dir[1], eye[2] + t * dir[2]};
        const float c[3] = {p[0] * 0.5f, p[1] * 0.5f, p[2] * 0.5f};
        float vol[4];
        const int literal = (a->flags & VO_FLAG_LITERAL_WGSL) != 0;
        xor_noise_volume_ex(c, a->proc_time, vol, literal);
        n++;
        const float vc = vol[0] / 2.0f;
        /* raycast_compute.wgsl:78-79: pow(a, 3.0) then smoothstep(0.0, 0.7, .) -- specified as a * a * a and the reciprocal form;
         * the literal reading calls powf and divides */
        float va = literal ? powf(vol[3], 3.0f) : (vol[3] * vol[3]) * vol[3];
        va = literal ? smoothstep_literal(0.0f, 0.7f, va) : smoothstepf(0.0f, 0.7f, va);
        const float w = (1.0f - A) * va;
        for (int k = 0; k < 3; k++) C[k] = C[k] + w * vc;
        A = A + w;
        if (A >= 0.95f && !(a->flags & VO_FLAG_NO_EARLY_OUT)) break;
        if (!(dt > 0.0f)) break;
    }
    out[0] = C[0]; out[1] = C[1]; out[2] = C[2];
    *steps = n;
}

static void pixel_compute(const vo_render_args *a, uint32_t gx, uint32_t gy, float out[4], uint32_t *steps) {
    /* coord = global_id + offset (:102); the tile offset is already folded into (gx,gy) */
    float eye[3], dir[3], th[2];
    vo_ray_compute(a->camera, a->width, a->height, (float)gx, (float)gy, eye, dir);
    const float clear[4] = {0.023f, 0.02f, 0.02f, 0.0f}; /* :118 */
    *steps = 0;
    vo_intersect_box(eye, dir, -1.0f, 1.0f, th);
    if (!(th[0] < th[1])) { /* :123,127 */
        out[0] = clear[0];
        out[1] = clear[1];
        out[2] = clear[2];
        out[3] = 1.0f;
        return;
    }
    th[0] = vmax(th[0], 0.0f);
    /* get_col2 :62-97 */
    float C[3] = {clear[0], clear[1], clear[2]}, A = 0.1f;
    const float bs[3] = {(float)a->nx, (float)a->ny, (float)a->nz};
    float dtx = 1.0f / (bs[0] * fabsf(dir[0]));
    float dty = 1.0f / (bs[1] * fabsf(dir[1]));
    float dtz = 1.0f / (bs[2] * fabsf(dir[2]));
    float dt = a->dt_scale * vmax(vmin(dtx, vmin(dty, dtz)), 0.01f);
    const float hb[3] = {bs[0] / 2.0f, bs[1] / 2.0f, bs[2] / 2.0f};
    float l1[3] = {-2.0f, -2.0f, -1.0f}, l2[3] = {1.0f, 1.0f, -1.0f};
    normalize3(l1);
    normalize3(l2);
    uint32_t n = 0;
    for (float t = th[0]; t < th[1]; t = t + dt) {
        float p[3] = {eye[0] + t * dir[0], eye[1] + t * dir[1], eye[2] + t * dir[2]};
        int sx = trunc_i32((p[0] + 1.0f) * hb[0]);
        int sy = trunc_i32((p[1] + 1.0f) * hb[1]);
        int sz = trunc_i32((p[2] + 1.0f) * hb[2]);
        float vc[4], nm[4];
        load_rgba16f((const uint16_t *)a->volume, a->nx, a->ny, a->nz, sx, sy, sz, vc);
        load_rgba16f((const uint16_t *)a->volume2, a->nx, a->ny, a->nz, sx, sy, sz, nm);
        n++;
        /* shade = max(0, dot(light, normal)), light = (0,-1,0) */
        float sh = vmax(0.0f, (0.0f * nm[0] + -1.0f * nm[1]) + 0.0f * nm[2]);
        float shade[3] = {sh, sh, sh};
        /* :78-79,82.  Specified reading: pow(a, 3.0) as a*a*a, smoothstep through the reciprocal of its span and one fma.
         * VO_FLAG_LITERAL_WGSL: powf, smoothstep with its divide, every operation rounded on its own (mix, dot and the
         * compositing below are already the text's own operations in the text's order). */
        const int literal = (a->flags & VO_FLAG_LITERAL_WGSL) != 0;
        float va = literal ? powf(vc[3], 3.0f) : (vc[3] * vc[3]) * vc[3];
        va = literal ? smoothstep_literal(0.0f, 0.7f, va) : smoothstepf(0.0f, 0.7f, va);
        float dl = vmax(dot3(nm, l1), 0.0f);
        float ss = literal ? smoothstep_literal(0.3f, 1.5f, dot3(p, l2)) : smoothstepf(0.3f, 1.5f, dot3(p, l2));
        float dirl[3] = {3.0f * 1.0f * dl * ss, 3.0f * 0.1f * dl * ss, 3.0f * 0.13f * dl * ss};
        float col[3] = {vc[0] + dirl[0], vc[1] + dirl[1], vc[2] + dirl[2]};
        float bl = 0.9f * vmin(vmax(0.5f - 0.5f * nm[1], 0.0f), 1.0f);
        const float blc[3] = {bl * 0.0f, bl * 0.0f, bl * 0.6f};
        for (int k = 0; k < 3; k++) shade[k] = shade[k] * (1.0f - 0.2f) + blc[k] * 0.2f; /* mix */
        float w = (1.0f - A) * va;
        for (int k = 0; k < 3; k++) {
            float tmp = C[k] + w * col[k] * shade[k];
            C[k] = tmp + clear[k] * clear[3] * (1.0f - va);
        }
        A = A + w * (1.0f - clear[3]);
        if (A >= 0.95f && !(a->flags & VO_FLAG_NO_EARLY_OUT)) break;
        if (!(dt > 0.0f)) break;
    }
    out[0] = C[0];
    out[1] = C[1];
    out[2] = C[2];
    out[3] = 1.0f;
    *steps = n;
}

/* ------------------------------------------------------------------------- */

int vo_render(const vo_render_args *a) {
    if (!a || !a->camera || !a->out_rgba) return -1;
    if (a->mode != VO_MODE_PROCEDURAL && (!a->volume || a->nx == 0 || a->ny == 0 || a->nz == 0)) return -1;
    if (a->width == 0 || a->height == 0) return -1;
    if (a->mode == VO_MODE_NAIVE_TRILINEAR && a->format != VO_FMT_R8_UNORM && a->format != VO_FMT_R16_FLOAT)
        return -2;
    if (a->mode == VO_MODE_COMPUTE_NEAREST && (a->format != VO_FMT_RGBA16F_PAIR || !a->volume2)) return -2;
    int64_t x0 = a->tile_x, y0 = a->tile_y;
    int64_t x1 = x0 + (int64_t)a->tile_w, y1 = y0 + (int64_t)a->tile_h;
    int64_t cx0 = x0 < 0 ? 0 : x0, cy0 = y0 < 0 ? 0 : y0;
    int64_t cx1 = x1 > (int64_t)a->width ? (int64_t)a->width : x1;
    int64_t cy1 = y1 > (int64_t)a->height ? (int64_t)a->height : y1;
    if (cx1 <= cx0 || cy1 <= cy0) return 0; /* off-screen tile: stores dropped (A12) */
#ifdef _OPENMP
    int nthreads = a->threads > 0 ? a->threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int64_t y = cy0; y < cy1; y++) {
        for (int64_t x = cx0; x < cx1; x++) {
            size_t pix = (size_t)y * a->width + (size_t)x;
            uint32_t st = 0, sm = 0;
            if (a->mode == VO_MODE_NAIVE_TRILINEAR)
                pixel_naive(a, (uint32_t)x, (uint32_t)y, a->out_rgba + 4 * pix, &st, &sm);
            else if (a->mode == VO_MODE_PROCEDURAL) {
                pixel_procedural(a, (uint32_t)x, (uint32_t)y, a->out_rgba + 4 * pix, &st);
                sm = st;
            } else {
                pixel_compute(a, (uint32_t)x, (uint32_t)y, a->out_rgba + 4 * pix, &st);
                sm = st;
            }
            if (a->out_steps) a->out_steps[pix] = st;
            if (a->out_sampled) a->out_sampled[pix] = sm;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* xor volume generator: shaders/xor.wgsl                                     */

/* sin(h) for hash(): the WGSL sin of arguments up to ~1e5 is implementation-sensitive and the
 * hash multiplies its error by 43758, so the specification pins it: Cody-Waite
 * reduction and minimax polynomials in f64, rounded once to f32 (== the correctly rounded f32
 * sine except in ~1e-8 of cases, identically so wherever this op sequence runs). */
static float sin_spec(float h) {
    const double x = (double)h;
    const double k = rint(x * 0.63661977236758134308);             /* 2/pi */
    double r = fma(-k, 1.57079632673412561417e+00, x);            /* pi/2 high */
    r = fma(-k, 6.07710050650619224932e-11, r);                   /* pi/2 low */
    const double r2 = r * r;
    double sp = 1.58969099521155010221e-10;                       /* sin: r + r^3 * P(r^2) */
    sp = fma(sp, r2, -2.50507602534068634195e-08);
    sp = fma(sp, r2, 2.75573137070700676789e-06);
    sp = fma(sp, r2, -1.98412698298579493134e-04);
    sp = fma(sp, r2, 8.33333333332248946124e-03);
    sp = fma(sp, r2, -1.66666666666666324348e-01);
    const double sn = fma(r * r2, sp, r);
    double cp = -1.13596475577881948265e-11;                      /* cos: 1 - r^2/2 + r^4 * Q(r^2) */
    cp = fma(cp, r2, 2.08757232129817482790e-09);
    cp = fma(cp, r2, -2.75573143513906633035e-07);
    cp = fma(cp, r2, 2.48015872894767294178e-05);
    cp = fma(cp, r2, -1.38888888888741095749e-03);
    cp = fma(cp, r2, 4.16666666666666019037e-02);
    const double cs = fma(r2 * r2, cp, fma(-0.5, r2, 1.0));
    const long q = (long)k & 3;
    const double v = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
    return (float)v;
}

float vo_sin_spec(float h) { return sin_spec(h); }

static inline float fractf_(float x) { return x - floorf(x); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; } /* WGSL mix */

static float xor_hash(float h) { return fractf_(sin_spec(h) * 43758.5453123f); } /* xor.wgsl:18-20 */

static float xor_noise(const float x[3]) { /* xor.wgsl:22-33 */
    float p[3] = {floorf(x[0]), floorf(x[1]), floorf(x[2])};
    float f[3] = {fractf_(x[0]), fractf_(x[1]), fractf_(x[2])};
    for (int i = 0; i < 3; i++) f[i] = f[i] * f[i] * (3.0f - 2.0f * f[i]);
    float n = p[0] + p[1] * 157.0f + 113.0f * p[2];
    return mixf(mixf(mixf(xor_hash(n + 0.0f), xor_hash(n + 1.0f), f[0]), mixf(xor_hash(n + 157.0f), xor_hash(n + 158.0f), f[0]), f[1]),
                mixf(mixf(xor_hash(n + 113.0f), xor_hash(n + 114.0f), f[0]), mixf(xor_hash(n + 270.0f), xor_hash(n + 271.0f), f[0]), f[1]),
                f[2]);
}

static float xor_fbm(const float p0[3]) { /* xor.wgsl:35-44 */
    float p[3] = {p0[0], p0[1], p0[2]};
    float f = 0.5000f * xor_noise(p);
    for (int i = 0; i < 3; i++) p[i] = p[i] * 2.01f;
    f = f + 0.2500f * xor_noise(p);
    for (int i = 0; i < 3; i++) p[i] = p[i] * 2.02f;
    f = f + 0.1250f * xor_noise(p);
    return f;
}

/* literal != 0: the falloff's smoothstep(0.5, 0.25, length(coord)) with its divide (xor.wgsl:59 as written); the hash's sine stays
 * the specified one either way (a device's own sine is not a reading of the text but another function: SURVEY 8d C3) */
static void xor_noise_volume_ex(const float c[3], float time, float out[4], int literal) { /* xor.wgsl:55-61 */
    float off[3] = {1.0f, sin_spec(time * 1.0f) * 0.1f, 21.0f};
    float pos[3] = {(c[0] + off[0]) * 32.0f, (c[1] + off[1]) * 32.0f, (c[2] + off[2]) * 32.0f};
    float val = xor_fbm(pos);
    float len = sqrtf((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
    float alpha = val * (literal ? smoothstep_literal(0.5f, 0.25f, len) : smoothstepf(0.5f, 0.25f, len));
    out[0] = out[1] = out[2] = val;
    out[3] = alpha;
}
static void xor_noise_volume(const float c[3], float time, float out[4]) { xor_noise_volume_ex(c, time, out, 0); }

void vo_volume_xor(uint32_t nx, uint32_t ny, uint32_t nz, float time, uint16_t *density, uint16_t *normals) {
    const float dims[3] = {(float)nx, (float)ny, (float)nz};
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t z = 0; z < (int64_t)nz; z++)
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < nx; x++) {
                const float id[3] = {(float)x, (float)y, (float)z};
                float c[3];
                for (int i = 0; i < 3; i++) c[i] = (id[i] - dims[i] / 2.0f) / dims[i]; /* xor.wgsl:72 */
                float vol[4], a[3];
                xor_noise_volume(c, time, vol);
                /* gradient(coord, 0.0001): xor.wgsl:63-67 */
                for (int k = 0; k < 3; k++) {
                    float q[3] = {c[0], c[1], c[2]}, t4[4];
                    q[k] = q[k] - 0.0001f;
                    xor_noise_volume(q, time, t4);
                    a[k] = vol[3] - t4[3];
                }
                float nl = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
                float nrm[3] = {a[0] / nl, a[1] / nl, a[2] / nl}; /* normalize(0) = NaN, as on a GPU */
                float ln = sqrtf((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
                size_t o = 4 * ((size_t)x + (size_t)nx * (y + (size_t)ny * (size_t)z));
                density[o + 0] = vo_f32_to_f16(vol[0] / 2.0f);
                density[o + 1] = vo_f32_to_f16(vol[1] / 2.0f);
                density[o + 2] = vo_f32_to_f16(vol[2] / 2.0f);
                density[o + 3] = vo_f32_to_f16(vol[3]);
                normals[o + 0] = vo_f32_to_f16(nrm[0]);
                normals[o + 1] = vo_f32_to_f16(nrm[1]);
                normals[o + 2] = vo_f32_to_f16(nrm[2]);
                normals[o + 3] = vo_f32_to_f16(ln);
            }
}

/* ------------------------------------------------------------------------- */
/* Present pass: shaders/present.wgsl                                         */

static inline float aces_film(float x) { /* present.wgsl:33-35 */
    float num = x * (2.51f * x + 0.03f);
    float den = x * (2.43f * x + 0.59f) + 0.14f;
    return vmin(vmax(num / den, 0.0f), 1.0f);
}

static inline float present_srgb(float c) { /* present.wgsl:23-30: branch-free, exponent 0.41666 */
    float sel = ceilf(c - 0.0031308f);
    float under = 12.92f * c;
    float over = 1.055f * powf(c, 0.41666f) - 0.055f;
    return under * (1.0f - sel) + over * sel; /* mix(under, over, sel) */
}

void vo_present(const float *bb, uint32_t bw, uint32_t bh, uint32_t w, uint32_t h, uint8_t *out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t y = 0; y < (int64_t)h; y++)
        for (uint32_t x = 0; x < w; x++) {
            /* fullscreen triangle (present.wgsl:98-104): uv at the pixel centre */
            float uvx = ((float)x + 0.5f) / (float)w, uvy = ((float)y + 0.5f) / (float)h;
            float ux = fmaf(uvx, (float)bw, -0.5f), uy = fmaf(uvy, (float)bh, -0.5f);
            float flx = floorf(ux), fly = floorf(uy);
            float fx = ux - flx, fy = uy - fly;
            if (fx >= 1.0f) fx = 0x1.fffffep-1f;
            if (fy >= 1.0f) fy = 0x1.fffffep-1f;
            int ix = (int)flx, iy = (int)fly;
            int x0 = clampi(ix, 0, (int)bw - 1), x1 = clampi(ix + 1, 0, (int)bw - 1);
            int y0 = clampi(iy, 0, (int)bh - 1), y1 = clampi(iy + 1, 0, (int)bh - 1);
            uint8_t *o = out + 4 * ((size_t)y * w + x);
            for (int c = 0; c < 4; c++) {
                float t00 = bb[4 * ((size_t)y0 * bw + x0) + c], t10 = bb[4 * ((size_t)y0 * bw + x1) + c];
                float t01 = bb[4 * ((size_t)y1 * bw + x0) + c], t11 = bb[4 * ((size_t)y1 * bw + x1) + c];
                float a = lerp_fma(t00, t10, fx), b = lerp_fma(t01, t11, fx);
                float v = lerp_fma(a, b, fy);
                if (c < 3) v = present_srgb(aces_film(v));
                v = vmin(vmax(v, 0.0f), 1.0f);
                o[c] = (uint8_t)floorf(v * 255.0f + 0.5f);
            }
        }
}

/* ------------------------------------------------------------------------- */
/* Deterministic volumes (integer-only)                                       */

static inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

static inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed) {
    return lowbias32(seed ^ (x * 0x9E3779B1U + y * 0x85EBCA77U + z * 0xC2B2AE3DU));
}

/* integer trilinear value noise on 12-bit coordinates; lattice cell = 2^sh units; returns 0..255 */
static uint32_t vnoise(uint32_t X, uint32_t Y, uint32_t Z, uint32_t sh, uint32_t seed) {
    uint32_t m = (1u << sh) - 1, S = 1u << sh;
    uint32_t cx = X >> sh, cy = Y >> sh, cz = Z >> sh;
    uint32_t fx = X & m, fy = Y & m, fz = Z & m;
    uint64_t acc = 0;
    for (uint32_t dz = 0; dz < 2; dz++)
        for (uint32_t dy = 0; dy < 2; dy++)
            for (uint32_t dx = 0; dx < 2; dx++) {
                uint64_t w = (uint64_t)(dx ? fx : S - fx) * (dy ? fy : S - fy) * (dz ? fz : S - fz);
                acc += w * (hash3(cx + dx, cy + dy, cz + dz, seed) >> 24);
            }
    return (uint32_t)(acc >> (3 * sh));
}

static inline uint8_t standin_voxel(uint32_t x, uint32_t y, uint32_t z, uint32_t nx, uint32_t ny, uint32_t nz,
                                    uint32_t seed) {
    /* centre-sampled 12-bit coordinates: 16 units per voxel at 256^3 */
    int32_t X = (int32_t)(((2 * (uint64_t)x + 1) * 2048) / nx);
    int32_t Y = (int32_t)(((2 * (uint64_t)y + 1) * 2048) / ny);
    int32_t Z = (int32_t)(((2 * (uint64_t)z + 1) * 2048) / nz);
    uint32_t n_lo = vnoise((uint32_t)X, (uint32_t)Y, (uint32_t)Z, 9, seed ^ 0x1111u);  /* 32-voxel cells */
    uint32_t n_hi = vnoise((uint32_t)X, (uint32_t)Y, (uint32_t)Z, 7, seed ^ 0x2222u);  /* 8-voxel cells */
    /* pot: dense ellipsoid, always >= 230 */
    {
        int64_t dx = X - 2048, dy = Y - 600, dz = Z - 2048;
        int64_t e = dx * dx + 7 * dy * dy + dz * dz;
        if (e < 1400 * 1400) return (uint8_t)(232 + (n_hi >> 4));
    }
    /* trunk: bent tapering cylinder, mid density */
    if (Y >= 900 && Y < 2600) {
        int32_t h = Y - 900;
        int64_t cx = 2048 + ((int64_t)h * h) / 8000, cz = 2048 - h / 6;
        int64_t rr = 230 - h / 12;
        int64_t dx = X - cx, dz = Z - cz;
        if (dx * dx + dz * dz < rr * rr) return (uint8_t)(110 + (n_hi >> 2));
    }
    /* canopy: noise-thresholded ellipsoid, low..mid density, smooth */
    {
        int64_t dx = X - 2150, dy = Y - 2850, dz = Z - 1950;
        /* radii 1750 / 1050 / 1750 -> normalise to 0..256 at the surface */
        int64_t q = (dx * dx * 256) / (1750 * 1750) + (dy * dy * 256) / (1050 * 1050) + (dz * dz * 256) / (1750 * 1750);
        if (q < 256) {
            int32_t f = (int32_t)((2 * n_lo + n_hi) / 3);
            int32_t d = f - (int32_t)(q / 3) - 52;
            if (d > 0) {
                int32_t v = 28 + 2 * d;
                return (uint8_t)(v > 225 ? 225 : v);
            }
        }
    }
    /* air: white noise 0..20 (exactly transparent) + 0.2 % speckle 26..41 */
    uint32_t h = hash3(x, y, z, seed ^ 0x3333u);
    if ((h & 0x1ffu) == 0) return (uint8_t)(26 + ((h >> 9) & 15));
    return (uint8_t)((h >> 16) % 21);
}

void vo_volume_standin_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint8_t *out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t z = 0; z < (int64_t)nz; z++)
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < nx; x++)
                out[(size_t)x + (size_t)nx * (y + (size_t)ny * (size_t)z)] =
                    standin_voxel(x, y, (uint32_t)z, nx, ny, nz, seed);
}

/* Dense ball at the centre of the fog volumes ("dense-core variant" of C4 / C5): radius a quarter of the
 * smallest dimension, tested in doubled integer coordinates. */
static int in_dense_core(uint32_t x, uint32_t y, uint32_t z, uint32_t nx, uint32_t ny, uint32_t nz) {
    int64_t dx = 2 * (int64_t)x + 1 - (int64_t)nx, dy = 2 * (int64_t)y + 1 - (int64_t)ny, dz = 2 * (int64_t)z + 1 - (int64_t)nz;
    int64_t m = nx < ny ? (nx < nz ? nx : nz) : (ny < nz ? ny : nz);
    int64_t r = m / 2;
    return dx * dx + dy * dy + dz * dz < r * r;
}

void vo_volume_fog_core_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint32_t lo, uint32_t span, int core,
                           uint8_t *out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t z = 0; z < (int64_t)nz; z++)
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < nx; x++) {
                uint32_t h = hash3(x, y, (uint32_t)z, seed) >> 8;
                int dense = core && in_dense_core(x, y, (uint32_t)z, nx, ny, nz);
                out[(size_t)x + (size_t)nx * (y + (size_t)ny * (size_t)z)] = (uint8_t)(dense ? 232u + h % 24u : lo + h % span);
            }
}

void vo_volume_fog_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint32_t lo, uint32_t span,
                      uint8_t *out) {
    vo_volume_fog_core_u8(nx, ny, nz, seed, lo, span, 0, out);
}

void vo_volume_fog_core_f16(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, int core, uint16_t *out) {
    /* f16 bit patterns 0x2D1F (0.08) .. 0x2FAE (0.12): monotone in value, integer-only; core 0x3B9A (0.95) .. 0x3BD9 */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t z = 0; z < (int64_t)nz; z++)
        for (uint32_t y = 0; y < ny; y++)
            for (uint32_t x = 0; x < nx; x++) {
                uint32_t h = hash3(x, y, (uint32_t)z, seed) >> 8;
                int dense = core && in_dense_core(x, y, (uint32_t)z, nx, ny, nz);
                out[(size_t)x + (size_t)nx * (y + (size_t)ny * (size_t)z)] = (uint16_t)(dense ? 0x3B9Au + h % 64u : 0x2D1Fu + h % 656u);
            }
}

void vo_volume_fog_f16(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint16_t *out) {
    vo_volume_fog_core_f16(nx, ny, nz, seed, 0, out);
}

/* ------------------------------------------------------------------------- */

uint32_t vo_dispatch_optimal(uint32_t len, uint32_t subgroup) {
    /* src/utils/mod.rs:15-18 */
    uint32_t padded = (subgroup - len % subgroup) % subgroup;
    return (len + padded) / subgroup;
}

void vo_image_dimentions(uint32_t w, uint32_t h, uint32_t align, uint32_t out4[4]) {
    /* src/utils/mod.rs:99-113 */
    h = h - (h % 2);
    w = w - (w % 2);
    uint32_t unpadded = w * 4;
    uint32_t pad = (align - unpadded % align) % align;
    out4[0] = w;
    out4[1] = h;
    out4[2] = unpadded;
    out4[3] = unpadded + pad;
}
