/*
 * vokselis_oracle.h -- CPU restatement of the vokselis raycast hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (vokselis_amd/, the
 * C-ABI library, the C++ host) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (pudnax/vokselis, Rust + wgpu + WGSL) ships
 * no tests, no golden vectors and cannot be built or run here (no Rust, no
 * Vulkan ICD, the bonsai .raw is absent from the checkout).  This oracle is
 * pinned only against a second, independent numpy restatement
 * (oracle/np_restatement.py) and the fixtures under tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  The arithmetic is a *specification*: each float
 * operation is an IEEE-754 binary32 op in the order written, built with
 * -ffp-contract=off, and fused multiply-adds appear only where fmaf() is
 * written.  The HIP kernels reproduce the opacity path op-for-op, so loop
 * trip counts and the accumulated alpha are bit-identical; only cos()/pow()
 * (colour, never control flow) differ in the last ulps.
 */
#ifndef VOKSELIS_ORACLE_H
#define VOKSELIS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/camera.rs:5-11 -- 144-byte CameraUniform, matrices column-major. */
typedef struct vo_camera_uniform {
    float view_position[4];
    float proj_view[16];
    float inv_proj[16]; /* = inverse(proj * view); the name is the reference's */
} vo_camera_uniform;

/* src/context/global_ubo.rs:52-65 -- 48-byte Uniform. */
typedef struct vo_uniform {
    float pos[3];
    uint32_t frame;
    float resolution[2];
    float mouse[2];
    uint32_t mouse_pressed;
    float time;
    float time_delta;
    float _padding;
} vo_uniform;

enum { VO_FMT_R8_UNORM = 0, VO_FMT_R16_FLOAT = 1, VO_FMT_RGBA16F_PAIR = 2 };
enum { VO_MODE_NAIVE_TRILINEAR = 0, VO_MODE_COMPUTE_NEAREST = 1, VO_MODE_PROCEDURAL = 2 /* no volume, see pixel_procedural */ };

/* Flags for vo_render. */
enum {
    VO_FLAG_NO_EARLY_OUT = 1, /* count/march S_nominal: ignore the alpha>=0.95 break */
    VO_FLAG_TAPNORM_PER_TAP = 2, /* ablation: normalise u8 taps (c/255) before the lerps */
    VO_FLAG_RAW_UNORM8 = 4, /* vo_sample_trilinear: leave filtered R8Unorm taps on their 0..255 scale (the march's own call) */
    /* NAIVE mode: evaluate raycast_naive.wgsl:96-119 AS WRITTEN instead of as specified -- texel coordinate p*n - 0.5 with two
     * roundings, every R8Unorm tap divided by 255, lerps a + f*(b - a) without fused operations, smoothstep's (v - 0.1) / (1.2 -
     * 0.1) as a true divide, the background term kept, libm cosf / powf.  The yardstick for the distance between "what the
     * shader says" and "what the three implementations agree on" (DESIGN.md 2.1); never the checker of the HIP path.
     * COMPUTE_NEAREST / PROCEDURAL (round 6): raycast_compute.wgsl:62-97 as written -- pow(a, 3.0) through powf, both smoothsteps (and
     * xor.wgsl:59's, in PROCEDURAL) with their divide and no fused operation; mix, dot and the compositing are the text's own
     * operations in the text's order in either reading. */
    VO_FLAG_LITERAL_WGSL = 8,
    VO_FLAG_TRANSFER_R1 = 16 /* NAIVE mode: the round-1 text of the transfer function (vo_transfer_alpha_r1) in an otherwise specified march */
};

typedef struct vo_render_args {
    const vo_camera_uniform *camera;
    const void *volume;   /* R8: u8[nz][ny][nx]; R16F: u16 bits; PAIR: density rgba16f */
    const void *volume2;  /* PAIR only: normals rgba16f */
    uint32_t nx, ny, nz;
    int format;
    int mode;
    uint32_t width, height;      /* full image */
    int32_t tile_x, tile_y;      /* tile origin (may be off-screen) */
    uint32_t tile_w, tile_h;
    float dt_scale;
    int flags;
    int threads;                 /* OpenMP threads, <=0: runtime default */
    float *out_rgba;             /* [height][width][4] f32, only tile pixels written */
    uint32_t *out_steps;         /* optional [height][width]: loop iterations executed */
    uint32_t *out_sampled;       /* optional: iterations with at least one tap > 25 (u8) */
    float proc_time;             /* PROCEDURAL: un.time of noise_volume (the reference runs it at 0) */
} vo_render_args;

/* Camera::new + fix_eye + build_projection_view_matrix + get_proj_view_matrix
 * (src/camera.rs:93-113,148-171) with glam 0.20.5's look_at_rh / perspective_rh. */
void vo_camera_uniform_build(float zoom, float pitch, float yaw, const float target[3],
                             float aspect, vo_camera_uniform *out);
void vo_camera_eye(float zoom, float pitch, float yaw, const float target[3], float eye[3]);

/* shaders/raycast_naive.wgsl:50-61 (lo=0,hi=1) and raycast_compute.wgsl:42-53 (lo=-1,hi=1). */
void vo_intersect_box(const float orig[3], const float dir[3], float lo, float hi, float t01[2]);

/* textureSampleLevel on R8Unorm / R16Float D3 with a linear clamp-to-edge sampler
 * (shaders/raycast_naive.wgsl:102, src/context/volume_texture.rs:39-66). */
float vo_sample_trilinear(const void *vol, uint32_t nx, uint32_t ny, uint32_t nz, int format,
                          const float p[3], int flags, int *any_above_25);

/* shaders/raycast_naive.wgsl:106-107 (clamp with low>high read as min(0.9,v)). */
float vo_transfer_alpha(float x, int raw_unorm8);
float vo_transfer_alpha_r1(float x, int raw_unorm8); /* the round-1 text (three roundings); yardstick only */
float vo_transfer_alpha_literal(float x);            /* the shader's text with its divide; yardstick only */
/* shaders/raycast_naive.wgsl:70-81. */
void vo_vertigo(float a, float rgb[3]);
/* shaders/raycast_naive.wgsl:63-68. */
float vo_linear_to_srgb(float x);

/* One pixel's ray for the NAIVE mode (replaces vs_main + rasteriser). */
void vo_ray_naive(const vo_camera_uniform *cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                  float eye[3], float dir[3]);
/* raycast_compute.wgsl:99-116. */
void vo_ray_compute(const vo_camera_uniform *cam, uint32_t W, uint32_t H, float cx, float cy,
                    float eye[3], float dir[3]);

/* Whole frame / tile.  Returns 0, or <0 on bad arguments. */
int vo_render(const vo_render_args *a);

/* RNE f32 -> f16 bits and back (rgba16float backbuffer, src/context/hdr_backbuffer.rs:10). */
uint16_t vo_f32_to_f16(float f);
float vo_f16_to_f32(uint16_t h);
void vo_rgba32f_to_rgba16f(const float *src, uint16_t *dst, size_t n_floats);

/* Deterministic inputs (integer-only, bit-exact with oracle/volumes.py and the HIP generator). */
void vo_volume_standin_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint8_t *out);
void vo_volume_fog_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint32_t lo,
                      uint32_t span, uint8_t *out);
void vo_volume_fog_f16(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint16_t *out);
/* The same fogs with a dense ball at the centre when core != 0 ("dense-core variant" of C4 / C5). */
void vo_volume_fog_core_u8(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, uint32_t lo, uint32_t span, int core,
                           uint8_t *out);
void vo_volume_fog_core_f16(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, int core, uint16_t *out);

/* xor example's volume generator (next row N3): shaders/xor.wgsl:18-78 `cs_main` at un.time = t
 * (the reference runs it once with time = 0).  Writes nx*ny*nz rgba16f texels (4 x u16)
 * of density (vol.rgb/2, vol.a) and normals (normal, length(normal)).  hash()'s sin is specified as
 * the correctly rounded f32 sine (evaluated in f64), which makes the volume reproducible. */
float vo_sin_spec(float h); /* the specified sine of hash(), exposed for tests */
void vo_volume_xor(uint32_t nx, uint32_t ny, uint32_t nz, float time, uint16_t *density, uint16_t *normals);

/* Present pass (next row N1): shaders/present.wgsl:23-35,111-119 with the linear clamp-to-edge sampler of
 * src/context/present_pipeline.rs:95-104.  backbuffer: [bh][bw][4] f32 (already rounded through f16
 * by the caller when the surface is rgba16float); out: [h][w][4] u8 in RGBA order (the Rgba8Unorm
 * copy that capture_frame reads, src/context.rs:339-359).  unorm8 = floor(c*255 + 0.5). */
void vo_present(const float *backbuffer, uint32_t bw, uint32_t bh, uint32_t w, uint32_t h, uint8_t *out_rgba8);

/* src/utils/mod.rs:15-18 and :99-117 */
uint32_t vo_dispatch_optimal(uint32_t len, uint32_t subgroup);
void vo_image_dimentions(uint32_t w, uint32_t h, uint32_t align, uint32_t out4[4]);

#ifdef __cplusplus
}
#endif
#endif
