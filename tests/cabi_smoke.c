/* cabi_smoke.c -- a plain C99 consumer of include/vokselis_hip.h: no C++, no HIP headers, only the C-ABI.
 * Generates a small fog volume on the device, renders one frame with vk_render and the same camera twice through
 * vk_render_batch, reads everything back and checks that the three frames agree bit for bit and are not empty.
 * Built and run by tests/test_parity_gpu.py::test_group_api_and_plain_c_consumer (gcc ... -lvokselis_hip). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vokselis_hip.h"

#define CHECK(ctx, call)                                                                 \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != VK_OK) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vk_last_error(ctx));           \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

/* CameraUniform for eye (1.6, 1.2, 1.9) looking at the cube centre, 90 degree fov, aspect 1: view_pos, P*V and its
 * inverse, column-major (src/camera.rs:5-11).  Built here in double with a textbook look-at / perspective / Gauss-Jordan:
 * any finite camera will do for this smoke test. */
static void mat_mul(const double *a, const double *b, double *o) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a[k * 4 + r] * b[c * 4 + k];
            o[c * 4 + r] = s;
        }
}
static int mat_inv(const double *m, double *o) {
    double a[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c; }
    for (int i = 0; i < 4; i++) {
        int p = i;
        for (int r = i + 1; r < 4; r++) if (fabs(a[r][i]) > fabs(a[p][i])) p = r;
        if (fabs(a[p][i]) < 1e-12) return 1;
        for (int c = 0; c < 8; c++) { double t = a[i][c]; a[i][c] = a[p][c]; a[p][c] = t; }
        double d = a[i][i];
        for (int c = 0; c < 8; c++) a[i][c] /= d;
        for (int r = 0; r < 4; r++) if (r != i) { double f = a[r][i]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o[c * 4 + r] = a[r][4 + c];
    return 0;
}
static void camera_blob(float *blob) {
    const double eye[3] = {1.6, 1.2, 1.9}, at[3] = {0.5, 0.5, 0.5};
    double f[3] = {eye[0] - at[0], eye[1] - at[1], eye[2] - at[2]};
    double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int i = 0; i < 3; i++) f[i] /= fl;
    double s[3] = {1.0 * f[2] - 0.0 * f[1], 0.0 * f[0] - 0.0 * f[2], 0.0 * f[1] - 1.0 * f[0]}; /* cross((0,1,0), f) */
    double sl = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (int i = 0; i < 3; i++) s[i] /= sl;
    double u[3] = {f[1] * s[2] - f[2] * s[1], f[2] * s[0] - f[0] * s[2], f[0] * s[1] - f[1] * s[0]};
    double V[16] = {s[0], u[0], f[0], 0, s[1], u[1], f[1], 0, s[2], u[2], f[2], 0,
                    -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]), -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]),
                    -(f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]), 1};
    const double n = 0.1, fa = 100.0, h = 1.0, w = 1.0, r = fa / (n - fa);
    double P[16] = {w, 0, 0, 0, 0, h, 0, 0, 0, 0, r, -1, 0, 0, r * n, 0};
    double PV[16], INV[16];
    mat_mul(P, V, PV);
    mat_inv(PV, INV);
    blob[0] = (float)eye[0]; blob[1] = (float)eye[1]; blob[2] = (float)eye[2]; blob[3] = 1.0f;
    for (int i = 0; i < 16; i++) { blob[4 + i] = (float)PV[i]; blob[20 + i] = (float)INV[i]; }
}

int main(void) {
    enum { W = 96, H = 96, N = 32 };
    if (vk_abi_version() != VK_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    vk_ctx *ctx = NULL;
    if (vk_ctx_create(0, &ctx) != VK_OK) { fprintf(stderr, "vk_ctx_create: %s\n", vk_last_error(NULL)); return 1; }
    float cam[36];
    camera_blob(cam);
    CHECK(ctx, vk_backbuffer_resize(ctx, W, H, VK_OUT_RGBA32F));
    CHECK(ctx, vk_volume_generate(ctx, VK_GEN_FOG, N, N, N, VK_FMT_R8_UNORM, 7u, 30u, 40u, VK_LAYOUT_AUTO));
    CHECK(ctx, vk_set_camera(ctx, cam));
    CHECK(ctx, vk_render(ctx, VK_MODE_NAIVE_TRILINEAR, 0, 0, W, H, 0.5f, 0));
    float *one = (float *)malloc(sizeof(float) * 4 * W * H), *two = (float *)malloc(sizeof(float) * 4 * W * H * 2);
    if (!one || !two) return 1;
    CHECK(ctx, vk_readback(ctx, one, sizeof(float) * 4 * W));
    float cams[72];
    memcpy(cams, cam, 144); memcpy(cams + 36, cam, 144);
    void *frames = NULL;
    CHECK(ctx, vk_device_alloc(ctx, sizeof(float) * 4 * W * H * 2, &frames));
    CHECK(ctx, vk_render_batch(ctx, VK_MODE_NAIVE_TRILINEAR, 2, cams, 32, 0, 1, 0.5f, 0, frames, 0, 0, NULL, NULL));
    CHECK(ctx, vk_device_download(ctx, two, frames, sizeof(float) * 4 * W * H * 2));
    CHECK(ctx, vk_device_free(ctx, frames));
    int lit = 0;
    for (int i = 0; i < W * H; i++) lit += one[4 * i] > 0.0f || one[4 * i + 1] > 0.0f || one[4 * i + 2] > 0.0f;
    if (lit < W * H / 20) { fprintf(stderr, "frame is empty (%d lit pixels)\n", lit); return 1; }
    if (memcmp(one, two, sizeof(float) * 4 * W * H) != 0 || memcmp(one, two + 4 * W * H, sizeof(float) * 4 * W * H) != 0) {
        fprintf(stderr, "vk_render_batch frames differ from vk_render\n");
        return 1;
    }
    /* the interactive shape (ABI 5): three frames in flight, the present fused into the pass; the frame that stays in its slot is the frame
     * above bit for bit, and its presented Rgba8 image is the one vk_present makes of it, to one 8-bit step */
    {
        uint64_t id[4] = {0, 0, 0, 0};
        uint32_t cw = 0, ch = 0, pitch = 0;
        CHECK(ctx, vk_present(ctx, W, H, 0));  /* of the frame rendered above */
        CHECK(ctx, vk_capture_frame(ctx, NULL, 0, &cw, &ch, &pitch));  /* sizes: even-rounded, rows padded to 256 bytes (src/utils/mod.rs:99-117) */
        if (cw != W || ch != H || pitch != 512) { fprintf(stderr, "capture dims %ux%u pitch %u\n", cw, ch, pitch); return 1; }
        unsigned char *shot_a = (unsigned char *)malloc((size_t)pitch * ch), *shot_b = (unsigned char *)malloc((size_t)pitch * ch);
        if (!shot_a || !shot_b) return 1;
        CHECK(ctx, vk_capture_frame(ctx, shot_a, (size_t)pitch * ch, NULL, NULL, NULL));
        CHECK(ctx, vk_ctx_frames_in_flight(ctx, 3));
        for (int k = 0; k < 4; k++) {
            CHECK(ctx, vk_set_camera(ctx, cam));
            CHECK(ctx, vk_frame_begin(ctx, &id[k]));
            CHECK(ctx, vk_render(ctx, VK_MODE_NAIVE_TRILINEAR, 0, 0, W, H, 0.5f, VK_RENDER_PRESENT));
            CHECK(ctx, vk_frame_end(ctx));
        }
        if (!(id[0] >= 1 && id[3] == id[0] + 3)) { fprintf(stderr, "frame ids %llu .. %llu\n", (unsigned long long)id[0], (unsigned long long)id[3]); return 1; }
        if (vk_frame_readback(ctx, id[0], two, sizeof(float) * 4 * W) != VK_ERR_INVALID) { fprintf(stderr, "a frame whose slot was retaken could still be read\n"); return 1; }
        CHECK(ctx, vk_frame_wait(ctx, id[0]));
        CHECK(ctx, vk_frame_readback(ctx, id[2], two, sizeof(float) * 4 * W));
        if (memcmp(one, two, sizeof(float) * 4 * W * H) != 0) { fprintf(stderr, "a frame in flight differs from vk_render's\n"); return 1; }
        CHECK(ctx, vk_frame_capture(ctx, id[3], shot_b, (size_t)pitch * ch, NULL, NULL, NULL));
        for (int i = 0; i < (int)(pitch * ch); i++) {
            int d = (int)shot_a[i] - (int)shot_b[i];
            if (d < -1 || d > 1) { fprintf(stderr, "fused present differs from vk_present by %d at byte %d\n", d, i); return 1; }
        }
        CHECK(ctx, vk_ctx_frames_in_flight(ctx, 1));
        free(shot_a); free(shot_b);
    }
    /* error behaviour across the boundary: codes, never aborts */
    if (vk_render(ctx, 77, 0, 0, W, H, 0.5f, 0) != VK_ERR_INVALID) { fprintf(stderr, "bad mode was accepted\n"); return 1; }
    if (vk_dispatch_optimal(1920, 8) != 240 || vk_dispatch_optimal(1081, 8) != 136) { fprintf(stderr, "dispatch_optimal\n"); return 1; }
    CHECK(ctx, vk_ctx_destroy(ctx));
    free(one); free(two);
    printf("cabi_smoke: OK (%d lit pixels)\n", lit);
    return 0;
}
