// vokselis.hpp -- headless C++ host above the C-ABI (include/vokselis_hip.h), restating the public
// surface of the reference crate (src/lib.rs:13-18,37-49): Camera, CameraUniform, Uniform,
// HdrBackBuffer, VolumeTexture, Context, Demo, run -- same names, argument meaning and call order.
// The reference host is Rust; this image has no Rust toolchain, so the compiled host is C++17.
// Window, input, hot reload and the present pass are out of scope.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vokselis_hip.h"

namespace vokselis {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("vokselis_hip error " + std::to_string(c) + ": " + m), code(c) {}
};

inline void check(vk_ctx *ctx, int rc) {
    if (rc != VK_OK) {
        const char *m = vk_last_error(ctx);
        throw Error(rc, m ? m : "");
    }
}

// src/utils/mod.rs:15-18
inline uint32_t dispatch_optimal(uint32_t len, uint32_t subgroup_size) {
    uint32_t padded = (subgroup_size - len % subgroup_size) % subgroup_size;
    return (len + padded) / subgroup_size;
}

// src/utils/mod.rs:91-118 (the reference's spelling)
struct ImageDimentions {
    uint32_t width, height, unpadded_bytes_per_row, padded_bytes_per_row;
    ImageDimentions(uint32_t w, uint32_t h, uint32_t align) {
        height = h - (h % 2);
        width = w - (w % 2);
        unpadded_bytes_per_row = width * 4;
        uint32_t row_padding = (align - unpadded_bytes_per_row % align) % align;
        padded_bytes_per_row = unpadded_bytes_per_row + row_padding;
    }
    uint64_t linear_size() const { return (uint64_t)padded_bytes_per_row * height; }
};

// ---- glam 0.20.5 pieces used by src/camera.rs (column-major Mat4) --------------------------------
struct Vec3 { float x, y, z; };
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(float s, Vec3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 v) { float l = std::sqrt(dot(v, v)); return {v.x / l, v.y / l, v.z / l}; }

struct Mat4 {
    float m[16];  // column-major: m[c*4 + r]
    static Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
        Vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
        return {{s.x, u.x, -f.x, 0, s.y, u.y, -f.y, 0, s.z, u.z, -f.z, 0, -dot(s, eye), -dot(u, eye), dot(f, eye), 1}};
    }
    static Mat4 perspective_rh(float fovy, float aspect, float z_near, float z_far) {  // depth 0..1
        float sn = std::sin(0.5f * fovy), cs = std::cos(0.5f * fovy);
        float h = cs / sn, w = h / aspect, r = z_far / (z_near - z_far);
        return {{w, 0, 0, 0, 0, h, 0, 0, 0, 0, r, -1, 0, 0, r * z_near, 0}};
    }
    Mat4 operator*(const Mat4 &b) const {
        Mat4 o{};
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) {
                float s = m[r] * b.m[c * 4];
                s = s + m[4 + r] * b.m[c * 4 + 1];
                s = s + m[8 + r] * b.m[c * 4 + 2];
                s = s + m[12 + r] * b.m[c * 4 + 3];
                o.m[c * 4 + r] = s;
            }
        return o;
    }
    // General inverse by cofactors, binary32, products and sums left to right as written: the one order every host
    // of this build uses (vokselis_amd/camera.py writes the same expressions), so that the same orbit gives the
    // same 144 bytes everywhere.  glam's own SIMD order cannot be reproduced offline.
    Mat4 inverse() const {
        float a[16];
        a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
        a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
        a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
        a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
        a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
        a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
        a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
        a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
        a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
        a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
        a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
        a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
        a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
        a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
        a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
        a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
        const float det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
        const float rdet = 1.0f / det;
        Mat4 o{};
        for (int i = 0; i < 16; i++) o.m[i] = a[i] * rdet;
        return o;
    }
    static Mat4 identity() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}; }
};

// src/camera.rs:5-21
struct CameraUniform {
    float view_position[4] = {0, 0, 0, 0};
    float proj_view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float inv_proj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
static_assert(sizeof(CameraUniform) == 144, "CameraUniform is 144 bytes (src/camera.rs:5-11)");

// src/camera.rs:74-172
class Camera {
  public:
    static constexpr float ZFAR = 100.f, ZNEAR = 0.1f, FOVY = 3.14159265358979323846f / 2.0f;
    float zoom, pitch, yaw, aspect;
    Vec3 target, eye{0, 0, 0}, up{0, 1, 0};
    bool updated = false;

    Camera(float zoom_, float pitch_, float yaw_, Vec3 target_, float aspect_)
        : zoom(zoom_), pitch(pitch_), yaw(yaw_), aspect(aspect_), target(target_) { fix_eye(); }
    Mat4 build_projection_view_matrix() const {
        return Mat4::perspective_rh(FOVY, aspect, ZNEAR, ZFAR) * Mat4::look_at_rh(eye, target, up);
    }
    void set_zoom(float z) { zoom = std::fmin(std::fmax(z, 0.3f), ZFAR / 2.f); fix_eye(); updated = true; }
    void add_zoom(float d) { set_zoom(zoom + d); }
    void set_pitch(float p) {
        const float eps = 1.1920929e-7f, half_pi = 3.14159265358979323846f / 2.0f;
        pitch = std::fmin(std::fmax(p, -half_pi + eps), half_pi - eps); fix_eye(); updated = true;
    }
    void add_pitch(float d) { set_pitch(pitch + d); }
    void set_yaw(float y) { yaw = y; fix_eye(); updated = true; }
    void add_yaw(float d) { set_yaw(yaw + d); }
    void set_aspect(uint32_t w, uint32_t h) { aspect = (float)w / (float)h; updated = true; }
    CameraUniform get_proj_view_matrix() const {
        CameraUniform u;
        Mat4 pv = build_projection_view_matrix(), inv = pv.inverse();
        u.view_position[0] = eye.x; u.view_position[1] = eye.y; u.view_position[2] = eye.z; u.view_position[3] = 1.0f;
        std::memcpy(u.proj_view, pv.m, 64);
        std::memcpy(u.inv_proj, inv.m, 64);
        return u;
    }

  private:
    void fix_eye() {
        float pc = std::cos(pitch);
        eye = target - zoom * Vec3{std::sin(yaw) * pc, std::sin(pitch), std::cos(yaw) * pc};
    }
};

// src/context/global_ubo.rs:52-81
struct Uniform {
    float pos[3] = {0, 0, 0};
    uint32_t frame = 0;
    float resolution[2] = {1920.0f, 780.f};
    float mouse[2] = {0, 0};
    uint32_t mouse_pressed = 0;
    float time = 0.f;
    float time_delta = 1.f / 60.f;
    float _padding = 0.f;
};
static_assert(sizeof(Uniform) == 48, "Uniform is 48 bytes (src/context/global_ubo.rs:52-65)");

// src/context/hdr_backbuffer.rs:10-11
struct HdrBackBuffer {
    static constexpr uint32_t DEFAULT_W = 1280, DEFAULT_H = 720;
    uint32_t width = DEFAULT_W, height = DEFAULT_H;
    int format = VK_OUT_RGBA16F;
};

// src/utils/frame_counter.rs
struct FrameCounter {
    uint32_t frame_count = 0;
    double accum_time = 0, delta = 1.0 / 60.0;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    float time_delta() const { return (float)delta; }
    double record() {
        auto now = std::chrono::steady_clock::now();
        delta = std::chrono::duration<double>(now - last).count();
        last = now; accum_time += delta; frame_count++;
        return delta;
    }
};

// src/context.rs:38-67,225-249 -- headless
class Context {
  public:
    Camera camera;
    Uniform global_uniform;
    HdrBackBuffer render_backbuffer;
    uint32_t width, height;
    // fuse_present: when the window has the backbuffer's size, RaycastPipeline::record presents from the pass's own epilogue
    // (VK_RENDER_PRESENT) and the render() that follows records nothing: demo.render + context.render (src/lib.rs:178-182) in one launch
    bool fuse_present = false;
    bool pass_presented = false;

    Context(uint32_t w, uint32_t h, const Camera *cam = nullptr, int device = 0, HdrBackBuffer bb = HdrBackBuffer())
        : camera(cam ? *cam : Camera(1.f, 0.5f, 1.f, {0.f, 0.f, 0.f}, (float)w / (float)h)), render_backbuffer(bb), width(w), height(h) {
        int rc = vk_ctx_create(device, &ctx_);
        if (rc != VK_OK) throw Error(rc, vk_last_error(nullptr));
        check(ctx_, vk_backbuffer_resize(ctx_, bb.width, bb.height, bb.format));
        timeline_ = std::chrono::steady_clock::now();
    }
    ~Context() { if (ctx_) vk_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    vk_ctx *handle() const { return ctx_; }

    // Context::update: uniform every frame, camera when `updated` -- and on the first frame (F10)
    void update(const FrameCounter &fc) {
        global_uniform.time = std::chrono::duration<float>(std::chrono::steady_clock::now() - timeline_).count();
        global_uniform.time_delta = fc.time_delta();
        global_uniform.frame = fc.frame_count;
        global_uniform.resolution[0] = (float)width; global_uniform.resolution[1] = (float)height;
        check(ctx_, vk_set_uniform(ctx_, &global_uniform));
        if (camera.updated || first_frame_) {
            CameraUniform cu = camera.get_proj_view_matrix();
            check(ctx_, vk_set_camera(ctx_, &cu));
            camera.updated = false; first_frame_ = false;
        }
    }
    void resize(uint32_t w, uint32_t h) { width = w; height = h; camera.set_aspect(w, h); }  // context.rs:238-249
    void sync() { check(ctx_, vk_ctx_sync(ctx_)); }
    // Frames in flight: the reference's queue runs ahead of the GPU (src/lib.rs:178-194), bounded by the swapchain
    // (get_current_texture, src/context.rs:252).  k surfaces, each on a stream of its own; 1 = one surface (the default).
    void frames_in_flight(uint32_t k) { check(ctx_, vk_ctx_frames_in_flight(ctx_, k)); }
    uint64_t frame_begin() { uint64_t id = 0; check(ctx_, vk_frame_begin(ctx_, &id)); return id; }  // the acquire
    void frame_end() { check(ctx_, vk_frame_end(ctx_)); }                                          // submit + present
    void frame_wait(uint64_t id) { check(ctx_, vk_frame_wait(ctx_, id)); }
    std::pair<std::vector<uint8_t>, ImageDimentions> capture_frame_of(uint64_t id) {
        ImageDimentions dims(width, height, 256);
        std::vector<uint8_t> out(dims.linear_size(), 0);
        check(ctx_, vk_frame_capture(ctx_, id, out.data(), out.size(), nullptr, nullptr, nullptr));
        return {out, dims};
    }
    void capture_frame_into(uint64_t id, std::vector<uint8_t> &out) {  // (the recorder's reused buffer)
        ImageDimentions dims(width, height, 256);
        if (out.size() != dims.linear_size()) out.assign(dims.linear_size(), 0);
        check(ctx_, vk_frame_capture(ctx_, id, out.data(), out.size(), nullptr, nullptr, nullptr));
    }
    std::string get_info() const {
        char name[256]; int cus = 0, is950 = 0; size_t mem = 0;
        check(ctx_, vk_device_info(ctx_, name, sizeof name, &cus, &is950, &mem));
        return std::string("Device name: ") + name + "\nBackend: HIP\nCompute units: " + std::to_string(cus) +
               "\nScreen format: " + (render_backbuffer.format == VK_OUT_RGBA16F ? "Rgba16Float" : "Rgba32Float");
    }
    // Context::render (context.rs:251-297): the present pass -- backbuffer -> ACES + sRGB -> Rgba8 at the
    // window size.  There is no surface to present to on a compute node.
    void render() {
        if (pass_presented) { pass_presented = false; return; }  // (the raycast pass has written the presented image itself)
        check(ctx_, vk_present(ctx_, width, height, 0));
    }
    uint32_t pass_flags(uint32_t flags) {  // what RaycastPipeline adds to a pass's flags under fuse_present
        if (fuse_present && width == render_backbuffer.width && height == render_backbuffer.height && !(flags & VK_RENDER_COUNT)) { pass_presented = true; return flags | VK_RENDER_PRESENT; }
        return flags;
    }
    // capture_frame (context.rs:299-302, screenshot.rs:37-77): the presented Rgba8 frame, rows padded to 256 B
    std::pair<std::vector<uint8_t>, ImageDimentions> capture_frame() {
        ImageDimentions dims(width, height, 256);
        std::vector<uint8_t> out(dims.linear_size(), 0);
        check(ctx_, vk_capture_frame(ctx_, out.data(), out.size(), nullptr, nullptr, nullptr));
        return {out, dims};
    }
    std::vector<float> read_backbuffer_f32() {
        const HdrBackBuffer &bb = render_backbuffer;
        size_t n = (size_t)bb.width * bb.height * 4;
        std::vector<float> out(n);
        if (bb.format == VK_OUT_RGBA32F) {
            check(ctx_, vk_readback(ctx_, out.data(), (size_t)bb.width * 16));
        } else {
            std::vector<uint16_t> h(n);
            check(ctx_, vk_readback(ctx_, h.data(), (size_t)bb.width * 8));
            for (size_t i = 0; i < n; i++) out[i] = half_to_float(h[i]);
        }
        return out;
    }
    static float half_to_float(uint16_t h) {
        uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu, b;
        if (e == 0) { if (!m) b = s; else { int k = -1; do { m <<= 1; k++; } while (!(m & 0x400u)); b = s | ((uint32_t)(112 - k) << 23) | ((m & 0x3ffu) << 13); } }
        else if (e == 31) b = s | 0x7f800000u | (m << 13);
        else b = s | ((e + 112) << 23) | (m << 13);
        float f; std::memcpy(&f, &b, 4); return f;
    }

  private:
    vk_ctx *ctx_ = nullptr;
    std::chrono::steady_clock::time_point timeline_;
    bool first_frame_ = true;
};

// src/context/volume_texture.rs:32-89
class VolumeTexture {
  public:
    uint32_t nx, ny, nz;
    int format;
    // dense x-fastest voxels (index x + nx*(y + ny*z))
    VolumeTexture(Context &ctx, const void *data, uint32_t nx_, uint32_t ny_, uint32_t nz_, int fmt = VK_FMT_R8_UNORM,
                  int layout = VK_LAYOUT_AUTO, const void *data2 = nullptr) : nx(nx_), ny(ny_), nz(nz_), format(fmt) {
        check(ctx.handle(), vk_volume_upload(ctx.handle(), data, data2, nx, ny, nz, fmt, layout));
    }
    // drop-in for the reference's include_bytes!("bonsai_256x256x256_uint8.raw") (absent from the checkout)
    static VolumeTexture from_raw(Context &ctx, const std::string &path, uint32_t nx = 256, uint32_t ny = 256, uint32_t nz = 256) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + path);
        std::vector<char> buf((size_t)nx * ny * nz);
        f.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) throw std::runtime_error(path + ": expected " + std::to_string(buf.size()) + " bytes");
        return VolumeTexture(ctx, buf.data(), nx, ny, nz);
    }
    static VolumeTexture generate(Context &ctx, int kind, uint32_t nx, uint32_t ny, uint32_t nz, int fmt = VK_FMT_R8_UNORM,
                                  uint32_t seed = 0x5EED0001u, uint32_t lo = 20, uint32_t span = 12, int layout = VK_LAYOUT_AUTO) {
        check(ctx.handle(), vk_volume_generate(ctx.handle(), kind, nx, ny, nz, fmt, seed, lo, span, layout));
        return VolumeTexture(nx, ny, nz, fmt);
    }

    // XorCompute::new + record (examples/xor/xor_compute.rs): density + normals from shaders/xor.wgsl
    static VolumeTexture generate_xor(Context &ctx, uint32_t nx = 256, uint32_t ny = 256, uint32_t nz = 256, float time = 0.f) {
        check(ctx.handle(), vk_volume_generate_xor(ctx.handle(), nx, ny, nz, time));
        return VolumeTexture(nx, ny, nz, VK_FMT_RGBA16F_PAIR);
    }

  private:
    VolumeTexture(uint32_t a, uint32_t b, uint32_t c, int f) : nx(a), ny(b), nz(c), format(f) {}
};

// examples/bonsai/raycast.rs (render pipeline) / examples/xor/raycast.rs (compute single + tile)
struct RaycastPipeline {
    int mode = VK_MODE_NAIVE_TRILINEAR;
    float dt_scale = 1.0f;
    uint32_t flags = 0;
    void record(Context &ctx) const {
        const HdrBackBuffer &bb = ctx.render_backbuffer;
        check(ctx.handle(), vk_render(ctx.handle(), mode, 0, 0, bb.width, bb.height, dt_scale, ctx.pass_flags(flags)));
    }
    void record_tile(Context &ctx, int32_t x, int32_t y, uint32_t w, uint32_t h) const {
        check(ctx.handle(), vk_render(ctx.handle(), mode, x, y, w, h, dt_scale, ctx.pass_flags(flags)));
    }
};

// trait Demo (src/lib.rs:37-43).  `init` is the static constructor D::init(Context&).
struct Demo {
    virtual ~Demo() = default;
    virtual void update(Context &) {}
    virtual void render(Context &) {}
};

// run::<D> (src/lib.rs:45-208) without the window: Context::update -> Demo::update -> Demo::render.
// in_flight > 1: the loop runs up to that many frames ahead of the GPU, every frame on a surface of its own.
// on_frame(ctx, id), if given, runs after each frame has been submitted -- the place of the reference's recorder
// (`if recording_status { context.capture_frame() ... }`, src/lib.rs:196-199), which with frames in flight captures a frame a few ids back.
struct NoFrameHook { void operator()(Context &, uint64_t) const {} };
template <class D, class OnFrame = NoFrameHook>
std::unique_ptr<D> run_headless(Context &ctx, uint32_t frames, double *mean_frame_ms = nullptr, uint32_t in_flight = 1, OnFrame on_frame = OnFrame()) {
    FrameCounter fc;
    if (in_flight > 1) ctx.frames_in_flight(in_flight);
    std::unique_ptr<D> demo = D::init(ctx);
    ctx.sync();  // (volume set-up is not frame time)
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < frames; i++) {
        ctx.update(fc);
        demo->update(ctx);
        fc.record();
        const uint64_t id = ctx.frame_begin();
        demo->render(ctx);
        ctx.render();  // src/lib.rs:178-182: demo.render, then context.render (present)
        ctx.frame_end();
        on_frame(ctx, id);
    }
    ctx.sync();
    if (mean_frame_ms) *mean_frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (frames ? frames : 1);
    return demo;
}

}  // namespace vokselis
