// xor.cpp -- headless counterpart of `cargo run --example xor` (examples/xor/main.rs): the procedural
// volume (shaders/xor.wgsl) marched by the compute raycast (shaders/raycast_compute.wgsl), either as
// one `single` dispatch or as the reference's loop of 256-pixel `tile` dispatches with offsets.
//   xor [--frames N] [--size WxH] [--mode single|tile|procedural] [--volume N] [--time T] [--ppm out.ppm] [--in-flight K] [--fuse-present]
// `procedural` marches shaders/xor.wgsl's density function itself, no volume; --time pins un.time.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "vokselis.hpp"

using namespace vokselis;

enum class Mode { SinglePass, Tile, Procedural };  // examples/xor/main.rs:14-18 (F1 toggles the first two there)
static bool g_pin_time = false;
static float g_time = 0.f;
static Mode g_mode = Mode::SinglePass;
static uint32_t g_vol = 256;
static const uint32_t TILE_SIZE = 256;  // examples/xor/main.rs:12

struct Xor : Demo {
    std::unique_ptr<VolumeTexture> xor_texture;
    RaycastPipeline raycast;
    std::vector<std::pair<int32_t, int32_t>> offsets;
    static std::unique_ptr<Xor> init(Context &ctx) {  // examples/xor/main.rs:41-162
        auto self = std::make_unique<Xor>();
        if (g_mode == Mode::Procedural) {
            self->raycast = RaycastPipeline{VK_MODE_PROCEDURAL, 1.0f, 0};  // no volume at all
            return self;
        }
        self->xor_texture = std::make_unique<VolumeTexture>(VolumeTexture::generate_xor(ctx, g_vol, g_vol, g_vol, 0.f));
        self->raycast = RaycastPipeline{VK_MODE_COMPUTE_NEAREST, 1.0f, 0};
        const HdrBackBuffer &bb = ctx.render_backbuffer;
        for (uint32_t y = 0; y < bb.height / TILE_SIZE + 1; y++)      // :82-92, including the off-screen column/row
            for (uint32_t x = 0; x < bb.width / TILE_SIZE + 1; x++) self->offsets.push_back({(int32_t)(x * TILE_SIZE), (int32_t)(y * TILE_SIZE)});
        return self;
    }
    void update(Context &ctx) override {
        if (g_pin_time) {  // Context::update has just uploaded the clock; a pinned un.time replaces it
            ctx.global_uniform.time = g_time;
            check(ctx.handle(), vk_set_uniform(ctx.handle(), &ctx.global_uniform));
        }
    }
    void render(Context &ctx) override {  // examples/xor/main.rs:210-262
        if (g_mode != Mode::Tile) raycast.record(ctx);
        else for (auto &o : offsets) raycast.record_tile(ctx, o.first, o.second, TILE_SIZE, TILE_SIZE);
    }
};

int main(int argc, char **argv) {
    uint32_t frames = 100, w = 1280, h = 720, in_flight = 1;
    bool fuse_present = false;
    std::string ppm;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--size") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--mode") { std::string m = next(); g_mode = m == "tile" ? Mode::Tile : (m == "procedural" ? Mode::Procedural : Mode::SinglePass); }
        else if (a == "--time") { g_time = (float)std::atof(next()); g_pin_time = true; }
        else if (a == "--volume") g_vol = (uint32_t)std::atoi(next());
        else if (a == "--ppm") ppm = next();
        else if (a == "--fuse-present") fuse_present = true;
        else if (a == "--in-flight") in_flight = (uint32_t)std::atoi(next());  // frames in flight (vk_ctx_frames_in_flight)
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        Camera camera(3.f, -0.5f, 1.f, {0.f, 0.f, 0.f}, (float)w / (float)h);  // examples/xor/main.rs:273-279
        HdrBackBuffer bb; bb.width = w; bb.height = h;
        Context ctx(w, h, &camera, 0, bb);
        ctx.fuse_present = fuse_present;  // (the present pass in the raycast pass's epilogue: VK_RENDER_PRESENT)
        std::printf("%s\n", ctx.get_info().c_str());
        double ms = 0;
        auto demo = run_headless<Xor>(ctx, frames, &ms, in_flight ? in_flight : 1);
        std::printf("Avg frame time %.4fms over %u frames (%s)\n", ms, frames, g_mode == Mode::Tile ? "Tile" : (g_mode == Mode::Procedural ? "Procedural" : "SinglePass"));
        auto shot = ctx.capture_frame();
        uint64_t sum = 0;
        for (uint8_t b : shot.first) sum += b;
        std::printf("capture_frame: %ux%u, padded row %u B, byte sum %llu\n", shot.second.width, shot.second.height,
                    shot.second.padded_bytes_per_row, (unsigned long long)sum);
        if (!ppm.empty()) {
            std::ofstream f(ppm, std::ios::binary);
            f << "P6\n" << shot.second.width << " " << shot.second.height << "\n255\n";
            for (uint32_t y = 0; y < shot.second.height; y++)
                for (uint32_t x = 0; x < shot.second.width; x++)
                    f.write((const char *)&shot.first[(size_t)y * shot.second.padded_bytes_per_row + x * 4], 3);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "xor: %s\n", e.what());
        return 1;
    }
    return 0;
}
