// bonsai.cpp -- headless counterpart of `cargo run --example bonsai` (examples/bonsai/main.rs):
// the same Demo (volume + raycast pipeline), the same camera, rendered N frames into the backbuffer.
//   bonsai [--frames N] [--size WxH] [--dt S] [--raw bonsai_256x256x256_uint8.raw] [--ppm out.ppm]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "vokselis.hpp"

using namespace vokselis;

static std::string g_raw;
static float g_dt = 1.0f;

struct Bonsai : Demo {
    std::unique_ptr<VolumeTexture> volume_texture;
    RaycastPipeline pipeline;
    static std::unique_ptr<Bonsai> init(Context &ctx) {  // examples/bonsai/main.rs:16-25
        auto self = std::make_unique<Bonsai>();
        if (!g_raw.empty()) self->volume_texture = std::make_unique<VolumeTexture>(VolumeTexture::from_raw(ctx, g_raw));
        else self->volume_texture = std::make_unique<VolumeTexture>(VolumeTexture::generate(ctx, VK_GEN_BONSAI_STANDIN, 256, 256, 256));
        self->pipeline = RaycastPipeline{VK_MODE_NAIVE_TRILINEAR, g_dt, 0};
        return self;
    }
    void render(Context &ctx) override { pipeline.record(ctx); }  // examples/bonsai/main.rs:27-57
};

int main(int argc, char **argv) {
    uint32_t frames = 100, w = 1280, h = 720;
    std::string ppm;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--size") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--dt") g_dt = (float)std::atof(next());
        else if (a == "--raw") g_raw = next();
        else if (a == "--ppm") ppm = next();
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        // examples/bonsai/main.rs:64-74: 1280x720 window, Camera::new(1., 0.5, 1., (0.5,0.5,0.5), w/h)
        Camera camera(1.f, 0.5f, 1.f, {0.5f, 0.5f, 0.5f}, (float)w / (float)h);
        HdrBackBuffer bb; bb.width = w; bb.height = h;
        Context ctx(w, h, &camera, 0, bb);
        std::printf("%s\n", ctx.get_info().c_str());
        double ms = 0;
        auto demo = run_headless<Bonsai>(ctx, frames, &ms);
        std::printf("Avg frame time %.4fms over %u frames\n", ms, frames);  // src/utils/frame_counter.rs:23-24
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
        std::fprintf(stderr, "bonsai: %s\n", e.what());
        return 1;
    }
    return 0;
}
