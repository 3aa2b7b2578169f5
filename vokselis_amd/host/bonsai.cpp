// bonsai.cpp -- headless counterpart of `cargo run --example bonsai` (examples/bonsai/main.rs):
// the same Demo (volume + raycast pipeline), the same camera, rendered N frames into the backbuffer.
//   bonsai [--frames N] [--size WxH] [--dt S] [--raw bonsai_256x256x256_uint8.raw] [--ppm out.ppm]
//          [--f32] [--dump-rgba file] [--dump-steps file]   parity surface (rgba32f) + per-pixel trip counts, raw
//          [--gpus N] [--batch B] [--peer-direct]            the frame's tiles over N GPUs of this node (vk_group_*); --peer-direct: the GPUs
//                                                            store into GPU 0's frames themselves instead of gather + un-tile
//          [--in-flight K] [--orbit] [--fuse-present] [--record]                        K frames in flight (vk_ctx_frames_in_flight); --orbit: the camera turns by 2 pi / 1024 every frame;
//                                                            --record: capture_frame of every frame, K - 1 frames behind (the recorder, src/lib.rs:196-199)
//          [--camera-blobs orbits.txt out.bin]              no GPU: one 144-byte CameraUniform per "zoom pitch yaw tx ty tz aspect" line
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>

#include "vokselis.hpp"

using namespace vokselis;

static std::string g_raw;
static float g_dt = 1.0f;
static bool g_orbit = false;

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
    void update(Context &ctx) override { if (g_orbit) ctx.camera.add_yaw(6.28318f / 1024.f); }  // (a mouse drag: src/lib.rs:166-171)
    void render(Context &ctx) override { pipeline.record(ctx); }  // examples/bonsai/main.rs:27-57
};

// The frame's tiles over the node's GPUs (examples/xor/main.rs:235-254 generalised): one context per GPU inside a
// vk_group, the volume replicated, `batch` frames per launch / gather / un-tile.
static int run_group(int n_gpus, uint32_t frames, uint32_t batch, uint32_t w, uint32_t h, bool peer_direct) {
    std::vector<int> ords(n_gpus);
    for (int i = 0; i < n_gpus; i++) ords[i] = i;
    vk_group *g = nullptr;
    if (vk_group_create(n_gpus, ords.data(), &g) != VK_OK) { std::fprintf(stderr, "bonsai: vk_group_create: %s\n", vk_last_error(nullptr)); return 1; }
    int rc = 1;
    try {
        std::vector<char> raw;
        if (!g_raw.empty()) {
            std::ifstream f(g_raw, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + g_raw);
            raw.resize((size_t)256 * 256 * 256);
            f.read(raw.data(), (std::streamsize)raw.size());
            if ((size_t)f.gcount() != raw.size()) throw std::runtime_error(g_raw + ": expected 16777216 bytes");
        }
        for (int i = 0; i < n_gpus; i++) {
            vk_ctx *c = vk_group_ctx(g, i);
            check(c, vk_backbuffer_resize(c, w, h, VK_OUT_RGBA16F));
            if (!raw.empty()) check(c, vk_volume_upload(c, raw.data(), nullptr, 256, 256, 256, VK_FMT_R8_UNORM, VK_LAYOUT_AUTO));
            else check(c, vk_volume_generate(c, VK_GEN_BONSAI_STANDIN, 256, 256, 256, VK_FMT_R8_UNORM, 0x5EED0001u, 0, 1, VK_LAYOUT_AUTO));
        }
        vk_ctx *root = vk_group_ctx(g, 0);
        check(root, vk_partition_wire(root, VK_WIRE_RGB));  // tiles travel as colour only: alpha is 1 in every pixel (6 bytes instead of 8 per link and pixel)
        // --peer-direct: every GPU's march stores its tiles straight into GPU 0's frames over xGMI (no gather, no un-tile)
        if (peer_direct && vk_group_peer_direct(g, 1) != VK_OK) throw std::runtime_error(vk_group_last_error(g));
        Camera camera(1.f, 0.5f, 1.f, {0.5f, 0.5f, 0.5f}, (float)w / (float)h);
        // every frame of a launch its own camera: consecutive frames of an orbit (yaw step 2 pi / 1024), as bench.py's headline marches them
        std::vector<CameraUniform> cams;
        for (uint32_t j = 0; j < batch; j++) cams.push_back(Camera(1.f, 0.5f, 1.f + 6.28318f * (float)j / 1024.f, {0.5f, 0.5f, 0.5f}, (float)w / (float)h).get_proj_view_matrix());
        (void)camera;
        void *out = nullptr;
        check(root, vk_device_alloc(root, (size_t)batch * w * h * 8, &out));
        auto launch = [&]() { if (vk_group_render(g, VK_MODE_NAIVE_TRILINEAR, batch, cams.data(), 64, g_dt, 0, out) != VK_OK) throw std::runtime_error(vk_group_last_error(g)); };
        launch();
        if (vk_group_sync(g) != VK_OK) throw std::runtime_error(vk_group_last_error(g));
        const uint32_t n_batches = (frames + batch - 1) / batch;
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t b = 0; b < n_batches; b++) launch();
        if (vk_group_sync(g) != VK_OK) throw std::runtime_error(vk_group_last_error(g));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (n_batches * batch);
        std::vector<uint16_t> first((size_t)w * h * 4);
        check(root, vk_device_download(root, first.data(), out, first.size() * 2));
        uint64_t sum = 0;
        for (uint16_t v : first) sum += v;
        std::printf("tiles: %s\n", peer_direct ? "peer-direct stores into GPU 0's frames" : "gathered over RCCL, un-tiled on GPU 0");
        std::printf("GPUs %d, %u frames per launch\nAvg frame time %.4fms over %u frames\nframe 0 half-word sum %llu\n", n_gpus, batch, ms, n_batches * batch,
                    (unsigned long long)sum);
        check(root, vk_device_free(root, out));
        rc = 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "bonsai: %s\n", e.what());
    }
    vk_group_destroy(g);
    return rc;
}

// `zoom pitch yaw tx ty tz aspect` per line -> 144 bytes per line (no GPU involved)
static int dump_camera_blobs(const std::string &in, const std::string &out) {
    std::ifstream f(in);
    std::ofstream o(out, std::ios::binary);
    if (!f || !o) { std::fprintf(stderr, "bonsai: cannot open %s / %s\n", in.c_str(), out.c_str()); return 1; }
    float z, p, y, tx, ty, tz, a;
    while (f >> z >> p >> y >> tx >> ty >> tz >> a) {
        CameraUniform u = Camera(z, p, y, {tx, ty, tz}, a).get_proj_view_matrix();
        o.write(reinterpret_cast<const char *>(&u), sizeof u);
    }
    return 0;
}

int main(int argc, char **argv) {
    uint32_t frames = 100, w = 1280, h = 720, batch = 8, in_flight = 1;
    int gpus = 0;
    bool f32 = false, peer_direct = false, fuse_present = false, record = false;
    std::string ppm, dump_rgba, dump_steps;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--size") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--dt") g_dt = (float)std::atof(next());
        else if (a == "--raw") g_raw = next();
        else if (a == "--ppm") ppm = next();
        else if (a == "--f32") f32 = true;
        else if (a == "--dump-rgba") dump_rgba = next();
        else if (a == "--dump-steps") dump_steps = next();
        else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--peer-direct") peer_direct = true;
        else if (a == "--in-flight") in_flight = (uint32_t)std::max(1, std::atoi(next()));
        else if (a == "--orbit") g_orbit = true;
        else if (a == "--fuse-present") fuse_present = true;
        else if (a == "--record") record = true;
        else if (a == "--batch") batch = (uint32_t)std::max(1, std::atoi(next()));
        else if (a == "--camera-blobs") { std::string in = next(); return dump_camera_blobs(in, next()); }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (gpus > 0) return run_group(gpus, frames, batch, w, h, peer_direct);
    try {
        // examples/bonsai/main.rs:64-74: 1280x720 window, Camera::new(1., 0.5, 1., (0.5,0.5,0.5), w/h)
        Camera camera(1.f, 0.5f, 1.f, {0.5f, 0.5f, 0.5f}, (float)w / (float)h);
        HdrBackBuffer bb; bb.width = w; bb.height = h;
        if (f32) bb.format = VK_OUT_RGBA32F;
        Context ctx(w, h, &camera, 0, bb);
        ctx.fuse_present = fuse_present;  // (the present pass in the raycast pass's epilogue: VK_RENDER_PRESENT)
        std::printf("%s\n", ctx.get_info().c_str());
        double ms = 0;
        // --record: the reference's recorder takes capture_frame() of every frame (src/lib.rs:196-199); with frames in flight it takes the frame
        // K - 1 ids back -- finished or nearly so -- while the newer ones execute
        std::vector<uint8_t> rec;
        uint64_t recorded = 0, rec_sum = 0;
        auto recorder = [&](Context &c, uint64_t id) {
            if (!record || id < in_flight) return;
            c.capture_frame_into(id - (in_flight - 1), rec);
            recorded++;
            rec_sum += rec[rec.size() / 2];
        };
        auto demo = run_headless<Bonsai>(ctx, frames, &ms, in_flight, recorder);
        std::printf("Avg frame time %.4fms over %u frames (%u in flight)\n", ms, frames, in_flight);  // src/utils/frame_counter.rs:23-24
        if (record) std::printf("recorded %llu frames (probe byte sum %llu)\n", (unsigned long long)recorded, (unsigned long long)rec_sum);
        auto shot = ctx.capture_frame();
        uint64_t sum = 0;
        for (uint8_t b : shot.first) sum += b;
        std::printf("capture_frame: %ux%u, padded row %u B, byte sum %llu\n", shot.second.width, shot.second.height,
                    shot.second.padded_bytes_per_row, (unsigned long long)sum);
        if (!dump_rgba.empty() || !dump_steps.empty()) {
            // the hot path's own output (before the present pass), with the kernel's per-pixel trip counts
            RaycastPipeline counted = demo->pipeline;
            counted.flags |= VK_RENDER_COUNT;
            counted.record(ctx);
            if (!dump_rgba.empty()) {
                std::vector<float> px = ctx.read_backbuffer_f32();
                std::ofstream(dump_rgba, std::ios::binary).write(reinterpret_cast<const char *>(px.data()), (std::streamsize)(px.size() * 4));
            }
            if (!dump_steps.empty()) {
                std::vector<uint32_t> st((size_t)w * h);
                check(ctx.handle(), vk_readback_steps(ctx.handle(), st.data()));
                std::ofstream(dump_steps, std::ios::binary).write(reinterpret_cast<const char *>(st.data()), (std::streamsize)(st.size() * 4));
            }
        }
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
