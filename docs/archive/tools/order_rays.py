"""Estimate rays per tile of the heaviest-first order (G x G): launch time of a single frame and of batches with one
camera per frame, and the host's cost per camera."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS, B = 1920, 1080, 64, 32
cam0 = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam0, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
nb = 6
orbit = [[V.Camera(1.0, 0.5 + 0.1 * ((i * B + j) % 7) / 7.0, 1.0 + 6.28318 * (i * B + j) / (nb * B), (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)] for i in range(nb)]
for vol in ("standin", "fog"):
    (V.VolumeTexture.generate_fog if vol == "fog" else V.VolumeTexture.generate_standin)(ctx, (256,) * 3); ctx.update()
    pipe = V.RaycastPipeline(dt_scale=0.5)
    for G in (3, 2, 1, 3, 2, 1):
        ctx.set_param("order_rays", G)
        for _ in range(20): pipe.record(ctx)
        ctx.sync(); best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            for _ in range(50): pipe.record(ctx)
            ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / 50)
        V.render_batch(ctx, pipe, orbit[0], frames.data_ptr(), tile_size=TS); ctx.sync()
        h0 = time.perf_counter()
        for cams in orbit[:3]: V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=TS)
        host = (time.perf_counter() - h0) / 3; ctx.sync()
        t0 = time.perf_counter()
        for cams in orbit: V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=TS)
        ctx.sync(); wall = (time.perf_counter() - t0) / (nb * B)
        print(json.dumps({"volume": vol, "rays_per_tile": G * G, "single_frame_ms": round(best, 4), "orbit_batch32_ms_per_frame": round(wall * 1e3, 4), "host_us_per_camera": round(host / B * 1e6, 1)}), flush=True)
ctx.close()
