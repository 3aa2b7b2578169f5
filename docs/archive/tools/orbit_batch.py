"""Batched launches with a different camera in every frame (an orbit) against the same camera repeated: the per-camera
host work (tile order, cull rectangle, descriptors) must not show in the frame rate."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS, B = 1920, 1080, 64, 32
ctx = V.Context(W, H, V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H), backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
same = [V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()] * B
n_batches = 8
orbit = [[V.Camera(1.0, 0.5 + 0.1 * ((i * B + j) % 7) / 7.0, 1.0 + 6.28318 * (i * B + j) / (n_batches * B), (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)] for i in range(n_batches)]
for name, batches in (("same camera", [same] * n_batches), ("orbit, every frame its own camera", orbit), ("orbit again (orders memoised?)", orbit)):
    V.render_batch(ctx, pipe, batches[0], frames.data_ptr(), tile_size=TS); ctx.sync()
    t0 = time.perf_counter(); th = 0.0
    for cams in batches:
        h0 = time.perf_counter()
        V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=TS)
        th += time.perf_counter() - h0
    ctx.sync(); el = time.perf_counter() - t0
    # host cost proper: three calls into an idle 4-slot ring (no call waits for the GPU)
    ctx.sync(); h0 = time.perf_counter()
    for cams in batches[:3]:
        V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=TS)
    th3 = (time.perf_counter() - h0) / 3; ctx.sync()
    print(json.dumps({"case": name, "host_ms_per_call_idle_ring": round(th3 * 1e3, 3), "ms_per_frame_wall": round(el / (n_batches * B) * 1e3, 4), "host_ms_per_call": round(th / n_batches * 1e3, 3)}))
ctx.close()
