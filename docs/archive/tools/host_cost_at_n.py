"""Host cost per camera of vk_render_batch against the GPU time of a rank's share (N = 8 emulated on one GPU: rank r of 8, compact output), with every
frame its own camera (an orbit): the host must not be what a rank of 8 waits for.  Prints, per frames-per-launch, the host's microseconds per camera (wall
time of the call, no synchronisation) and the GPU's microseconds per frame for ranks 0 and 5."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS, N = 1920, 1080, 64, 8
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
ctx.set_root_skip(2)
cap = V.partition_slots(W, H, TS, N, 2)
for B in (32, 64, 208, 256):
    cams = [[V.Camera(1.0, 0.5, 1.0 + 6.28318 * (i * B + j) / 4096, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)] for i in range(6)]
    out = torch.empty((cap, B, TS, TS, 4), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    for rank in (0, 5):
        for c in cams[:2]:
            V.render_batch(ctx, pipe, c, out.data_ptr(), tile_size=TS, rank=rank, nranks=N, compact=True, slot_capacity=cap)
        ctx.sync()
        h0 = time.perf_counter()
        for c in cams:
            V.render_batch(ctx, pipe, c, out.data_ptr(), tile_size=TS, rank=rank, nranks=N, compact=True, slot_capacity=cap)
        host = (time.perf_counter() - h0) / (len(cams) * B) * 1e6
        ctx.sync()
        wall = (time.perf_counter() - h0) / (len(cams) * B) * 1e6
        ctx.timer_begin()
        V.render_batch(ctx, pipe, cams[0], out.data_ptr(), tile_size=TS, rank=rank, nranks=N, compact=True, slot_capacity=cap)
        ctx.timer_end()
        gpu = ctx.timer_elapsed_ms() / B * 1e3
        print(json.dumps({"frames_per_launch": B, "rank": rank, "of": N, "host_us_per_camera": round(host, 2), "gpu_us_per_frame": round(gpu, 2), "wall_us_per_frame_6_launches": round(wall, 2)}), flush=True)
ctx.close()
