"""C5 (2048^3 u8, 3840x2160) cut over N = 8 ranks, emulated on one GPU: each rank's compact launch of B frames (consecutive
orbit cameras) timed alone, and the root's un-tile -- the configuration BASELINE names for 8 GPUs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS, n = 3840, 2160, 64, 2048
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=V.FMT_R8_UNORM, seed=0x5EED0005, layout=V.LAYOUT_AUTO)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)

def timeit(fn, iters=3):
    fn(); ctx.sync(); best = 1e9
    for _ in range(2):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

B = 4
blobs = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
whole = timeit(lambda: V.render_batch(ctx, pipe, blobs, frames.data_ptr(), tile_size=TS)) / B
print(json.dumps({"whole_frames_ms_per_frame": round(whole, 3), "frames_per_launch": B}), flush=True)
for nr, k in ((8, 0), (8, 8), (4, 0), (2, 0)):
    ctx.set_root_skip(k)
    cap = V.partition_slots(W, H, TS, nr, k)
    buf = torch.empty((cap, B, TS, TS, 4), dtype=torch.float16, device="cuda")
    per = [timeit(lambda: V.render_batch(ctx, pipe, blobs, buf.data_ptr(), tile_size=TS, rank=r, nranks=nr, compact=True, slot_capacity=cap)) / B for r in range(nr)]
    bid, act = V.render_batch(ctx, pipe, blobs, buf.data_ptr(), tile_size=TS, rank=0, nranks=nr, compact=True, slot_capacity=cap)
    gathered = torch.empty((nr, act, B, TS, TS, 4), dtype=torch.float16, device="cuda")
    un = timeit(lambda: V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr())) / B
    print(json.dumps({"nranks": nr, "root_skip": k, "root_march_ms": round(per[0], 3), "slowest_peer_ms": round(max(per[1:]), 3), "fastest_peer_ms": round(min(per[1:]), 3),
                      "untile_ms": round(un, 3), "bound_ms": round(max(per[0] + un, max(per[1:])), 3), "speedup": round(whole / max(per[0] + un, max(per[1:])), 2),
                      "active_slots": act, "peer_MB_per_frame": round(act * TS * TS * 8 / 1e6, 1)}), flush=True)
    del buf, gathered
ctx.set_root_skip(0)
ctx.close()
