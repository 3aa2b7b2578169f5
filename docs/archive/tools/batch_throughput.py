"""C2 frames per launch: one launch spanning B frames (vk_render_batch) against B single launches, and the
slowest rank of N emulated on this GPU (compact tiles)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS = 1920, 1080, 64
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
blob = cam.get_proj_view_matrix()
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_begin()
    for _ in range(iters): fn()
    ctx.timer_end()
    return ctx.timer_elapsed_ms() / iters

print(json.dumps({"single_launch_ms": round(timeit(lambda: pipe.record(ctx), 100), 4)}))
cap = V.partition_slots(W, H, TS, 1)
for B in (2, 4, 8, 16, 32):
    frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
    ms = timeit(lambda: V.render_batch(ctx, pipe, [blob] * B, frames.data_ptr(), tile_size=TS))
    print(json.dumps({"batch": B, "whole_frames_ms_per_frame": round(ms / B, 4)}))
for B in (16, 32):
    frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
    for nr in (1, 2, 4, 8):
        for k in ((0,) if nr == 1 else (0, 2, 3, 4, 6)):
            ctx.set_root_skip(k)
            capk = V.partition_slots(W, H, TS, nr, k)
            buf = torch.empty((capk, B, TS, TS, 4), dtype=torch.float16, device="cuda")
            gathered = torch.empty((nr, capk, B, TS, TS, 4), dtype=torch.float16, device="cuda")
            per_rank = []
            for r in range(nr):
                per_rank.append(timeit(lambda: V.render_batch(ctx, pipe, [blob] * B, buf.data_ptr(), tile_size=TS, rank=r, nranks=nr, compact=True, slot_capacity=capk)))
            bid, act = V.render_batch(ctx, pipe, [blob] * B, buf.data_ptr(), tile_size=TS, rank=0, nranks=nr, compact=True, slot_capacity=capk)
            g2 = gathered[:, :act].contiguous()
            un_full = timeit(lambda: V.untile_batch(ctx, bid, g2.data_ptr(), act, frames.data_ptr()))
            # as the driver calls it: the frame buffer still holds the un-tile of the batch before last (a still camera here)
            un = timeit(lambda: V.untile_batch(ctx, bid, g2.data_ptr(), act, frames.data_ptr(), prev_batch_id=bid))
            root = per_rank[0] + un
            print(json.dumps({"batch": B, "nranks": nr, "root_skip": k, "root_march+untile_us_per_frame": round(root / B * 1e3, 1), "untile_us_per_frame": round(un / B * 1e3, 1), "untile_full_us_per_frame": round(un_full / B * 1e3, 1),
                              "slowest_peer_us_per_frame": round(max(per_rank[1:] or [0]) / B * 1e3, 1), "bound_us_per_frame": round(max(root, max(per_rank)) / B * 1e3, 1)}))
ctx.set_root_skip(0)
ctx.close()
