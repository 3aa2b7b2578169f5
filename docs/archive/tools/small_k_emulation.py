"""The driver's own invocation is `bench.py --gpus N --steps 20 --warmup 5`: one timed region holds 20 frames.  How
should the N > 1 driver cut those 20 frames into launches?  Emulated on one GPU: each rank's launch of B frames timed
alone (compact tiles, calibrated-looking root_skip), and the root's un-tile of B frames."""
import sys, os, json
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
    for _ in range(3):
        fn()
    ctx.sync(); ctx.timer_begin()
    for _ in range(iters):
        fn()
    ctx.timer_end()
    return ctx.timer_elapsed_ms() / iters


for nr, k in ((2, 6), (4, 3), (8, 2), (8, 0)):
    for B in (20, 10, 7, 5):
        ctx.set_root_skip(k)
        capk = V.partition_slots(W, H, TS, nr, k)
        frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
        buf = torch.empty((capk, B, TS, TS, 4), dtype=torch.float16, device="cuda")
        gathered = torch.empty((nr, capk, B, TS, TS, 4), dtype=torch.float16, device="cuda")
        per_rank = [timeit(lambda: V.render_batch(ctx, pipe, [blob] * B, buf.data_ptr(), tile_size=TS, rank=r, nranks=nr, compact=True, slot_capacity=capk)) for r in range(nr)]
        bid, act = V.render_batch(ctx, pipe, [blob] * B, buf.data_ptr(), tile_size=TS, rank=0, nranks=nr, compact=True, slot_capacity=capk)
        g2 = gathered[:, :act].contiguous()
        un = timeit(lambda: V.untile_batch(ctx, bid, g2.data_ptr(), act, frames.data_ptr()))
        print(json.dumps({"nranks": nr, "root_skip": k, "frames_per_launch": B, "root_march_ms": round(per_rank[0], 4), "slowest_peer_ms": round(max(per_rank[1:]), 4),
                          "untile_ms": round(un, 4), "active_slots": act, "peer_bytes_MB": round(act * B * TS * TS * 8 / 1e6, 2)}), flush=True)
ctx.set_root_skip(0)
ctx.close()
