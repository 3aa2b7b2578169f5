"""How long is a single C2 frame's critical path on an otherwise empty machine?  The frame's heaviest 64x64 tile (64 waves, one or two per CU) rendered on
its own, then growing windows around it, against the whole frame -- plain and tolerance walk.  If the lone tile takes as long as the frame, the frame is
its longest waves' dependent chains and nothing else."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N

W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()


def t(fn, iters, groups=3):
    for _ in range(3): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


for walk, fl in (("exact", 0), ("fast_walk", V.RENDER_FAST_WALK)):
    ctx.reset_step_counts()
    V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT | N.RENDER_DEBUG_TRIPS).record(ctx); ctx.sync()
    trips = ctx.read_steps().reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))          # per 8x8 block: the wave's trips
    by, bx = np.unravel_index(np.argmax(trips), trips.shape)
    p = V.RaycastPipeline(dt_scale=0.5, flags=fl)
    for _ in range(300): p.record(ctx)
    whole = t(lambda: p.record(ctx), 50)
    print(json.dumps({"walk": walk, "whole_frame_ms": round(whole, 4), "heaviest_block": [int(bx), int(by)], "its_trips": int(trips[by, bx]),
                      "blocks_ge_100_trips": int((trips >= 100).sum()), "blocks_ge_150": int((trips >= 150).sum())}), flush=True)
    for tw, th in ((8, 8), (64, 64), (128, 128), (256, 256), (512, 512), (1024, 512)):
        x0 = int(np.clip(bx * 8 + 4 - tw // 2, 0, W - tw)) // 8 * 8
        y0 = int(np.clip(by * 8 + 4 - th // 2, 0, H - th)) // 8 * 8
        ms = t(lambda: p.record(ctx, tile=(x0, y0, tw, th)), 50)
        sub = trips[y0 // 8:(y0 + th) // 8, x0 // 8:(x0 + tw) // 8]
        print(json.dumps({"walk": walk, "window": [x0, y0, tw, th], "waves": int(sub.size), "max_trips": int(sub.max()), "sum_trips": int(sub.sum()), "ms": round(ms, 4)}), flush=True)
ctx.close()
