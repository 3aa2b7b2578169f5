"""Four lanes per ray on an otherwise empty machine: windows of 512 x 64 and 512 x 128 pixels around the frame's heaviest 8x8 block (8 / 16
tiles: the smallest launches in which the four-lane path can be switched on), one lane per ray against four (multi_thr 0: every block of
the leading tiles), and the trips of the window's longest wave in both modes (COUNT build)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
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


ctx.set_param("multi_tiles", 0)
V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | N.RENDER_DEBUG_TRIPS).record(ctx); ctx.sync()
trips = ctx.read_steps().reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
by, bx = np.unravel_index(np.argmax(trips), trips.shape)
p = V.RaycastPipeline(dt_scale=0.5)
for _ in range(300): p.record(ctx)
for tw, th in ((512, 64), (512, 128), (1024, 256)):
    x0 = int(np.clip(bx * 8 + 4 - tw // 2, 0, W - tw)) // 64 * 64
    y0 = int(np.clip(by * 8 + 4 - th // 2, 0, H - th)) // 64 * 64
    for tiles, thr in ((0, 0), (4096, 0), (4096, 150), (4096, 110)):
        ctx.set_param("multi_tiles", tiles); ctx.set_param("multi_thr", thr)
        ms = t(lambda: p.record(ctx, tile=(x0, y0, tw, th)), 50)
        ctx.reset_step_counts()
        for _ in range(2): V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | N.RENDER_DEBUG_TRIPS).record(ctx, tile=(x0, y0, tw, th))
        ctx.sync()
        tr = ctx.read_steps()[y0:y0 + th, x0:x0 + tw]
        cen = ctx.simt_census()
        print(json.dumps({"window": [x0, y0, tw, th], "multi_tiles": tiles, "multi_thr": thr, "ms": round(ms, 4), "max_lane_trips": int(tr.max()), "wave_trips_total_2_launches": cen["wave_loop_iters"]}), flush=True)
ctx.close()
