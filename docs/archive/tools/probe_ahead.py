"""LF_PROBE_AHEAD (vk_march.hpp: AHEAD): the next position's distance byte requested under this trip's sample, against the plain trip.
C2 on the bonsai stand-in, single frames and 64 orbit frames per launch, both walk modes, interleaved; frames must not change by a bit."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

W, H, TS = 1920, 1080, 64
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
frames = torch.empty((64, H, W, 4), dtype=torch.float16, device="cuda")
orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(64)]


def t(fn, iters, groups=3):
    for _ in range(3): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


for walk, fl in (("exact", 0), ("fast_walk", V.RENDER_FAST_WALK)):
    p = V.RaycastPipeline(dt_scale=0.5, flags=fl)
    for _ in range(300): p.record(ctx)
    ref = None
    for rep in range(2):
        for pa in (0, 1):
            ctx.set_param("probe_ahead", pa)
            single = t(lambda: p.record(ctx), 50)
            img = ctx.read_backbuffer().view(np.uint16)
            ref = img if ref is None else ref
            orb = t(lambda: V.render_batch(ctx, p, orbit, frames.data_ptr(), tile_size=TS), 4) / 64
            print(json.dumps({"walk": walk, "probe_ahead": pa, "single_ms": round(single, 4), "orbit64_ms_per_frame": round(orb, 5), "bitwise_equal": bool((img == ref).all())}), flush=True)
ctx.set_param("probe_ahead", 0)
ctx.close()
