"""Depth of the record kernel's request ring (pair_ring: 4, 6 or 8 buffers -- 3, 5, 7 steps in flight): the xor example's frame at 720p and 1080p,
single launches and 8 frames per launch, two interleaved repetitions; frames must not change by a bit."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V


def t(ctx, fn, iters, groups=3):
    for _ in range(5): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


for W, H in ((1280, 720), (1920, 1080)):
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update(); ctx.sync()
    blob = cam.get_proj_view_matrix()
    fr = torch.empty((8, H, W, 4), dtype=torch.float16, device="cuda")
    p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
    for _ in range(200): p.record(ctx)
    ref = None
    for rep in range(2):
        for ring in (4, 42, 6):
            ctx.set_param("pair_ring", ring)
            ms = t(ctx, lambda: p.record(ctx), 50)
            img = ctx.read_backbuffer().view(np.uint16)
            ref = img if ref is None else ref
            msb = t(ctx, lambda: V.render_batch(ctx, p, [blob] * 8, fr.data_ptr(), tile_size=64), 8) / 8
            print(json.dumps({"frame": f"{W}x{H}", "pair_ring": ring, "single_ms": round(ms, 4), "batch8_ms_per_frame": round(msb, 4), "bitwise_equal": bool((img == ref).all())}), flush=True)
    ctx.set_param("pair_ring", 0)
    ctx.close()
