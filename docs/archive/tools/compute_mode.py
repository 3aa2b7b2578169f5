"""COMPUTE_NEAREST mode (raycast_compute.wgsl) on the xor volume: time, steps, algorithmic bytes -- with the record kernel's exact
empty-space skipping (round 4) and without it (VK_RENDER_NO_SKIP), single frames and eight frames per launch, interleaved."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)   # examples/xor/main.rs:273-279
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    t0 = time.perf_counter()
    V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update(); ctx.sync()
    setup = time.perf_counter() - t0
    blob = cam.get_proj_view_matrix()
    fr = torch.empty((8, H, W, 4), dtype=torch.float16, device="cuda")
    for name, fl, wm in (("no skip", V.RENDER_NO_SKIP, 4), ("skip >=2", 0, 2), ("skip >=4", 0, 4), ("skip >=8", 0, 8), ("skip >=16", 0, 16), ("no skip", V.RENDER_NO_SKIP, 4), ("skip >=4", 0, 4)):
        ctx.set_param("pair_walk_min", wm)
        ctx.reset_step_counts(); V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=fl | V.RENDER_COUNT).record(ctx); s_ref, s_samp = ctx.step_counts()
        p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=fl)
        for _ in range(200): p.record(ctx)
        ms = t(ctx, lambda: p.record(ctx), 50)
        msb = t(ctx, lambda: V.render_batch(ctx, p, [blob] * 8, fr.data_ptr(), tile_size=64), 8) / 8
        by = s_samp * 16 + W * H * 8
        print(f"{W}x{H} {name:8s}: {ms*1e3:6.1f} us/frame single, {msb*1e3:6.1f} at 8 per launch; steps {s_ref} (fetching {s_samp} = {s_samp/s_ref:.3f}), "
              f"{s_ref/ms/1e6:.1f} G ray-steps/s, algorithmic {by/(ms*1e-3)/1e9:.0f} GB/s = {by/(ms*1e-3)/1e9/8000:.3f} of 8 TB/s (single), volume set-up {setup*1e3:.0f} ms", flush=True)
    ctx.close()
