"""C2 in a few seconds: single frame and 128 orbit frames per launch, exact and tolerance walk, with a checksum of the f32 frame and
the step counters -- the A/B probe for kernel experiments (tools/ab.py runs it against library variants, interleaved).
usage: tools/c2_quick.py [--reps N] [--no-fast]"""
import sys, os, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np, torch
import vokselis_amd as V

W, H, DT, B = 1920, 1080, 0.5, 128
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
modes = [("exact", 0)] + ([] if "--no-fast" in sys.argv else [("fast", V.RENDER_FAST_WALK)])


def t(ctx, fn, iters, groups=3):
    for _ in range(2): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
out = {"lib": os.environ.get("VK_LIB", "product")}
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
for name, fl in modes:
    ctx.reset_step_counts()
    V.RaycastPipeline(dt_scale=DT, flags=fl | V.RENDER_COUNT).record(ctx)
    img, steps, sc = ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts()
    out[name + "_crc"] = "%08x" % zlib.crc32(img.tobytes())
    out[name + "_steps_crc"] = "%08x" % zlib.crc32(steps.tobytes())
    out[name + "_s_ref"], out[name + "_s_sampled"] = int(sc[0]), int(sc[1])
ctx.close()
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
pipes = {name: V.RaycastPipeline(dt_scale=DT, flags=fl) for name, fl in modes}
for _ in range(300): pipes["exact"].record(ctx)
for rep in range(reps):
    for name, p in pipes.items():
        out.setdefault(name + "_single_ms", []).append(round(t(ctx, lambda: p.record(ctx), 50), 4))
        out.setdefault(name + "_orbit128_ms_per_frame", []).append(round(t(ctx, lambda: V.render_batch(ctx, p, orbit, frames.data_ptr(), tile_size=64), 3) / B, 5))
ctx.close()
print(json.dumps(out), flush=True)
