"""Skip kernels: the caps on a walk's steps per trip (walk_cap: in a trip in which other lanes sample; walk_cap_all: in a trip in which every lane
walks), re-swept on the round-5 walk (integer step count, three accumulators).  C2, single frame and 128 orbit frames per launch, exact walk; the
frame must not change by a bit.  usage: tools/walk_caps.py [caps] [caps_all]"""
import sys, os, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np, torch
import vokselis_amd as V

W, H, DT, B = 1920, 1080, 0.5, 128
caps = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6,8,10,12,16").split(",")]
caps_all = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,12,16,24").split(",")]


def t(ctx, fn, iters, groups=3):
    for _ in range(2): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
p = V.RaycastPipeline(dt_scale=DT)
for _ in range(300): p.record(ctx)
ref = None
for rep in range(2):
    for c in caps:
        for ca in caps_all:
            if ca < c: continue
            ctx.set_param("walk_cap", c); ctx.set_param("walk_cap_all", ca)
            p.record(ctx); crc = "%08x" % zlib.crc32(ctx.read_backbuffer().tobytes())
            ref = ref or crc
            print(json.dumps({"walk_cap": c, "walk_cap_all": ca, "single_ms": round(t(ctx, lambda: p.record(ctx), 40), 4),
                              "orbit128_ms_per_frame": round(t(ctx, lambda: V.render_batch(ctx, p, orbit, frames.data_ptr(), tile_size=64), 3) / B, 5), "same_frame": crc == ref}), flush=True)
ctx.close()
