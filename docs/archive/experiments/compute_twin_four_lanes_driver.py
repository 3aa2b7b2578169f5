"""Compute twin (raycast_compute.wgsl) with four lanes per ray in single-frame launches (pair_quad) against the one-lane record kernel: the xor
example's frame at 720p and 1080p, other cameras; f32 frame and per-pixel iteration counts must not change by a bit; single-frame times."""
import sys, os, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np
import vokselis_amd as V


def t(ctx, fn, iters, groups=3):
    for _ in range(5): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


for (W, H) in ((1280, 720), (1920, 1080)):
    for cam_args in ((3.0, -0.5, 1.0), (2.04, -0.401, 4.556), (1.2, 0.3, 2.0)):
        cam = V.Camera(cam_args[0], cam_args[1], cam_args[2], (0.0, 0.0, 0.0), W / H)
        res = {}
        ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update()
        for q in (0, 1):
            ctx.set_param("pair_quad", q)
            ctx.reset_step_counts()
            V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=V.RENDER_COUNT).record(ctx)
            img, steps, sc = ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts()
            res[q] = ("%08x" % zlib.crc32(img.tobytes()), "%08x" % zlib.crc32(steps.tobytes()), int(sc[0]), int(sc[1]))
            V.RaycastPipeline(V.MODE_COMPUTE_NEAREST).record(ctx)
            res[q] += ("%08x" % zlib.crc32(ctx.read_backbuffer().tobytes()),)
        ctx.close()
        ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
        V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update()
        p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
        for _ in range(100): p.record(ctx)
        ms = {}
        for rep in range(2):
            for q in (0, 1):
                ctx.set_param("pair_quad", q)
                ms.setdefault(q, []).append(round(t(ctx, lambda: p.record(ctx), 40), 4))
        ctx.close()
        print(json.dumps({"size": [W, H], "camera": cam_args, "one_lane_ms": ms[0], "four_lanes_ms": ms[1], "bitwise_equal": res[0] == res[1], "one_lane": res[0], "four_lanes": res[1]}), flush=True)
