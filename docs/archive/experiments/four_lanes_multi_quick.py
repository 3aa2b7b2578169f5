"""Four lanes per ray (vk_multi.hpp) for the leading tiles of a single-frame launch: C2 single frame against multi_tiles, with a checksum
of the f32 frame and of the per-pixel iteration counts (must not change by a bit).  usage: tools/multi_quick.py [tiles, comma separated]"""
import sys, os, json, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np
import vokselis_amd as V

W, H, DT = 1920, 1080, 0.5
tiles = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,8,16,32,64,128,192").split(",")]
thrs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "110").split(",")]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)


def t(ctx, fn, iters, groups=3):
    for _ in range(3): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
crc = {}
ref = None
for n in tiles:
  for thr in thrs:
    ctx.set_param("multi_tiles", n); ctx.set_param("multi_thr", thr)
    for rep in range(3):  # (the second launch is the first that has a history to act on)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=DT, flags=V.RENDER_COUNT).record(ctx)
        img, steps, sc = ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts()
        crc[(n, thr)] = ("%08x" % zlib.crc32(img.tobytes()), "%08x" % zlib.crc32(steps.tobytes()), int(sc[0]), int(sc[1]))
        ref = ref or crc[(n, thr)]
        if crc[(n, thr)] != ref:
            print("MISMATCH at multi_tiles", n, "thr", thr, "launch", rep, crc[(n, thr)], ref, flush=True)
ctx.close()
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
p = V.RaycastPipeline(dt_scale=DT)
for _ in range(300): p.record(ctx)
for rep in range(2):
    for n in tiles:
      for thr in (thrs if n else thrs[:1]):
        ctx.set_param("multi_tiles", n); ctx.set_param("multi_thr", thr)
        ms = t(ctx, lambda: p.record(ctx), 50)
        print(json.dumps({"multi_tiles": n, "multi_thr": thr, "single_ms": round(ms, 4), "crc": crc[(n, thr)], "crc16f": "%08x" % zlib.crc32(ctx.read_backbuffer().tobytes())}), flush=True)
ctx.close()
