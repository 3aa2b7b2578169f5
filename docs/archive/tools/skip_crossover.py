"""The cost of the skip machinery against the share of skippable cells: 256^3 u8 @1920x1080 (C2's shape), fog 26..40 with
16^3 blocks knocked out with probability p.  Columns: no-skip, probing on every trip, the adaptive default."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

W, H, n = 1920, 1080, 256
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)


def holes(p, seed=3, block=16):
    rng = np.random.default_rng(seed)
    vol = rng.integers(26, 41, (n, n, n), dtype=np.uint8)
    nb = n // block
    m = rng.random((nb, nb, nb)) < p
    vol[np.repeat(np.repeat(np.repeat(m, block, 0), block, 1), block, 2)] = 10
    return vol


def timeit(flags, iters=30):
    p = V.RaycastPipeline(dt_scale=0.5, flags=flags)
    for _ in range(3): p.record(ctx)
    ctx.sync(); ctx.timer_begin()
    for _ in range(iters): p.record(ctx)
    ctx.timer_end()
    return ctx.timer_elapsed_ms() / iters


cases = [("fog 20..31 (census 0 %)", None)] + [("holes p=%.2f" % p, p) for p in (0.0, 0.05, 0.1, 0.2, 0.4, 0.6, 0.8)] + [("bonsai stand-in", "standin")]
for name, p in cases:
    if p is None:
        V.VolumeTexture.generate_fog(ctx, (n,) * 3)
    elif p == "standin":
        V.VolumeTexture.generate_standin(ctx, (n,) * 3)
    else:
        V.VolumeTexture(ctx, holes(p))
    ctx.update()
    ef = V.native.C.c_double()
    V.native.check(ctx.handle, V.native.lib().vk_volume_empty_fraction(ctx.handle, V.native.C.byref(ef)))
    ctx.reset_step_counts()
    V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_FORCE_SKIP).record(ctx)
    s_ref, s_ad = ctx.step_counts()
    ctx.reset_step_counts()
    V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS).record(ctx)
    _, s_ex = ctx.step_counts()
    t_ns, t_al, t_ad, t_auto = timeit(V.RENDER_NO_SKIP), timeit(V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS), timeit(V.RENDER_FORCE_SKIP), timeit(0)
    print(json.dumps({"volume": name, "empty_cells": round(ef.value, 3), "noskip_ms": round(t_ns, 4), "probe_always_ms": round(t_al, 4), "adaptive_ms": round(t_ad, 4),
                      "default_ms": round(t_auto, 4), "adaptive_vs_noskip": round(t_ad / t_ns, 3), "s_ref": s_ref, "s_sampled_exact": s_ex, "s_sampled_adaptive": s_ad}), flush=True)
ctx.close()
