"""Staged march: how often should the slab search try a slab one cell thicker than the last one that fitted?  (stage_grow_every: 1 = every round,
round 2's policy.)  C4 / C5 single frames; the frame must not change by a bit.  usage: tools/staged_grow.py <c4|c5> [periods]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
periods = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
ref = None
for camname, cam in (("bonsai", V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)), ("diagonal", V.Camera(1.2, 0.6, 0.8, (0.5, 0.5, 0.5), W / H)), ("far", V.Camera(2.5, 0.3, 2.0, (0.5, 0.5, 0.5), W / H))):
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
    ctx.update()
    ref = None
    for g in periods:
        ctx.set_param("stage_grow_every", g)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
        cen = ctx.simt_census()
        p = V.RaycastPipeline(dt_scale=0.5)
        for _ in range(3): p.record(ctx)
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            for _ in range(4): p.record(ctx)
            ctx.timer_end()
            best = min(best, ctx.timer_elapsed_ms() / 4)
        img = ctx.read_backbuffer()
        ref = img.copy() if ref is None else ref
        same = bool((img.view(np.uint16) == ref.view(np.uint16)).all())
        print(json.dumps({"case": which, "camera": camname, "grow_every": g, "ms": round(best, 3), "rounds": cen["wave_loop_iters"], "mean_T": round(cen["wave_sample_execs"] / max(cen["wave_loop_iters"], 1), 2), "bitwise": same}), flush=True)
        assert same
    ctx.close()
