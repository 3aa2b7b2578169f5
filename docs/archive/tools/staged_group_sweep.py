"""Parameters of the group-window staged march on C5 / C4 (stage_group 1): slab cap, growth period, LDS per wave; the per-wave march at its
defaults is the first line."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

def t(ctx, fn, iters, groups=3, warm=2):
    for _ in range(warm): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
pipe_c = V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT)
it = 3 if which == "c5" else 8
def run(tag):
    ms = t(ctx, lambda: pipe.record(ctx), it)
    sha = hashlib.sha256(ctx.read_backbuffer().tobytes()).hexdigest()[:12]
    ctx.reset_step_counts(); pipe_c.record(ctx); c = ctx.simt_census()
    print(json.dumps({**tag, "ms": round(ms, 4), "sha": sha, "wave_rounds": c["wave_loop_iters"], "mean_T": round(c["wave_sample_execs"] / max(c["wave_loop_iters"], 1), 2)}), flush=True)
run({"case": which, "stage_group": 0})
ctx.set_param("stage_group", 1)
for cap in (0, 6400, 8192, 10240):
    for T0 in (8, 12, 16, 24):
        for ge in (1, 4):
            ctx.set_param("stage_cap_bytes", cap); ctx.set_param("stage_slab_cells", T0); ctx.set_param("stage_grow_every", ge)
            run({"case": which, "stage_group": 1, "cap_per_wave": cap or "auto", "slab_cells": T0, "grow_every": ge})
ctx.close()
