"""Issue priority by ray length (s_setprio at ray set-up): single-frame and batched launch times with and without,
cell kernels on C2's frame and the staged kernel on C4 / C5 (argument: c2 | c4 | c5)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vokselis_amd as V
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
W, H = (3840, 2160) if which == "c5" else (1920, 1080)
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)


def run(fn, iters, warm):
    for _ in range(warm): fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end()
        best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


def ab(name, p, iters, warm, batch=0):
    ref = None
    cams = None
    if batch:
        cams = [cam.get_proj_view_matrix()] * batch
        out = torch.empty((batch, H, W, 4), dtype=torch.float16, device="cuda")
    for prio in ((0, 1, 2, 0, 1, 2) if which != 'c2' else (0, 1, 0, 1)):
        ctx.set_param("wave_prio", prio)
        if batch:
            ms = run(lambda: V.render_batch(ctx, p, cams, out.data_ptr()), iters, warm) / batch
            eq = None
        else:
            ms = run(lambda: p.record(ctx), iters, warm)
            img = ctx.read_backbuffer().copy()
            if ref is None: ref = img
            eq = bool((img.view(np.uint16) == ref.view(np.uint16)).all())
        print(json.dumps({"case": name, "batch": batch or 1, "wave_prio": prio, "ms_per_frame": round(ms, 4), "bitwise_equal": eq}), flush=True)


if which == "c2":
    for vol in ("fog", "standin"):
        (V.VolumeTexture.generate_fog if vol == "fog" else V.VolumeTexture.generate_standin)(ctx, (256,) * 3); ctx.update()
        for kind, flags in (("dense", V.RENDER_NO_SKIP), ("default", 0)):
            ab(f"{vol} {kind}", V.RaycastPipeline(dt_scale=0.5, flags=flags), 50, 20)
            ab(f"{vol} {kind}", V.RaycastPipeline(dt_scale=0.5, flags=flags), 6, 3, batch=8)
else:
    n, fmt, seed = (1024, V.FMT_R16_FLOAT, 0x5EED0004) if which == "c4" else (2048, V.FMT_R8_UNORM, 0x5EED0005)
    V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED); ctx.update()
    ab(which, V.RaycastPipeline(dt_scale=0.5), 10 if which == "c4" else 4, 3)
ctx.close()
