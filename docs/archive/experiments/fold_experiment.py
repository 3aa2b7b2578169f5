"""Folded launch of the cell kernels (one wave per hardware slot, each marching blocks w, 2F-1-w, 2F+w, ... of the
heaviest-first order) against one block per wave: single-frame launch time on C2's frame, every frame checked bitwise."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)


def run(p, iters=50, warm=20):
    for _ in range(warm): p.record(ctx)
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(iters): p.record(ctx)
        ctx.timer_end()
        best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


folds = [int(a) for a in sys.argv[1:]] or [0, 8192, 4096, 6144, 2048, 16384]
for vol in ("fog", "standin"):
    (V.VolumeTexture.generate_fog if vol == "fog" else V.VolumeTexture.generate_standin)(ctx, (256,) * 3); ctx.update()
    for kind, flags in (("dense", V.RENDER_NO_SKIP), ("default", 0), ("forced skip", V.RENDER_FORCE_SKIP)):
        ref = None
        for F in folds:
            ctx.set_param("fold_waves", F)
            p = V.RaycastPipeline(dt_scale=0.5, flags=flags)
            ms = run(p)
            img = ctx.read_backbuffer().copy()
            if ref is None: ref = img
            print(json.dumps({"volume": vol, "kernel": kind, "fold_waves": F, "ms": round(ms, 4), "bitwise_equal": bool((img.view(np.uint16) == ref.view(np.uint16)).all())}), flush=True)
ctx.close()
