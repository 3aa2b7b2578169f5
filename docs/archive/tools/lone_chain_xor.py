"""The xor example's compute-mode frame (1280x720, 256^3 records): its heaviest 8x8 block alone on the machine, growing windows, the whole frame."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

for W, H in ((1280, 720),):
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update(); ctx.sync()

    def t(fn, iters, groups=3):
        for _ in range(3): fn()
        ctx.sync(); best = 1e9
        for _ in range(groups):
            ctx.timer_begin()
            for _ in range(iters): fn()
            ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
        return best

    for name, fl in (("skip", 0), ("no skip", V.RENDER_NO_SKIP)):
        ctx.reset_step_counts()
        V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=fl | V.RENDER_COUNT).record(ctx); ctx.sync()
        steps = ctx.read_steps().reshape(H // 8, 8, W // 8, 8)
        smax = steps.max(axis=(1, 3))
        by, bx = np.unravel_index(np.argmax(smax), smax.shape)
        p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=fl)
        for _ in range(300): p.record(ctx)
        whole = t(lambda: p.record(ctx), 50)
        print(json.dumps({"case": name, "whole_frame_ms": round(whole, 4), "heaviest_block": [int(bx), int(by)], "its_longest_ray_steps": int(smax[by, bx]),
                          "mean_steps_per_ray": float(steps.mean())}), flush=True)
        for tw, th in ((8, 8), (64, 64), (256, 256), (512, 512), (1280, 360)):
            x0 = int(np.clip(bx * 8 + 4 - tw // 2, 0, W - tw)) // 8 * 8
            y0 = int(np.clip(by * 8 + 4 - th // 2, 0, H - th)) // 8 * 8
            ms = t(lambda: p.record(ctx, tile=(x0, y0, tw, th)), 50)
            print(json.dumps({"case": name, "window": [x0, y0, tw, th], "waves": (tw // 8) * (th // 8), "ms": round(ms, 4)}), flush=True)
    ctx.close()
