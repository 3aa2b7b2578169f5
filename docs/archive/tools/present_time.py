"""Present pass (present.wgsl: bilinear resample + ACES + sRGB + RGBA8) and the root's un-tile: launch time and bytes moved."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V
for (W, H) in ((1920, 1080), (3840, 2160)):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    for bb in ((W, H), (W // 2, H // 2)):
        ctx = V.Context(W, H, cam, backbuffer=bb, out_format=V.OUT_RGBA16F)
        V.VolumeTexture.generate_standin(ctx, (128,) * 3); ctx.update()
        V.RaycastPipeline(dt_scale=0.5).record(ctx)
        for _ in range(10): ctx.render()
        ctx.sync(); best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            for _ in range(50): ctx.render()
            ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / 50)
        alg = bb[0] * bb[1] * 8 + W * H * 4
        print(json.dumps({"present": f"{bb[0]}x{bb[1]} rgba16f -> {W}x{H} rgba8", "us": round(best * 1e3, 2), "algorithmic_GBps": round(alg / best / 1e6, 1)}), flush=True)
        ctx.close()
