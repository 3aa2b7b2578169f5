"""Staged march: an odd LDS row pitch for the window ("an odd LDS row pitch against the 61 % bank conflicts (C4)").
Rows of an even number of 16-byte pieces get one more piece, so consecutive window rows -- the pixel rows of a wave -- start on
different banks.  C4 / C5 single frames, row_pad 0 / 1, LDS budget auto and fixed; the frame must not change by a bit.
usage: tools/staged_row_pad.py <c4|c5> [caps, comma separated; 0 = auto]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
caps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,8192,10240,12288,16384").split(",")]
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
ctx.update()
ref = None
for cap in caps:
    for pad in (0, 1):
        ctx.set_param("stage_cap_bytes", cap); ctx.set_param("stage_row_pad", pad)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
        s_ref, _ = ctx.step_counts(); cen = ctx.simt_census()
        p = V.RaycastPipeline(dt_scale=0.5)
        for _ in range(3): p.record(ctx)
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            for _ in range(5): p.record(ctx)
            ctx.timer_end()
            best = min(best, ctx.timer_elapsed_ms() / 5)
        img = ctx.read_backbuffer()
        if ref is None:
            ref = img.copy()
        same = bool((img.view(np.uint16) == ref.view(np.uint16)).all())
        print(json.dumps({"case": which, "cap": cap or "auto", "row_pad": pad, "ms": round(best, 3), "rounds": cen["wave_loop_iters"], "fallback_rounds": cen["wave_skip_iters"],
                          "mean_T": round(cen["wave_sample_execs"] / max(cen["wave_loop_iters"], 1), 2), "bitwise_equal_to_first": same}), flush=True)
        assert same
ctx.close()
