"""C4 / C5 on the staged layout: frame time against the LDS window budget and the slab thickness."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
caps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8192,12288,16384,24576,32768").split(",")]
slabs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "6,8,12").split(",")]
mask = int(sys.argv[4]) if len(sys.argv) > 4 else 7
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005),
                      "c5small": (1024, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
ctx.set_param("stage_copies_mask", mask)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
ctx.update()
for sl in slabs:
    for cap in caps:
        ctx.set_param("stage_cap_bytes", cap); ctx.set_param("stage_slab_cells", sl)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
        s_ref, _ = ctx.step_counts(); cen = ctx.simt_census()
        p = V.RaycastPipeline(dt_scale=0.5)
        p.record(ctx); ctx.sync()
        ctx.timer_begin()
        for _ in range(5): p.record(ctx)
        ctx.timer_end()
        ms = ctx.timer_elapsed_ms() / 5
        print(json.dumps({"case": which, "copies": mask, "slab": sl, "cap": cap, "ms": round(ms, 3), "Gsteps_s": round(s_ref / ms / 1e6, 1), "rounds": cen["wave_loop_iters"],
                          "fallback_rounds": cen["wave_skip_iters"], "mean_T": round(cen["wave_sample_execs"] / max(cen["wave_loop_iters"], 1), 2)}), flush=True)
ctx.close()
