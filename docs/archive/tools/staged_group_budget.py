"""Staged march, one window per 256-thread group: how small may the group's window be?  The per-wave budget (10 KiB for C4) leaves
3-4 waves per SIMD; a group of four waves sharing a 16 x 16 pixel window needs far less LDS per wave for the same slab, so a small
group window buys occupancy (round 5: profiles/r05_ubench_lds_gather.txt -- the fill-wait-march model tops out at ~0.57 VALU occupancy
at 3-4 waves per SIMD).  C4 / C5 single frames; per-wave (stage_group 0, auto) against group windows of cap x 4 bytes.
usage: tools/staged_group_budget.py <c4|c5> [per-wave caps for the group, comma separated]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
caps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2560,3072,4096,5120,6144,8192,10240").split(",")]
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
ctx.update()
ref = None
for rep in range(2):
    for group, cap in [(0, 0)] + [(1, c) for c in caps]:
        ctx.set_param("stage_group", group); ctx.set_param("stage_cap_bytes", cap)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
        cen = ctx.simt_census()
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
        print(json.dumps({"case": which, "group": group, "cap_per_wave": cap or "auto", "group_lds": cap * 4 if group else None, "ms": round(best, 3), "rounds": cen["wave_loop_iters"],
                          "fallback_rounds": cen["wave_skip_iters"], "mean_T": round(cen["wave_sample_execs"] / max(cen["wave_loop_iters"], 1), 2), "bitwise_equal_to_first": same}), flush=True)
        assert same
ctx.close()
