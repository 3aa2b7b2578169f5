"""Dense (no-skip) kernel on C2-fog, single frame: launch time against a cap on the waves per SIMD (LDS padding)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (256,) * 3); ctx.update()
for kind, flags in (("fog dense", V.RENDER_NO_SKIP),):
    for pad in (0, 1024, 2048, 2816, 3712, 5120, 7168):
        ctx.set_param("naive_lds_pad", pad)
        p = V.RaycastPipeline(dt_scale=0.5, flags=flags)
        for _ in range(5): p.record(ctx)
        ctx.sync(); ctx.timer_begin()
        for _ in range(50): p.record(ctx)
        ctx.timer_end()
        lds = 3084 + pad
        print(json.dumps({"case": kind, "lds_per_wave": lds, "waves_per_simd_cap": min(8, (163840 // lds) // 4), "ms": round(ctx.timer_elapsed_ms() / 50, 4)}))
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
for pad in (0, 1024, 2816, 5120):
    ctx.set_param("naive_lds_pad", pad)
    p = V.RaycastPipeline(dt_scale=0.5)
    for _ in range(5): p.record(ctx)
    ctx.sync(); ctx.timer_begin()
    for _ in range(50): p.record(ctx)
    ctx.timer_end()
    print(json.dumps({"case": "stand-in skip", "lds_per_wave": 3084 + pad, "waves_per_simd_cap": min(8, (163840 // (3084 + pad)) // 4), "ms": round(ctx.timer_elapsed_ms() / 50, 4)}))
ctx.close()
