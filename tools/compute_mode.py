"""COMPUTE_NEAREST mode (raycast_compute.wgsl) on the xor volume: time, steps, algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V
for W, H in ((1280, 720), (1920, 1080)):
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)   # examples/xor/main.rs:273-279
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0); ctx.update()
    pc = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=V.RENDER_COUNT)
    ctx.reset_step_counts(); pc.record(ctx); s_ref, s_samp = ctx.step_counts()
    p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
    for _ in range(5): p.record(ctx)
    ctx.sync(); ctx.timer_begin()
    for _ in range(50): p.record(ctx)
    ctx.timer_end(); ms = ctx.timer_elapsed_ms() / 50
    by = s_samp * 16 + W * H * 8; ms_s = ms * 1e-3
    print(f"{W}x{H}: {ms*1e3:.1f} us/frame, steps {s_ref} (fetching {s_samp}), {s_ref/ms/1e6:.1f} G steps/s, algorithmic {by/ms_s/1e9:.0f} GB/s = {by/ms_s/1e9/8000:.3f} of 8 TB/s")
    ctx.close()
