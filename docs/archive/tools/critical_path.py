"""How long does the frame's heaviest 8x8 block take when it runs alone?  (critical path vs throughput)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
def timeit(p, tile, it=100):
    for _ in range(5): p.record(ctx, tile)
    ctx.sync(); ctx.timer_begin()
    for _ in range(it): p.record(ctx, tile)
    ctx.timer_end(); return ctx.timer_elapsed_ms() / it * 1e3
for name, fl in (("skip", 0), ("noskip", V.RENDER_NO_SKIP)):
    pc = V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT)
    ctx.reset_step_counts(); pc.record(ctx); steps = ctx.read_steps()
    p = V.RaycastPipeline(dt_scale=0.5, flags=fl)
    print(f"[{name}] full frame: {timeit(p, None):.1f} us; max steps/ray {steps.max()}")
    # per-8x8-block max steps; heaviest blocks
    blk = steps[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    ys, xs = np.unravel_index(np.argsort(-blk.astype(np.int64), axis=None)[:3], blk.shape)
    for by, bx in zip(ys, xs):
        print(f"   8x8 block at ({bx * 8},{by * 8}) max steps {blk[by, bx]}: alone {timeit(p, (int(bx * 8), int(by * 8), 8, 8)):.1f} us; its 64x64 tile alone {timeit(p, (int(bx * 8) // 64 * 64, int(by * 8) // 64 * 64, 64, 64)):.1f} us")
    print(f"   one empty 8x8 block (0,0): {timeit(p, (0, 0, 8, 8)):.1f} us")
ctx.close()
