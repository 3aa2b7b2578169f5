"""Tile order vs frame time for the no-skip kernel (fog and stand-in): heaviest-first, alternating heavy/light,
row-major, lightest-first."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N
W, H = 1920, 1080
for name, gen, flags in (("fog no-skip", lambda c: V.VolumeTexture.generate_fog(c, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS), V.RENDER_NO_SKIP),
                         ("stand-in no-skip", lambda c: V.VolumeTexture.generate_standin(c, (256,) * 3), V.RENDER_NO_SKIP),
                         ("stand-in skip", lambda c: V.VolumeTexture.generate_standin(c, (256,) * 3), 0)):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    gen(ctx); ctx.update()
    pipe = V.RaycastPipeline(dt_scale=0.5, flags=flags)
    def timeit(it=200):
        for _ in range(10): pipe.record(ctx)
        ctx.sync(); ctx.timer_begin()
        for _ in range(it): pipe.record(ctx)
        ctx.timer_end(); return ctx.timer_elapsed_ms() / it
    base = timeit()
    order = ctx.partition_order(64); na = ctx.partition_active(64, 1)[0]
    act, rest = order[:na].copy(), order[na:].copy()
    def apply(o, label):
        o = np.concatenate([o, rest]).astype(np.uint32)
        N.check(ctx.handle, N.lib().vk_debug_set_tile_order(ctx.handle, o.ctypes.data_as(C.POINTER(C.c_uint32)), len(o)))
        print(f"  {label}: {timeit():.4f} ms")
    print(f"{name}: library (heaviest-first + snake) {base:.4f} ms, active tiles {na}")
    # undo the snake to get the plain sorted list
    srt = act.copy()
    for g in range(8, na - 7, 16): srt[g:g + 8] = srt[g:g + 8][::-1]
    half = (na + 1) // 2
    alt = np.empty(na, dtype=srt.dtype); alt[0::2] = srt[:half]; alt[1::2] = srt[half:][::-1][: na - half]
    apply(alt, "alternating heavy / light")
    blk = np.concatenate([srt[i::4] for i in range(4)])   # four interleaved quarters: heavy..light repeated 4 times
    apply(blk, "four heavy-to-light sweeps")
    apply(np.sort(act), "row-major")
    apply(srt[::-1].copy(), "lightest-first")
    apply(srt, "heaviest-first, no snake")
    ctx.close()
