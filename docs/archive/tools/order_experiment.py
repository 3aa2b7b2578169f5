"""Does a measured-cost tile order beat the geometric (nominal-steps) estimate?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS); ctx.update()
def timeit(p, it=200):
    for _ in range(10): p.record(ctx)
    ctx.sync(); ctx.timer_begin()
    for _ in range(it): p.record(ctx)
    ctx.timer_end(); return ctx.timer_elapsed_ms() / it
pipe = V.RaycastPipeline(dt_scale=0.5)
print("geometric order ms:", timeit(pipe))
# per-pixel executed lookups+samples as cost proxy: use per-pixel steps? need lookups: approximate with the trace durations
pc = V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
pc.record(ctx); ctx.sync()
nb = 30 * 17 * 64
buf = np.zeros(nb * 4, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb))
st, en = buf[0::4].astype(np.int64), buf[1::4].astype(np.int64)
dur = np.where(en > 0, en - st, 0).astype(np.float64)   # per logical block (position-major)
order = ctx.partition_order(64)
cost_by_pos = dur.reshape(-1, 64)
for agg, f in (("sum", cost_by_pos.sum(1)), ("max", cost_by_pos.max(1))):
    new_pos = np.argsort(-f, kind="stable")           # positions sorted by measured cost
    new_order = order[new_pos].astype(np.uint32)
    N.check(ctx.handle, N.lib().vk_debug_set_tile_order(ctx.handle, new_order.ctypes.data_as(C.POINTER(C.c_uint32)), len(new_order)))
    print(f"measured-cost order ({agg}) ms:", timeit(pipe))
rev = order[::-1].astype(np.uint32).copy()
N.check(ctx.handle, N.lib().vk_debug_set_tile_order(ctx.handle, rev.ctypes.data_as(C.POINTER(C.c_uint32)), len(rev)))
print("lightest-first order ms:", timeit(pipe))
ident = np.arange(len(order), dtype=np.uint32)
N.check(ctx.handle, N.lib().vk_debug_set_tile_order(ctx.handle, ident.ctypes.data_as(C.POINTER(C.c_uint32)), len(ident)))
print("row-major order ms:", timeit(pipe))
ctx.close()
