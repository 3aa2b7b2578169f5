"""Wave timeline of one frame: when do the 8x8 blocks start/end, how many are in flight?"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N
W, H = 1920, 1080
flags = V.RENDER_NO_SKIP if (len(sys.argv) > 1 and sys.argv[1] == "noskip") else 0
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS); ctx.update()
pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=0.5, flags=flags | V.RENDER_COUNT)
pipe.record(ctx); ctx.sync()
nb = 30 * 17 * 64
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
pipe.record(ctx); ctx.sync()
buf = np.zeros(nb * 2, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb))
st, en = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
ok = en > 0
t0 = st[ok].min()
st, en = (st[ok] - t0) / 100.0, (en[ok] - t0) / 100.0   # us
dur = en - st
print("blocks traced", ok.sum(), "frame span us", en.max())
print("duration us percentiles 50/90/99/max:", np.percentile(dur, [50, 90, 99, 100]).round(1))
print("start us percentiles 50/90/99/max:", np.percentile(st, [50, 90, 99, 100]).round(1))
edges = np.linspace(0, en.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    print(f"t={mid:7.1f} us  in flight {int(((st <= mid) & (en > mid)).sum()):6d}")
long_ = np.argsort(-dur)[:10]
print("longest blocks: start, end:", [(round(st[i], 1), round(en[i], 1)) for i in long_])
ctx.close()
