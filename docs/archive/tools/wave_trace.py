"""Wave timeline of one frame: when do the 8x8 blocks start/end, how many are in flight?"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N
W, H = 1920, 1080
flags = V.RENDER_NO_SKIP if ("noskip" in sys.argv[1:]) else 0
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
if "fog" in sys.argv[1:]:
    V.VolumeTexture.generate_fog(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS)
else:
    V.VolumeTexture.generate_standin(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS)
ctx.update()
pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=0.5, flags=flags | V.RENDER_COUNT)
pipe.record(ctx); ctx.sync()
nb = 30 * 17 * 64
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
pipe.record(ctx); ctx.sync()
buf = np.zeros(nb * 4, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb))
st, en, where, work = buf[0::4].astype(np.int64), buf[1::4].astype(np.int64), buf[2::4], buf[3::4]
ok = en > 0
where = where[ok]
work = work[ok]
t0 = st[ok].min()
st, en = (st[ok] - t0) / 100.0, (en[ok] - t0) / 100.0   # us
dur = en - st
print("blocks traced", ok.sum(), "frame span us", en.max())
print("duration us percentiles 50/90/99/max:", np.percentile(dur, [50, 90, 99, 100]).round(1))
print("start us percentiles 50/90/99/max:", np.percentile(st, [50, 90, 99, 100]).round(1))
print("waves resident per SIMD at start of each block: computed below from placement")
edges = np.linspace(0, en.max(), 31)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    print(f"t={mid:7.1f} us  in flight {int(((st <= mid) & (en > mid)).sum()):6d}")
long_ = np.argsort(-dur)[:10]
print("longest blocks: start, end:", [(round(st[i], 1), round(en[i], 1)) for i in long_])
hw = (where & np.uint64(0xffffffff)).astype(np.int64); xcc = (where >> np.uint64(32)).astype(np.int64) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk, inv = np.unique(key, return_inverse=True)
load = np.bincount(inv, weights=dur)
heavy = np.bincount(inv, weights=(dur > np.percentile(dur, 90)).astype(float))
full = buf[2::4]
print("placement of the first logical blocks (lb: xcc se sh cu simd wave):")
for lb in list(range(0, 40)) + list(range(64, 72)) + list(range(512, 520)):
    w_ = int(full[lb]); h_ = w_ & 0xffffffff
    print(lb, (w_ >> 32) & 0xf, (h_ >> 13) & 7, (h_ >> 12) & 1, (h_ >> 8) & 15, (h_ >> 4) & 3, h_ & 15, end=" | ")
print()
print("distinct SIMDs seen", len(uk), "xcc values", np.unique(xcc), "se", np.unique(se), "cu", np.unique(cu))
print("per-SIMD summed wave time: mean %.0f  p50 %.0f  p90 %.0f  max %.0f   (frame span %.0f)" % (load.mean(), np.percentile(load, 50), np.percentile(load, 90), load.max(), en.max()))
print("heavy (top-10%%) waves per SIMD: mean %.2f max %d  hist %s" % (heavy.mean(), heavy.max(), np.bincount(heavy.astype(int)).tolist()))
wo = (work & np.uint64(0xfffff)).astype(np.float64); wi = ((work >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64); ws = (work >> np.uint64(40)).astype(np.float64)
est = wo * 27 + wi * 5 + ws * 33          # issue-slot estimate per wave
wl = np.bincount(inv, weights=est)
print("per-SIMD estimated work (issue slots): mean %.0f p50 %.0f p90 %.0f max %.0f  -> max/mean %.2f" % (wl.mean(), np.percentile(wl, 50), np.percentile(wl, 90), wl.max(), wl.max() / wl.mean()))
cu_key = key // 4; ucu, icu = np.unique(cu_key, return_inverse=True); wcu = np.bincount(icu, weights=est)
print("per-CU estimated work: mean %.0f max %.0f min %.0f -> max/mean %.2f" % (wcu.mean(), wcu.max(), wcu.min(), wcu.max() / wcu.mean()))
xk = xcc; wx = np.bincount(xk, weights=est); print("per-XCD estimated work:", (wx / wx.mean()).round(3).tolist())
percu = np.bincount((key // 4).astype(np.int64) - (key // 4).min(), weights=dur); percu = percu[percu > 0]
print("per-CU summed wave time: mean %.0f max %.0f min %.0f" % (percu.mean(), percu.max(), percu.min()))
ctx.close()
# heaviest blocks by estimated issue slots: where are they and what do they execute?
top = np.argsort(-est)[:8]
lbs = np.nonzero(ok)[0]
for i in top:
    print(f"block lb={lbs[i]} est {est[i]:.0f}: loop trips {wo[i]:.0f}, walk trips {wi[i]:.0f}, sample execs {ws[i]:.0f}, duration(instrumented) {dur[i]:.0f} us")
