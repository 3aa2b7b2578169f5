"""How many rays are still alive in the long waves of a single C2 frame, trip by trip?  (several lanes per ray.)
From the COUNT build's per-trip logs (live lanes | samplers << 7 ...): for the waves with the most trips, the number of trips made with
at most 32 / 16 / 8 live rays -- the trips in which a wave could give every remaining ray 2 / 4 / 8 lanes without a second wave."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N

W, H, CAP = 1920, 1080, 1024
cam = V.Camera(1.0, 0.5, 1.0 + (float(sys.argv[1]) if len(sys.argv) > 1 else 0.0), (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3)
ctx.update()
pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_PROBE_ALWAYS)
ctx.set_param("trip_log_cap", CAP)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
ctx.reset_step_counts(); pipe.record(ctx); ctx.sync()
nb = 30 * 17 * 64
buf = np.zeros(nb * CAP // 2, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb * (CAP // 8)))
ctx.close()
log = buf.view(np.uint32).reshape(nb, CAP)
live, samp = (log & 127).astype(np.int64), ((log >> 7) & 127).astype(np.int64)
trips = (live > 0).sum(1)
order = np.argsort(-trips)
print("waves with trips: %d; trips total %d; max %d" % (int((trips > 0).sum()), int(trips.sum()), int(trips.max())))
print("%8s %6s %8s %8s %8s %8s | %s" % ("wave", "trips", "live<=32", "live<=16", "live<=8", "live<=4", "sampling trips among live<=16 / mean samplers there"))
for w in order[:12]:
    l, s = live[w, :trips[w]], samp[w, :trips[w]]
    m16 = l <= 16
    print("%8d %6d %8d %8d %8d %8d | %d / %.1f" % (w, trips[w], int((l <= 32).sum()), int(m16.sum()), int((l <= 8).sum()), int((l <= 4).sum()), int((s[m16] > 0).sum()), float(s[m16].mean()) if m16.any() else 0))
# frame-wide: share of wave-trips by live count
tot = int(trips.sum())
for thr in (32, 16, 8, 4):
    print("wave-trips with live <= %2d: %.3f of all" % (thr, float(((live > 0) & (live <= thr)).sum()) / tot))
# by decile of wave length
for lo, hi in ((150, 9999), (100, 150), (60, 100), (0, 60)):
    sel = (trips >= lo) & (trips < hi)
    if sel.any():
        l = live[sel]
        t = int((l > 0).sum())
        print("waves with %d <= trips < %d: %d waves, %d trips, share with live <= 32: %.2f, <= 16: %.2f, <= 8: %.2f" % (lo, hi, int(sel.sum()), t, float(((l > 0) & (l <= 32)).sum()) / t, float(((l > 0) & (l <= 16)).sum()) / t, float(((l > 0) & (l <= 8)).sum()) / t))
