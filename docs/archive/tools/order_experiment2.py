"""Upper bound of what a better tile order could give: orders built from the TRUE per-tile work (wave-level trip
counters of an instrumented frame), heaviest-first + snake, and LPT-balanced over the 8 XCDs."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3, layout=V.LAYOUT_PACKED_PAIRS); ctx.update()
def timeit(p, it=300):
    for _ in range(10): p.record(ctx)
    ctx.sync(); ctx.timer_begin()
    for _ in range(it): p.record(ctx)
    ctx.timer_end(); return ctx.timer_elapsed_ms() / it
pipe = V.RaycastPipeline(dt_scale=0.5)
print("library order ms:", timeit(pipe))
pc = V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
pc.record(ctx); ctx.sync()
nb = 30 * 17 * 64
buf = np.zeros(nb * 4, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb))
work = buf[3::4]
wo = (work & np.uint64(0xfffff)).astype(np.float64); wi = ((work >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64); ws = (work >> np.uint64(40)).astype(np.float64)
est = (wo * 28 + wi * 9 + ws * 33 + (wo > 0) * 300).reshape(-1, 64)     # per (order position, block)
order = ctx.partition_order(64)
n_active = ctx.partition_active(64, 1)[0]
tile_cost = est.sum(1)                                                  # by position
print("active tiles", n_active, "cost of active", tile_cost[:n_active].sum(), "inactive", tile_cost[n_active:].sum())
def apply(new_pos, label):
    new_order = order[np.asarray(new_pos)].astype(np.uint32)
    N.check(ctx.handle, N.lib().vk_debug_set_tile_order(ctx.handle, new_order.ctypes.data_as(C.POINTER(C.c_uint32)), len(new_order)))
    print(f"{label}: {timeit(pipe):.4f} ms")
act = np.arange(n_active); rest = np.arange(n_active, len(order))
srt = act[np.argsort(-tile_cost[:n_active], kind="stable")]
apply(np.concatenate([srt, rest]), "true-work heaviest-first")
sn = srt.copy()
for g in range(8, len(sn) - 7, 16): sn[g:g + 8] = sn[g:g + 8][::-1]
apply(np.concatenate([sn, rest]), "true-work heaviest-first + snake")
# LPT over 8 XCD bins, then interleave bins round by round (position q -> XCD q % 8)
bins = [[] for _ in range(8)]; load = np.zeros(8)
for p in srt:
    b = int(np.argmin(load + (np.array([len(x) for x in bins]) > (len(srt) + 7) // 8 - 1) * 1e18)); bins[b].append(p); load[b] += tile_cost[p]
print("LPT XCD loads / mean:", (load / load.mean()).round(3))
inter = []
for r in range(max(len(x) for x in bins)):
    for b in range(8):
        inter.append(bins[b][r] if r < len(bins[b]) else None)
inter = [x for x in inter]
# fill holes (None) with inactive tiles so the XCD mapping is preserved
rest_list = list(rest); out = []
for x in inter:
    out.append(x if x is not None else rest_list.pop())
apply(np.concatenate([np.array(out), np.array(rest_list, dtype=int)]), "true-work LPT over XCDs")
# proxy check: outer trips only (what a production kernel can count for free on the scalar unit)
def lpt(cost_vec, label):
    srt2 = act[np.argsort(-cost_vec[:n_active], kind="stable")]
    rounds, rem = n_active // 8, n_active % 8
    bins2 = [[] for _ in range(8)]; load2 = np.zeros(8)
    for p_ in srt2:
        cand = [b for b in range(8) if len(bins2[b]) < rounds + (1 if b < rem else 0)]
        b = min(cand, key=lambda j: load2[j]); bins2[b].append(p_); load2[b] += cost_vec[p_]
    out2 = []
    for r_ in range(rounds + 1):
        for b in range(8):
            if r_ < len(bins2[b]): out2.append(bins2[b][r_])
    true_load = np.array([tile_cost[bins2[b]].sum() for b in range(8)])
    print(label, "true XCD loads / mean:", (true_load / true_load.mean()).round(3))
    apply(np.concatenate([np.array(out2), rest]), label)
lpt(tile_cost, "LPT no holes, true work")
lpt((wo.reshape(-1, 64) + 4 * (wo.reshape(-1, 64) > 0)).sum(1), "LPT no holes, outer trips only")
lpt((wo.reshape(-1, 64) * 28 + wi.reshape(-1, 64) * 9).sum(1), "LPT no holes, trips + walk")
ctx.close()
