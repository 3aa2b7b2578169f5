"""Does an XCD-coherent tile order cut the staged march's refetch (C5: each brick fetched 2.6 times) and its time?  The launch deals runs of
four 32 x 32-pixel tiles (positions 32 c + 4 x .. + 3 of the heaviest-first order) to XCD x; here the active tiles are cut into 8 spatial
clusters (pie sectors around the silhouette's centroid, or vertical / horizontal bands, equal tile counts), each kept heaviest-first, and
interleaved so that XCD x marches cluster x only (vk_debug_set_tile_order).  Prints ms per frame for the product order and each clustering;
frames are compared bitwise.  usage: tools/xcd_order.py [c5|c4]"""
import sys, os, json, zlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
n, fmt, W, H, seed = {"c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005), "c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004)}[which]
TS = 32
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed); ctx.update()
p = V.RaycastPipeline(dt_scale=0.5)


def t(iters=6, groups=3):
    for _ in range(2): p.record(ctx)
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): p.record(ctx)
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best


p.record(ctx)
ref = zlib.crc32(ctx.read_backbuffer().tobytes())
order = ctx.partition_order(TS)  # (the order vk_render uses for the staged march: 32-pixel tiles)
n_act, _ = ctx.partition_active(TS)
tx = (W + TS - 1) // TS
out = {"config": which, "tiles": int(order.size), "active": int(n_act), "product_ms": round(t(), 4)}
act = order[:n_act].astype(np.int64)
cx, cy = (act % tx).astype(np.float64), (act // tx).astype(np.float64)


# nominal cost of a tile: steps of the ray through its centre (the arithmetic of vk_order.hip's estimate, one ray per tile)
blob = np.frombuffer(cam.get_proj_view_matrix(), np.float32).astype(np.float64)
eye, m = blob[0:3], blob[20:36]
px, py = (cx + 0.5) * TS, (cy + 0.5) * TS
X, Y = 2.0 * px / W - 1.0, 1.0 - 2.0 * py / H
qw = 1.0 / (m[3] * X + m[7] * Y + m[11] + m[15])
d = np.stack([(m[k] * X + m[4 + k] * Y + m[8 + k] + m[12 + k]) * qw - eye[k] for k in range(3)], 1)
with np.errstate(divide="ignore", invalid="ignore"):
    ta, tb = (0.0 - eye) / d, (1.0 - eye) / d
t0 = np.maximum(np.minimum(ta, tb).max(1), 0.0); t1 = np.maximum(ta, tb).min(1)
cost = np.maximum(t1 - t0, 0.0) * (np.abs(d) * n).max(1) + 1.0


def clustered(key):
    """8 clusters of (almost) equal COST, contiguous in `key`, each keeping the product order's relative (heaviest-first) order, interleaved in runs of 4."""
    o = np.argsort(key, kind="stable")
    cum = np.cumsum(cost[o]); cl = np.empty(n_act, np.int64)
    cl[o] = np.minimum((cum / cum[-1] * 8).astype(np.int64), 7)
    lists = [list(act[cl == x]) for x in range(8)]  # act is in heaviest-first order already
    new, c = [], 0
    while any(lists):
        for x in range(8):
            run, lists[x] = lists[x][:4], lists[x][4:]
            new += run + [None] * (4 - len(run))
    # holes (a cluster ran out): fill with what is left, order preserved
    placed = [v for v in new if v is not None]
    assert sorted(placed) == sorted(act.tolist())
    comp = [v for v in new if v is not None]
    return np.array(comp + order[n_act:].tolist(), dtype=np.uint32)


for name, key in (("sectors", np.arctan2(cy - cy.mean(), cx - cx.mean())), ("vbands", cx * 1000 + cy), ("hbands", cy * 1000 + cx)):
    new = clustered(key)
    V.native.check(ctx.handle, V.native.lib().vk_debug_set_tile_order(ctx.handle, new.ctypes.data_as(C.POINTER(C.c_uint32)), new.size))
    p.record(ctx)
    same = zlib.crc32(ctx.read_backbuffer().tobytes()) == ref
    out[name + "_ms"] = round(t(), 4)
    out[name + "_bitwise"] = same
print(json.dumps(out), flush=True)
ctx.close()
