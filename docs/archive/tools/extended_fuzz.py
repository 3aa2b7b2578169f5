"""Extended seeded fuzz on the GPU, beyond what the test suite runs every round: internal consistency, bit for bit, of the paths that changed in round 3.
  (a) staged layout, windows per wave and per 256-thread group, random LDS budgets and slab caps  ==  the dense LINEAR kernel (frames and per-pixel trip counts)
  (b) skip kernel (cells + distance maps)  ==  the same layout without skipping
  (c) a batch of cameras through an emulated partition (2..8 ranks, weighted deals, colour-only tiles on the wire) + un-tile  ==  single vk_render frames,
      with more than eight frames per launch (frame runs per XCD)
usage: tools/extended_fuzz.py [seed] [trials]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)

def cam_blob(W, H):
    kind = int(rng.integers(0, 4))
    if kind == 0: a = (float(rng.uniform(0.6, 3.0)), float(rng.uniform(-1.4, 1.4)), float(rng.uniform(0, 6.28)), (0.5, 0.5, 0.5))
    elif kind == 1: a = (float(rng.uniform(0.05, 0.4)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(0, 6.28)), tuple(float(x) for x in rng.uniform(0.3, 0.7, 3)))
    elif kind == 2: a = (1.5, 0.0, float(rng.integers(0, 4)) * 1.5707963, (0.5, 0.5, 0.5))
    else: a = (1.2, float(rng.uniform(-0.05, 0.05)), float(rng.uniform(0, 6.28)), (0.5, float(rng.choice([0.02, 0.98])), 0.5))
    return V.Camera(a[0], a[1], a[2], a[3], W / H).get_proj_view_matrix()

def volume(dims, f16):
    z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    v = np.zeros(x.shape, np.float32)
    for _ in range(int(rng.integers(1, 5))):
        c = rng.uniform(0.1, 0.9, 3) * np.array(dims); rad = rng.uniform(1.0, max(2.0, 0.45 * min(dims)))
        v = np.maximum(v, np.where((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 < rad * rad, rng.uniform(0.15, 1.0), 0.0))
    v = np.maximum(v, rng.uniform(0, 1, x.shape) < 0.003)  # speckle
    if rng.uniform() < 0.3: v = np.maximum(v, rng.uniform(0.05, 0.2))  # fog: nothing to skip
    return v.astype(np.float16) if f16 else np.round(v * 255).astype(np.uint8)

def render(vol, W, H, cam, dt, layout, flags=0, params=()):
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        for k, v in params: ctx.set_param(k, v)
        V.VolumeTexture(ctx, vol, layout=layout)
        ctx.set_camera_blob(cam)
        V.RaycastPipeline(dt_scale=dt, flags=flags | V.RENDER_COUNT).record(ctx)
        return ctx.read_backbuffer().view(np.uint32).copy(), ctx.read_steps().copy()
    finally:
        ctx.close()

t0 = time.time(); done = {"staged": 0, "skip": 0, "batch": 0}
for trial in range(trials):
    which = trial % 3
    W, H = int(rng.integers(3, 14)) * 8 + int(rng.integers(0, 8)), int(rng.integers(3, 12)) * 8 + int(rng.integers(0, 8))
    dims = tuple(int(x) for x in rng.integers(3, 80, 3))
    f16 = bool(rng.integers(0, 2))
    dt = float(rng.choice([0.13, 0.37, 0.5, 1.0, 2.1]))
    vol = volume(dims, f16)
    cam = cam_blob(W, H)
    tag = {"trial": trial, "dims": dims, "W": W, "H": H, "f16": f16, "dt": dt}
    if which == 0:
        ref, rs = render(vol, W, H, cam, dt, V.LAYOUT_LINEAR)
        for grp in (0, 1):
            params = (("stage_group", grp), ("stage_cap_bytes", int(rng.choice([0, 1024, 2048, 4096, 8192, 16384]))), ("stage_slab_cells", int(rng.choice([0, 1, 2, 5, 12, 27]))))
            img, st = render(vol, W, H, cam, dt, V.LAYOUT_STAGED, params=params)
            assert (img == ref).all() and (st == rs).all(), ("staged", tag, params)
        done["staged"] += 1
    elif which == 1:
        lay = V.LAYOUT_PACKED if f16 else int(rng.choice([V.LAYOUT_PACKED, V.LAYOUT_PACKED_PAIRS]))
        a, sa = render(vol, W, H, cam, dt, lay, flags=V.RENDER_FORCE_SKIP)
        b, sb = render(vol, W, H, cam, dt, lay, flags=V.RENDER_NO_SKIP)
        c, sc = render(vol, W, H, cam, dt, lay, flags=V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS)
        assert (a == b).all() and (sa == sb).all() and (a == c).all() and (sc == sb).all(), ("skip", tag)
        done["skip"] += 1
    else:
        B, ts = int(rng.integers(2, 22)), int(rng.choice([16, 32, 64]))
        nr, k = int(rng.integers(1, 9)), int(rng.choice([0, 0, 2, 3, 5]))
        cams = [cam_blob(W, H) for _ in range(B)]
        fmt = V.OUT_RGBA16F if rng.integers(0, 2) else V.OUT_RGBA32F
        tdt = torch.float16 if fmt == V.OUT_RGBA16F else torch.float32
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=fmt)
        try:
            V.VolumeTexture(ctx, vol, layout=V.LAYOUT_STAGED if rng.integers(0, 3) == 0 else V.LAYOUT_AUTO)
            pipe = V.RaycastPipeline(dt_scale=dt)
            singles = []
            for c_ in cams:
                ctx.set_camera_blob(c_); pipe.record(ctx); singles.append(ctx.read_backbuffer().copy())
            wire = int(rng.integers(0, 2)); ch = 3 if wire else 4
            ctx.set_wire(wire); ctx.set_root_skip(k if nr > 1 else 0)
            cap = V.partition_slots(W, H, ts, nr, k if nr > 1 else 0)
            gathered = None
            for r in range(nr):
                buf = torch.full((cap, B, ts * ts * ch), 3.0, dtype=tdt, device="cuda"); torch.cuda.synchronize()
                bid, act = V.render_batch(ctx, pipe, cams, buf.data_ptr(), tile_size=ts, rank=r, nranks=nr, compact=True, slot_capacity=cap)
                if gathered is None:
                    gathered = torch.zeros((nr, act, B, ts * ts * ch), dtype=tdt, device="cuda"); torch.cuda.synchronize()
                ctx.sync(); gathered[r] = buf[:act]
            frames = torch.zeros((B, H, W, 4), dtype=tdt, device="cuda"); torch.cuda.synchronize()
            V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr()); ctx.sync()
            got = frames.cpu().numpy()
            for j in range(B):
                assert (got[j].view(np.uint8) == singles[j].view(np.uint8)).all(), ("batch", tag, B, ts, nr, k, wire, j)
            # whole frames in one launch
            ctx.set_wire(0); ctx.set_root_skip(0)
            V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=ts); ctx.sync()
            got = frames.cpu().numpy()
            for j in range(B):
                assert (got[j].view(np.uint8) == singles[j].view(np.uint8)).all(), ("whole", tag, B, ts, j)
        finally:
            ctx.close()
        done["batch"] += 1
print(json.dumps({"seed": seed, "trials": trials, "passed": done, "seconds": round(time.time() - t0, 1)}))
