"""Skip kernels: cap on the steps a walk may take in a trip that also has sampling lanes (walk_cap; 0 = none).  C2 stand-in
(default policy), the p = 0.4 / 0.6 knocked-out fogs of the crossover table, single frames, batches of 32 of one camera, and (round 3) the
headline's shape: 64 consecutive orbit frames per launch."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

W, H, TS = 1920, 1080, 64
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
blob = cam.get_proj_view_matrix()
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
frames = torch.empty((64, H, W, 4), dtype=torch.float16, device="cuda")
orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(64)]

def t(fn, iters, groups=3):
    for _ in range(3): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

def knocked(p, seed=7):
    rng = np.random.default_rng(seed)
    v = rng.integers(26, 41, (256, 256, 256), dtype=np.uint8)
    k = rng.random((16, 16, 16)) < p
    v[np.kron(k, np.ones((16, 16, 16), bool))] = 0
    return v

cases = [("standin", lambda: V.VolumeTexture.generate_standin(ctx, (256,) * 3), 0)]
if "--more" in sys.argv:
    cases += [("fog p=0.4 (forced skip)", lambda: V.VolumeTexture(ctx, knocked(0.4)), V.RENDER_FORCE_SKIP), ("fog p=0.8", lambda: V.VolumeTexture(ctx, knocked(0.8)), 0)]
for name, mk, fl in cases:
    mk(); ctx.update()
    pipe = V.RaycastPipeline(dt_scale=0.5, flags=fl)
    ref = None
    for cap, cap_all in ((8, 12), (0, 0), (4, 12), (4, 8), (6, 12), (8, 8), (8, 16), (8, 24), (12, 12), (12, 16), (16, 16), (8, 12)):
        ctx.set_param("walk_cap", cap); ctx.set_param("walk_cap_all", cap_all)
        for _ in range(400): pipe.record(ctx)   # clocks: ~60 ms
        single = t(lambda: pipe.record(ctx), 50)
        img = ctx.read_backbuffer().view(np.uint16)
        ref = img if ref is None else ref
        same = bool((img == ref).all())
        batch = t(lambda: V.render_batch(ctx, pipe, [blob] * 32, frames.data_ptr(), tile_size=TS), 6) / 32
        orb = t(lambda: V.render_batch(ctx, pipe, orbit, frames.data_ptr(), tile_size=TS), 4) / 64
        print(json.dumps({"volume": name, "walk_cap": cap, "walk_cap_all": cap_all, "single_frame_ms": round(single, 4), "batch32_ms_per_frame": round(batch, 4), "orbit64_ms_per_frame": round(orb, 4), "bitwise_equal_to_uncapped": same}), flush=True)
ctx.set_param("walk_cap", 0)
ctx.close()
