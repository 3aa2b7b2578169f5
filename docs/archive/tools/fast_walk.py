"""VK_RENDER_FAST_WALK (tolerance mode: a skip advances t and p by one fma each) against the default, bit-exact walk.
C1 and C2 on the bonsai stand-in: per-channel difference of the f32 frames, pixels whose iteration count changed, single-frame
and 64-orbit-frames-per-launch times of both modes (interleaved), and the knocked-out fogs of the crossover table."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

TS = 64


def t(ctx, fn, iters, groups=3):
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


def compare(W, H, dt, mk, name, flags=0):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    mk(ctx); ctx.update()
    res = {}
    for mode, fl in (("exact", flags), ("fast", flags | V.RENDER_FAST_WALK)):
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=dt, flags=fl | V.RENDER_COUNT).record(ctx)
        res[mode] = (ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts())
    a, b = res["exact"], res["fast"]
    d = np.abs(a[0] - b[0])
    out = {"case": name, "max_abs_diff": float(d.max()), "p9999_abs_diff": float(np.quantile(d, 0.9999)), "pixels_steps_changed": int((a[1] != b[1]).sum()),
           "max_step_change": int(np.abs(a[1].astype(np.int64) - b[1].astype(np.int64)).max()), "hit_pixels": int((a[1] > 0).sum()),
           "s_ref_exact": a[2][0], "s_ref_fast": b[2][0], "s_sampled_exact": a[2][1], "s_sampled_fast": b[2][1]}
    ctx.close()
    # timing on the reference-shaped surface
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    mk(ctx); ctx.update()
    blob = cam.get_proj_view_matrix()
    frames = torch.empty((64, H, W, 4), dtype=torch.float16, device="cuda")
    orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(64)]
    pe, pf = V.RaycastPipeline(dt_scale=dt, flags=flags), V.RaycastPipeline(dt_scale=dt, flags=flags | V.RENDER_FAST_WALK)
    for _ in range(300): pe.record(ctx)
    for rep in range(2):
        for mode, p in (("exact", pe), ("fast", pf)):
            out["%s_single_ms_%d" % (mode, rep)] = round(t(ctx, lambda: p.record(ctx), 50), 4)
            out["%s_orbit64_ms_per_frame_%d" % (mode, rep)] = round(t(ctx, lambda: V.render_batch(ctx, p, orbit, frames.data_ptr(), tile_size=TS), 4) / 64, 5)
    ctx.close()
    print(json.dumps(out), flush=True)


standin = lambda ctx: V.VolumeTexture.generate_standin(ctx, (256,) * 3)
compare(512, 512, 1.0, standin, "C1 stand-in 512x512 dt 1")
compare(1920, 1080, 0.5, standin, "C2 stand-in 1920x1080 dt 0.5")
if "--more" in sys.argv:
    for p in (0.6, 0.8):
        vol = knocked(p)
        compare(1920, 1080, 0.5, lambda ctx: V.VolumeTexture(ctx, vol), "fog p=%.1f 1920x1080 dt 0.5" % p, V.RENDER_FORCE_SKIP)
