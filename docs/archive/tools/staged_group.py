"""Staged march: one LDS window per wave (round 2) against one per 256-thread group of four waves (`stage_group`), C4 / C5 on three views,
single frames and four orbit frames per launch.  Frames must not change by a bit; rounds and mean slab thickness from the COUNT build."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

def t(ctx, fn, iters, groups=3, warm=2):
    for _ in range(warm): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
cases = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005), "small": (256, V.FMT_R8_UNORM, 640, 360, 0x5EED0006)}
n, fmt, W, H, seed = cases[which]
views = {"bonsai": (1.0, 0.5, 1.0), "diagonal": (1.0, 0.9, 0.785), "far": (2.5, 0.3, 0.4)}
cam0 = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam0, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
pipe_c = V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT)
for vname, (zoom, pitch, yaw) in views.items():
    cam = V.Camera(zoom, pitch, yaw, (0.5, 0.5, 0.5), W / H)
    ctx.set_camera_blob(cam.get_proj_view_matrix())
    ref = None
    for grp in (0, 1, 2, 0, 1, 2):
        ctx.set_param("stage_group", grp)
        ms = t(ctx, lambda: pipe.record(ctx), 3 if which == "c5" else 8)
        img = ctx.read_backbuffer().view(np.uint16)
        sha = hashlib.sha256(img.tobytes()).hexdigest()[:16]
        ref = sha if ref is None else ref
        ctx.reset_step_counts(); pipe_c.record(ctx); s_ref, s_samp = ctx.step_counts(); c = ctx.simt_census()
        print(json.dumps({"case": which, "view": vname, "stage_group": grp, "ms": round(ms, 4), "sha": sha, "bitwise": sha == ref, "S_ref": s_ref,
                          "wave_rounds": c["wave_loop_iters"], "rounds_from_global": c["wave_skip_iters"], "mean_T": round(c["wave_sample_execs"] / max(c["wave_loop_iters"], 1), 2)}), flush=True)
        assert sha == ref
B = 4
frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda"); torch.cuda.synchronize()
cams = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
ref = None
for grp in (0, 1, 2, 0, 1, 2):
    ctx.set_param("stage_group", grp)
    ms = t(ctx, lambda: V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=64), 2, warm=1) / B
    ctx.sync(); torch.cuda.synchronize()
    sha = hashlib.sha256(frames.cpu().numpy().tobytes()).hexdigest()[:16]
    ref = sha if ref is None else ref
    print(json.dumps({"case": which, "view": "orbit x4 per launch", "stage_group": grp, "ms_per_frame": round(ms, 4), "sha": sha, "bitwise": sha == ref}), flush=True)
    assert sha == ref
ctx.close()
