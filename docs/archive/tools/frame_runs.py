"""Batched launches under a moving camera: which frames of a tile position should one XCD march?  frame_runs = 0: frames x, x + 8, x + 16 ... (frame index
fastest over the XCDs, round 2); 1: a run of consecutive frames.  C2 (64 and 52 orbit frames per launch, and one camera repeated), C2 fog dense at 8, C4 / C5 at
4 orbit frames per launch.  The frames must not change by a bit."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vokselis_amd as V

def t(ctx, fn, iters, groups=3, warm=3):
    for _ in range(warm): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cases = {"c2": (256, V.FMT_R8_UNORM, 1920, 1080, "standin", 0, [(64, "orbit"), (52, "orbit"), (64, "still"), (32, "orbit")]),
         "c2fog": (256, V.FMT_R8_UNORM, 1920, 1080, "fog", V.RENDER_NO_SKIP, [(8, "orbit"), (8, "still")]),
         "c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, "fog4", 0, [(4, "orbit"), (8, "orbit")]),
         "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, "fog5", 0, [(4, "orbit"), (8, "orbit")])}
n, fmt, W, H, kind, fl, shapes = cases[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
if kind == "standin": V.VolumeTexture.generate_standin(ctx, (n,) * 3)
elif kind == "fog": V.VolumeTexture.generate_fog(ctx, (n,) * 3)
else: V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=0x5EED0004 if kind == "fog4" else 0x5EED0005)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5, flags=fl)
for B, mode in shapes:
    frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda"); torch.cuda.synchronize()
    cams = [V.Camera(1.0, 0.5, 1.0 + (6.28318 * j / 1024 if mode == "orbit" else 0.0), (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
    ref = None
    for runs in (0, 1, 0, 1):
        ctx.set_param("frame_runs", runs)
        ms = t(ctx, lambda: V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=64), 6 if which in ("c2", "c2fog") else 2, warm=4 if which in ("c2", "c2fog") else 1) / B
        ctx.sync(); torch.cuda.synchronize()
        img = frames.cpu().numpy().view(np.uint16)
        ref = img if ref is None else ref
        same = bool((img == ref).all())
        print(json.dumps({"case": which, "frames_per_launch": B, "cameras": mode, "frame_runs": runs, "ms_per_frame": round(ms, 5), "bitwise": same}), flush=True)
        assert same
    del frames
ctx.close()
