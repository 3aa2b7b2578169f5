"""A/B of the skip walk: closed form (v_n = fma(n - 1, d, v1), the product) against the sequential additions of rounds 1-3 (a second build of the
library with -DVK_WALK_SEQUENTIAL, tools/walk_closed_form.sh).  argv[1]: path of the library to load; prints C2 figures as JSON lines."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vokselis_amd import _native as N
if len(sys.argv) > 1 and sys.argv[1] != "-":
    N.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch, hashlib
import vokselis_amd as V

def t(ctx, fn, iters, groups=3, warm=3):
    for _ in range(warm): fn()
    ctx.sync(); best = 1e9
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

W, H = 1920, 1080
tag = os.path.basename(N.LIB_PATH)
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
caps = [tuple(int(x) for x in a.split("/")) for a in sys.argv[2:]] or [(8, 12)]
for kind in ("standin", "fog"):
    if kind == "standin": V.VolumeTexture.generate_standin(ctx, (256,) * 3)
    else: V.VolumeTexture.generate_fog(ctx, (256,) * 3)
    ctx.update()
    for fl, name in ((0, "default"), (V.RENDER_FORCE_SKIP, "force_skip")):
        if kind == "standin" and fl: continue
        pipe = V.RaycastPipeline(dt_scale=0.5, flags=fl)
        for cap, cap_all in caps:
            ctx.set_param("walk_cap", cap); ctx.set_param("walk_cap_all", cap_all)
            out = {"lib": tag, "volume": kind, "policy": name, "walk_cap": cap, "walk_cap_all": cap_all}
            ctx.reset_step_counts()
            V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT).record(ctx)
            out["S_ref"], out["S_sampled"] = ctx.step_counts(); out.update(ctx.simt_census())
            out["single_ms"] = round(t(ctx, lambda: pipe.record(ctx), 20), 5)
            out["sha_single"] = hashlib.sha256(ctx.read_backbuffer().tobytes()).hexdigest()[:16]
            for B, mode in ((64, "orbit"), (64, "still")) if kind == "standin" else ((8, "orbit"),):
                frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda"); torch.cuda.synchronize()
                cams = [V.Camera(1.0, 0.5, 1.0 + (6.28318 * j / 1024 if mode == "orbit" else 0.0), (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
                out[f"{mode}{B}_ms_per_frame"] = round(t(ctx, lambda: V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=64), 6, warm=4) / B, 5)
                ctx.sync(); torch.cuda.synchronize()
                out[f"sha_{mode}{B}"] = hashlib.sha256(frames.cpu().numpy().tobytes()).hexdigest()[:16]
                del frames
            print(json.dumps(out), flush=True)
ctx.close()
