"""Exploratory GPU probe: parity + timings of kernel variants (not part of the test suite)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from oracle import oracle as O

def run(name, vol, W, H, dt_scale, layout, flags, outfmt, iters=20, check=True, aspect=None):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), (W / H) if aspect is None else aspect)
    blob = cam.get_proj_view_matrix()
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=outfmt)
    V.VolumeTexture(ctx, vol, layout=layout)
    ctx.update()
    pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=dt_scale, flags=flags | V.RENDER_COUNT)
    ctx.reset_step_counts()
    pipe.record(ctx)
    s_ref, s_samp = ctx.step_counts()
    img = ctx.read_backbuffer().astype(np.float32)
    steps = ctx.read_steps()
    res = {"name": name, "S_ref": s_ref, "S_sampled": s_samp}
    if check:
        ref, rsteps, rsamp = O.render(blob, vol, W, H, dt_scale=dt_scale)
        if outfmt == V.OUT_RGBA16F:
            ref = O.rgba32f_to_rgba16f(ref).view(np.float16).astype(np.float32)
        d = np.abs(img - ref)
        res.update(max_err=float(d.max()), n_bad=int((d.max(axis=2) > 1e-4).sum()), steps_equal=bool((steps == rsteps).all()),
                   n_step_diff=int((steps != rsteps).sum()), S_ref_oracle=int(rsteps.sum()), S_samp_oracle=int(rsamp.sum()))
    pipe2 = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=dt_scale, flags=flags)
    for _ in range(5): pipe2.record(ctx)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(iters): pipe2.record(ctx)
    ctx.timer_end()
    ms = ctx.timer_elapsed_ms() / iters
    res.update(ms=ms, Gsteps_ref=s_ref / ms / 1e6, Gsteps_sampled=s_samp / ms / 1e6, frac_sampled=(s_samp * 8) / (ms * 1e-3) / 8e12)
    print(json.dumps(res), flush=True)
    ctx.close()
    return res

if __name__ == "__main__":
    t = time.time(); vol = O.volume_standin_u8(256); print("standin gen s", time.time() - t, flush=True)
    fog = O.volume_fog_u8(256)
    P8, P16, LIN = V.LAYOUT_PACKED, V.LAYOUT_PACKED_PAIRS, V.LAYOUT_LINEAR
    run("C1 P8 skip f32", vol, 512, 512, 1.0, P8, 0, V.OUT_RGBA32F)
    run("C1 P16 skip f32", vol, 512, 512, 1.0, P16, 0, V.OUT_RGBA32F)
    run("C1 P16 skip SAFE f32", vol, 512, 512, 1.0, P16, V.RENDER_SAFE, V.OUT_RGBA32F)
    run("C1 P8 noskip f32", vol, 512, 512, 1.0, P8, V.RENDER_NO_SKIP, V.OUT_RGBA32F)
    run("C1 linear f32", vol, 512, 512, 1.0, LIN, 0, V.OUT_RGBA32F)
    run("C2 P8 skip f32", vol, 1920, 1080, 0.5, P8, 0, V.OUT_RGBA32F)
    run("C2 P16 skip f32", vol, 1920, 1080, 0.5, P16, 0, V.OUT_RGBA32F)
    run("C2 P8 skip f16", vol, 1920, 1080, 0.5, P8, 0, V.OUT_RGBA16F, check=False)
    run("C2 P16 skip f16", vol, 1920, 1080, 0.5, P16, 0, V.OUT_RGBA16F, check=False)
    run("C2 P8 noskip f16", vol, 1920, 1080, 0.5, P8, V.RENDER_NO_SKIP, V.OUT_RGBA16F, check=False)
    run("C2 P16 noskip f16", vol, 1920, 1080, 0.5, P16, V.RENDER_NO_SKIP, V.OUT_RGBA16F, check=False)
    run("C2fog P8 skip f16", fog, 1920, 1080, 0.5, P8, 0, V.OUT_RGBA16F, check=True)
    run("C2fog P8 noskip f16", fog, 1920, 1080, 0.5, P8, V.RENDER_NO_SKIP, V.OUT_RGBA16F, check=False)
    run("C2fog P16 noskip f16", fog, 1920, 1080, 0.5, P16, V.RENDER_NO_SKIP, V.OUT_RGBA16F, check=False)
    run("C2fog P16 noskip SAFE f16", fog, 1920, 1080, 0.5, P16, V.RENDER_NO_SKIP | V.RENDER_SAFE, V.OUT_RGBA16F, check=False)
