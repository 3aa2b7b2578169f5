"""BASELINE configs C4 (1024^3 f16, 1920x1080) and C5 (2048^3 u8, 3840x2160) on one GPU: set-up time,
frame time, step counts, and the size-independent exactness property skip == no-skip."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

def run(name, dims, fmt, W, H, iters=5, seed=0x5EED0004, layout=V.LAYOUT_AUTO):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    for kv in os.environ.get("VK_PARAMS", "").split(","):  # e.g. VK_PARAMS=stage_cap_bytes=8192,stage_kmax_log2=3
        if "=" in kv:
            ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    t0 = time.perf_counter()
    V.VolumeTexture.generate_fog(ctx, dims, fmt=fmt, seed=seed, layout=layout); ctx.sync()
    setup = time.perf_counter() - t0
    ctx.update()
    res = {"case": name, "layout": layout, "setup_s": round(setup, 2)}
    imgs = {}
    modes = (("auto", 0),) if layout in (V.LAYOUT_STAGED, V.LAYOUT_BRICKED, V.LAYOUT_QUADS) else (("auto", 0), ("skip", V.RENDER_FORCE_SKIP), ("noskip", V.RENDER_NO_SKIP))
    for mode, fl in modes:
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT).record(ctx)
        s_ref, s_samp = ctx.step_counts()
        imgs[mode] = ctx.read_backbuffer().view(np.uint16).copy()
        p = V.RaycastPipeline(dt_scale=0.5, flags=fl)
        p.record(ctx); ctx.sync()
        ctx.timer_begin()
        for _ in range(iters): p.record(ctx)
        ctx.timer_end()
        ms = ctx.timer_elapsed_ms() / iters
        b_step = 8 if fmt == V.FMT_R8_UNORM else 16
        res[mode] = {"ms": round(ms, 3), "S_ref": s_ref, "S_sampled": s_samp, "Gsteps_per_s": round(s_ref / ms / 1e6, 1),
                     "alg_TBps": round((s_samp * b_step + W * H * 8) / (ms * 1e-3) / 1e12, 3)}
    if "skip" in imgs:
        res["skip_equals_noskip_bitwise"] = bool((imgs["skip"] == imgs["noskip"]).all() and (imgs["auto"] == imgs["noskip"]).all())
    res["nonblack_px"] = int((imgs["auto"][..., :3] != 0).any(axis=2).sum())
    res["census"] = ctx.simt_census()
    import zlib
    res["crc32"] = zlib.crc32(imgs["auto"].tobytes())
    print(json.dumps(res), flush=True)
    ctx.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    lay = {"auto": V.LAYOUT_AUTO, "p8": V.LAYOUT_PACKED, "p16": V.LAYOUT_PACKED_PAIRS, "b9": V.LAYOUT_BRICKED, "q": V.LAYOUT_QUADS, "s8": V.LAYOUT_STAGED}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
    if which in ("c4", "all"): run("C4 1024^3 f16 1920x1080", (1024,) * 3, V.FMT_R16_FLOAT, 1920, 1080, layout=lay)
    if which in ("c5small", "all"): run("C5-lite 1024^3 u8 3840x2160", (1024,) * 3, V.FMT_R8_UNORM, 3840, 2160, seed=0x5EED0005, layout=lay)
    if which in ("c5", "all"): run("C5 2048^3 u8 3840x2160", (2048,) * 3, V.FMT_R8_UNORM, 3840, 2160, iters=3, seed=0x5EED0005, layout=lay)
