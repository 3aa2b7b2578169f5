"""Frames in flight in a few seconds: ms per frame of the one-vk_render-per-frame surface, every frame a new orbit camera, on one
surface / one stream (today's vk_render loop) and on rings of K = 1..4 surfaces (vk_ctx_frames_in_flight), for C2 (bonsai stand-in
256^3, 1080p, dt_scale 0.5) and the xor example's own frame (256^3 pair, 1280x720).  Wall time around N frames, best of three.
usage: tools/fif_quick.py [c4] [c5] [--frames N] [--present | --fused]   (--present: vk_present after every vk_render; --fused: VK_RENDER_PRESENT)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import torch  # noqa: F401  (one HIP runtime per process: see vokselis_amd/_native.py)
import vokselis_amd as V

n = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 256
present = "--present" in sys.argv
fused = "--fused" in sys.argv


def stream_of_frames(ctx, pipe, cams, frames_api):
    """ms per frame, wall time: set_camera -> [frame_begin] -> vk_render [-> vk_present] -> [frame_end] for every camera."""
    def once(k):
        for j in range(k):
            ctx.set_camera_blob(cams[j % len(cams)])
            if frames_api:
                ctx.frame_begin()
            pipe.record(ctx)
            if present:
                ctx.render()
            if frames_api:
                ctx.frame_end()
    once(min(64, n))
    ctx.sync()
    best, host = 1e9, 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        once(n)
        t1 = time.perf_counter()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
        host = min(host, (t1 - t0) / n * 1e3)
    return best, host


big = [a for a in sys.argv[1:] if a in ("c4", "c5")]  # the beyond-cache configs instead of C2 / xor (fewer frames: a frame is milliseconds)
out = {"lib": os.environ.get("VK_LIB", "product"), "params": os.environ.get("VK_PARAMS", ""), "frames": n, "present": "fused" if fused else present}
CASES = [("c2", (1920, 1080), lambda c: V.VolumeTexture.generate_standin(c, (256,) * 3), V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5))),
         ("xor720p", (1280, 720), lambda c: V.VolumeTexture.generate_xor(c, (256,) * 3, 0.0), V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)))]
if big:
    n = min(n, 48)
    CASES = [c for c in (("c4", (1920, 1080), lambda c: V.VolumeTexture.generate_fog(c, (1024,) * 3, fmt=V.FMT_R16_FLOAT, seed=0x5EED0004), V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5))),
                         ("c5", (3840, 2160), lambda c: V.VolumeTexture.generate_fog(c, (2048,) * 3, fmt=V.FMT_R8_UNORM, seed=0x5EED0005), V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5)))) if c[0] in big]
for name, (w, h), mk, mode, dt, cam0 in CASES:
    ctx = V.Context(w, h, backbuffer=(w, h), out_format=V.OUT_RGBA16F)
    for kv in os.environ.get("VK_PARAMS", "").split(","):  # library knobs: VK_PARAMS=name=value,...
        if "=" in kv:
            ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    mk(ctx)
    z, p, y, t = cam0
    cams = [V.Camera(z, p, y + 6.28318 * j / 1024, t, w / h).get_proj_view_matrix() for j in range(128)]
    pipe = V.RaycastPipeline(mode, dt_scale=dt, flags=V.RENDER_PRESENT if fused else 0)
    ctx.set_camera_blob(cams[0])
    for _ in range(8 if big else 200):
        pipe.record(ctx)  # clocks
    r5 = lambda t: (round(t[0], 5), round(t[1], 5))  # (ms per frame, of which the host spent submitting)
    res = {"plain": r5(stream_of_frames(ctx, pipe, cams, False))}
    for k in (1, 2, 3, 4):
        ctx.frames_in_flight(k)
        res["k%d" % k] = r5(stream_of_frames(ctx, pipe, cams, True))
    ctx.frames_in_flight(1)
    res["plain_again"] = r5(stream_of_frames(ctx, pipe, cams, False))
    out[name] = res
    ctx.close()
print(json.dumps(out), flush=True)
