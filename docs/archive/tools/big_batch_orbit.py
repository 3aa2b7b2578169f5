"""C4 / C5 several frames per launch: the same camera repeated (frames share their brick fetches) against consecutive
frames of an orbit (yaw step 2 pi / 1024) and against unrelated views."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
n, fmt, W, H, seed, B = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004, 4), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005, 2)}[which]
mk = lambda zoom, pitch, yaw: V.Camera(zoom, pitch, yaw, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
ctx = V.Context(W, H, V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H), backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_AUTO); ctx.update()
p = V.RaycastPipeline(dt_scale=0.5)
fr = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")


def t(fn, it=4):
    fn(); fn(); ctx.sync(); best = 1e9
    for _ in range(2):
        ctx.timer_begin()
        for _ in range(it): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / it)
    return best


single = t(lambda: p.record(ctx), 6)
cases = {"same camera x%d" % B: [mk(1.0, 0.5, 1.0)] * B,
         "orbit, yaw step 2pi/1024": [mk(1.0, 0.5, 1.0 + 6.28318 * j / 1024) for j in range(B)],
         "unrelated views": [mk(1.0, 0.5, 1.0), mk(1.0, -0.5, 2.6), mk(1.0, 0.2, 4.1), mk(1.0, 0.9, 5.5)][:B]}
print(json.dumps({"case": which, "single_frame_ms": round(single, 3)}))
for name, cams in cases.items():
    ms = t(lambda: V.render_batch(ctx, p, cams, fr.data_ptr(), tile_size=64)) / B
    singles = 0.0
    for c in set(cams):
        ctx.set_camera_blob(c); singles += t(lambda: p.record(ctx), 4) * cams.count(c)
    print(json.dumps({"case": which, "batch": name, "ms_per_frame": round(ms, 3), "same_frames_one_per_launch_ms": round(singles / B, 3)}), flush=True)
ctx.close()
