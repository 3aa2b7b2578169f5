"""A stream of single-frame launches for rocprofv3 --kernel-trace: tools/prof_fif.py <c2|xor> <K> [frames] -- every frame a new orbit
camera, K frames in flight (0: the plain vk_render loop on one stream).  tools/fif_overlap.py reads the trace's kernel intervals."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import vokselis_amd as V

which, k = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if which == "c2":
    w, h, mode, dt, cam0 = 1920, 1080, V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5))
else:
    w, h, mode, dt, cam0 = 1280, 720, V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0))
ctx = V.Context(w, h, backbuffer=(w, h), out_format=V.OUT_RGBA16F)
for kv in os.environ.get("VK_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
if which == "c2":
    V.VolumeTexture.generate_standin(ctx, (256,) * 3)
else:
    V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0)
z, p, y, t = cam0
cams = [V.Camera(z, p, y + 6.28318 * j / 1024, t, w / h).get_proj_view_matrix() for j in range(n)]
pipe = V.RaycastPipeline(mode, dt_scale=dt)
if k > 0:
    ctx.frames_in_flight(k)
for rep in range(3):  # (the first repetitions settle the clocks; the last one is what fif_overlap.py reads)
    for cb in cams:
        ctx.set_camera_blob(cb)
        if k > 0:
            ctx.frame_begin()
        pipe.record(ctx)
        if k > 0:
            ctx.frame_end()
    ctx.sync()
ctx.close()
