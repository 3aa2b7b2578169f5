"""A few frames of one configuration, for rocprofv3: tools/prof_frames.py <c2|c2fog|c4|c5|c5small|xor|c3> <layout> [frames]
(VK_PARAMS=name=value,... sets library knobs; VK_NOSKIP=1: the dense kernel; VK_BATCH=n: n frames per launch, consecutive orbit cameras).
xor: the compute twin on the xor example's own frame (256^3 rgba16f pair, 1280x720); c3: the procedural mode at 1920x1080."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V

which, lay = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if which in ("xor", "c3"):
    W, H = (1280, 720) if which == "xor" else (1920, 1080)
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    if which == "xor":
        V.VolumeTexture.generate_xor(ctx, (256,) * 3, 0.0)
    ctx.update()
    ctx.set_camera_blob(cam.get_proj_view_matrix())
    p = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST if which == "xor" else V.MODE_PROCEDURAL)
    for _ in range(frames):
        p.record(ctx)
    ctx.sync()
    ctx.close()
    sys.exit(0)
layout = {"auto": V.LAYOUT_AUTO, "p8": V.LAYOUT_PACKED, "p16": V.LAYOUT_PACKED_PAIRS, "b9": V.LAYOUT_BRICKED, "q": V.LAYOUT_QUADS, "s8": V.LAYOUT_STAGED,
          "lin": V.LAYOUT_LINEAR}[lay]
n, fmt, W, H, seed, kind = {"c2": (256, V.FMT_R8_UNORM, 1920, 1080, 0x5EED0001, "standin"), "c2fog": (256, V.FMT_R8_UNORM, 1920, 1080, 0x5EED0002, "fog"),
                            "c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004, "fog"), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005, "fog"),
                            "c5small": (1024, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005, "fog")}[which]
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
for kv in os.environ.get("VK_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
if kind == "standin":
    V.VolumeTexture.generate_standin(ctx, (n,) * 3, layout=layout)
else:
    V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=layout)
ctx.update()
flags = V.RENDER_NO_SKIP if os.environ.get("VK_NOSKIP") else 0
p = V.RaycastPipeline(dt_scale=0.5, flags=flags)
nb = int(os.environ.get("VK_BATCH", "0"))
if nb:
    import torch
    fr = torch.empty((nb, H, W, 4), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    blobs = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(nb)]
    for _ in range(frames):
        V.render_batch(ctx, p, blobs, fr.data_ptr(), tile_size=64)
else:
    for _ in range(frames):
        p.record(ctx)
ctx.sync()
ctx.close()
