"""A few frames of one configuration, for rocprofv3: tools/prof_frames.py <c2|c2fog|c4|c5|c5small> <layout> [frames]
(VK_PARAMS=name=value,... sets library knobs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V

which, lay = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
for _ in range(frames):
    p.record(ctx)
ctx.sync()
ctx.close()
