"""Where does the staged march fall back to global taps?  Coarse map of per-pixel fallback steps on C4."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
for kv in os.environ.get("VK_PARAMS", "").split(","):
    if "=" in kv:
        ctx.set_param(kv.split("=")[0], float(kv.split("=")[1]))
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=V.FMT_R16_FLOAT, seed=0x5EED0004, layout=V.LAYOUT_STAGED)
ctx.update()
ctx.reset_step_counts()
V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | 32).record(ctx)
fb = ctx.read_steps().astype(np.int64)
print("census", ctx.simt_census(), "fallback lane-steps", int(fb.sum()))
ctx.reset_step_counts()
V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
st = ctx.read_steps().astype(np.int64)
print("lane-steps", int(st.sum()))
# 64x64 tile map: fallback share in percent
ty, tx = (H + 63) // 64, (W + 63) // 64
for j in range(ty):
    row = []
    for i in range(tx):
        a = fb[j * 64:(j + 1) * 64, i * 64:(i + 1) * 64].sum(); b = st[j * 64:(j + 1) * 64, i * 64:(i + 1) * 64].sum()
        row.append("  ." if b == 0 else "%3d" % (100 * a // max(b, 1)))
    print("".join(row))
ctx.close()
