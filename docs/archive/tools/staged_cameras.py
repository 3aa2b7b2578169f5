"""Staged layout: LDS window budget / slab thickness across camera positions (the defaults must not be tuned to one view)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
settings = [(0, 0) if a == "auto" else tuple(int(x) for x in a.split("/")) for a in (sys.argv[2:] or ["auto", "8192/8", "6144/6", "5120/6"])]
n, fmt, W, H, seed = {"c4": (1024, V.FMT_R16_FLOAT, 1920, 1080, 0x5EED0004), "c5": (2048, V.FMT_R8_UNORM, 3840, 2160, 0x5EED0005)}[which]
cams = {"bonsai (1,.5,1)": (1.0, 0.5, 1.0), "axis-aligned (1,0,0)": (1.0, 0.0, 0.0), "diagonal (1.2,.6155,.7854)": (1.2, 0.6155, 0.7854),
        "far (2.5,.3,2)": (2.5, 0.3, 2.0), "close (0.6,-.4,4)": (0.6, -0.4, 4.0), "top (1,1.5,.3)": (1.0, 1.5, 0.3)}
ctx = V.Context(W, H, V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H), backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,) * 3, fmt=fmt, seed=seed, layout=V.LAYOUT_STAGED)
p = V.RaycastPipeline(dt_scale=0.5)
for name, (zoom, pitch, yaw) in cams.items():
    ctx.set_camera_blob(V.Camera(zoom, pitch, yaw, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix())
    row = {"camera": name}
    for cap, slab in settings:
        ctx.set_param("stage_cap_bytes", cap); ctx.set_param("stage_slab_cells", slab)
        for _ in range(2): p.record(ctx)
        ctx.sync(); best = 1e9
        for _ in range(2):
            ctx.timer_begin()
            for _ in range(3): p.record(ctx)
            ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / 3)
        row[f"{cap}/{slab}" if cap else "auto"] = round(best, 3)
    print(json.dumps(row), flush=True)
ctx.close()
