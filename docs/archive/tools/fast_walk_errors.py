"""Tolerance walk (VK_RENDER_FAST_WALK) against the bit-exact walk, pixel by pixel: how large are the differences on pixels whose
iteration count did NOT change (position drift only), how many early-outs flip, and how close to 0.95 were those rays' alphas?
C1 and C2 on the bonsai stand-in, a few orbit cameras."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import variant
variant.use_variant_from_env()
import numpy as np
import vokselis_amd as V

for (W, H, dt) in ((512, 512, 1.0), (1920, 1080, 0.5)):
    for yaw in (1.0, 1.7, 2.9, 4.4):
        cam = V.Camera(1.0, 0.5, yaw, (0.5, 0.5, 0.5), W / H)
        ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
        res = {}
        for mode, fl in (("exact", 0), ("fast", V.RENDER_FAST_WALK)):
            V.RaycastPipeline(dt_scale=dt, flags=fl | V.RENDER_COUNT).record(ctx)
            res[mode] = (ctx.read_backbuffer().copy(), ctx.read_steps().copy())
        ctx.close()
        d = np.abs(res["exact"][0] - res["fast"][0]).max(axis=-1)
        same = res["exact"][1] == res["fast"][1]
        hit = res["exact"][1] > 0
        ds = d[same & hit]
        out = {"W": W, "yaw": yaw, "hit": int(hit.sum()), "flips": int((~same).sum()), "same_max": float(ds.max()),
               "same_gt_1e-4": int((ds > 1e-4).sum()), "same_gt_5e-5": int((ds > 5e-5).sum()), "same_gt_2e-5": int((ds > 2e-5).sum()),
               "same_p999": float(np.quantile(ds, 0.999)), "flip_max": float(d[~same].max()) if (~same).any() else 0.0}
        print(json.dumps(out), flush=True)
