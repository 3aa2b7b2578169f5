import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
for name, fl in (("skip", 0), ("noskip", V.RENDER_NO_SKIP)):
    V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT | 16).record(ctx)
    trips = ctx.read_steps()
    hit = trips[trips > 0]
    print(name, "rays with trips", hit.size, "mean", hit.mean().round(1), "pcts 50/75/90/99/max", np.percentile(hit, [50, 75, 90, 99, 100]))
    for B in (32, 64, 96, 128, 192, 256):
        alive = [(hit > B * k).sum() for k in range(1, 20) if (hit > B * k).any()]
        blk = trips[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8)
        wave_trips = blk.max(axis=(1, 3)); lane_sum = blk.sum(axis=(1, 3))
        print(f"  B={B}: rays continuing after pass k: {alive}")
    wave_trips = trips[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).astype(np.int64)
    print("  wave-level trips now:", wave_trips.sum(), " lane trips / 64:", hit.sum() / 64, " -> lane utilisation", hit.sum() / 64 / wave_trips.sum())
ctx.close()
