import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
import vokselis_amd as V
W,H,n=3840,2160,2048
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_fog(ctx, (n,)*3, fmt=V.FMT_R8_UNORM, seed=0x5EED0005); ctx.update()
ctx.reset_step_counts()
V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
s_ref, s_samp = ctx.step_counts()
c = ctx.simt_census()
waves = sum(1 for _ in range(0))
print(json.dumps({"s_ref": s_ref, "census": c}))
rounds, sumT = c["wave_loop_iters"], c["wave_sample_execs"]
print("rounds (wave-level)", rounds, "mean T", sumT / rounds, "lane-steps per wave-round", s_ref / rounds, "fallback rounds", c["wave_skip_iters"])
ctx.close()
