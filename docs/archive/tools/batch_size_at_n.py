"""Per-rank march time per frame against the frames per launch, for a rank of N = 8 / 4 (emulated on one GPU): does a longer launch amortise the tail of a 1/N share?"""
import sys, os, json
sys.path.insert(0, '/root/repo')
import torch
import vokselis_amd as V
W, H, TS = 1920, 1080, 64
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
blob = cam.get_proj_view_matrix()
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    ctx.sync(); best=1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best=min(best, ctx.timer_elapsed_ms()/iters)
    return best
for _ in range(300): pipe.record(ctx)
for nr, k in ((8, 2), (8, 3), (4, 3)):
    ctx.set_root_skip(k)
    for B in (16, 32, 64, 128):
        cap = V.partition_slots(W, H, TS, nr, k)
        buf = torch.empty((cap, B, TS, TS, 4), dtype=torch.float16, device="cuda")
        per = [timeit(lambda: V.render_batch(ctx, pipe, [blob] * B, buf.data_ptr(), tile_size=TS, rank=r, nranks=nr, compact=True, slot_capacity=cap)) for r in (0, 1, nr - 1)]
        print(json.dumps({"nranks": nr, "root_skip": k, "frames_per_launch": B, "root_us_per_frame": round(per[0] / B * 1e3, 2), "peer1_us_per_frame": round(per[1] / B * 1e3, 2), "last_peer_us_per_frame": round(per[2] / B * 1e3, 2)}), flush=True)
ctx.set_root_skip(0); ctx.close()
