"""Colour-only tiles on the wire (vk_partition_wire): what it costs the march and the un-tile, what it takes off a peer's link.
C2, 8 ranks emulated on this GPU (each rank's compact launch and the root's un-tile timed alone), 128 orbit frames per launch."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V

W, H, TS, NR, B = 1920, 1080, 64, 8, 128
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3)
ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
cams = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]

def timeit(fn, iters=6, warm=3):
    for _ in range(warm): fn()
    ctx.sync(); best = 1e9
    for _ in range(3):
        ctx.timer_begin()
        for _ in range(iters): fn()
        ctx.timer_end(); best = min(best, ctx.timer_elapsed_ms() / iters)
    return best

frames = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
for k in (0, 2):
    ctx.set_root_skip(k)
    cap = V.partition_slots(W, H, TS, NR, k)
    for rep in range(2):
        for wire, name in ((V.WIRE_RGBA, "rgba"), (V.WIRE_RGB, "rgb")):
            ctx.set_wire(wire)
            ch = 3 if wire == V.WIRE_RGB else 4
            buf = torch.empty((cap, B, TS * TS * ch), dtype=torch.float16, device="cuda")
            per_rank = [timeit(lambda: V.render_batch(ctx, pipe, cams, buf.data_ptr(), tile_size=TS, rank=r, nranks=NR, compact=True, slot_capacity=cap)) for r in (0, 1, NR - 1)]
            bid, act = V.render_batch(ctx, pipe, cams, buf.data_ptr(), tile_size=TS, rank=0, nranks=NR, compact=True, slot_capacity=cap)
            gathered = torch.zeros((NR, act, B, TS * TS * ch), dtype=torch.float16, device="cuda"); torch.cuda.synchronize()
            un = timeit(lambda: V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr()))
            mb = act * TS * TS * ctx.wire_pixel_bytes / 1e6
            print(json.dumps({"root_skip": k, "wire": name, "march_us_per_frame_rank0_rank1_rank7": [round(x / B * 1e3, 2) for x in per_rank], "untile_full_us_per_frame": round(un / B * 1e3, 2),
                              "active_slots": act, "MB_per_peer_and_frame": round(mb, 3), "link_us_per_frame_at_76.8_and_57.6_GBps": [round(mb / 76.8 * 1e3, 1), round(mb / 57.6 * 1e3, 1)]}), flush=True)
ctx.close()
