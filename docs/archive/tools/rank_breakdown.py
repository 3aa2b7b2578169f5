"""Why does rank r of 8 take ~130 us?  Its tiles alone (vk_render per tile) vs. the partition launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import vokselis_amd as V
W, H, ts, N, r = 1920, 1080, 64, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
def timeit(fn, it=50):
    for _ in range(5): fn()
    ctx.sync(); ctx.timer_begin()
    for _ in range(it): fn()
    ctx.timer_end(); return ctx.timer_elapsed_ms() / it * 1e3
order = ctx.partition_order(ts); n_active = ctx.partition_active(ts, N)[0]
tx = (W + ts - 1) // ts
mine = [int(order[q]) for q in range(r, n_active, N)]
alone = []
for t in mine:
    x0, y0 = (t % tx) * ts, (t // tx) * ts
    alone.append(timeit(lambda: pipe.record(ctx, (x0, y0, ts, ts)), 20))
print("rank", r, "tiles", len(mine), "each alone us:", [round(a, 1) for a in alone])
slots = V.partition_slots(W, H, ts, N)
buf = torch.zeros((slots, ts, ts, 4), dtype=torch.float16, device="cuda")
print("partition launch us:", round(timeit(lambda: pipe.record_partition(ctx, ts, r, N, buf.data_ptr())), 1))
# the same tiles as one rectangular-region-free launch is not expressible; approximate with the first k tiles of the rank via N' = large
for k in (1, 2, 4, 8, 16):
    # rank r of N*? : emulate by rendering only first k tiles sequentially (sum) -- lower bound on serial cost
    print(f"first {k} tiles sequential sum us: {sum(alone[:k]):.1f}")
ctx.close()
