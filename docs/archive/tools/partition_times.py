"""Per-rank march time of the C2 frame when its tiles are dealt to N ranks (single-GPU emulation: rank r of N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V
W, H, ts = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
for N in (1, 2, 4, 8):
    slots = V.partition_slots(W, H, ts, N)
    buf = torch.zeros((slots, ts, ts, 4), dtype=torch.float16, device="cuda")
    res = []
    for r in range(N):
        for _ in range(5): pipe.record_partition(ctx, ts, r, N, buf.data_ptr())
        ctx.sync(); ctx.timer_begin()
        for _ in range(50): pipe.record_partition(ctx, ts, r, N, buf.data_ptr())
        ctx.timer_end(); res.append(ctx.timer_elapsed_ms() / 50 * 1e3)
    print(f"tile {ts}: N={N}: per-rank march us: max {max(res):.1f} min {min(res):.1f} -> speedup bound {res[0] if N==1 else 0:.0f}", [round(x, 1) for x in res])
ctx.close()
