"""Per-rank march THROUGHPUT of the C2 frame dealt to N ranks with F frames in flight (single-GPU emulation of
rank r of N; no gather): us per frame for the slowest rank."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vokselis_amd as V
W, H, ts = 1920, 1080, 64
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F, stream=main.cuda_stream)
    V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
    pipe = V.RaycastPipeline(dt_scale=0.5)
    side = [torch.cuda.Stream() for _ in range(F)]
    for N in (1, 2, 4, 8):
        slots = V.partition_slots(W, H, ts, N)
        bufs = [torch.zeros((slots, ts, ts, 4), dtype=torch.float16, device="cuda") for _ in range(F)]
        res, cpu = [], []
        for r in range(N):
            for k in range(2 * F): pipe.record_partition(ctx, ts, r, N, bufs[k % F].data_ptr(), stream=side[k % F].cuda_stream)
            torch.cuda.synchronize()
            K = 400
            t0 = time.perf_counter()
            for k in range(K): pipe.record_partition(ctx, ts, r, N, bufs[k % F].data_ptr(), stream=side[k % F].cuda_stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append((t2 - t0) / K * 1e6); cpu.append((t1 - t0) / K * 1e6)
        print(f"F={F} N={N}: us/frame per rank: max {max(res):.1f} min {min(res):.1f} (host enqueue {max(cpu):.1f}); N=1-serial/this = {174.0 / max(res):.2f}x")
    ctx.close()
