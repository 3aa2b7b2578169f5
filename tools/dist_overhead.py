"""CPU/launch overhead of one TileParallelRenderer.submit() (world = 1, tiny frame): the floor of the N > 1 path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import vokselis_amd as V
from vokselis_amd.dist import TileParallelRenderer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for (W, H) in ((256, 144), (1920, 1080)):
        cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
        ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F, stream=stream.cuda_stream)
        V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
        pipe = V.RaycastPipeline(dt_scale=0.5)
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        tpr = TileParallelRenderer(ctx, pipe, tile_size=64, batch=B)
        for k in range(50): tpr.submit(k)
        tpr.flush(); torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        for k in range(K): tpr.submit(k)
        t_cpu = time.perf_counter() - t0
        tpr.flush(); torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        for k in range(K): pipe.record(ctx)
        t_cpu2 = time.perf_counter() - t0
        ctx.sync(); t_all2 = time.perf_counter() - t0
        print(f"{W}x{H}: submit() CPU {t_cpu / K * 1e6:.1f} us/frame, end-to-end {t_all / K * 1e6:.1f} us/frame | plain record CPU {t_cpu2 / K * 1e6:.1f}, end-to-end {t_all2 / K * 1e6:.1f}")
        if W == 1920:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            for k in range(500): tpr.submit(k)
            pr.disable(); tpr.flush(); torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
        ctx.close()
dist.destroy_process_group()
