"""Frame time with a camera that changes every frame (orbit): the per-camera host work (tile order, cull
rectangle, table upload) is on the frame's critical path here."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
pipe = V.RaycastPipeline(dt_scale=0.5)
for k in range(20): pipe.record(ctx)
ctx.sync()
K = 300
t0 = time.perf_counter()
for k in range(K): pipe.record(ctx)
ctx.sync(); t_static = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for k in range(K):
    ctx.camera.set_yaw(1.0 + 0.002 * k); ctx.update()
    pipe.record(ctx)
ctx.sync(); t_orbit = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for k in range(K):
    ctx.camera.set_yaw(1.0 + 0.002 * k); ctx.update()
t_host = (time.perf_counter() - t0) / K * 1e3
print(f"static camera {t_static:.4f} ms/frame; orbiting camera {t_orbit:.4f} ms/frame (camera math + upload alone {t_host:.4f})")
ctx.close()
