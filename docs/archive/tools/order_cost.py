"""Host cost of a new camera: set camera + heaviest-first tile order + table upload, per estimate-ray count."""
import sys, os, time, json
sys.path.insert(0, '/root/repo')
import vokselis_amd as V
W, H = 1920, 1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3); ctx.update()
N = 400
blobs = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 2048, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(N)]
for rays in (3, 2, 1):
    ctx.set_param("order_rays", rays)
    ctx.sync(); t0 = time.perf_counter()
    for j in range(N):
        ctx.set_camera_blob(blobs[j]); ctx.partition_active(64, 1)
    dt = (time.perf_counter() - t0) / N * 1e6
    ctx.sync()
    print(json.dumps({"order_rays": rays, "us_per_new_camera(set_camera + order + upload)": round(dt, 2)}))
t0 = time.perf_counter()
for j in range(N):
    ctx.set_camera_blob(blobs[j])
print("set_camera_blob alone us", round((time.perf_counter() - t0) / N * 1e6, 2))
t0 = time.perf_counter()
for j in range(N):
    ctx.partition_active(64, 1)
print("partition_active memoised us", round((time.perf_counter() - t0) / N * 1e6, 2))
ctx.close()
