"""How many of a frame's 64x64 tiles does a partition march and move (the silhouette cull), C2 camera."""
import sys; sys.path.insert(0,'/root/repo')
import vokselis_amd as V
W,H=1920,1080
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,)*3); ctx.update()
print("active tiles, slots@8:", ctx.partition_active(64, 8), "of", ((W + 63) // 64) * ((H + 63) // 64))
