"""Run one raycast configuration N times (for rocprofv3).  usage: prof_case.py <standin|fog> <skip|noskip> [iters] [W H dt]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from oracle import oracle as O

which, skip = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W, H, dt = (int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080, 0.5)
layout = V.LAYOUT_LINEAR if skip == "linear" else (V.LAYOUT_PACKED_PAIRS if skip.endswith("16") else V.LAYOUT_PACKED)
vol = O.volume_standin_u8(256) if which == "standin" else O.volume_fog_u8(256)
cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture(ctx, vol, layout=layout)
ctx.update()
pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=dt, flags=(V.RENDER_NO_SKIP if skip.startswith("noskip") else 0))
for _ in range(iters):
    pipe.record(ctx)
ctx.sync()
ctx.close()
