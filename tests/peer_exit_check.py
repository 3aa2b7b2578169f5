"""TEST INFRASTRUCTURE: one rank of a two-rank job (both on GPU 0, rendezvous and tiles over gloo) in which rank 1 ENDS before a gather.
The survivor must find out within BatchTileRenderer's time limit: PeerLostError from submit() / flush(), exit code 3 -- never a hang.
Started twice by tests/test_peer_loss_gpu.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment."""
import datetime
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import vokselis_amd as V
from vokselis_amd.dist import BatchTileRenderer, PeerLostError

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
W, H = 320, 180
ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (64,) * 3)
cams = [V.Camera(1.0, 0.5, 1.0 + 0.1 * j, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(4)]
ctx.set_camera_blob(cams[0])
got = []
btr = BatchTileRenderer(ctx, V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR), tile_size=32, batch=2, root=0, transport="torch", via_host=True, root_skip=0,
                        timeout_s=8.0, on_batch=lambda first, n, fr: got.append((first, n)))
# one healthy batch: both ranks march, gather, the root un-tiles
for c in cams[:2]:
    btr.submit(c)
btr.flush()
dist.barrier()
if rank == 0:
    assert got == [(0, 2)], got
    print("peer_exit_check: healthy batch delivered", flush=True)
if rank == 1:
    print("peer_exit_check: rank 1 ends now, before the next gather", flush=True)
    os._exit(0)  # (no clean-up, no goodbye: a process that died)
t0 = time.monotonic()
try:
    for c in cams[2:]:
        btr.submit(c)
    btr.flush()
except PeerLostError as e:
    print("peer_exit_check: survivor raised PeerLostError after %.1f s: %s" % (time.monotonic() - t0, str(e)[:200]), flush=True)
    try:
        btr.close()  # (a dead renderer closes without draining)
        ctx.close()
    finally:
        os._exit(3)
print("peer_exit_check: the gather with a dead peer RETURNED", flush=True)
os._exit(1)
