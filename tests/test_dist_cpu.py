"""CPU suite, part 3: the N > 1 path on the gloo backend, world_size 2 (one process per rank).
Each rank fills its compact tile buffer from the oracle's frame (test double for the HIP march),
the production FrameGather moves it, and the root un-tiles: the result must be the oracle frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, ts, q):
    import torch.distributed as dist

    from oracle import oracle as O
    from vokselis_amd import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol = O.volume_standin_u8(32)
        cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
        # two frames per collective call, two batches in flight through the two buffer sets
        B = 2
        dts = (1.0, 0.5, 0.75)  # 3 frames: one full batch + one partial (flushed with count = 1)
        fg = D.FrameGather(W, H, ts, torch.float32, torch.device("cpu"), root=0, batch=B)
        tx, _ = D.tiles_xy(W, H, ts)
        compact, gathered = fg.buffers()
        works = []
        for k, dt in enumerate(dts):
            g, b = divmod(k, B)
            buf = compact[g % 2][b].numpy()
            for j, t in enumerate(D.local_tiles(W, H, ts, rank, world)):
                x0, y0 = (t % tx) * ts, (t // tx) * ts
                full, _, _ = O.render(cam, vol, W, H, dt_scale=dt, tile=(x0, y0, ts, ts), want_counts=False)
                tile = full[y0:y0 + ts, x0:x0 + ts]
                buf[j, :tile.shape[0], :tile.shape[1]] = tile
            if b == B - 1 or k == len(dts) - 1:
                works.append(fg.start(g % 2, None, b + 1))
        for w in works:
            w.wait()
        if rank == 0:
            for k, dt in enumerate(dts):
                g, b = divmod(k, B)
                frame = D.untile_reference(gathered[g % 2][:, b].numpy(), W, H, ts)
                ref, _, _ = O.render(cam, vol, W, H, dt_scale=dt, want_counts=False)
                q.put((k, float(np.abs(frame - ref).max()), bool((frame == ref).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H,ts", [(96, 64, 16), (72, 40, 32)])
def test_gloo_world2_tile_gather(W, H, ts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, ts, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(3))
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[2] for g in got), got  # bit-identical to the single-process oracle frame
