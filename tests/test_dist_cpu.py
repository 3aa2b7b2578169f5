"""CPU suite, part 3: the N > 1 path on the gloo backend, world_size 2 (one process per rank).
Each rank fills its compact batch buffer ([slot][frame][ts][ts], the layout vk_render_batch writes) from the
oracle's frames (test double for the HIP march), the production TorchTileGather moves the active prefix, and the
root un-tiles with the numpy statement of vk_untile_batch: the result must be the oracle frames."""
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
        dts = (1.0, 0.5, 0.75)  # a batch of three frames
        B = len(dts)
        tx, _ = D.tiles_xy(W, H, ts)
        cap = D.n_slots(W, H, ts, world)
        send = torch.zeros((cap, B, ts, ts, 4), dtype=torch.float32)
        for b, dt in enumerate(dts):
            for j, t in enumerate(D.local_tiles(W, H, ts, rank, world)):
                x0, y0 = (t % tx) * ts, (t // tx) * ts
                full, _, _ = O.render(cam, vol, W, H, dt_scale=dt, tile=(x0, y0, ts, ts), want_counts=False)
                tile = full[y0:y0 + ts, x0:x0 + ts]
                send[j, b, :tile.shape[0], :tile.shape[1]] = torch.from_numpy(np.ascontiguousarray(tile))
        tg = D.TorchTileGather(root=0)
        recv = torch.zeros((world, cap, B, ts, ts, 4), dtype=torch.float32) if rank == 0 else None
        tg.gather(send, recv)
        if rank == 0:
            frames = D.untile_batch_reference(recv.numpy(), W, H, ts)
            for b, dt in enumerate(dts):
                ref, _, _ = O.render(cam, vol, W, H, dt_scale=dt, want_counts=False)
                q.put((b, float(np.abs(frames[b] - ref).max()), bool((frames[b] == ref).all())))
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
    assert all(g[2] for g in got), got  # bit-identical to the single-process oracle frames
