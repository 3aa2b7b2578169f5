"""GPU suite: the framebuffer partition -- tiles dealt over ranks, batched launches, gather, un-tile, the single-process group, the N > 1 branches under the
RCCL stand-in, two-rank rehearsals of bench.py (the reference's tile loop, examples/xor/main.rs:77-95,235-254, generalised).  Frames are bitwise those of one GPU."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import TOL, V, _captured_rgb, _holes_volume, _orbit_cameras, _render_with_params, _synced, gpu_render, layouts  # noqa: F401

pytestmark = pytest.mark.gpu


def test_partition_untile_equals_frame(V, O):
    """Multi-GPU scheme on one GPU: every rank's partition, concatenated, un-tiles to the frame."""
    from vokselis_amd import dist as D

    vol = O.volume_standin_u8(64)
    W, H, ts = 200, 136, 32
    cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ref, _, _ = O.render(cam, vol, W, H, dt_scale=0.5)
    import torch

    for world in (1, 2, 3, 8):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture(ctx, vol)
            ctx.set_camera_blob(cam)
            slots = V.partition_slots(W, H, ts, world)
            gathered = _synced(torch.zeros((world, slots, ts, ts, 4), dtype=torch.float32, device="cuda"))
            pipe = V.RaycastPipeline(dt_scale=0.5)
            for r in range(world):
                pipe.record_partition(ctx, ts, r, world, gathered[r].data_ptr())
            ctx.sync()
            order = ctx.partition_order(ts)
            assert sorted(order.tolist()) == list(range(len(order)))  # a permutation of the tiles
            n_active, n_slots_active = ctx.partition_active(ts, world)
            assert 0 < n_active < len(order) and n_slots_active == -(-n_active // world)
            g_host = gathered.cpu().numpy()
            g_host[:, n_slots_active:] = np.nan  # slots beyond the active ones are never read
            host = D.untile_reference(g_host, W, H, ts, order, n_active)
            assert np.abs(host - ref).max() <= TOL
            V.native.check(ctx.handle, V.native.lib().vk_untile(ctx.handle, gathered.data_ptr(), ts, world, slots))
            img = ctx.read_backbuffer()
            assert (img == host).all()
            # the same partition with colour-only tiles (VK_WIRE_RGB): three quarters of the bytes, the same frame
            ctx.set_wire(V.WIRE_RGB)
            lean = _synced(torch.full((world, slots, ts * ts * 3), np.nan, dtype=torch.float32, device="cuda"))
            for r in range(world):
                pipe.record_partition(ctx, ts, r, world, lean[r].data_ptr())
            V.native.check(ctx.handle, V.native.lib().vk_untile(ctx.handle, lean.data_ptr(), ts, world, slots))
            img = ctx.read_backbuffer()
            assert (img.view(np.uint32) == host.view(np.uint32)).all()
            ctx.set_wire(V.WIRE_RGBA)
        finally:
            ctx.close()


def test_silhouette_cull_never_drops_a_hit_tile(V, O):
    """Tiles the cube's projected silhouette (convex hull of its corners, 2 px of margin) cannot reach are never marched
    nor gathered; the root clears them.  60 seeded cameras -- far, close, grazing, nearly axis-aligned, inside -- at two
    tile sizes: compact partition + vk_untile must give the frame vk_render writes (which marches every pixel), bitwise."""
    import torch

    rng = np.random.default_rng(0xC011)
    vol = O.volume_fog_u8(24, seed=5)
    W, H = 208, 120
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, vol)
        pipe = V.RaycastPipeline(dt_scale=1.0)
        fewer = 0
        for case in range(60):
            zoom = float(rng.choice([0.3, 0.8, 1.0, 1.6, 3.0, 6.0]))
            pitch = float(rng.uniform(-1.5, 1.5)) if case % 5 else float(rng.choice([0.0, 1e-3, 1.5]))
            yaw = float(rng.uniform(0, 6.283)) if case % 7 else float(rng.choice([0.0, 1.5708, 3.1416]))
            tgt = tuple(float(v) for v in (rng.uniform(0.2, 0.8, 3) if case % 3 else (0.5, 0.5, 0.5)))
            ctx.set_camera_blob(O.camera_blob(zoom, pitch, yaw, tgt, W / H))
            pipe.record(ctx)
            whole = ctx.read_backbuffer()
            for ts in (16, 32):
                slots = V.partition_slots(W, H, ts, 1)
                gathered = _synced(torch.full((1, slots, ts, ts, 4), float("nan"), dtype=torch.float32, device="cuda"))
                pipe.record_partition(ctx, ts, 0, 1, gathered.data_ptr())
                n_active, _ = ctx.partition_active(ts, 1)
                V.native.check(ctx.handle, V.native.lib().vk_untile(ctx.handle, gathered.data_ptr(), ts, 1, slots))
                img = ctx.read_backbuffer()
                assert (img == whole).all(), (case, ts, zoom, pitch, yaw, tgt)
                # the hull is at least as tight as the bounding rectangle, and tighter somewhere
                tx, ty = -(-W // ts), -(-H // ts)
                hit = (whole[..., :3] != 0).any(axis=-1)
                touched = sum(bool(hit[j * ts:(j + 1) * ts, i * ts:(i + 1) * ts].any()) for j in range(ty) for i in range(tx))
                assert touched <= n_active <= tx * ty
                if hit.any():
                    ys, xs = np.nonzero(hit)
                    rect = (xs.max() // ts - xs.min() // ts + 1) * (ys.max() // ts - ys.min() // ts + 1)
                    fewer += n_active < rect
        assert fewer > 10  # (the rectangle alone would keep all of them)
    finally:
        ctx.close()


def test_render_batch_equals_single_frames(V, O):
    """vk_render_batch: B frames with B different cameras in ONE launch (whole frames at N = 1; per-rank compact
    tiles + vk_untile_batch for N in {1, 2, 3, 8} emulated on this GPU) -- every frame bitwise equal to vk_render's,
    on the cell layout with skipping (u8), on the staged bricks (f16) and in the compute twin."""
    import torch

    W, H, ts = 320, 200, 32
    cams = [V.Camera(1.0 + 0.05 * k, 0.5 - 0.08 * k, 1.0 + 0.35 * k, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for k in range(6)]
    cams.insert(3, cams[2])  # a repeated camera inside the batch
    cases = [("standin u8 / cells+skip", O.volume_standin_u8(64), None, V.LAYOUT_AUTO, V.MODE_NAIVE_TRILINEAR, V.OUT_RGBA16F, cams),
             ("fog f16 / staged bricks", O.volume_fog_f16(48), None, V.LAYOUT_STAGED, V.MODE_NAIVE_TRILINEAR, V.OUT_RGBA32F, cams)]
    for name, vol, vol2, lay, mode, fmt, cc in cases:
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=fmt)
        try:
            V.VolumeTexture(ctx, vol, vol2, layout=lay)
            pipe = V.RaycastPipeline(mode, dt_scale=0.5)
            tdt = torch.float16 if fmt == V.OUT_RGBA16F else torch.float32
            singles = []
            for c in cc:
                ctx.set_camera_blob(c)
                pipe.record(ctx)
                singles.append(ctx.read_backbuffer().copy())
            B = len(cc)
            frames = _synced(torch.zeros((B, H, W, 4), dtype=tdt, device="cuda"))
            V.render_batch(ctx, pipe, cc, frames.data_ptr(), tile_size=ts)
            ctx.sync()
            got = frames.cpu().numpy()
            for k in range(B):
                assert (got[k].view(np.uint8) == singles[k].view(np.uint8)).all(), (name, "whole frames", k)
            for nr, k in ((1, 0), (2, 0), (3, 2), (8, 3), (8, 0), (2, 5)):
                ctx.set_root_skip(k)
                cap = V.partition_slots(W, H, ts, nr, k)
                gathered = None
                for r in range(nr):
                    buf = _synced(torch.zeros((cap, B, ts, ts, 4), dtype=tdt, device="cuda"))
                    bid, act = V.render_batch(ctx, pipe, cc, buf.data_ptr(), tile_size=ts, rank=r, nranks=nr, compact=True, slot_capacity=cap)
                    if gathered is None:
                        gathered = _synced(torch.zeros((nr, act, B, ts, ts, 4), dtype=tdt, device="cuda"))
                    ctx.sync()
                    gathered[r] = buf[:act]  # what the rank would send: a contiguous prefix
                frames.zero_()
                torch.cuda.synchronize()  # torch's copies and fill (its own stream) before the library reads / writes them
                V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr())
                ctx.sync()
                got = frames.cpu().numpy()
                for j in range(B):
                    assert (got[j].view(np.uint8) == singles[j].view(np.uint8)).all(), (name, "ranks", nr, "root_skip", k, j)
            ctx.set_root_skip(0)
        finally:
            ctx.close()
    # more than eight frames: every XCD takes a run of consecutive frames of a tile position (frame_runs, the default) -- a relabelling of which
    # block renders which frame, for counts that are and are not multiples of eight, whole frames and a partition's compact tiles
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        V.VolumeTexture(ctx, O.volume_standin_u8(64))
        pipe = V.RaycastPipeline(dt_scale=0.5)
        many = [V.Camera(1.0, 0.5, 1.0 + 0.11 * k, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for k in range(24)]
        singles = []
        for c in many:
            ctx.set_camera_blob(c)
            pipe.record(ctx)
            singles.append(ctx.read_backbuffer().copy())
        for B in (9, 19, 24):
            for runs in (1, 0):
                ctx.set_param("frame_runs", runs)
                # (the partition's tiles also as colour only -- VK_WIRE_RGB: (r, g) plane + b plane per record, alpha restored by the un-tile)
                wire = V.WIRE_RGB if runs else V.WIRE_RGBA
                ch = 3 if wire == V.WIRE_RGB else 4
                ctx.set_wire(wire)
                assert ctx.wire_pixel_bytes == 2 * ch
                frames = _synced(torch.zeros((B, H, W, 4), dtype=torch.float16, device="cuda"))
                V.render_batch(ctx, pipe, many[:B], frames.data_ptr(), tile_size=ts)
                ctx.sync()
                got = frames.cpu().numpy()
                for k in range(B):
                    assert (got[k].view(np.uint8) == singles[k].view(np.uint8)).all(), ("frame runs", runs, B, "whole frames", k)
                nr = 3
                cap = V.partition_slots(W, H, ts, nr, 0)
                gathered = None
                for r in range(nr):
                    buf = _synced(torch.full((cap, B, ts * ts * ch), 7.0, dtype=torch.float16, device="cuda"))
                    bid, act = V.render_batch(ctx, pipe, many[:B], buf.data_ptr(), tile_size=ts, rank=r, nranks=nr, compact=True, slot_capacity=cap)
                    if gathered is None:
                        gathered = _synced(torch.zeros((nr, act, B, ts * ts * ch), dtype=torch.float16, device="cuda"))
                    ctx.sync()
                    gathered[r] = buf[:act]
                frames.zero_()
                torch.cuda.synchronize()
                V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr())
                ctx.sync()
                got = frames.cpu().numpy()
                for k in range(B):
                    assert (got[k].view(np.uint8) == singles[k].view(np.uint8)).all(), ("frame runs", runs, B, "ranks", nr, k)
        ctx.set_param("frame_runs", 1)
        # a batch dealt in one wire format (the last one above: whole pixels) is not un-tiled in another
        ctx.set_wire(V.WIRE_RGB)
        with pytest.raises(V.VokselisError):
            V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr())
        ctx.set_wire(V.WIRE_RGBA)
    finally:
        ctx.close()
    # the compute twin (records layout) and the procedural mode (no volume) through the same batched launch
    xcams = [V.Camera(3.0 + 0.1 * k, -0.5 + 0.1 * k, 1.0 + 0.4 * k, (0.0, 0.0, 0.0), W / H).get_proj_view_matrix() for k in range(4)]
    for mode, dt in ((V.MODE_COMPUTE_NEAREST, 1.0), (V.MODE_PROCEDURAL, 3.0)):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            if mode == V.MODE_COMPUTE_NEAREST:
                V.VolumeTexture.generate_xor(ctx, (64, 64, 64), 0.0)
            pipe = V.RaycastPipeline(mode, dt_scale=dt)
            singles = []
            for c in xcams:
                ctx.set_camera_blob(c)
                pipe.record(ctx)
                singles.append(ctx.read_backbuffer().copy())
            frames = _synced(torch.zeros((len(xcams), H, W, 4), dtype=torch.float32, device="cuda"))
            V.render_batch(ctx, pipe, xcams, frames.data_ptr(), tile_size=ts)
            ctx.sync()
            got = frames.cpu().numpy()
            for j in range(len(xcams)):
                assert (got[j].view(np.uint32) == singles[j].view(np.uint32)).all(), (mode, j)
        finally:
            ctx.close()
    # error behaviour: counters are per frame, capacity is checked
    ctx = V.Context(64, 64, backbuffer=(64, 64), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, O.volume_fog_u8(16))
        buf = _synced(torch.zeros((4, 64, 64, 4), device="cuda"))
        cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0)
        with pytest.raises(V.VokselisError):
            V.render_batch(ctx, V.RaycastPipeline(flags=V.RENDER_COUNT), [cam], buf.data_ptr())
        # whole-frame addressing with nranks > 1 is a rank's share written at its place in frames that live elsewhere (peer-direct tiles,
        # vk_group_peer_direct): two "ranks" of one context fill one buffer, rank 0 clearing the tiles the silhouette cannot reach
        V.render_batch(ctx, V.RaycastPipeline(), [cam] * 4, buf.data_ptr())
        ctx.sync()
        whole = buf.cpu().numpy().copy()
        buf.fill_(-3.0); torch.cuda.synchronize()
        for rk in (1, 0):
            V.render_batch(ctx, V.RaycastPipeline(), [cam] * 4, buf.data_ptr(), tile_size=16, rank=rk, nranks=2)
        ctx.sync()
        assert (buf.cpu().numpy().view(np.uint32) == whole.view(np.uint32)).all()
        with pytest.raises(V.VokselisError):
            V.render_batch(ctx, V.RaycastPipeline(), [cam], buf.data_ptr(), compact=True, slot_capacity=0)
    finally:
        ctx.close()


def test_c2_full_size_batch_and_eight_way_partition(V, O):
    """The headline configuration at its own size (256^3 stand-in, 1920x1080, dt 0.5, rgba16f): frames of a batched launch
    and of an 8-rank partition with the weighted deal (emulated on this GPU, gathered by copies) are bitwise equal to
    vk_render's frames, whose trip counts are the oracle's."""
    import torch

    W, H, ts = 1920, 1080, 64
    cams = [V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix(),
            V.Camera(1.3, 0.2, 2.1, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()]
    cams = [cams[0], cams[0], cams[1], cams[0]]
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        V.VolumeTexture.generate_standin(ctx, (256,) * 3)
        pipe = V.RaycastPipeline(dt_scale=0.5)
        singles = []
        for c in cams:
            ctx.set_camera_blob(c)
            pipe.record(ctx)
            singles.append(ctx.read_backbuffer().view(np.uint16).copy())
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)  # camera 0 again
        steps = ctx.read_steps()
        _, rsteps, _ = O.render(cams[0], O.volume_standin_u8(256), W, H, dt_scale=0.5, tile=(640, 300, 640, 64))
        assert (steps[300:364, 640:1280] == rsteps[300:364, 640:1280]).all()
        B = len(cams)
        frames = _synced(torch.zeros((B, H, W, 4), dtype=torch.float16, device="cuda"))
        V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=ts)
        ctx.sync()
        got = frames.cpu().numpy().view(np.uint16)
        for j in range(B):
            assert (got[j] == singles[j]).all(), ("batch", j)
        nr, k = 8, 2
        ctx.set_root_skip(k)
        cap = V.partition_slots(W, H, ts, nr, k)
        gathered = None
        for r in range(nr):
            buf = _synced(torch.zeros((cap, B, ts, ts, 4), dtype=torch.float16, device="cuda"))
            bid, act = V.render_batch(ctx, pipe, cams, buf.data_ptr(), tile_size=ts, rank=r, nranks=nr, compact=True, slot_capacity=cap)
            if gathered is None:
                gathered = _synced(torch.zeros((nr, act, B, ts, ts, 4), dtype=torch.float16, device="cuda"))
            ctx.sync()
            gathered[r] = buf[:act]
        frames.zero_()
        torch.cuda.synchronize()  # torch's copies and fill (its own stream) before the library reads / writes them
        V.untile_batch(ctx, bid, gathered.data_ptr(), act, frames.data_ptr())
        ctx.sync()
        got = frames.cpu().numpy().view(np.uint16)
        for j in range(B):
            assert (got[j] == singles[j]).all(), ("partition", j)
    finally:
        ctx.close()


def test_batch_tile_renderer_over_rccl_world1(V, O):
    """The N > 1 driver (vokselis_amd.dist.BatchTileRenderer) as a world of one over the library's own RCCL
    communicator (vk_comm_init_rank / vk_gather_tiles): batches of 4 frames, a new camera every frame, a partial
    batch at the end; every delivered frame bitwise equal to vk_render's frame for that camera."""
    import torch
    import torch.distributed as dist

    from vokselis_amd.dist import BatchTileRenderer

    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    W, H = 640, 360
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        V.VolumeTexture.generate_standin(ctx, (128,) * 3)
        cams = _orbit_cameras(V, 11, W / H)
        pipe = V.RaycastPipeline(dt_scale=0.5)
        want = []
        for c in cams:
            ctx.set_camera_blob(c)
            pipe.record(ctx)
            want.append(ctx.read_backbuffer().view(np.uint16).copy())
        got = {}

        def on_batch(first, count, frames):
            f = frames.cpu().numpy().view(np.uint16)
            for j in range(count):
                got[first + j] = f[j].copy()

        with torch.cuda.stream(torch.cuda.Stream()):
            r = BatchTileRenderer(ctx, pipe, tile_size=64, batch=4, transport="rccl", on_batch=on_batch)
            for c in cams:
                r.submit(c)
            r.close()
            ctx.set_stream(None)
        assert sorted(got) == list(range(11))
        for k in range(11):
            assert (got[k] == want[k]).all(), k
        # driven from torch's default stream: the renderer makes (and enters) a stream of its own
        got.clear()
        r = BatchTileRenderer(ctx, pipe, tile_size=64, batch=4, transport="rccl", on_batch=on_batch)
        assert r.march_stream.cuda_stream != 0
        for c in cams:
            r.submit(c)
        r.close()
        ctx.set_stream(None)
        assert sorted(got) == list(range(11))
        for k in range(11):
            assert (got[k] == want[k]).all(), k
    finally:
        ctx.close()
        if created:
            dist.destroy_process_group()


def _btr_two_ranks_worker(rank, world, port, q):
    """One of two processes sharing cuda:0: the production BatchTileRenderer with rank/world = (rank, 2); only the
    wire differs (gloo through host memory -- RCCL refuses two ranks on one device)."""
    import os

    import numpy as np
    import torch
    import torch.distributed as dist

    import vokselis_amd as V
    from vokselis_amd.dist import BatchTileRenderer

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H = 640, 360
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
        V.VolumeTexture.generate_standin(ctx, (128,) * 3)
        cams = [V.Camera(1.0 + 0.03 * k, 0.5 - 0.05 * k, 1.0 + 0.3 * k, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for k in range(11)]
        pipe = V.RaycastPipeline(dt_scale=0.5)
        want = []
        for c in cams:  # (both ranks: with a rotating root either may be handed a batch)
            ctx.set_camera_blob(c)
            pipe.record(ctx)
            want.append(ctx.read_backbuffer().view(np.uint16).copy())
        bad, seen = [], []

        def on_batch(first, count, frames):
            f = frames.cpu().numpy().view(np.uint16)
            for j in range(count):
                seen.append(first + j)
                if not (f[j] == want[first + j]).all():
                    bad.append(first + j)

        with torch.cuda.stream(torch.cuda.Stream()):
            r = BatchTileRenderer(ctx, pipe, tile_size=64, batch=4, transport="torch", via_host=True, on_batch=on_batch if rank == 0 else None)
            for c in cams:
                r.submit(c)
            r.close()
            ctx.set_stream(None)
        fixed = (list(seen), list(bad))
        del seen[:], bad[:]
        # the same stream of frames with a rotating root: launch g is assembled on rank g mod 2
        with torch.cuda.stream(torch.cuda.Stream()):
            r = BatchTileRenderer(ctx, pipe, tile_size=64, batch=4, root="rotate", transport="torch", via_host=True, on_batch=on_batch)
            assert r.root_skip == 0
            for c in cams:
                r.submit(c)
            r.close()
            ctx.set_stream(None)
        ctx.close()
        q.put((rank, fixed[0], fixed[1], list(seen), list(bad)))
    finally:
        dist.destroy_process_group()


def test_batch_tile_renderer_two_ranks_one_gpu(V, O):
    """The N > 1 driver with two real ranks (two processes on this GPU): every frame's tiles dealt to both, batches of
    4 frames with a different camera each, a partial batch; every frame delivered on the root bitwise equal to vk_render's."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_btr_two_ranks_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(300)
        assert p_.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got[0][0] == 0 and got[0][1] == list(range(11)) and got[0][2] == [], got
    assert got[1][0] == 1 and got[1][1] == [], got
    # rotating root: launches 0 and 2 (frames 0-3, 8-10) land on rank 0, launch 1 (frames 4-7) on rank 1, all bitwise equal
    assert got[0][3] == [0, 1, 2, 3, 8, 9, 10] and got[0][4] == [], got
    assert got[1][3] == [4, 5, 6, 7] and got[1][4] == [], got


def test_group_api_and_plain_c_consumer(V, O, tmp_path):
    """vk_group_* (one process, one context per GPU, ncclCommInitAll) on the GPUs this box has, and a plain-C program
    (tests/cabi_smoke.c, gcc, no C++ / HIP headers) linked against the library: both must reproduce vk_render."""
    import ctypes as C
    import os
    import subprocess

    import torch

    L = V.native.lib()
    W, H = 320, 200
    n_gpus = torch.cuda.device_count()
    ords = (C.c_int * n_gpus)(*range(n_gpus))
    g = C.c_void_p()
    assert L.vk_group_create(n_gpus, ords, C.byref(g)) == 0
    try:
        assert L.vk_group_size(g) == n_gpus
        vol = O.volume_standin_u8(48)
        for i in range(n_gpus):
            c = C.c_void_p(L.vk_group_ctx(g, i))
            V.native.check(c, L.vk_backbuffer_resize(c, W, H, V.OUT_RGBA32F))
            V.native.check(c, L.vk_volume_upload(c, vol.ctypes.data, None, 48, 48, 48, V.FMT_R8_UNORM, V.LAYOUT_AUTO))
        cams = _orbit_cameras(V, 5, W / H)
        root = C.c_void_p(L.vk_group_ctx(g, 0))
        out = C.c_void_p()
        V.native.check(root, L.vk_device_alloc(root, 5 * W * H * 16, C.byref(out)))
        rc = L.vk_group_render(g, V.MODE_NAIVE_TRILINEAR, 5, b"".join(cams), 32, 0.5, 0, out)
        assert rc == 0, L.vk_group_last_error(g)
        assert L.vk_group_sync(g) == 0
        got = np.empty((5, H, W, 4), np.float32)
        V.native.check(root, L.vk_device_download(root, got.ctypes.data, out, got.nbytes))
        V.native.check(root, L.vk_device_free(root, out))
    finally:
        L.vk_group_destroy(g)
    for k, cam in enumerate(cams):
        img, _, _ = gpu_render(V, cam, vol, W, H, dt=0.5, want_steps=False)
        assert (img.view(np.uint32) == got[k].view(np.uint32)).all(), k
    # the plain-C consumer
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cabi_smoke")
    lib_dir = os.path.join(root_dir, "vokselis_amd", "_lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(root_dir, "include"), os.path.join(root_dir, "tests", "cabi_smoke.c"),
                    "-L", lib_dir, "-lvokselis_hip", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_smoke: OK" in r.stdout, r.stdout


def test_procedural_partition(V, O):
    """PROCEDURAL needs no volume: the partition calls accept it on a context without one, use the same tile order as
    the render call (one order, not two), and partition + un-tile reproduces the frame."""
    import torch

    W, H, ts = 160, 96, 32
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        ctx.update()
        pipe = V.RaycastPipeline(V.MODE_PROCEDURAL, dt_scale=4.0)
        pipe.record(ctx)
        want = ctx.read_backbuffer().copy()
        act, slots = ctx.partition_active(ts, 2, V.MODE_PROCEDURAL)
        assert act == 15 and slots == 8
        order = ctx.partition_order(ts, V.MODE_PROCEDURAL)
        assert sorted(order.tolist()) == list(range(15))
        cap = V.partition_slots(W, H, ts, 2)
        gathered = _synced(torch.zeros((2, cap, ts, ts, 4), device="cuda"))
        for r in range(2):
            pipe.record_partition(ctx, ts, r, 2, gathered[r].data_ptr())
        V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
        V.native.check(ctx.handle, V.native.lib().vk_untile(ctx.handle, gathered.data_ptr(), ts, 2, cap))
        assert (ctx.read_backbuffer().view(np.uint32) == want.view(np.uint32)).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("how", ["plain", "torchrun"])
def test_bench_multi_rank_flow_rehearsal(V, O, how):
    """bench.py's N > 1 flow end to end with two ranks -- both on this one GPU, rendezvous over gloo, tiles through
    torch.distributed (rehearsal): the weighted deal, the pipelined gather + un-tile, max-over-ranks timing, the contiguous
    >= 100-frame window and the JSON contract.
      plain:    `python bench.py --gpus 2 ...` typed exactly like the N = 1 line -- bench.py starts its own ranks as a child
                torch.distributed.run (and, seeing one GPU for two ranks, rehearses);
      torchrun: the launch line the driver uses for N > 1.
    (The library's RCCL communicator needs one GPU per rank: world 1 in test_batch_tile_renderer_over_rccl_world1, its
    multi-peer branches under the stand-in of test_multi_peer_branches_under_fake_rccl, real peers on the driver's 8-GPU node.)"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VK_BENCH_REHEARSAL")}
    if how == "plain":
        cmd = [sys.executable] + tail
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env["VK_BENCH_REHEARSAL"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["unit"] == "Mray-steps/s" and d["value"] > 0
    assert d["config"]["s_ref_config_camera"] == 148393048  # the C2 frame, as at N = 1
    assert "rehearsal" in d and d["config"]["transport"].startswith("torch.distributed")
    # exactly K steps are timed; a step is one launch of frames_per_launch frames: one contiguous window of >= 100 frames
    assert d["launches_per_region"] == 5 and d["timed_frames"] == 5 * d["frames_per_launch"] >= 100
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 / d["timed_region_s"] - 1.0) < 1e-5 and abs(d["ms_per_frame"] * d["frames_per_launch"] / d["ms_per_step"] - 1.0) < 1e-9
    assert abs(d["timed_region_s"] * d["value"] * 1e6 / (d["config"]["s_ref_per_frame"] * d["timed_frames"]) - 1.0) < 1e-5
    for key in ("metric", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "still_camera", "rotating_root"):
        assert key in d, key
    assert d["rotating_root"]["value"] > 0 and d["config"]["wire"]["format"] == "rgb" and d["config"]["wire"]["bytes_per_pixel"] == 6


@pytest.mark.parametrize("what", ["config_c5", "c5_at_n"])
def test_bench_c5_two_rank_rehearsal(V, O, what):
    """BASELINE's 8-GPU configuration (C5: 2048^3 u8, 3840x2160, replicated volume, framebuffer tiles over the ranks) through bench.py's N > 1
    flow with two ranks on this one GPU (2 x 26 GB of bricks fit): the JSON contract of
      config_c5: `bench.py --gpus 2 --config c5` -- C5 as the line's own workload;
      c5_at_n:   `bench.py --gpus 2` -- the C2 line the driver's scaling run produces, with C5 through the same partition + gather + un-tile in
                 extras.c5_at_n (fixed root and rotating root; a time-limited child job started once the C2 ranks have left their process group),
                 so that the first real 8-GPU run yields BASELINE's own 8-GPU configuration too.
    A test of the flow, not a measurement.  Generalises the reference's tile loop, examples/xor/main.rs:77-95,235-254."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    cmd += ["--config", "c5", "--no-extras"] if what == "config_c5" else ["--no-rotate"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VK_BENCH_REHEARSAL")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "Mray-steps/s" and d["value"] > 0 and "rehearsal" in d
    assert d["launches_per_region"] == 4 and d["timed_frames"] == 4 * d["frames_per_launch"] and d["timed_region_s"] > 0
    assert abs(d["timed_region_s"] / d["steps"] * 1e3 / d["ms_per_step"] - 1.0) < 1e-5
    for key in ("metric", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    if what == "config_c5":
        assert d["config"]["workload"].startswith("C5") and d["frames_per_launch"] * d["launches_per_region"] >= d["timed_frames"]
        # the fog never reaches the early-out: every ray takes its nominal iterations, ~6.2e9 of them per frame
        assert 5.5e9 < d["config"]["s_ref_config_camera"] < 6.8e9 and d["config"]["s_sampled_config_camera"] == d["config"]["s_ref_config_camera"]
        assert d["rotating_root"]["value"] > 0
    else:
        assert d["config"]["s_ref_config_camera"] == 148393048
        c5 = d["extras"]["c5_at_n"]  # (a child `bench.py --gpus 2 --config c5` with a time limit: it cannot take the C2 line down)
        assert "error" not in c5, c5
        assert c5["workload"].startswith("C5") and c5["n_gpus"] == 2 and 5.5e9 < c5["s_ref_per_frame"] < 6.8e9 and "rehearsal" in c5
        for mode in ("fixed_root", "rotating_root"):
            assert c5[mode]["value"] > 0 and c5[mode]["ms_per_frame"] > 0, c5[mode]


def test_multi_peer_branches_under_fake_rccl(V, O):
    """The branches that only run with more than one peer -- vk_group_render's n > 1 path and vk_gather_tiles' root branch --
    executed on this one GPU through a single-process stand-in for RCCL (tests/fake_rccl.cpp, bound via VK_RCCL_LIB): n = 2, 3, 8
    contexts, root_skip 0 / 2 / 3, every frame bitwise equal to vk_render's.  In a child process, so that this process keeps
    the real RCCL for the other tests.  Generalises the reference's tile loop, examples/xor/main.rs:235-254."""
    import os
    import subprocess
    import sys

    import __graft_entry__ as g

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VK_RCCL_LIB=g.build_fake_rccl())
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "shim_multi_rank_check.py")], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "shim_multi_rank_check: OK" in r.stdout, r.stdout
    assert r.stdout.count("vk_group_render n=") == 9 and r.stdout.count("vk_gather_tiles n=") == 5, r.stdout
