"""The C-ABI entry points no other GPU module calls by name: surface queries, device-resident uploads, the timers, the debug hooks -- and
that reshaping a context over and over does not leak device memory.  Through the shared library, on an MI355X box."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import V, _synced  # noqa: F401  (V: fixture)

pytestmark = pytest.mark.gpu


def _fog(n, seed=11):
    return np.random.default_rng(seed).integers(20, 60, (n, n, n), dtype=np.uint8)


def _camera(V, W, H, k=0):
    return V.Camera(1.0 + 0.1 * k, 0.5, 1.1, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()


class _DevicePtr:
    """A raw device pointer as something torch.as_tensor reads without copying."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def test_backbuffer_info_follows_the_surface(V):
    lib, W, H = V.native.lib(), 96, 64
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        w, h, fmt, ptr = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_void_p()
        V.native.check(ctx.handle, lib.vk_backbuffer_info(ctx.handle, C.byref(w), C.byref(h), C.byref(fmt), C.byref(ptr)))
        assert (w.value, h.value, fmt.value) == (W, H, V.OUT_RGBA32F) and ptr.value
        # every out-pointer is optional
        V.native.check(ctx.handle, lib.vk_backbuffer_info(ctx.handle, None, None, None, None))
        ctx.resize_backbuffer(128, 40, V.OUT_RGBA16F)
        V.native.check(ctx.handle, lib.vk_backbuffer_info(ctx.handle, C.byref(w), C.byref(h), C.byref(fmt), C.byref(ptr)))
        assert (w.value, h.value, fmt.value) == (128, 40, V.OUT_RGBA16F) and ptr.value
        # the pointer IS the surface vk_render writes: what torch reads through it equals vk_readback
        import torch

        V.VolumeTexture(ctx, _fog(32))
        ctx.set_camera_blob(_camera(V, 128, 40))
        V.RaycastPipeline(dt_scale=1.0).record(ctx)
        img = ctx.read_backbuffer()
        ctx.sync()
        through_ptr = torch.as_tensor(_DevicePtr(ptr.value, (40, 128, 4), "<f2"), device="cuda").cpu().numpy()
        assert (through_ptr.view(np.uint16) == img.view(np.uint16).reshape(40, 128, 4)).all()
        assert np.isfinite(img.astype(np.float32)).all() and img.astype(np.float32)[..., :3].max() > 0
    finally:
        ctx.close()


@pytest.mark.parametrize("layout", ["LIN", "P8", "S8"])
def test_upload_from_device_memory_equals_upload_from_the_host(V, layout):
    import torch

    from gpu_helpers import layouts

    lib, W, H, n = V.native.lib(), 96, 64, 48
    vol = _fog(n)
    lay = layouts(V)[layout]
    frames = []
    for from_device in (False, True):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            if from_device:
                d = _synced(torch.from_numpy(vol).cuda())
                V.native.check(ctx.handle, lib.vk_volume_upload_device(ctx.handle, d.data_ptr(), None, n, n, n, V.FMT_R8_UNORM, lay))
                ctx.sync()
                del d  # the library keeps its own re-laid copy
            else:
                V.VolumeTexture(ctx, vol, layout=lay)
            ctx.set_camera_blob(_camera(V, W, H))
            V.RaycastPipeline(dt_scale=0.5).record(ctx)
            frames.append(ctx.read_backbuffer())
        finally:
            ctx.close()
    assert (frames[0].view(np.uint32) == frames[1].view(np.uint32)).all()
    assert frames[0][..., :3].max() > 0


def test_upload_from_device_memory_refuses_what_upload_refuses(V):
    lib = V.native.lib()
    ctx = V.Context(64, 64, backbuffer=(64, 64), out_format=V.OUT_RGBA32F)
    try:
        assert lib.vk_volume_upload_device(ctx.handle, None, None, 8, 8, 8, V.FMT_R8_UNORM, V.LAYOUT_AUTO) != 0
        import torch

        d = _synced(torch.zeros(8 * 8 * 8, dtype=torch.uint8, device="cuda"))
        assert lib.vk_volume_upload_device(ctx.handle, d.data_ptr(), None, 0, 8, 8, V.FMT_R8_UNORM, V.LAYOUT_AUTO) != 0
        assert lib.vk_volume_upload_device(ctx.handle, d.data_ptr(), None, 8, 8, 8, 99, V.LAYOUT_AUTO) != 0
        assert lib.vk_volume_upload_device(ctx.handle, d.data_ptr(), None, 8, 8, 8, V.FMT_RGBA16F_PAIR, V.LAYOUT_AUTO) != 0  # needs the second array
    finally:
        ctx.close()


def test_empty_fraction_counts_the_transparent_cells(V):
    """bonsai's transfer function (raycast_naive.wgsl:106-108) is exactly 0 for v <= 0.1 of 255... the library reports the share of cells it may
    skip; on a volume with 16^3 holes of value 10 that share is the holes' (cells whose 2x2x2 taps are all transparent: the holes' interiors)."""
    from gpu_helpers import _holes_volume

    lib, n = V.native.lib(), 64
    frac = C.c_double(-1.0)
    ctx = V.Context(64, 64, backbuffer=(64, 64), out_format=V.OUT_RGBA32F)
    try:
        assert lib.vk_volume_empty_fraction(ctx.handle, C.byref(frac)) != 0  # no volume yet
        assert lib.vk_volume_empty_fraction(ctx.handle, None) != 0
        vol = _holes_volume(n, 0.5)
        V.VolumeTexture(ctx, vol, layout=V.LAYOUT_PACKED)
        V.native.check(ctx.handle, lib.vk_volume_empty_fraction(ctx.handle, C.byref(frac)))
        share = float((vol == 10).mean())
        assert 0.0 < frac.value <= share + 1e-9, (frac.value, share)
        assert frac.value > 0.5 * share, (frac.value, share)  # 16^3 holes: at least the interiors (15/16)^3 less the block granularity
        # nothing to skip in a volume without a transparent cell, everything in one that is transparent throughout
        V.VolumeTexture(ctx, np.full((n, n, n), 40, np.uint8), layout=V.LAYOUT_PACKED)
        V.native.check(ctx.handle, lib.vk_volume_empty_fraction(ctx.handle, C.byref(frac)))
        assert frac.value == 0.0
        V.VolumeTexture(ctx, np.full((n, n, n), 10, np.uint8), layout=V.LAYOUT_PACKED)
        V.native.check(ctx.handle, lib.vk_volume_empty_fraction(ctx.handle, C.byref(frac)))
        assert frac.value == 1.0
    finally:
        ctx.close()


def test_timer_brackets_the_launches_between_begin_and_end(V):
    W, H = 512, 512
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        V.VolumeTexture(ctx, _fog(64))
        ctx.set_camera_blob(_camera(V, W, H))
        pipe = V.RaycastPipeline(dt_scale=0.25)
        pipe.record(ctx)
        ctx.sync()

        def bracket(n):
            ctx.timer_begin()
            for _ in range(n):
                pipe.record(ctx)
            ctx.timer_end()
            return ctx.timer_elapsed_ms()

        one = min(bracket(1) for _ in range(7))
        eight = min(bracket(8) for _ in range(7))
        assert 0.0 < one < 50.0
        assert eight > 2.0 * one, (one, eight)  # eight launches, in order on one stream (a launch's fixed cost is paid once per bracket)
        assert eight < 8.0 * one + 1.0, (one, eight)
        # an empty bracket is (almost) nothing
        assert 0.0 <= min(bracket(0) for _ in range(3)) < max(one, 0.05)
    finally:
        ctx.close()


def test_debug_tile_order_takes_permutations_only_and_the_frame_does_not_depend_on_it(V):
    lib, W, H, ts = V.native.lib(), 256, 192, 32
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        ctx.set_param("render_tile", ts)
        V.VolumeTexture(ctx, _fog(64))
        ctx.set_camera_blob(_camera(V, W, H, 10))  # from far enough for clear tiles around the silhouette
        pipe = V.RaycastPipeline(dt_scale=0.5)
        pipe.record(ctx)
        ref = ctx.read_backbuffer()
        order = ctx.partition_order(ts)
        n = order.size
        assert n == (W // ts) * (H // ts) and sorted(order.tolist()) == list(range(n))
        as_u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))  # noqa: E731
        n_act, _ = ctx.partition_active(ts)
        assert 0 < n_act < n, (n_act, n)  # the volume's silhouette covers a part of the frame
        rng = np.random.default_rng(5)
        shuffled = np.concatenate([rng.permutation(order[:n_act]), rng.permutation(order[n_act:])]).astype(np.uint32)
        for perm in (np.concatenate([order[:n_act][::-1], order[n_act:][::-1]]).astype(np.uint32), shuffled):
            V.native.check(ctx.handle, lib.vk_debug_set_tile_order(ctx.handle, as_u32(perm), n))
            assert (ctx.partition_order(ts) == perm).all()  # the table in use is the one set
            V.native.check(ctx.handle, lib.vk_backbuffer_clear(ctx.handle))
            pipe.record(ctx)
            assert (ctx.read_backbuffer().view(np.uint32) == ref.view(np.uint32)).all()
        bad = order.copy(); bad[0] = bad[1]  # a tile twice, another one never
        assert lib.vk_debug_set_tile_order(ctx.handle, as_u32(bad), n) != 0
        bad = order.copy(); bad[3] = n  # no such tile
        assert lib.vk_debug_set_tile_order(ctx.handle, as_u32(bad), n) != 0
        bad = shuffled.copy(); bad[[0, n - 1]] = bad[[n - 1, 0]]  # a permutation, but an active tile behind the marched positions
        assert lib.vk_debug_set_tile_order(ctx.handle, as_u32(bad), n) != 0
        assert lib.vk_debug_set_tile_order(ctx.handle, as_u32(order), n - 1) != 0
        assert lib.vk_debug_set_tile_order(ctx.handle, None, n) != 0
        # a refused table leaves the one in use alone
        pipe.record(ctx)
        assert (ctx.read_backbuffer().view(np.uint32) == ref.view(np.uint32)).all()
    finally:
        ctx.close()


def test_wave_trace_stamps_every_marched_block(V):
    lib, W, H = V.native.lib(), 128, 128
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, _fog(48), layout=V.LAYOUT_LINEAR)
        ctx.set_camera_blob(_camera(V, W, H))
        n_blocks = (W // 8) * (H // 8)
        out = (C.c_uint64 * (4 * n_blocks))()
        assert lib.vk_debug_wave_trace(ctx.handle, 0, out, n_blocks) != 0  # nothing traced yet
        V.native.check(ctx.handle, lib.vk_debug_wave_trace(ctx.handle, 1, None, 0))
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_NO_SKIP).record(ctx)
        steps = ctx.read_steps()
        V.native.check(ctx.handle, lib.vk_debug_wave_trace(ctx.handle, 0, out, n_blocks))
        rec = np.frombuffer(out, np.uint64).reshape(n_blocks, 4)
        stamped = rec[:, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)
        # every block whose rays marched is stamped (blocks outside the volume's screen hull leave before the stamp)
        marched = int((steps.reshape(H // 8, 8, W // 8, 8).sum(axis=(1, 3)) > 0).sum())
        assert 0 < marched <= int(stamped.sum()) <= n_blocks
        r = rec[stamped]
        assert (r[:, 0] <= r[:, 1]).all()  # start <= end
        assert ((r[:, 2] >> np.uint64(32)) < 8).all()  # XCC_ID: 8 XCDs
        assert len(np.unique(r[:, 2] >> np.uint64(32))) > 1  # and the launch used more than one of them
        trips = r[:, 3] & np.uint64((1 << 20) - 1)
        assert int((trips > 0).sum()) == marched  # wave-level march-loop trips: exactly the blocks with a ray that iterated
        # the wave-level trips of a block are at least its longest ray's iterations
        assert int(trips.sum()) >= int(steps.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).sum())
        assert lib.vk_debug_wave_trace(ctx.handle, 0, out, 10 ** 9) != 0  # more blocks than were traced
    finally:
        ctx.close()


def test_reshaping_a_context_again_and_again_does_not_leak_device_memory(V):
    import torch

    W, H = 640, 360
    vol = _fog(64)

    def cycle(ctx, i):
        k = 1 + i % 4
        ctx.frames_in_flight(k)
        ctx.resize_backbuffer(W + 64 * (i % 3), H + 40 * (i % 2), V.OUT_RGBA16F if i % 2 else V.OUT_RGBA32F)
        ctx.set_camera_blob(_camera(V, W, H, i))
        if i % 5 == 0:
            V.VolumeTexture(ctx, vol, layout=V.LAYOUT_AUTO if i % 10 else V.LAYOUT_STAGED)
        pipe = V.RaycastPipeline(dt_scale=1.0, flags=V.RENDER_PRESENT if i % 3 == 0 else 0)
        for _ in range(k + 1):
            f = ctx.frame_begin()
            pipe.record(ctx)
            ctx.frame_end()
        ctx.frame_wait(f)

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, vol)
        for i in range(12):  # every shape once: the allocator's pools are warm after this
            cycle(ctx, i)
        before = free_bytes()
        for i in range(120):
            cycle(ctx, i)
        after = free_bytes()
        assert before - after < (8 << 20), f"{(before - after) / 2**20:.1f} MiB of device memory lost over 120 reshapes"
    finally:
        ctx.close()
    # ... and contexts themselves: create, render, destroy
    before = free_bytes()
    for i in range(20):
        c = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
        try:
            c.frames_in_flight(1 + i % 4)
            V.VolumeTexture(c, vol)
            c.set_camera_blob(_camera(V, W, H, i))
            V.RaycastPipeline().record(c)
            c.sync()
        finally:
            c.close()
    assert before - free_bytes() < (8 << 20)
