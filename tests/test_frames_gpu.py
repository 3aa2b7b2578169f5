"""GPU suite, frames in flight (vk_ctx_frames_in_flight / vk_frame_*): the reference's submission model -- one pass per frame, the
queue running ahead of the GPU (src/lib.rs:178-194, src/context.rs:118,252) -- on a ring of K frame surfaces, each on its own stream.

Bars: every frame bitwise the frame the same calls produce with one surface (and within 1e-4 of the oracle); a frame's surface is not
written again before the frame K later begins; error codes, never crashes, for misuse.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
W, H, DT = 480, 270, 0.5


@pytest.fixture(scope="module")
def V(hip_built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU: the gpu suite must run on an MI355X box")
    import vokselis_amd

    return vokselis_amd


def orbit(V, n, aspect=W / H, target=(0.5, 0.5, 0.5), zoom=1.0, pitch=0.5):
    return [V.Camera(zoom, pitch + 0.02 * j, 1.0 + 0.05 * j, target, aspect).get_proj_view_matrix() for j in range(n)]


def bonsai_ctx(V, k=1, out=None):
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F if out is None else out)
    V.VolumeTexture.generate_standin(ctx, (256,) * 3)
    if k > 1:
        ctx.frames_in_flight(k)
    return ctx


def test_frames_in_flight_bitwise_and_oracle(V, O):
    cams = orbit(V, 7)
    pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT)
    # one surface, one stream: the frames every ring has to reproduce
    ctx = bonsai_ctx(V)
    ref, shots = [], []
    try:
        for cb in cams:
            ctx.set_camera_blob(cb)
            pipe.record(ctx)
            ctx.render()
            ref.append(ctx.read_backbuffer())
            shots.append(ctx.capture_frame()[0])
    finally:
        ctx.close()
    want, wsteps, _ = O.render(cams[0], O.volume_standin_u8(256), W, H, dt_scale=DT)
    assert np.abs(ref[0] - want).max() <= TOL
    for k in (1, 2, 3, 4):
        ctx = bonsai_ctx(V, k)
        try:
            ids, ptrs = [], []
            for j, cb in enumerate(cams):
                ctx.set_camera_blob(cb)  # a camera the caller did not know a frame earlier
                fid = ctx.frame_begin()
                pipe.record(ctx)
                ctx.render()
                ctx.frame_end()
                ids.append(fid)
                ptrs.append(ctx.frame_info(fid)["backbuffer"])
                if j >= 1 and k >= 2:
                    # the hazard: frame j (another camera) is in flight behind frame j - 1; j - 1's surface holds j - 1's pixels
                    prev = ctx.read_frame(ids[j - 1])
                    assert (prev.view(np.uint8) == ref[j - 1].view(np.uint8)).all(), (k, j)
            assert ids == list(range(ids[0], ids[0] + len(cams)))
            # the ring: k distinct surfaces, frame j + k on frame j's
            assert len(set(ptrs[:k])) == k and all(ptrs[j] == ptrs[j - k] for j in range(k, len(cams)))
            for j in range(len(cams)):
                if j >= len(cams) - k:  # still held
                    got = ctx.read_frame(ids[j])
                    assert (got.view(np.uint8) == ref[j].view(np.uint8)).all(), (k, j)
                    assert ctx.capture_frame_of(ids[j])[0] == shots[j], (k, j)
                    assert ctx.frame_info(ids[j])["complete"]
                else:  # its surface went to frame j + k
                    with pytest.raises(V.VokselisError):
                        ctx.read_frame(ids[j])
                    ctx.frame_wait(ids[j])  # (completed long ago: not an error)
            # the current surface is the last frame's
            assert (ctx.read_backbuffer().view(np.uint8) == ref[-1].view(np.uint8)).all()
            with pytest.raises(V.VokselisError):
                ctx.frame_wait(ids[-1] + 1)
        finally:
            ctx.close()


def test_frames_in_flight_tile_loop(V):
    """The xor example's frame -- 18 tile dispatches with offsets (examples/xor/main.rs:235-254) -- with frames in flight: every tile
    launch takes a slot of the tile-order ring (16 slots), so the ring comes round INSIDE every frame and the next frame, on another
    stream, rewrites slots the previous frame's launches read."""
    w, h, ts = 1280, 720, 256
    cams = [V.Camera(3.0, -0.5 + 0.03 * j, 1.0 + 0.1 * j, (0.0, 0.0, 0.0), w / h).get_proj_view_matrix() for j in range(5)]
    offsets = [(x * ts, y * ts) for y in range(h // ts + 1) for x in range(w // ts + 1)]
    assert len(offsets) == 18
    pipe = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
    frames = {}
    for k in (1, 3):
        ctx = V.Context(w, h, backbuffer=(w, h), out_format=V.OUT_RGBA16F)
        try:
            V.VolumeTexture.generate_xor(ctx, (64,) * 3, 0.0)
            if k > 1:
                ctx.frames_in_flight(k)
            got = []
            ids = []
            for cb in cams:
                ctx.set_camera_blob(cb)
                ids.append(ctx.frame_begin())
                for (x, y) in offsets:
                    pipe.record(ctx, (x, y, ts, ts))
                ctx.frame_end()
                if k == 1:
                    got.append(ctx.read_backbuffer())
            if k > 1:
                got = [None] * (len(cams) - k) + [ctx.read_frame(i) for i in ids[-k:]]
            frames[k] = got
        finally:
            ctx.close()
    for j in range(len(cams) - 3, len(cams)):
        assert (frames[3][j].view(np.uint8) == frames[1][j].view(np.uint8)).all(), j
    assert frames[1][0].astype(np.float32)[..., :3].max() > 0.1  # (something was drawn)


def test_crowded_ring_variants_are_bitwise_equal(V):
    """A frame of a ring in which three frames execute (k = 4) is launched with the leaner kernel variants -- the naive march without probe-ahead,
    the compute twin with its smallest request ring: the frames are those of one surface, bit for bit."""
    cases = [("bonsai", 1920, 1080, lambda c: V.VolumeTexture.generate_standin(c, (256,) * 3), V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5))),
             ("xor", 1280, 720, lambda c: V.VolumeTexture.generate_xor(c, (128,) * 3, 0.0), V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)))]
    for name, w, h, mk, mode, dt, (z, p, y, t) in cases:
        cams = [V.Camera(z, p + 0.02 * j, y + 0.07 * j, t, w / h).get_proj_view_matrix() for j in range(6)]
        pipe = V.RaycastPipeline(mode, dt_scale=dt)
        ctx = V.Context(w, h, backbuffer=(w, h), out_format=V.OUT_RGBA16F)
        try:
            mk(ctx)
            ref = []
            for cb in cams:
                ctx.set_camera_blob(cb)
                pipe.record(ctx)
                ref.append(ctx.read_backbuffer())
            ctx.frames_in_flight(4)
            ids = []
            for cb in cams:
                ctx.set_camera_blob(cb)
                ids.append(ctx.frame_begin())
                pipe.record(ctx)
                ctx.frame_end()
            for j in range(2, 6):
                assert (ctx.read_frame(ids[j]).view(np.uint16) == ref[j].view(np.uint16)).all(), (name, j)
        finally:
            ctx.close()


def test_frames_in_flight_misuse_is_an_error_code(V):
    import torch

    ctx = bonsai_ctx(V)
    try:
        for bad in (0, V.native.MAX_FRAMES_IN_FLIGHT + 1):
            with pytest.raises(V.VokselisError):
                ctx.frames_in_flight(bad)
        with pytest.raises(V.VokselisError):
            ctx.frame_end()  # no frame open
        ctx.frames_in_flight(2)
        with pytest.raises(V.VokselisError):
            ctx.set_stream(torch.cuda.Stream().cuda_stream)  # the ring's surfaces run on streams of the context's own
        ctx.set_camera_blob(orbit(V, 1)[0])
        fid = ctx.frame_begin()
        with pytest.raises(V.VokselisError):
            ctx.frame_begin()  # one frame open at a time
        with pytest.raises(V.VokselisError):
            ctx.frames_in_flight(3)
        with pytest.raises(V.VokselisError):
            ctx.resize_backbuffer(64, 64)
        with pytest.raises(V.VokselisError):
            ctx.read_frame(fid)  # still open
        V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT).record(ctx)
        ctx.frame_end()
        a = ctx.read_frame(fid)
        # a new volume while frames are in flight: the ring is drained, the next frame sees the new volume
        V.VolumeTexture.generate_fog(ctx, (64,) * 3)
        f2 = ctx.frame_begin()
        V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT).record(ctx)
        ctx.frame_end()
        b = ctx.read_frame(f2)
        assert not (a == b).all()
        # a smaller ring, then one surface again: earlier ids are forgotten, the context keeps working
        ctx.frames_in_flight(1)
        with pytest.raises(V.VokselisError):
            ctx.read_frame(f2)
        f3 = ctx.frame_begin()
        V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT).record(ctx)
        ctx.frame_end()
        assert f3 > f2 and (ctx.read_frame(f3).view(np.uint8) == b.view(np.uint8)).all()
        # resize with a ring: every surface takes the new shape
        ctx.frames_in_flight(3)
        ctx.resize_backbuffer(96, 64)
        ids = []
        for _ in range(3):
            ids.append(ctx.frame_begin())
            V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT).record(ctx)
            ctx.frame_end()
        got = [ctx.read_frame(i) for i in ids]
        assert got[0].shape == (64, 96, 4) and all((g.view(np.uint8) == got[0].view(np.uint8)).all() for g in got)
    finally:
        ctx.close()
    # a context on a caller's stream cannot grow a ring
    s = torch.cuda.Stream()
    ctx = V.Context(W, H, backbuffer=(W, H), stream=s.cuda_stream)
    try:
        with pytest.raises(V.VokselisError):
            ctx.frames_in_flight(2)
        ctx.frames_in_flight(1)
    finally:
        ctx.close()


def test_run_headless_in_flight(V):
    """run::<D> (src/lib.rs:45-208) with the queue running ahead: the last frame equals the one-surface run's."""

    class Bonsai(V.Demo):
        @classmethod
        def init(cls, ctx):
            self = cls()
            self.volume = V.VolumeTexture.generate_standin(ctx, (64,) * 3)
            self.pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR)
            return self

        def update(self, ctx):
            ctx.camera.add_yaw(0.1)  # a mouse drag (src/lib.rs:166-171)

        def render(self, ctx):
            self.pipe.record(ctx)

    shots = {}
    for k in (1, 3):
        seen = []
        cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 160 / 90)
        ctx, _ = V.run_headless(Bonsai, frames=5, camera=cam, width=160, height=90, backbuffer=(160, 90), in_flight=k,
                                on_frame=lambda c, fid: seen.append(fid))
        try:
            assert seen == [1, 2, 3, 4, 5]
            shots[k] = (ctx.read_backbuffer(), ctx.capture_frame()[0])
        finally:
            ctx.close()
    assert (shots[1][0].view(np.uint8) == shots[3][0].view(np.uint8)).all() and shots[1][1] == shots[3][1]
    # the same loop with the present pass in the raycast pass's epilogue (Context.fuse_present): context.render records nothing, the
    # captured frame is the two-pass one to within one 8-bit step on the f32-off-centre columns / rows
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 160 / 90)
    ctx, _ = V.run_headless(Bonsai, frames=5, camera=cam, width=160, height=90, backbuffer=(160, 90), in_flight=4, fuse_present=True)
    try:
        fused = np.frombuffer(ctx.capture_frame()[0], np.uint8).astype(np.int32)
        assert (ctx.read_backbuffer().view(np.uint8) == shots[1][0].view(np.uint8)).all()
    finally:
        ctx.close()
    d = np.abs(fused - np.frombuffer(shots[1][1], np.uint8).astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() > 0.99


# ---- present fused into the raycast pass's epilogue (VK_RENDER_PRESENT*) ------------------------------------------------------


def _shot(ctx, fid=None):
    buf, dims = ctx.capture_frame() if fid is None else ctx.capture_frame_of(fid)
    rows = np.frombuffer(buf, np.uint8).reshape(dims.height, dims.padded_bytes_per_row)
    return rows[:, :dims.unpadded_bytes_per_row].reshape(dims.height, dims.width, 4).copy()


def _centred(n):
    """Columns (rows) of an n-wide present at the backbuffer's own size whose f32 sample coordinate (present.wgsl's uv * size - 0.5, as
    vk_present and the oracle compute it) is exactly the texel's centre."""
    x = np.arange(n, dtype=np.float32)
    uv = (x + np.float32(0.5)) / np.float32(n)
    u = (uv.astype(np.float64) * np.float64(n) - 0.5).astype(np.float32)  # fmaf(uv, n, -0.5): the product is exact in binary64
    return (np.floor(u) == x) & (u - np.floor(u) == 0)


def _fused_cases(V):
    naive = lambda c: V.VolumeTexture.generate_standin(c, (256,) * 3)
    return [
        # (name, w, h, volume, mode, dt, camera, tiles or None)
        ("bonsai 720p", 1280, 720, naive, V.MODE_NAIVE_TRILINEAR, 1.0, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5)), None),
        ("bonsai 480x270", 480, 270, naive, V.MODE_NAIVE_TRILINEAR, 0.5, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5)), None),
        ("bonsai 203x117: ragged blocks on both edges", 203, 117, naive, V.MODE_NAIVE_TRILINEAR, 1.0, (1.0, 0.5, 1.0, (0.5, 0.5, 0.5)), None),
        ("xor 333x95", 333, 95, lambda c: V.VolumeTexture.generate_xor(c, (64,) * 3, 0.0), V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)), None),
        ("staged", 320, 180, lambda c: V.VolumeTexture.generate_fog(c, (96,) * 3, layout=V.LAYOUT_STAGED, dense_core=True), V.MODE_NAIVE_TRILINEAR, 0.5,
         (1.0, 0.5, 1.0, (0.5, 0.5, 0.5)), None),
        ("xor single", 1280, 720, lambda c: V.VolumeTexture.generate_xor(c, (128,) * 3, 0.0), V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)), None),
        ("xor tiles", 1280, 720, lambda c: V.VolumeTexture.generate_xor(c, (128,) * 3, 0.0), V.MODE_COMPUTE_NEAREST, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)),
         [(x * 256, y * 256, 256, 256) for y in range(3) for x in range(6)]),
        ("procedural", 160, 90, lambda c: None, V.MODE_PROCEDURAL, 1.0, (3.0, -0.5, 1.0, (0.0, 0.0, 0.0)), None),
    ]


def test_present_fused_equals_render_then_present(V, O):
    """VK_RENDER_PRESENT: the pass's own lanes apply ACES + sRGB and write the Rgba8 (and Bgra8) targets.  Against vk_render followed by
    vk_present at the backbuffer's size: the HDR backbuffer bitwise; the presented image equal on every pixel vk_present samples at a
    texel centre, within one 8-bit step on the few whose f32 coordinate lands ~1e-7 off centre; within one step of the oracle's present
    everywhere.  VK_RENDER_PRESENT_ONLY leaves the backbuffer alone."""
    for name, w, h, mkvol, mode, dt, cam, tiles in _fused_cases(V):
        for fmt in (V.OUT_RGBA16F, V.OUT_RGBA32F):
            ctx = V.Context(w, h, backbuffer=(w, h), out_format=fmt)
            try:
                mkvol(ctx)
                z, p, y, t = cam
                ctx.set_camera_blob(V.Camera(z, p, y, t, w / h).get_proj_view_matrix())

                def draw(flags):
                    pipe = V.RaycastPipeline(mode, dt_scale=dt, flags=flags)
                    for tl in (tiles or [None]):
                        pipe.record(ctx, tl)

                draw(0)
                ctx.render()  # vk_present(w, h)
                bb0, two_pass = ctx.read_backbuffer(), _shot(ctx)
                V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
                draw(V.RENDER_PRESENT | V.RENDER_PRESENT_BGRA)
                bb1, fused = ctx.read_backbuffer(), _shot(ctx)
                assert (bb1.view(np.uint8) == bb0.view(np.uint8)).all(), name
                centre = _centred(h)[:, None] & _centred(w)[None, :]
                d = np.abs(fused.astype(np.int32) - two_pass.astype(np.int32)).max(axis=2)
                assert (d[centre[:d.shape[0], :d.shape[1]]] == 0).all(), (name, fmt)
                assert d.max() <= 1, (name, fmt, d.max())
                want = O.present(bb0.astype(np.float32), w, h)[:fused.shape[0], :fused.shape[1]].astype(np.int32)
                do = np.abs(fused.astype(np.int32) - want)
                assert do.max() <= 1 and (do == 0).mean() > 0.995, (name, fmt, do.max(), (do == 0).mean())
                assert (fused[..., 3] == 255).all() and fused[..., :3].max() > 30
                # PRESENT_ONLY: the backbuffer keeps what it held
                V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
                cleared = ctx.read_backbuffer()
                draw(V.RENDER_PRESENT_ONLY)
                assert (ctx.read_backbuffer().view(np.uint8) == cleared.view(np.uint8)).all(), name
                assert (_shot(ctx) == fused).all(), name
            finally:
                ctx.close()


def test_present_fused_misuse_and_frames_in_flight(V):
    import torch

    w, h = 480, 270
    ctx = bonsai_ctx(V, 3, out=V.OUT_RGBA16F)
    try:
        cams = orbit(V, 5)
        ref = []
        for cb in cams:  # two passes, one surface at a time
            ctx.set_camera_blob(cb)
            fid = ctx.frame_begin()
            V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT).record(ctx)
            ctx.render()
            ctx.frame_end()
            ref.append(_shot(ctx, fid))
        ids = []
        for cb in cams:  # fused, three in flight
            ctx.set_camera_blob(cb)
            ids.append(ctx.frame_begin())
            V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT, flags=V.RENDER_PRESENT).record(ctx)
            ctx.frame_end()
        for j in (2, 3, 4):
            d = np.abs(_shot(ctx, ids[j]).astype(np.int32) - ref[j].astype(np.int32))
            assert d.max() <= 1 and (d == 0).mean() > 0.999, j
        # compact / batched passes have no presented image
        buf = torch.empty((64 * 64 * 64 * 8,), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(V.VokselisError):
            V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT, flags=V.RENDER_PRESENT).record_partition(ctx, 64, 0, 1, buf.data_ptr())
        with pytest.raises(V.VokselisError):
            V.render_batch(ctx, V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT, flags=V.RENDER_PRESENT_ONLY), [cams[0]], buf.data_ptr())
    finally:
        ctx.close()


# ---- C3 with the device's own sine (VK_RENDER_DEVICE_SINE): a tolerance mode with stated bars ---------------------------------------------


def test_procedural_device_sine_tolerance_mode(V, O):
    """What a GPU running xor.wgsl as written computes -- hash = fract(sin(h) * 43758.5) with the HARDWARE sine -- beside the specified sine the
    oracle shares.  The hash amplifies the sine's error at arguments up to ~8e5 into a different noise field, so nothing is compared pixel by
    pixel; the bars are statistical: the same frame in the large (8 x 8-blurred correlation >= 0.95, mean colour of the lit pixels within 2 %),
    mean |d| <= 0.05, and every pixel whose ray misses the box clear-coloured in both.  The specified march stays the one held to the oracle."""
    W, H = 1920, 1080
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H).get_proj_view_matrix()
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        ctx.set_camera_blob(cam)
        ctx.reset_step_counts()
        V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_COUNT).record(ctx)
        spec, steps = ctx.read_backbuffer(), ctx.read_steps()
        V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_DEVICE_SINE).record(ctx)
        dev = ctx.read_backbuffer()
        with pytest.raises(V.VokselisError):
            V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_DEVICE_SINE | V.RENDER_COUNT).record(ctx)
    finally:
        ctx.close()
    # the specified march is the oracle's (a tile of it, here; the whole frame in test_procedural_full_size_tiles)
    tile = (W // 2 - 32, H // 2 - 32, 64, 64)
    ref, rsteps = O.render_procedural(cam, W, H, tile=tile)
    sl = (slice(tile[1], tile[1] + 64), slice(tile[0], tile[0] + 64))
    assert (steps[sl] == rsteps[sl]).all() and np.abs(spec[sl] - ref[sl]).max() <= TOL
    clear = np.array([0.023, 0.02, 0.02, 1.0], np.float32)
    miss = steps == 0
    assert (spec[miss] == clear).all() and (dev[miss] == clear).all() and (dev[..., 3] == 1).all()
    lit = (spec[..., :3] != clear[:3]).any(-1)
    assert 0.05 < lit.mean() < 0.2
    d = np.abs(spec - dev)[..., :3]
    assert d[lit].mean() <= 0.05 and d.max() < 0.5, (d[lit].mean(), d.max())
    assert d.max() > 1e-3  # (it IS another noise field: a frame equal to the specified one would mean the flag did nothing)
    ms, md = spec[lit][:, :3].mean(0), dev[lit][:, :3].mean(0)
    assert np.abs(md / ms - 1.0).max() <= 0.02, (ms, md)
    blur = lambda im: im[..., 0].reshape(H // 8, 8, W // 8, 8).mean((1, 3)).ravel()
    assert np.corrcoef(blur(spec), blur(dev))[0, 1] >= 0.95
