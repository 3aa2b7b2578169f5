"""GPU suite: the HIP path, called through the C-ABI, against the CPU oracle and the golden vectors.

Bars: per-channel |dRGBA| <= 1e-4 on the f32 surface (north star); loop trip counts (S_ref) and
the count of tap-fetching steps (S_sampled) are integer work and must be *identical*.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def V(hip_built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU: the gpu suite must run on an MI355X box")
    import vokselis_amd

    return vokselis_amd


def gpu_render(V, cam_blob, vol, W, H, *, dt=1.0, layout=None, flags=0, out=None, tile=None, vol2=None, mode=None,
               want_steps=True):
    out = V.OUT_RGBA32F if out is None else out
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=out)
    try:
        V.VolumeTexture(ctx, vol, vol2, layout=V.LAYOUT_AUTO if layout is None else layout)
        ctx.set_camera_blob(cam_blob)
        ctx.reset_step_counts()
        if not (flags & V.RENDER_NO_SKIP):
            # exercise the skip path whatever the volume's empty share, probing on every trip so that S_sampled is exactly
            # the count of steps that can contribute (the adaptive policy has its own test)
            flags |= V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS
        pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR if mode is None else mode, dt_scale=dt,
                                 flags=flags | (V.RENDER_COUNT if want_steps else 0))
        pipe.record(ctx, tile)
        img = ctx.read_backbuffer()
        steps = ctx.read_steps() if want_steps else None
        counts = ctx.step_counts() if want_steps else None
        if want_steps:
            # the production (uninstrumented) kernel must reproduce the instrumented frame bit for bit
            V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
            V.RaycastPipeline(pipe.mode, dt_scale=dt, flags=flags).record(ctx, tile)
            again = ctx.read_backbuffer()
            if tile is None:
                assert (again.view(np.uint8) == img.view(np.uint8)).all(), "production path differs from the instrumented one"
        return img, steps, counts
    finally:
        ctx.close()


def _synced(t):
    """A tensor torch has just filled on ITS current stream, handed to the library, which writes on a non-blocking stream of
    its own: without a synchronisation nothing orders the fill before the library's kernels (torch's streams and the
    context's do not synchronise with the legacy default stream)."""
    import torch

    torch.cuda.synchronize()
    return t


def layouts(V):
    return {"P8": V.LAYOUT_PACKED, "P16": V.LAYOUT_PACKED_PAIRS, "LIN": V.LAYOUT_LINEAR, "B9": V.LAYOUT_BRICKED, "Q": V.LAYOUT_QUADS,
            "S8": V.LAYOUT_STAGED}


# ---------------------------------------------------------------------------------------------


def test_golden_vectors_every_layout(V, golden, cameras, golden_volumes):
    g = golden["naive_64x64"]
    keys = sorted({k.rsplit("__", 1)[0] for k in g.files})
    for key in keys:
        vname, cname, dts = key.split("__")
        for lname, lay in layouts(V).items():
            img, steps, (s_ref, s_samp) = gpu_render(V, cameras[cname], golden_volumes[vname], 64, 64, dt=float(dts[2:]), layout=lay)
            assert np.abs(img - g[key + "__rgba"]).max() <= TOL, (key, lname)
            assert (steps == g[key + "__steps"]).all(), (key, lname)
            assert s_ref == int(g[key + "__steps"].astype(np.int64).sum())
            if lname in ("P8", "P16"):  # exact empty-space skipping fetches taps only where a tap can contribute
                assert s_samp == int(g[key + "__sampled"].astype(np.int64).sum()), (key, lname)


@pytest.mark.parametrize("W,H,dt,aspect", [(512, 512, 1.0, 1.0), (1920, 1080, 0.5, 16 / 9)])
def test_bonsai_standin_full_frame(V, O, W, H, dt, aspect):
    """BASELINE configs C1 and C2 on the 256^3 stand-in, every pixel, against the oracle."""
    vol = O.volume_standin_u8(256)
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), aspect).get_proj_view_matrix()
    ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=dt)
    for lay in (V.LAYOUT_PACKED_PAIRS, V.LAYOUT_PACKED):
        img, steps, (s_ref, s_samp) = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay)
        assert np.abs(img - ref).max() <= TOL
        assert (steps == rsteps).all()
        assert s_ref == int(rsteps.sum()) and s_samp == int(rsamp.sum())
        assert (img[..., 3] == 1).all()
    miss = rsteps == 0
    assert (img[miss][:, :3] == 0).all()  # clear colour BLACK, alpha 1 (examples/bonsai/main.rs:41)


def test_skip_is_exact(V, O):
    """Size-independent property: skipping changes no pixel bit and no trip count."""
    vol = O.volume_standin_u8(256)
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 16 / 9).get_proj_view_matrix()
    a, sa, (ra, ma) = gpu_render(V, cam, vol, 1920, 1080, dt=0.5, layout=V.LAYOUT_PACKED_PAIRS)
    b, sb, (rb, mb) = gpu_render(V, cam, vol, 1920, 1080, dt=0.5, layout=V.LAYOUT_PACKED_PAIRS, flags=V.RENDER_NO_SKIP)
    c, sc, _ = gpu_render(V, cam, vol, 1920, 1080, dt=0.5, layout=V.LAYOUT_PACKED_PAIRS, flags=V.RENDER_SAFE)
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and (a.view(np.uint32) == c.view(np.uint32)).all()
    assert (sa == sb).all() and (sa == sc).all() and ra == rb
    assert ma < mb == rb  # without skipping every iteration fetches taps


def test_paced_walks_are_exact(V, O):
    """A walk may stop anywhere: what is not skipped now is probed again.  Whatever the caps on a walk's length (in a trip in
    which other lanes sample / in which every lane walks; 0 = none), the frame, the per-pixel iteration counts and the
    number of sampled steps stay what they are -- on the stand-in and on a half-empty volume, u8 and f16, fast and SAFE."""
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 16 / 9).get_proj_view_matrix()
    W, H = 640, 360
    half = _holes_volume(96, 0.5, seed=11)
    for vol, fl in ((O.volume_standin_u8(128), 0), (half, 0), (half, V.RENDER_SAFE), (half.astype(np.float16) / np.float16(255), 0)):
        ref, rsteps, _ = O.render(cam, vol, W, H, dt_scale=0.5)
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture(ctx, vol)
            ctx.set_camera_blob(cam)
            first = None
            for cap, cap_all in ((8, 12), (0, 0), (1, 1), (3, 5), (8, 0), (0, 4), (200, 200)):
                ctx.set_param("walk_cap", cap); ctx.set_param("walk_cap_all", cap_all)
                for policy in (V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS, V.RENDER_FORCE_SKIP):
                    ctx.reset_step_counts()
                    V.RaycastPipeline(dt_scale=0.5, flags=fl | policy | V.RENDER_COUNT).record(ctx)
                    img, steps, counts = ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts()
                    assert (steps == rsteps).all() and counts[0] == int(rsteps.sum()), (cap, cap_all, policy)
                    if first is None:
                        first = (img, counts)
                        assert np.abs(img - ref).max() <= TOL
                    assert (img.view(np.uint32) == first[0].view(np.uint32)).all(), (cap, cap_all, policy)
                    if policy & V.RENDER_PROBE_ALWAYS:
                        assert counts[1] == first[1][1]  # exactly the steps that can contribute, however the walks were cut
        finally:
            ctx.close()


def _holes_volume(n, p_empty, seed=3, block=16):
    """u8 fog 26..40 (every cell contributes) with 16^3 blocks knocked out to value 10 (exactly transparent) with
    probability p_empty: the share of skippable cells is close to p_empty."""
    rng = np.random.default_rng(seed)
    vol = rng.integers(26, 41, (n, n, n), dtype=np.uint8)
    nb = n // block
    holes = rng.random((nb, nb, nb)) < p_empty
    mask = np.repeat(np.repeat(np.repeat(holes, block, 0), block, 1), block, 2)
    vol[mask] = 10
    return vol


def test_adaptive_skip_policy_is_exact(V, O):
    """The default skip policy probes in windows and runs dense stretches where nothing can be skipped (fog), so that
    skipping never costs more than a few per cent.  Whatever it decides the frame is the same bit for bit: against
    the no-skip kernel, the probe-always kernel and the oracle, on fog, on volumes with 20 % and 60 % of their
    cells knocked out, and on the bonsai stand-in; its tap-fetching steps lie between the exact count and S_ref."""
    W, H = 320, 200
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
    vols = {"fog": O.volume_fog_u8(96), "holes20": _holes_volume(96, 0.2), "holes60": _holes_volume(96, 0.6), "standin": O.volume_standin_u8(96),
            "f16core": None}
    z, y, x = np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij")
    r2 = (x - 30) ** 2 + (y - 28) ** 2 + (z - 34) ** 2
    vols["f16core"] = np.where(r2 < 150, 0.95, np.where(r2 < 700, 0.3, 0.05)).astype(np.float16)
    for name, vol in vols.items():
        ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=0.5)
        res = {}
        for mode, fl in (("noskip", V.RENDER_NO_SKIP), ("always", V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS), ("adaptive", V.RENDER_FORCE_SKIP),
                         ("adaptive_safe", V.RENDER_FORCE_SKIP | V.RENDER_SAFE)):
            ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
            try:
                V.VolumeTexture(ctx, vol, layout=V.LAYOUT_PACKED if vol.dtype == np.float16 else V.LAYOUT_PACKED_PAIRS)
                ctx.set_camera_blob(cam)
                ctx.reset_step_counts()
                V.RaycastPipeline(dt_scale=0.5, flags=fl | V.RENDER_COUNT).record(ctx)
                img, steps, (s_ref, s_samp) = ctx.read_backbuffer(), ctx.read_steps(), ctx.step_counts()
                V.RaycastPipeline(dt_scale=0.5, flags=fl).record(ctx)  # the production kernel
                assert (ctx.read_backbuffer().view(np.uint32) == img.view(np.uint32)).all(), (name, mode)
                res[mode] = (img, steps, s_ref, s_samp)
            finally:
                ctx.close()
        for mode in ("always", "adaptive", "adaptive_safe"):
            assert (res[mode][0].view(np.uint32) == res["noskip"][0].view(np.uint32)).all(), (name, mode)
            assert (res[mode][1] == rsteps).all() and res[mode][2] == int(rsteps.sum()), (name, mode)
        assert np.abs(res["adaptive"][0] - ref).max() <= TOL
        assert res["always"][3] == int(rsamp.sum())
        assert res["always"][3] <= res["adaptive"][3] <= res["noskip"][3] == res["noskip"][2], name
        if name in ("holes60", "standin"):
            assert res["adaptive"][3] < 0.8 * res["noskip"][3], name  # it still skips where there is something to skip


def test_fog_never_terminates_early(V, O):
    """C2-fog: alpha/step <= 1.4e-3, so S_ref == S_nominal."""
    vol = O.volume_fog_u8(256)
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 16 / 9).get_proj_view_matrix()
    img, steps, (s_ref, _) = gpu_render(V, cam, vol, 1920, 1080, dt=0.5)
    ref, rsteps, _ = O.render(cam, vol, 1920, 1080, dt_scale=0.5)
    _, nsteps, _ = O.render(cam, vol, 1920, 1080, dt_scale=0.5, flags=O.FLAG_NO_EARLY_OUT, want_counts=True)
    assert (steps == rsteps).all() and (rsteps == nsteps).all()
    assert np.abs(img - ref).max() <= TOL
    assert abs(s_ref / 1.92e8 - 1) < 0.01 and steps.max() == 513


def test_rgba16f_surface(V, O, cameras, golden_volumes):
    """The reference-shaped rgba16float backbuffer: RNE of the f32 result.  Colour
    differs from the oracle in the last f32 ulps, so a value may land on the other side of an f16
    rounding boundary: allow one f16 ulp, require almost all pixels identical."""
    vol = golden_volumes["standin"]
    ref, _, _ = O.render(cameras["bonsai_1x1"], vol, 64, 64)
    want = O.rgba32f_to_rgba16f(ref)
    img, _, _ = gpu_render(V, cameras["bonsai_1x1"], vol, 64, 64, out=V.OUT_RGBA16F)
    got = img.view(np.uint16)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.995
    assert (got[..., 3] == 0x3C00).all()


def test_tiles_and_offscreen(V, O, cameras, golden_volumes):
    """Tile rectangles (A12): any origin, partially or wholly off-screen, compose to the frame."""
    vol = golden_volumes["standin"]
    W, H = 100, 76
    cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ref, rsteps, _ = O.render(cam, vol, W, H)
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, vol)
        ctx.set_camera_blob(cam)
        pipe = V.RaycastPipeline(dt_scale=1.0)
        for ty in range(-16, H + 16, 40):      # 40x40 tiles from a negative origin
            for tx in range(-16, W + 16, 40):
                pipe.record(ctx, (tx, ty, 40, 40))
        pipe.record(ctx, (W + 5, 3, 40, 40))  # wholly off-screen: dropped
        img = ctx.read_backbuffer()
        assert np.abs(img - ref).max() <= TOL
        # a single interior tile leaves the rest of the (cleared) backbuffer alone
        V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
        pipe.record(ctx, (30, 20, 17, 9))
        img = ctx.read_backbuffer()
        mask = np.zeros((H, W), bool); mask[20:29, 30:47] = True
        assert np.abs(img[mask] - ref[mask]).max() <= TOL
        assert (img[~mask] == [0, 0, 0, 1]).all()
    finally:
        ctx.close()


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


def _orbit_cameras(V, n, aspect):
    return [V.Camera(1.0 + 0.03 * k, 0.5 - 0.05 * k, 1.0 + 0.3 * k, (0.5, 0.5, 0.5), aspect).get_proj_view_matrix() for k in range(n)]


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


def test_f16_volume(V, O, golden, cameras):
    g = golden["naive_f16_64x64"]
    vol = O.volume_fog_f16(32)
    for lay in (V.LAYOUT_PACKED, V.LAYOUT_LINEAR, V.LAYOUT_BRICKED, V.LAYOUT_QUADS, V.LAYOUT_STAGED):
        img, steps, _ = gpu_render(V, cameras["bonsai_1x1"], vol, 64, 64, dt=0.5, layout=lay)
        assert np.abs(img - g["rgba"]).max() <= TOL and (steps == g["steps"]).all()
    # a dense-core f16 volume exercises the early-out and the skip map's 0.1 threshold
    z, y, x = np.meshgrid(np.arange(48), np.arange(48), np.arange(48), indexing="ij")
    r2 = (x - 24) ** 2 + (y - 20) ** 2 + (z - 28) ** 2
    core = np.where(r2 < 100, 0.95, np.where(r2 < 400, 0.3, 0.05)).astype(np.float16)
    ref, rsteps, rsamp = O.render(cameras["bonsai_1x1"], core, 96, 96, dt_scale=0.5)
    img, steps, (s_ref, s_samp) = gpu_render(V, cameras["bonsai_1x1"], core, 96, 96, dt=0.5)
    assert np.abs(img - ref).max() <= TOL and (steps == rsteps).all() and s_samp == int(rsamp.sum())


def _render_with_params(V, cam, vol, W, H, dt, layout, params=(), flags=0):
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        for k, v in params:
            ctx.set_param(k, v)
        V.VolumeTexture(ctx, vol, layout=layout)
        ctx.set_camera_blob(cam)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=dt, flags=flags | V.RENDER_COUNT).record(ctx)
        img, steps = ctx.read_backbuffer(), ctx.read_steps()
        census = ctx.simt_census()
        V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
        V.RaycastPipeline(dt_scale=dt, flags=flags).record(ctx)
        assert (ctx.read_backbuffer().view(np.uint32) == img.view(np.uint32)).all(), "production kernel differs from the instrumented one"
        return img, steps, census
    finally:
        ctx.close()


def test_staged_bricks_equal_linear_bitwise(V, O, golden_volumes):
    """VK_LAYOUT_STAGED (8^3 bricks staged through LDS, vk_staged.hpp) against the dense LINEAR kernel: same taps,
    same arithmetic, so frames and trip counts are bitwise equal -- for every window size (LDS budget down to
    one that forces single-step rounds and the global-memory fallback), round length and set of brick copies,
    on u8 and f16 volumes, cubic and not, with cameras outside, inside and axis-aligned."""
    z, y, x = np.meshgrid(np.arange(48), np.arange(40), np.arange(56), indexing="ij")
    r2 = (x - 24) ** 2 + (y - 20) ** 2 + (z - 28) ** 2
    core = np.where(r2 < 100, 0.95, np.where(r2 < 400, 0.3, 0.05)).astype(np.float16)  # [nz=48][ny=40][nx=56]
    vols = {"standin64": O.volume_standin_u8(64), "fog_f16_32": O.volume_fog_f16(32), "core_f16": core,
            "fog_u8_40x24x56": O.volume_fog_u8((40, 24, 56), seed=7, lo=20, span=12), "checker": golden_volumes["checker"]}
    cams = {"bonsai": O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.5),
            "inside": O.camera_blob(0.3, 0.4, 2.5, (0.5, 0.5, 0.5), 1.5),
            "top": O.camera_blob(1.2, 1.45, 0.3, (0.5, 0.5, 0.5), 1.5),
            "far": O.camera_blob(3.0, -0.6, 4.0, (0.5, 0.5, 0.5), 1.5)}
    knobs = {"default": (), "tiny_window": (("stage_cap_bytes", 1024),), "thin_slabs": (("stage_slab_cells", 2), ("stage_cap_bytes", 4096)),
             "thick_slabs": (("stage_slab_cells", 27), ("stage_cap_bytes", 32768)), "copy_x_only": (("stage_copies_mask", 1),),
             "copy_y_only": (("stage_copies_mask", 2),), "copy_z_only": (("stage_copies_mask", 4),),
             # one window per 256-thread group of four waves (raymarch_staged_group_kernel): default budget, a budget that forces thin slabs and
             # the global-memory fallback, thin and thick slabs
             "group": (("stage_group", 1),), "group_tiny_window": (("stage_group", 1), ("stage_cap_bytes", 1024)),
             "group_thin_slabs": (("stage_group", 1), ("stage_slab_cells", 2), ("stage_cap_bytes", 4096)),
             "group_thick_slabs": (("stage_group", 1), ("stage_slab_cells", 27), ("stage_cap_bytes", 16384)), "group_copy_x_only": (("stage_group", 1), ("stage_copies_mask", 1))}
    W, H = 96, 64
    saw_fallback = saw_short = False
    for vname, vol in vols.items():
        for cname, cam in cams.items():
            for dt in (1.0, 0.37):
                ref, rsteps, _ = _render_with_params(V, cam, vol, W, H, dt, V.LAYOUT_LINEAR)
                for kname, params in knobs.items():
                    if kname != "default" and (cname, dt) not in (("bonsai", 0.37), ("top", 1.0)):
                        continue
                    img, steps, cen = _render_with_params(V, cam, vol, W, H, dt, V.LAYOUT_STAGED, params)
                    assert (steps == rsteps).all(), (vname, cname, dt, kname)
                    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), (vname, cname, dt, kname)
                    saw_fallback |= cen["wave_skip_iters"] > 0          # rounds served from global memory
                    saw_short |= kname == "tiny_window" and cen["wave_sample_execs"] < 8 * cen["wave_loop_iters"]  # slabs thinner than asked for
    assert saw_fallback and saw_short  # the degenerate paths were exercised, not just the fast one
    # and against the oracle
    ref, rsteps, _ = O.render(cams["bonsai"], vols["core_f16"], W, H, dt_scale=0.37)
    img, steps, _ = _render_with_params(V, cams["bonsai"], vols["core_f16"], W, H, 0.37, V.LAYOUT_STAGED)
    assert np.abs(img - ref).max() <= TOL and (steps == rsteps).all()


def test_staged_fuzz_dims_cameras_dt(V, O):
    """Seeded fuzz of the staged layout: volume dims that are no multiple of the 8-voxel brick (padding, partial pieces,
    the u8 copy's 16-voxel pieces), anisotropic, down to 3 voxels; cameras anywhere around and inside; dt from 0.11 to
    2.3; default and small LDS budgets.  Frames and trip counts bitwise equal to the dense linear kernel."""
    rng = np.random.default_rng(20261004)
    W, H = 72, 56
    for case in range(24):
        dims = tuple(int(v) for v in rng.choice([3, 5, 9, 17, 23, 31, 37, 50, 64], 3))  # (nx, ny, nz)
        f16 = bool(case & 1)
        vol = O.volume_fog_f16(dims, seed=100 + case) if f16 else O.volume_standin_u8(dims)
        if case % 5 == 0 and not f16:
            vol = O.volume_fog_u8(dims, seed=7 + case, lo=18, span=30)
        cam = O.camera_blob(float(rng.uniform(0.2, 3.5)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-6.0, 6.0)),
                            tuple(float(v) for v in rng.uniform(0.2, 0.8, 3)), W / H)
        dt = float(rng.choice([0.11, 0.37, 0.5, 1.0, 2.3]))
        params = (("stage_cap_bytes", 2048),) if case % 3 == 0 else ()
        if case % 2 == 0 or case % 7 == 0:
            params = params + (("stage_group", 1),)  # windows shared by the four waves of a group
        ref, rsteps, _ = _render_with_params(V, cam, vol, W, H, dt, V.LAYOUT_LINEAR)
        img, steps, _ = _render_with_params(V, cam, vol, W, H, dt, V.LAYOUT_STAGED, params)
        assert (steps == rsteps).all(), (case, dims, dt)
        assert (img.view(np.uint32) == ref.view(np.uint32)).all(), (case, dims, dt)


def test_device_fog_generator_is_bit_identical(V, O):
    cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0)
    for fmt, host in ((V.FMT_R8_UNORM, O.volume_fog_u8((40, 24, 56), seed=99, lo=20, span=12)),
                      (V.FMT_R16_FLOAT, O.volume_fog_f16((40, 24, 56), seed=99))):
        ref, rsteps, _ = O.render(cam, host, 80, 80, dt_scale=0.5)
        ctx = V.Context(80, 80, backbuffer=(80, 80), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture.generate_fog(ctx, (40, 24, 56), fmt=fmt, seed=99)
            ctx.set_camera_blob(cam)
            V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
            assert np.abs(ctx.read_backbuffer() - ref).max() <= TOL and (ctx.read_steps() == rsteps).all()
        finally:
            ctx.close()
    # the dense-core variants: device generator == oracle generator, through the cells and the staged bricks
    for fmt, host in ((V.FMT_R8_UNORM, O.volume_fog_u8((48, 40, 56), seed=7, dense_core=True)),
                      (V.FMT_R16_FLOAT, O.volume_fog_f16((48, 40, 56), seed=7, dense_core=True))):
        ref, rsteps, _ = O.render(cam, host, 80, 80, dt_scale=0.5)
        assert rsteps.max() > 2 * rsteps[40, 40] > 0  # the centre ray ends in the core
        for lay in (V.LAYOUT_AUTO, V.LAYOUT_STAGED):
            ctx = V.Context(80, 80, backbuffer=(80, 80), out_format=V.OUT_RGBA32F)
            try:
                V.VolumeTexture.generate_fog(ctx, (48, 40, 56), fmt=fmt, seed=7, layout=lay, dense_core=True)
                ctx.set_camera_blob(cam)
                V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
                assert np.abs(ctx.read_backbuffer() - ref).max() <= TOL and (ctx.read_steps() == rsteps).all()
            finally:
                ctx.close()
    # the bonsai stand-in made on the device renders bit-identically to the host-uploaded one
    host = O.volume_standin_u8((72, 40, 56), seed=3)
    a, sa, _ = gpu_render(V, cam, host, 80, 80, dt=0.5)
    ctx = V.Context(80, 80, backbuffer=(80, 80), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture.generate_standin(ctx, (72, 40, 56), seed=3)
        ctx.set_camera_blob(cam)
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
        assert (ctx.read_backbuffer() == a).all() and (ctx.read_steps() == sa).all()
    finally:
        ctx.close()


def test_cameras_dims_and_dt(V, O):
    """Non-cubic dims, eye inside the volume, grazing and axis-aligned views, several dt_scale."""
    rng = np.random.default_rng(11)
    for dims, cam_args, W, H, dt in [((40, 64, 24), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.5), 120, 80, 1.0),
                                     ((96, 32, 48), (0.8, -0.3, 4.0, (0.4, 0.5, 0.6), 1.0), 96, 96, 0.7),
                                     ((64, 64, 64), (0.31, 0.2, 2.5, (0.5, 0.5, 0.5), 1.0), 80, 80, 0.25),
                                     ((33, 17, 65), (2.0, 1.2, 0.3, (0.5, 0.5, 0.5), 0.75), 60, 80, 2.0),
                                     ((64, 64, 64), (1.5, 0.0, 0.0, (0.5, 0.5, 0.5), 1.0), 64, 64, 1.0),
                                     ((1, 1, 1), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0), 32, 32, 0.5),
                                     ((2, 3, 5), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0), 32, 32, 0.1),
                                     # very short and very long steps: the skip walk's rounding margins scale with both
                                     ((64, 64, 64), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0), 40, 40, 0.02),
                                     ((128, 96, 64), (1.3, 0.4, 2.0, (0.5, 0.5, 0.5), 1.0), 48, 48, 0.013),
                                     ((48, 48, 48), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0), 48, 48, 3.5)]:
        nx, ny, nz = dims
        vol = (O.volume_standin_u8(dims, seed=5) if min(dims) >= 17 else rng.integers(0, 256, (nz, ny, nx)).astype(np.uint8))
        cam = O.camera_blob(*cam_args)
        ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=dt)
        for lay in (V.LAYOUT_PACKED_PAIRS, V.LAYOUT_PACKED):
            for fl in (0, V.RENDER_SAFE):
                img, steps, (_, s_samp) = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay, flags=fl)
                assert np.abs(img - ref).max() <= TOL, (dims, lay, fl)
                assert (steps == rsteps).all(), (dims, lay, fl)
                assert s_samp == int(rsamp.sum())
        for lay in (V.LAYOUT_BRICKED, V.LAYOUT_QUADS):  # 9^3 dense bricks; 2x2 (y,z) quads, one load per sample
            img, steps, _ = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay)
            assert np.abs(img - ref).max() <= TOL and (steps == rsteps).all(), (dims, lay)


def test_trip_budget_against_the_oracle_loop(V, O):
    """The march carries an integer trip budget computed by vk::count_trips on the device (vokselis_amd/csrc/vk_trips.hpp) instead of the loop variable of
    raycast_naive.wgsl:101.  Through air (alpha exactly 0: no early-out) a pixel's step count IS that trip count, through fog nearly so: per-pixel equality with
    the oracle -- which runs the loop itself, `for (t = t0; t < t1; t += dt)` -- for eyes far outside (t in the binades [2, 4) .. [8, 16)), inside the volume (t starts at 0 and climbs
    through every binade), on a face, and for step lengths from 0.02 to 3 cells; dense march and skip kernel (nothing to skip: forced), two layouts."""
    rng = np.random.default_rng(5)
    n = 48
    vol = rng.integers(26, 32, (n, n, n)).astype(np.uint8)  # fog: alpha per step ~1e-3
    air = rng.integers(0, 26, (n, n, n)).astype(np.uint8)   # air: alpha exactly 0 in every step
    cases = [(6.0, 0.3, 1.0, (0.5, 0.5, 0.5)), (11.0, -0.7, 4.0, (0.5, 0.5, 0.5)), (3.1, 1.3, 2.2, (0.5, 0.5, 0.5)),   # far: t0 ~ 2.6 .. 10.5
             (0.2, 0.4, 0.9, (0.5, 0.5, 0.5)), (0.05, -0.2, 5.0, (0.3, 0.6, 0.5)),                                       # inside: t0 = 0
             (0.5, 0.0, 0.0, (0.5, 0.5, 0.5)), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5))]                                        # eye on the z = 0 face; the bonsai camera
    W, H = 72, 56
    for ci, (zoom, pitch, yaw, target) in enumerate(cases):
        cam = O.camera_blob(zoom, pitch, yaw, target, W / H)
        for dt in ((0.02, 0.37, 1.0, 3.0) if ci % 2 == 0 else (0.11, 0.5, 1.9)) + ((0.003,) if ci in (1, 3) else ()):  # (0.003: ~28 000 iterations per ray)
            ref, rsteps, _ = O.render(cam, vol, W, H, dt_scale=dt)
            for lay, fl in ((V.LAYOUT_PACKED_PAIRS, V.RENDER_NO_SKIP), (V.LAYOUT_PACKED_PAIRS, V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS), (V.LAYOUT_PACKED, V.RENDER_SAFE)):
                img, steps, _ = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay, flags=fl)
                assert (steps == rsteps).all(), (ci, dt, lay, fl, int((steps != rsteps).sum()))
                assert np.abs(img - ref).max() <= TOL, (ci, dt, lay, fl)
            # ... and through air (every tap <= 25: alpha exactly 0, no early-out whatever the length): the dense kernel makes every iteration, the skip
            # kernel makes none of them but walks -- its walks are clamped to the iterations left, so the count it reports is the budget itself
            ref0, rsteps0, _ = O.render(cam, air, W, H, dt_scale=dt)
            for fl in (V.RENDER_NO_SKIP, V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS, 0):
                img, steps, (_, s_samp) = gpu_render(V, cam, air, W, H, dt=dt, layout=V.LAYOUT_PACKED_PAIRS, flags=fl)
                assert (steps == rsteps0).all(), (ci, dt, fl, int((steps != rsteps0).sum()))
                assert (img == ref0).all() and (fl == V.RENDER_NO_SKIP or s_samp == 0), (ci, dt, fl)


def test_skip_fuzz_cameras_dims_dt(V, O):
    """Seeded fuzz of the skip path: random volume dims, cameras (outside, inside, axis-aligned, grazing), image
    sizes and dt_scale; skip == no-skip bitwise, trip counts and tap-fetching steps identical to the oracle."""
    rng = np.random.default_rng(20261003)
    cases = 0
    for trial in range(24):
        dims = tuple(int(x) for x in rng.integers(5, 72, 3))
        W, H = int(rng.integers(24, 96)), int(rng.integers(24, 96))
        kind = trial % 4
        if kind == 0:    # ordinary orbit
            cam_args = (float(rng.uniform(0.7, 2.5)), float(rng.uniform(-1.4, 1.4)), float(rng.uniform(0, 6.28)), (0.5, 0.5, 0.5), W / H)
        elif kind == 1:  # eye inside the volume
            cam_args = (float(rng.uniform(0.05, 0.4)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(0, 6.28)),
                        tuple(float(x) for x in rng.uniform(0.3, 0.7, 3)), W / H)
        elif kind == 2:  # axis-aligned views: direction components that are exactly zero on the centre rays
            cam_args = (1.5, 0.0, float(rng.integers(0, 4)) * 1.5707963, (0.5, 0.5, 0.5), 1.0)
        else:            # grazing: looking along a face
            cam_args = (1.2, float(rng.uniform(-0.05, 0.05)), float(rng.uniform(0, 6.28)), (0.5, float(rng.choice([0.02, 0.98])), 0.5), W / H)
        dt = float(rng.choice([0.15, 0.5, 1.0, 1.7]))
        vol = O.volume_standin_u8(dims, seed=int(rng.integers(1, 1 << 30))) if min(dims) >= 17 else rng.integers(0, 60, (dims[2], dims[1], dims[0])).astype(np.uint8)
        cam = O.camera_blob(*cam_args)
        ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=dt)
        for lay in (V.LAYOUT_PACKED_PAIRS, V.LAYOUT_PACKED):
            a, sa, (_, ma) = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay)
            b, sb, _ = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay, flags=V.RENDER_NO_SKIP)
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), (trial, dims, cam_args, dt, lay)
            assert (sa == rsteps).all() and (sb == rsteps).all(), (trial, dims, cam_args, dt, lay)
            assert ma == int(rsamp.sum()), (trial, dims, cam_args, dt, lay)
            assert np.abs(a - ref).max() <= TOL, (trial, dims, cam_args, dt, lay)
            cases += 1
    assert cases == 48
    # f16 volumes (threshold 0.1 for an empty cell): blobs in air
    for trial in range(8):
        dims = tuple(int(x) for x in rng.integers(9, 60, 3))
        W, H = int(rng.integers(32, 80)), int(rng.integers(32, 80))
        z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
        vol = np.full(x.shape, 0.02, np.float32)
        for _ in range(4):
            c = rng.uniform(0.2, 0.8, 3) * np.array(dims); rad = rng.uniform(2, 0.3 * min(dims))
            d2 = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2
            vol = np.maximum(vol, np.where(d2 < rad * rad, rng.uniform(0.15, 1.0), 0.0))
        vol = vol.astype(np.float16)
        cam = O.camera_blob(float(rng.uniform(0.3, 2.0)), float(rng.uniform(-1.2, 1.2)), float(rng.uniform(0, 6.28)), (0.5, 0.5, 0.5), W / H)
        dt = float(rng.choice([0.3, 0.5, 1.0]))
        ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=dt)
        a, sa, (_, ma) = gpu_render(V, cam, vol, W, H, dt=dt, layout=V.LAYOUT_PACKED)
        b, sb, _ = gpu_render(V, cam, vol, W, H, dt=dt, layout=V.LAYOUT_PACKED, flags=V.RENDER_NO_SKIP)
        assert (a.view(np.uint32) == b.view(np.uint32)).all() and (sa == rsteps).all() and (sb == rsteps).all(), (trial, dims, dt)
        assert ma == int(rsamp.sum()) and np.abs(a - ref).max() <= TOL, (trial, dims, dt)


def test_compute_nearest_mode(V, O, golden, cameras):
    """raycast_compute.wgsl `single` and `tile` (A10-A12) against the golden vectors."""
    g = golden["compute_128x72"]
    den, nrm = g["density"].view(np.float16), g["normals"].view(np.float16)
    # AUTO / PACKED: bricked 16-byte (density, normals) records, pipelined kernel; LINEAR: the two dense volumes
    first = None
    for lay in (V.LAYOUT_AUTO, V.LAYOUT_PACKED, V.LAYOUT_LINEAR):
        img, steps, _ = gpu_render(V, cameras["xor_16x9"], den, 128, 72, vol2=nrm, mode=V.MODE_COMPUTE_NEAREST, layout=lay)
        assert np.abs(img - g["rgba"]).max() <= TOL and (steps == g["steps"]).all(), lay
        first = img if first is None else first
        assert (img.view(np.uint32) == first.view(np.uint32)).all(), "the record layout changes no bit"
    # non-multiple-of-4 dims, a tile that hangs off the image, and long steps (speculative request far outside); the second volume has
    # holes of exactly zero opacity with NaN normals in them (what the xor generator writes where the gradient vanishes), negative
    # opacities and lone contributing voxels: the record kernel's skip map must not change a bit or a count
    rng = np.random.default_rng(3)
    d2 = rng.random((19, 10, 33, 4), np.float32).astype(np.float16); n2 = (rng.random((19, 10, 33, 4), np.float32) * 2 - 1).astype(np.float16)
    d3 = rng.random((40, 27, 33, 4), np.float32); n3 = (rng.random((40, 27, 33, 4), np.float32) * 2 - 1)
    hole = rng.random((40, 27, 33)) < 0.97
    hole[10:30, 5:20, 8:25] = True
    d3[..., 3][hole] = np.where(rng.random(int(hole.sum())) < 0.5, 0.0, -0.25)
    n3[hole & (rng.random((40, 27, 33)) < 0.5)] = np.nan
    d3[20, 12, 16, 3] = 0.9  # a lone voxel deep inside the hole
    d3, n3 = d3.astype(np.float16), n3.astype(np.float16)
    for (dv, nv) in ((d2, n2), (d3, n3)):
        for dt in (1.0, 7.5, 0.3):
            ref, rsteps, _ = O.render(cameras["xor_16x9"], dv, 96, 54, mode=O.MODE_COMPUTE_NEAREST, volume2=nv, dt_scale=dt)
            got = {}
            for lay, fl in ((V.LAYOUT_PACKED, 0), (V.LAYOUT_PACKED, V.RENDER_NO_SKIP), (V.LAYOUT_LINEAR, 0)):
                img, steps, (sr, ss) = gpu_render(V, cameras["xor_16x9"], dv, 96, 54, vol2=nv, mode=V.MODE_COMPUTE_NEAREST, layout=lay, dt=dt, flags=fl)
                assert np.abs(img - ref).max() <= TOL and (steps == rsteps).all(), (dt, lay, fl)
                got[(lay, fl)] = (img, ss)
            assert (got[(V.LAYOUT_PACKED, 0)][0].view(np.uint32) == got[(V.LAYOUT_PACKED, V.RENDER_NO_SKIP)][0].view(np.uint32)).all(), dt
            assert (got[(V.LAYOUT_PACKED, 0)][0].view(np.uint32) == got[(V.LAYOUT_LINEAR, 0)][0].view(np.uint32)).all(), dt
            if dv is d3:
                assert got[(V.LAYOUT_PACKED, 0)][1] <= got[(V.LAYOUT_PACKED, V.RENDER_NO_SKIP)][1] and (dt > 1.0 or got[(V.LAYOUT_PACKED, 0)][1] < got[(V.LAYOUT_PACKED, V.RENDER_NO_SKIP)][1]), dt  # (a wave-level affair: only the big hole is walked, and not with 9.6 voxels per step)
    # the reference's tile loop: (H/256+1) x (W/256+1) offsets, here with 64-px tiles incl. off-screen ones
    ctx = V.Context(128, 72, backbuffer=(128, 72), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture(ctx, den, nrm)
        ctx.set_camera_blob(cameras["xor_16x9"])
        pipe = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
        for y in range(72 // 64 + 1):
            for x in range(128 // 64 + 1):
                pipe.record(ctx, (x * 64, y * 64, 64, 64))
        assert np.abs(ctx.read_backbuffer() - g["rgba"]).max() <= TOL
    finally:
        ctx.close()


def test_compute_fuzz_cameras_dims_dt(V, O):
    """Seeded fuzz of the compute twin's record kernel (request ring + exact skipping): random dims, blobs with exactly-zero and negative
    opacity around them and NaN normals in the holes, cameras outside / inside / axis-aligned, image sizes and dt_scale.  Skip == no skip ==
    the literal twin bitwise, iteration counts identical to the oracle; the same frames in one launch of several (the ring's other shape)."""
    import torch

    rng = np.random.default_rng(20261004)
    for trial in range(12):
        dims = tuple(int(x) for x in rng.integers(8, 48, 3))  # (nx, ny, nz)
        W, H = int(rng.integers(32, 96)), int(rng.integers(32, 96))
        z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
        den = rng.random(x.shape + (4,), np.float32)
        op = np.where(rng.random(x.shape) < 0.5, 0.0, -0.25).astype(np.float32)
        for _ in range(3):
            c = rng.uniform(0.2, 0.8, 3) * np.array(dims); rad = rng.uniform(2, 0.35 * min(dims))
            d2 = (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2
            op = np.where(d2 < rad * rad, rng.uniform(0.2, 1.0), op)
        den[..., 3] = op
        nrm = (rng.random(x.shape + (4,), np.float32) * 2 - 1)
        nrm[(op <= 0) & (rng.random(x.shape) < 0.5)] = np.nan
        den, nrm = den.astype(np.float16), nrm.astype(np.float16)
        kind = trial % 3
        if kind == 0:
            cam_args = (float(rng.uniform(2.0, 4.0)), float(rng.uniform(-1.3, 1.3)), float(rng.uniform(0, 6.28)), (0.0, 0.0, 0.0), W / H)
        elif kind == 1:  # eye inside the box
            cam_args = (float(rng.uniform(0.1, 0.6)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(0, 6.28)), tuple(float(v) for v in rng.uniform(-0.3, 0.3, 3)), W / H)
        else:            # axis-aligned
            cam_args = (3.0, 0.0, float(rng.integers(0, 4)) * 1.5707963, (0.0, 0.0, 0.0), 1.0)
        dt = float(rng.choice([0.3, 1.0, 2.5]))
        cam = O.camera_blob(*cam_args)
        ref, rsteps, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm, dt_scale=dt)
        got = []
        for lay, fl in ((V.LAYOUT_PACKED, 0), (V.LAYOUT_PACKED, V.RENDER_NO_SKIP), (V.LAYOUT_LINEAR, 0)):
            img, steps, _ = gpu_render(V, cam, den, W, H, vol2=nrm, mode=V.MODE_COMPUTE_NEAREST, layout=lay, dt=dt, flags=fl)
            assert (steps == rsteps).all() and np.abs(img - ref).max() <= TOL, (trial, dims, cam_args, dt, lay, fl)
            got.append(img)
        assert (got[0].view(np.uint32) == got[1].view(np.uint32)).all() and (got[0].view(np.uint32) == got[2].view(np.uint32)).all(), (trial, dims, cam_args, dt)
        # three frames in one launch against the three single launches
        cams = [cam] + [O.camera_blob(cam_args[0], cam_args[1], cam_args[2] + 0.3 * k, cam_args[3], cam_args[4]) for k in (1, 2)]
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture(ctx, den, nrm, layout=V.LAYOUT_PACKED)
            pipe = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, dt_scale=dt)
            singles = []
            for c in cams:
                ctx.set_camera_blob(c); pipe.record(ctx); singles.append(ctx.read_backbuffer().copy())
            assert (singles[0].view(np.uint32) == got[0].view(np.uint32)).all(), (trial, "the default policy's frame")
            frames = _synced(torch.zeros((3, H, W, 4), dtype=torch.float32, device="cuda"))
            V.render_batch(ctx, pipe, cams, frames.data_ptr(), tile_size=32)
            ctx.sync()
            out = frames.cpu().numpy()
            for k in range(3):
                assert (out[k].view(np.uint32) == singles[k].view(np.uint32)).all(), (trial, dims, cam_args, dt, "frame", k)
        finally:
            ctx.close()


def test_procedural_mode(V, O, golden, cameras):
    """C3: no volume, density from xor.wgsl's noise_volume at the sample position.  Trip counts are
    integer work and must be identical; RGBA within 1e-4 (measured ~1e-7: the specified sine is shared)."""
    g = golden["procedural_96x54"]
    ctx = V.Context(96, 54, backbuffer=(96, 54), out_format=V.OUT_RGBA32F)
    try:
        ctx.set_camera_blob(cameras["xor_16x9"])          # no volume uploaded, Uniform.time = 0
        for dt, kr, ks in ((1.0, "rgba", "steps"), (2.5, "rgba_dt2p5", "steps_dt2p5")):
            ctx.reset_step_counts()
            V.RaycastPipeline(V.MODE_PROCEDURAL, dt_scale=dt, flags=V.RENDER_COUNT).record(ctx)
            img, steps = ctx.read_backbuffer(), ctx.read_steps()
            assert (steps == g[ks]).all()
            assert np.abs(img - g[kr]).max() <= TOL
            assert ctx.step_counts()[0] == int(g[ks].astype(np.int64).sum())
        # a tile that hangs off the image, uninstrumented, against the oracle at the same time value
        V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
        V.RaycastPipeline(V.MODE_PROCEDURAL).record(ctx, (64, 32, 64, 64))
        ref, _ = O.render_procedural(cameras["xor_16x9"], 96, 54, tile=(64, 32, 64, 64))
        img = ctx.read_backbuffer()
        assert np.abs(img[32:, 64:] - ref[32:, 64:]).max() <= TOL
    finally:
        ctx.close()
    # larger frame, other camera, nonzero Uniform.time (xor.wgsl's un.time)
    W, H = 320, 180
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        ctx.global_uniform.time = 0.75
        V.native.check(ctx.handle, V.native.lib().vk_set_uniform(ctx.handle, ctx.global_uniform.to_bytes()))
        ctx.set_camera_blob(cam.get_proj_view_matrix())
        V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_COUNT).record(ctx)
        img, steps = ctx.read_backbuffer(), ctx.read_steps()
        ref, rsteps = O.render_procedural(cam.get_proj_view_matrix(), W, H, time=0.75)
        assert (steps == rsteps).all() and np.abs(img - ref).max() <= TOL
    finally:
        ctx.close()


def test_error_behaviour(V, O, cameras, golden_volumes):
    ctx = V.Context(64, 64, backbuffer=(64, 64), out_format=V.OUT_RGBA32F)
    try:
        pipe = V.RaycastPipeline()
        with pytest.raises(V.VokselisError, match="no volume"):
            pipe.record(ctx)
        V.VolumeTexture(ctx, golden_volumes["standin"])
        with pytest.raises(V.VokselisError, match="no camera"):
            pipe.record(ctx)
        ctx.set_camera_blob(cameras["bonsai_1x1"])
        with pytest.raises(V.VokselisError, match="dt_scale"):
            V.RaycastPipeline(dt_scale=0.0).record(ctx)
        with pytest.raises(V.VokselisError, match="COMPUTE_NEAREST"):
            V.RaycastPipeline(V.MODE_COMPUTE_NEAREST).record(ctx)
        bad = np.frombuffer(cameras["bonsai_1x1"], np.float32).copy(); bad[5] = np.nan
        with pytest.raises(V.VokselisError, match="non-finite"):
            ctx.set_camera_blob(bad.tobytes())
        with pytest.raises(V.VokselisError, match="no camera"):  # a rejected blob does not linger
            pipe.record(ctx)
        with pytest.raises(ValueError, match="144 bytes"):  # a short buffer never reaches the C side
            ctx.set_camera_blob(cameras["bonsai_1x1"][:100])
        ctx.set_camera_blob(cameras["bonsai_1x1"])
        pipe.record(ctx, (0, 0, 0, 0))  # empty tile is a no-op
        info = ctx.get_info()
        assert info["gfx950"] and info["compute_units"] == 256
        # the wire format of a partition's tiles: an enum of two, 16 or 12 bytes per rgba32f pixel
        with pytest.raises(V.VokselisError, match="VK_WIRE"):
            ctx.set_wire(5)
        assert ctx.wire_pixel_bytes == 16
        ctx.set_wire(V.WIRE_RGB)
        assert ctx.wire_pixel_bytes == 12
        ctx.set_wire(V.WIRE_RGBA)
        with pytest.raises(V.VokselisError, match="trip_log_cap"):
            ctx.set_param("trip_log_cap", 12)  # (a multiple of 8)
    finally:
        ctx.close()


def test_headless_demo_loop(V, O):
    """The reference's frame order: Context.update -> Demo.update -> Demo.render (src/lib.rs:75-79,178-181)."""
    vol = O.volume_standin_u8(64)
    calls = []

    class Bonsai(V.Demo):
        @classmethod
        def init(cls, ctx):
            self = cls()
            self.volume_texture = V.VolumeTexture(ctx, vol)
            self.pipeline = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=1.0)
            calls.append("init")
            return self

        def update(self, ctx):
            calls.append("update")

        def render(self, ctx):
            calls.append("render")
            self.pipeline.record(ctx)

    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 160 / 90)
    ctx, demo = V.run_headless(Bonsai, frames=3, camera=cam, width=160, height=90, backbuffer=(160, 90), out_format=V.OUT_RGBA32F)
    try:
        assert calls == ["init"] + ["update", "render"] * 3
        ref, _, _ = O.render(cam.get_proj_view_matrix(), vol, 160, 90)
        assert np.abs(ctx.read_backbuffer() - ref).max() <= TOL
        buf, dims = ctx.capture_frame()  # run_headless presents after every Demo.render, like the reference
        assert len(buf) == dims.linear_size() and dims.padded_bytes_per_row == 768
    finally:
        ctx.close()


def test_cpp_host_bonsai_example(V, O, tmp_path):
    """The compiled C++ host (vokselis_amd/host: Context / Demo / run_headless / bonsai) drives the same
    C-ABI; its captured frame matches the oracle's frame after the same 8-bit quantisation."""
    import os
    import subprocess

    import __graft_entry__ as g

    g.build_host()
    exe = os.path.join(g.ROOT, "vokselis_amd", "_lib", "bonsai")
    ppm = tmp_path / "bonsai.ppm"
    r = subprocess.run([exe, "--frames", "3", "--size", "320x180", "--dt", "1.0", "--ppm", str(ppm)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Avg frame time" in r.stdout and "gfx950" in r.stdout
    raw = ppm.read_bytes()
    hdr, data = raw.split(b"\n255\n", 1)
    assert hdr == b"P6\n320 180"
    img = np.frombuffer(data, np.uint8).reshape(180, 320, 3).astype(np.int32)
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 320 / 180).get_proj_view_matrix()
    ref, _, _ = O.render(cam, O.volume_standin_u8(256), 320, 180, dt_scale=1.0)
    # the surface is rgba16f: quantise the oracle through f16 first, like the backbuffer, then present
    ref16 = O.rgba32f_to_rgba16f(ref).view(np.float16).astype(np.float32)
    want = O.present(ref16, 320, 180)[..., :3].astype(np.int32)
    d = np.abs(img - want)
    assert d.max() <= 1 and (d == 0).mean() > 0.995
    # the hot path's own surface: the C++ host builds byte-identical camera blobs (one builder), so its
    # f32 frame matches the oracle at the north star's 1e-4 and the kernel's trip counts are the oracle's
    rgba, stp = tmp_path / "rgba.bin", tmp_path / "steps.bin"
    r = subprocess.run([exe, "--frames", "1", "--size", "320x180", "--dt", "1.0", "--f32", "--dump-rgba", str(rgba), "--dump-steps", str(stp)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(rgba, np.float32).reshape(180, 320, 4)
    gsteps = np.fromfile(stp, np.uint32).reshape(180, 320)
    ref, rsteps, _ = O.render(O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 320 / 180), O.volume_standin_u8(256), 320, 180, dt_scale=1.0)
    assert np.abs(got - ref).max() <= TOL and (gsteps == rsteps).all()
    # the same frames through the group API on the GPUs of this box
    import torch
    r = subprocess.run([exe, "--gpus", str(torch.cuda.device_count()), "--frames", "16", "--batch", "4", "--size", "320x180"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Avg frame time" in r.stdout, r.stdout + r.stderr
    # a missing GPU library / device is an error exit, not a silent fallback
    r = subprocess.run([exe, "--raw", "/nonexistent.raw", "--frames", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_present_pass_and_capture_frame(V, O):
    """Next rows N1/N2: present.wgsl (bilinear resample, ACESFilm, branch-free sRGB, Rgba8) and
    capture_frame's byte layout (even-rounded size, 256-B row pitch)."""
    vol = O.volume_standin_u8(64)
    for (bw, bh), (w, h) in [((160, 90), (160, 90)), ((160, 90), (213, 121)), ((128, 72), (64, 36)), ((96, 96), (95, 33))]:
        cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), w / h)
        for fmt in (V.OUT_RGBA16F, V.OUT_RGBA32F):
            ctx = V.Context(w, h, cam, backbuffer=(bw, bh), out_format=fmt)
            try:
                V.VolumeTexture(ctx, vol)
                ctx.update()
                V.RaycastPipeline(dt_scale=1.0).record(ctx)
                ctx.render()
                buf, dims = ctx.capture_frame()
                assert (dims.width, dims.height) == (w - w % 2, h - h % 2) and dims.padded_bytes_per_row % 256 == 0
                assert len(buf) == dims.linear_size()
                rows = np.frombuffer(buf, np.uint8).reshape(dims.height, dims.padded_bytes_per_row)
                got = rows[:, :dims.unpadded_bytes_per_row].reshape(dims.height, dims.width, 4).astype(np.int32)
                assert (rows[:, dims.unpadded_bytes_per_row:] == 0).all()
                want = O.present(ctx.read_backbuffer().astype(np.float32), w, h)[:dims.height, :dims.width].astype(np.int32)
                d = np.abs(got - want)
                assert d.max() <= 1 and (d == 0).mean() > 0.995, ((bw, bh), (w, h), fmt, d.max(), (d == 0).mean())
                assert (got[..., 3] == 255).all()
            finally:
                ctx.close()


def test_xor_generator_and_example(V, O, tmp_path):
    """Next rows N3/N4: shaders/xor.wgsl on the device is bit-identical to the oracle's generator (the
    hash's sine is specified), and the xor example (generator + compute raycast, SinglePass and Tile
    modes) reproduces the oracle's frame."""
    import os
    import subprocess

    import __graft_entry__ as g

    n, W, H = 64, 320, 180
    den, nrm = O.volume_xor(n, 0.0)
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H).get_proj_view_matrix()
    ref, rsteps, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm)
    assert (rsteps > 0).mean() > 0.1
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture.generate_xor(ctx, (n, n, n), 0.0)
        ctx.set_camera_blob(cam)
        V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=V.RENDER_COUNT).record(ctx)
        img, steps = ctx.read_backbuffer(), ctx.read_steps()
        assert (steps == rsteps).all()           # identical volumes -> identical trip counts
        assert np.abs(img - ref).max() <= TOL
        # device-generated == host-uploaded oracle volume, bit for bit
        V.VolumeTexture(ctx, den, nrm)
        V.RaycastPipeline(V.MODE_COMPUTE_NEAREST).record(ctx)
        assert (ctx.read_backbuffer() == img).all()
    finally:
        ctx.close()
    # the compiled example, both modes of examples/xor/main.rs:14-18
    g.build_host()
    exe = os.path.join(g.ROOT, "vokselis_amd", "_lib", "xor")
    ref16 = O.rgba32f_to_rgba16f(ref).view(np.float16).astype(np.float32)
    want = O.present(ref16, W, H)[..., :3].astype(np.int32)
    for mode in ("single", "tile"):
        ppm = tmp_path / f"xor_{mode}.ppm"
        r = subprocess.run([exe, "--frames", "2", "--size", f"{W}x{H}", "--volume", str(n), "--mode", mode, "--ppm", str(ppm)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        data = ppm.read_bytes().split(b"\n255\n", 1)[1]
        got = np.frombuffer(data, np.uint8).reshape(H, W, 3).astype(np.int32)
        d = np.abs(got - want)
        assert d.max() <= 2 and (d == 0).mean() > 0.99, (mode, d.max(), (d == 0).mean())
    # the C3 surrogate through the compiled host: no volume, un.time pinned, presented like any other frame
    refp, _ = O.render_procedural(cam, W, H, time=0.5)
    wantp = O.present(O.rgba32f_to_rgba16f(refp).view(np.float16).astype(np.float32), W, H)[..., :3].astype(np.int32)
    ppm = tmp_path / "xor_procedural.ppm"
    r = subprocess.run([exe, "--frames", "2", "--size", f"{W}x{H}", "--mode", "procedural", "--time", "0.5", "--ppm", str(ppm)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    gotp = np.frombuffer(ppm.read_bytes().split(b"\n255\n", 1)[1], np.uint8).reshape(H, W, 3).astype(np.int32)
    dp = np.abs(gotp - wantp)
    assert dp.max() <= 2 and (dp == 0).mean() > 0.99, (dp.max(), (dp == 0).mean())


def test_large_volume_layouts_agree(V, O):
    """Beyond the cache-resident sizes: a 640^3 u8 fog (262 M voxels) rendered through four independent
    layouts/kernels (9^3 dense bricks, 2x2 quads, 8-B cells, dense linear) gives bitwise-identical frames and
    trip counts, and a tile of it matches the CPU oracle."""
    n, W, H = 640, 960, 540
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
    imgs, steps = {}, {}
    for name, lay in (("b9", V.LAYOUT_BRICKED), ("q", V.LAYOUT_QUADS), ("p8", V.LAYOUT_PACKED), ("lin", V.LAYOUT_LINEAR)):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture.generate_fog(ctx, (n, n, n), seed=0x5EED0005, layout=lay)
            ctx.set_camera_blob(cam)
            V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_FORCE_SKIP).record(ctx)
            imgs[name], steps[name] = ctx.read_backbuffer(), ctx.read_steps()
        finally:
            ctx.close()
    for other in ("q", "p8", "lin"):
        assert (imgs["b9"].view(np.uint32) == imgs[other].view(np.uint32)).all(), other
        assert (steps["b9"] == steps[other]).all(), other
    assert steps["b9"].max() == 2 * n + 1  # dt_scale 0.5: <= 2n+1 iterations
    tile = (448, 238, 64, 64)
    ref, rsteps, _ = O.render(cam, O.volume_fog_u8(n, seed=0x5EED0005), W, H, dt_scale=0.5, tile=tile)
    ys, xs = slice(tile[1], tile[1] + 64), slice(tile[0], tile[0] + 64)
    assert (steps["b9"][ys, xs] == rsteps[ys, xs]).all() and rsteps[ys, xs].min() > 0
    assert np.abs(imgs["b9"][ys, xs] - ref[ys, xs]).max() <= TOL


@pytest.mark.parametrize("name,n,f16,W,H,seed,tile", [
    ("C4", 1024, True, 1920, 1080, 0x5EED0004, (1216, 416, 64, 64)),
    ("C5", 2048, False, 3840, 2160, 0x5EED0005, (2496, 864, 64, 64)),
])
def test_baseline_configs_full_size(V, O, name, n, f16, W, H, seed, tile):
    """BASELINE configs C4 (1024^3 fp16 @1920x1080) and C5 (2048^3 uint8 @3840x2160) at their own size on one GPU:
    AUTO picks the staged 8^3 bricks; the frame and the per-pixel trip counts are bitwise equal through three
    independent layouts/kernels (staged bricks through LDS, dense 9^3 bricks, dense linear -- 64-bit offsets
    everywhere); a 64x64 tile of it matches the CPU oracle on the host-generated volume."""
    import ctypes as C

    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
    fmt = V.FMT_R16_FLOAT if f16 else V.FMT_R8_UNORM
    imgs, steps = {}, {}
    for lname, lay in (("auto", V.LAYOUT_AUTO), ("b9", V.LAYOUT_BRICKED), ("lin", V.LAYOUT_LINEAR)):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture.generate_fog(ctx, (n, n, n), fmt=fmt, seed=seed, layout=lay)
            if lname == "auto":
                got_layout = C.c_int()
                V.native.check(ctx.handle, V.native.lib().vk_volume_info(ctx.handle, None, None, C.byref(got_layout), None))
                assert got_layout.value == V.LAYOUT_STAGED
            ctx.set_camera_blob(cam)
            ctx.reset_step_counts()
            V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
            imgs[lname], steps[lname] = ctx.read_backbuffer(), ctx.read_steps()
            s_ref, s_samp = ctx.step_counts()
            assert s_ref == s_samp == int(steps[lname].astype(np.int64).sum())  # fog: every iteration fetches its taps
            if lname == "auto":  # the production (uninstrumented) kernel gives the same frame
                V.RaycastPipeline(dt_scale=0.5).record(ctx)
                assert (ctx.read_backbuffer().view(np.uint32) == imgs["auto"].view(np.uint32)).all()
                # ... and so does the 8-way partition of it (how BASELINE runs C5): every rank's compact launch emulated here,
                # gathered side by side, un-tiled -- two frames per launch, the second one the same camera again
                import torch

                nr, ts, B = 8, 64, 2
                cap = V.partition_slots(W, H, ts, nr)
                gathered = _synced(torch.zeros((nr, cap, B, ts, ts, 4), dtype=torch.float32, device="cuda"))
                frames = _synced(torch.zeros((B, H, W, 4), dtype=torch.float32, device="cuda"))
                pipe = V.RaycastPipeline(dt_scale=0.5)
                for r in range(nr):
                    bid, act = V.render_batch(ctx, pipe, [cam] * B, gathered[r].data_ptr(), tile_size=ts, rank=r, nranks=nr, compact=True, slot_capacity=cap)
                assert 0 < act <= cap
                ctx.sync()  # the library's launches (its stream) before torch reads their output ...
                packed = gathered[:, :act].contiguous()
                torch.cuda.synchronize()  # ... and torch's copy (its stream) before the library reads it
                V.untile_batch(ctx, bid, packed.data_ptr(), act, frames.data_ptr())
                ctx.sync()
                out = frames.cpu().numpy()
                for b in range(B):
                    assert (out[b].view(np.uint32) == imgs["auto"].view(np.uint32)).all(), (name, "partition", b)
                del gathered, frames, packed
        finally:
            ctx.close()
    for other in ("b9", "lin"):
        assert (imgs["auto"].view(np.uint32) == imgs[other].view(np.uint32)).all(), (name, other)
        assert (steps["auto"] == steps[other]).all(), (name, other)
    assert steps["auto"].max() == 2 * n + 1  # dt_scale 0.5: <= 2n+1 iterations
    host = O.volume_fog_f16(n, seed=seed) if f16 else O.volume_fog_u8(n, seed=seed)
    ref, rsteps, _ = O.render(cam, host, W, H, dt_scale=0.5, tile=tile)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert (steps["auto"][ys, xs] == rsteps[ys, xs]).all()
    assert rsteps[ys, xs].min() > 100  # the tile lies inside the cube's silhouette
    assert np.abs(imgs["auto"][ys, xs] - ref[ys, xs]).max() <= TOL
    # ... and the shader's text AS WRITTEN (VO_FLAG_LITERAL_WGSL: two-rounding coordinate, unfused lerps, smoothstep's divide, libm), on
    # the same tile -- the staged kernels (f16 taps on C4, u8 taps on C5) held to the literal reading: no trip count moves, <= 1e-5
    lit, lsteps, _ = O.render(cam, host, W, H, dt_scale=0.5, tile=tile, flags=O.FLAG_LITERAL_WGSL)
    assert (steps["auto"][ys, xs] == lsteps[ys, xs]).all(), (name, "literal trips")
    assert np.abs(imgs["auto"][ys, xs] - lit[ys, xs]).max() <= 1e-5, (name, np.abs(imgs["auto"][ys, xs] - lit[ys, xs]).max())


@pytest.mark.parametrize("name,n,f16,W,H,seed,tile", [
    ("C4-core", 1024, True, 1920, 1080, 0x5EED0004, (1056, 508, 64, 64)),
    ("C5-core", 2048, False, 3840, 2160, 0x5EED0005, (2160, 1048, 64, 64)),
])
def test_baseline_configs_dense_core_full_size(V, O, name, n, f16, W, H, seed, tile):
    """The "dense-core variant" of C4 / C5 at full size: the fog with a dense ball at the centre, so that the
    rays through the middle of the image leave the loop by the opacity early-out while their neighbours march on.  The
    staged bricks (AUTO) and the dense linear layout give bitwise-identical frames and trip counts (default and forced
    skipping are the same kernel here); a 64x64 tile across the ball's silhouette matches the CPU oracle."""
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
    fmt = V.FMT_R16_FLOAT if f16 else V.FMT_R8_UNORM
    imgs, steps, totals = {}, {}, {}
    for lname, lay in (("auto", V.LAYOUT_AUTO), ("lin", V.LAYOUT_LINEAR)):
        ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
        try:
            V.VolumeTexture.generate_fog(ctx, (n, n, n), fmt=fmt, seed=seed, layout=lay, dense_core=True)
            ctx.set_camera_blob(cam)
            ctx.reset_step_counts()
            V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT).record(ctx)
            imgs[lname], steps[lname] = ctx.read_backbuffer(), ctx.read_steps()
            totals[lname] = ctx.step_counts()
            if lname == "auto":  # the production (uninstrumented) kernel gives the same frame
                V.RaycastPipeline(dt_scale=0.5).record(ctx)
                assert (ctx.read_backbuffer().view(np.uint32) == imgs["auto"].view(np.uint32)).all()
        finally:
            ctx.close()
    assert (imgs["auto"].view(np.uint32) == imgs["lin"].view(np.uint32)).all(), name
    assert (steps["auto"] == steps["lin"]).all() and totals["auto"] == totals["lin"], name
    assert totals["auto"][0] == int(steps["auto"].astype(np.int64).sum())
    centre = int(steps["auto"][H // 2, W // 2])
    assert 0 < centre < n and steps["auto"].max() == 2 * n + 1  # the centre ray stops in the ball; rays beside it cross the cube
    host = O.volume_fog_f16(n, seed=seed, dense_core=True) if f16 else O.volume_fog_u8(n, seed=seed, dense_core=True)
    ref, rsteps, _ = O.render(cam, host, W, H, dt_scale=0.5, tile=tile)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert (steps["auto"][ys, xs] == rsteps[ys, xs]).all()
    assert rsteps[ys, xs].min() > 100 and 2 * rsteps[ys, xs].min() < rsteps[ys, xs].max()  # the tile straddles the silhouette
    assert np.abs(imgs["auto"][ys, xs] - ref[ys, xs]).max() <= TOL
    # the literal reading of the shader on the same tile, where rays END by the alpha >= 0.95 early-out (the trip count is the
    # sensitive quantity here): no trip count moves, <= 1e-5 per channel
    lit, lsteps, _ = O.render(cam, host, W, H, dt_scale=0.5, tile=tile, flags=O.FLAG_LITERAL_WGSL)
    assert (steps["auto"][ys, xs] == lsteps[ys, xs]).all(), (name, "literal trips", int((steps["auto"][ys, xs] != lsteps[ys, xs]).sum()))
    assert np.abs(imgs["auto"][ys, xs] - lit[ys, xs]).max() <= 1e-5, (name, np.abs(imgs["auto"][ys, xs] - lit[ys, xs]).max())


def test_raw_loader_round_trip(V, O, tmp_path):
    """The drop-in loaders for the reference's `bonsai_256x256x256_uint8.raw` (volume_texture.rs:33 include_bytes!, absent
    from the checkout): a synthetic .raw written to disk and loaded through VolumeTexture.from_raw (Python host) and
    `bonsai --raw` (C++ host) renders exactly like the same bytes uploaded directly; a short file is an error."""
    import os
    import subprocess

    import __graft_entry__ as g

    vol = O.volume_standin_u8((128, 128, 64))  # 1 MiB
    small = tmp_path / "vol_128x128x64_uint8.raw"
    vol.tofile(small)
    cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.5)
    want, wsteps, _ = gpu_render(V, cam, vol, 192, 128, dt=0.5)
    ctx = V.Context(192, 128, backbuffer=(192, 128), out_format=V.OUT_RGBA32F)
    try:
        vt = V.VolumeTexture.from_raw(ctx, str(small), dims=(128, 128, 64))
        assert vt.dims == (128, 128, 64)
        ctx.set_camera_blob(cam)
        V.RaycastPipeline(dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_FORCE_SKIP).record(ctx)
        assert (ctx.read_backbuffer().view(np.uint32) == want.view(np.uint32)).all() and (ctx.read_steps() == wsteps).all()
        with pytest.raises(ValueError):
            V.VolumeTexture.from_raw(ctx, str(small), dims=(256, 256, 256))
    finally:
        ctx.close()
    # C++ host: the stand-in written as the reference's 16 MiB file gives the frame the built-in generator gives
    g.build_host()
    exe = os.path.join(g.ROOT, "vokselis_amd", "_lib", "bonsai")
    big = tmp_path / "bonsai_256x256x256_uint8.raw"
    O.volume_standin_u8(256).tofile(big)
    outs = []
    for extra in ([], ["--raw", str(big)]):
        f = tmp_path / ("rgba%d.bin" % len(outs))
        r = subprocess.run([exe, "--frames", "1", "--size", "256x144", "--f32", "--dump-rgba", str(f)] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(np.fromfile(f, np.float32))
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all() and outs[0].max() > 0


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


# ---------------------------------------------------------------------------------------------
# round 4: the reference-held pin and the full-size cases bench.py times


def _captured_rgb(ctx):
    buf, dims = ctx.capture_frame()
    rows = np.frombuffer(buf, np.uint8).reshape(dims.height, dims.padded_bytes_per_row)
    return rows[:, :dims.unpadded_bytes_per_row].reshape(dims.height, dims.width, 4)


def test_volume_png_pin_hip(V, O):
    """The reference's `volume.png` through the HIP path alone: vk_volume_generate_xor (xor.wgsl, t = 0) -> vk_render
    COMPUTE_NEAREST into the 1280x720 backbuffer -> vk_present to the capture's 958x1050 window, at the camera
    oracle/volume_png.py fitted.  Against the committed oracle frame (<= 1 LSB) and against the capture (same loose bars
    as the oracle's own CPU test: background exact, blurred correlation, silhouette box, side of the pink light)."""
    from oracle import volume_png as VP

    pin = VP.load_pin()
    cap = pin["capture_blur_ds"].astype(np.float32)
    want = VP.load_oracle_frame().astype(np.int32)
    for fmt, lsb, same in ((V.OUT_RGBA32F, 1, 0.995), (V.OUT_RGBA16F, 2, 0.97)):  # rgba16f is the reference's own surface (hdr_backbuffer.rs:10)
        ctx = V.Context(VP.WIN_W, VP.WIN_H, backbuffer=(VP.BB_W, VP.BB_H), out_format=fmt)
        try:
            V.VolumeTexture.generate_xor(ctx, (VP.XOR_N,) * 3, 0.0)
            ctx.set_camera_blob(pin["camera"].tobytes())
            V.RaycastPipeline(V.MODE_COMPUTE_NEAREST).record(ctx)
            ctx.render()
            got = _captured_rgb(ctx)
        finally:
            ctx.close()
        assert got.shape == (VP.WIN_H, VP.WIN_W, 4) and (got[..., 3] == 255).all()
        assert (got[0, 0, :3] == VP.BACKGROUND).all() and (got[-1, -1, :3] == VP.BACKGROUND).all()
        d = np.abs(got[..., :3].astype(np.int32) - want)
        assert d.max() <= lsb and (d == 0).mean() > same, (fmt, d.max(), (d == 0).mean())
        m = VP.metrics(got, cap)
        VP.check(m)
        assert abs(m["corr"] - float(pin["corr"])) < 5e-3 and abs(m["mean_abs"] - float(pin["mean_abs"])) < 0.05, m


def test_bonsai_png_pin_hip(V, O):
    """The naive path's reference-held pin through the HIP kernels (oracle/bonsai_png.py): the palette curve -- uniform volumes v = 0 .. 255
    under a saturating ray -- rendered by vk_render equals the oracle's, the colours of the reference's `bonsai.png` lie inside its hull, and
    the capture's greenest colour is the HIP curve's point for v = 179 to one LSB."""
    from oracle import bonsai_png as BP

    pin = BP.load_pin()
    W, H = 16, 9
    ctx = V.Context(W, H, V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H), backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        pipe = V.RaycastPipeline(dt_scale=1.0)

        def one(v):
            V.VolumeTexture(ctx, np.full((32, 32, 32), v, np.uint8))
            ctx.update()
            pipe.record(ctx)
            return ctx.read_backbuffer()[H // 2, W // 2].copy()

        curve = BP.curve_backbuffer(one)
    finally:
        ctx.close()
    ref = BP.oracle_curve()
    assert np.abs(curve - ref).max() <= TOL
    hull = BP.hull_of(curve)
    both = np.vstack([pin["sample"], pin["extremes"]])
    assert BP.inside_share(both, hull, BP.decode_double_srgb) >= BP.BAR_INSIDE
    assert BP.inside_share(both, hull, BP.decode_aces_srgb) <= BP.BAR_INSIDE_ACES
    v, d = BP.nearest_on_curve(BP.GREENEST, curve)
    assert d <= BP.BAR_GREENEST and 170 <= v <= 190, (v, d)


def test_xor_example_full_size(V, O):
    """The reference's own xor configuration at its own size: 256^3 pair volume, 1280x720, camera (3, -0.5, 1, 0)
    (examples/xor/main.rs:232-233,273-279) -- the frame bench.py times as `xor_compute_nearest_720p`.  Every pixel and every
    trip count against the oracle, through the record kernel (AUTO) and the literal twin (LINEAR), `single` and the 3 x 6
    `tile` loop with its wholly off-screen column (examples/xor/main.rs:77-95,235-254)."""
    W, H, n = 1280, 720, 256
    den, nrm = O.volume_xor(n, 0.0)
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H).get_proj_view_matrix()
    ref, rsteps, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm)
    s_ref = int(rsteps.astype(np.int64).sum())
    assert s_ref == 21175162 and rsteps.max() <= 293 and abs(int((rsteps > 0).sum()) - 180_000) < 5_000
    frames, sampled = {}, {}
    for name, lay, fl in (("records", V.LAYOUT_AUTO, 0), ("records_noskip", V.LAYOUT_AUTO, V.RENDER_NO_SKIP), ("literal", V.LAYOUT_LINEAR, 0)):
        img, steps, (sr, ss) = gpu_render(V, cam, den, W, H, vol2=nrm, mode=V.MODE_COMPUTE_NEAREST, layout=lay, flags=fl)
        assert (steps == rsteps).all() and sr == s_ref, name
        assert np.abs(img - ref).max() <= TOL, (name, np.abs(img - ref).max())
        frames[name], sampled[name] = img, ss
    assert (frames["records"].view(np.uint32) == frames["literal"].view(np.uint32)).all()
    # the record kernel against raycast_compute.wgsl:62-97 AS WRITTEN (VO_FLAG_LITERAL_WGSL: pow(a, 3.0) through powf, both smoothsteps
    # with their divide, nothing fused): every pixel of the example's own frame -- no trip count moves, <= 1e-5 per channel
    lit, lsteps, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm, flags=O.FLAG_LITERAL_WGSL)
    assert (lsteps == rsteps).all(), int((lsteps != rsteps).sum())
    assert np.abs(frames["records"] - lit).max() <= 1e-5, np.abs(frames["records"] - lit).max()
    # exact empty-space skipping of the record kernel (round 4): not a bit changes, and the steps that fetch and shade are those whose
    # record can contribute plus a rim of one or two voxels -- the blob fills half of the cube, a ray sees far less of it
    assert (frames["records"].view(np.uint32) == frames["records_noskip"].view(np.uint32)).all()
    assert sampled["records_noskip"] == s_ref and sampled["literal"] == s_ref
    a3 = den[..., 3].astype(np.float32) ** 3
    assert 0.3 < float((a3 > 0).mean()) < 0.6
    assert sampled["records"] < 0.75 * s_ref, (sampled["records"], s_ref)
    # three 64x64 tiles by name: centre, silhouette, hanging off the right edge
    hit = rsteps > 0
    ys, xs = np.nonzero(hit)
    sil_x = int(xs.min()) - 32
    for tx, ty in ((W // 2 - 32, H // 2 - 32), (sil_x, H // 2 - 32), (W - 32, H // 2 - 32)):
        t_img, t_steps, _ = gpu_render(V, cam, den, W, H, vol2=nrm, mode=V.MODE_COMPUTE_NEAREST, tile=(tx, ty, 64, 64))
        x1 = min(tx + 64, W)
        assert np.abs(t_img[ty:ty + 64, tx:x1] - ref[ty:ty + 64, tx:x1]).max() <= TOL
        assert (t_steps[ty:ty + 64, tx:x1] == rsteps[ty:ty + 64, tx:x1]).all()
    # the device generator + the reference's tile loop: TILE_SIZE 256, (H/256+1) x (W/256+1) = 3 x 6 offsets
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        V.VolumeTexture.generate_xor(ctx, (n,) * 3, 0.0)
        ctx.set_camera_blob(cam)
        pipe = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
        for y in range(H // 256 + 1):
            for x in range(W // 256 + 1):
                pipe.record(ctx, (x * 256, y * 256, 256, 256))
        assert (ctx.read_backbuffer().view(np.uint32) == frames["records"].view(np.uint32)).all()
    finally:
        ctx.close()


def test_procedural_full_size_tiles(V, O):
    """C3 at the size bench.py times it (1920x1080): one interior and one silhouette 64x64 tile and a 64-row strip through the
    middle against `render_procedural`; trip counts identical."""
    W, H = 1920, 1080
    cam = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H).get_proj_view_matrix()
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        ctx.set_camera_blob(cam)
        ctx.reset_step_counts()
        V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_COUNT).record(ctx)
        img, steps = ctx.read_backbuffer(), ctx.read_steps()
    finally:
        ctx.close()
    row = steps[H // 2]
    xs = np.nonzero(row)[0]
    assert xs.size > 300
    tiles = [(W // 2 - 32, H // 2 - 32, 64, 64), (int(xs.min()) - 32, H // 2 - 32, 64, 64), (0, H // 2 - 32, W, 64)]
    for (tx, ty, tw, th) in tiles:
        ref, rsteps = O.render_procedural(cam, W, H, tile=(tx, ty, tw, th))
        sl = (slice(ty, ty + th), slice(tx, tx + tw))
        assert (steps[sl] == rsteps[sl]).all(), (tx, ty)
        assert np.abs(img[sl] - ref[sl]).max() <= TOL, (tx, ty, np.abs(img[sl] - ref[sl]).max())
        # the literal reading (powf, the march's and xor.wgsl:59's smoothsteps with their divide): no trip count moves, <= 1e-5
        lit, lsteps = O.render_procedural(cam, W, H, tile=(tx, ty, tw, th), flags=O.FLAG_LITERAL_WGSL)
        assert (steps[sl] == lsteps[sl]).all(), (tx, ty, "literal trips")
        assert np.abs(img[sl] - lit[sl]).max() <= 1e-5, (tx, ty, np.abs(img[sl] - lit[sl]).max())
    assert rsteps[H // 2 - 32:H // 2 + 32].max() > 150


def test_rgba16f_full_size_c2(V, O):
    """The reference-shaped surface at the headline's size: C2 (1920x1080, dt_scale 0.5) written as rgba16f equals the
    oracle's f32 frame after the same round-to-nearest-even conversion, up to one f16 step where the f32 values themselves
    differ by the 1e-4 budget (an f16 ulp at 0.5..1 is 4.9e-4)."""
    W, H = 1920, 1080
    vol = O.volume_standin_u8(256)
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
    ref, rsteps, _ = O.render(cam, vol, W, H, dt_scale=0.5)
    want = O.rgba32f_to_rgba16f(ref)
    img16, _, _ = gpu_render(V, cam, vol, W, H, dt=0.5, out=V.OUT_RGBA16F, want_steps=False)
    img32, _, _ = gpu_render(V, cam, vol, W, H, dt=0.5, out=V.OUT_RGBA32F, want_steps=False)
    got = img16.view(np.uint16)
    # the kernel's f16 store is RNE of its own f32 value, bit for bit
    assert (got == O.rgba32f_to_rgba16f(img32)).all()
    dbits = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert dbits.max() <= 1 and (dbits == 0).mean() > 0.999, (dbits.max(), (dbits == 0).mean())
    assert np.abs(img16.astype(np.float32) - want.view(np.float16).astype(np.float32)).max() <= 4.9e-4
    assert (got[..., 3] == 0x3C00).all()  # alpha 1.0


def test_hip_against_literal_wgsl(V, O):
    """The kernels against the shader's text AS WRITTEN (VO_FLAG_LITERAL_WGSL: two-rounding texel coordinate, per-tap /255,
    unfused lerps, true divide in smoothstep, libm cos/pow) rather than against the specified reading the oracle shares with
    them: C1 whole and a 640x480 crop of C2.  No pixel changes its trip count; <= 1e-5 per channel."""
    vol = O.volume_standin_u8(256)
    for (W, H, dt, tile) in ((512, 512, 1.0, None), (1920, 1080, 0.5, (640, 300, 640, 480))):
        cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
        lit, lsteps, _ = O.render(cam, vol, W, H, dt_scale=dt, tile=tile, flags=O.FLAG_LITERAL_WGSL)
        img, steps, _ = gpu_render(V, cam, vol, W, H, dt=dt)
        sl = (slice(None), slice(None)) if tile is None else (slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2]))
        agree = steps[sl] == lsteps[sl]
        assert agree.all(), int((~agree).sum())
        d = np.abs(img[sl] - lit[sl])
        assert d.max() <= 1e-5, d.max()
        assert (lsteps[sl] > 0).mean() > 0.2


def test_fast_walk_tolerance_mode(V, O, golden, cameras, golden_volumes):
    """VK_RENDER_FAST_WALK against the ORACLE: skips advance t and p in closed form, so frames are no longer bit-identical to
    the default mode -- they stay inside the contract's 1e-4 except where a ray's last `t < t1` or its alpha >= 0.95 early-out
    lands on the other side (profiles/r04_walk_modes.txt).  t is kept exact, so a ray's iteration count is the reference's unless
    its early-out flips.  Bars: an iteration count moves by at most one, on < 0.05 % of the rays (small cubes: < 1 %); < 0.2 % of the
    pixels are further than 1e-4 from the oracle; the mean difference stays below 1e-5; S_sampled (integer work) within 0.2 %.
    The mode changes nothing for kernels that do not skip."""
    fast = V.RENDER_FAST_WALK

    def check(img, steps, ref, rsteps, what, px_bar=2e-3, steps_bar=5e-4):
        d = np.abs(img - ref)
        ds = np.abs(steps.astype(np.int64) - rsteps.astype(np.int64))
        hit = max(int((rsteps > 0).sum()), 1)
        rep = (what, float(d.max()), float(d.mean()), float((d.max(axis=-1) > TOL).mean()), int((ds > 0).sum()), hit)
        assert ds.max() <= 1, rep
        assert (ds > 0).sum() <= max(steps_bar * hit, 2), rep
        assert (d.max(axis=-1) > TOL).mean() <= px_bar and d.mean() <= 1e-5 and d.max() <= 0.1, rep
        assert (img[..., 3] == 1).all()
        return rep

    vol = O.volume_standin_u8(256)
    for (W, H, dt, aspect) in ((512, 512, 1.0, 1.0), (1920, 1080, 0.5, 16 / 9)):
        cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), aspect).get_proj_view_matrix()
        ref, rsteps, rsamp = O.render(cam, vol, W, H, dt_scale=dt)
        for lay in (V.LAYOUT_PACKED_PAIRS, V.LAYOUT_PACKED):
            img, steps, (s_ref, s_samp) = gpu_render(V, cam, vol, W, H, dt=dt, layout=lay, flags=fast)
            check(img, steps, ref, rsteps, ("standin", W, H, lay))
            assert abs(s_ref - int(rsteps.sum())) <= 1e-6 * s_ref and abs(s_samp - int(rsamp.sum())) <= 2e-3 * s_samp
        # a kernel that does not skip ignores the flag: bitwise the exact frame
        a, _, _ = gpu_render(V, cam, vol, W, H, dt=dt, flags=V.RENDER_NO_SKIP | fast, want_steps=False)
        b, _, _ = gpu_render(V, cam, vol, W, H, dt=dt, flags=V.RENDER_NO_SKIP, want_steps=False)
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    # the golden vectors (32^3 volumes, 64x64: few rays, coarse cells -- looser bars) incl. the f16 volume
    g = golden["naive_64x64"]
    for key in sorted({k.rsplit("__", 1)[0] for k in g.files}):
        vname, cname, dts = key.split("__")
        for lay in (V.LAYOUT_PACKED, V.LAYOUT_PACKED_PAIRS):
            img, steps, _ = gpu_render(V, cameras[cname], golden_volumes[vname], 64, 64, dt=float(dts[2:]), layout=lay, flags=fast)
            check(img, steps, g[key + "__rgba"], g[key + "__steps"], (key, lay), px_bar=2e-2, steps_bar=0.01)
    gf = golden["naive_f16_64x64"]
    img, steps, _ = gpu_render(V, cameras["bonsai_1x1"], O.volume_fog_f16(32), 64, 64, dt=0.5, layout=V.LAYOUT_PACKED, flags=fast)
    check(img, steps, gf["rgba"], gf["steps"], "f16 fog", px_bar=2e-2, steps_bar=0.01)
    # seeded cameras / dims / step sizes (the cases of test_skip_fuzz_cameras_dims_dt)
    rng = np.random.default_rng(20240611)
    for trial in range(6):
        dims = tuple(int(v) for v in rng.integers(9, 70, 3))
        v = (rng.random(dims[::-1]) < 0.15).astype(np.uint8) * rng.integers(26, 255, dims[::-1], dtype=np.uint8)
        W, H = int(rng.integers(40, 200)), int(rng.integers(40, 120))
        cam = O.camera_blob(float(rng.uniform(0.3, 2.0)), float(rng.uniform(-1.2, 1.2)), float(rng.uniform(0, 6.28)), (0.5, 0.5, 0.5), W / H)
        dt = float(rng.choice([0.3, 0.5, 1.0]))
        ref, rsteps, _ = O.render(cam, v, W, H, dt_scale=dt)
        img, steps, _ = gpu_render(V, cam, v, W, H, dt=dt, layout=V.LAYOUT_PACKED, flags=fast)
        check(img, steps, ref, rsteps, (trial, dims, dt), px_bar=3e-2, steps_bar=0.01)
