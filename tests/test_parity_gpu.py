"""GPU suite: the naive march (raycast_naive.wgsl) -- cell layouts, f16 volumes, the staged 8^3 bricks, BASELINE configs at full size, skipping,
the tolerance walk -- called through the C-ABI, against the CPU oracle and the golden vectors.

Bars: per-channel |dRGBA| <= 1e-4 on the f32 surface (north star); loop trip counts (S_ref) and
the count of tap-fetching steps (S_sampled) are integer work and must be *identical*.
"""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import TOL, V, _captured_rgb, _holes_volume, _orbit_cameras, _render_with_params, _synced, gpu_render, layouts  # noqa: F401

pytestmark = pytest.mark.gpu


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
