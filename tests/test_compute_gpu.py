"""GPU suite: the compute twin (raycast_compute.wgsl `single` / `tile`), the procedural mode (C3) and the xor generator (xor.wgsl) against the oracle:
|dRGBA| <= 1e-4, trip counts identical, the generator bit for bit."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import TOL, V, _captured_rgb, _holes_volume, _orbit_cameras, _render_with_params, _synced, gpu_render, layouts  # noqa: F401

pytestmark = pytest.mark.gpu


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
