"""GPU suite: the host surface above the C-ABI -- error codes, Demo / run_headless, the compiled C++ host, present pass and capture_frame's byte layout, the raw
loader -- and the two reference-held pins (volume.png, the colours of bonsai.png) through the HIP path."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import TOL, V, _captured_rgb, _holes_volume, _orbit_cameras, _render_with_params, _synced, gpu_render, layouts  # noqa: F401

pytestmark = pytest.mark.gpu


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
