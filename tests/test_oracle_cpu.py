"""CPU suite, part 1: the C oracle against the committed golden vectors and the independent numpy
restatement.  (The reference has no tests and cannot run here: parity is unpinned -- these pin the
two restatements to each other and to tests/golden/.)"""
import numpy as np
import pytest


def _cases(npz):
    keys = sorted({k.rsplit("__", 1)[0] for k in npz.files})
    return keys


def test_oracle_matches_golden_naive(O, golden, cameras, golden_volumes):
    g = golden["naive_64x64"]
    assert len(_cases(g)) >= 20
    for key in _cases(g):
        vname, cname, dts = key.split("__")
        dt = float(dts[2:])
        rgba, steps, samp = O.render(cameras[cname], golden_volumes[vname], 64, 64, dt_scale=dt)
        assert np.abs(rgba - g[key + "__rgba"]).max() <= 1e-6, key
        assert (steps == g[key + "__steps"]).all(), key
        assert (samp == g[key + "__sampled"]).all(), key


def test_oracle_matches_golden_f16_volume(O, golden, cameras):
    g = golden["naive_f16_64x64"]
    rgba, steps, _ = O.render(cameras["bonsai_1x1"], O.volume_fog_f16(32), 64, 64, dt_scale=0.5)
    assert np.abs(rgba - g["rgba"]).max() <= 1e-6
    assert (steps == g["steps"]).all()
    assert steps.max() <= 2 * 32 + 1  # dt_scale 0.5 on a 32^3 volume: <= 2n+1 iterations


def test_oracle_matches_golden_compute(O, golden, cameras):
    g = golden["compute_128x72"]
    den, nrm = g["density"].view(np.float16), g["normals"].view(np.float16)
    rgba, steps, _ = O.render(cameras["xor_16x9"], den, 128, 72, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm)
    assert np.abs(rgba - g["rgba"]).max() <= 1e-6
    assert (steps == g["steps"]).all()
    # `tile` entry point with a dynamic offset; the part beyond the image is dropped (A12)
    t_rgba, t_steps, _ = O.render(cameras["xor_16x9"], den, 128, 72, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm,
                                  tile=(96, 48, 64, 64))
    assert np.abs(t_rgba - g["tile_rgba"]).max() <= 1e-6
    assert (t_steps == g["tile_steps"]).all()
    assert (t_rgba[:48] == 0).all() and (t_rgba[:, :96] == 0).all()  # untouched outside the tile
    miss = steps == 0
    assert np.allclose(rgba[miss][:, :3], [0.023, 0.02, 0.02]) and (rgba[..., 3] == 1).all()


def test_oracle_matches_golden_procedural(O, R, golden, cameras):
    """C3 (no volume): the compute march over xor.wgsl's noise_volume -- C oracle vs the numpy-made fixture, and
    the two restatements on a fresh camera / time (their sines are evaluated independently)."""
    g = golden["procedural_96x54"]
    rgba, steps = O.render_procedural(cameras["xor_16x9"], 96, 54)
    assert np.abs(rgba - g["rgba"]).max() <= 1e-6 and (steps == g["steps"]).all()
    rgba, steps = O.render_procedural(cameras["xor_16x9"], 96, 54, dt_scale=2.5)
    assert np.abs(rgba - g["rgba_dt2p5"]).max() <= 1e-6 and (steps == g["steps_dt2p5"]).all()
    assert 0.1 < (steps > 0).mean() < 0.3 and steps.max() <= 347     # <= 2*sqrt(3)/0.01 steps through [-1,1]^3
    miss = steps == 0
    assert np.allclose(rgba[miss][:, :3], [0.023, 0.02, 0.02]) and (rgba[..., 3] == 1).all()
    cam = O.camera_blob(2.2, 0.3, 2.0, (0.1, -0.1, 0.0), 4 / 3)
    a, sa = O.render_procedural(cam, 40, 30, time=1.25)
    b, sb = R.render_procedural(cam, 40, 30, time=1.25)
    assert np.abs(a - b).max() <= 1e-5 and (sa != sb).mean() <= 0.01


def test_two_restatements_agree_on_fresh_inputs(O, R):
    """Inputs that are not in the fixtures: other sizes, non-cubic dims, other cameras / dt."""
    for dims, cam_args, W, H, dt in [((40, 64, 24), (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.5), 60, 40, 1.0),
                                     (48, (0.8, -0.3, 4.0, (0.4, 0.5, 0.6), 1.0), 48, 48, 0.7),
                                     (24, (2.0, 1.2, 0.3, (0.5, 0.5, 0.5), 0.75), 36, 48, 2.0)]:
        vol = O.volume_standin_u8(dims, seed=77)
        assert (vol == R.volume_standin_u8(dims, seed=77)).all()
        cam = O.camera_blob(*cam_args)
        a, sa, ma = O.render(cam, vol, W, H, dt_scale=dt)
        b, sb, mb = R.render_naive(cam, vol, W, H, dt_scale=dt)
        assert np.abs(a - b).max() <= 1e-6
        assert (sa == sb).all() and (ma == mb).all()
        assert sa.sum() > 0


def test_units_intersect_box(O, golden):
    import ctypes as C
    u = golden["units"]
    for o, d, t0, t1 in zip(u["box_o"], u["box_d"], u["box_t0"], u["box_t1"]):
        out = (C.c_float * 2)()
        O.lib().vo_intersect_box((C.c_float * 3)(*o), (C.c_float * 3)(*d), 0.0, 1.0, out)
        assert np.float32(out[0]) == t0 or (np.isnan(out[0]) and np.isnan(t0))
        assert np.float32(out[1]) == t1 or (np.isnan(out[1]) and np.isnan(t1))
    # axis-parallel ray through the box: +-inf slabs on the degenerate axes, finite on z
    out = (C.c_float * 2)()
    O.lib().vo_intersect_box((C.c_float * 3)(0.5, 0.5, -1.0), (C.c_float * 3)(0.0, 0.0, 1.0), 0.0, 1.0, out)
    assert (out[0], out[1]) == (1.0, 2.0)
    # eye inside the box: t0 < 0 before the max(t0, 0)
    O.lib().vo_intersect_box((C.c_float * 3)(0.5, 0.5, 0.5), (C.c_float * 3)(0.0, -1.0, 0.0), 0.0, 1.0, out)
    assert out[0] == -0.5 and out[1] == 0.5
    # box behind the ray: the slab test still reports t0 < t1 (both negative); after max(t0, 0)
    # the march loop runs zero times, so the pixel is black like a miss (raycast_naive.wgsl:91-101)
    O.lib().vo_intersect_box((C.c_float * 3)(2.0, 2.0, 2.0), (C.c_float * 3)(0.57735026, 0.57735026, 0.57735026), 0.0, 1.0, out)
    assert out[0] < out[1] < 0
    # true miss: slabs do not overlap
    O.lib().vo_intersect_box((C.c_float * 3)(2.0, 2.0, 0.5), (C.c_float * 3)(0.0, -1.0, 0.0), 0.0, 1.0, out)
    assert out[0] > out[1]


def test_units_trilinear_clamp_to_edge(O, golden, golden_volumes):
    import ctypes as C
    u = golden["units"]
    ramp, stand = golden_volumes["ramp_x"], golden_volumes["standin"]
    for p, e_ramp, e_stand in zip(u["tri_pts"], u["tri_ramp_x"], u["tri_standin"]):
        pp = (C.c_float * 3)(*p)
        got = O.lib().vo_sample_trilinear(ramp.ctypes.data, 32, 32, 32, O.FMT_R8_UNORM, pp, 0, None)
        assert abs(got - e_ramp) <= 1e-7
        got = O.lib().vo_sample_trilinear(stand.ctypes.data, 32, 32, 32, O.FMT_R8_UNORM, pp, 0, None)
        assert abs(got - e_stand) <= 1e-7
    # texel centre returns the texel; outside the first/last centre the edge texel is held
    v = ramp
    centre = (C.c_float * 3)((5 + 0.5) / 32, 0.5, 0.5)
    assert O.lib().vo_sample_trilinear(v.ctypes.data, 32, 32, 32, 0, centre, 0, None) == pytest.approx(v[0, 0, 5] / 255, abs=1e-7)
    for x, ref in ((0.0, v[0, 0, 0]), (0.2 / 32, v[0, 0, 0]), (1.0, v[0, 0, 31]), (31.9 / 32, v[0, 0, 31])):
        got = O.lib().vo_sample_trilinear(v.ctypes.data, 32, 32, 32, 0, (C.c_float * 3)(x, 0.5, 0.5), 0, None)
        assert got == pytest.approx(ref / 255, abs=1e-7)


def test_units_srgb_and_transfer(O, golden):
    u = golden["units"]
    for x, y in zip(u["srgb_x"], u["srgb_y"]):
        assert abs(O.lib().vo_linear_to_srgb(float(x)) - y) <= 2e-7
    assert O.lib().vo_linear_to_srgb(0.0031308) == pytest.approx(12.92 * 0.0031308, rel=1e-6)  # the knee
    for r, a in zip(u["alpha_r"], u["alpha_a"]):
        assert O.lib().vo_transfer_alpha(float(r), 0) == a
    for x, a in zip(u["alpha_raw"], u["alpha_a_raw"]):  # filtered R8Unorm taps on their 0..255 scale
        assert O.lib().vo_transfer_alpha(float(x), 1) == a
    # u8 <= 25 is exactly transparent, 26 is not; opacity saturates at min(0.9, r)
    assert O.lib().vo_transfer_alpha(25.0, 1) == 0.0 and O.lib().vo_transfer_alpha(25.49, 1) == 0.0
    assert O.lib().vo_transfer_alpha(26.0, 1) > 0.0
    assert O.lib().vo_transfer_alpha(0.0999755859375, 0) == 0.0 and O.lib().vo_transfer_alpha(0.1002197265625, 0) > 0.0  # the f16 neighbours of 0.1
    amax = O.lib().vo_transfer_alpha(1.0, 0)
    assert amax == O.lib().vo_transfer_alpha(0.9, 0) and amax == pytest.approx(0.8174, abs=2e-4)
    assert O.lib().vo_transfer_alpha(255.0, 1) == pytest.approx(amax, abs=1e-6) and O.lib().vo_transfer_alpha(255.0, 1) == O.lib().vo_transfer_alpha(229.5, 1)
    # the two scales are the same function: |alpha(x/255) - alpha_raw(x)| stays at rounding level
    xs = np.linspace(0, 255, 511, dtype=np.float32)
    d = [abs(O.lib().vo_transfer_alpha(float(x), 1) - O.lib().vo_transfer_alpha(float(np.float32(x) / np.float32(255.0)), 0)) for x in xs]
    assert max(d) <= 2e-6


def test_tap_normalisation_order_is_immaterial(O, cameras, golden_volumes):
    """The specification normalises R8Unorm once after filtering; normalising each tap first (the
    literal reading of the sampler) changes the image by far less than the 1e-4 budget."""
    for name in ("standin", "ramp_y"):
        a, sa, _ = O.render(cameras["bonsai_1x1"], golden_volumes[name], 64, 64)
        b, sb, _ = O.render(cameras["bonsai_1x1"], golden_volumes[name], 64, 64, flags=O.FLAG_TAPNORM_PER_TAP)
        same = sa == sb
        assert same.mean() > 0.995  # an early-out can flip by one step on a handful of rays
        assert np.abs(a - b)[same].max() <= 2e-6


def test_f16_conversions(O):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 10.0 ** rng.integers(-8, 5, 5000),
                        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 2.98e-8, 2.99e-8, np.inf, -np.inf, 0.7], np.float32)])
    want = x.astype(np.float16).view(np.uint16)
    got = np.array([O.lib().vo_f32_to_f16(float(v)) for v in x], np.uint16)
    assert (got == want).all()
    h = np.arange(0, 65536, 7, dtype=np.uint16)
    back = np.array([O.lib().vo_f16_to_f32(int(v)) for v in h], np.float32)
    ref = h.view(np.float16).astype(np.float32)
    assert ((back == ref) | (np.isnan(back) & np.isnan(ref))).all()


def test_nominal_step_counts_match_survey(O, cameras):
    """512x512 dt 1.0 -> 54.3 % hit, 2.16e7 nominal steps, <= 257 per ray."""
    vol = np.zeros((256, 256, 256), np.uint8)
    _, steps, _ = O.render(cameras["bonsai_1x1"], vol, 512, 512, dt_scale=1.0, flags=O.FLAG_NO_EARLY_OUT)
    assert abs((steps > 0).mean() - 0.543) < 0.002
    assert abs(steps.sum() / 2.16e7 - 1) < 0.01
    assert steps.max() == 257


def test_present_oracle_basics(O):
    """present.wgsl: ACESFilm + branch-free sRGB on a few known values, identity-size resample."""
    bb = np.zeros((4, 6, 4), np.float32)
    bb[..., 3] = 1.0
    bb[1, 2, :3] = [0.5, 0.0031308, 10.0]
    bb[2, 3, :3] = [0.001, 0.18, 1.0]
    out = O.present(bb, 6, 4)
    assert out.shape == (4, 6, 4) and (out[..., 3] == 255).all() and (out[0, 0, :3] == 0).all()

    def ref(x):
        a = min(max((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0), 1.0)
        s = 12.92 * a if a <= 0.0031308 else 1.055 * a ** 0.41666 - 0.055
        return int(np.floor(min(max(s, 0.0), 1.0) * 255 + 0.5))

    for (y, x) in ((1, 2), (2, 3)):
        assert [abs(int(out[y, x, c]) - ref(float(bb[y, x, c]))) <= 1 for c in range(3)] == [True] * 3
    # downsampling by 2 averages 2x2 texel blocks (uv lands on texel corners)
    small = O.present(np.full((8, 8, 4), 0.18, np.float32), 4, 4)
    assert (small[..., :3] == ref(0.18)).all()


def test_xor_generator_oracle(O):
    """shaders/xor.wgsl restated: the hash's sine is the correctly rounded f32 sine; the volume has the
    reference's structure (alpha only inside radius 0.5, NaN normals where the gradient vanishes)."""
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-1e5, 1e5, 4000), np.arange(0.0, 400.0)]).astype(np.float32)
    got = np.array([O.lib().vo_sin_spec(float(v)) for v in x], np.float32)
    assert (got == np.sin(x.astype(np.float64)).astype(np.float32)).all()
    den, nrm = O.volume_xor(24, 0.0)
    a = den[..., 3].astype(np.float32)
    z, y, xx = np.meshgrid(*[(np.arange(24) - 12) / 24.0] * 3, indexing="ij")
    r = np.sqrt(xx * xx + y * y + z * z)
    assert (a[r > 0.5] == 0).all() and (a[r < 0.2] > 0).all()
    assert (den[..., 0] == den[..., 1]).all() and (den[..., 0] == den[..., 2]).all()
    n3 = nrm[..., :3].astype(np.float32)
    assert np.isnan(n3[r > 0.51]).all()                     # normalize(0): the reference stores NaN there too
    ok = ~np.isnan(n3).any(axis=-1)
    assert np.abs(np.linalg.norm(n3[ok], axis=-1) - 1).max() < 2e-3


def test_golden_volumes_rebuild(tmp_path, golden_volumes):
    """The committed fixtures are what oracle/gen_golden.py produces TODAY: regenerate into a scratch directory and compare
    every array bit for bit (a fixture edited by hand, or a specification changed without regenerating, fails here); and the
    closed-form volumes the tests rebuild (conftest.adversarial_volumes) are the generator's own."""
    import os

    from oracle import gen_golden as G

    from conftest import GOLDEN, adversarial_volumes

    G.main(str(tmp_path))
    # (volume_png_pin.npz / bonsai_png_colours.npz come from oracle/volume_png.py / bonsai_png.py and are re-derived by their own tests)
    names = sorted(n for n in os.listdir(GOLDEN) if n.endswith(".npz") and not n.startswith(("volume_png", "bonsai_png")))
    assert names == sorted(n for n in os.listdir(tmp_path) if n.endswith(".npz"))
    for n in names:
        a, b = np.load(os.path.join(GOLDEN, n)), np.load(os.path.join(tmp_path, n))
        assert sorted(a.files) == sorted(b.files), n
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), (n, k)
    mine, theirs = adversarial_volumes(32), G.adversarial_volumes(32)
    assert sorted(mine) == sorted(theirs)
    for k in mine:
        assert (mine[k] == theirs[k]).all(), k


def test_transfer_respecification_is_bounded(O, cameras, golden_volumes):
    """Round 2 re-specified the transfer function (one fused op carrying the sample's scale; goldens regenerated).  The
    round-1 text is kept as vo_transfer_alpha_r1: over EVERY u8-scale input on a fine grid and every finite non-negative
    f16 the two differ by at most 2 ulp of alpha's range, and on the golden cameras the two formulations give identical
    per-pixel trip counts and images within 2e-6 -- the re-specification moved no exit and no pixel beyond rounding."""
    L = O.lib()
    xs = np.concatenate([np.arange(0, 256, dtype=np.float32), np.linspace(0, 255, 100001, dtype=np.float32)])
    assert max(abs(L.vo_transfer_alpha(float(x), 1) - L.vo_transfer_alpha_r1(float(x), 1)) for x in xs) <= 2.5e-7
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    assert max(abs(L.vo_transfer_alpha(float(x), 0) - L.vo_transfer_alpha_r1(float(x), 0)) for x in h) <= 2.5e-7
    # exact zeros: what the skip maps call transparent (every tap <= 25) is exactly transparent under both texts; the one input
    # on which they part is the threshold itself, x = 25.5 = 0.1 * 255, where f32(25.5 * f32(1/255)) lands one ulp above 0.1f:
    # the round-1 text answers 1.4e-16 there, the fused one 0 -- fourteen orders of magnitude below anything a pixel shows
    for x in (0.0, 25.0, 25.49):
        assert L.vo_transfer_alpha(x, 1) == 0.0 and L.vo_transfer_alpha_r1(x, 1) == 0.0
    assert L.vo_transfer_alpha(25.5, 1) == 0.0 and 0.0 <= L.vo_transfer_alpha_r1(25.5, 1) < 1e-15
    assert L.vo_transfer_alpha(26.0, 1) > 0.0 and L.vo_transfer_alpha_r1(26.0, 1) > 0.0
    for cam in ("bonsai_1x1", "inside", "axis"):
        for name in ("standin", "ramp_x", "checker", "fog"):
            for dt in (1.0, 0.5):
                a, sa, _ = O.render(cameras[cam], golden_volumes[name], 64, 64, dt_scale=dt)
                b, sb, _ = O.render(cameras[cam], golden_volumes[name], 64, 64, dt_scale=dt, flags=O.FLAG_TRANSFER_R1)
                assert (sa == sb).all(), (cam, name, dt)
                assert np.abs(a - b).max() <= 2e-6, (cam, name, dt)


def test_literal_wgsl_yardstick(O):
    """VO_FLAG_LITERAL_WGSL evaluates raycast_naive.wgsl:96-119 as written (unfused p*n - 0.5, every tap / 255, unfused lerps,
    smoothstep with its divide, the background term, libm cos / pow); the specified reading (fused coordinate, one scale,
    one-fma transfer) is what oracle, numpy restatement and HIP kernels share.  Distance between the two on C1 (512x512,
    dt_scale 1) and on a 640x480 crop of C2 (1920x1080, dt_scale 0.5), bonsai stand-in 256^3: NO pixel changes its trip
    count, max |dRGBA| 1.4e-6 (bar: 1e-4) -- the numbers DESIGN.md 2.1 quotes."""
    vol = O.volume_standin_u8(256)
    for W, H, dt, aspect, tile in ((512, 512, 1.0, 1.0, None), (1920, 1080, 0.5, 16 / 9, (640, 300, 640, 480))):
        blob = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), aspect)
        a, sa, _ = O.render(blob, vol, W, H, dt_scale=dt, tile=tile)
        b, sb, _ = O.render(blob, vol, W, H, dt_scale=dt, tile=tile, flags=O.FLAG_LITERAL_WGSL)
        differ = np.argwhere(sa != sb)
        assert len(differ) <= 16, differ[:20]  # (measured: none)
        same = sa == sb
        assert np.abs(a - b)[same].max() <= 1e-5 and np.abs(a - b).max() <= 1e-4
        assert (sa > 0).sum() > 100000
    # the transfer function alone, every input: |alpha_specified - alpha_literal| stays at rounding level
    L = O.lib()
    xs = np.linspace(0, 255, 50001, dtype=np.float32)
    assert max(abs(L.vo_transfer_alpha(float(x), 1) - L.vo_transfer_alpha_literal(float(np.float32(x) / np.float32(255.0)))) for x in xs) <= 2.5e-7


def test_literal_wgsl_yardstick_compute_and_procedural(O):
    """VO_FLAG_LITERAL_WGSL for the other two kernel families: raycast_compute.wgsl:62-97 as written -- pow(a, 3.0) through powf, both
    smoothsteps with their divide and no fused operation -- and, for the procedural mode, xor.wgsl:59's falloff the same way.  Against the
    specified reading (a*a*a, reciprocal-and-fma smoothstep) on the xor example at 640x360 (128^3 pair) and C3 at 320x180: NO pixel changes
    its trip count, max |dRGBA| stays at rounding level (measured 6e-8 / 1.2e-7; bar 1e-5)."""
    n, W, H = 128, 640, 360
    den, nrm = O.volume_xor(n, 0.0)
    cam = O.camera_blob(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
    a, sa, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm)
    b, sb, _ = O.render(cam, den, W, H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm, flags=O.FLAG_LITERAL_WGSL)
    assert (sa == sb).all() and (sa > 0).mean() > 0.15
    assert 0 < np.abs(a - b).max() <= 1e-5  # (the two readings are different arithmetic: not bitwise equal)
    pa, psa = O.render_procedural(cam, 320, 180)
    pb, psb = O.render_procedural(cam, 320, 180, flags=O.FLAG_LITERAL_WGSL)
    assert (psa == psb).all() and (psa > 0).mean() > 0.15
    assert 0 < np.abs(pa - pb).max() <= 1e-5


def test_volume_png_pin(O):
    """The one reference-held pin (oracle/volume_png.py): xor.wgsl generator at t = 0 -> raycast_compute.wgsl at 1280x720 ->
    present.wgsl to the capture's 958x1050, at the camera fitted to the reference's `volume.png`.  The capture itself travels
    as a blurred 1/8-scale fixture; where /root/reference exists the fixture is re-derived from the file."""
    import os

    from oracle import volume_png as VP

    pin = VP.load_pin()
    cap = pin["capture_blur_ds"].astype(np.float32)
    if os.path.exists(VP.REF_PNG):
        from PIL import Image

        png = np.array(Image.open(VP.REF_PNG).convert("RGB"))
        assert png.shape == (VP.WIN_H, VP.WIN_W, 3)
        assert (png[0, 0] == VP.BACKGROUND).all() and (png[-1, -1] == VP.BACKGROUND).all()
        assert np.abs(VP.blurred_ds(png) - cap).max() <= 0.13  # f16 storage of values <= 255
    zoom, pitch, yaw = (float(v) for v in pin["orbit"])
    assert VP.camera(zoom, pitch, yaw) == pin["camera"].tobytes()
    frame = VP.render_oracle(zoom, pitch, yaw)
    # background: clear colour (0.023, 0.02, 0.02) -> ACES -> branch-free sRGB -> 8 bit, exactly the capture's
    assert (frame[0, 0, :3] == VP.BACKGROUND).all() and (frame[-1, -1, :3] == VP.BACKGROUND).all() and (frame[..., 3] == 255).all()
    m = VP.metrics(frame, cap)
    VP.check(m)
    assert abs(m["corr"] - float(pin["corr"])) < 5e-3 and abs(m["mean_abs"] - float(pin["mean_abs"])) < 0.05
    # the committed frame is this frame (libm's powf may move a pixel by one step between machines)
    want = VP.load_oracle_frame()
    d = np.abs(frame[..., :3].astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    # the example's own initial camera (examples/xor/main.rs:273-279) shows the same blob 1.5x smaller: the fit moved zoom, not shape
    f0 = VP.render_oracle(3.0, -0.5, 1.0)
    b0 = VP._bbox(VP.blurred_ds(f0))
    b1 = VP._bbox(VP.blurred_ds(frame))
    ratio = (b1[3] - b1[2]) / (b0[3] - b0[2])
    assert 1.35 < ratio < 1.75, ratio


def test_bonsai_png_palette_pin(O):
    """The second reference-held pin, for the naive path (oracle/bonsai_png.py): every colour of the reference's `bonsai.png` is a convex
    combination of black and the oracle's palette curve -- uniform volumes v = 0 .. 255 under raycast_naive.wgsl:101-123 -- once the capture's
    present pass is read as one more linear_to_srgb; its greenest colour (the inside of the pot) IS a point of that curve."""
    import os

    from oracle import bonsai_png as BP

    pin = BP.load_pin()
    sample, ext = pin["sample"], pin["extremes"]
    assert sample.shape == (BP.SAMPLE, 3) and ext.shape == (27, 3)
    if os.path.exists(BP.REF_PNG):
        from PIL import Image

        png = np.array(Image.open(BP.REF_PNG).convert("RGB"))
        s2, e2, n_nz = BP.colours_of(png)
        assert (s2 == sample).all() and (e2 == ext).all() and n_nz == int(pin["non_black"])
        assert (png[0, 0] == 0).all() and (png[0, 0] == pin["background"]).all()  # LoadOp::Clear(BLACK), examples/bonsai/main.rs:41
    assert tuple(int(v) for v in ext[26]) == BP.GREENEST
    curve = BP.oracle_curve()
    assert curve.shape == (256, 3) and (curve[:20] == 0).all()  # values up to 0.1 * 255 are exactly transparent
    hull = BP.hull_of(curve)
    both = np.vstack([sample, ext])
    assert BP.inside_share(both, hull, BP.decode_double_srgb) >= BP.BAR_INSIDE
    assert BP.inside_share(both, hull, BP.decode_aces_srgb) <= BP.BAR_INSIDE_ACES  # (0.77 on the whole capture: HEAD's tone map is not what took it)
    v, d = BP.nearest_on_curve(BP.GREENEST, curve)
    assert d <= BP.BAR_GREENEST and 170 <= v <= 190, (v, d)
    # nothing a backbuffer value <= 1 can produce through ACES + sRGB exceeds 232; the capture goes up to 249
    assert int(np.round(255 * BP.present_srgb(np.float32(2.51 + 0.03) / np.float32(2.43 + 0.59 + 0.14)))) <= 232 < int(both.max())
