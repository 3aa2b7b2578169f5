"""Generates tests/golden/*.npz from the numpy restatement (oracle/np_restatement.py).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.  The reference cannot run here (no Rust / Vulkan),
so these are not reference outputs: they are known-answer vectors of the *second, independent*
restatement, cross-checked against the C oracle at generation time (agreement <= 1e-6 in RGBA,
identical step counts), and committed so that the C oracle and the HIP path are both pinned to
the same numbers on the GPU box, where neither /root/reference nor this script's numpy path is
needed.  Run:  python -m oracle.gen_golden
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import np_restatement as R  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def adversarial_volumes(n=32):
    """Named u8 volumes [n,n,n] built from closed-form rules (rebuilt identically by the tests)."""
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    vols = {
        "all0": np.zeros((n, n, n), np.uint8),
        "all25": np.full((n, n, n), 25, np.uint8),   # largest exactly transparent value
        "all26": np.full((n, n, n), 26, np.uint8),   # smallest contributing value
        "all255": np.full((n, n, n), 255, np.uint8),
        "impulse": np.zeros((n, n, n), np.uint8),
        "ramp_x": ((x * 255) // (n - 1)).astype(np.uint8),
        "ramp_y": ((y * 255) // (n - 1)).astype(np.uint8),
        "ramp_z": ((z * 255) // (n - 1)).astype(np.uint8),
        "checker": (((x + y + z) & 1) * 255).astype(np.uint8),
    }
    vols["impulse"][n // 2, n // 3, n // 4] = 255
    return vols


CAMERAS = {
    # name: (zoom, pitch, yaw, target, aspect)
    "bonsai_1x1": (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0),            # examples/bonsai/main.rs:68-74
    "bonsai_16x9": (1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 16.0 / 9.0),
    "inside": (0.3, 0.2, 2.5, (0.5, 0.5, 0.5), 1.0),                  # eye inside the unit cube
    "axis": (1.5, 0.0, 0.0, (0.5, 0.5, 0.5), 1.0),                    # looks straight down -z: dir components 0
    "xor_16x9": (3.0, -0.5, 1.0, (0.0, 0.0, 0.0), 16.0 / 9.0),        # examples/xor/main.rs:273-279
}


def main(out_dir: str | None = None):
    """Writes the fixtures into `out_dir` (default: tests/golden).  tests/test_oracle_cpu.py::test_golden_volumes_rebuild runs
    it into a scratch directory and holds the committed fixtures to what it produces today."""
    global OUT
    if out_dir is not None:
        OUT = out_dir
    os.makedirs(OUT, exist_ok=True)
    cams = {k: O.camera_blob(*v) for k, v in CAMERAS.items()}
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **{k: np.frombuffer(v, np.uint8) for k, v in cams.items()})

    # ---- naive (bonsai) path -------------------------------------------------------------
    cases = {}
    standin32 = R.volume_standin_u8(32)
    assert (standin32 == O.volume_standin_u8(32)).all()
    vols = dict(adversarial_volumes(32), standin=standin32, fog=R.volume_fog_u8(32))
    W = H = 64
    for vname, vol in vols.items():
        for cname in (("bonsai_1x1", "inside", "axis") if vname in ("standin", "ramp_x", "checker") else ("bonsai_1x1",)):
            for dt in ((1.0, 0.5) if vname in ("standin", "fog", "all26") else (1.0,)):
                rgba, steps, samp = R.render_naive(cams[cname], vol, W, H, dt_scale=dt)
                c_rgba, c_steps, c_samp = O.render(cams[cname], vol, W, H, dt_scale=dt)
                err = float(np.abs(rgba - c_rgba).max())
                assert err <= 1e-6, (vname, cname, dt, err)
                assert (steps == c_steps).all() and (samp == c_samp).all(), (vname, cname, dt)
                key = f"{vname}__{cname}__dt{dt}"
                cases[key + "__rgba"] = rgba
                cases[key + "__steps"] = steps.astype(np.uint16)
                cases[key + "__sampled"] = samp.astype(np.uint16)
                print(f"naive {key}: |numpy - C| = {err:.2e}, S_ref = {int(steps.sum())}")
    np.savez_compressed(os.path.join(OUT, "naive_64x64.npz"), **cases)

    # f16 volume
    f16v = R.volume_fog_f16(32)
    rgba, steps, samp = R.render_naive(cams["bonsai_1x1"], f16v, W, H, dt_scale=0.5)
    c_rgba, c_steps, _ = O.render(cams["bonsai_1x1"], f16v, W, H, dt_scale=0.5)
    assert np.abs(rgba - c_rgba).max() <= 1e-6 and (steps == c_steps).all()
    np.savez_compressed(os.path.join(OUT, "naive_f16_64x64.npz"), rgba=rgba, steps=steps.astype(np.uint16))

    # ---- compute (xor) path --------------------------------------------------------------
    rng = np.random.default_rng(20240607)
    n = 16
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    rad = np.sqrt((x - 7.5) ** 2 + (y - 7.5) ** 2 + (z - 7.5) ** 2) / 8.0
    den = np.stack([rng.random((n, n, n)), rng.random((n, n, n)), rng.random((n, n, n)),
                    np.clip(1.2 - rad, 0, 1) * rng.random((n, n, n))], axis=-1).astype(np.float16)
    nrm = (rng.random((n, n, n, 4)) * 2 - 1).astype(np.float16)
    Wc, Hc = 128, 72
    rgba, steps = R.render_compute(cams["xor_16x9"], den, nrm, Wc, Hc)
    c_rgba, c_steps, _ = O.render(cams["xor_16x9"], den, Wc, Hc, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm)
    assert np.abs(rgba - c_rgba).max() <= 1e-6 and (steps == c_steps).all()
    # tile with an offset that runs off the right/bottom edge (examples/xor/main.rs:82-92)
    t_rgba, t_steps = R.render_compute(cams["xor_16x9"], den, nrm, Wc, Hc, tile=(96, 48, 64, 64))
    np.savez_compressed(os.path.join(OUT, "compute_128x72.npz"), density=den.view(np.uint16), normals=nrm.view(np.uint16),
                        rgba=rgba, steps=steps.astype(np.uint16), tile_rgba=t_rgba, tile_steps=t_steps.astype(np.uint16))
    print(f"compute: S = {int(steps.sum())}, hit = {(steps > 0).mean():.3f}")

    # ---- procedural density (C3): no volume, the compute march over xor.wgsl's noise_volume ----------
    Wp, Hp = 96, 54
    p_rgba, p_steps = R.render_procedural(cams["xor_16x9"], Wp, Hp)
    cp_rgba, cp_steps = O.render_procedural(cams["xor_16x9"], Wp, Hp)
    assert np.abs(p_rgba - cp_rgba).max() <= 1e-6 and (p_steps == cp_steps).all()
    h_rgba, h_steps = R.render_procedural(cams["xor_16x9"], Wp, Hp, dt_scale=2.5)
    np.savez_compressed(os.path.join(OUT, "procedural_96x54.npz"), rgba=p_rgba, steps=p_steps.astype(np.uint16),
                        rgba_dt2p5=h_rgba, steps_dt2p5=h_steps.astype(np.uint16))
    print(f"procedural: S = {int(p_steps.sum())}, hit = {(p_steps > 0).mean():.3f}")

    # ---- unit vectors --------------------------------------------------------------------
    o = np.array([[-0.2385, 0.0206, 0.0258], [0.5, 0.5, -1.0], [0.5, 0.5, 0.5], [2.0, 2.0, 2.0], [0.25, 0.75, -3.0]], np.float32)
    d = np.array([[0.7, 0.5, 0.5], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 1.0, 1.0], [0.0, 0.0, 1.0]], np.float32)
    d = d / np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    t0, t1 = R.intersect_box([o[:, 0], o[:, 1], o[:, 2]], [d[:, 0], d[:, 1], d[:, 2]], 0.0, 1.0)
    pts = np.array([[0.5 / 32, 0.5 / 32, 0.5 / 32], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [1.0 / 32, 0.3, 0.7],
                    [0.999, 0.001, 0.5], [31.5 / 32, 31.5 / 32, 31.5 / 32], [0.26, 0.51, 0.77]], np.float32)
    tri, _ = R.sample_trilinear(vols["ramp_x"], [pts[:, 0], pts[:, 1], pts[:, 2]])
    tri_s, _ = R.sample_trilinear(standin32, [pts[:, 0], pts[:, 1], pts[:, 2]])
    xs = np.array([0.0, 0.001, 0.0031307, 0.0031308, 0.0031309, 0.01, 0.2, 0.5, 0.95, 1.0], np.float32)
    rs = np.array([0.0, 25 / 255, 25.5 / 255, 26 / 255, 0.1, 0.100001, 0.5, 0.9, 0.95, 1.0], np.float32)
    np.savez_compressed(os.path.join(OUT, "units.npz"), box_o=o, box_d=d, box_t0=t0, box_t1=t1, tri_pts=pts, tri_ramp_x=tri,
                        tri_standin=tri_s, srgb_x=xs, srgb_y=R.linear_to_srgb(xs), alpha_r=rs, alpha_a=R.transfer_alpha(rs),
                        alpha_raw=(rs * np.float32(255.0)).astype(np.float32), alpha_a_raw=R.transfer_alpha((rs * np.float32(255.0)).astype(np.float32), raw_unorm8=True))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {OUT}: {tot / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
