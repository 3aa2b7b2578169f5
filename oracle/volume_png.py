"""The one reference-held pin of the oracle: `volume.png` (reference README.md:21).

TEST INFRASTRUCTURE ONLY.  `volume.png` is a 958x1050 window capture of `cargo run --example xor`.  Every stage that
produced it can be regenerated from the reference's own sources, all of which the oracle restates:

    shaders/xor.wgsl:18-78             cs_main at t = 0 (examples/xor/xor_compute.rs:199)  -> vo_volume_xor
    shaders/raycast_compute.wgsl:62-131 render / get_col2 into the fixed 1280x720 backbuffer            -> vo_render(COMPUTE_NEAREST)
    shaders/present.wgsl:98-119        bilinear resample to the window, ACES, branch-free sRGB, 8 bit   -> vo_present
    examples/xor/main.rs:273-279       Camera::new(3, -0.5, 1, 0, aspect) -- then moved by the user

What the capture does NOT record is the camera at the moment of the screenshot (the user had zoomed and orbited), so the
three orbit parameters are fitted (`fit()` below: Nelder-Mead on the blurred image difference) and stored with the
fixtures.  The fine grain of the picture cannot be matched by anybody: the density is `fract(sin(h) * 43758.5453)`
(xor.wgsl:18-20), which amplifies the last bits of the capturing GPU's own `sin` by 4e4.  What CAN be matched, and is
asserted by tests/test_oracle_cpu.py::test_volume_png_pin (oracle) and tests/test_parity_gpu.py::test_volume_png_pin_hip
(HIP path): the background colour exactly (clear colour -> ACES -> sRGB -> 8 bit), the silhouette's box, the blurred
picture (correlation, mean difference) and the side the pink directional light falls on (handedness, the -H/W y scale,
the near-plane eye, light directions).

Run in the build container (the only place /root/reference exists):  python -m oracle.volume_png
It writes tests/golden/volume_png_pin.npz (camera, a blurred 1/8-scale copy of the capture, metrics) and
tests/golden/volume_png_oracle_frame.png (the oracle's own 958x1050 frame at the fitted camera).
"""
from __future__ import annotations

import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
REF_PNG = "/root/reference/volume.png"
GOLDEN_NPZ = os.path.join(ROOT, "tests", "golden", "volume_png_pin.npz")
GOLDEN_FRAME = os.path.join(ROOT, "tests", "golden", "volume_png_oracle_frame.png")

WIN_W, WIN_H = 958, 1050  # the capture's size = the window's inner size when F11 was pressed
BB_W, BB_H = 1280, 720  # HdrBackBuffer::DEFAULT_RESOLUTION (src/context/hdr_backbuffer.rs:10-11)
XOR_N = 256  # examples/xor/xor_compute.rs: 256^3 pair volume
TARGET = (0.0, 0.0, 0.0)
BACKGROUND = (30, 26, 26)  # what (0.023, 0.02, 0.02) becomes through present.wgsl
BLUR_SIGMA = 6.0
DS = 8  # the committed copy of the capture is blurred and then decimated by this factor

# acceptance bars
BAR_CORR = 0.93
BAR_MEAN_ABS = 3.5  # of 255, on the blurred pictures
BAR_BBOX = 0.03  # of the window's width / height
BAR_PINK = 0.08  # distance of the pink centroids, in units of the silhouette's diagonal


def _blur(a: np.ndarray, sigma: float = BLUR_SIGMA) -> np.ndarray:
    from scipy.ndimage import gaussian_filter

    return gaussian_filter(a.astype(np.float32), (sigma, sigma, 0))


def blurred_ds(rgb_u8: np.ndarray) -> np.ndarray:
    """[H,W,3] u8 -> blurred, decimated f32 [ceil(H/DS), ceil(W/DS), 3] (the form the capture is committed in)."""
    return np.ascontiguousarray(_blur(rgb_u8[..., :3])[DS // 2::DS, DS // 2::DS])


def camera(zoom: float, pitch: float, yaw: float) -> bytes:
    from . import oracle as O

    # the camera's aspect is the WINDOW's (src/context.rs:248), the backbuffer stays 1280x720
    return O.camera_blob(float(zoom), float(pitch), float(yaw), TARGET, WIN_W / WIN_H)


_volume = None


def xor_volume():
    global _volume
    if _volume is None:
        from . import oracle as O

        _volume = O.volume_xor(XOR_N, 0.0)
    return _volume


def render_oracle(zoom: float, pitch: float, yaw: float) -> np.ndarray:
    """The reference's whole xor frame through the oracle: [WIN_H, WIN_W, 4] u8."""
    from . import oracle as O

    den, nrm = xor_volume()
    rgba, _, _ = O.render(camera(zoom, pitch, yaw), den, BB_W, BB_H, mode=O.MODE_COMPUTE_NEAREST, volume2=nrm, want_counts=False)
    return O.present(rgba, WIN_W, WIN_H)


def _bbox(blur_img: np.ndarray, thresh: float = 6.0):
    """Silhouette box of a blurred picture: pixels whose colour is more than `thresh`/255 away from the background."""
    d = np.abs(blur_img - np.asarray(BACKGROUND, np.float32)).max(axis=-1)
    ys, xs = np.nonzero(d > thresh)
    return float(xs.min()), float(xs.max()), float(ys.min()), float(ys.max())


def _pink_centroid(blur_img: np.ndarray):
    """Centroid of the pink directional light: where red exceeds blue by more than the background's own 4/255."""
    w = np.clip(blur_img[..., 0] - blur_img[..., 2] - (BACKGROUND[0] - BACKGROUND[2]) - 2.0, 0.0, None)
    yy, xx = np.mgrid[0:w.shape[0], 0:w.shape[1]]
    s = float(w.sum())
    return float((w * xx).sum() / s), float((w * yy).sum() / s), s


def metrics(frame_u8: np.ndarray, capture_blur_ds: np.ndarray) -> dict:
    """Compare a rendered 958x1050 frame with the committed (blurred, decimated) copy of the capture."""
    mine = blurred_ds(frame_u8)
    ref = capture_blur_ds.astype(np.float32)
    assert mine.shape == ref.shape, (mine.shape, ref.shape)
    bx = np.array(_bbox(mine)) * DS
    rx = np.array(_bbox(ref)) * DS
    cx, cy = 0.5 * (rx[0] + rx[1]), 0.5 * (rx[2] + rx[3])
    diag = float(np.hypot(rx[1] - rx[0], rx[3] - rx[2]))
    pm = _pink_centroid(mine)
    pr = _pink_centroid(ref)
    return {
        "corr": float(np.corrcoef(mine.ravel(), ref.ravel())[0, 1]),
        "mean_abs": float(np.abs(mine - ref).mean()),
        "bbox_mine": bx.tolist(),
        "bbox_capture": rx.tolist(),
        "bbox_err_x": float(max(abs(bx[0] - rx[0]), abs(bx[1] - rx[1])) / WIN_W),
        "bbox_err_y": float(max(abs(bx[2] - rx[2]), abs(bx[3] - rx[3])) / WIN_H),
        "pink_mine": [pm[0] * DS, pm[1] * DS],
        "pink_capture": [pr[0] * DS, pr[1] * DS],
        "pink_dist": float(np.hypot(pm[0] - pr[0], pm[1] - pr[1]) * DS / diag),
        # the side of the silhouette's centre the light falls on: +1 right / -1 left, +1 below / -1 above (image y is down)
        "pink_side_mine": [int(np.sign(pm[0] * DS - cx)), int(np.sign(pm[1] * DS - cy))],
        "pink_side_capture": [int(np.sign(pr[0] * DS - cx)), int(np.sign(pr[1] * DS - cy))],
        "pink_mass_ratio": float(pm[2] / pr[2]),
    }


def check(m: dict) -> None:
    """The acceptance bars, shared by the CPU and the GPU test."""
    assert m["corr"] >= BAR_CORR, m
    assert m["mean_abs"] <= BAR_MEAN_ABS, m
    assert m["bbox_err_x"] <= BAR_BBOX and m["bbox_err_y"] <= BAR_BBOX, m
    assert m["pink_side_mine"] == m["pink_side_capture"], m
    assert m["pink_dist"] <= BAR_PINK, m


def load_pin() -> dict:
    z = np.load(GOLDEN_NPZ)
    return {k: z[k] for k in z.files}


def load_oracle_frame() -> np.ndarray:
    from PIL import Image

    return np.array(Image.open(GOLDEN_FRAME).convert("RGB"))


def fit(capture_rgb: np.ndarray, start=(2.05, -0.4, 4.4), verbose=True):
    """Nelder-Mead over (zoom, pitch, yaw) on the blurred difference, restarted once from its own optimum (the objective is
    rough at the scale of the noise's grain)."""
    from scipy.optimize import minimize

    ref = blurred_ds(capture_rgb)
    rb = ref[..., 0] - ref[..., 2]
    n = [0]

    def f(x):
        mine = blurred_ds(render_oracle(*x))
        e = float(np.abs(mine - ref).mean() + np.abs((mine[..., 0] - mine[..., 2]) - rb).mean())
        n[0] += 1
        if verbose:
            print(f"  fit {n[0]:3d}: zoom {x[0]:.4f} pitch {x[1]:.4f} yaw {x[2]:.4f} -> {e:.4f}", flush=True)
        return e

    x = np.asarray(start, float)
    for step in (0.1, 0.03):
        simplex = [x, x + [step, 0, 0], x + [0, step, 0], x + [0, 0, 1.5 * step]]
        r = minimize(f, x, method="Nelder-Mead", options=dict(xatol=1e-3, fatol=2e-3, initial_simplex=simplex, maxfev=120))
        x = r.x
    return tuple(float(v) for v in x)


def main() -> None:
    from PIL import Image

    if not os.path.exists(REF_PNG):
        sys.exit("the capture lives under /root/reference: run this in the build container")
    cap = np.array(Image.open(REF_PNG).convert("RGB"))
    assert cap.shape == (WIN_H, WIN_W, 3)
    assert tuple(cap[0, 0]) == BACKGROUND and tuple(cap[-1, -1]) == BACKGROUND
    zoom, pitch, yaw = fit(cap)
    # f32 values are what every host hands to the blob builder
    zoom, pitch, yaw = (float(np.float32(v)) for v in (zoom, pitch, yaw))
    frame = render_oracle(zoom, pitch, yaw)
    cap_ds = blurred_ds(cap)
    m = metrics(frame, cap_ds)
    print("fitted camera: zoom %.6f pitch %.6f yaw %.6f" % (zoom, pitch, yaw))
    for k, v in m.items():
        print(f"  {k}: {v}")
    check(m)
    np.savez_compressed(
        GOLDEN_NPZ,
        orbit=np.array([zoom, pitch, yaw], np.float32),
        camera=np.frombuffer(camera(zoom, pitch, yaw), np.uint8),
        capture_blur_ds=cap_ds.astype(np.float16),  # 132 x 120 x 3: the capture, blurred (sigma 6 px) and decimated by 8
        background=np.array(BACKGROUND, np.uint8),
        corr=np.float32(m["corr"]),
        mean_abs=np.float32(m["mean_abs"]),
    )
    Image.fromarray(frame[..., :3]).save(GOLDEN_FRAME, optimize=True)
    print("wrote", GOLDEN_NPZ, os.path.getsize(GOLDEN_NPZ), "bytes;", GOLDEN_FRAME, os.path.getsize(GOLDEN_FRAME), "bytes")


if __name__ == "__main__":
    main()
