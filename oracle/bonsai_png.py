"""A second reference-held pin of the oracle, for the NAIVE path: the colours of `bonsai.png` (reference README.md:15).

TEST INFRASTRUCTURE ONLY.  `bonsai.png` is a 1280x720 capture of `cargo run --example bonsai`.  Its volume
(`bonsai_256x256x256_uint8.raw`, src/context/volume_texture.rs:33) is not in the checkout, so the picture cannot be regenerated --
but its COLOURS can be checked against what the shader can produce at all.  raycast_naive.wgsl:101-123 composites

    C = 0.5 A + 0.5 sum_i w_i cos(6.28318 (c a_i + d)),   sum_i w_i = A <= 1,   a_i = transfer(sample_i) in [0, a_max]

i.e. every pixel is a convex combination of black and the points P(a) = 0.5 + 0.5 cos(6.28318 (c a + d)) of ONE curve, whatever the volume
holds; then `linear_to_srgb` (:121-123), the rgba16f backbuffer, the present pass, 8 bits.  The oracle renders the curve itself: uniform
volumes of value v = 0 .. 255 seen through a ray that saturates (A >= 0.95: the early-out), C(v) = A_v P(a_v).

What the capture shows (measured in the build container, `python -m oracle.bonsai_png`):
  * with the present pass read as ONE MORE `linear_to_srgb` and no tone map, 99.97 % of the capture's 297 738 non-black pixels lie inside the
    convex hull of {0} and the oracle's curve (each pixel taken with its +-1 LSB box); read as ACES + sRGB (present.wgsl at the checkout's
    HEAD, which `volume.png` obeys) 77 % do, as a single sRGB 43 %, as linear values 10 %.  The capture has green values up to 249; through the
    ACES curve nothing above 232 can come out of a backbuffer value <= 1: the picture predates the tone map.
  * the capture's greenest colour, (135, 246, 135) -- the inside of the pot, where rays saturate in uniform material -- is the curve's point
    for v = 179 to within one LSB per channel.
So the palette's constants, C = 0.5 A + 0.5 G, the early-out's saturation and the shader's own sRGB step are the reference's.  What this does NOT
pin: the transfer function's edges and SURVEY F8's min(0.9, v) (they decide WHICH points of the curve a volume reaches, not the curve).

The capture's colours travel as a fixture (tests/golden/bonsai_png_colours.npz: a seeded sample of its non-black pixels and its extreme
colours along 26 directions, its greenest); where /root/reference exists the test re-derives them from the file.
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
REF_PNG = "/root/reference/bonsai.png"
GOLDEN_NPZ = os.path.join(ROOT, "tests", "golden", "bonsai_png_colours.npz")

SAMPLE = 20000  # non-black pixels kept in the fixture
SEED = 20261004
NON_BLACK = 8  # a pixel counts when its largest channel exceeds this
GREENEST = (135, 246, 135)

# acceptance bars
BAR_INSIDE = 0.999  # share of the sampled colours inside the hull, present pass = one more sRGB
BAR_INSIDE_ACES = 0.85  # ... and at most this share when the capture is decoded as ACES + sRGB (it measures 0.77): the test discriminates
BAR_GREENEST = 1  # LSB per channel between the capture's greenest colour and the curve


def present_srgb(c):
    """present.wgsl:23-30 (branch-free, exponent 0.41666), as oracle/vokselis_oracle.c:present_srgb."""
    c = np.asarray(c, np.float32)
    sel = np.ceil(c - np.float32(0.0031308))
    under = np.float32(12.92) * c
    over = np.float32(1.055) * np.power(np.maximum(c, 0), np.float32(0.41666)) - np.float32(0.055)
    return under * (1 - sel) + over * sel


def present_srgb_inv(s):
    s = np.asarray(s, np.float64)
    return np.where(s <= 12.92 * 0.0031308, s / 12.92, ((s + 0.055) / 1.055) ** (1.0 / 0.41666))


def aces_inv(y):
    """Inverse of present.wgsl:33-35 on [0, 1)."""
    y = np.minimum(np.asarray(y, np.float64), 0.99)
    a, b, c = 2.51 - 2.43 * y, 0.03 - 0.59 * y, -0.14 * y
    return (-b + np.sqrt(np.maximum(b * b - 4 * a * c, 0))) / (2 * a)


def curve_backbuffer(render_uniform) -> np.ndarray:
    """[256, 3] f32: the backbuffer colour (after the shader's linear_to_srgb) of a saturated ray through uniform material v = 0 .. 255.
    `render_uniform(v)` renders a 32^3 volume full of v and returns the rgb of a pixel whose ray crosses it."""
    return np.stack([np.asarray(render_uniform(v), np.float32)[:3] for v in range(256)])


def oracle_curve() -> np.ndarray:
    from . import oracle as O

    W, H = 16, 9
    blob = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)  # examples/bonsai/main.rs:68-74

    def one(v):
        rgba, _, _ = O.render(blob, np.full((32, 32, 32), v, np.uint8), W, H, dt_scale=1.0)
        return rgba[H // 2, W // 2]

    return curve_backbuffer(one)


def hull_of(curve_bb: np.ndarray):
    """Half-spaces (A, b) of conv({0} u C(v) u C(v) / 0.95), C = the linear composite colour (the shader's sRGB step undone)."""
    from scipy.spatial import ConvexHull

    C = present_srgb_inv(curve_bb)
    h = ConvexHull(np.vstack([np.zeros((1, 3)), C, C / 0.95]))
    return h.equations[:, :3], h.equations[:, 3]


def inside_share(colours_u8: np.ndarray, hull, decode) -> float:
    """Share of the colours whose +-1 LSB box meets the hull, after `decode` (8-bit value / 255 -> linear composite colour)."""
    A, b = hull
    c = colours_u8.astype(np.float64)

    def ins(p):
        return ((decode(np.clip(p, 0, 255) / 255.0) @ A.T + b) <= 1e-3).all(axis=1)

    return float((ins(c) | ins(c - 1.0) | ins(c + 1.0)).mean())


def decode_double_srgb(p):
    return present_srgb_inv(present_srgb_inv(p))


def decode_aces_srgb(p):
    return present_srgb_inv(aces_inv(present_srgb_inv(p)))


def nearest_on_curve(colour, curve_bb: np.ndarray):
    """(v, Linf in LSB) of the curve's 8-bit point nearest to `colour`, the present pass read as one more sRGB."""
    c8 = np.round(np.clip(present_srgb(curve_bb), 0, 1) * 255).astype(int)
    d = np.abs(c8 - np.asarray(colour, int)).max(axis=1)
    return int(np.argmin(d)), int(d.min())


def colours_of(png_rgb: np.ndarray):
    """The fixture's content from the capture: a seeded sample of the non-black pixels, and the extreme colours along 26 directions."""
    px = png_rgb.reshape(-1, 3)
    nz = px[px.max(axis=1) > NON_BLACK]
    rng = np.random.default_rng(SEED)
    sample = nz[rng.choice(len(nz), SAMPLE, replace=False)]
    dirs = np.array([(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)], np.int64)
    ext = np.stack([nz[np.argmax(nz.astype(np.int64) @ d)] for d in dirs])
    ext = np.vstack([ext, nz[np.argmax(nz[:, 1].astype(np.int64) - np.maximum(nz[:, 0], nz[:, 2]))][None]])  # ... and the greenest: G - max(R, B)
    return sample.astype(np.uint8), ext.astype(np.uint8), len(nz)


def load_pin():
    return np.load(GOLDEN_NPZ)


def main():
    from PIL import Image

    png = np.array(Image.open(REF_PNG).convert("RGB"))
    assert png.shape == (720, 1280, 3)
    sample, ext, n_nz = colours_of(png)
    curve = oracle_curve()
    hull = hull_of(curve)
    allnz = png.reshape(-1, 3)[png.reshape(-1, 3).max(axis=1) > NON_BLACK]
    out = {"non_black": n_nz, "inside_double_srgb_all": inside_share(allnz, hull, decode_double_srgb), "inside_aces_srgb_all": inside_share(allnz, hull, decode_aces_srgb),
           "inside_single_srgb_all": inside_share(allnz, hull, present_srgb_inv), "inside_double_srgb_sample": inside_share(sample, hull, decode_double_srgb),
           "greenest": nearest_on_curve(GREENEST, curve), "max_channel": int(png.max())}
    print(out)
    np.savez_compressed(GOLDEN_NPZ, sample=sample, extremes=ext, non_black=np.int64(n_nz), background=png[0, 0].astype(np.uint8),
                        inside_double_srgb_all=np.float64(out["inside_double_srgb_all"]), inside_aces_srgb_all=np.float64(out["inside_aces_srgb_all"]))
    print("wrote", GOLDEN_NPZ, os.path.getsize(GOLDEN_NPZ), "bytes")


if __name__ == "__main__":
    main()
