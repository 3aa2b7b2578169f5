"""Deterministic synthetic input volumes (host side, numpy; integer arithmetic only).

The reference embeds `bonsai_256x256x256_uint8.raw` with include_bytes!
(src/context/volume_texture.rs:33) and the file is absent from the checkout, so the
bench and the tests render these instead; the real file is a drop-in via
`VolumeTexture.from_raw`.  Bit-identical to the device generator (`vk_volume_generate`) and to the
oracle's C generator (asserted by the tests).  Layout: [nz, ny, nx], x fastest
(volume_texture.rs:50-59).
"""
from __future__ import annotations

import numpy as np


def _lowbias32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def _hash3(x, y, z, seed):
    with np.errstate(over="ignore"):
        k = (x.astype(np.uint32) * np.uint32(0x9E3779B1) + y.astype(np.uint32) * np.uint32(0x85EBCA77)
             + z.astype(np.uint32) * np.uint32(0xC2B2AE3D)).astype(np.uint32)
        return _lowbias32(np.uint32(seed) ^ k)


def _vnoise(X, Y, Z, sh, seed):
    """Integer trilinear value noise on 12-bit coordinates, lattice cell 2^sh units -> 0..255."""
    m, S = (1 << sh) - 1, 1 << sh
    cx, cy, cz = X >> sh, Y >> sh, Z >> sh
    fx, fy, fz = (X & m).astype(np.uint64), (Y & m).astype(np.uint64), (Z & m).astype(np.uint64)
    acc = np.zeros(X.shape, np.uint64)
    for dz in range(2):
        for dy in range(2):
            for dx in range(2):
                w = (fx if dx else S - fx) * (fy if dy else S - fy) * (fz if dz else S - fz)
                acc += w * (_hash3(cx + dx, cy + dy, cz + dz, seed) >> np.uint32(24)).astype(np.uint64)
    return (acc >> np.uint64(3 * sh)).astype(np.int64)


def _dims(n):
    return (n, n, n) if np.isscalar(n) else tuple(n)


def bonsai_standin(n=256, seed=0x5EED0001) -> np.ndarray:
    """256^3-style u8 stand-in for the bonsai CT: dense pot (>= 232), mid-density bent trunk, smooth
    noise-thresholded canopy, air = white noise 0..20 (exactly transparent under the reference's
    transfer function) with 0.2 % speckle 26..41.  ~78 % of the voxels are <= 25."""
    nx, ny, nz = _dims(n)
    out = np.empty((nz, ny, nx), np.uint8)
    y, x = np.meshgrid(np.arange(ny, dtype=np.int64), np.arange(nx, dtype=np.int64), indexing="ij")
    X = ((2 * x + 1) * 2048) // nx  # centre-sampled 12-bit coordinates: 16 units per voxel at 256^3
    Y = ((2 * y + 1) * 2048) // ny
    for z in range(nz):
        Z = np.full_like(X, ((2 * z + 1) * 2048) // nz)
        n_lo = _vnoise(X, Y, Z, 9, seed ^ 0x1111)
        n_hi = _vnoise(X, Y, Z, 7, seed ^ 0x2222)
        h = _hash3(x, y, np.full_like(x, z), seed ^ 0x3333).astype(np.int64)
        v = np.where((h & 0x1FF) == 0, 26 + ((h >> 9) & 15), (h >> 16) % 21)
        dx, dy, dz = X - 2150, Y - 2850, Z - 1950
        q = (dx * dx * 256) // (1750 * 1750) + (dy * dy * 256) // (1050 * 1050) + (dz * dz * 256) // (1750 * 1750)
        d = (2 * n_lo + n_hi) // 3 - q // 3 - 52
        v = np.where((q < 256) & (d > 0), np.minimum(28 + 2 * d, 225), v)
        hh = Y - 900
        ddx, ddz, rr = X - (2048 + (hh * hh) // 8000), Z - (2048 - hh // 6), 230 - hh // 12
        v = np.where((Y >= 900) & (Y < 2600) & (ddx * ddx + ddz * ddz < rr * rr), 110 + (n_hi >> 2), v)
        dx, dy, dz = X - 2048, Y - 600, Z - 2048
        v = np.where(dx * dx + 7 * dy * dy + dz * dz < 1400 * 1400, 232 + (n_hi >> 4), v)
        out[z] = v.astype(np.uint8)
    return out


def _dense_core(x, y, z, nx, ny, nz):
    """The dense ball of the "dense-core" fog variants: radius a quarter of the smallest dimension."""
    dx, dy, dz = 2 * x.astype(np.int64) + 1 - nx, 2 * y.astype(np.int64) + 1 - ny, 2 * z.astype(np.int64) + 1 - nz
    r = min(nx, ny, nz) // 2
    return dx * dx + dy * dy + dz * dz < r * r


def fog_u8(n=256, seed=0x5EED0002, lo=20, span=12, dense_core=False) -> np.ndarray:
    """Uniform u8 in [lo, lo+span): alpha per step <= 1.4e-3, no ray ever reaches 0.95 (C2-fog); dense_core: 232..255 inside the ball."""
    nx, ny, nz = _dims(n)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    h = _hash3(x, y, z, seed) >> np.uint32(8)
    v = lo + h % np.uint32(span)
    if dense_core:
        v = np.where(_dense_core(x, y, z, nx, ny, nz), np.uint32(232) + h % np.uint32(24), v)
    return v.astype(np.uint8)


def fog_f16(n=256, seed=0x5EED0004, dense_core=False) -> np.ndarray:
    """f16 bit patterns 0x2D1F (0.08) .. 0x2FAE (0.12) (C4); dense_core: 0x3B9A (0.95) .. inside the ball."""
    nx, ny, nz = _dims(n)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    h = _hash3(x, y, z, seed) >> np.uint32(8)
    v = np.uint32(0x2D1F) + h % np.uint32(656)
    if dense_core:
        v = np.where(_dense_core(x, y, z, nx, ny, nz), np.uint32(0x3B9A) + h % np.uint32(64), v)
    return v.astype(np.uint16).view(np.float16)
