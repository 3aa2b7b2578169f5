"""Independent numpy (float32) restatement of the reference raycast shaders.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/vokselis_oracle.h).

Written from shaders/raycast_naive.wgsl / raycast_compute.wgsl and the arithmetic specification
in SURVEY.md Appendix A, *not* from vokselis_oracle.c: it is vectorised over rays (one masked
numpy step per march iteration) and exists only to cross-check the C oracle in this container
and to generate the fixtures under tests/golden/ (oracle/gen_golden.py).  numpy has no fused
multiply-add, so fma(a, b, c) is evaluated as float32(float64(a) * float64(b) + float64(c)); the
product is exact in float64, which leaves a ~2^-29 chance of a double-rounding difference per
operation -- the agreement criterion is therefore <= 1e-6, with step counts equal.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def _mat_vec(m16: np.ndarray, v4):
    """column-major 4x4 times vec4, accumulated ((c0*x + c1*y) + c2*z) + c3*w in float32."""
    cols = m16.reshape(4, 4)
    out = []
    for r in range(4):
        s = cols[0, r] * v4[0]
        s = s + cols[1, r] * v4[1]
        s = s + cols[2, r] * v4[2]
        s = s + cols[3, r] * v4[3]
        out.append(s.astype(np.float32))
    return out


def _normalize(d):
    ln = np.sqrt(((d[0] * d[0] + d[1] * d[1]).astype(np.float32) + d[2] * d[2]).astype(np.float32)).astype(np.float32)
    return [(c / ln).astype(np.float32) for c in d]


def intersect_box(o, d, lo, hi):
    """raycast_naive.wgsl:50-61; numpy minimum/maximum would propagate NaN, fmin/fmax do not."""
    tmin, tmax = [], []
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(3):
            inv = (f32(1.0) / d[i]).astype(np.float32)
            a = ((f32(lo) - o[i]) * inv).astype(np.float32)
            b = ((f32(hi) - o[i]) * inv).astype(np.float32)
            tmin.append(np.fmin(a, b))
            tmax.append(np.fmax(a, b))
    t0 = np.fmax(tmin[0], np.fmax(tmin[1], tmin[2]))
    t1 = np.fmin(tmax[0], np.fmin(tmax[1], tmax[2]))
    return t0, t1


def sample_trilinear(vol: np.ndarray, p, raw=False):
    """Linear, clamp-to-edge sample of a [nz,ny,nx] u8 (R8Unorm) or f16 volume at p in [0,1]^3."""
    nz, ny, nx = vol.shape
    dims = (nx, ny, nz)
    fr, i0, i1 = [], [], []
    for a in range(3):
        u = fma(p[a], f32(dims[a]), f32(-0.5))
        fl = np.floor(u)
        f = (u - fl).astype(np.float32)
        f = np.where(f >= f32(1.0), np.nextafter(f32(1.0), f32(0.0)), f)  # GPU fract semantic
        ii = np.nan_to_num(fl, nan=0.0, posinf=2**31 - 1, neginf=-2**31).astype(np.int64)
        fr.append(f)
        i0.append(np.clip(ii, 0, dims[a] - 1))
        i1.append(np.clip(ii + 1, 0, dims[a] - 1))
    is_u8 = vol.dtype == np.uint8

    def tap(ix, iy, iz):
        return vol[iz, iy, ix].astype(np.float32)

    t = [tap(i0[0], i0[1], i0[2]), tap(i1[0], i0[1], i0[2]), tap(i0[0], i1[1], i0[2]), tap(i1[0], i1[1], i0[2]),
         tap(i0[0], i0[1], i1[2]), tap(i1[0], i0[1], i1[2]), tap(i0[0], i1[1], i1[2]), tap(i1[0], i1[1], i1[2])]
    nonempty = np.zeros(p[0].shape, bool)
    for k in range(8):
        nonempty |= (t[k] > 25) if is_u8 else (t[k] > f32(0.1))

    def lerp(a, b, f):
        return fma(f, (b - a).astype(np.float32), a)

    c00, c10 = lerp(t[0], t[1], fr[0]), lerp(t[2], t[3], fr[0])
    c01, c11 = lerp(t[4], t[5], fr[0]), lerp(t[6], t[7], fr[0])
    c0, c1 = lerp(c00, c10, fr[1]), lerp(c01, c11, fr[1])
    r = lerp(c0, c1, fr[2])
    if is_u8 and not raw:  # raw: the filtered taps stay on their 0..255 scale (the march: transfer_alpha carries the 1/255)
        r = (r * (f32(1.0) / f32(255.0))).astype(np.float32)
    return r, nonempty


def transfer_alpha(x, raw_unorm8=False):
    """min(0.9, v) (clamp with low > high) then smoothstep(0.1, 1.2, .), with the affine part as ONE fused op
    whose constants carry the sample's scale: x is a value (c = 0.9, k1 = 1/1.1) or filtered R8Unorm taps on their
    0..255 scale (c = 229.5, k1 = 1/(255*1.1)); k2 = -0.1/1.1 (constants rounded from float64)."""
    c = f32(229.5) if raw_unorm8 else f32(0.9)
    k1 = f32(1.0 / (255.0 * 1.1)) if raw_unorm8 else f32(1.0 / 1.1)
    k2 = f32(-0.1 / 1.1)
    s = fma(np.fmin(np.asarray(x, np.float32), c), k1, k2)
    s = np.fmin(np.fmax(s, f32(0.0)), f32(1.0))
    return ((s * s).astype(np.float32) * fma(f32(-2.0), s, f32(3.0))).astype(np.float32)


def vertigo(a):
    tau = f32(6.28318)
    c = (f32(1.0), f32(1.7), f32(0.4))
    d = (f32(0.0), f32(0.15), f32(0.20))
    return [(f32(0.5) + f32(0.5) * np.cos((tau * (c[k] * a + d[k]).astype(np.float32)).astype(np.float32))).astype(np.float32)
            for k in range(3)]


def linear_to_srgb(x):
    x = x.astype(np.float32)
    hi = (f32(1.055) * np.power(np.maximum(x, f32(1e-30)), f32(1.0) / f32(2.4)).astype(np.float32) - f32(0.055)).astype(np.float32)
    return np.where(x <= f32(0.0031308), (f32(12.92) * x).astype(np.float32), hi)


def render_naive(camera_blob: bytes, vol: np.ndarray, W: int, H: int, dt_scale: float = 1.0, tile=None):
    """fs_main of raycast_naive.wgsl for every pixel of `tile` (default: the full frame).
    Returns (rgba [H,W,4] f32, steps [H,W] u32, sampled [H,W] u32)."""
    cam = np.frombuffer(camera_blob, np.float32)
    eye = [cam[0], cam[1], cam[2]]
    inv_proj = cam[20:36]
    tx, ty, tw, th = (0, 0, W, H) if tile is None else tile
    xs = np.arange(max(tx, 0), min(tx + tw, W))
    ys = np.arange(max(ty, 0), min(ty + th, H))
    rgba = np.zeros((H, W, 4), np.float32)
    steps = np.zeros((H, W), np.uint32)
    sampled = np.zeros((H, W), np.uint32)
    if xs.size == 0 or ys.size == 0:
        return rgba, steps, sampled
    X, Y = np.meshgrid(xs, ys)
    fx = (X.astype(np.float32) + f32(0.5)).ravel()
    fy = (Y.astype(np.float32) + f32(0.5)).ravel()
    ndcx = ((f32(2.0) * fx) / f32(W) - f32(1.0)).astype(np.float32)
    ndcy = (f32(1.0) - (f32(2.0) * fy) / f32(H)).astype(np.float32)
    one = np.ones_like(ndcx)
    q = _mat_vec(inv_proj, [ndcx, ndcy, one, one])
    d = _normalize([((q[i] / q[3]).astype(np.float32) - eye[i]).astype(np.float32) for i in range(3)])
    o = [np.full_like(ndcx, eye[i]) for i in range(3)]
    t0, t1 = intersect_box(o, d, 0.0, 1.0)
    hit = ~(t0 > t1)
    t0 = np.fmax(t0, f32(0.0))
    nz, ny, nx = vol.shape
    with np.errstate(divide="ignore"):
        dtv = [(f32(1.0) / (f32(n) * np.abs(d[i])).astype(np.float32)).astype(np.float32) for i, n in enumerate((nx, ny, nz))]
    dt = (f32(dt_scale) * np.fmin(dtv[0], np.fmin(dtv[1], dtv[2]))).astype(np.float32)
    p = [(o[i] + (t0 * d[i]).astype(np.float32)).astype(np.float32) for i in range(3)]
    st = [(d[i] * dt).astype(np.float32) for i in range(3)]
    n = ndcx.size
    C = [np.zeros(n, np.float32) for _ in range(3)]
    A = np.zeros(n, np.float32)
    t = t0.copy()
    nst = np.zeros(n, np.uint32)
    nsm = np.zeros(n, np.uint32)
    with np.errstate(invalid="ignore"):
        active = hit & (t < t1)
    while active.any():
        idx = np.nonzero(active)[0]
        pa = [p[i][idx] for i in range(3)]
        r, nonempty = sample_trilinear(vol, pa, raw=True)
        a = transfer_alpha(r, raw_unorm8=(vol.dtype == np.uint8))
        rgb = vertigo(a)
        nst[idx] += 1
        nsm[idx] += nonempty.astype(np.uint32)
        w = ((f32(1.0) - A[idx]) * a).astype(np.float32)
        for k in range(3):
            C[k][idx] = (C[k][idx] + (w * rgb[k]).astype(np.float32)).astype(np.float32)
        A[idx] = (A[idx] + w).astype(np.float32)
        done = A[idx] >= f32(0.95)
        cont = idx[~done]
        for i in range(3):
            p[i][cont] = (p[i][cont] + st[i][cont]).astype(np.float32)
        t[cont] = (t[cont] + dt[cont]).astype(np.float32)
        active[idx[done]] = False
        active[cont] = t[cont] < t1[cont]
    out = np.zeros((n, 4), np.float32)
    for k in range(3):
        out[:, k] = np.where(hit, linear_to_srgb(C[k]), f32(0.0))
    out[:, 3] = 1.0
    rgba[np.ix_(ys, xs)] = out.reshape(ys.size, xs.size, 4)
    steps[np.ix_(ys, xs)] = nst.reshape(ys.size, xs.size)
    sampled[np.ix_(ys, xs)] = nsm.reshape(ys.size, xs.size)
    return rgba, steps, sampled


# ------------------------------------------------------------------------------------------------
# raycast_compute.wgsl


def _smoothstep(e0, e1, x):
    inv = f32(1.0) / (f32(e1) - f32(e0))
    s = ((x - f32(e0)) * inv).astype(np.float32)
    s = np.fmin(np.fmax(s, f32(0.0)), f32(1.0))
    return ((s * s).astype(np.float32) * fma(f32(-2.0), s, f32(3.0))).astype(np.float32)


def _dot3(a, b):
    return (((a[0] * b[0]).astype(np.float32) + (a[1] * b[1]).astype(np.float32)).astype(np.float32) + (a[2] * b[2]).astype(np.float32)).astype(np.float32)


def render_compute(camera_blob: bytes, density: np.ndarray, normals: np.ndarray, W: int, H: int, dt_scale: float = 1.0,
                   tile=None):
    """`single` / `tile` of raycast_compute.wgsl (render + get_col2).  density/normals: f16 [nz,ny,nx,4]."""
    cam = np.frombuffer(camera_blob, np.float32)
    inv_proj = cam[20:36]
    tx, ty, tw, th = (0, 0, W, H) if tile is None else tile
    xs = np.arange(max(tx, 0), min(tx + tw, W))
    ys = np.arange(max(ty, 0), min(ty + th, H))
    rgba = np.zeros((H, W, 4), np.float32)
    steps = np.zeros((H, W), np.uint32)
    if xs.size == 0 or ys.size == 0:
        return rgba, steps
    X, Y = np.meshgrid(xs, ys)
    cx, cy = X.astype(np.float32).ravel(), Y.astype(np.float32).ravel()
    dx, dy = f32(W), f32(H)
    aspect = dy / dx
    sx = ((f32(2.0) * cx) / dx - f32(1.0)).astype(np.float32)
    sy = ((f32(2.0) * cy) / dy - f32(1.0)).astype(np.float32)
    sy = (sy * -aspect).astype(np.float32)
    zero, one = np.zeros_like(sx), np.ones_like(sx)
    vp = _mat_vec(inv_proj, [sx, sy, zero, one])
    vt = _mat_vec(inv_proj, [sx, sy, one, one])
    eye = [(vp[i] / vp[3]).astype(np.float32) for i in range(3)]
    d = _normalize([((vt[i] / vt[3]).astype(np.float32) - eye[i]).astype(np.float32) for i in range(3)])
    clear = (f32(0.023), f32(0.02), f32(0.02))
    t0, t1 = intersect_box(eye, d, -1.0, 1.0)
    with np.errstate(invalid="ignore"):
        hit = t0 < t1
    t0 = np.fmax(t0, f32(0.0))
    nz, ny, nx = density.shape[:3]
    bs = (f32(nx), f32(ny), f32(nz))
    with np.errstate(divide="ignore"):
        dtv = [(f32(1.0) / (bs[i] * np.abs(d[i])).astype(np.float32)).astype(np.float32) for i in range(3)]
    dt = (f32(dt_scale) * np.fmax(np.fmin(dtv[0], np.fmin(dtv[1], dtv[2])), f32(0.01))).astype(np.float32)
    hb = [b / f32(2.0) for b in bs]
    l1 = _normalize([np.array(f32(-2.0)), np.array(f32(-2.0)), np.array(f32(-1.0))])
    l2 = _normalize([np.array(f32(1.0)), np.array(f32(1.0)), np.array(f32(-1.0))])
    n = sx.size
    C = [np.full(n, clear[k], np.float32) for k in range(3)]
    A = np.full(n, f32(0.1), np.float32)
    t = t0.copy()
    nst = np.zeros(n, np.uint32)
    with np.errstate(invalid="ignore"):
        active = hit & (t < t1)
    den = density.astype(np.float32)
    nrm = normals.astype(np.float32)
    while active.any():
        idx = np.nonzero(active)[0]
        p = [(eye[i][idx] + (t[idx] * d[i][idx]).astype(np.float32)).astype(np.float32) for i in range(3)]
        ii = [np.trunc(((p[i] + f32(1.0)).astype(np.float32) * hb[i]).astype(np.float32)) for i in range(3)]
        ii = [np.nan_to_num(v, nan=0.0).astype(np.int64) for v in ii]
        inb = (ii[0] >= 0) & (ii[1] >= 0) & (ii[2] >= 0) & (ii[0] < nx) & (ii[1] < ny) & (ii[2] < nz)
        jx, jy, jz = [np.clip(ii[i], 0, (nx, ny, nz)[i] - 1) for i in range(3)]
        vc = np.where(inb[:, None], den[jz, jy, jx], f32(0.0)).astype(np.float32)
        nm = np.where(inb[:, None], nrm[jz, jy, jx], f32(0.0)).astype(np.float32)
        nst[idx] += 1
        sh = np.fmax(f32(0.0), _dot3([f32(0.0), f32(-1.0), f32(0.0)], [nm[:, 0], nm[:, 1], nm[:, 2]]))
        va = ((vc[:, 3] * vc[:, 3]).astype(np.float32) * vc[:, 3]).astype(np.float32)
        va = _smoothstep(0.0, 0.7, va)
        dl = np.fmax(_dot3([nm[:, 0], nm[:, 1], nm[:, 2]], l1), f32(0.0))
        ss = _smoothstep(0.3, 1.5, _dot3(p, l2))
        lc = (f32(1.0), f32(0.1), f32(0.13))
        col = [(vc[:, k] + (((f32(3.0) * lc[k]) * dl).astype(np.float32) * ss).astype(np.float32)).astype(np.float32) for k in range(3)]
        bl = (f32(0.9) * np.fmin(np.fmax((f32(0.5) - (f32(0.5) * nm[:, 1]).astype(np.float32)).astype(np.float32), f32(0.0)), f32(1.0))).astype(np.float32)
        blc = (f32(0.0), f32(0.0), f32(0.6))
        shade = [((sh * (f32(1.0) - f32(0.2))).astype(np.float32) + ((bl * blc[k]).astype(np.float32) * f32(0.2)).astype(np.float32)).astype(np.float32)
                 for k in range(3)]
        w = ((f32(1.0) - A[idx]) * va).astype(np.float32)
        for k in range(3):
            C[k][idx] = (C[k][idx] + ((w * col[k]).astype(np.float32) * shade[k]).astype(np.float32)).astype(np.float32)
        A[idx] = (A[idx] + w).astype(np.float32)
        done = A[idx] >= f32(0.95)
        cont = idx[~done]
        t[cont] = (t[cont] + dt[cont]).astype(np.float32)
        active[idx[done]] = False
        active[cont] = t[cont] < t1[cont]
    out = np.zeros((n, 4), np.float32)
    for k in range(3):
        out[:, k] = np.where(hit, C[k], clear[k])
    out[:, 3] = 1.0
    rgba[np.ix_(ys, xs)] = out.reshape(ys.size, xs.size, 4)
    steps[np.ix_(ys, xs)] = nst.reshape(ys.size, xs.size)
    return rgba, steps


# ------------------------------------------------------------------------------------------------
# C3: procedural density, no volume -- the compute twin's march over shaders/xor.wgsl:18-61


def _xor_hash(h):
    """xor.wgsl:18-20, fract(sin(h) * 43758.5453123); the sine is numpy's float64 sine rounded to f32 (the C
    oracle uses its own f64 Cody-Waite evaluation: the two agree wherever neither sits on a rounding tie)."""
    s = np.sin(h.astype(np.float64)).astype(np.float32)
    v = (s * f32(43758.5453123)).astype(np.float32)
    return (v - np.floor(v)).astype(np.float32)


def _xor_mix(a, b, t):
    return ((a * (f32(1.0) - t).astype(np.float32)).astype(np.float32) + (b * t).astype(np.float32)).astype(np.float32)


def _xor_noise(x):
    p = [np.floor(v).astype(np.float32) for v in x]
    f = [(v - np.floor(v)).astype(np.float32) for v in x]
    f = [((v * v).astype(np.float32) * (f32(3.0) - (f32(2.0) * v).astype(np.float32)).astype(np.float32)).astype(np.float32) for v in f]
    n = ((p[0] + (p[1] * f32(157.0)).astype(np.float32)).astype(np.float32) + (f32(113.0) * p[2]).astype(np.float32)).astype(np.float32)
    h = lambda o: _xor_hash((n + f32(o)).astype(np.float32))
    return _xor_mix(_xor_mix(_xor_mix(h(0.0), h(1.0), f[0]), _xor_mix(h(157.0), h(158.0), f[0]), f[1]),
                    _xor_mix(_xor_mix(h(113.0), h(114.0), f[0]), _xor_mix(h(270.0), h(271.0), f[0]), f[1]), f[2])


def _xor_fbm(p):
    f = (f32(0.5) * _xor_noise(p)).astype(np.float32)
    p = [(v * f32(2.01)).astype(np.float32) for v in p]
    f = (f + (f32(0.25) * _xor_noise(p)).astype(np.float32)).astype(np.float32)
    p = [(v * f32(2.02)).astype(np.float32) for v in p]
    return (f + (f32(0.125) * _xor_noise(p)).astype(np.float32)).astype(np.float32)


def xor_noise_volume(c, time=0.0):
    """xor.wgsl:55-61 -> (val, alpha) at generator coordinates c (three f32 arrays)."""
    off = (f32(1.0), (np.sin(np.float64(f32(time))).astype(np.float32) * f32(0.1)).astype(np.float32), f32(21.0))
    pos = [((c[i] + off[i]).astype(np.float32) * f32(32.0)).astype(np.float32) for i in range(3)]
    val = _xor_fbm(pos)
    ln = np.sqrt((((c[0] * c[0]).astype(np.float32) + (c[1] * c[1]).astype(np.float32)).astype(np.float32) + (c[2] * c[2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return val, (val * _smoothstep(0.5, 0.25, ln)).astype(np.float32)


def render_procedural(camera_blob: bytes, W: int, H: int, dt_scale: float = 1.0, time: float = 0.0):
    """vo_render(mode=PROCEDURAL): render()/get_col2() of raycast_compute.wgsl with the texel loads replaced by
    noise_volume(p/2), colour = density.rgb/2, no normals.  Returns (rgba [H,W,4] f32, steps [H,W] u32)."""
    cam = np.frombuffer(camera_blob, np.float32)
    inv_proj = cam[20:36]
    X, Y = np.meshgrid(np.arange(W), np.arange(H))
    cx, cy = X.astype(np.float32).ravel(), Y.astype(np.float32).ravel()
    dx, dy = f32(W), f32(H)
    sx = ((f32(2.0) * cx) / dx - f32(1.0)).astype(np.float32)
    sy = ((f32(2.0) * cy) / dy - f32(1.0)).astype(np.float32)
    sy = (sy * -(dy / dx)).astype(np.float32)
    zero, one = np.zeros_like(sx), np.ones_like(sx)
    vp = _mat_vec(inv_proj, [sx, sy, zero, one])
    vt = _mat_vec(inv_proj, [sx, sy, one, one])
    eye = [(vp[i] / vp[3]).astype(np.float32) for i in range(3)]
    d = _normalize([((vt[i] / vt[3]).astype(np.float32) - eye[i]).astype(np.float32) for i in range(3)])
    clear = (f32(0.023), f32(0.02), f32(0.02))
    t0, t1 = intersect_box(eye, d, -1.0, 1.0)
    with np.errstate(invalid="ignore"):
        hit = t0 < t1
    t0 = np.fmax(t0, f32(0.0))
    with np.errstate(divide="ignore"):
        dtv = [(f32(1.0) / (f32(256.0) * np.abs(d[i])).astype(np.float32)).astype(np.float32) for i in range(3)]
    dt = (f32(dt_scale) * np.fmax(np.fmin(dtv[0], np.fmin(dtv[1], dtv[2])), f32(0.01))).astype(np.float32)
    n = sx.size
    C = [np.full(n, clear[k], np.float32) for k in range(3)]
    A = np.full(n, f32(0.1), np.float32)
    t = t0.copy()
    nst = np.zeros(n, np.uint32)
    with np.errstate(invalid="ignore"):
        active = hit & (t < t1)
    while active.any():
        idx = np.nonzero(active)[0]
        p = [(eye[i][idx] + (t[idx] * d[i][idx]).astype(np.float32)).astype(np.float32) for i in range(3)]
        val, alpha = xor_noise_volume([(v * f32(0.5)).astype(np.float32) for v in p], time)
        nst[idx] += 1
        vc = (val / f32(2.0)).astype(np.float32)
        va = _smoothstep(0.0, 0.7, ((alpha * alpha).astype(np.float32) * alpha).astype(np.float32))
        w = ((f32(1.0) - A[idx]) * va).astype(np.float32)
        for k in range(3):
            C[k][idx] = (C[k][idx] + (w * vc).astype(np.float32)).astype(np.float32)
        A[idx] = (A[idx] + w).astype(np.float32)
        done = A[idx] >= f32(0.95)
        cont = idx[~done]
        t[cont] = (t[cont] + dt[cont]).astype(np.float32)
        active[idx[done]] = False
        active[cont] = t[cont] < t1[cont]
    out = np.zeros((n, 4), np.float32)
    for k in range(3):
        out[:, k] = np.where(hit, C[k], clear[k])
    out[:, 3] = 1.0
    return out.reshape(H, W, 4), nst.reshape(H, W)


# ------------------------------------------------------------------------------------------------
# deterministic volumes (integer-only; must be bit-identical to vo_volume_* in the C oracle)


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
    m = (1 << sh) - 1
    S = 1 << sh
    cx, cy, cz = X >> sh, Y >> sh, Z >> sh
    fx, fy, fz = (X & m).astype(np.uint64), (Y & m).astype(np.uint64), (Z & m).astype(np.uint64)
    acc = np.zeros(X.shape, np.uint64)
    for dz in range(2):
        for dy in range(2):
            for dx in range(2):
                w = (fx if dx else S - fx) * (fy if dy else S - fy) * (fz if dz else S - fz)
                acc += w * (_hash3(cx + dx, cy + dy, cz + dz, seed) >> np.uint32(24)).astype(np.uint64)
    return (acc >> np.uint64(3 * sh)).astype(np.int64)


def volume_standin_u8(n, seed=0x5EED0001) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    out = np.empty((nz, ny, nx), np.uint8)
    y, x = np.meshgrid(np.arange(ny, dtype=np.int64), np.arange(nx, dtype=np.int64), indexing="ij")
    X = ((2 * x + 1) * 2048) // nx
    Y = ((2 * y + 1) * 2048) // ny
    for z in range(nz):
        Z = np.full_like(X, ((2 * z + 1) * 2048) // nz)
        n_lo = _vnoise(X, Y, Z, 9, seed ^ 0x1111)
        n_hi = _vnoise(X, Y, Z, 7, seed ^ 0x2222)
        h = _hash3(x, y, np.full_like(x, z), seed ^ 0x3333).astype(np.int64)
        v = np.where((h & 0x1FF) == 0, 26 + ((h >> 9) & 15), (h >> 16) % 21)  # air + speckle
        # canopy
        dx, dy, dz = X - 2150, Y - 2850, Z - 1950
        q = (dx * dx * 256) // (1750 * 1750) + (dy * dy * 256) // (1050 * 1050) + (dz * dz * 256) // (1750 * 1750)
        f = (2 * n_lo + n_hi) // 3
        d = f - q // 3 - 52
        v = np.where((q < 256) & (d > 0), np.minimum(28 + 2 * d, 225), v)
        # trunk
        hh = Y - 900
        cxx = 2048 + (hh * hh) // 8000
        czz = 2048 - hh // 6
        rr = 230 - hh // 12
        ddx, ddz = X - cxx, Z - czz
        v = np.where((Y >= 900) & (Y < 2600) & (ddx * ddx + ddz * ddz < rr * rr), 110 + (n_hi >> 2), v)
        # pot
        dx, dy, dz = X - 2048, Y - 600, Z - 2048
        v = np.where(dx * dx + 7 * dy * dy + dz * dz < 1400 * 1400, 232 + (n_hi >> 4), v)
        out[z] = v.astype(np.uint8)
    return out


def volume_fog_u8(n, seed=0x5EED0002, lo=20, span=12) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    return (lo + (_hash3(x, y, z, seed) >> np.uint32(8)) % np.uint32(span)).astype(np.uint8)


def volume_fog_f16(n, seed=0x5EED0004) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    bits = (np.uint32(0x2D1F) + (_hash3(x, y, z, seed) >> np.uint32(8)) % np.uint32(656)).astype(np.uint16)
    return bits.view(np.float16)
