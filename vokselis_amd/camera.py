"""Orbit camera -- host mirror of the reference's `Camera` / `CameraUniform` (src/camera.rs).

Same names, argument meaning and clamping as the reference; f32 arithmetic throughout.  The
144-byte blob from `get_proj_view_matrix()` is what `vk_set_camera` takes (byte-identical layout
to `CameraUniform`, src/camera.rs:5-11: view_position[4], proj_view[4][4], inv_proj[4][4],
matrices column-major).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

f32 = np.float32

# One camera-blob builder: every host builds the 144 bytes with the same binary32 operations in the
# same order, so the same orbit gives the same bytes -- and therefore the same rays and trip counts -- from Python,
# from the C++ host and from the test oracle.  Sines and cosines are the C library's binary32 sinf / cosf (numpy's own
# float32 sine may differ from it in the last place), the inverse is the f32 cofactor expansion written out below.
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _fn in (_libm.sinf, _libm.cosf):
    _fn.restype = ctypes.c_float
    _fn.argtypes = [ctypes.c_float]


def sinf(x) -> np.float32:
    return f32(_libm.sinf(float(f32(x))))


def cosf(x) -> np.float32:
    return f32(_libm.cosf(float(f32(x))))


def _v3(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).reshape(3)


def _normalize(v: np.ndarray) -> np.ndarray:
    ln = np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]), dtype=np.float32)
    return (v / ln).astype(np.float32)


def _cross(a, b) -> np.ndarray:
    return np.array(
        [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=np.float32
    )


def _dot(a, b) -> np.float32:
    return f32(f32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def look_at_rh(eye, center, up) -> np.ndarray:
    """glam 0.20.5 Mat4::look_at_rh; returns a [4 columns][4 rows] f32 array."""
    f = _normalize(_v3(center) - _v3(eye))
    s = _normalize(_cross(f, _v3(up)))
    u = _cross(s, f)
    e = _v3(eye)
    return np.array(
        [[s[0], u[0], -f[0], 0], [s[1], u[1], -f[1], 0], [s[2], u[2], -f[2], 0],
         [-_dot(s, e), -_dot(u, e), _dot(f, e), 1]], dtype=np.float32)


def perspective_rh(fovy, aspect, z_near, z_far) -> np.ndarray:
    """glam 0.20.5 Mat4::perspective_rh (depth 0..1); [column][row]."""
    half = f32(0.5) * f32(fovy)
    sn, cs = sinf(half), cosf(half)
    h = f32(cs / sn)
    w = f32(h / f32(aspect))
    r = f32(f32(z_far) / f32(f32(z_near) - f32(z_far)))
    return np.array([[w, 0, 0, 0], [0, h, 0, 0], [0, 0, r, -1], [0, 0, f32(r * f32(z_near)), 0]], dtype=np.float32)


def mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a * b for [column][row] matrices, accumulated column by column like glam."""
    out = np.zeros((4, 4), np.float32)
    for c in range(4):
        acc = a[0] * b[c, 0]
        acc = acc + a[1] * b[c, 1]
        acc = acc + a[2] * b[c, 2]
        acc = acc + a[3] * b[c, 3]
        out[c] = acc
    return out


def mat4_inverse(m: np.ndarray) -> np.ndarray:
    """General 4x4 inverse by cofactors, binary32 throughout, products and sums left to right as written (glam's
    SIMD ordering cannot be reproduced offline; this order is the one every host of this build
    uses).  m and the result are [column][row]."""
    m = [f32(v) for v in np.asarray(m, np.float32).reshape(16)]
    a = [f32(0)] * 16
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10]
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10]
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9]
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9]
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10]
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10]
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9]
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9]
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6]
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6]
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5]
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5]
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6]
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6]
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5]
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5]
    det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12]
    with np.errstate(divide="ignore", invalid="ignore"):
        rdet = f32(1.0) / det
        return np.array([v * rdet for v in a], dtype=np.float32).reshape(4, 4)


class Camera:
    ZFAR = f32(100.0)
    ZNEAR = f32(0.1)
    FOVY = f32(math.pi / 2.0)
    UP = (0.0, 1.0, 0.0)

    def __init__(self, zoom, pitch, yaw, target, aspect):
        # Camera::new, src/camera.rs:93-107
        self.zoom = f32(zoom)
        self.pitch = f32(pitch)
        self.yaw = f32(yaw)
        self.eye = np.zeros(3, np.float32)
        self.target = _v3(target)
        self.up = _v3(self.UP)
        self.aspect = f32(aspect)
        self.updated = False
        self._fix_eye()

    @classmethod
    def new(cls, zoom, pitch, yaw, target, aspect) -> "Camera":
        return cls(zoom, pitch, yaw, target, aspect)

    def build_projection_view_matrix(self) -> np.ndarray:
        # src/camera.rs:109-113
        view = look_at_rh(self.eye, self.target, self.up)
        proj = perspective_rh(self.FOVY, self.aspect, self.ZNEAR, self.ZFAR)
        return mat4_mul(proj, view)

    def set_zoom(self, zoom):
        self.zoom = f32(min(max(f32(zoom), f32(0.3)), f32(self.ZFAR / f32(2.0))))
        self._fix_eye()
        self.updated = True

    def add_zoom(self, delta):
        self.set_zoom(f32(self.zoom + f32(delta)))

    def set_pitch(self, pitch):
        eps = np.finfo(np.float32).eps
        lo = f32(f32(-math.pi) / f32(2.0) + eps)
        hi = f32(f32(math.pi) / f32(2.0) - eps)
        self.pitch = f32(min(max(f32(pitch), lo), hi))
        self._fix_eye()
        self.updated = True

    def add_pitch(self, delta):
        self.set_pitch(f32(self.pitch + f32(delta)))

    def set_yaw(self, yaw):
        self.yaw = f32(yaw)
        self._fix_eye()
        self.updated = True

    def add_yaw(self, delta):
        self.set_yaw(f32(self.yaw + f32(delta)))

    def _fix_eye(self):
        # src/camera.rs:148-157
        pitch_cos = cosf(self.pitch)
        v = np.array([sinf(self.yaw) * pitch_cos, sinf(self.pitch), cosf(self.yaw) * pitch_cos], dtype=np.float32)
        self.eye = (self.target - self.zoom * v).astype(np.float32)

    def set_aspect(self, width: int, height: int):
        self.aspect = f32(f32(width) / f32(height))
        self.updated = True

    def get_proj_view_matrix(self) -> bytes:
        """144-byte CameraUniform (src/camera.rs:164-171)."""
        pv = self.build_projection_view_matrix()
        blob = np.concatenate(
            [np.array([self.eye[0], self.eye[1], self.eye[2], 1.0], np.float32), pv.reshape(16),
             mat4_inverse(pv).reshape(16)]
        ).astype(np.float32)
        return blob.tobytes()


CAMERA_UNIFORM_DEFAULT = np.concatenate(
    [np.zeros(4, np.float32), np.eye(4, dtype=np.float32).reshape(16), np.eye(4, dtype=np.float32).reshape(16)]
).tobytes()  # CameraUniform::default, src/camera.rs:13-21
