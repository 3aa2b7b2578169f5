"""ctypes binding of oracle/vokselis_oracle.c (the C restatement of the reference WGSL).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/vokselis_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvokselis_oracle.so")

FMT_R8_UNORM, FMT_R16_FLOAT, FMT_RGBA16F_PAIR = 0, 1, 2
MODE_NAIVE_TRILINEAR, MODE_COMPUTE_NEAREST, MODE_PROCEDURAL = 0, 1, 2
FLAG_NO_EARLY_OUT, FLAG_TAPNORM_PER_TAP = 1, 2
FLAG_LITERAL_WGSL, FLAG_TRANSFER_R1 = 8, 16  # yardsticks: the shader's text as written / the round-1 text of the transfer function


class CameraUniform(C.Structure):
    _fields_ = [("view_position", C.c_float * 4), ("proj_view", C.c_float * 16), ("inv_proj", C.c_float * 16)]


class RenderArgs(C.Structure):
    _fields_ = [
        ("camera", C.POINTER(CameraUniform)),
        ("volume", C.c_void_p),
        ("volume2", C.c_void_p),
        ("nx", C.c_uint32),
        ("ny", C.c_uint32),
        ("nz", C.c_uint32),
        ("format", C.c_int),
        ("mode", C.c_int),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("tile_x", C.c_int32),
        ("tile_y", C.c_int32),
        ("tile_w", C.c_uint32),
        ("tile_h", C.c_uint32),
        ("dt_scale", C.c_float),
        ("flags", C.c_int),
        ("threads", C.c_int),
        ("out_rgba", C.c_void_p),
        ("out_steps", C.c_void_p),
        ("out_sampled", C.c_void_p),
        ("proc_time", C.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    import hashlib

    h = hashlib.sha256()
    for name in ("vokselis_oracle.c", "vokselis_oracle.h", "Makefile"):
        with open(os.path.join(_HERE, name), "rb") as f:
            h.update(f.read())
    key, stamp = h.hexdigest()[:16], _SO + ".stamp"
    # keyed on the sources' content (mtimes do not survive a sync to another box)
    if force or not os.path.exists(_SO) or not os.path.exists(stamp) or open(stamp).read().strip() != key:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(key + "\n")
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.vo_render.argtypes = [C.POINTER(RenderArgs)]
        L.vo_render.restype = C.c_int
        L.vo_camera_uniform_build.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_float,
                                              C.POINTER(CameraUniform)]
        L.vo_camera_uniform_build.restype = None
        L.vo_camera_eye.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vo_intersect_box.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                       C.POINTER(C.c_float)]
        L.vo_sample_trilinear.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                          C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        L.vo_sample_trilinear.restype = C.c_float
        L.vo_transfer_alpha.argtypes = [C.c_float, C.c_int]
        L.vo_transfer_alpha.restype = C.c_float
        L.vo_transfer_alpha_r1.argtypes = [C.c_float, C.c_int]
        L.vo_transfer_alpha_r1.restype = C.c_float
        L.vo_transfer_alpha_literal.argtypes = [C.c_float]
        L.vo_transfer_alpha_literal.restype = C.c_float
        L.vo_vertigo.argtypes = [C.c_float, C.POINTER(C.c_float)]
        L.vo_linear_to_srgb.argtypes = [C.c_float]
        L.vo_linear_to_srgb.restype = C.c_float
        L.vo_f32_to_f16.argtypes = [C.c_float]
        L.vo_f32_to_f16.restype = C.c_uint16
        L.vo_f16_to_f32.argtypes = [C.c_uint16]
        L.vo_f16_to_f32.restype = C.c_float
        L.vo_rgba32f_to_rgba16f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vo_volume_standin_u8.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
        L.vo_volume_fog_u8.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.vo_volume_fog_f16.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
        L.vo_volume_fog_core_u8.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.vo_volume_fog_core_f16.argtypes = [C.c_uint32] * 4 + [C.c_int, C.c_void_p]
        L.vo_dispatch_optimal.argtypes = [C.c_uint32, C.c_uint32]
        L.vo_dispatch_optimal.restype = C.c_uint32
        L.vo_image_dimentions.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.vo_volume_xor.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
        L.vo_sin_spec.argtypes = [C.c_float]
        L.vo_sin_spec.restype = C.c_float
        L.vo_present.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.vo_ray_naive.argtypes = [C.POINTER(CameraUniform), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vo_ray_compute.argtypes = [C.POINTER(CameraUniform), C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


# --------------------------------------------------------------------------------------------


def camera_blob(zoom, pitch, yaw, target, aspect) -> bytes:
    """144-byte CameraUniform for Camera::new(zoom, pitch, yaw, target, aspect) (src/camera.rs:93-171)."""
    cu = CameraUniform()
    tgt = (C.c_float * 3)(*[float(t) for t in target])
    lib().vo_camera_uniform_build(zoom, pitch, yaw, tgt, aspect, C.byref(cu))
    return bytes(cu)


def camera_from_blob(blob: bytes) -> CameraUniform:
    assert len(blob) == 144
    return CameraUniform.from_buffer_copy(blob)


def render_procedural(camera: bytes, width: int, height: int, *, dt_scale: float = 1.0, time: float = 0.0, tile=None,
                      flags: int = 0, threads: int = 0):
    """C3: the compute twin's march over the xor example's density function, no volume (pixel_procedural)."""
    cu = camera_from_blob(camera)
    rgba = np.zeros((height, width, 4), np.float32)
    steps = np.zeros((height, width), np.uint32)
    a = RenderArgs()
    a.camera = C.pointer(cu)
    a.mode = MODE_PROCEDURAL
    a.width, a.height = width, height
    tx, ty, tw, th = (0, 0, width, height) if tile is None else tile
    a.tile_x, a.tile_y, a.tile_w, a.tile_h = tx, ty, tw, th
    a.dt_scale, a.flags, a.threads, a.proc_time = dt_scale, flags, threads, time
    a.out_rgba, a.out_steps = rgba.ctypes.data, steps.ctypes.data
    rc = lib().vo_render(C.byref(a))
    if rc != 0:
        raise RuntimeError(f"vo_render failed: {rc}")
    return rgba, steps


def render(camera: bytes, volume: np.ndarray, width: int, height: int, *, dt_scale: float = 1.0,
           mode: int = MODE_NAIVE_TRILINEAR, fmt: int | None = None, volume2: np.ndarray | None = None,
           tile=None, flags: int = 0, threads: int = 0, want_counts: bool = True, out: np.ndarray | None = None):
    """Render a frame (or a tile of it) with the C oracle.

    volume: u8 [nz,ny,nx] (R8Unorm), u16/f16 [nz,ny,nx] (R16Float) or f16 [nz,ny,nx,4] (+volume2).
    Returns (rgba f32 [H,W,4], steps u32 [H,W], sampled u32 [H,W]).  Pixels outside `tile`
    keep the contents of `out` (zeros when `out` is None).
    """
    cu = camera_from_blob(camera)
    vol = np.ascontiguousarray(volume)
    if fmt is None:
        if vol.dtype == np.uint8:
            fmt = FMT_R8_UNORM
        elif vol.ndim == 4:
            fmt = FMT_RGBA16F_PAIR
        else:
            fmt = FMT_R16_FLOAT
    if fmt != FMT_R8_UNORM:
        vol = vol.view(np.uint16)
    nz, ny, nx = vol.shape[:3]
    rgba = np.zeros((height, width, 4), np.float32) if out is None else out
    assert rgba.dtype == np.float32 and rgba.shape == (height, width, 4) and rgba.flags.c_contiguous
    steps = np.zeros((height, width), np.uint32)
    sampled = np.zeros((height, width), np.uint32)
    a = RenderArgs()
    a.camera = C.pointer(cu)
    a.volume = vol.ctypes.data
    v2 = None
    if volume2 is not None:
        v2 = np.ascontiguousarray(volume2).view(np.uint16)
        a.volume2 = v2.ctypes.data
    a.nx, a.ny, a.nz = nx, ny, nz
    a.format, a.mode = fmt, mode
    a.width, a.height = width, height
    tx, ty, tw, th = (0, 0, width, height) if tile is None else tile
    a.tile_x, a.tile_y, a.tile_w, a.tile_h = tx, ty, tw, th
    a.dt_scale = dt_scale
    a.flags = flags
    a.threads = threads
    a.out_rgba = rgba.ctypes.data
    a.out_steps = steps.ctypes.data if want_counts else None
    a.out_sampled = sampled.ctypes.data if want_counts else None
    rc = lib().vo_render(C.byref(a))
    if rc != 0:
        raise RuntimeError(f"vo_render failed: {rc}")
    return rgba, steps, sampled


def rgba32f_to_rgba16f(rgba: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(rgba, np.float32)
    dst = np.empty(src.shape, np.uint16)
    lib().vo_rgba32f_to_rgba16f(src.ctypes.data, dst.ctypes.data, src.size)
    return dst


def volume_standin_u8(n, seed=0x5EED0001) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    out = np.empty((nz, ny, nx), np.uint8)
    lib().vo_volume_standin_u8(nx, ny, nz, seed, out.ctypes.data)
    return out


def volume_fog_u8(n, seed=0x5EED0002, lo=20, span=12, dense_core=False) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    out = np.empty((nz, ny, nx), np.uint8)
    lib().vo_volume_fog_core_u8(nx, ny, nz, seed, lo, span, 1 if dense_core else 0, out.ctypes.data)
    return out


def volume_fog_f16(n, seed=0x5EED0004, dense_core=False) -> np.ndarray:
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    out = np.empty((nz, ny, nx), np.uint16)
    lib().vo_volume_fog_core_f16(nx, ny, nz, seed, 1 if dense_core else 0, out.ctypes.data)
    return out.view(np.float16)


def present(backbuffer: np.ndarray, width: int, height: int) -> np.ndarray:
    """Present pass (present.wgsl): backbuffer [bh,bw,4] f32 -> [height,width,4] u8 RGBA."""
    bb = np.ascontiguousarray(backbuffer, np.float32)
    out = np.empty((height, width, 4), np.uint8)
    lib().vo_present(bb.ctypes.data, bb.shape[1], bb.shape[0], width, height, out.ctypes.data)
    return out


def volume_xor(n, time: float = 0.0):
    """xor.wgsl cs_main: (density, normals) as float16 [nz,ny,nx,4]."""
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    den = np.empty((nz, ny, nx, 4), np.uint16)
    nrm = np.empty((nz, ny, nx, 4), np.uint16)
    lib().vo_volume_xor(nx, ny, nz, time, den.ctypes.data, nrm.ctypes.data)
    return den.view(np.float16), nrm.view(np.float16)
