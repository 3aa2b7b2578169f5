"""Headless host mirror of the reference's `Context` / `Demo` / `run` surface over the C-ABI.

Reference: src/context.rs (Context), src/context/global_ubo.rs (Uniform),
src/context/hdr_backbuffer.rs (HdrBackBuffer), src/context/volume_texture.rs (VolumeTexture),
src/lib.rs:37-49 (`trait Demo`, `run`).  Windowing, input, hot reload and the present pass are out
of scope (no display on a compute node); the call order of the frame loop is kept:
`Context.update()` -> `Demo.update()` -> `Demo.render()` (src/lib.rs:75-79,178-181).
"""
from __future__ import annotations

import ctypes as C
import struct
import time
from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .camera import Camera


def dispatch_optimal(length: int, subgroup_size: int) -> int:
    """src/utils/mod.rs:15-18"""
    padded = (subgroup_size - length % subgroup_size) % subgroup_size
    return (length + padded) // subgroup_size


@dataclass
class ImageDimentions:
    """src/utils/mod.rs:91-118 (the reference's spelling)."""
    width: int
    height: int
    unpadded_bytes_per_row: int
    padded_bytes_per_row: int

    @classmethod
    def new(cls, width: int, height: int, align: int) -> "ImageDimentions":
        height = max(height - height % 2, 0)
        width = max(width - width % 2, 0)
        unpadded = width * 4
        row_padding = (align - unpadded % align) % align
        return cls(width, height, unpadded, unpadded + row_padding)

    def linear_size(self) -> int:
        return self.padded_bytes_per_row * self.height


@dataclass
class Uniform:
    """48-byte global uniform, src/context/global_ubo.rs:52-81."""
    pos: tuple = (0.0, 0.0, 0.0)
    frame: int = 0
    resolution: tuple = (1920.0, 780.0)
    mouse: tuple = (0.0, 0.0)
    mouse_pressed: int = 0
    time: float = 0.0
    time_delta: float = 1.0 / 60.0
    _padding: float = 0.0

    def to_bytes(self) -> bytes:
        b = struct.pack("<3fI2f2fI3f", *self.pos, self.frame, *self.resolution, *self.mouse, self.mouse_pressed,
                        self.time, self.time_delta, self._padding)
        assert len(b) == 48
        return b


@dataclass
class HdrBackBuffer:
    """src/context/hdr_backbuffer.rs:10-11 -- fixed 1280x720 rgba16float in the reference."""
    DEFAULT_RESOLUTION = (1280, 720)
    width: int = 1280
    height: int = 720
    format: int = N.OUT_RGBA16F


class FrameCounter:
    """src/utils/frame_counter.rs:1-40 (mean frame time)."""

    def __init__(self):
        self.frame_count = 0
        self.accum_time = 0.0
        self._last = time.perf_counter()
        self._delta = 1.0 / 60.0

    def time_delta(self) -> float:
        return self._delta

    def record(self) -> float:
        now = time.perf_counter()
        self._delta = now - self._last
        self._last = now
        self.accum_time += self._delta
        self.frame_count += 1
        return self._delta


class Context:
    """Device + per-frame uniform/camera upload + backbuffer (src/context.rs:38-67,225-249)."""

    def __init__(self, width: int = 1280, height: int = 720, camera: Camera | None = None, device: int = 0,
                 backbuffer: tuple | None = None, out_format: int = N.OUT_RGBA16F, stream=None):
        L = N.lib()
        self._h = C.c_void_p()
        rc = L.vk_ctx_create(device, C.byref(self._h))
        if rc != N.VK_OK:
            msg = L.vk_last_error(None)
            raise N.VokselisError(rc, msg.decode() if msg else "")
        self.stream_handle = None
        if stream is not None:
            N.check(self._h, L.vk_ctx_set_stream(self._h, C.c_void_p(stream)))
            self.stream_handle = stream
        self.width, self.height = width, height
        # Context::new: Camera::new(1., 0.5, 1., (0.,0.,0.), w/h) when none is given (src/context.rs:124-132)
        self.camera = camera if camera is not None else Camera(1.0, 0.5, 1.0, (0.0, 0.0, 0.0), width / height)
        self.camera_epoch = 0  # bumped whenever a camera blob is uploaded (per-camera caches key on it)
        self.camera_blob = None  # the 144 bytes last uploaded
        self.global_uniform = Uniform()
        bw, bh = backbuffer if backbuffer is not None else HdrBackBuffer.DEFAULT_RESOLUTION
        self.render_backbuffer = HdrBackBuffer(bw, bh, out_format)
        N.check(self._h, L.vk_backbuffer_resize(self._h, bw, bh, out_format))
        self._timeline = time.perf_counter()
        self._first_frame = True
        self.in_flight = 1
        # fuse_present: when the window has the backbuffer's size, RaycastPipeline.record presents from the pass's own epilogue
        # (VK_RENDER_PRESENT) and the render() that follows records nothing -- demo.render + context.render (src/lib.rs:178-182) in one launch
        self.fuse_present = False
        self._pass_presented = False

    # -- lifetime
    def close(self):
        if self._h:
            N.lib().vk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def get_info(self) -> dict:
        """Context::get_info, src/context.rs:183-203."""
        name = C.create_string_buffer(256)
        cus, is950, mem = C.c_int(), C.c_int(), C.c_size_t()
        N.check(self._h, N.lib().vk_device_info(self._h, name, 256, C.byref(cus), C.byref(is950), C.byref(mem)))
        return {"device_name": name.value.decode(), "compute_units": cus.value, "gfx950": bool(is950.value),
                "total_mem_bytes": mem.value, "backend": "HIP"}

    # -- per frame
    def update(self, frame_counter: FrameCounter | None = None):
        """Context::update, src/context.rs:225-236.  The camera is uploaded unconditionally on the
        first frame (the reference's `updated` gate leaves frame 0 with an identity camera, F10)."""
        gu = self.global_uniform
        gu.time = time.perf_counter() - self._timeline
        if frame_counter is not None:
            gu.time_delta = frame_counter.time_delta()
            gu.frame = frame_counter.frame_count
        gu.resolution = (float(self.width), float(self.height))
        N.check(self._h, N.lib().vk_set_uniform(self._h, gu.to_bytes()))
        if self.camera.updated or self._first_frame:
            self.camera_blob = self.camera.get_proj_view_matrix()
            N.check(self._h, N.lib().vk_set_camera(self._h, self.camera_blob))
            self.camera.updated = False
            self._first_frame = False
            self.camera_epoch += 1

    def resize(self, width: int, height: int):
        """Context::resize, src/context.rs:238-249: the *window* size drives the camera aspect; the
        backbuffer keeps its own resolution (F6)."""
        self.width, self.height = width, height
        self.camera.set_aspect(width, height)

    def resize_backbuffer(self, width: int, height: int, out_format: int | None = None):
        fmt = self.render_backbuffer.format if out_format is None else out_format
        N.check(self._h, N.lib().vk_backbuffer_resize(self._h, width, height, fmt))
        self.render_backbuffer = HdrBackBuffer(width, height, fmt)

    def set_stream(self, hip_stream: int | None):
        """Run this context's work on a caller-owned HIP stream (None: the context's own)."""
        N.check(self._h, N.lib().vk_ctx_set_stream(self._h, C.c_void_p(hip_stream)))
        self.stream_handle = hip_stream

    def set_camera_blob(self, blob: bytes):
        if len(blob) != 144:  # the C side reads exactly one CameraUniform (src/camera.rs:5-11)
            raise ValueError("a camera blob is 144 bytes, got %d" % len(blob))
        N.check(self._h, N.lib().vk_set_camera(self._h, bytes(blob)))
        self.camera_blob = bytes(blob)
        self._first_frame = False
        self.camera_epoch += 1

    def sync(self):
        N.check(self._h, N.lib().vk_ctx_sync(self._h))

    # -- frames in flight: the queue running ahead of the GPU (src/lib.rs:178-194), bounded by the swapchain (src/context.rs:118,252)
    def frames_in_flight(self, k: int):
        """vk_ctx_frames_in_flight: a ring of k frame surfaces, each on its own stream (1: one surface, the default)."""
        N.check(self._h, N.lib().vk_ctx_frames_in_flight(self._h, int(k)))
        self.in_flight = int(k)

    def frame_begin(self) -> int:
        """Surface::get_current_texture (src/context.rs:252): the next surface of the ring (blocks while the frame that used it k
        frames ago is still running); the render / present / read calls that follow address it.  Returns the frame's id."""
        fid = C.c_uint64()
        N.check(self._h, N.lib().vk_frame_begin(self._h, C.byref(fid)))
        return fid.value

    def frame_end(self):
        """queue.submit + frame.present() (src/context.rs:294-296)."""
        N.check(self._h, N.lib().vk_frame_end(self._h))

    def frame_wait(self, frame_id: int):
        N.check(self._h, N.lib().vk_frame_wait(self._h, frame_id))

    def read_frame(self, frame_id: int) -> np.ndarray:
        """vk_frame_readback: that frame's backbuffer (waits for that frame only), while its surface still holds it."""
        bb = self.render_backbuffer
        out = np.empty((bb.height, bb.width, 4), np.float32 if bb.format == N.OUT_RGBA32F else np.float16)
        N.check(self._h, N.lib().vk_frame_readback(self._h, frame_id, out.ctypes.data, out.strides[0]))
        return out

    def capture_frame_of(self, frame_id: int):
        """vk_frame_capture: capture_frame of that frame's presented Rgba8 image."""
        dims = ImageDimentions.new(self.width, self.height, 256)
        buf = np.zeros(dims.linear_size(), np.uint8)
        N.check(self._h, N.lib().vk_frame_capture(self._h, frame_id, buf.ctypes.data, buf.size, None, None, None))
        return buf.tobytes(), dims

    def frame_info(self, frame_id: int) -> dict:
        bb, r8, done = C.c_void_p(), C.c_void_p(), C.c_int()
        N.check(self._h, N.lib().vk_frame_info(self._h, frame_id, C.byref(bb), C.byref(r8), C.byref(done)))
        return {"backbuffer": bb.value, "rgba8": r8.value, "complete": bool(done.value)}

    def set_root_skip(self, k: int):
        """vk_partition_root_skip: rank 0 sits out every k-th round of the tile deal (0: never)."""
        N.check(self._h, N.lib().vk_partition_root_skip(self._h, int(k)))

    def set_wire(self, wire: int):
        """vk_partition_wire: what a pixel of this context's compact tiles holds (WIRE_RGBA: the backbuffer's pixel; WIRE_RGB:
        its three colour channels -- alpha is 1 in every pixel the path writes -- a quarter less to move between GPUs)."""
        N.check(self._h, N.lib().vk_partition_wire(self._h, int(wire)))

    @property
    def wire_pixel_bytes(self) -> int:
        n = C.c_uint32(0)
        N.check(self._h, N.lib().vk_wire_pixel_bytes(self._h, C.byref(n)))
        return int(n.value)

    def set_param(self, name: str, value: float):
        """Debug / tuning knob of the library (vk_debug_set_param)."""
        N.check(self._h, N.lib().vk_debug_set_param(self._h, name.encode(), float(value)))

    # -- results
    def read_backbuffer(self) -> np.ndarray:
        """The backbuffer as [H, W, 4] float32 (RGBA32F) or float16 (RGBA16F)."""
        bb = self.render_backbuffer
        dt = np.float32 if bb.format == N.OUT_RGBA32F else np.float16
        out = np.empty((bb.height, bb.width, 4), dt)
        N.check(self._h, N.lib().vk_readback(self._h, out.ctypes.data, out.strides[0]))
        return out

    def render(self):
        """Context::render (src/context.rs:251-297): the present pass -- backbuffer -> ACES + sRGB ->
        Rgba8 at the window size.  (No surface to present to on a compute node.)"""
        if self._pass_presented:  # the raycast pass has written the presented image itself (fuse_present)
            self._pass_presented = False
            return
        N.check(self._h, N.lib().vk_present(self._h, self.width, self.height, 0))

    def capture_frame(self):
        """Context::capture_frame (src/context.rs:299-302, screenshot.rs:37-77): the presented Rgba8
        frame as padded rows + its ImageDimentions."""
        w, h, pitch = C.c_uint32(), C.c_uint32(), C.c_uint32()
        N.check(self._h, N.lib().vk_capture_frame(self._h, None, 0, C.byref(w), C.byref(h), C.byref(pitch)))
        dims = ImageDimentions.new(self.width, self.height, 256)
        assert (dims.width, dims.height, dims.padded_bytes_per_row) == (w.value, h.value, pitch.value)
        buf = np.zeros(dims.linear_size(), np.uint8)
        N.check(self._h, N.lib().vk_capture_frame(self._h, buf.ctypes.data, buf.size, None, None, None))
        return buf.tobytes(), dims

    def partition_active(self, tile_size: int, nranks: int = 1, mode: int = N.MODE_NAIVE_TRILINEAR):
        """(active tiles, active slots per rank): only those leading positions of the order are marched /
        gathered; the rest of the frame is clear colour."""
        a, b = C.c_uint32(), C.c_uint32()
        N.check(self._h, N.lib().vk_partition_active(self._h, mode, tile_size, nranks, C.byref(a), C.byref(b)))
        return a.value, b.value

    def partition_order(self, tile_size: int, mode: int = N.MODE_NAIVE_TRILINEAR) -> np.ndarray:
        """Heaviest-first tile order of the frame partition (position -> row-major tile id)."""
        bb = self.render_backbuffer
        n = ((bb.width + tile_size - 1) // tile_size) * ((bb.height + tile_size - 1) // tile_size)
        out = np.empty(n, np.uint32)
        N.check(self._h, N.lib().vk_partition_order(self._h, mode, tile_size, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def step_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        N.check(self._h, N.lib().vk_step_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def simt_census(self) -> dict:
        out = (C.c_uint64 * 4)()
        N.check(self._h, N.lib().vk_simt_census(self._h, out))
        return {"wave_loop_iters": out[0], "wave_skip_iters": out[1], "wave_sample_execs": out[2], "lane_loop_iters": out[3]}

    def reset_step_counts(self):
        N.check(self._h, N.lib().vk_step_counts_reset(self._h))

    def read_steps(self) -> np.ndarray:
        bb = self.render_backbuffer
        out = np.empty((bb.height, bb.width), np.uint32)
        N.check(self._h, N.lib().vk_readback_steps(self._h, out.ctypes.data))
        return out

    def timer_begin(self):
        N.check(self._h, N.lib().vk_timer_begin(self._h))

    def timer_end(self):
        N.check(self._h, N.lib().vk_timer_end(self._h))

    def timer_elapsed_ms(self) -> float:
        ms = C.c_float()
        N.check(self._h, N.lib().vk_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value


class VolumeTexture:
    """src/context/volume_texture.rs:32-89: uploads the dense x-fastest volume ([nz,ny,nx])."""

    def __init__(self, ctx: Context, data: np.ndarray, data2: np.ndarray | None = None, layout: int = N.LAYOUT_AUTO):
        vol = np.ascontiguousarray(data)
        if vol.dtype == np.uint8 and vol.ndim == 3:
            fmt = N.FMT_R8_UNORM
        elif vol.dtype in (np.float16, np.uint16) and vol.ndim == 3:
            fmt = N.FMT_R16_FLOAT
        elif vol.dtype in (np.float16, np.uint16) and vol.ndim == 4 and vol.shape[3] == 4 and data2 is not None:
            fmt = N.FMT_RGBA16F_PAIR
        else:
            raise ValueError("volume must be u8[nz,ny,nx], f16[nz,ny,nx] or a pair of f16[nz,ny,nx,4]")
        nz, ny, nx = vol.shape[:3]
        v2 = None
        if data2 is not None:
            v2 = np.ascontiguousarray(data2)
            if v2.shape != vol.shape or v2.dtype.itemsize != 2:
                raise ValueError("normals volume must match the density volume")
        N.check(ctx.handle, N.lib().vk_volume_upload(ctx.handle, vol.ctypes.data, v2.ctypes.data if v2 is not None else None,
                                                    nx, ny, nz, fmt, layout))
        self.dims = (nx, ny, nz)
        self.format = fmt

    @classmethod
    def from_raw(cls, ctx: Context, path: str, dims=(256, 256, 256), layout: int = N.LAYOUT_AUTO) -> "VolumeTexture":
        """Drop-in for the reference's `bonsai_256x256x256_uint8.raw` (absent from the checkout, F3)."""
        nx, ny, nz = dims
        raw = np.fromfile(path, dtype=np.uint8)
        if raw.size != nx * ny * nz:
            raise ValueError(f"{path}: expected {nx * ny * nz} bytes, found {raw.size}")
        return cls(ctx, raw.reshape(nz, ny, nx), layout=layout)

    @classmethod
    def generate(cls, ctx: Context, kind: int, dims, fmt=N.FMT_R8_UNORM, seed=0x5EED0001, lo=20, span=12,
                 layout: int = N.LAYOUT_AUTO) -> "VolumeTexture":
        """Deterministic synthetic volume made on the device (GEN_FOG / GEN_BONSAI_STANDIN)."""
        nx, ny, nz = dims
        N.check(ctx.handle, N.lib().vk_volume_generate(ctx.handle, kind, nx, ny, nz, fmt, seed, lo, span, layout))
        self = cls.__new__(cls)
        self.dims, self.format = (nx, ny, nz), fmt
        return self

    @classmethod
    def generate_fog(cls, ctx: Context, dims, fmt=N.FMT_R8_UNORM, seed=0x5EED0002, lo=20, span=12,
                     layout: int = N.LAYOUT_AUTO, dense_core: bool = False) -> "VolumeTexture":
        """Fog (C2-fog, C4, C5); dense_core: with the dense ball at the centre (the configs' early-out variant)."""
        return cls.generate(ctx, N.GEN_FOG_DENSE_CORE if dense_core else N.GEN_FOG, dims, fmt, seed, lo, span, layout)

    @classmethod
    def generate_xor(cls, ctx: Context, dims=(256, 256, 256), time: float = 0.0) -> "VolumeTexture":
        """XorCompute (examples/xor/xor_compute.rs): shaders/xor.wgsl cs_main -> density + normals rgba16f."""
        nx, ny, nz = dims
        N.check(ctx.handle, N.lib().vk_volume_generate_xor(ctx.handle, nx, ny, nz, time))
        self = cls.__new__(cls)
        self.dims, self.format = (nx, ny, nz), N.FMT_RGBA16F_PAIR
        return self

    @classmethod
    def generate_standin(cls, ctx: Context, dims=(256, 256, 256), seed=0x5EED0001, layout: int = N.LAYOUT_AUTO) -> "VolumeTexture":
        return cls.generate(ctx, N.GEN_BONSAI_STANDIN, dims, N.FMT_R8_UNORM, seed, 0, 1, layout)


class RaycastPipeline:
    """examples/bonsai/raycast.rs (render pipeline) and examples/xor/raycast.rs (compute
    `single` / `tile`): records one raycast pass into the context's backbuffer."""

    def __init__(self, mode: int = N.MODE_NAIVE_TRILINEAR, dt_scale: float = 1.0, flags: int = 0):
        self.mode, self.dt_scale, self.flags = mode, dt_scale, flags

    def record(self, ctx: Context, tile=None):
        bb = ctx.render_backbuffer
        tx, ty, tw, th = (0, 0, bb.width, bb.height) if tile is None else tile
        flags = self.flags
        if ctx.fuse_present and (ctx.width, ctx.height) == (bb.width, bb.height) and not (flags & N.RENDER_COUNT):
            flags |= N.RENDER_PRESENT
            ctx._pass_presented = True
        N.check(ctx.handle, N.lib().vk_render(ctx.handle, self.mode, tx, ty, tw, th, self.dt_scale, flags))

    def record_partition(self, ctx: Context, tile_size: int, rank: int, nranks: int, compact_ptr: int):
        """March this rank's tiles of the current camera's frame into a compact buffer (several frames: render_batch)."""
        N.check(ctx.handle, N.lib().vk_render_partition(ctx.handle, self.mode, tile_size, rank, nranks, self.dt_scale,
                                                       self.flags, C.c_void_p(compact_ptr)))


def render_batch(ctx: Context, pipe: RaycastPipeline, cameras, out_ptr: int, *, tile_size: int = 64, rank: int = 0, nranks: int = 1,
                 compact: bool = False, slot_capacity: int = 0):
    """vk_render_batch: len(cameras) frames (144-byte blobs) in one launch.  Returns (batch_id, active slots per rank)."""
    if not cameras or any(len(c) != 144 for c in cameras):
        raise ValueError("render_batch: every camera is one 144-byte CameraUniform blob")
    blob = b"".join(cameras)
    n = len(blob) // 144
    bid, act = C.c_uint32(), C.c_uint32()
    N.check(ctx.handle, N.lib().vk_render_batch(ctx.handle, pipe.mode, n, blob, tile_size, rank, nranks, pipe.dt_scale, pipe.flags,
                                               C.c_void_p(out_ptr), 1 if compact else 0, slot_capacity, C.byref(bid), C.byref(act)))
    return bid.value, act.value


def untile_batch(ctx: Context, batch_id: int, gathered_ptr: int, n_slots: int, out_ptr: int, prev_batch_id: int = 0):
    """vk_untile_batch / vk_untile_batch_over: `prev_batch_id` names the batch whose un-tiled frames `out_ptr` still holds,
    untouched (0: unknown -- every inactive tile is cleared)."""
    N.check(ctx.handle, N.lib().vk_untile_batch_over(ctx.handle, batch_id, C.c_void_p(gathered_ptr), n_slots, C.c_void_p(out_ptr), int(prev_batch_id)))


def partition_slots(width: int, height: int, tile_size: int, nranks: int, root_skip: int = 0) -> int:
    n = C.c_uint32()
    rc = N.lib().vk_partition_slots_weighted(width, height, tile_size, nranks, root_skip, C.byref(n))
    if rc != N.VK_OK:
        raise N.VokselisError(rc, "vk_partition_slots: bad arguments")
    return n.value


class Demo:
    """`trait Demo` (src/lib.rs:37-43): init / update / render; resize and update_input are
    window-side and therefore absent."""

    @classmethod
    def init(cls, ctx: Context) -> "Demo":
        return cls()

    def update(self, ctx: Context):
        pass

    def render(self, ctx: Context):
        pass


def run_headless(demo_cls, frames: int = 1, camera: Camera | None = None, width: int = 1280, height: int = 720,
                 backbuffer: tuple | None = None, out_format: int = N.OUT_RGBA16F, device: int = 0, in_flight: int = 1, on_frame=None,
                 fuse_present: bool = False):
    """`run::<D>` (src/lib.rs:45-208) without the window: N frames of
    Context.update -> Demo.update -> Demo.render, then returns (ctx, demo).
    in_flight > 1: the loop runs up to that many frames ahead of the GPU, as the reference's queue does (src/lib.rs:178-194), every
    frame on a surface of its own; on_frame(ctx, frame_id), if given, is called after each frame has been submitted.
    fuse_present: the present pass rides in the raycast pass's epilogue when the window has the backbuffer's size (Context.fuse_present)."""
    ctx = Context(width, height, camera, device=device, backbuffer=backbuffer, out_format=out_format)
    if in_flight > 1:
        ctx.frames_in_flight(in_flight)
    ctx.fuse_present = bool(fuse_present)
    fc = FrameCounter()
    demo = demo_cls.init(ctx)
    for _ in range(frames):
        ctx.update(fc)
        demo.update(ctx)
        fc.record()
        fid = ctx.frame_begin()
        demo.render(ctx)
        ctx.render()  # src/lib.rs:178-182: demo.render, then context.render (present pass)
        ctx.frame_end()
        if on_frame is not None:
            on_frame(ctx, fid)
    ctx.sync()
    return ctx, demo
