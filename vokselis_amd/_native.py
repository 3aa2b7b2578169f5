"""ctypes binding of the C-ABI in include/vokselis_hip.h (libvokselis_hip.so).

There is no CPU fallback: if the shared library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libvokselis_hip.so")

VK_OK = 0
FMT_R8_UNORM, FMT_R16_FLOAT, FMT_RGBA16F_PAIR = 0, 1, 2
MODE_NAIVE_TRILINEAR, MODE_COMPUTE_NEAREST, MODE_PROCEDURAL = 0, 1, 2
OUT_RGBA32F, OUT_RGBA16F = 0, 1
LAYOUT_AUTO, LAYOUT_LINEAR, LAYOUT_PACKED, LAYOUT_PACKED_PAIRS, LAYOUT_BRICKED, LAYOUT_QUADS, LAYOUT_STAGED = 0, 1, 2, 3, 4, 5, 6
RENDER_NO_SKIP, RENDER_COUNT, RENDER_SAFE, RENDER_FORCE_SKIP = 1, 2, 4, 8
RENDER_DEBUG_TRIPS = 16
RENDER_DEBUG_FALLBACK, RENDER_PROBE_ALWAYS, RENDER_FAST_WALK = 32, 64, 128
RENDER_PRESENT, RENDER_PRESENT_BGRA, RENDER_PRESENT_ONLY = 256, 512, 1024
RENDER_DEVICE_SINE = 2048
WIRE_RGBA, WIRE_RGB = 0, 1
MAX_FRAMES_IN_FLIGHT = 4
GEN_FOG, GEN_BONSAI_STANDIN, GEN_FOG_DENSE_CORE = 0, 1, 2

# every symbol include/vokselis_hip.h declares: name -> (restype, argtypes)
_u32, _i32, _f32, _vp, _sz = C.c_uint32, C.c_int32, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "vk_abi_version": (C.c_int, []),
    "vk_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "vk_ctx_destroy": (C.c_int, [_vp]),
    "vk_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "vk_ctx_sync": (C.c_int, [_vp]),
    "vk_device_info": (C.c_int, [_vp, C.c_char_p, _sz, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz)]),
    "vk_last_error": (C.c_char_p, [_vp]),
    "vk_volume_upload": (C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32, C.c_int, C.c_int]),
    "vk_volume_upload_device": (C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32, C.c_int, C.c_int]),
    "vk_volume_generate": (C.c_int, [_vp, C.c_int, _u32, _u32, _u32, C.c_int, _u32, _u32, _u32, C.c_int]),
    "vk_volume_generate_xor": (C.c_int, [_vp, _u32, _u32, _u32, _f32]),
    "vk_volume_empty_fraction": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "vk_volume_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz)]),
    "vk_set_uniform": (C.c_int, [_vp, _vp]),
    "vk_set_camera": (C.c_int, [_vp, _vp]),
    "vk_backbuffer_resize": (C.c_int, [_vp, _u32, _u32, C.c_int]),
    "vk_backbuffer_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_int), C.POINTER(_vp)]),
    "vk_backbuffer_clear": (C.c_int, [_vp]),
    "vk_ctx_frames_in_flight": (C.c_int, [_vp, _u32]),
    "vk_frame_begin": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "vk_frame_end": (C.c_int, [_vp]),
    "vk_frame_wait": (C.c_int, [_vp, C.c_uint64]),
    "vk_frame_readback": (C.c_int, [_vp, C.c_uint64, _vp, _sz]),
    "vk_frame_capture": (C.c_int, [_vp, C.c_uint64, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "vk_frame_info": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int)]),
    "vk_render": (C.c_int, [_vp, C.c_int, _i32, _i32, _u32, _u32, _f32, _u32]),
    "vk_partition_slots": (C.c_int, [_u32, _u32, _u32, _u32, C.POINTER(_u32)]),
    "vk_partition_slots_weighted": (C.c_int, [_u32, _u32, _u32, _u32, _u32, C.POINTER(_u32)]),
    "vk_partition_wire": (C.c_int, [_vp, C.c_int]),
    "vk_wire_pixel_bytes": (C.c_int, [_vp, C.POINTER(_u32)]),
    "vk_tiles_active": (C.c_int, [_vp, C.c_int, _u32, _u32, _u32, C.POINTER(C.c_ubyte), C.POINTER(_u32)]),
    "vk_partition_root_skip": (C.c_int, [_vp, _u32]),
    "vk_render_partition": (C.c_int, [_vp, C.c_int, _u32, _u32, _u32, _f32, _u32, _vp]),
    "vk_partition_order": (C.c_int, [_vp, C.c_int, _u32, C.POINTER(_u32), _u32]),
    "vk_partition_active": (C.c_int, [_vp, C.c_int, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "vk_untile": (C.c_int, [_vp, _vp, _u32, _u32, _u32]),
    "vk_render_batch": (C.c_int, [_vp, C.c_int, _u32, _vp, _u32, _u32, _u32, _f32, _u32, _vp, C.c_int, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "vk_untile_batch": (C.c_int, [_vp, _u32, _vp, _u32, _vp]),
    "vk_untile_batch_over": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _u32]),
    "vk_comm_available": (C.c_int, []),
    "vk_comm_unique_id": (C.c_int, [_vp]),
    "vk_comm_init_rank": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "vk_comm_destroy": (C.c_int, [_vp]),
    "vk_comm_abort": (C.c_int, [_vp]),
    "vk_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vk_gather_tiles": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, _vp]),
    "vk_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "vk_group_destroy": (C.c_int, [_vp]),
    "vk_group_size": (C.c_int, [_vp]),
    "vk_group_ctx": (_vp, [_vp, C.c_int]),
    "vk_group_render": (C.c_int, [_vp, C.c_int, _u32, _vp, _u32, _f32, _u32, _vp]),
    "vk_group_sync": (C.c_int, [_vp]),
    "vk_group_peer_direct": (C.c_int, [_vp, C.c_int]),
    "vk_group_last_error": (C.c_char_p, [_vp]),
    "vk_device_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "vk_device_free": (C.c_int, [_vp, _vp]),
    "vk_device_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "vk_present": (C.c_int, [_vp, _u32, _u32, C.c_int]),
    "vk_capture_frame": (C.c_int, [_vp, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "vk_readback": (C.c_int, [_vp, _vp, _sz]),
    "vk_step_counts": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vk_step_counts_reset": (C.c_int, [_vp]),
    "vk_simt_census": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "vk_debug_set_tile_order": (C.c_int, [_vp, C.POINTER(_u32), _u32]),
    "vk_debug_set_param": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "vk_debug_wave_trace": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64), _sz]),
    "vk_readback_steps": (C.c_int, [_vp, _vp]),
    "vk_timer_begin": (C.c_int, [_vp]),
    "vk_timer_end": (C.c_int, [_vp]),
    "vk_timer_elapsed_ms": (C.c_int, [_vp, C.POINTER(_f32)]),
    "vk_dispatch_optimal": (_u32, [_u32, _u32]),
}


class VokselisError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vokselis_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load libvokselis_hip.so; raise (never fall back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64; if this library
        # (linked against /opt/rocm's) is loaded first, a later `import torch` brings a second runtime into the
        # process and whichever initialises second reports "no ROCm-capable device".  Loaded after torch, the
        # dynamic loader resolves this library's libamdhip64.so.7 to the copy already in the process.  Processes
        # that never import torch (the C++ host, a plain C consumer) are unaffected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(ctx, rc: int):
    if rc != VK_OK:
        msg = lib().vk_last_error(ctx)
        raise VokselisError(rc, msg.decode() if msg else "")
