"""vokselis_amd -- MI355X-native (gfx950) volume raymarcher behind the surface of pudnax/vokselis.

Only the raycast hot path and the host surface that drives it live here:
  csrc/      hand-written HIP kernels + the C-ABI (include/vokselis_hip.h)
  _native    ctypes binding of that C-ABI (fails loudly when the .so is missing)
  camera     Camera / CameraUniform            (src/camera.rs)
  context    Context, Uniform, HdrBackBuffer, VolumeTexture, RaycastPipeline, Demo, run_headless
  volumes    deterministic synthetic volumes (bonsai stand-in, fog)
  dist       tile-parallel multi-GPU frame (one process per GPU, RCCL gather over xGMI)
"""
from . import _native as native
from . import volumes
from ._native import (FMT_R8_UNORM, FMT_R16_FLOAT, FMT_RGBA16F_PAIR, LAYOUT_AUTO, LAYOUT_LINEAR, LAYOUT_PACKED, LAYOUT_PACKED_PAIRS, LAYOUT_BRICKED, LAYOUT_QUADS, LAYOUT_STAGED, RENDER_SAFE, RENDER_FORCE_SKIP,
                      GEN_BONSAI_STANDIN, GEN_FOG, GEN_FOG_DENSE_CORE, MODE_COMPUTE_NEAREST, MODE_NAIVE_TRILINEAR, MODE_PROCEDURAL, OUT_RGBA16F, OUT_RGBA32F, RENDER_COUNT,
                      RENDER_NO_SKIP, RENDER_PROBE_ALWAYS, RENDER_FAST_WALK, RENDER_PRESENT, RENDER_PRESENT_BGRA, RENDER_PRESENT_ONLY, RENDER_DEVICE_SINE, WIRE_RGB, WIRE_RGBA, VokselisError)
from .camera import Camera
from .context import (Context, Demo, FrameCounter, HdrBackBuffer, ImageDimentions, RaycastPipeline, Uniform,
                      VolumeTexture, dispatch_optimal, partition_slots, render_batch, run_headless, untile_batch)

__all__ = [
    "native", "volumes", "GEN_BONSAI_STANDIN", "GEN_FOG", "GEN_FOG_DENSE_CORE", "Camera", "Context", "Demo", "FrameCounter", "HdrBackBuffer", "ImageDimentions", "RaycastPipeline",
    "Uniform", "VolumeTexture", "dispatch_optimal", "partition_slots", "render_batch", "untile_batch", "run_headless", "VokselisError",
    "FMT_R8_UNORM", "FMT_R16_FLOAT", "FMT_RGBA16F_PAIR", "LAYOUT_AUTO", "LAYOUT_LINEAR", "LAYOUT_PACKED", "LAYOUT_PACKED_PAIRS", "LAYOUT_BRICKED", "LAYOUT_QUADS", "LAYOUT_STAGED", "RENDER_SAFE", "RENDER_FORCE_SKIP",
    "MODE_COMPUTE_NEAREST", "MODE_NAIVE_TRILINEAR", "MODE_PROCEDURAL", "OUT_RGBA16F", "OUT_RGBA32F", "RENDER_COUNT", "RENDER_NO_SKIP", "RENDER_PROBE_ALWAYS", "RENDER_FAST_WALK", "RENDER_PRESENT", "RENDER_PRESENT_BGRA", "RENDER_PRESENT_ONLY", "RENDER_DEVICE_SINE", "WIRE_RGB", "WIRE_RGBA",
]
