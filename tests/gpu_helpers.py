"""Shared by the GPU test modules: the `V` fixture (the package, on a box that has an MI355X) and the render helpers."""
import ctypes as C

import numpy as np
import pytest

TOL = 1e-4


@pytest.fixture(scope="module")
def V(hip_built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU: the gpu suite must run on an MI355X box")
    import vokselis_amd

    return vokselis_amd


def gpu_render(V, cam_blob, vol, W, H, *, dt=1.0, layout=None, flags=0, out=None, tile=None, vol2=None, mode=None,
               want_steps=True):
    out = V.OUT_RGBA32F if out is None else out
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=out)
    try:
        V.VolumeTexture(ctx, vol, vol2, layout=V.LAYOUT_AUTO if layout is None else layout)
        ctx.set_camera_blob(cam_blob)
        ctx.reset_step_counts()
        if not (flags & V.RENDER_NO_SKIP):
            # exercise the skip path whatever the volume's empty share, probing on every trip so that S_sampled is exactly
            # the count of steps that can contribute (the adaptive policy has its own test)
            flags |= V.RENDER_FORCE_SKIP | V.RENDER_PROBE_ALWAYS
        pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR if mode is None else mode, dt_scale=dt,
                                 flags=flags | (V.RENDER_COUNT if want_steps else 0))
        pipe.record(ctx, tile)
        img = ctx.read_backbuffer()
        steps = ctx.read_steps() if want_steps else None
        counts = ctx.step_counts() if want_steps else None
        if want_steps:
            # the production (uninstrumented) kernel must reproduce the instrumented frame bit for bit
            V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
            V.RaycastPipeline(pipe.mode, dt_scale=dt, flags=flags).record(ctx, tile)
            again = ctx.read_backbuffer()
            if tile is None:
                assert (again.view(np.uint8) == img.view(np.uint8)).all(), "production path differs from the instrumented one"
        return img, steps, counts
    finally:
        ctx.close()


def _synced(t):
    """A tensor torch has just filled on ITS current stream, handed to the library, which writes on a non-blocking stream of
    its own: without a synchronisation nothing orders the fill before the library's kernels (torch's streams and the
    context's do not synchronise with the legacy default stream)."""
    import torch

    torch.cuda.synchronize()
    return t


def layouts(V):
    return {"P8": V.LAYOUT_PACKED, "P16": V.LAYOUT_PACKED_PAIRS, "LIN": V.LAYOUT_LINEAR, "B9": V.LAYOUT_BRICKED, "Q": V.LAYOUT_QUADS,
            "S8": V.LAYOUT_STAGED}


def _holes_volume(n, p_empty, seed=3, block=16):
    """u8 fog 26..40 (every cell contributes) with 16^3 blocks knocked out to value 10 (exactly transparent) with
    probability p_empty: the share of skippable cells is close to p_empty."""
    rng = np.random.default_rng(seed)
    vol = rng.integers(26, 41, (n, n, n), dtype=np.uint8)
    nb = n // block
    holes = rng.random((nb, nb, nb)) < p_empty
    mask = np.repeat(np.repeat(np.repeat(holes, block, 0), block, 1), block, 2)
    vol[mask] = 10
    return vol


def _orbit_cameras(V, n, aspect):
    return [V.Camera(1.0 + 0.03 * k, 0.5 - 0.05 * k, 1.0 + 0.3 * k, (0.5, 0.5, 0.5), aspect).get_proj_view_matrix() for k in range(n)]


def _render_with_params(V, cam, vol, W, H, dt, layout, params=(), flags=0):
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    try:
        for k, v in params:
            ctx.set_param(k, v)
        V.VolumeTexture(ctx, vol, layout=layout)
        ctx.set_camera_blob(cam)
        ctx.reset_step_counts()
        V.RaycastPipeline(dt_scale=dt, flags=flags | V.RENDER_COUNT).record(ctx)
        img, steps = ctx.read_backbuffer(), ctx.read_steps()
        census = ctx.simt_census()
        V.native.check(ctx.handle, V.native.lib().vk_backbuffer_clear(ctx.handle))
        V.RaycastPipeline(dt_scale=dt, flags=flags).record(ctx)
        assert (ctx.read_backbuffer().view(np.uint32) == img.view(np.uint32)).all(), "production kernel differs from the instrumented one"
        return img, steps, census
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------
# round 4: the reference-held pin and the full-size cases bench.py times


def _captured_rgb(ctx):
    buf, dims = ctx.capture_frame()
    rows = np.frombuffer(buf, np.uint8).reshape(dims.height, dims.padded_bytes_per_row)
    return rows[:, :dims.unpadded_bytes_per_row].reshape(dims.height, dims.width, 4)
