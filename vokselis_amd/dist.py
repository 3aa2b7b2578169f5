"""Tile-parallel multi-GPU frame: one process per GPU, framebuffer tiles interleaved over ranks,
one RCCL gather of tile pixels to the root over xGMI.

The reference has no multi-GPU path; what it does have is the xor example's framebuffer tiling
with per-tile pixel offsets (examples/xor/main.rs:12,77-95,235-253).  That scheme is promoted
here to the partition: tile t (row-major, `tile_size`^2 pixels) belongs to rank t % world, each
rank renders its tiles into a compact [n_slots, ts, ts, 4] buffer (`vk_render_partition`), the
root gathers the buffers and scatters them into its backbuffer (`vk_untile`).  Rays are
independent, so there is no reduction -- the gather is the only collective.  Tiles are interleaved
(not contiguous strips) because ~70 % of a 16:9 frame misses the cube and opacity varies.

Frames are pipelined: the gather of frame k runs on the collective's stream while frame k+1 is
marched, and the root un-tiles frame k after launching frame k+1.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .context import Context, RaycastPipeline, partition_slots


def tiles_xy(width: int, height: int, tile_size: int):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def tile_owner(pos: int, world: int) -> int:
    """Rank owning position `pos` of the tile order."""
    return pos % world


def local_tiles(width: int, height: int, tile_size: int, rank: int, world: int, order=None):
    """Row-major tile ids of this rank in slot order: slot j <-> position rank + j*world of `order`
    (the library's heaviest-first order, `Context.partition_order`; identity when None)."""
    tx, ty = tiles_xy(width, height, tile_size)
    pos = range(rank, tx * ty, world)
    return list(pos) if order is None else [int(order[q]) for q in pos]


def n_slots(width: int, height: int, tile_size: int, world: int) -> int:
    tx, ty = tiles_xy(width, height, tile_size)
    return (tx * ty + world - 1) // world


class FrameGather:
    """The collective half: fixed-size gather of every rank's compact tile buffer to `root`.
    Device-agnostic (RCCL for cuda tensors, gloo for the CPU tests)."""

    def __init__(self, width: int, height: int, tile_size: int, channels_dtype, device, root: int = 0, group=None):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.root = root
        self.ts = tile_size
        self.slots = n_slots(width, height, tile_size, self.world)
        shape = (self.slots, tile_size, tile_size, 4)
        # two compact buffers: frame k+1 is rendered while frame k is in flight
        self.compact = [torch.zeros(shape, dtype=channels_dtype, device=device) for _ in range(2)]
        self.gathered = None
        if self.rank == root:
            self.gathered = [torch.zeros((self.world,) + shape, dtype=channels_dtype, device=device) for _ in range(2)]
        self._views = {}  # (buffer, n_slots) -> (send view, [recv views]): slicing tensors costs microseconds per frame

    def start(self, k: int, n_slots: int | None = None):
        """Launch the gather of the first `n_slots` slots of compact[k % 2] (async); returns the work
        handle.  Slots beyond the active ones hold nothing worth moving (`Context.partition_active`)."""
        n = self.slots if n_slots is None else min(n_slots, self.slots)
        key = (k % 2, n)
        v = self._views.get(key)
        if v is None:
            buf = self.compact[k % 2][:n]
            out = [self.gathered[k % 2][r, :n] for r in range(self.world)] if self.rank == self.root else None
            v = self._views[key] = (buf, out)
        buf, out = v
        if self.rank == self.root:
            return self.dist.gather(buf, gather_list=out, dst=self.root, group=self.group, async_op=True)
        return self.dist.gather(buf, dst=self.root, group=self.group, async_op=True)


class TileParallelRenderer:
    """March + gather + un-tile for one frame stream on this rank's GPU."""

    def __init__(self, ctx: Context, pipeline: RaycastPipeline, tile_size: int = 64, root: int = 0, group=None):
        import torch

        self.torch = torch
        self.ctx, self.pipe = ctx, pipeline
        bb = ctx.render_backbuffer
        dtype = torch.float32 if bb.format == N.OUT_RGBA32F else torch.float16
        dev = torch.device("cuda", torch.cuda.current_device())
        self.fg = FrameGather(bb.width, bb.height, tile_size, dtype, dev, root=root, group=group)
        assert self.fg.slots == partition_slots(bb.width, bb.height, tile_size, self.fg.world)
        self._pending = None  # (frame index, work) whose un-tile is still owed
        self._active, self._active_key = None, None

    @property
    def is_root(self) -> bool:
        return self.fg.rank == self.fg.root

    def submit(self, k: int):
        """Frame k: march this rank's tiles, start their gather; finish frame k-1 on the root."""
        fg = self.fg
        self.pipe.record_partition(self.ctx, fg.ts, fg.rank, fg.world, fg.compact[k % 2].data_ptr())
        if self._active_key != id(self.ctx) or self.ctx.camera.updated or self._active is None:
            self._active = self.ctx.partition_active(fg.ts, fg.world, self.pipe.mode)[1]  # per camera; cached in the library too
            self._active_key = id(self.ctx)
        n_slots = self._active
        if n_slots == 0:
            self._finish_pending()
            self._pending = (k, None)
            return
        work = fg.start(k, n_slots)
        self._finish_pending()
        self._pending = (k, work)

    def _finish_pending(self):
        if self._pending is None:
            return
        k, work = self._pending
        if work is not None:
            work.wait()  # orders the current stream after the collective
        if self.is_root:
            N.check(self.ctx.handle, N.lib().vk_untile(self.ctx.handle, self.fg.gathered[k % 2].data_ptr(), self.fg.ts,
                                                      self.fg.world, self.fg.slots))
        self._pending = None

    def flush(self):
        self._finish_pending()


def untile_reference(gathered: np.ndarray, width: int, height: int, tile_size: int, order=None, n_active=None) -> np.ndarray:
    """numpy statement of vk_untile: [world, n_slots, ts, ts, C] -> [H, W, C]; tiles at positions
    >= n_active of the order are clear colour (0,0,0,1)."""
    world = gathered.shape[0]
    tx, ty = tiles_xy(width, height, tile_size)
    pos = np.arange(tx * ty) if order is None else np.argsort(np.asarray(order))  # tile id -> position
    out = np.zeros((height, width, gathered.shape[-1]), gathered.dtype)
    for y0 in range(0, height, tile_size):
        for x0 in range(0, width, tile_size):
            q = int(pos[(y0 // tile_size) * tx + x0 // tile_size])
            h, w = min(tile_size, height - y0), min(tile_size, width - x0)
            if n_active is not None and q >= n_active:
                out[y0:y0 + h, x0:x0 + w] = [0, 0, 0, 1][: out.shape[-1]]
            else:
                out[y0:y0 + h, x0:x0 + w] = gathered[q % world, q // world, :h, :w]
    return out
