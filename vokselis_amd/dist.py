"""Tile-parallel multi-GPU frame: one process per GPU, framebuffer tiles interleaved over ranks,
one RCCL gather of tile pixels to the root over xGMI.

The reference has no multi-GPU path; what it does have is the xor example's framebuffer tiling
with per-tile pixel offsets (examples/xor/main.rs:12,77-95,235-253).  That scheme is promoted
here to the partition: tile t (row-major, `tile_size`^2 pixels) belongs to rank t % world, each
rank renders its tiles into a compact [n_slots, ts, ts, 4] buffer (`vk_render_partition`), the
root gathers the buffers and scatters them into its backbuffer (`vk_untile`).  Rays are
independent, so there is no reduction -- the gather is the only collective.  Tiles are interleaved
(not contiguous strips) because ~70 % of a 16:9 frame misses the cube and opacity varies.

Frames are pipelined in batches: the gather of batch g (several frames per collective call) runs on
the collective's stream while batch g+1 is marched, and the root un-tiles batch g after launching g+1.
"""
from __future__ import annotations

import ctypes as C

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
    """The collective half: gather of every rank's compact tile buffers to `root`, `batch` frames per
    collective.  Device-agnostic (RCCL for cuda tensors, gloo for the CPU tests).

    A collective call costs ~100 us of host time through torch.distributed -- more than a whole C2
    frame takes to march on one GPU -- so frames are moved `batch` at a time: rank buffers are
    [batch, n, ts, ts, 4] (n = active slots of the current camera), two sets, so batch g+1 is marched
    while batch g is on the wire.  The root receives [world, batch, n, ts, ts, 4]."""

    def __init__(self, width: int, height: int, tile_size: int, channels_dtype, device, root: int = 0, group=None, batch: int = 1):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.root = root
        self.ts = tile_size
        self.batch = max(1, int(batch))
        self.slots = n_slots(width, height, tile_size, self.world)
        self.dtype, self.device = channels_dtype, device
        self._bufs = {}   # n -> (compact[2], gathered[2] or None)
        self._views = {}  # (n, set, count) -> (send view, [recv views]): slicing tensors costs microseconds per frame

    def buffers(self, n: int | None = None):
        """(compact sets, gathered sets) for n slots per frame (default: all slots)."""
        n = self.slots if n is None else min(int(n), self.slots)
        v = self._bufs.get(n)
        if v is None:
            torch = self.torch
            shape = (self.batch, n, self.ts, self.ts, 4)
            compact = [torch.zeros(shape, dtype=self.dtype, device=self.device) for _ in range(2)]
            gathered = None
            if self.rank == self.root:
                gathered = [torch.zeros((self.world,) + shape, dtype=self.dtype, device=self.device) for _ in range(2)]
            v = self._bufs[n] = (compact, gathered)
        return v

    def start(self, set_: int, n: int | None = None, count: int | None = None):
        """Launch the gather of the first `count` frames of compact set `set_` (async); returns the work handle."""
        n = self.slots if n is None else min(int(n), self.slots)
        count = self.batch if count is None else count
        key = (n, set_, count)
        v = self._views.get(key)
        if v is None:
            compact, gathered = self.buffers(n)
            buf = compact[set_][:count]
            out = [gathered[set_][r, :count] for r in range(self.world)] if self.rank == self.root else None
            v = self._views[key] = (buf, out)
        buf, out = v
        if self.rank == self.root:
            return self.dist.gather(buf, gather_list=out, dst=self.root, group=self.group, async_op=True)
        return self.dist.gather(buf, dst=self.root, group=self.group, async_op=True)


class TileParallelRenderer:
    """March + gather + un-tile for one frame stream on this rank's GPU."""

    def __init__(self, ctx: Context, pipeline: RaycastPipeline, tile_size: int = 64, root: int = 0, group=None, batch: int = 1,
                 frames_in_flight: int = 1, on_frame=None, gather_cls=None):
        """`batch` frames travel per gather call; up to `frames_in_flight` of them are marched concurrently,
        each on its own stream into its own slice of the batch buffer.  A rank's share of a small frame is a
        few hundred waves whose length is set by the slowest one; overlapping consecutive frames is what keeps
        the GPU busy (the reference, like any wgpu app, also keeps more than one frame in flight)."""
        import torch

        self.torch = torch
        self.ctx, self.pipe = ctx, pipeline
        # The collective is ordered against torch's *current* stream; the march and the un-tile must run
        # on that same stream or the gather could start before the tiles are written.
        # (The legacy default stream has handle 0, which the C-ABI reads as "the context's own": drive
        # the renderer inside `with torch.cuda.stream(s)` for a real stream s, as bench.py does.)
        cur = torch.cuda.current_stream().cuda_stream
        if not cur:
            raise RuntimeError("TileParallelRenderer needs a non-default torch stream: create it inside `with torch.cuda.stream(torch.cuda.Stream())`")
        if ctx.stream_handle != cur:
            ctx.set_stream(cur)
        self._stream = cur
        bb = ctx.render_backbuffer
        dtype = torch.float32 if bb.format == N.OUT_RGBA32F else torch.float16
        dev = torch.device("cuda", torch.cuda.current_device())
        # gather_cls: a FrameGather subclass (tests move the buffers over gloo through host memory, so that several
        # ranks can share one GPU); the production class hands the device buffers to RCCL
        self.fg = (gather_cls or FrameGather)(bb.width, bb.height, tile_size, dtype, dev, root=root, group=group, batch=batch)
        assert self.fg.slots == partition_slots(bb.width, bb.height, tile_size, self.fg.world)
        self._esize = 4 if bb.format == N.OUT_RGBA32F else 2
        self._pending = None  # (set, n, count, work, order epoch) of the batch whose un-tile is still owed
        self._epoch = 0
        self._active, self._active_key = None, None
        self._n = None        # active slots of the batch being filled
        self._set, self._filled = 0, 0
        self.on_frame = on_frame  # root only: called as on_frame(k) right after frame k (in submit order) was un-tiled
        self._submitted, self._delivered = 0, 0
        self._fif = max(1, min(int(frames_in_flight), self.fg.batch))
        self._main = torch.cuda.current_stream()
        self._side, self._done, self._free = [], [], [None, None]
        if self._fif > 1:
            self._side = [torch.cuda.Stream() for _ in range(self._fif)]
            self._done = [[torch.cuda.Event() for _ in range(self.fg.batch)] for _ in range(2)]  # frame b of set s marched
            self._free = [None, None]  # set s may be overwritten once this event (recorded on the main stream) has passed

    @property
    def is_root(self) -> bool:
        return self.fg.rank == self.fg.root

    def submit(self, k: int = 0):
        """Next frame: march this rank's tiles into the current batch; when the batch is full, start its
        gather and finish the previous batch on the root.  (`k` is informational: frames are taken in call order.)"""
        fg = self.fg
        if self.torch.cuda.current_stream().cuda_stream != self._stream:
            raise RuntimeError("TileParallelRenderer must be driven on the torch stream it was created on")
        key = (id(self.ctx), self.ctx.camera_epoch)
        if self._active_key != key:
            # A new camera re-deals the tiles (other order, maybe another active set): the batch being filled is
            # closed at this point, so a batch holds frames of ONE order, and it is un-tiled later under that
            # order's epoch (the library keeps the tables of the last 16 orders) -- the pipeline keeps running.
            if self._filled:
                self._launch_batch()
            self._active = self.ctx.partition_active(fg.ts, fg.world, self.pipe.mode)[1]  # per uploaded camera; cached in the library too
            self._active_key = key
            e = C.c_uint32()
            N.check(self.ctx.handle, N.lib().vk_partition_epoch(self.ctx.handle, C.byref(e)))
            self._epoch = e.value
        n = self._active
        self._n = n
        if n > 0:
            compact, _ = fg.buffers(n)
            frame_bytes = n * fg.ts * fg.ts * 4 * self._esize
            dst = compact[self._set].data_ptr() + self._filled * frame_bytes
            if self._fif > 1:
                st = self._side[self._filled % self._fif]
                if self._filled < self._fif:  # first use of this stream in the batch: the set must be free, the camera uploaded
                    st.wait_stream(self._main) if self._free[self._set] is None else st.wait_event(self._free[self._set])
                self.pipe.record_partition(self.ctx, fg.ts, fg.rank, fg.world, dst, stream=st.cuda_stream)
                self._done[self._set][self._filled].record(st)
            else:
                self.pipe.record_partition(self.ctx, fg.ts, fg.rank, fg.world, dst)
        self._filled += 1
        if self._filled == fg.batch:
            self._launch_batch()

    def _launch_batch(self):
        fg, n, count, set_ = self.fg, self._n, self._filled, self._set
        if self._fif > 1 and n > 0:
            for b in range(count):
                self._main.wait_event(self._done[set_][b])  # the gather (ordered after the main stream) sees every frame
        work = fg.start(set_, n, count) if n > 0 else None
        self._finish_pending()
        self._pending = (set_, n, count, work, self._epoch)
        self._set, self._filled = set_ ^ 1, 0

    def _finish_pending(self):
        if self._pending is None:
            return
        set_, n, count, work, epoch = self._pending
        if work is not None:
            work.wait()  # orders the current stream after the collective
        if self._fif > 1:
            ev = self._free[set_] or self.torch.cuda.Event()
            ev.record(self._main)  # the send buffers of this set are free from here on
            self._free[set_] = ev
        if self.is_root:
            fg = self.fg
            if n > 0:
                base = fg.buffers(n)[1][set_].data_ptr()
                frame_bytes = n * fg.ts * fg.ts * 4 * self._esize
                for b in range(count):  # every frame of the batch materialises in the root's backbuffer, in order
                    N.check(self.ctx.handle, N.lib().vk_untile_epoch(self.ctx.handle, base + b * frame_bytes, fg.ts, fg.world, fg.batch * n, epoch))
                    if self.on_frame is not None:
                        self.on_frame(self._delivered + b)
            else:
                # no tile touches the cube: the un-tile only clears (it reads no slot), once per frame
                dummy = fg.buffers(1)[1][0].data_ptr()
                for b in range(count):
                    N.check(self.ctx.handle, N.lib().vk_untile_epoch(self.ctx.handle, dummy, fg.ts, fg.world, fg.batch, epoch))
                    if self.on_frame is not None:
                        self.on_frame(self._delivered + b)
        self._delivered += count
        self._pending = None

    def flush(self):
        if self._filled:
            self._launch_batch()
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
