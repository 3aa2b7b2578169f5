"""Tile-parallel multi-GPU frames: one process per GPU, framebuffer tiles dealt over the ranks, one gather of
finished tiles to the root over xGMI.

The reference has no multi-GPU path; what it does have is the xor example's framebuffer tiling with per-tile pixel
offsets (examples/xor/main.rs:12,77-95,235-253).  That scheme is promoted here to the partition: position q of a
frame's heaviest-first tile order belongs to rank q % world, slot q // world.  Rays are independent, so there is no
reduction -- the gather is the only collective.  Tiles are interleaved (not contiguous strips) because ~70 % of a
16:9 frame misses the cube and opacity varies.

Frames travel in batches: ONE launch marches this rank's tiles of B frames (`vk_render_batch`, each frame with its
own camera), ONE gather moves them ([slot][frame][ts][ts]: the active slots are a contiguous prefix), ONE launch on
the root un-tiles them (`vk_untile_batch`).  The gather goes through the library's own RCCL communicator
(`vk_gather_tiles`: a grouped send/recv, no Python per call beyond the ctypes hop) on a second stream, so the wire
time of batch g hides behind the march of batch g+1; `transport="torch"` moves the same buffers through
`torch.distributed` instead (gloo in the CPU tests, or when several test ranks share one GPU).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context, RaycastPipeline, partition_slots, render_batch, untile_batch


def tiles_xy(width: int, height: int, tile_size: int):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def deal_pos(rank: int, slot: int, world: int, root_skip: int = 0) -> int:
    """Position of the tile order that (rank, slot) marches.  Rounds give one position to every rank; with
    root_skip = k >= 2 rank 0 sits out every k-th round (vk_partition_root_skip; deal_pos of vk_common.hpp)."""
    k = root_skip
    if k < 2:
        return rank + slot * world
    rnd = slot if rank else slot + slot // (k - 1)
    return rnd * world - rnd // k + rank - (1 if rnd % k == k - 1 else 0)


def deal_owner(pos: int, world: int, root_skip: int = 0):
    """(rank, slot) owning position `pos` (the inverse of deal_pos)."""
    k = root_skip
    if k < 2:
        return pos % world, pos // world
    G = k * world - 1
    g, o = divmod(pos, G)
    if o < (k - 1) * world:
        rj, rank = divmod(o, world)
    else:
        rj, rank = k - 1, o - (k - 1) * world + 1
    return rank, (g * k + rj if rank else g * (k - 1) + rj)


def deal_rounds(tiles: int, world: int, root_skip: int = 0) -> int:
    """Rounds needed to deal `tiles` positions = slots of a non-root rank."""
    k = root_skip
    if k < 2:
        return (tiles + world - 1) // world
    r = tiles // world
    while r * world - r // k < tiles:
        r += 1
    return r


def tile_owner(pos: int, world: int, root_skip: int = 0) -> int:
    """Rank owning position `pos` of the tile order."""
    return deal_owner(pos, world, root_skip)[0]


def local_tiles(width: int, height: int, tile_size: int, rank: int, world: int, order=None, root_skip: int = 0):
    """Row-major tile ids of this rank in slot order (`order`: the library's heaviest-first order,
    `Context.partition_order`; identity when None)."""
    tx, ty = tiles_xy(width, height, tile_size)
    out = []
    for slot in range(deal_rounds(tx * ty, world, root_skip)):
        q = deal_pos(rank, slot, world, root_skip)
        if q < tx * ty and deal_owner(q, world, root_skip) == (rank, slot):
            out.append(q if order is None else int(order[q]))
    return out


def n_slots(width: int, height: int, tile_size: int, world: int, root_skip: int = 0) -> int:
    tx, ty = tiles_xy(width, height, tile_size)
    return deal_rounds(tx * ty, world, root_skip)


class PeerLostError(RuntimeError):
    """A gather did not complete: a peer has gone (its process ended, its GPU hung) or the wire is stuck.  Raised on the surviving ranks
    within the renderer's time limit instead of leaving them parked inside a collective; the library communicator, if any, has been
    aborted (vk_comm_abort), so the process can report and leave -- a retry belongs to a NEW job (fresh processes, a fresh rendezvous)."""


class TorchTileGather:
    """The collective through torch.distributed: every rank's [n, B, ts, ts, 4] prefix to the root's
    [world, n, B, ts, ts, 4].  Device-agnostic (gloo for CPU tensors, RCCL for cuda tensors); `via_host` stages
    cuda tensors through host memory so that several test ranks can share one GPU over gloo."""

    def __init__(self, group=None, root: int = 0, via_host: bool = False):
        import torch.distributed as dist

        self.dist, self.group, self.root, self.via_host = dist, group, root, via_host
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def _gather(self, send, out, root, timeout_s):
        """dist.gather with a time limit: a peer that is gone makes it fail (gloo: connection closed) or never finish (timeout)."""
        import datetime

        try:
            work = self.dist.gather(send, gather_list=out, dst=root, group=self.group, async_op=True)
            if timeout_s is None:
                work.wait()
            else:
                work.wait(datetime.timedelta(seconds=float(timeout_s)))
        except Exception as e:  # noqa: BLE001  (torch raises RuntimeError / DistBackendError here, with backend-specific texts)
            raise PeerLostError("rank %d: the tile gather to rank %d failed or did not finish within %s s: %s" % (self.rank, root, timeout_s, str(e)[:300])) from e

    def gather(self, send, recv, root=None, timeout_s=None):
        """send: tensor [n, B, ...]; recv (on the root): tensor [world, n, B, ...].  Blocking, for at most `timeout_s` seconds (None: no
        limit); PeerLostError when it fails or runs out of time.  `root`: this gather's destination (default: the one given at construction)."""
        root = self.root if root is None else root
        if self.via_host:
            s = send.cpu()
            out = [s.new_empty(s.shape) for _ in range(self.world)] if self.rank == root else None
            self._gather(s, out, root, timeout_s)
            if self.rank == root:
                for r in range(self.world):
                    recv[r].copy_(out[r])
            return
        out = [recv[r] for r in range(self.world)] if self.rank == root else None
        self._gather(send, out, root, timeout_s)


class BatchTileRenderer:
    """March + gather + un-tile of a frame stream on this rank's GPU, `batch` frames per launch and per gather.

    submit(camera_blob) queues a frame; a full batch is launched at once.  On the root, finished batches arrive in
    `on_batch(first_frame_index, count, frames)` with frames a [count, H, W, 4] cuda tensor view (valid until the
    next-but-one batch is launched; READ-ONLY: the next un-tile into that buffer relies on what this one left there and
    rewrites only the tiles whose state changed).  flush() launches a partial batch and drains.

    root = "rotate": launch g's frames are assembled on rank g mod world instead of always on one root (on_batch then fires on
    that rank).  A fixed root receives 7/8 of every frame over the ONE link each peer has to it, and at 8 GPUs on the 1080p
    configuration those links, not the march, set the pace; rotating the destination spreads the same bytes over all
    56 directed links of the node -- for consumers that are themselves per GPU (an encoder, a NIC), not for a window on GPU 0."""

    def __init__(self, ctx: Context, pipeline: RaycastPipeline, tile_size: int = 64, batch: int = 16, root: int = 0, group=None,
                 transport: str = "rccl", on_batch=None, via_host: bool = False, root_skip="auto", wire: int = N.WIRE_RGB, timeout_s: float | None = 120.0):
        import torch
        import torch.distributed as dist

        self.torch = torch
        # Time limit of every wait on a gather (PeerLostError beyond it; None: wait for ever, the behaviour before round 6).  A gather of one
        # batch takes milliseconds: the limit only has to be longer than a peer's slowest march of a batch.
        self.timeout_s = timeout_s
        self._dead = False
        self.rotate = root == "rotate"
        root = 0 if self.rotate else int(root)
        self.ctx, self.pipe, self.ts, self.batch, self.root = ctx, pipeline, tile_size, max(1, int(batch)), root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.on_batch = on_batch
        bb = ctx.render_backbuffer
        self.W, self.H = bb.width, bb.height
        self.dtype = torch.float32 if bb.format == N.OUT_RGBA32F else torch.float16
        self.esize = 4 if bb.format == N.OUT_RGBA32F else 2
        # Tiles travel as colour only (alpha is 1 in every pixel of this path): 6 bytes per rgba16f pixel instead of 8 on the one
        # xGMI link each peer has to the root.  Every rank constructs this object with the same arguments.
        self._wire_before = N.WIRE_RGB if ctx.wire_pixel_bytes in (6, 12) else N.WIRE_RGBA  # restored by close()
        ctx.set_wire(wire)
        self.wire, self.ch = wire, (3 if wire == N.WIRE_RGB else 4)
        self.tile_elems = tile_size * tile_size * self.ch  # elements of one (slot, frame) record
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self._group = group
        # The march, the un-tile and (torch transport) the gather run on ONE stream of this object's choosing: torch's
        # current stream when that is a real stream, else a new one (handle 0, the legacy default stream, would read as
        # "the context's own" in the C-ABI).  Every call below enters it itself, so the caller's current stream does
        # not matter; frames handed to on_batch are ordered on `march_stream`.
        cur = torch.cuda.current_stream()
        if not cur.cuda_stream:
            cur = torch.cuda.Stream()
        if ctx.stream_handle != cur.cuda_stream:
            ctx.set_stream(cur.cuda_stream)
        self.march_stream = cur
        self.comm_stream = torch.cuda.Stream()
        self.transport = transport
        if transport == "rccl":
            # The library's own communicator: rank 0 makes the id, torch.distributed (already up) ships its 128 bytes.
            # ncclCommInitRank blocks until every rank has joined, so no rank may enter it unless all of them can: every
            # rank first proves that it can load RCCL (vk_comm_available: no side effect -- an id made and thrown away would
            # leave a bootstrap listener waiting for peers that never come) and the ranks agree on that; rank 0 alone makes
            # the id; they agree again on the outcome of the join itself.  A rank that fails raises on EVERY rank instead
            # of leaving its peers parked inside a collective, and the ranks that did join leave the communicator first.
            idbuf = (C.c_ubyte * 128)()
            why = None
            try:
                N.check(None, N.lib().vk_comm_available())
                if self.rank == 0:
                    N.check(None, N.lib().vk_comm_unique_id(idbuf))
            except Exception as e:  # noqa: BLE001
                why = "rank %d cannot use RCCL: %r" % (self.rank, e)
            self._agree(why, "loading RCCL")
            obj = [bytes(idbuf) if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(obj, src=0, group=group)
            why, joined = None, False
            try:
                N.check(ctx.handle, N.lib().vk_comm_init_rank(ctx.handle, obj[0], self.rank, self.world))
                joined = True
            except Exception as e:  # noqa: BLE001
                why = "rank %d could not join the communicator: %r" % (self.rank, e)
            try:
                self._agree(why, "joining the communicator")
            except RuntimeError:
                if joined:
                    N.lib().vk_comm_destroy(ctx.handle)
                raise
            self.tg = None
        elif transport == "torch":
            self.tg = TorchTileGather(group, root, via_host)
        else:
            raise ValueError("transport is 'rccl' or 'torch'")
        # The root also receives and un-tiles every frame: give it a lighter share of the march (rank 0 sits out every
        # k-th round of the deal).  "auto": k from this GPU's own timings of one batch's march and un-tile.
        with torch.cuda.stream(self.march_stream):
            # (a rotating root has no rank that un-tiles more than the others: an even deal)
            self.root_skip = 0 if self.rotate else (self._calibrate_root_skip() if root_skip == "auto" else int(root_skip))
        ctx.set_root_skip(self.root_skip if self.world > 1 else 0)
        self.cap = partition_slots(self.W, self.H, tile_size, self.world, self.root_skip if self.world > 1 else 0)
        shape = (self.cap, self.batch, self.tile_elems)
        # The buffers are filled (zeroed) on `march_stream`, not on the caller's current stream: torch's streams do not
        # synchronise with the legacy default stream, and a zero-fill still queued there could land AFTER the first march
        # has written its tiles (seen once in ~15 runs of the test that drives this class from the default stream).
        with torch.cuda.stream(self.march_stream):
            self.send = [torch.zeros(shape, dtype=self.dtype, device=self.dev) for _ in range(2)]
            self.recv = self.frames = None
            if self.is_root or self.rotate:
                self.recv = [torch.zeros((self.world * self.cap * self.batch, self.tile_elems), dtype=self.dtype, device=self.dev) for _ in range(2)]
                self.frames = [torch.zeros((self.batch, self.H, self.W, 4), dtype=self.dtype, device=self.dev) for _ in range(2)]
        self.marched = [torch.cuda.Event() for _ in range(2)]   # set s: tiles written
        self.moved = [torch.cuda.Event() for _ in range(2)]     # set s: gather done (send[s] free, recv[s] valid)
        self._used = [False, False]
        self._frames_bid = [0, 0]  # the batch last un-tiled into frames[s] (0: none yet -- the buffer holds zeros, not the clear colour)
        self._cams, self._set = [], 0
        self._pending = None  # (set, batch id, active slots, count, first index, root of that launch)
        self._submitted = 0
        self._launches = 0

    @property
    def is_root(self) -> bool:
        return self.rank == self.root

    def _agree(self, why, what: str):
        """Every rank reaches this point with its own outcome (`why`: None = fine); all of them leave it with the same
        verdict: either nobody failed, or everybody raises with the failing ranks' messages."""
        import torch.distributed as dist

        if self.world == 1:
            if why is not None:
                raise RuntimeError(why)
            return
        all_why = [None] * self.world
        dist.all_gather_object(all_why, why, group=self._group)
        bad = [w for w in all_why if w is not None]
        if bad:
            raise RuntimeError("%s failed on %d of %d ranks: %s" % (what, len(bad), self.world, "; ".join(bad)))

    def _calibrate_root_skip(self) -> int:
        """k = (M + U) / (U * world) balances the root's march share + un-tile against a peer's share, with M the time
        to march one whole frame's tiles and U the time to un-tile one frame, both measured here on a 4-frame batch."""
        import torch.distributed as dist

        if self.world == 1 or self.root != 0:  # (uniform over the ranks: everybody joins the broadcast below)
            return 0
        k = [0]
        if self.is_root and self.ctx.camera_blob is not None:
            try:  # (a failure here must still reach the broadcast: the other ranks are waiting in it)
                k[0] = self._measure_root_share()
            except Exception as e:  # noqa: BLE001
                k[0] = "calibration of the root's share failed on the root: %r" % (e,)
        dist.broadcast_object_list(k, src=self.root, group=self._group)
        if isinstance(k[0], str):
            raise RuntimeError(k[0])
        return int(k[0])

    def _measure_root_share(self) -> int:
        torch = self.torch
        B = min(4, self.batch)
        cap1 = partition_slots(self.W, self.H, self.ts, 1)
        tiles = torch.zeros((cap1, B, self.tile_elems), dtype=self.dtype, device=self.dev)
        frames = torch.zeros((B, self.H, self.W, 4), dtype=self.dtype, device=self.dev)
        cams = [self.ctx.camera_blob] * B
        self.ctx.set_root_skip(0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for _ in range(2):  # the second pass is the measurement
            ev[0].record()
            bid, act = render_batch(self.ctx, self.pipe, cams, tiles.data_ptr(), tile_size=self.ts, rank=0, nranks=1, compact=True, slot_capacity=cap1)
            ev[1].record()
            untile_batch(self.ctx, bid, tiles.data_ptr(), act, frames.data_ptr())
            ev[2].record()
            torch.cuda.current_stream().synchronize()
        m, u = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        if not (u > 0 and m > 0):
            return 0
        kk = int(round((m + u) / (u * self.world)))
        return 0 if kk > 64 else max(2, kk)

    def submit(self, camera_blob: bytes | None = None):
        blob = camera_blob if camera_blob is not None else self.ctx.camera_blob
        if blob is None:
            raise RuntimeError("no camera: pass a 144-byte blob or upload one to the context")
        self._cams.append(blob)
        if len(self._cams) == self.batch:
            self._launch()

    def _lost(self, what: str):
        """The survivor's way out: abort the library communicator (never destroy it: that waits for the dead transfer), mark the renderer
        dead, raise.  The caller reports and ends the process."""
        self._dead = True
        if self.transport == "rccl":
            try:
                N.lib().vk_comm_abort(self.ctx.handle)
            except Exception:  # noqa: BLE001
                pass
        raise PeerLostError("rank %d: %s did not complete within %s s -- a peer is gone or its gather is stuck; the communicator has been aborted" % (self.rank, what, self.timeout_s))

    def _wait_event(self, ev, what: str):
        """Host-side wait for a cuda event with the renderer's time limit (a stream-side wait on a gather that never completes would park
        every later launch behind it, and the host in the library's next blocking call)."""
        import time

        if self.timeout_s is None:
            ev.synchronize()
            return
        t0 = time.monotonic()
        t_end = t0 + float(self.timeout_s)
        while not ev.query():
            now = time.monotonic()
            if now > t_end:
                self._lost(what)
            if now - t0 > 2e-3:  # (a gather in its last microseconds is polled, not slept on: the next launch waits for this)
                time.sleep(2e-4)

    def _launch(self):
        if self._dead:
            raise PeerLostError("rank %d: this renderer lost a peer earlier; start a new job" % self.rank)
        with self.torch.cuda.stream(self.march_stream):
            self._launch_on_stream()

    def _launch_on_stream(self):
        s, cams = self._set, self._cams
        count = len(cams)
        if len(cams) < self.batch:  # a partial batch marches (and moves) whole batches: pad with the last camera
            cams = cams + [cams[-1]] * (self.batch - len(cams))
        if self._used[s]:
            self._wait_event(self.moved[s], "the gather of two batches ago")  # (bounded, on the host: normally long done)
            self.march_stream.wait_event(self.moved[s])  # the set's previous gather has read its tiles
        # Rank 0's share of a gather to rank 0 sits at the very start of the receive buffer whatever the active-slot count turns out to be: it
        # marches straight into it (vk_gather_tiles then finds send == its own segment and copies nothing -- at a world of one that copy was
        # the whole frame's tiles, 2.3 us per C2 frame).  The un-tile of this set's previous batch has read recv[s] on this stream already.
        in_place = self.transport == "rccl" and self.rank == 0 and self.is_root and not self.rotate
        send_ptr = self.recv[s].data_ptr() if in_place else self.send[s].data_ptr()
        bid, act = render_batch(self.ctx, self.pipe, cams, send_ptr, tile_size=self.ts, rank=self.rank, nranks=self.world,
                                compact=True, slot_capacity=self.cap)
        self.marched[s].record(self.march_stream)
        # the gather of this batch, on the communication stream
        n_px = act * self.batch * self.ts * self.ts
        root = (self._launches % self.world) if self.rotate else self.root
        self._launches += 1
        mine = self.rank == root
        if self.transport == "rccl":
            self.comm_stream.wait_event(self.marched[s])
            recv_ptr = self.recv[s].data_ptr() if mine else None
            N.check(self.ctx.handle, N.lib().vk_gather_tiles(self.ctx.handle, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), n_px, root,
                                                            C.c_void_p(self.comm_stream.cuda_stream)))
            self.moved[s].record(self.comm_stream)
        else:
            self.marched[s].synchronize()
            if act > 0:
                recv = self.recv[s][: self.world * act * self.batch].view(self.world, act, self.batch, self.tile_elems) if mine else None
                try:
                    self.tg.gather(self.send[s][:act], recv, root, self.timeout_s)
                except PeerLostError:
                    self._dead = True
                    raise
            self.moved[s].record(self.march_stream)
        self._used[s] = True
        # the previous batch is on the root by now (its gather overlapped this march): un-tile and deliver it
        self._finish_pending()
        self._pending = (s, bid, act, count, self._submitted, root)
        self._submitted += count
        self._cams, self._set = [], s ^ 1

    def _finish_pending(self):
        if self._pending is None:
            return
        s, bid, act, count, first, root = self._pending
        self._pending = None
        if self.rank != root:
            return
        # (a stream-side wait: the host goes on preparing the next batch while this gather is on the wire.  A gather that never completes is
        # found by the bounded host-side wait on this set's `moved` event at the launch two batches on, or by flush().)
        self.march_stream.wait_event(self.moved[s])
        # frames[s] still holds what this object un-tiled into it two batches ago: only tiles whose state changed are cleared
        untile_batch(self.ctx, bid, self.recv[s].data_ptr(), act, self.frames[s].data_ptr(), prev_batch_id=self._frames_bid[s])
        self._frames_bid[s] = bid
        if self.on_batch is not None:
            self.on_batch(first, count, self.frames[s][:count])

    def flush(self):
        """Launch a partial batch, deliver what is pending and wait -- for at most `timeout_s` per wait -- until this rank's part of every
        gather is through.  PeerLostError when a peer is gone."""
        if self._dead:
            raise PeerLostError("rank %d: this renderer lost a peer earlier; start a new job" % self.rank)
        if self._cams:
            self._launch()
        with self.torch.cuda.stream(self.march_stream):
            self._finish_pending()
        for s in (0, 1):  # a non-root rank's sends, and whatever the un-tile left on the march stream
            if self._used[s]:
                self._wait_event(self.moved[s], "a gather")
        done = self.torch.cuda.Event()
        done.record(self.march_stream)
        self._wait_event(done, "the un-tile behind the last gather")

    def close(self):
        if self._dead:  # nothing to drain: the communicator is gone; give the context its wire format back and leave
            self.ctx.set_wire(self._wire_before)
            return
        self.flush()
        if self.transport == "rccl":
            self.torch.cuda.synchronize()
            N.check(self.ctx.handle, N.lib().vk_comm_destroy(self.ctx.handle))
        # the wire format is a property of the context: later users of its compact paths get back what they had
        self.torch.cuda.synchronize()
        self.ctx.set_wire(self._wire_before)


def untile_reference(gathered: np.ndarray, width: int, height: int, tile_size: int, order=None, n_active=None, root_skip: int = 0) -> np.ndarray:
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
                r, sl = deal_owner(q, world, root_skip)
                out[y0:y0 + h, x0:x0 + w] = gathered[r, sl, :h, :w]
    return out


def untile_batch_reference(gathered: np.ndarray, width: int, height: int, tile_size: int, orders=None, n_active=None, root_skip: int = 0) -> np.ndarray:
    """numpy statement of vk_untile_batch: [world, n_slots, B, ts, ts, C] -> [B, H, W, C] (per-frame orders / active counts)."""
    B = gathered.shape[2]
    return np.stack([untile_reference(gathered[:, :, b], width, height, tile_size, None if orders is None else orders[b],
                                      None if n_active is None else n_active[b], root_skip) for b in range(B)])
