"""Child process of tests/test_parity_gpu.py::test_multi_peer_branches_under_fake_rccl.

Runs with VK_RCCL_LIB=tests/_build/libfake_rccl.so (a single-process stand-in for RCCL: tests/fake_rccl.cpp), so that
the branches of the library that only execute with more than one peer do execute on a one-GPU box:

  * vk_group_create / vk_group_render with n = 2, 3, 8 contexts, all on GPU 0, root_skip 0 / 2 / 3:
    the n > 1 branch of vk_group_render (weighted deal, per-rank compact launches, grouped send/recv of the active
    prefix, un-tile on the root) -- every frame bitwise equal to vk_render's;
  * vk_comm_init_rank on n separate contexts + vk_render_batch(compact) + vk_gather_tiles + vk_untile_batch:
    the root branch of vk_gather_tiles (self copy + one receive per peer inside one group) with the gather on a second
    stream, as vokselis_amd.dist.BatchTileRenderer drives it (one process per GPU there, one process here).

It generalises the reference's tile loop (examples/xor/main.rs:235-254: one dispatch per tile) to one peer per share of
the tiles.  Prints one line per case and "shim_multi_rank_check: OK"; any mismatch raises.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    shim = os.environ.get("VK_RCCL_LIB", "")
    assert shim and os.path.exists(shim), "VK_RCCL_LIB must name the fake RCCL library"
    import torch

    import vokselis_amd as V
    from oracle import oracle as O

    O.build()
    L = V.native.lib()
    check = V.native.check
    fake = C.CDLL(shim)
    fake.fake_rccl_unmatched.restype = C.c_ulonglong

    def stats():
        t, b = C.c_ulonglong(), C.c_ulonglong()
        fake.fake_rccl_stats(C.byref(t), C.byref(b))
        return t.value, b.value

    W, H, ts, nvox, dt = 328, 200, 32, 48, 0.5  # (W not a multiple of the tile: ragged right-hand tiles)
    vol = O.volume_standin_u8(nvox)
    B = 5
    cams = [V.Camera(1.0, 0.5 + 0.03 * j, 1.0 + 0.21 * j, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(B)]
    # the frames vk_render makes, one context
    want = []
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F)
    V.VolumeTexture(ctx, vol)
    for cam in cams:
        ctx.set_camera_blob(cam)
        V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=dt).record(ctx)
        want.append(ctx.read_backbuffer().copy())
    ctx.close()
    want = np.stack(want)
    assert len({w.tobytes() for w in want}) == B, "the cameras must differ"

    # ---- (b) one process for the node: vk_group_* ------------------------------------------------------------
    for n in (2, 3, 8):
        for root_skip in (0, 2, 3):
            t0, b0 = stats()
            ords = (C.c_int * n)(*([0] * n))
            g = C.c_void_p()
            assert L.vk_group_create(n, ords, C.byref(g)) == 0, L.vk_last_error(None)
            try:
                for i in range(n):
                    c = C.c_void_p(L.vk_group_ctx(g, i))
                    check(c, L.vk_backbuffer_resize(c, W, H, V.OUT_RGBA32F))
                    check(c, L.vk_volume_upload(c, vol.ctypes.data, None, nvox, nvox, nvox, V.FMT_R8_UNORM, V.LAYOUT_AUTO))
                root = C.c_void_p(L.vk_group_ctx(g, 0))
                check(root, L.vk_partition_root_skip(root, root_skip))
                wire = V.WIRE_RGB if (n + root_skip) % 2 else V.WIRE_RGBA  # colour-only tiles on the wire in half of the cases
                check(root, L.vk_partition_wire(root, wire))
                # a member's communicator is the group's: the per-rank entry points must refuse it
                assert L.vk_comm_destroy(root) != 0
                out = C.c_void_p()
                check(root, L.vk_device_alloc(root, B * W * H * 16, C.byref(out)))
                for rep in range(2):  # twice: the second call reuses the group's buffers and the batch slots
                    rc = L.vk_group_render(g, V.MODE_NAIVE_TRILINEAR, B, b"".join(cams), ts, dt, 0, out)
                    assert rc == 0, L.vk_group_last_error(g)
                    assert L.vk_group_sync(g) == 0
                    got = np.empty((B, H, W, 4), np.float32)
                    check(root, L.vk_device_download(root, got.ctypes.data, out, got.nbytes))
                    assert (got.view(np.uint32) == want.view(np.uint32)).all(), ("vk_group_render", n, root_skip, rep)
                t_g, _ = stats()
                # peer-direct (round 4): every member stores its tiles straight into the root's frames -- no staging buffer, no transfer, no
                # un-tile; here the "peers" share GPU 0, so the stores are local, but offsets, the deal, the root's clearing strips and the
                # cross-stream ordering are the ones a node runs.  Stale contents must not survive: poison the frames first.
                assert L.vk_group_peer_direct(g, 1) == 0, L.vk_group_last_error(g)
                for rep in range(2):
                    frames_t = torch.full((B, H, W, 4), -7.0, dtype=torch.float32, device="cuda")  # (the root is GPU 0, torch's device)
                    torch.cuda.synchronize()
                    rc = L.vk_group_render(g, V.MODE_NAIVE_TRILINEAR, B, b"".join(cams), ts, dt, 0, C.c_void_p(frames_t.data_ptr()))
                    assert rc == 0, L.vk_group_last_error(g)
                    assert L.vk_group_sync(g) == 0
                    got = frames_t.cpu().numpy()
                    assert (got.view(np.uint32) == want.view(np.uint32)).all(), ("vk_group_render peer-direct", n, root_skip, rep)
                assert stats()[0] == t_g, "peer-direct frames move nothing through RCCL"
                assert L.vk_group_peer_direct(g, 0) == 0
                check(root, L.vk_device_free(root, out))
            finally:
                L.vk_group_destroy(g)
            t1, b1 = stats()
            assert t1 - t0 == 2 * (n - 1), ("one send/recv pair per peer and call", n, t1 - t0)
            assert fake.fake_rccl_unmatched() == 0
            print("vk_group_render n=%d root_skip=%d wire=%s: %d frames bitwise (gathered and peer-direct), %d transfers, %.2f MB moved" % (n, root_skip, "rgb" if wire else "rgba", B, t1 - t0, (b1 - b0) / 1e6))

    # ---- (a) one context per rank: vk_comm_init_rank + vk_gather_tiles (root branch) --------------------------
    # (root: where this gather is assembled -- rank 0, or another rank as BatchTileRenderer(root="rotate") does from launch to launch)
    for n, root_skip, peers_first, wire, root in ((2, 0, False, V.WIRE_RGBA, 0), (3, 2, True, V.WIRE_RGB, 0), (8, 3, False, V.WIRE_RGB, 0),
                                                  (3, 0, True, V.WIRE_RGBA, 2), (8, 0, False, V.WIRE_RGB, 5)):
        ch = 3 if wire == V.WIRE_RGB else 4
        t0, b0 = stats()
        idbuf = (C.c_ubyte * 128)()
        check(None, L.vk_comm_unique_id(idbuf))
        ctxs, comm_streams, march_streams = [], [], []
        for r in range(n):
            march_streams.append(torch.cuda.Stream())
            c = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA32F, stream=march_streams[r].cuda_stream)
            V.VolumeTexture(c, vol)
            check(c.handle, L.vk_comm_init_rank(c.handle, bytes(idbuf), r, n))
            c.set_root_skip(root_skip)
            c.set_wire(wire)
            assert c.wire_pixel_bytes == 4 * ch
            ctxs.append(c)
            comm_streams.append(torch.cuda.Stream())
        rk, nr = C.c_int(), C.c_int()
        check(ctxs[n - 1].handle, L.vk_comm_info(ctxs[n - 1].handle, C.byref(rk), C.byref(nr)))
        assert (rk.value, nr.value) == (n - 1, n)
        cap = V.partition_slots(W, H, ts, n, root_skip)
        pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=dt)
        send = [torch.zeros((cap, B, ts * ts * ch), dtype=torch.float32, device="cuda") for _ in range(n)]
        recv = torch.zeros((n * cap * B, ts * ts * ch), dtype=torch.float32, device="cuda")
        frames = torch.zeros((B, H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        bids, acts = [], []
        for r in range(n):
            bid, act = V.render_batch(ctxs[r], pipe, cams, send[r].data_ptr(), tile_size=ts, rank=r, nranks=n, compact=True, slot_capacity=cap)
            bids.append(bid); acts.append(act)
        assert len(set(acts)) == 1, acts  # every rank derives the same active-slot count from the cameras
        act = acts[0]
        n_px = act * B * ts * ts
        # the gather on a second stream per rank, ordered behind that rank's march by an event, as the driver does it
        order = [r for r in range(n) if r != root] + [root] if peers_first else [root] + [r for r in range(n) if r != root]
        for r in order:
            ev = torch.cuda.Event()
            ev.record(march_streams[r])
            comm_streams[r].wait_event(ev)
            check(ctxs[r].handle, L.vk_gather_tiles(ctxs[r].handle, C.c_void_p(send[r].data_ptr()), C.c_void_p(recv.data_ptr() if r == root else None), n_px, root,
                                                    C.c_void_p(comm_streams[r].cuda_stream)))
        assert fake.fake_rccl_unmatched() == 0
        done = torch.cuda.Event()
        done.record(comm_streams[root])
        march_streams[root].wait_event(done)
        V.untile_batch(ctxs[root], bids[root], recv.data_ptr(), act, frames.data_ptr())
        ctxs[root].sync()
        torch.cuda.synchronize()
        got = frames.cpu().numpy()
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), ("vk_gather_tiles", n, root_skip, root)
        # error behaviour of the gather: a root without a receive buffer, a root out of range
        assert L.vk_gather_tiles(ctxs[root].handle, C.c_void_p(send[root].data_ptr()), None, n_px, root, None) != 0
        assert L.vk_gather_tiles(ctxs[1].handle, C.c_void_p(send[1].data_ptr()), None, n_px, n, None) != 0
        for c in ctxs:
            check(c.handle, L.vk_comm_destroy(c.handle))
            c.close()
        t1, b1 = stats()
        assert t1 - t0 == n - 1
        assert b1 - b0 == (n - 1) * n_px * 4 * ch, "every peer moves its active prefix, in the wire format"
        print("vk_gather_tiles n=%d root_skip=%d root=%d wire=%s (%s): %d frames bitwise, %d transfers, %.2f MB moved"
              % (n, root_skip, root, "rgb" if wire else "rgba", "peers post first" if peers_first else "root posts first", B, t1 - t0, (b1 - b0) / 1e6))
    print("shim_multi_rank_check: OK")


if __name__ == "__main__":
    main()
