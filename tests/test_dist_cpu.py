"""CPU suite, part 3: the N > 1 path on the gloo backend, world_size 2 (one process per rank).
Each rank fills its compact batch buffer ([slot][frame][ts][ts], the layout vk_render_batch writes) from the
oracle's frames (test double for the HIP march), the production TorchTileGather moves the active prefix, and the
root un-tiles with the numpy statement of vk_untile_batch: the result must be the oracle frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_weighted_deal_is_a_bijection():
    """deal_pos / deal_owner (Python mirror of vk_common.hpp): every position has exactly one (rank, slot), slots are
    dense per rank, the root's share shrinks to (k-1)/k of a peer's."""
    from vokselis_amd import dist as D

    for world in (1, 2, 3, 4, 8):
        for k in (0, 2, 3, 5, 16):
            if world == 1 and k:
                continue
            tiles = 510
            seen = {}
            for q in range(tiles):
                r, sl = D.deal_owner(q, world, k)
                assert 0 <= r < world and D.deal_pos(r, sl, world, k) == q
                assert (r, sl) not in seen
                seen[(r, sl)] = q
            rounds = D.deal_rounds(tiles, world, k)
            per_rank = [sorted(sl for (r, sl) in seen if r == rr) for rr in range(world)]
            for rr in range(world):
                assert per_rank[rr] == list(range(len(per_rank[rr]))) and len(per_rank[rr]) <= rounds
            assert rounds - 1 <= max(len(p) for p in per_rank) <= rounds  # (the last round may hold the root's tile only)
            if k >= 2 and world > 1:
                assert abs(len(per_rank[0]) / len(per_rank[1]) - (k - 1) / k) < 0.05


def _worker(rank, world, port, W, H, ts, q):
    import torch.distributed as dist

    from oracle import oracle as O
    from vokselis_amd import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol = O.volume_standin_u8(32)
        cam = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
        dts = (1.0, 0.5, 0.75)  # a batch of three frames
        B = len(dts)
        tx, _ = D.tiles_xy(W, H, ts)
        k = 3  # the root sits out every third round of the deal
        cap = D.n_slots(W, H, ts, world, k)
        send = torch.zeros((cap, B, ts, ts, 4), dtype=torch.float32)
        for b, dt in enumerate(dts):
            for j, t in enumerate(D.local_tiles(W, H, ts, rank, world, root_skip=k)):
                x0, y0 = (t % tx) * ts, (t // tx) * ts
                full, _, _ = O.render(cam, vol, W, H, dt_scale=dt, tile=(x0, y0, ts, ts), want_counts=False)
                tile = full[y0:y0 + ts, x0:x0 + ts]
                send[j, b, :tile.shape[0], :tile.shape[1]] = torch.from_numpy(np.ascontiguousarray(tile))
        tg = D.TorchTileGather(root=0)
        recv = torch.zeros((world, cap, B, ts, ts, 4), dtype=torch.float32) if rank == 0 else None
        tg.gather(send, recv)
        if rank == 0:
            frames = D.untile_batch_reference(recv.numpy(), W, H, ts, root_skip=k)
            for b, dt in enumerate(dts):
                ref, _, _ = O.render(cam, vol, W, H, dt_scale=dt, want_counts=False)
                q.put((b, float(np.abs(frames[b] - ref).max()), bool((frames[b] == ref).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,H,ts", [(96, 64, 16), (72, 40, 32)])
def test_gloo_world2_tile_gather(W, H, ts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, ts, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(3))
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[2] for g in got), got  # bit-identical to the single-process oracle frames


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...`, typed exactly like the N = 1 line (no torchrun, no WORLD_SIZE): bench.py must start its
    two ranks itself, as a CHILD torch.distributed.run, before anything touches a GPU.  This container has none, so each rank
    must then stop with the no-GPU message (there is no CPU fallback) and the parent must leave with the child's non-zero code --
    which proves the launch path end to end without a device.  (On the GPU box the same line runs to its JSON:
    tests/test_parity_gpu.py::test_bench_multi_rank_flow_rehearsal[plain].)"""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu suite")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    # the ranks did start: each stops with the message -- or is ended by the launcher because its peer already had (then the launcher's
    # report names both ranks)
    said = r.stderr.count("bench.py needs an MI355X: no GPU visible")
    assert said >= 2 or (said >= 1 and "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr), r.stderr[-2000:]
    assert "must be launched with torch.distributed.run" not in r.stderr


def test_bench_window_rule():
    """bench.py times EXACTLY K steps; a step is one launch of `batch` frames (every frame its own camera), so the window is one contiguous
    run of K * batch frames -- >= 100 at the default batch sizes for the driver's K = 20."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "n_launch = K" in src and "timed_frames = K * batch" in src
    for cfg_batch in (128, 4):
        assert 20 * cfg_batch >= 80


def test_fake_rccl_builds_and_exports_the_bound_symbols():
    """The single-process stand-in for RCCL the gpu suite binds through VK_RCCL_LIB (tests/fake_rccl.cpp) builds here and
    exports the ten entry points vk_comm.hip resolves (no call without a GPU)."""
    import ctypes as C

    import __graft_entry__ as g

    if not (os.path.isdir("/opt/rocm/include/hip") and os.path.exists("/opt/rocm/lib/libamdhip64.so")):
        pytest.skip("no ROCm headers / runtime on this box: the stand-in links libamdhip64")
    lib = C.CDLL(g.build_fake_rccl())
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                 "ncclGetErrorString", "fake_rccl_stats", "fake_rccl_unmatched"):
        assert getattr(lib, name) is not None
    src = open(os.path.join(ROOT, "vokselis_amd", "csrc", "vk_comm.hip")).read()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString"):
        assert 'sym("%s")' % name in src, name
