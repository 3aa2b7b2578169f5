"""GPU suite, multi-GPU failure behaviour (no second GPU needed): a rank that has lost its peer gets an ERROR within a bounded time, never a
hang -- by a time limit around every wait on a gather (BatchTileRenderer.timeout_s, PeerLostError, vk_comm_abort), and by a limit on
bench.py's N > 1 window."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def V(hip_built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU: the gpu suite must run on an MI355X box")
    import vokselis_amd

    return vokselis_amd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_survivor_of_a_dead_peer_raises_within_its_time_limit(V):
    """Two ranks over gloo on this GPU; rank 1's process ends before the second batch's gather.  Rank 0's submit() / flush() must raise
    PeerLostError (exit code 3 of tests/peer_exit_check.py) well inside a minute."""
    port = _free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    t0 = time.monotonic()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_exit_check.py")], env=dict(env, RANK=str(r)), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=180))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    took = time.monotonic() - t0
    assert procs[1].returncode == 0 and "rank 1 ends now" in outs[1][0], outs[1]
    assert procs[0].returncode == 3, (procs[0].returncode, outs[0][0][-1500:], outs[0][1][-1500:])
    assert "healthy batch delivered" in outs[0][0] and "survivor raised PeerLostError" in outs[0][0]
    assert took < 120, took


def test_stuck_gather_on_the_library_communicator_is_aborted(V):
    """The library's own RCCL communicator (a world of one here), with the communication stream held up for seconds behind a spinning kernel --
    what a gather whose peer never answers looks like from this side: flush() raises PeerLostError at the renderer's time limit, the
    communicator is aborted (vk_comm_abort, not vk_comm_destroy, which would wait), and the context goes on rendering."""
    import torch
    import torch.distributed as dist

    from vokselis_amd.dist import BatchTileRenderer, PeerLostError

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    W, H = 320, 180
    ctx = V.Context(W, H, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        V.VolumeTexture.generate_standin(ctx, (64,) * 3)
        cams = [V.Camera(1.0, 0.5, 1.0 + 0.1 * j, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(2)]
        ctx.set_camera_blob(cams[0])
        pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR)
        got = []
        btr = BatchTileRenderer(ctx, pipe, tile_size=32, batch=2, transport="rccl", timeout_s=0.5, on_batch=lambda first, n, fr: got.append(fr.clone()))
        for c in cams:
            btr.submit(c)
        btr.flush()
        assert len(got) == 1
        want = got[0].cpu().numpy()
        # the "peer that never answers": the stream the gather runs on spins for seconds
        # (a spinning kernel of ~2 s.  Unlike an RCCL kernel waiting for a dead peer it cannot be told to stop, and ncclCommAbort drains the
        # device: the error arrives when the spin ends at the latest -- the bound asserted here -- and at the time limit when the wait is RCCL's)
        with torch.cuda.stream(btr.comm_stream):
            torch.cuda._sleep(int(5e9))
        t0 = time.monotonic()
        with pytest.raises(PeerLostError):
            for c in cams:
                btr.submit(c)
            btr.flush()
        assert 0.45 < time.monotonic() - t0 < 4.0
        rank, nranks = V.native.C.c_int(-1), V.native.C.c_int(-1)
        V.native.check(ctx.handle, V.native.lib().vk_comm_info(ctx.handle, V.native.C.byref(rank), V.native.C.byref(nranks)))
        assert nranks.value == 0  # aborted: no communicator left
        with pytest.raises(PeerLostError):
            btr.submit(cams[0])  # a renderer that lost a peer stays dead
            btr.submit(cams[1])
        btr.close()
        torch.cuda.synchronize()  # (the spin ends)
        # the context itself is intact: a new renderer on it delivers the same frames
        got.clear()
        b2 = BatchTileRenderer(ctx, pipe, tile_size=32, batch=2, transport="rccl", timeout_s=20.0, on_batch=lambda first, n, fr: got.append(fr.clone()))
        for c in cams:
            b2.submit(c)
        b2.flush()
        b2.close()
        assert len(got) == 1 and (got[0].cpu().numpy().view(np.uint16) == want.view(np.uint16)).all()
    finally:
        ctx.close()
        if created:
            dist.destroy_process_group()


def test_bench_two_ranks_with_a_killed_rank_ends_with_an_error(V):
    """`bench.py --gpus 2` (rehearsal: both ranks on this GPU over gloo) in which rank 1 ends in the middle of the run: the job must END,
    with a non-zero exit code and a message -- by the gloo error on the survivor, the launcher's tear-down, or bench.py's own limit on the
    N > 1 window (VK_BENCH_WINDOW_LIMIT_S) -- instead of parking rank 0 in a barrier."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VK_BENCH_REHEARSAL")}
    env.update(VK_BENCH_TEST_END_RANK="1", VK_BENCH_WINDOW_LIMIT_S="40")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-rotate", "--batch", "16"]
    t0 = time.monotonic()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    took = time.monotonic() - t0
    assert r.returncode != 0, r.stdout[-1500:]
    assert took < 300, took
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert not lines or "value" not in json.loads(lines[-1]), "a job that lost a rank must not print a result line"
    assert "VK_BENCH_TEST_END_RANK" in r.stderr or "did not complete within" in r.stderr or "PeerLostError" in r.stderr or "Connection closed" in r.stderr, r.stderr[-3000:]
