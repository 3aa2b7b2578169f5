#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mray-steps/s of the volume raycast on the
256^3 uint8 bonsai (stand-in) at 1920x1080, dt_scale 0.5 ("512 steps/ray"), plus the achieved
fraction of the HBM-read roofline (BASELINE.json / SURVEY.md 8d, config C2).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(typed without torchrun, `--gpus N` with N > 1 starts its own ranks as a child torch.distributed.run and relays the line)

A "step" is one LAUNCH: one pass of the raycast over a batch of `batch` frames, every frame its own camera (vk_render_batch; the
reference keeps frames in flight through its queue, src/lib.rs:178-194).  K steps are timed, exactly.  At N = 1 the launch writes
whole frames.  At N > 1 the SAME frames are partitioned: 64x64-pixel tiles dealt heaviest-first over the ranks, every rank marches its
tiles of the batch in one launch, one RCCL gather (the library's own communicator, vk_gather_tiles) brings them to rank 0 over xGMI on
a second stream while the next batch is marched, rank 0 un-tiles (scaling: strong -- the frames are fixed).
`latency` holds the other shape of the same workload, the reference's own: ONE vk_render per frame, each frame a camera the caller did
not know a frame earlier -- on one stream, and with 2 / 3 / 4 frames in flight (vk_ctx_frames_in_flight), with and without the present pass.

value   = S_ref * M / t  [Mray-steps/s]: S_ref = loop iterations the reference shader executes for one frame (with its
          alpha >= 0.95 early-out), counted by the kernel itself in untimed counting launches (the mean over the
          launch's cameras) and equal to the oracle's count (tests).  Volume resident in HBM.  ONE contiguous timed
          window of exactly K steps (M = K * batch frames; `timed_frames`, `launches_per_region`), the same rule at
          every N.  The window is repeated 3 times and the median reported (`repeats`, `repeat_ms_per_step` in run
          order); an untimed pre-roll of the same path brings the GPU to its sustained clocks first (`preroll_frames`).
          The frames of a launch are consecutive frames of an orbit around the config's camera (yaw step 2 pi / 1024),
          every frame its own camera; the same launch with ONE camera repeated is reported beside it (`still_camera`).
roofline: algorithmic bytes of one launch = batch * (S_sampled * B_step + W*H * 8 B), B_step = 8 B (8 u8 taps) or 16 B
          (f16), over the launch's mean duration from HIP events on the launch stream, against 8 TB/s.
          See DESIGN.md "Measurement".
cpu_baseline: the C oracle (oracle/, a port of the reference WGSL -- the reference's wgpu/Vulkan path cannot run:
          no Rust, no Vulkan ICD) timed on the host cores for the same C2 frame.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TILE = 64


def tile_of(cfg) -> int:
    """Tile size of the batched launches: 64 (C2: 0.0689 / 0.0647 / 0.0671 ms per frame at 32 / 64 / 128; C5 the same at 32 and 64); the f16
    staged march gains from finer tiles (C4, four frames per launch: 1.50 ms per frame at 32 against 1.556 at 64)."""
    return 32 if cfg["fmt"] != "u8" else 64


PROF = "r06"  # prefix of the PMC-derived files under profiles/ this line quotes (tools/prof.sh, tools/pmc_traffic.py, tools/utilisation.py)
# (a window per wave or one per 256-thread group of four waves, chosen per launch: vk_launch_staged.hip -- C5 and 4-frame C4 launches take the group kernel)
STAGED_KERNELS = "vk::raymarch_staged_kernel | vk::raymarch_staged_group_kernel"
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0      # measured float4-copy ceiling, same guide
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMD-32, max clock (same guide); an f32 full-rate wave64 instruction holds a SIMD for 2 cycles,
                               # a half-rate one (f64 fma, cvt, fract, min/max, v_fma_mix, 3-operand integer ops) for 4, a transcendental for 8
                               # (profiles/r03_ubench_valu_issue_rate.txt)
F64_LANE_OPS_PEAK = SIMDS * 16 * CLOCK_HZ  # 39.3 T f64 lane-operations/s = 78.6 TFLOP/s of fma


def valu_floor_frac(steps, valu_per_step, mean_cost, ms):
    """VALU pipe occupancy a kernel cannot be below: (steps / 64 full waves) x vector instructions per step x their mean issue cost, over
    the SIMD cycles of the launch at the maximum clock.  Partly empty waves and a lower sustained clock both raise the true figure
    (PMC-based: profiles/r03_utilisation.txt)."""
    return steps / 64.0 * valu_per_step * mean_cost / (SIMDS * CLOCK_HZ * ms * 1e-3)
B_RAY = 8                  # rgba16f per ray

# BASELINE.json configs that fit one GPU: volume edge, format, image, seed, bytes per step
CONFIGS = {
    "c2": dict(n=256, fmt="u8", W=1920, H=1080, kind="standin", seed=0x5EED0001, b_step=8,
               name="C2: bonsai stand-in 256^3 uint8 (device-generated, seed 0x5EED0001), 1920x1080, bonsai camera (1,.5,1,(.5,.5,.5)), "
                    "NAIVE_TRILINEAR, dt_scale 0.5 (<=513 steps/ray), rgba16f out",
               metric="Mray-steps/s on 256^3 uint8 @1920x1080; achieved % HBM-read roofline"),
    "c4": dict(n=1024, fmt="f16", W=1920, H=1080, kind="fog", seed=0x5EED0004, b_step=16,
               name="C4: fog 1024^3 fp16 (device-generated, seed 0x5EED0004), 1920x1080, bonsai camera, NAIVE_TRILINEAR, dt_scale 0.5 "
                    "(<=2049 steps/ray), rgba16f out",
               metric="Mray-steps/s on 1024^3 fp16 @1920x1080 (C4); achieved % HBM-read roofline"),
    "c5": dict(n=2048, fmt="u8", W=3840, H=2160, kind="fog", seed=0x5EED0005, b_step=8,
               name="C5: fog 2048^3 uint8 (device-generated, seed 0x5EED0005), 3840x2160, bonsai camera, NAIVE_TRILINEAR, dt_scale 0.5 "
                    "(<=4097 steps/ray), rgba16f out",
               metric="Mray-steps/s on 2048^3 uint8 @3840x2160 (C5); achieved % HBM-read roofline"),
}
DT_SCALE = 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step is one launch of --batch frames (every frame its own camera)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed warm-up steps (launches) before the window")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="BASELINE config to run (the metric is quoted on c2)")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch (default 128 for c2 -- 32 N at N > 1, at least 128, at most 256 --, 4 for c4 and c5)")
    ap.add_argument("--no-skip", action="store_true", help="disable exact empty-space skipping in the timed path")
    ap.add_argument("--fast-walk", action="store_true", help="run the timed path in VK_RENDER_FAST_WALK (tolerance mode: skips advance in closed form; not bit-exact)")
    ap.add_argument("--layout", default="auto", choices=["auto", "pairs", "packed", "bricked", "staged"], help="volume layout (auto: the library's choice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements")
    ap.add_argument("--no-preroll", action="store_true", help="skip the untimed pre-roll that brings the GPU to its sustained clocks before the W warm-up frames")
    ap.add_argument("--no-rotate", action="store_true", help="N > 1: skip the second window with a rotating root (rotating_root in the JSON line)")
    ap.add_argument("--headline-only", action="store_true", help="only the headline's launches (no still-camera window, no single-frame launches): what tools/prof.sh profiles, so that a kernel's mean duration under rocprofv3 is the headline launch's")
    ap.add_argument("--force-dist", action="store_true", help="use the partition + gather driver even at N = 1 (the like-for-like baseline of the N > 1 lines)")
    return ap.parse_args()


def time_launches(ctx, fn, iters, warm=3, groups=3, min_warm_ms=40.0):
    """Mean duration of one call of fn() from HIP events on the launch stream: the best of `groups` back-to-back groups
    of `iters` calls, after `warm` calls and as many more as fill `min_warm_ms` of GPU time (the clocks need ~10 ms of
    sustained load to settle, and a side measurement may follow seconds of host-only work)."""
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_begin()
    fn()
    ctx.timer_end()
    one = max(ctx.timer_elapsed_ms(), 1e-3)
    for _ in range(min(200, int(min_warm_ms / one))):
        fn()
    ctx.sync()
    best = None
    for _ in range(groups):
        ctx.timer_begin()
        for _ in range(iters):
            fn()
        ctx.timer_end()
        ms = ctx.timer_elapsed_ms() / iters
        best = ms if best is None else min(best, ms)
    return best


def count_steps(ctx, V, flags):
    ctx.reset_step_counts()
    V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags | V.RENDER_COUNT).record(ctx)
    return ctx.step_counts()


def effective_cpus() -> int:
    """Host threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(blob):
    """The oracle (CPU port of the reference shader) on the C2 frame, all host threads."""
    import numpy as np

    from oracle import oracle as O

    O.build()
    W, H, n = 1920, 1080, 256
    vol = O.volume_standin_u8(n)
    threads = effective_cpus()
    O.render(blob, vol, W, H // 8, dt_scale=DT_SCALE, threads=threads, want_counts=False)  # page in
    times, s_ref = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        _, steps, _ = O.render(blob, vol, W, H, dt_scale=DT_SCALE, threads=threads)
        times.append(time.perf_counter() - t0)
        s_ref = int(steps.sum())
    t_all = float(np.median(times))
    t0 = time.perf_counter()
    _, st1, _ = O.render(blob, vol, W, H, dt_scale=DT_SCALE, threads=1, tile=(0, H // 2 - 32, W, 64))
    t_one = time.perf_counter() - t0
    # C1 (512x512, dt_scale 1), the reference's own CPU-runnable case, on the same threads (C1 and C2)
    blob1 = O.camera_blob(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), 1.0)
    t1s, s1 = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        _, steps1, _ = O.render(blob1, vol, 512, 512, dt_scale=1.0, threads=threads)
        t1s.append(time.perf_counter() - t0)
        s1 = int(steps1.sum())
    return {
        "c1_value": s1 / float(np.median(t1s)) / 1e6, "c1_sample": "3 full C1 frames (512x512, dt_scale 1; median), same threads", "c1_s_ref": s1,
        "value": s_ref / t_all / 1e6, "unit": "Mray-steps/s", "cores": threads, "kind": "port",
        "sample": f"3 full C2 frames (median), OpenMP dynamic over rows, {threads} threads; oracle/vokselis_oracle.c",
        "one_thread_value": int(st1.sum()) / t_one / 1e6, "one_thread_sample": "rows 508..571 of the C2 frame, 1 thread",
        "s_ref": s_ref,
    }, s_ref


def make_volume(V, ctx, cfg, layout):
    fmt = V.FMT_R8_UNORM if cfg["fmt"] == "u8" else V.FMT_R16_FLOAT
    t0 = time.perf_counter()
    if cfg["kind"] == "standin":
        V.VolumeTexture.generate_standin(ctx, (cfg["n"],) * 3, layout=layout)
    else:
        V.VolumeTexture.generate_fog(ctx, (cfg["n"],) * 3, fmt=fmt, seed=cfg["seed"], layout=layout)
    ctx.sync()
    return time.perf_counter() - t0


def big_config_extra(V, torch, local_rank, key, frames=6):
    """One BASELINE config on this GPU, single-frame launches: set-up time, step counts, launch time, roofline fraction."""
    cfg = CONFIGS[key]
    W, H = cfg["W"], cfg["H"]
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)
    ctx = V.Context(W, H, cam, device=local_rank, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
    try:
        setup = make_volume(V, ctx, cfg, V.LAYOUT_AUTO)
        ctx.update()
        s_ref, s_samp = count_steps(ctx, V, 0)
        p = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE)
        ms = time_launches(ctx, lambda: p.record(ctx), frames, warm=2)
        alg = s_samp * cfg["b_step"] + W * H * B_RAY
        gb = alg / (ms * 1e-3) / 1e9
        # the same frames several per launch (vk_render_batch), as the headline is run: a single launch of a few rounds of
        # LDS-limited waves loses its tail
        nb = 4
        fr = torch.empty((nb, H, W, 4), dtype=torch.float16, device="cuda")
        # consecutive frames of an orbit (yaw step 2 pi / 1024), not one camera repeated: identical frames in one launch
        # share their brick fetches in L2 / Infinity Cache and run up to 20 % faster than any real frame stream
        # (docs/archive/tools/big_batch_orbit.py); unrelated views in one launch gain nothing over single launches
        blobs = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(nb)]
        ms_b = time_launches(ctx, lambda: V.render_batch(ctx, p, blobs, fr.data_ptr(), tile_size=tile_of(cfg)), 3, warm=1) / nb
        del fr
        # ... and one frame per vk_render, the camera turning, on one stream and with three / four frames in flight (vk_ctx_frames_in_flight)
        stream_ms = {}
        try:
            orbit = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(16)]
            nfr = 32 if key == "c4" else 12
            for k in (0, 3, 4):
                if k:
                    ctx.frames_in_flight(k)
                stream_ms["one_stream" if k == 0 else "in_flight_%d" % k] = frame_stream_ms(ctx, p, orbit, nfr, k)
            ctx.frames_in_flight(1)
        except Exception as e:  # noqa: BLE001
            stream_ms = {"error": repr(e)}
        ctx.set_camera_blob(cam.get_proj_view_matrix())
        dims = (V.native.C.c_uint32 * 3)()
        lay, nbytes = V.native.C.c_int(), V.native.C.c_size_t()
        V.native.check(ctx.handle, V.native.lib().vk_volume_info(ctx.handle, dims, None, V.native.C.byref(lay), V.native.C.byref(nbytes)))
        # the config's "dense-core variant" (SURVEY 8d): the same fog with a dense ball at the centre -- rays through the
        # middle of the image leave by the opacity early-out, so S_ref < S_nominal and waves run partly empty
        fmt = V.FMT_R8_UNORM if cfg["fmt"] == "u8" else V.FMT_R16_FLOAT
        V.VolumeTexture.generate_fog(ctx, (cfg["n"],) * 3, fmt=fmt, seed=cfg["seed"], layout=V.LAYOUT_AUTO, dense_core=True)
        c_ref, c_samp = count_steps(ctx, V, 0)
        ms_c = time_launches(ctx, lambda: p.record(ctx), frames, warm=2)
        alg_c = c_samp * cfg["b_step"] + W * H * B_RAY
        core = {"launch_ms": ms_c, "s_ref": c_ref, "s_sampled": c_samp, "Mray_steps_per_s": c_ref / ms_c / 1e3,
                "algorithmic_frac": alg_c / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS}
        phys = {}
        try:  # physical HBM side of the same kernel from the PMC passes (algorithmic bytes are served from LDS: their fraction says nothing about HBM)
            uj = json.load(open(os.path.join(ROOT, "profiles", PROF + "_utilisation.json"))).get(key, {})
            if "hbm_bytes_per_launch" in uj:
                phys = {"physical_hbm": {"measured_in_this_run": False, "bytes_per_launch": uj["hbm_bytes_per_launch"], "frac_of_peak_under_rocprof": uj["hbm_frac_of_peak"],
                                         "frac_of_peak_at_this_launch_ms": uj["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "refetch_factor": uj.get("refetch_factor"), "valu_pipe_occupancy": uj.get("valu_pipe_occupancy"),
                                         "source": "profiles/%s_utilisation.json (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes; refetch = bytes fetched / dense volume bytes)" % PROF}}
        except Exception:
            pass
        staged = lay.value == 6
        # A staged march serves its taps from LDS: the algorithmic bytes (8 taps per step) are not HBM traffic and their rate over the HBM peak is
        # not bounded by 1 -- it is named `algorithmic_frac`, never `frac`, and the physical side (bytes the counters saw) comes first.
        fkey = "algorithmic_frac" if staged else "frac"
        return {"workload": cfg["name"], **phys, "launch_ms": ms, "s_ref": s_ref, "s_sampled": s_samp, "Mray_steps_per_s": s_ref / ms / 1e3,
                "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": gb, fkey: gb / HBM_PEAK_GBS,
                **({"algorithmic_frac_note": "8 B (u8) / 16 B (f16) per step over 8 TB/s: an accounting of the taps the LDS windows serve, not a bound; the HBM side is `physical_hbm`"} if staged else {}),
                "volume_setup_s": setup, "dense_core": core,
                "batch": {"frames_per_launch": nb, "cameras": "consecutive frames of an orbit, yaw step 2pi/1024", "ms_per_frame": ms_b, "Mray_steps_per_s": s_ref / ms_b / 1e3, fkey: alg / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "frame_stream_ms_per_frame": stream_ms,
                "layout": {6: "staged 8^3 bricks through LDS, 3 copies", 4: "dense 9^3 bricks", 3: "cells, f16 pairs", 2: "cells"}.get(lay.value, str(lay.value)),
                "volume_device_bytes": nbytes.value, "kernel": STAGED_KERNELS if lay.value == 6 else "vk::raymarch_naive_kernel"}
    finally:
        ctx.close()


def c5_at_n(world, timeout_s=240):
    """BASELINE's own 8-GPU configuration (C5: 2048^3 u8, 3840x2160) through the N > 1 driver -- the same partition, gather and un-tile as the
    headline's C2 frames -- as a CHILD `bench.py --gpus N --config c5` with a time limit, started by rank 0 after this job's own ranks have left their
    process group: a second job full of collectives must not be able to take the headline's line down (a hang or a crash in it costs `extras.c5_at_n`
    an error string, nothing else).  Returns the `extras.c5_at_n` object."""
    import signal
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                        "MASTER_ADDR", "MASTER_PORT", "VK_BENCH_TRANSPORT") and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--config", "c5", "--steps", "8", "--warmup", "2", "--no-extras", "--no-cpu-baseline"]
    try:
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            so, se = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            return {"error": "the C5 job did not finish within %d s" % timeout_s}
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            return {"error": "the C5 job failed (rc %d): %s" % (proc.returncode, se[-400:])}
        d = json.loads(lines[-1])
        res = {"workload": d["config"]["workload"], "n_gpus": d["n_gpus"], "frames_per_launch": d["frames_per_launch"], "timed_frames": d["timed_frames"],
               "s_ref_per_frame": d["config"]["s_ref_config_camera"], "volume_setup_s": d.get("volume_setup_s"),
               "fixed_root": {"value": d["value"], "unit": d["unit"], "ms_per_frame": d["ms_per_frame"], "frac": d["roofline"]["frac"], "root_skip": d["config"].get("root_skip")},
               "note": "`bench.py --gpus %d --config c5` as a child job after this one's ranks left their process group: C5 frames through the same partition + "
                       "gather + un-tile as the headline's (64 x 64-pixel tiles dealt heaviest-first, volume replicated: 26 GB of 288 per GPU)" % world}
        if "rotating_root" in d:
            rr = d["rotating_root"]
            res["rotating_root"] = rr if "error" in rr else {"value": rr["value"], "unit": rr["unit"], "ms_per_frame": rr["ms_per_frame"]}
        if "rehearsal" in d:
            res["rehearsal"] = d["rehearsal"]
        return res
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def self_launch(args):
    """`python bench.py --gpus N` typed like the N = 1 line: start the N ranks as a CHILD torch.distributed.run (never an exec,
    and before anything in this process has touched the GPU), relay its output, leave with its exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import torch
        n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    except Exception:
        n_dev = 0
    if 0 < n_dev < args.gpus and env.get("VK_BENCH_REHEARSAL", "") != "1":
        if args.gpus > 6:
            sys.exit("bench.py --gpus %d: this box has %d GPU(s) and a rehearsal on one GPU takes at most 6 ranks" % (args.gpus, n_dev))
        print("[bench] %d GPU(s) visible for --gpus %d: REHEARSAL (every rank on GPU 0, tiles over gloo) -- a test of the flow, not a measurement"
              % (n_dev, args.gpus), file=sys.stderr)
        env["VK_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT)  # stdout / stderr inherited: the JSON line stays the last line
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    sys.exit(rc)


def single_process_group(n_gpus, batch, s_ref, timeout_s=150):
    """The same frames through the single-process host (vokselis_amd/host/bonsai --gpus N: one context per GPU inside a vk_group), gathered
    over RCCL and with peer-direct stores (every GPU writes GPU 0's frames itself: no gather, no un-tile), as CHILD processes with a
    time limit -- a path no one-GPU box can rehearse must not be able to take the line down.  Rank 0 only, after the timed windows."""
    import re
    import subprocess

    import __graft_entry__ as G

    out = {"note": "compiled host, one process, one context per GPU (vk_group_render), %d orbit frames per launch, wall time over >= 1024 frames; "
                   "the ranks of this run idle meanwhile" % batch}
    try:
        G.build_host()
    except Exception as e:  # noqa: BLE001
        return {"error": "host build failed: %r" % (e,)}
    exe = os.path.join(ROOT, "vokselis_amd", "_lib", "bonsai")
    frames = max(1024, 4 * batch)
    for name, extra in (("peer_direct", ["--peer-direct"]), ("gathered", [])):
        cmd = [exe, "--gpus", str(n_gpus), "--frames", str(frames), "--batch", str(batch), "--size", "1920x1080", "--dt", str(DT_SCALE)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            m = re.search(r"Avg frame time ([0-9.]+)ms", r.stdout)
            if r.returncode == 0 and m:
                ms = float(m.group(1))
                out[name] = {"ms_per_frame": ms, "value": s_ref / ms / 1e3, "unit": "Mray-steps/s"}
            else:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
        except subprocess.TimeoutExpired:
            out[name] = {"error": "no result within %d s" % timeout_s}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


def frame_stream_ms(ctx, pipe, cams, n, in_flight, present=False):
    """Wall time per frame of the one-vk_render-per-frame surface over n frames, every frame a camera of `cams` in turn (set just before the
    frame is recorded, as Context::update does).  in_flight = 0: plain vk_render calls on the context's one stream; k >= 1: inside
    vk_frame_begin / vk_frame_end on a ring of k surfaces.  Best of three windows after an untimed one."""
    def window(m):
        for j in range(m):
            ctx.set_camera_blob(cams[j % len(cams)])
            if in_flight:
                ctx.frame_begin()
            pipe.record(ctx)
            if present:
                ctx.render()
            if in_flight:
                ctx.frame_end()

    window(min(n, 64))
    ctx.sync()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        window(n)
        ctx.sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        best = ms if best is None else min(best, ms)
    return best


def latency_section(V, torch, ctx, stream, cfg, flags, s_ref_still, s_sampled_still, cam_list, s_ref_orbit, s_sampled_orbit):
    """The reference's own submission model (one pass per RedrawRequested, src/lib.rs:178-181) on the headline's workload: the duration of
    one single-frame launch on the config's camera, and the frame rate of a STREAM of such launches -- every frame a new camera of the
    headline's orbit -- on one stream and with frames in flight (vk_ctx_frames_in_flight), bare and followed by the present pass
    (two passes, and fused into the raycast's epilogue: VK_RENDER_PRESENT)."""
    W, H = cfg["W"], cfg["H"]
    n1 = 100 if cfg["n"] <= 256 else 24
    alg_still = s_sampled_still * cfg["b_step"] + W * H * B_RAY
    alg_orbit = s_sampled_orbit * cfg["b_step"] + W * H * B_RAY
    p1 = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags)
    evs1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n1)]
    for _ in range(5):
        p1.record(ctx)
    for a, b in evs1:
        a.record(stream)
        p1.record(ctx)
        b.record(stream)
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in evs1)
    ms1 = sum(d) / len(d)
    gb = alg_still / (ms1 * 1e-3) / 1e9
    lat = {"submission": "one vk_render per frame (the reference's model: one pass per RedrawRequested, src/lib.rs:178-181)",
           "launch_ms": ms1, "launch_ms_p10": d[n1 // 10], "launch_ms_p50": d[n1 // 2], "launch_ms_p90": d[(9 * n1) // 10],
           "value": s_ref_still / ms1 / 1e3, "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS,
           "note": "launch_ms: HIP events around each of %d back-to-back single-frame launches on the config's camera (kernel time; frames bitwise "
                   "those of the batched launches)" % n1}
    if cfg["n"] <= 256 and not (flags & (V.RENDER_NO_SKIP | V.RENDER_FAST_WALK)):
        pf = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags | V.RENDER_FAST_WALK)
        ms_f = time_launches(ctx, lambda: pf.record(ctx), 50, warm=10)
        lat["fast_walk"] = {"launch_ms": ms_f, "value": s_ref_still / ms_f / 1e3, "note": "the same launch in tolerance mode (VK_RENDER_FAST_WALK)"}
    # the stream of frames: wall time per frame, every frame its own orbit camera, set right before the frame is recorded
    n = 256 if cfg["n"] <= 256 else 16
    try:
        ctx.set_stream(None)  # (a ring's surfaces run on streams of the context's own)
        pf_ = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags | V.RENDER_PRESENT)
        stream_of = {"frames": n, "unit": "ms per frame, wall time, every frame a new camera of the headline's orbit",
                     "raycast": {}, "raycast_then_present": {}, "raycast_present_fused": {}}
        for k in (0, 2, 3, 4):
            if k:
                ctx.frames_in_flight(k)
            key = "one_stream" if k == 0 else "in_flight_%d" % k
            stream_of["raycast"][key] = frame_stream_ms(ctx, p1, cam_list, n, k)
            stream_of["raycast_then_present"][key] = frame_stream_ms(ctx, p1, cam_list, n, k, present=True)
            stream_of["raycast_present_fused"][key] = frame_stream_ms(ctx, pf_, cam_list, n, k)
        # the recorder's loop (src/lib.rs:196-199: capture_frame of every frame): the presented Rgba8 image of every frame read back over
        # PCIe into pinned host memory, three frames behind the one being recorded (K = 4, present fused) -- the PCIe-INCLUSIVE rate
        try:
            import ctypes as C

            dims = V.ImageDimentions.new(W, H, 256)
            host = torch.empty(dims.linear_size(), dtype=torch.uint8, pin_memory=True)
            ids = []

            def recorder_window(m):
                for j in range(m):
                    ctx.set_camera_blob(cam_list[j % len(cam_list)])
                    ids.append(ctx.frame_begin())
                    pf_.record(ctx)
                    ctx.frame_end()
                    if len(ids) > 3:
                        V.native.check(ctx.handle, V.native.lib().vk_frame_capture(ctx.handle, ids[-4], C.c_void_p(host.data_ptr()), host.numel(), None, None, None))
                ctx.sync()

            recorder_window(16)
            t0 = time.perf_counter()
            recorder_window(64)
            ms_rec = (time.perf_counter() - t0) / 64 * 1e3
            stream_of["recorder_loop"] = {"ms_per_frame": ms_rec, "host_GBps": dims.linear_size() / (ms_rec * 1e-3) / 1e9,
                                          "note": "in_flight_4, present fused, vk_frame_capture of every frame (8.3 MB of Rgba8 per 1080p frame) into pinned host memory, "
                                                  "three frames behind: bound by the copy over PCIe, the raycast hides under it"}
        except Exception as e:  # noqa: BLE001
            stream_of["recorder_loop"] = {"error": repr(e)}
        ctx.frames_in_flight(1)
        best = min(stream_of["raycast"], key=lambda k_: stream_of["raycast"][k_])
        msb = stream_of["raycast"][best]
        stream_of["best"] = {"shape": best, "ms_per_frame": msb, "value": s_ref_orbit / msb / 1e3, "frac": alg_orbit / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS}
        lat["in_flight"] = stream_of
    finally:
        ctx.frames_in_flight(1)
        ctx.set_stream(stream.cuda_stream)
    return lat


class Run(types.SimpleNamespace):
    """What one bench run threads through its stages: the plan (ranks, config, frames per launch, K), the process (torch, the process group, the
    library), the workload (context, volume, cameras, unit counts), the submitter of the timed windows and their timings."""


def plan(args):
    """Ranks, workload and window of this run: a STEP is one launch of `batch` frames, K steps are timed."""
    R = Run()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 flow on a box with one GPU (tests/test_parity_gpu.py): every rank on device 0, rendezvous over
    # gloo, tiles through torch.distributed.  Not a measurement -- the JSON says so.
    rehearsal = os.environ.get("VK_BENCH_REHEARSAL", "") == "1"
    if rehearsal:
        local_rank = 0
        os.environ["VK_BENCH_TRANSPORT"] = "torch"
    args.gpus = world  # (under a launcher the world it made is what runs)
    cfg = CONFIGS[args.config]
    global TILE
    TILE = tile_of(cfg)
    W, H = cfg["W"], cfg["H"]
    batch = args.batch or {"c2": 128, "c4": 4, "c5": 4}[args.config]  # (C2 per frame at 32 / 64 / 128 / 256 frames per launch: 0.0677 / 0.0659 / 0.0644 / 0.0645 ms)
    if not args.batch and args.config == "c2" and world > 1:
        # a rank's launch covers 1 / N of every frame: more frames per launch keep it from paying the launch's tail N times as
        # often (docs/archive/tools/batch_size_at_n.py, a rank of 8: 14.7 / 11.6 / 9.9 / 8.8 us per frame at 16 / 32 / 64 / 128 frames per launch)
        batch = min(256, max(128, 32 * world))
    batch = max(1, batch)
    # A STEP is one launch: one pass of the raycast over a batch of `batch` frames, every frame its own camera.  The timed window is
    # EXACTLY K steps = K launches = K * batch frames (>= 100 frames, SURVEY 8d, for any K at the default batch sizes of c2); the W
    # warm-up steps are W launches of the same kind.
    K = max(1, args.steps)
    n_launch = K
    timed_frames = K * batch
    if batch > 1024:
        sys.exit("bench.py: --batch is at most 1024 frames per launch (VK_MAX_BATCH_FRAMES)")
    R.H, R.K, R.W, R.args, R.batch, R.cfg, R.local_rank, R.n_launch, R.rank = H, K, W, args, batch, cfg, local_rank, n_launch, rank
    R.rehearsal, R.timed_frames, R.world = rehearsal, timed_frames, world
    return R


def open_process(R):
    """torch, the process group (N > 1: one process per GPU over RCCL; a rehearsal: every rank on GPU 0 over gloo), the library."""
    args, local_rank, rank, rehearsal, world = R.args, R.local_rank, R.rank, R.rehearsal, R.world
    import torch

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as G

    if rank == 0:
        G.build_hip()
    if world > 1:
        dist.barrier()
    import vokselis_amd as V
    from vokselis_amd.dist import BatchTileRenderer

    layout = {"auto": V.LAYOUT_AUTO, "pairs": V.LAYOUT_PACKED_PAIRS, "packed": V.LAYOUT_PACKED, "bricked": V.LAYOUT_BRICKED, "staged": V.LAYOUT_STAGED}[args.layout]
    flags = (V.RENDER_NO_SKIP if args.no_skip else 0) | (V.RENDER_FAST_WALK if args.fast_walk else 0)
    stream = torch.cuda.Stream()
    R.BatchTileRenderer, R.V, R.dist, R.flags, R.layout, R.stream, R.torch, R.use_dist = BatchTileRenderer, V, dist, flags, layout, stream, torch, use_dist


def open_workload(R):
    """Context, volume, the cameras of a launch and the units one frame processes (untimed counting launches).  Runs on R.stream."""
    H, V, W, batch, cfg, flags, layout, local_rank, stream = R.H, R.V, R.W, R.batch, R.cfg, R.flags, R.layout, R.local_rank, R.stream
    cam = V.Camera(1.0, 0.5, 1.0, (0.5, 0.5, 0.5), W / H)  # examples/bonsai/main.rs:68-74
    blob = cam.get_proj_view_matrix()
    ctx = V.Context(W, H, cam, device=local_rank, backbuffer=(W, H), out_format=V.OUT_RGBA16F, stream=stream.cuda_stream)
    info = ctx.get_info()
    t_volume = make_volume(V, ctx, cfg, layout)
    ctx.update()
    pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags)

    # untimed counting launches: the units one frame processes.  The config's own camera first (S_ref of THE frame BASELINE
    # names, equal to the oracle's count) ...
    s_ref_still, s_sampled_still = count_steps(ctx, V, flags)
    # ... then the frames the timed launches march: consecutive frames of an orbit around that camera (yaw step 2 pi /
    # 1024: src/camera.rs turns the camera on input), every frame its own camera.  A launch that repeats ONE camera skips
    # the per-camera host work (tile order, cull rectangle, descriptors) and shares every fetch between its frames in
    # L2 / Infinity Cache -- up to 20 % faster than any real frame stream on the beyond-cache configs
    # (docs/archive/tools/big_batch_orbit.py), 3 % on C2; it is reported beside the headline as `still_camera`.
    cam_list = [V.Camera(1.0, 0.5, 1.0 + 6.28318 * j / 1024, (0.5, 0.5, 0.5), W / H).get_proj_view_matrix() for j in range(batch)]
    tot_ref = tot_samp = 0
    for cb in cam_list:
        ctx.set_camera_blob(cb)
        a, b = count_steps(ctx, V, flags)
        tot_ref += a; tot_samp += b
    ctx.set_camera_blob(blob)
    s_ref, s_sampled = tot_ref / batch, tot_samp / batch  # means over the launch's frames
    R.blob, R.cam_list, R.ctx, R.info, R.pipe, R.s_ref, R.s_ref_still, R.s_sampled, R.s_sampled_still = blob, cam_list, ctx, info, pipe, s_ref, s_ref_still, s_sampled, s_sampled_still
    R.t_volume = t_volume


def make_submitter(R):
    """submit(timed) / flush(timed): one frame into the current launch / launch what is pending.  N = 1: whole frames through vk_render_batch;
    N > 1 (and --force-dist): BatchTileRenderer -- partition, gather, un-tile (R.btr: run_windows swaps it for the rotating-root renderer)."""
    BatchTileRenderer, H, V, W, batch, cam_list, ctx, dist, pipe = R.BatchTileRenderer, R.H, R.V, R.W, R.batch, R.cam_list, R.ctx, R.dist, R.pipe
    rehearsal, stream, torch, use_dist, world = R.rehearsal, R.stream, R.torch, R.use_dist, R.world
    btr = transport = None
    pipe_now = [pipe]      # (swapped for the tolerance-mode run)
    launch_ev = []  # (start, end) HIP events around every batch launch of the timed region
    if not use_dist:
        frames = torch.empty((batch, H, W, 4), dtype=torch.float16, device="cuda")
        cams_now = [cam_list]  # (swapped for the still-camera run)
        pending = [0]

        def submit(timed):
            pending[0] += 1
            if pending[0] == batch:
                flush(timed)

        def flush(timed=False):
            if pending[0] == 0:
                return
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            V.render_batch(ctx, pipe_now[0], cams_now[0], frames.data_ptr(), tile_size=TILE)  # a partial batch is padded to a whole one
            if timed:
                b.record(stream)
                launch_ev.append((a, b))
            pending[0] = 0
    else:
        # tiles move through the library's own RCCL communicator; should that fail to come up on every rank (librccl
        # not loadable, communicator refused) the same buffers go through torch.distributed's -- RCCL as well
        transport, why = os.environ.get("VK_BENCH_TRANSPORT", "rccl"), None
        try:
            btr = BatchTileRenderer(ctx, pipe, tile_size=TILE, batch=batch, root=0, transport=transport, via_host=rehearsal)
            ok = 1
        except Exception as e:  # noqa: BLE001
            btr, ok, why = None, 0, repr(e)
        if world > 1:
            okt = torch.tensor([ok], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = int(okt.item())
        if not ok:
            if btr is not None:
                btr.close()
            if transport == "torch":
                raise RuntimeError("tile gather could not be set up: %s" % why)
            print("[bench] library communicator unavailable (%s): gathering through torch.distributed" % why, file=sys.stderr)
            transport = "torch"
            btr = BatchTileRenderer(ctx, pipe, tile_size=TILE, batch=batch, root=0, transport=transport, via_host=rehearsal)

        btr_i = [0]
        cams_now = [cam_list]

        def submit(timed):
            R.btr.submit(cams_now[0][btr_i[0] % batch]); btr_i[0] += 1

        def flush(timed=False):
            R.btr.flush()
            btr_i[0] = 0
    R.btr, R.cams_now, R.flush, R.launch_ev, R.pipe_now, R.submit, R.transport = btr, cams_now, flush, launch_ev, pipe_now, submit, transport


def make_timed_region(R):
    """timed_region(k, timed): k frames through submit / flush between barriers and synchronisations, the maximum over ranks of the wall time."""
    dist, flush, rank, rehearsal, submit, torch, world = R.dist, R.flush, R.rank, R.rehearsal, R.submit, R.torch, R.world
    region_no = [0]
    # N > 1: a window is full of collectives, and a rank whose peer is gone would sit in one of them for ever (RCCL has no time limit of its
    # own here; torch's watchdog takes ten minutes).  Every window therefore runs under a limit: a rank still inside it after
    # VK_BENCH_WINDOW_LIMIT_S seconds (default 240; a window is a fraction of a second) says so and ends its process with a non-zero
    # code -- which ends the job through the launcher.  No retry from inside: a process that has touched the GPU is never re-executed.
    window_limit_s = float(os.environ.get("VK_BENCH_WINDOW_LIMIT_S", "240"))

    def window_overrun():
        print("[bench] rank %d: a window of the N = %d run did not complete within %.0f s (a peer gone, or a gather stuck): ending this rank"
              % (rank, world, window_limit_s), file=sys.stderr, flush=True)
        os._exit(74)

    def timed_region(k, timed):
        region_no[0] += 1
        if world > 1 and os.environ.get("VK_BENCH_TEST_END_RANK", "") == str(rank) and region_no[0] == 3:
            # TEST HOOK (tests/test_peer_loss_gpu.py): this rank's process ends here, without a word to its peers -- a rank that died
            print("[bench] VK_BENCH_TEST_END_RANK: rank %d ends before window %d" % (rank, region_no[0]), file=sys.stderr, flush=True)
            os._exit(0)
        guard = None
        if world > 1:
            import threading

            guard = threading.Timer(window_limit_s, window_overrun)
            guard.daemon = True
            guard.start()
        try:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                submit(timed)
            flush(timed)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        finally:
            if guard is not None:
                guard.cancel()
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el
    R.timed_region = timed_region


def run_windows(R):
    """Pre-roll, warm-up, the window of K steps (three times, median), and beside it the still-camera window, the tolerance walk, the rotating root."""
    BatchTileRenderer, K, V, args, batch, blob, btr, cam_list, cams_now = R.BatchTileRenderer, R.K, R.V, R.args, R.batch, R.blob, R.btr, R.cam_list, R.cams_now
    ctx, dist, flags, launch_ev, pipe, pipe_now = R.ctx, R.dist, R.flags, R.launch_ev, R.pipe, R.pipe_now
    rehearsal, timed_frames, timed_region = R.rehearsal, R.timed_frames, R.timed_region
    torch, transport, use_dist, world = R.torch, R.transport, R.use_dist, R.world
    # Pre-roll (untimed, every rank, the same path as the timed region): the GPU's clocks take ~10 ms of sustained load
    # to settle -- five back-to-back regions of 20 frames ran 0.097, 0.092, 0.089, 0.086, 0.085 ms per frame in that
    # order -- and the first RCCL transfers set up their channels.  One region of `batch` frames is measured (its
    # maximum over ranks is the same number everywhere), then as many more as fill ~50 ms.  The W warm-up frames follow.
    preroll_frames = 0
    if not args.no_preroll:
        timed_region(batch, False)  # (first use: tables, channels)
        el = timed_region(batch, False)
        n_pre = max(1, min(64, math.ceil(0.05 / max(el, 1e-4))))
        for _ in range(n_pre):
            timed_region(batch, False)
        preroll_frames = (2 + n_pre) * batch
    timed_region(args.warmup * batch, False) if args.warmup else None
    # The window of `timed_frames` frames, three times (a single window is at the mercy of one host hiccup -- a default
    # run on a busy box once reported 27.5 ms of wall time around 20.3 ms of launches); the median is reported.
    repeats = 3

    def measure():
        runs = []
        for _ in range(repeats):
            del launch_ev[:]
            runs.append((timed_region(timed_frames, True), [(a, b) for a, b in launch_ev]))
        order_ms = [r[0] / K * 1e3 for r in runs]  # in the order they ran
        runs.sort(key=lambda r: r[0])
        return runs[len(runs) // 2] + (order_ms,)

    elapsed, evs, run_order_ms = measure()
    root_skip_fixed = btr.root_skip if use_dist else 0
    wire_info = {"format": "rgb" if btr.ch == 3 else "rgba", "bytes_per_pixel": ctx.wire_pixel_bytes, "slot_capacity": btr.cap,
                 "tile_bytes": TILE * TILE * ctx.wire_pixel_bytes} if use_dist else None
    # the same window with the config's ONE camera in every frame (what round 2 reported as the headline)
    still_elapsed, still_evs = None, []
    if not args.headline_only:
        cams_now[0] = [blob] * batch
        timed_region(batch, False)
        still_elapsed, still_evs, _ = measure()
        cams_now[0] = cam_list
    # the same orbit window in VK_RENDER_FAST_WALK (tolerance mode: a skip advances t and p by one fma each; frames within 1e-4 of the
    # bit-exact default at the 99.99th percentile of the pixels, not bit-identical: tests/test_parity_gpu.py::test_fast_walk_tolerance_mode)
    fast_elapsed, fast_evs = None, []
    if not args.headline_only and not use_dist and not args.no_skip and not args.fast_walk and args.config == "c2":
        pipe_now[0] = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=flags | V.RENDER_FAST_WALK)
        timed_region(batch, False)
        fast_elapsed, fast_evs, _ = measure()
        pipe_now[0] = pipe
    # N > 1: the same window once more with a ROTATING root (launch g is assembled on rank g mod N): a fixed root takes 7/8 of every
    # frame over the one link each peer has to it, which at 8 GPUs is slower than the march; rotating spreads the same
    # bytes over every link of the node.  Reported beside `value`, which stays the gather to rank 0.
    rot_elapsed, rot_error = None, None
    if use_dist and world > 1 and not args.no_rotate:
        # Only the SET-UP may fail softly, and only for everybody at once: the ranks agree on its outcome (as for the first renderer
        # above) and drop `rotating_root` together.  The timed window itself is full of collectives; an exception inside it on one
        # rank would leave the others parked in a barrier, so it is not caught: it ends the job through the launcher.
        why = None
        try:
            btr.close()
            btr = R.btr = None
            btr = R.btr = BatchTileRenderer(ctx, pipe, tile_size=TILE, batch=batch, root="rotate", transport=transport, via_host=rehearsal)  # (submit / flush read R.btr)
        except Exception as e:  # noqa: BLE001
            why = repr(e)
        okt = torch.tensor([0 if why else 1], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()):
            timed_region(batch * min(world, 4), False)
            rot_elapsed, _, _ = measure()
        else:
            rot_error = why or "set-up of the rotating-root renderer failed on another rank"
            if btr is not None:
                btr.close()
                btr = None
    R.btr, R.elapsed, R.evs, R.fast_elapsed, R.fast_evs = btr, elapsed, evs, fast_elapsed, fast_evs
    R.preroll_frames, R.repeats, R.root_skip_fixed, R.rot_elapsed = preroll_frames, repeats, root_skip_fixed, rot_elapsed
    R.rot_error, R.run_order_ms, R.still_elapsed, R.still_evs, R.wire_info = rot_error, run_order_ms, still_elapsed, still_evs, wire_info


def headline_line(R):
    """The JSON line's own keys, `config` and `roofline` (rank 0; every rank computes its launch times)."""
    H, K, W, args, batch, cfg, elapsed, evs, fast_elapsed = R.H, R.K, R.W, R.args, R.batch, R.cfg, R.elapsed, R.evs, R.fast_elapsed
    fast_evs, info, n_launch, preroll_frames = R.fast_evs, R.info, R.n_launch, R.preroll_frames
    rank, rehearsal, repeats, root_skip_fixed, rot_elapsed = R.rank, R.rehearsal, R.repeats, R.root_skip_fixed, R.rot_elapsed
    rot_error, run_order_ms, s_ref, s_ref_still = R.rot_error, R.run_order_ms, R.s_ref, R.s_ref_still
    s_sampled, s_sampled_still, still_elapsed, still_evs, t_volume = R.s_sampled, R.s_sampled_still, R.still_elapsed, R.still_evs, R.t_volume
    timed_frames, torch, transport, use_dist, wire_info, world = R.timed_frames, R.torch, R.transport, R.use_dist, R.wire_info, R.world
    out = None
    n_launch_frames = batch  # frames one launch spans
    launch_ms = None
    if evs:
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) for a, b in evs)
        launch_ms = sum(d) / len(d)

    fast_launch_ms = None
    if fast_evs:
        torch.cuda.synchronize()
        fd_ = sorted(a.elapsed_time(b) for a, b in fast_evs)
        fast_launch_ms = sum(fd_) / len(fd_)
    still_launch_ms = None
    if still_evs:
        torch.cuda.synchronize()
        sd = sorted(a.elapsed_time(b) for a, b in still_evs)
        still_launch_ms = sum(sd) / len(sd)

    if rank == 0:
        n_px = W * H
        ms_per_step = elapsed / K * 1e3          # a step: one launch of `batch` frames
        ms_per_frame = elapsed / timed_frames * 1e3
        alg_frame = s_sampled * cfg["b_step"] + n_px * B_RAY
        out = {
            "metric": cfg["metric"],
            "value": s_ref * timed_frames / elapsed / 1e6,
            "unit": "Mray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_frames": timed_frames, "launches_per_region": n_launch, "frames_per_launch": batch,
            "ms_per_step": ms_per_step, "ms_per_frame": ms_per_frame,
            "step": "one launch = %d frames, every frame its own camera" % batch,
            "timed_region_s": float("%.6g" % elapsed),  # wall time of the window of K steps `value` is computed from (the median of the three repetitions)
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": cfg["name"],
                "layout": args.layout,
                "skip": not args.no_skip,
                "walk": "closed form, VK_RENDER_FAST_WALK (tolerance mode, NOT bit-exact)" if args.fast_walk else "the reference's own additions (bit-exact: trip counts identical to the oracle)",
                "frames_per_launch": batch,
                "partition": "one launch per batch of whole frames" if not use_dist else
                             f"{TILE}x{TILE} tiles dealt heaviest-first over {world} ranks, one launch + one RCCL gather (second stream) + one un-tile per batch of {batch} frames",
                "s_ref_per_frame": s_ref, "s_sampled_per_frame": s_sampled, "rays_per_frame": n_px,
                "s_ref_config_camera": s_ref_still, "s_sampled_config_camera": s_sampled_still,
                "cameras": "every frame its own camera: consecutive frames of an orbit around the config's camera, yaw step 2pi/1024 (step counts: mean over the launch's frames)",
                "submission": "batched: %d frames per launch, one camera per frame (vk_render_batch); the one-vk_render-per-frame shape of the same workload: `latency`" % batch,
                "window": "one contiguous window of exactly %d steps = %d launches of %d frames = %d frames" % (K, n_launch, batch, timed_frames),
                **({"transport": "library RCCL communicator (vk_gather_tiles)" if transport == "rccl" else "torch.distributed (RCCL)"} if use_dist else {}),
            },
            **({"rehearsal": "all ranks on ONE GPU over gloo: a test of the N > 1 flow, not a measurement"} if rehearsal else {}),
            "repeats": repeats, "repeat_ms_per_step": run_order_ms, "preroll_frames": preroll_frames,
            "device": info["device_name"], "volume_setup_s": t_volume,
            # the same window with the config's one camera repeated in every frame of every launch
            **({"still_camera": {"ms_per_step": still_elapsed / K * 1e3, "ms_per_frame": still_elapsed / timed_frames * 1e3, "value": s_ref_still * timed_frames / still_elapsed / 1e6,
                                 "s_ref_per_frame": s_ref_still, "s_sampled_per_frame": s_sampled_still,
                                 **({"launch_ms": still_launch_ms,
                                     "frac": (s_sampled_still * cfg["b_step"] + n_px * B_RAY) * batch / (still_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if still_launch_ms else {})}}
               if still_elapsed is not None else {}),
            # the headline's own window in tolerance mode (VK_RENDER_FAST_WALK), beside the bit-exact headline
            **({"fast_walk": {"ms_per_step": fast_elapsed / K * 1e3, "ms_per_frame": fast_elapsed / timed_frames * 1e3, "value": s_ref * timed_frames / fast_elapsed / 1e6,
                              **({"launch_ms": fast_launch_ms, "frac": alg_frame * batch / (fast_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if fast_launch_ms else {}),
                              "note": "same orbit window, skips advance the position in closed form (one fma per coordinate; the iteration count is exact: an integer "
                                      "budget): 930 of C2's 636 049 hit pixels differ from the bit-exact frame by more than 1e-4 (max 1.6e-3) at unchanged iteration counts "
                                      "-- position drift where a coordinate crosses a power of two inside a walk -- and 14 early-outs flip (<= 2.2e-2 there): outside the "
                                      "1e-4 contract, a side figure (profiles/r05_fast_walk_contract.txt); S_ref priced as the exact mode's"}}
               if fast_elapsed is not None else {}),
        }
        if launch_ms is not None:
            alg = alg_frame * n_launch_frames
            achieved = alg / (launch_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": ("vk::raymarch_naive_kernel" if args.config == "c2" and args.layout in ("auto", "pairs", "packed") else STAGED_KERNELS)
                          + " (one launch spanning %d frames: the active tiles are marched, strips at the end of the same grid write the clear"
                            " colour of the others)" % n_launch_frames,
                "launch_ms": launch_ms, "frames_per_launch": n_launch_frames, "launches_timed": len(evs),
                "algorithmic_bytes_per_launch": alg,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                # the same launch priced at the reference's own step count (every iteration of the reference loop reads 8 taps; skipped
                # iterations are provably alpha == 0): a throughput equivalence in GB/s, deliberately NOT divided by a peak -- it is no roofline
                "GBps_if_every_reference_step_fetched": (s_ref * cfg["b_step"] + n_px * B_RAY) * n_launch_frames / (launch_ms * 1e-3) / 1e9,
            }
            if cfg["n"] > 256 or args.layout == "staged":
                out["roofline"]["frac_kind"] = "algorithmic: the staged march serves its taps from LDS, so this is an accounting of taps over the HBM peak, not a bound (physical side: extras / profiles)"
            # HBM bytes per launch from the PMC passes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes; tools/prof.sh, tools/pmc_traffic.py)
            # of THIS launch shape: the same orbit, the same number of frames per launch.  No figure for another shape is
            # scaled to this one (round 2 did that); a shape that was not profiled reports null.
            prof = os.path.join(ROOT, "profiles", PROF + "_pmc_traffic.json")
            if os.path.exists(prof) and not args.no_skip and args.layout == "auto":
                try:
                    pj = json.load(open(prof)).get(args.config, {})
                    ent = pj.get("per_frames_per_launch", {}).get(str(n_launch_frames))
                    if ent is not None:
                        out["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_measured_in_this_run"] = False  # PMC counters need rocprofv3 around the process: quoted from profiles/
                        out["roofline"]["traffic_source"] = ent.get("source", "profiles/%s_pmc_traffic.json" % PROF)
                    else:
                        out["roofline"]["traffic_note"] = "no PMC pass at %d frames per launch (profiled: %s)" % (n_launch_frames, sorted(pj.get("per_frames_per_launch", {})))
                    out["roofline"]["compulsory_GBps"] = (cfg["n"] ** 3 * (1 if cfg["fmt"] == "u8" else 2) + n_px * B_RAY * n_launch_frames) / (launch_ms * 1e-3) / 1e9
                except Exception:
                    pass
            # the VALU side of the same kernel (it is bound by instruction issue, not by bytes): occupancy of the vector pipe and of the
            # issue slots from the PMC passes of this launch shape, priced with the measured issue classes
            try:
                uj = json.load(open(os.path.join(ROOT, "profiles", PROF + "_utilisation.json")))
                ent = uj.get({"c2": "default", "c4": "c4", "c5": "c5"}[args.config])
                if ent and not args.no_skip and args.layout == "auto":
                    out["roofline"]["issue"] = {"measured_in_this_run": False, **{k: ent[k] for k in ("valu_pipe_occupancy", "issue_slot_occupancy", "cycles_per_valu_instruction", "hot_loop_mean_issue_cycles", "source")}}
            except Exception:
                pass
        else:
            # N > 1 (and --force-dist): march, gather and un-tile of different batches overlap on every rank, so no single
            # kernel duration describes a step; the figure here is the whole job's algorithmic bytes over the wall time
            # against N GPUs' HBM.  The per-kernel roofline is the N = 1 line's.
            agg = alg_frame * timed_frames / elapsed / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": agg, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": agg / (HBM_PEAK_GBS * world), "traffic": None,
                               "kernel": "whole job (vk::raymarch_naive_kernel per rank + RCCL gather + un-tile, overlapped)",
                               "note": "aggregate over %d GPU(s): algorithmic bytes of the window's frames / wall time; per-kernel figure: see the N = 1 line" % world}
            if rot_error is not None:
                out["rotating_root"] = {"error": rot_error}
            if rot_elapsed is not None:
                out["rotating_root"] = {"value": s_ref * timed_frames / rot_elapsed / 1e6, "unit": "Mray-steps/s", "ms_per_step": rot_elapsed / K * 1e3, "ms_per_frame": rot_elapsed / timed_frames * 1e3,
                                        "note": "the same window with launch g assembled on rank g mod N (BatchTileRenderer(root='rotate')): complete frames end up "
                                                "round-robin over the GPUs instead of on rank 0; `value` above is the gather to rank 0"}
            out["scaling_baseline"] = "like-for-like N = 1 baseline of this driver: extras.dist_driver_world1 of the N = 1 line (same batch, same partition + gather + un-tile path)"
            out["config"]["root_skip"] = root_skip_fixed
            # what a peer puts on its link to the root per frame: its active slots, colour only (alpha is 1 in every pixel)
            out["config"]["wire"] = wire_info
    return out


def c2_side_measurements(R, out):
    """Untimed side measurements of the C2 line (rank 0, N = 1): the orbit stream, the present pass, dense / skip kernels on stand-in and fog,
    the N > 1 driver as a world of one, the compute twin on the xor example's frame, C3."""
    BatchTileRenderer, H, V, W, args, batch, blob, cam_list, cfg = R.BatchTileRenderer, R.H, R.V, R.W, R.args, R.batch, R.blob, R.cam_list, R.cfg
    ctx, dist, layout, local_rank, pipe, rank, s_ref, timed_frames, torch = R.ctx, R.dist, R.layout, R.local_rank, R.pipe, R.rank, R.s_ref, R.timed_frames, R.torch
    world = R.world
    # untimed side measurements (rank 0, N = 1)
    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2":
        extras = {}
        try:
            # the frame stream of an orbiting camera (src/camera.rs rotates on input): every frame of every batch its own
            # camera, so the per-camera host work (tile order, cull rectangle, descriptors) is inside the wall time
            nb, B = 8, 32
            orbit = [[V.Camera(1.0, 0.5 + 0.1 * ((i * B + j) % 7) / 7.0, 1.0 + 6.28318 * (i * B + j) / (nb * B), (0.5, 0.5, 0.5), W / H).get_proj_view_matrix()
                      for j in range(B)] for i in range(nb)]
            ofr = torch.empty((B, H, W, 4), dtype=torch.float16, device="cuda")
            V.render_batch(ctx, pipe, orbit[0], ofr.data_ptr(), tile_size=TILE)
            ctx.sync()
            h0 = time.perf_counter()
            for cams in orbit[:3]:
                V.render_batch(ctx, pipe, cams, ofr.data_ptr(), tile_size=TILE)
            host_us = (time.perf_counter() - h0) / (3 * B) * 1e6
            ctx.sync()
            t0 = time.perf_counter()
            for cams in orbit:
                V.render_batch(ctx, pipe, cams, ofr.data_ptr(), tile_size=TILE)
            ctx.sync()
            msf = (time.perf_counter() - t0) / (nb * B) * 1e3
            del ofr
            extras["orbit_one_camera_per_frame"] = {"ms_per_frame": msf, "frames": nb * B, "frames_per_launch": B, "host_us_per_camera": host_us,
                                                    "note": "wall time of 8 batches of 32 distinct cameras (one orbit); the views differ, so the march work per frame does too"}
        except Exception as e:
            extras["orbit_one_camera_per_frame"] = {"error": str(e)}
        try:
            # the step after the hot path: present pass, backbuffer rgba16f -> window-sized RGBA8
            ms_p = time_launches(ctx, lambda: ctx.render(), 50, warm=10)
            extras["present_1080p"] = {"launch_ms": ms_p, "algorithmic_GBps": W * H * 12 / (ms_p * 1e-3) / 1e9, "frac": W * H * 12 / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "note": "bilinear resample + ACES + sRGB + RGBA8, 8 B read + 4 B written per pixel; bound by its 3 divisions, 3 logs and 3 exps per pixel, not by HBM"}
        except Exception as e:
            extras["present_1080p"] = {"error": str(e)}
        it = 50
        p_ns = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=V.RENDER_NO_SKIP)
        p_sk = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=DT_SCALE, flags=V.RENDER_FORCE_SKIP)
        for name, mk in (("standin", None), ("fog", lambda: V.VolumeTexture.generate_fog(ctx, (cfg["n"],) * 3, layout=layout))):
            if mk is not None:
                mk()
            for mode, p, fl in (("noskip", p_ns, V.RENDER_NO_SKIP), ("skip", p_sk, V.RENDER_FORCE_SKIP)):
                sr, ss = count_steps(ctx, V, fl)
                ms = time_launches(ctx, lambda: p.record(ctx), 50, warm=20)
                gb = (ss * cfg["b_step"] + W * H * B_RAY) / (ms * 1e-3) / 1e9
                extras[f"{name}_{mode}"] = {"launch_ms": ms, "s_ref": sr, "s_sampled": ss, "Mray_steps_per_s": sr / ms / 1e3,
                                           "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS}
            # the sampling loop with every step fetching its taps, batched like the headline
            try:
                fr = torch.empty((8, H, W, 4), dtype=torch.float16, device="cuda")
                ms = time_launches(ctx, lambda: V.render_batch(ctx, p_ns, [blob] * 8, fr.data_ptr(), tile_size=TILE), 12, warm=4)
                sr, ss = extras[f"{name}_noskip"]["s_ref"], extras[f"{name}_noskip"]["s_sampled"]
                gb = (ss * cfg["b_step"] + W * H * B_RAY) * 8 / (ms * 1e-3) / 1e9
                extras[f"{name}_noskip_batch8"] = {"launch_ms": ms, "frames_per_launch": 8, "Mray_steps_per_s": sr * 8 / ms / 1e3, "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS}
                del fr
            except Exception as e:
                extras[f"{name}_noskip_batch8"] = {"error": str(e)}
        V.VolumeTexture.generate_standin(ctx, (cfg["n"],) * 3, layout=layout)
        ctx.sync()
        # the N > 1 driver as a world of one (partition + self-gather + un-tile): the like-for-like baseline of the N > 1 lines
        try:
            created = False
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29519")
                dist.init_process_group("gloo", rank=0, world_size=1)
                created = True
            b1 = BatchTileRenderer(ctx, pipe, tile_size=TILE, batch=batch, root=0, transport="rccl")
            for j in range(2 * batch):
                b1.submit(cam_list[j % batch])
            b1.flush()
            torch.cuda.synchronize()
            msfs = []
            for _ in range(3):  # the headline's rule: the same orbit, one contiguous window of `timed_frames`, median of three
                t0 = time.perf_counter()
                for j in range(timed_frames):
                    b1.submit(cam_list[j % batch])
                b1.flush()
                torch.cuda.synchronize()
                msfs.append((time.perf_counter() - t0) / timed_frames * 1e3)
            msf = sorted(msfs)[1]
            b1.close()
            extras["dist_driver_world1"] = {"ms_per_frame": msf, "Mray_steps_per_s": s_ref / msf / 1e3, "frames_per_launch": batch, "timed_frames": timed_frames,
                                            "note": "BatchTileRenderer at world 1 on the headline's orbit and window: compact tiles, gather to self, un-tile"}
            if created:
                dist.destroy_process_group()
        except Exception as e:
            extras["dist_driver_world1"] = {"error": str(e)}
        # the compute twin (raycast_compute.wgsl `single`) on the xor example's own configuration:
        # 256^3 rgba16f pair generated on the device, 1280x720, xor camera, dt = 0.01; 16 B per step
        try:
            cx = V.Context(1280, 720, V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), 1280 / 720), device=local_rank,
                           backbuffer=(1280, 720), out_format=V.OUT_RGBA16F)
            try:
                V.VolumeTexture.generate_xor(cx, (256,) * 3, 0.0)
                cx.update()
                cx.reset_step_counts()
                V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=V.RENDER_COUNT).record(cx)
                sr, ss = cx.step_counts()
                pc = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST)
                ms = time_launches(cx, lambda: pc.record(cx), it)
                gb = (ss * 16 + 1280 * 720 * B_RAY) / (ms * 1e-3) / 1e9
                extras["xor_compute_nearest_720p"] = {"launch_ms": ms, "s_ref": sr, "s_sampled": ss, "Mray_steps_per_s": sr / ms / 1e3,
                                                      "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS,
                                                      "note": "exact empty-space skipping (s_sampled < s_ref) and a four-deep request ring since round 4; the 720p frame is one partial round "
                                                              "of waves and lasts as long as its longest rays (profiles/r04_compute_twin_skip_and_ring.txt)",
                                                      "valu_per_step": 76, "valu_mean_issue_cycles": 3.07, "valu_frac_floor": valu_floor_frac(ss, 76, 3.07, ms)}
                pn = V.RaycastPipeline(V.MODE_COMPUTE_NEAREST, flags=V.RENDER_NO_SKIP)
                ms_n = time_launches(cx, lambda: pn.record(cx), it)
                extras["xor_compute_nearest_720p"]["no_skip"] = {"launch_ms": ms_n, "frac": (sr * 16 + 1280 * 720 * B_RAY) / (ms_n * 1e-3) / 1e9 / HBM_PEAK_GBS}
                # the same frame eight per launch: what the kernel does once the machine is full (a single 720p frame
                # is 14 400 waves, less than two rounds of the 8192 wave slots)
                xfr = torch.empty((8, 720, 1280, 4), dtype=torch.float16, device="cuda")
                xblob = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), 1280 / 720).get_proj_view_matrix()
                msb = time_launches(cx, lambda: V.render_batch(cx, pc, [xblob] * 8, xfr.data_ptr(), tile_size=TILE), 12, warm=4)
                gbb = (ss * 16 + 1280 * 720 * B_RAY) * 8 / (msb * 1e-3) / 1e9
                extras["xor_compute_nearest_720p_batch8"] = {"launch_ms": msb, "frames_per_launch": 8, "Mray_steps_per_s": sr * 8 / msb / 1e3,
                                                             "achieved_GBps": gbb, "frac": gbb / HBM_PEAK_GBS}
                del xfr
                # the xor example's own loop: one `single` dispatch per frame, the camera turning; one stream, and frames in flight
                xcams = [V.Camera(3.0, -0.5, 1.0 + 6.28318 * j / 1024, (0.0, 0.0, 0.0), 1280 / 720).get_proj_view_matrix() for j in range(128)]
                fl = {}
                for k in (0, 2, 3, 4):
                    if k:
                        cx.frames_in_flight(k)
                    fl["one_stream" if k == 0 else "in_flight_%d" % k] = frame_stream_ms(cx, pc, xcams, 256, k)
                cx.frames_in_flight(1)
                extras["xor_compute_nearest_720p"]["frame_stream_ms_per_frame"] = fl
            finally:
                cx.close()
        except Exception as e:  # a side measurement must not take the headline down
            extras["xor_compute_nearest_720p"] = {"error": str(e)}
        # C3: the procedural (no volume) configuration at 1920x1080, xor camera -- ALU work only:
        # 24 specified sines (f64 Cody-Waite, ~45 f64 ops each) + ~200 f32 flops per step, 0 volume bytes
        try:
            cam3 = V.Camera(3.0, -0.5, 1.0, (0.0, 0.0, 0.0), W / H)
            cp = V.Context(W, H, cam3, device=local_rank, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
            try:
                cp.set_camera_blob(cam3.get_proj_view_matrix())  # Uniform.time stays 0, as the reference runs xor.wgsl
                cp.reset_step_counts()
                V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_COUNT).record(cp)
                sr, _ = cp.step_counts()
                pp = V.RaycastPipeline(V.MODE_PROCEDURAL)
                ms = time_launches(cp, lambda: pp.record(cp), 5, warm=1)
                # per step (static count of the loop, tools/isa_hist.py): 552 f64 instructions (24 sines x 23: Cody-Waite reduction, two
                # Horner series, quadrant pick), 400 f32 / integer ones.  Fraction of the f64 vector peak = f64 lane-operations per
                # second / (1024 SIMDs x 16 lanes x 2.4 GHz); bound "valu" ("flops/step vs VALU peak, not HBM").
                f64_rate = 552.0 * sr / (ms * 1e-3)
                extras["c3_procedural_1080p"] = {"launch_ms": ms, "s_ref": sr, "Mray_steps_per_s": sr / ms / 1e3,
                                                 "Gsines_per_s": 24 * sr / ms / 1e6, "f64_instructions_per_step": 552, "other_valu_per_step": 400,
                                                 "bound": "valu (f64)", "f64_T_lane_ops_per_s": f64_rate / 1e12, "f64_frac_of_vector_peak": f64_rate / F64_LANE_OPS_PEAK,
                                                 "valu_frac_floor": valu_floor_frac(sr, 952, 3.77, ms),
                                                 "valu_frac_pmc": "0.80 (profiles/r03_utilisation.txt; 112 VGPRs: 4 waves per SIMD)", "volume_bytes_per_step": 0}
                # beside it, what the chip does with the shader AS WRITTEN: hash()'s sine through v_sin_f32 (VK_RENDER_DEVICE_SINE, a tolerance
                # mode: another noise field of the same statistics -- tests/test_frames_gpu.py holds the bars)
                pdv = V.RaycastPipeline(V.MODE_PROCEDURAL, flags=V.RENDER_DEVICE_SINE)
                ms_d = time_launches(cp, lambda: pdv.record(cp), 10, warm=2)
                extras["c3_procedural_1080p"]["device_sine"] = {
                    "launch_ms": ms_d, "Mray_steps_per_s_at_specified_step_count": sr / ms_d / 1e3, "speedup": ms / ms_d,
                    "note": "tolerance mode, NOT the specified arithmetic: hash = fract(sin(h) * 43758.5) with the hardware sine (v_sin_f32 after * 1/2pi and fract) "
                            "as a GPU running shaders/xor.wgsl:18-20 as written computes it; against the specified frame: mean |d| 0.017, max 0.17, mean colour "
                            "within 0.3 %, 8x8-blurred correlation 0.986 (tools/c3_device_sine.py)"}
            finally:
                cp.close()
        except Exception as e:
            extras["c3_procedural_1080p"] = {"error": str(e)}
        out["extras"] = extras


def finish_on_stream(R, out):
    """Co-headlines inside `roofline`, the CPU baseline, tear-down of renderer and context, the other single-GPU configs."""
    V, args, batch, blob, btr, ctx, local_rank, rank, rehearsal = R.V, R.args, R.batch, R.blob, R.btr, R.ctx, R.local_rank, R.rank, R.rehearsal
    s_ref, s_ref_still, torch, use_dist, world = R.s_ref, R.s_ref_still, R.torch, R.use_dist, R.world
    # co-headlines inside `roofline` (the driver's record keeps that object whole): the reference's submission model, one frame per
    # launch, and the sampling loop's own fraction -- the same kernel family with every step fetching its taps -- measured in this run
    if rank == 0 and "roofline" in out:
        sf = out.get("latency", {})
        if "launch_ms" in sf:
            out["roofline"]["single_frame"] = {"launch_ms": sf["launch_ms"], "Mray_steps_per_s": sf["value"], "frac": sf["frac"], "measured_in_this_run": True,
                                               **({"tolerance_walk_launch_ms": sf["fast_walk"]["launch_ms"]} if "fast_walk" in sf else {}),
                                               **({"frames_in_flight_best": sf["in_flight"]["best"]} if "best" in sf.get("in_flight", {}) else {})}
        ex = out.get("extras", {})
        if "frac" in ex.get("standin_noskip", {}):
            out["roofline"]["dense_kernel"] = {"frac_single_frame": ex["standin_noskip"]["frac"], "frac_8_frames_per_launch": ex.get("standin_noskip_batch8", {}).get("frac"),
                                               "measured_in_this_run": True,
                                               "note": "the same workload with VK_RENDER_NO_SKIP: every reference iteration fetches its 8 taps (S_sampled = S_ref); "
                                                       "exact skipping removes sampled bytes ~21x and time ~2.6x, so `frac` above falls by construction"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "c2":
        cb, s_cpu = cpu_baseline(blob)
        out["cpu_baseline"] = cb
        out["cpu_baseline"]["s_ref_matches_gpu"] = (s_cpu == s_ref_still)
    if use_dist and btr is not None:
        btr.close()
    if rank == 0 and world > 1 and not rehearsal and args.config == "c2" and not args.no_rotate:
        # beside the gather to rank 0 and the rotating root: the single-process group, gathered and peer-direct
        try:
            out["single_process_group"] = single_process_group(world, batch, s_ref)
        except Exception as e:  # noqa: BLE001
            out["single_process_group"] = {"error": repr(e)}
    ctx.close()
    # the other single-GPU BASELINE configs (their own contexts: the C2 volume is gone by now)
    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2":
        for key, name in (("c4", "c4_1024_f16_1080p"), ("c5", "c5_2048_u8_4k")):
            try:
                out["extras"][name] = big_config_extra(V, torch, local_rank, key)
            except Exception as e:
                out["extras"][name] = {"error": str(e)}


def finish(R, out):
    """Leave the process group; N > 1: C5 through the same driver as a child job; print the line."""
    args, dist, rank, use_dist, world = R.args, R.dist, R.rank, R.use_dist, R.world
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # N > 1: BASELINE's 8-GPU configuration through the same driver -- a child job, once this one's ranks are out of their collectives
    if rank == 0 and world > 1 and args.config == "c2" and not args.no_extras and os.environ.get("VK_BENCH_C5_AT_N", "1") != "0":
        out.setdefault("extras", {})["c5_at_n"] = c5_at_n(world)
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it first so the JSON line is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    R = plan(args)
    open_process(R)
    with R.torch.cuda.stream(R.stream):
        open_workload(R)
        make_submitter(R)
        make_timed_region(R)
        run_windows(R)
        out = headline_line(R)
        # One vk_render per frame -- the reference's own submission model -- on the same workload: `latency`
        if R.rank == 0 and R.world == 1 and not args.headline_only:
            try:
                out["latency"] = latency_section(R.V, R.torch, R.ctx, R.stream, R.cfg, R.flags, R.s_ref_still, R.s_sampled_still, R.cam_list, R.s_ref, R.s_sampled)
            except Exception as e:  # a side measurement must not take the headline down
                out["latency"] = {"error": repr(e)}
        c2_side_measurements(R, out)
        finish_on_stream(R, out)
    finish(R, out)


if __name__ == "__main__":
    main()
