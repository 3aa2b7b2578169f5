"""What would re-packing rays between the four waves of a 256-thread group buy the skip kernel?  

The COUNT build logs every trip of every wave of one C2 frame (live lanes, lanes that sample, samplers whose alpha is not 0, wave-level walk
iterations: `trip_log_cap`, debug bit 6).  From the logs:
  * the census as it is: executions of each part of a trip and the lanes they serve;
  * the census after re-packing, under two models of a group = the 2 x 2 neighbouring 8x8 blocks (16 x 16 px):
      lockstep  -- the four waves run trip k together, the group's samplers of that trip are packed into ceil(n / 64) sample executions;
      ideal     -- no alignment constraint at all: ceil(all samplers of the group / 64) executions (a lower bound no scheme can beat);
  * both priced with the issue-cycle costs of the parts of the loop (tools/isa_hist.py on the production kernel, profiles/r03_ubench_valu_issue_rate.txt),
    with and without the exchange's own cost per wave-trip.
"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vokselis_amd as V
from vokselis_amd import _native as N

W, H, CAP = 1920, 1080, 1024
# issue cycles per wave execution of each part (production kernel raymarch_naive_kernel<3,true,false,1,false>, blocks of the bounded march loop)
P_PROBE, P_BOUND, P_WALK_IT, P_REM, P_SAMPLE, P_PALETTE = 68.0, 70.0, 44.0, 25.0, 110.0, 50.0
# the exchange: ballot + mbcnt + one LDS atomic for the queue slot, ds_write_b128 of the request, two s_barrier, ds_read_b128 of the result per
# wave-trip (~10 instructions, two of them LDS); per packed execution a ds_read_b128 + ds_write_b128 on top of the sample itself
X_WAVE_TRIP, X_EXEC = 36.0, 16.0

cam = V.Camera(1.0, 0.5, 1.0 + (float(sys.argv[1]) if len(sys.argv) > 1 else 0.0), (0.5, 0.5, 0.5), W / H)
ctx = V.Context(W, H, cam, backbuffer=(W, H), out_format=V.OUT_RGBA16F)
V.VolumeTexture.generate_standin(ctx, (256,) * 3)
ctx.update()
pipe = V.RaycastPipeline(V.MODE_NAIVE_TRILINEAR, dt_scale=0.5, flags=V.RENDER_COUNT | V.RENDER_PROBE_ALWAYS)
ctx.set_param("trip_log_cap", CAP)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 1, None, 0))
ctx.reset_step_counts(); pipe.record(ctx); ctx.sync()
s_ref, s_samp = ctx.step_counts(); cen = ctx.simt_census()
nb = 30 * 17 * 64
buf = np.zeros(nb * CAP // 2, np.uint64)
N.check(ctx.handle, N.lib().vk_debug_wave_trace(ctx.handle, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), nb * (CAP // 8)))
ctx.close()
log = buf.view(np.uint32).reshape(nb // 64, 8, 8, CAP)          # [slot][sub_y][sub_x][trip]
live, samp, nz, wit = log & 127, (log >> 7) & 127, (log >> 14) & 127, (log >> 21) & 127
assert int(live.max()) <= 64 and int(samp.max()) <= 64
trips = int((live > 0).sum())
assert int((live[..., -1] > 0).sum()) == 0, "raise CAP"
walkers = live - samp
walk_exec, samp_exec, pal_exec = (wit > 0), (samp > 0), (nz > 0)

def price(n_trip, n_bound, n_walk_it, n_samp_exec, n_pal_exec, extra=0.0):
    return n_trip * P_PROBE + n_bound * (P_BOUND + P_REM) + n_walk_it * P_WALK_IT + n_samp_exec * P_SAMPLE + n_pal_exec * P_PALETTE + extra

n_bound = int(walk_exec.sum()); n_wit = int(np.maximum(wit.astype(np.int64) - 1, 0).sum())
before = price(trips, n_bound, n_wit, int(samp_exec.sum()), int(pal_exec.sum()))
out = {"S_ref": s_ref, "S_sampled": s_samp, "census_counters": cen,
       "as_is": {"wave_trips": trips, "live_lanes_per_trip": round(float(live.sum()) / trips, 1),
                 "walk_executions": n_bound, "walkers_per_walk_execution": round(float(walkers[walk_exec].sum()) / max(n_bound, 1), 1), "walk_iterations_beyond_first": n_wit,
                 "sample_executions": int(samp_exec.sum()), "lanes_per_sample_execution": round(float(samp.sum()) / max(int(samp_exec.sum()), 1), 1),
                 "palette_executions": int(pal_exec.sum()), "lanes_alpha_nonzero_per_palette_execution": round(float(nz.sum()) / max(int(pal_exec.sum()), 1), 1),
                 "sample_executions_with_at_most_4_lanes": int(((samp > 0) & (samp <= 4)).sum()),
                 "priced_cycles": before,
                 "share_probe": round(trips * P_PROBE / before, 3), "share_walk": round((n_bound * (P_BOUND + P_REM) + n_wit * P_WALK_IT) / before, 3),
                 "share_sample": round(int(samp_exec.sum()) * P_SAMPLE / before, 3), "share_palette": round(int(pal_exec.sum()) * P_PALETTE / before, 3)}}
# groups of 2 x 2 blocks
g = lambda a: a.reshape(nb // 64, 4, 2, 4, 2, CAP).transpose(0, 1, 3, 2, 4, 5).reshape(-1, 4, CAP).astype(np.int64)  # [group][wave][trip]
gs, gn, gl = g(samp), g(nz), g(live)
lock_samp_exec = int(np.ceil(gs.sum(1) / 64.0).sum())
lock_pal_exec = int(np.ceil(gn.sum(1) / 64.0).sum())            # packed by "alpha is not 0" as well: the best case
ideal_samp_exec = int(np.ceil(gs.sum((1, 2)) / 64.0).sum())
ideal_pal_exec = int(np.ceil(gn.sum((1, 2)) / 64.0).sum())
group_trips = int((gl.sum(1) > 0).sum())
for name, se, pe in (("lockstep", lock_samp_exec, lock_pal_exec), ("ideal", ideal_samp_exec, ideal_pal_exec)):
    free = price(trips, n_bound, n_wit, se, pe)
    paid = price(trips, n_bound, n_wit, se, pe, extra=trips * X_WAVE_TRIP + se * X_EXEC)
    out[name] = {"sample_executions": se, "palette_executions": pe, "lanes_per_sample_execution": round(float(samp.sum()) / max(se, 1), 1),
                 "priced_cycles_exchange_free": free, "vs_as_is_exchange_free": round(free / before, 3),
                 "priced_cycles_with_exchange": paid, "vs_as_is_with_exchange": round(paid / before, 3)}
# what lockstep costs in waiting: the group's trip lasts as long as its slowest wave's walk
gw = g(np.maximum(wit.astype(np.int64), 0))
out["lockstep"]["group_trips"] = group_trips
out["lockstep"]["waves_alive_per_group_trip"] = round(float((gl > 0).sum()) / group_trips, 2)
out["lockstep"]["walk_iterations_waited_for_per_walked"] = round(float((gw.max(1)[:, None, :] * (gl > 0)).sum()) / max(float(gw.sum()), 1.0), 2)
print(json.dumps(out, indent=1))
