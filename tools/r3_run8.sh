#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "skip or golden or bonsai or batch_equals or paced or fuzz" > gpurun_out/r03_pytest_gpu4.log 2>&1
rc=$?; tail -4 gpurun_out/r03_pytest_gpu4.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err || { tail -20 gpurun_out/r03_bench3.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench3.json') if l.startswith('{')][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "still", d["still_camera"]["ms_per_step"], "single", d["single_frame"]["launch_ms"])
for k in ("standin_skip","fog_skip","standin_noskip","orbit_one_camera_per_frame","dist_driver_world1"):
    print(k, {kk:vv for kk,vv in d["extras"][k].items() if kk in ("launch_ms","frac","ms_per_frame")})
PY
