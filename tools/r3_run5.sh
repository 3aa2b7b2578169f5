#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu2.log 2>&1
rc=$?; tail -5 gpurun_out/r03_pytest_gpu2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default2.json 2> gpurun_out/r03_bench_default2.err || { tail -20 gpurun_out/r03_bench_default2.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_default2.json') if l.startswith('{')][-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
for k in ("c3_procedural_1080p","c4_1024_f16_1080p","c5_2048_u8_4k","xor_compute_nearest_720p"):
    print(k, {kk:vv for kk,vv in d["extras"][k].items() if kk in ("launch_ms","frac","Mray_steps_per_s")})
PY
