#!/bin/bash
# round-3 evidence on HEAD: the gpu suite, the profiles, the two bench lines
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_pytest_gpu.log 2>&1
rc=$?; tail -4 gpurun_out/r03_final_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 700 bash tools/prof_r3.sh > gpurun_out/r03_final_prof_r3.log 2>&1 || { echo "prof_r3 failed"; tail -20 gpurun_out/r03_final_prof_r3.log; exit 1; }
echo "prof_r3 done"
if [ "${1:-}" = "all" ]; then
  timeout -k 10 900 bash tools/prof_r3_kernels.sh > gpurun_out/r03_final_prof_r3k.log 2>&1 || { echo "prof_r3k failed"; tail -20 gpurun_out/r03_final_prof_r3k.log; exit 1; }
  echo "prof_r3k done"
fi
timeout -k 10 400 python bench.py > gpurun_out/r03_final_bench_default.json 2> gpurun_out/r03_final_bench_default.err || { tail -20 gpurun_out/r03_final_bench_default.err; exit 1; }
echo "bench default done"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench_driver.json 2> gpurun_out/r03_final_bench_driver.err || { tail -20 gpurun_out/r03_final_bench_driver.err; exit 1; }
echo "bench driver done"
