#!/bin/bash
# round 3, first GPU call: clock-independent VALU issue table, the gpu suite, the bench lines
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 tools/ubench/_build/valu_rate > gpurun_out/r03_ubench_valu.txt 2>&1 || { echo "ubench failed"; tail -5 gpurun_out/r03_ubench_valu.txt; exit 1; }
echo "ubench done"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/r03_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -20 gpurun_out/r03_bench_default.err; exit 1; }
echo "bench default done"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver.json 2> gpurun_out/r03_bench_driver.err || { tail -20 gpurun_out/r03_bench_driver.err; exit 1; }
echo "bench driver done"
