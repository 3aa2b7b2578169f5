#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 tools/ubench/_build/valu_rate > gpurun_out/r03_ubench_valu2.txt 2>&1 || { echo "ubench failed"; tail -5 gpurun_out/r03_ubench_valu2.txt; exit 1; }
echo "ubench done"
timeout -k 10 700 bash tools/prof_r3.sh > gpurun_out/r03_prof_r3.log 2>&1 || { echo "prof_r3 failed"; tail -20 gpurun_out/r03_prof_r3.log; exit 1; }
echo "prof_r3 done"
timeout -k 10 900 bash tools/prof_r3_kernels.sh > gpurun_out/r03_prof_r3k.log 2>&1 || { echo "prof_r3k failed"; tail -20 gpurun_out/r03_prof_r3k.log; exit 1; }
echo "prof_r3k done"
