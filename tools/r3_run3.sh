#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
UB_VERBOSE=1 timeout -k 10 500 tools/ubench/_build/valu_rate 12 > gpurun_out/r03_ubench_valu3.txt 2>&1 || { echo "ubench failed"; tail -5 gpurun_out/r03_ubench_valu3.txt; exit 1; }
echo "ubench done"
timeout -k 10 300 python tools/staged_row_pad.py c4 > gpurun_out/r03_row_pad_c4.txt 2>&1 || { echo "row pad c4 failed"; tail -20 gpurun_out/r03_row_pad_c4.txt; exit 1; }
cat gpurun_out/r03_row_pad_c4.txt
timeout -k 10 400 python tools/staged_row_pad.py c5 0,5120,8192 > gpurun_out/r03_row_pad_c5.txt 2>&1 || { echo "row pad c5 failed"; tail -20 gpurun_out/r03_row_pad_c5.txt; exit 1; }
cat gpurun_out/r03_row_pad_c5.txt
