#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 tools/ubench/_build/valu_rate 12 cndmask,bfi,xor,floor,rndne > gpurun_out/r03_ubench_valu4.txt 2>&1 || { echo "ubench failed"; tail -5 gpurun_out/r03_ubench_valu4.txt; exit 1; }
cat gpurun_out/r03_ubench_valu4.txt
