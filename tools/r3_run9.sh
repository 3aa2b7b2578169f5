#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python tools/staged_grow.py c5 > gpurun_out/r03_staged_grow_c5.txt 2>&1 || { tail -20 gpurun_out/r03_staged_grow_c5.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r03_staged_grow_c5.txt
timeout -k 10 300 python tools/staged_grow.py c4 > gpurun_out/r03_staged_grow_c4.txt 2>&1 || { tail -20 gpurun_out/r03_staged_grow_c4.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r03_staged_grow_c4.txt
