#!/bin/bash
# FETCH_SIZE / kernel time of one big config: tools/prof_big.sh <c4|c5small|c5> <b9|q|p8|p16|auto>
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_big_$1_$2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc -- python3 $R/tools/big_configs.py $1 $2 > $out/log.txt 2>&1
python3 $R/tools/pmc_summary.py $out | grep -A2 "raymarch" | head -12
