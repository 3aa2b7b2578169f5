#!/bin/bash
# rocprofv3 runs of the bench command (N=1): kernel trace + stats, then PMC passes (each its own run).
# usage (on the GPU box): tools/prof_bench.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
tag=${1:-bench}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $ARGS > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 $ARGS > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $ARGS > $out/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS --output-format csv -d $out/pmc_misc -- python3 $ARGS > $out/pmc_misc.log 2>&1
python3 $R/tools/pmc_summary.py $out > $out/summary.txt 2>&1
tail -5 $out/trace.log
