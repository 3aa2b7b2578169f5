#!/bin/bash
# usage: tools/prof.sh <tag> <case args...>   -- kernel trace + PMC passes into gpurun_out/prof_<tag>/
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/prof_case.py "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/pmc1 -- python3 $R/tools/prof_case.py "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 $R/tools/prof_case.py "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc3 -- python3 $R/tools/prof_case.py "$@" > $out/pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $out/pmc4 -- python3 $R/tools/prof_case.py "$@" > $out/pmc4.log 2>&1
find $out -name "*.csv" | head -30
