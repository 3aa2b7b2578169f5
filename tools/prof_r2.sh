#!/bin/bash
# usage: tools/prof_r2.sh <tag> <prof_frames.py args...>  -- kernel trace + PMC passes (each in its own run) into gpurun_out/prof_<tag>/
set -u
tag=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/prof_frames.py "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $R/tools/prof_frames.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $out/p2 -- python3 $R/tools/prof_frames.py "$@" > $out/p2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p3 -- python3 $R/tools/prof_frames.py "$@" > $out/p3.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $out/p4 -- python3 $R/tools/prof_frames.py "$@" > $out/p4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p5 -- python3 $R/tools/prof_frames.py "$@" > $out/p5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p6 -- python3 $R/tools/prof_frames.py "$@" > $out/p6.log 2>&1
python3 $R/tools/pmc_summary.py $out > $out/summary.txt 2>&1
grep -v "pack_\|generate_\|clear_\|dist_pass\|build_" $out/summary.txt | head -80
