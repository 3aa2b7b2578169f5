#!/bin/bash
# occupancy / VALU / cache counters of one big config: tools/prof_big2.sh <c4|c5small|c5> <layout>
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_big2_$1_$2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $R/tools/big_configs.py $1 $2 > $out/log1.txt 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p2 -- python3 $R/tools/big_configs.py $1 $2 > $out/log2.txt 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $out/p3 -- python3 $R/tools/big_configs.py $1 $2 > $out/log3.txt 2>&1
python3 $R/tools/pmc_summary.py $out | grep -A9 "false, true, 1, false" | grep -v "^--" | head -40
tail -2 $out/log3.txt
