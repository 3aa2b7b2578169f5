#!/bin/bash
# quick instruction census of one config: tools/prof_valu.sh <tag> <prof_frames.py args...>
set -u
tag=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $R/tools/prof_frames.py "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/prof_frames.py "$@" > $out/trace.log 2>&1
python3 $R/tools/pmc_summary.py $out | grep -A9 "raymarch" | grep -v "pack_\|generate" | head -24
