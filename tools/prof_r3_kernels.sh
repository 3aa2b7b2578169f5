#!/bin/bash
# Round-3 evidence per kernel on HEAD: kernel stats + the SQ issue counters (three passes) + traffic passes for
#   c5 (staged u8, 3840x2160), c4 (staged f16), fogbatch (c2fog dense at 8 frames per launch), xor (the compute twin, 720p), c3 (procedural 1080p).
# usage (GPU box): tools/prof_r3_kernels.sh [cases...]   -> gpurun_out/prof_r3k/<case>/summary.txt
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r3k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cases=${@:-"c5 c4 fogbatch xor c3"}
for c in $cases; do
  unset VK_NOSKIP VK_BATCH
  case $c in
    c5) A="c5 s8 5";;
    c4) A="c4 s8 8";;
    fogbatch) A="c2fog p16 10"; export VK_NOSKIP=1 VK_BATCH=8;;
    xor) A="xor auto 30";;
    c3) A="c3 auto 6";;
    *) echo "unknown case $c"; exit 1;;
  esac
  o=$out/$c; rm -rf $o; mkdir -p $o
  P="python3 $R/tools/prof_frames.py $A"
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- $P > $o/trace.log 2>&1 || { echo "trace failed ($c)"; tail -5 $o/trace.log; exit 1; }
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $o/sq1 -- $P > $o/sq1.log 2>&1 || { echo "sq1 failed"; exit 1; }
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $o/sq2 -- $P > $o/sq2.log 2>&1 || { echo "sq2 failed"; exit 1; }
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $o/sq3 -- $P > $o/sq3.log 2>&1 || echo "sq3 failed (kept going)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- $P > $o/fetch.log 2>&1 || { echo "fetch failed"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- $P > $o/write.log 2>&1 || { echo "write failed"; exit 1; }
  cp $o/trace/*/*kernel_stats.csv $o/kernel_stats.csv 2>/dev/null
  python3 $R/tools/pmc_summary.py $o > $o/summary.txt 2>&1
  echo "== $c"; grep -v "pack_\|generate_\|clear_\|dist_pass\|build_\|xor_generate" $o/summary.txt | head -60
done
