#!/bin/bash
# Round-3 evidence for the headline: for each launch shape bench.py runs --
#   default:  `python bench.py`                      -> 64 frames per launch
#   driver:   `python bench.py --steps 20 --warmup 5` -> 52 frames per launch
# -- rocprofv3 kernel stats of the headline's launches alone (--headline-only: no still-camera window, no single-frame
# launches of the same kernel in the averages), then the PMC passes, each in its own run: FETCH_SIZE, WRITE_SIZE (HBM
# traffic of THIS shape), the SQ issue counters.  Output: gpurun_out/prof_r3/<shape>/..., summary.txt, and
# gpurun_out/prof_r3/r03_pmc_traffic_c2.json (tools/pmc_traffic.py).
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r3; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in default driver; do
  if [ $shape = default ]; then A=""; else A="--steps 20 --warmup 5"; fi
  B="python3 $R/bench.py --no-extras --no-cpu-baseline --headline-only $A"
  o=$out/$shape; mkdir -p $o
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- $B > $o/trace.log 2>&1 || { echo "trace failed ($shape)"; tail -5 $o/trace.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- $B > $o/fetch.log 2>&1 || { echo "fetch failed"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- $B > $o/write.log 2>&1 || { echo "write failed"; exit 1; }
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $o/sq1 -- $B > $o/sq1.log 2>&1 || { echo "sq1 failed"; exit 1; }
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $o/sq2 -- $B > $o/sq2.log 2>&1 || { echo "sq2 failed"; exit 1; }
  grep "^{" $o/trace.log | tail -1 > $o/bench_line_under_rocprof.json
  cp $o/trace/*/*kernel_stats.csv $o/kernel_stats.csv 2>/dev/null
  python3 $R/tools/pmc_summary.py $o > $o/summary.txt 2>&1
  echo "== $shape"; grep -v "pack_\|generate_\|clear_\|dist_pass\|build_" $o/summary.txt | head -40
done
python3 $R/tools/pmc_traffic.py $out c2 > $out/r03_pmc_traffic_c2.json && cat $out/r03_pmc_traffic_c2.json
