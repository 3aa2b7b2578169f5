#!/bin/bash
# Round-2 evidence: rocprofv3 kernel stats of the bench command, PMC traffic of its kernel, and kernel stats + PMC of C4 / C5 / C2-fog.
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r02; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-extras --no-cpu-baseline --steps 128 --warmup 32"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- $B > $out/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_fetch -- $B > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/bench_write -- $B > $out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/bench_sq -- $B > $out/bench_sq.log 2>&1
python3 $R/tools/pmc_summary.py $out > $out/bench_summary.txt 2>&1
for cfg in "c4 s8" "c5 s8" "c4 b9"; do
  set -- $cfg
  bash $R/tools/prof_r2.sh r02_$1_$2 $1 $2 6 > /dev/null 2>&1
done
VK_NOSKIP=1 bash $R/tools/prof_r2.sh r02_c2fog_noskip c2fog p16 20 > /dev/null 2>&1
grep -h "raymarch\|FETCH\|WRITE\|mean=" $out/bench_summary.txt | head -40
for d in r02_c4_s8 r02_c5_s8 r02_c4_b9 r02_c2fog_noskip; do echo "== $d"; grep -v "pack_\|generate_\|clear_\|dist_pass\|build_\|rocclr" $R/gpurun_out/prof_$d/summary.txt | head -45; done
