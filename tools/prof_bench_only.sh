#!/bin/bash
# rocprofv3 kernel stats of the bench command + PMC traffic of its kernel (each PMC set in its own run), into gpurun_out/prof_bench/
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_bench; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-extras --no-cpu-baseline --steps 128 --warmup 32"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- $B > $out/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_fetch -- $B > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/bench_write -- $B > $out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/bench_sq -- $B > $out/bench_sq.log 2>&1
python3 $R/tools/pmc_summary.py $out > $out/bench_summary.txt 2>&1
cp $out/bench_trace/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
tail -1 $out/bench_trace.log > $out/bench_line_under_rocprof.json
cat $out/bench_summary.txt
