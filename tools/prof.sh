#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats first, then the PMC passes, each in its own run (gpurun refuses --pmc together
# with the trace domains; FETCH_SIZE / WRITE_SIZE / the SQ sets do not fit one pass).  Run ON THE GPU BOX:
#
#   tools/prof.sh <tag> bench   [default driver ...]        the bench command's headline launches per launch shape
#                                                            (default: `bench.py`; driver: `bench.py --steps 20 --warmup 5`)
#   tools/prof.sh <tag> kernels [c5 c4 fogbatch xor c3 ...]  one configuration each through tools/prof_frames.py
#   tools/prof.sh <tag> case    <name> <prof_frames.py args> any other prof_frames.py invocation (VK_PARAMS / VK_NOSKIP / VK_BATCH apply)
#
# Output: gpurun_out/prof_<tag>/<name>/{trace,fetch,write,sq1,sq2,sq3}/, kernel_stats.csv, summary.txt (tools/pmc_summary.py); for
# `bench` also bench_line_under_rocprof.json per shape and pmc_traffic_c2.json (tools/pmc_traffic.py).  Copy what is to be judged
# into profiles/ under the round's prefix.
set -u
tag=${1:?tag}; what=${2:?bench|kernels|case}; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; out=$R/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
SQ3="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"

passes() {  # passes <dir> <program and arguments...>: the program itself after `--`, never a wrapper
  local o=$1; shift; rm -rf $o; mkdir -p $o
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- "$@" > $o/trace.log 2>&1 || { echo "trace failed ($o)"; tail -5 $o/trace.log; return 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- "$@" > $o/fetch.log 2>&1 || { echo "fetch failed ($o)"; return 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- "$@" > $o/write.log 2>&1 || { echo "write failed ($o)"; return 1; }
  rocprofv3 --pmc $SQ1 --output-format csv -d $o/sq1 -- "$@" > $o/sq1.log 2>&1 || { echo "sq1 failed ($o)"; return 1; }
  rocprofv3 --pmc $SQ2 --output-format csv -d $o/sq2 -- "$@" > $o/sq2.log 2>&1 || { echo "sq2 failed ($o)"; return 1; }
  rocprofv3 --pmc $SQ3 --output-format csv -d $o/sq3 -- "$@" > $o/sq3.log 2>&1 || echo "sq3 failed ($o; kept going)"
  cp $o/trace/*/*kernel_stats.csv $o/kernel_stats.csv 2>/dev/null
  python3 $R/tools/pmc_summary.py $o > $o/summary.txt 2>&1
  grep -v "pack_\|generate_\|clear_\|dist_pass\|build_" $o/summary.txt | head -60
}

case $what in
  bench)
    for shape in ${@:-default driver}; do
      if [ $shape = default ]; then A=""; else A="--steps 20 --warmup 5"; fi
      echo "== bench $shape"
      passes $out/$shape python3 $R/bench.py --no-extras --no-cpu-baseline --headline-only $A ${BENCH_ARGS:-} || exit 1
      grep "^{" $out/$shape/trace.log | tail -1 > $out/$shape/bench_line_under_rocprof.json
    done
    python3 $R/tools/pmc_traffic.py $out c2 > $out/pmc_traffic_c2.json && cat $out/pmc_traffic_c2.json;;
  kernels)
    for c in ${@:-c5 c4 fogbatch xor c3}; do
      unset VK_NOSKIP VK_BATCH
      case $c in
        c5) A="c5 s8 5";; c4) A="c4 s8 8";; xor) A="xor auto 30";; c3) A="c3 auto 6";;
        fogbatch) A="c2fog p16 10"; export VK_NOSKIP=1 VK_BATCH=8;;
        *) echo "unknown case $c"; exit 1;;
      esac
      echo "== $c"; passes $out/$c python3 $R/tools/prof_frames.py $A || exit 1
    done;;
  case)
    name=${1:?name}; shift
    echo "== $name"; passes $out/$name python3 $R/tools/prof_frames.py "$@";;
  *) echo "bench | kernels | case"; exit 1;;
esac
