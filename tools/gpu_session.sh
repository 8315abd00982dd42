#!/bin/bash
# One parameterised runner for the GPU sessions (replaces the per-session gpu_round2_*.sh scripts): every step writes under
# gpurun_out/<tag>_*, the files worth judging are then copied into profiles/.
#
#   tools/gpu_session.sh <tag> <step> [<step> ...]
#
# steps:
#   pytest[:<-k expression>]        python -m pytest tests -m gpu (-k ...)            -> <tag>_pytest.txt
#   smoke                           __graft_entry__.smoke()                            -> <tag>_smoke.txt
#   bench:<config>[:<extra args>]   python bench.py --config <config> <extra>          -> <tag>_bench_<config>.json / .err
#   prof:<config>[:<extra args>]    the same under rocprofv3 --kernel-trace --stats    -> <tag>_bench_<config>.json + _kernel_stats.txt
#   akaze                           tools/akaze_perf.py (+ kernel stats)              -> <tag>_akaze_perf.txt + _akaze_kernel_stats.txt
#   tool:<script.py>[:<args>]       python tools/<script.py> <args>                   -> <tag>_<script>.txt
#   proftool:<script.py>[:<args>]   the same under rocprofv3 --kernel-trace --stats    -> <tag>_<script>.txt + _<script>_kernel_stats.txt
#   gridtrace:<script.py>[:<args>] rocprofv3 --kernel-trace CSV of python tools/<script.py>, durations bucketed by (kernel, grid)
#   exe:<binary>[:<args>]           a prebuilt binary of the tree (tools/ubench/*.bin)   -> <tag>_<name>.txt
#   pmc:<group+group>:<config>[:<extra args>]   separate rocprofv3 --pmc passes, one per '+'-separated group of comma-separated counters (--kernel-trace only)
#                                                                                        -> <tag>_pmc_<config>.txt
# Every step runs under its own `timeout`; nothing here kills by pattern.
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=$1; shift
stats() {   # $1 = profile dir, $2 = output file
  local db; db=$(find $1 -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > $2 2>&1 && head -${STATS_HEAD:-24} $2 | cut -c1-150
}
for step in "$@"; do
  IFS=':' read -r kind a1 a2 a3 <<< "$step"
  echo "=== [$T] $step"
  case $kind in
    pytest)
      # (the whole log is kept beside the tail: a run that dies with a signal says which test it was in only at the top of its dump)
      if [ -n "$a1" ]; then timeout ${PYTEST_TIMEOUT:-1500} python -X faulthandler -m pytest tests -m gpu -q -x -v -k "$a1" > gpurun_out/${T}_pytest_full.txt 2>&1
      else timeout ${PYTEST_TIMEOUT:-1500} python -X faulthandler -m pytest tests -m gpu -q -v > gpurun_out/${T}_pytest_full.txt 2>&1; fi
      echo "rc=$?" >> gpurun_out/${T}_pytest_full.txt
      grep -v "PASSED\|^$" gpurun_out/${T}_pytest_full.txt | tail -${PYTEST_TAIL:-15} | tee gpurun_out/${T}_pytest.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/${T}_smoke.txt ;;
    bench)
      timeout ${BENCH_TIMEOUT:-1200} python bench.py --config $a1 $a2 > gpurun_out/${T}_bench_$a1.json 2> gpurun_out/${T}_bench_$a1.err
      echo "rc=$?"; cut -c1-1500 gpurun_out/${T}_bench_$a1.json; tail -3 gpurun_out/${T}_bench_$a1.err ;;
    prof)
      rm -rf /tmp/prof_$a1
      timeout ${BENCH_TIMEOUT:-1200} rocprofv3 --kernel-trace --stats -d /tmp/prof_$a1 -- python bench.py --config $a1 $a2 > gpurun_out/${T}_bench_$a1.json 2> gpurun_out/${T}_bench_$a1.err
      echo "rc=$?"; cut -c1-1500 gpurun_out/${T}_bench_$a1.json; tail -3 gpurun_out/${T}_bench_$a1.err
      stats /tmp/prof_$a1 gpurun_out/${T}_bench_${a1}_kernel_stats.txt ;;
    akaze)
      rm -rf /tmp/prof_ak
      timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ak -- python tools/akaze_perf.py > gpurun_out/${T}_akaze_perf_profiled.txt 2>&1
      stats /tmp/prof_ak gpurun_out/${T}_akaze_kernel_stats.txt
      timeout 600 python tools/akaze_perf.py 2>&1 | grep "^{" | cut -c1-400 | tee gpurun_out/${T}_akaze_perf.txt ;;
    tool)
      n=$(basename $a1 .py)
      timeout ${TOOL_TIMEOUT:-900} python tools/$a1 $a2 2>&1 | tail -${TOOL_TAIL:-40} | cut -c1-400 | tee gpurun_out/${T}_$n.txt ;;
    proftool)
      n=$(basename $a1 .py); rm -rf /tmp/prof_$n
      timeout ${TOOL_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -- python tools/$a1 $a2 2>&1 | grep -v "^W\|rocprofv3" | tail -${TOOL_TAIL:-40} | cut -c1-400 | tee gpurun_out/${T}_$n.txt
      stats /tmp/prof_$n gpurun_out/${T}_${n}_kernel_stats.txt ;;
    gridtrace)   # gridtrace:<script.py>[:<args>]: per-launch durations by (kernel, grid) -> <tag>_<script>_by_grid.txt
      n=$(basename $a1 .py); rm -rf /tmp/tr_$n
      timeout ${TOOL_TIMEOUT:-900} rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -- python tools/$a1 $a2 > /tmp/tr_$n.log 2>&1
      grep "^{" /tmp/tr_$n.log | cut -c1-300
      python tools/trace_by_grid.py /tmp/tr_$n "r3dm::" ${GRID_MIN_US:-100} | tee gpurun_out/${T}_${n}_by_grid.txt | head -${GRID_HEAD:-70} ;;
    pmctool)   # pmctool:<group+group+...>:<script.py>[:<args>]: one rocprofv3 --pmc pass per group (counters of a group comma-separated)
      n=$(basename $a2 .py)
      for grp in ${a1//+/ }; do
        rm -rf /tmp/pmt_$n
        timeout ${TOOL_TIMEOUT:-600} rocprofv3 --pmc ${grp//,/ } --kernel-trace --output-format csv -d /tmp/pmt_$n -- python tools/$a2 $a3 > /tmp/pmt_$n.log 2>&1
        echo "## pass: ${grp//,/ } (rc=$?)"; python tools/pmc_summary.py /tmp/pmt_$n ${PMC_BY_GRID:+--by-grid} 2>&1 | grep -E "${PMC_FILTER:-r3dm}" | head -${PMC_HEAD:-12}
      done | tee gpurun_out/${T}_pmc_$n.txt ;;
    pmc)   # pmc:<group+group+...>:<config>[:<extra args>]: one rocprofv3 --pmc pass per group (counters of a group comma-separated)
      for grp in ${a1//+/ }; do
        rm -rf /tmp/pmc_pass
        timeout ${BENCH_TIMEOUT:-900} rocprofv3 --pmc ${grp//,/ } --kernel-trace --output-format csv -d /tmp/pmc_pass -- python bench.py --config $a2 $a3 --no-cpu-baseline > /tmp/pmc_pass.log 2>&1
        echo "## pass: ${grp//,/ } (rc=$?)"; python tools/pmc_summary.py /tmp/pmc_pass 2>&1 | grep -v "stage_" | head -${PMC_HEAD:-8}
      done | tee gpurun_out/${T}_pmc_$a2.txt ;;
    exe)   # exe:<path of a prebuilt binary in the tree>[:<args>]                      -> <tag>_<name>.txt
      n=$(basename $a1 .bin)
      timeout ${TOOL_TIMEOUT:-300} $a1 $a2 2>&1 | tail -${TOOL_TAIL:-40} | cut -c1-300 | tee gpurun_out/${T}_$n.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
