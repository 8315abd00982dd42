#!/bin/bash
# Round 2, GPU session D: E-filter bisect (7 variants), the whole -m gpu suite, features-stage concurrency probe, the two
# multi-GPU configs as one GPU of an 8-GPU node runs them (c4 / c5 --emulate-world 8).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-d}
for lib in product dev dbg t1 t2 t3 t4 vmwait nanfill; do
  case $lib in product|dev) arg=$lib;; *) arg=regard3d_amd/libr3dm_bisect_$lib.so;; esac
  timeout 120 python tools/efilter_probe2.py $arg 2 > /tmp/probe_$lib.log 2>&1
  echo "[$lib] E-filter reps without mismatch: $(grep ' E rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -c ' E rep' /tmp/probe_$lib.log); F/H clean: $(grep -E ' (F|H) rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -cE ' (F|H) rep' /tmp/probe_$lib.log)  $(tail -1 /tmp/probe_$lib.log | grep -i error | cut -c1-200)"
done | tee gpurun_out/r02_${T}_efilter_bisect.txt
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -15 gpurun_out/r02_${T}_pytest.log | cut -c1-250
rm -rf /tmp/prof_ak; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ak -- python tools/akaze_perf.py > gpurun_out/r02_${T}_akaze_perf.txt 2>&1
cat gpurun_out/r02_${T}_akaze_perf.txt | grep "^{" | cut -c1-300
db=$(find /tmp/prof_ak -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_akaze_kernel_stats.txt 2>&1
prof() {
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 2500 gpurun_out/r02_${T}_$tag.json
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
}
prof bench_c4_emu8 python bench.py --config c4 --emulate-world 8 --steps 1 --warmup 1
prof bench_c5_emu8 python bench.py --config c5 --emulate-world 8 --steps 1 --warmup 1 --cpu-seconds 2
