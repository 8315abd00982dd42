#!/bin/bash
# Round 2, GPU session E: essential-matrix filter with the sample handed over in LDS (vs the two old call shapes), the whole
# -m gpu suite, detector probe (captured launch sequence, batched pruning, K contexts), liop144 bench, filter timings on C2.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-e}
for lib in product old olddbg; do
  case $lib in product|dev) arg=$lib;; *) arg=regard3d_amd/libr3dm_bisect_$lib.so;; esac
  timeout 120 python tools/efilter_probe2.py $arg 3 > /tmp/probe_$lib.log 2>&1
  echo "[$lib] E-filter reps without mismatch: $(grep ' E rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -c ' E rep' /tmp/probe_$lib.log); F/H clean: $(grep -E ' (F|H) rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -cE ' (F|H) rep' /tmp/probe_$lib.log)  $(tail -1 /tmp/probe_$lib.log | grep -i error | cut -c1-200)"
done | tee gpurun_out/r02_${T}_efilter_variants.txt
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -12 gpurun_out/r02_${T}_pytest.log | cut -c1-250
rm -rf /tmp/prof_ak; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ak -- python tools/akaze_perf.py > gpurun_out/r02_${T}_akaze_perf.txt 2>&1
grep "^{" gpurun_out/r02_${T}_akaze_perf.txt | cut -c1-300; grep -v "^{" gpurun_out/r02_${T}_akaze_perf.txt | tail -5 | cut -c1-300
db=$(find /tmp/prof_ak -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_akaze_kernel_stats.txt 2>&1
prof() {
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 2200 gpurun_out/r02_${T}_$tag.json
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
}
prof bench_liop144 python bench.py --config liop144 --steps 2 --cpu-seconds 6
prof filters_c2 python tools/gpu_perf.py --images 200 --reps 1 --all-filters
