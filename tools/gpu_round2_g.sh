#!/bin/bash
# Round 2, GPU session G: the essential-matrix kernel as shipped (own TU, SGPR spills to memory) + the LDS hand-over variant,
# filter timings of both on the whole C2 graph, then the whole -m gpu suite.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-g}
for lib in product ldsflag; do
  case $lib in product|dev) arg=$lib;; *) arg=regard3d_amd/libr3dm_bisect_$lib.so;; esac
  timeout 120 python tools/efilter_probe2.py $arg 3 > /tmp/probe_$lib.log 2>&1
  echo "[$lib] E-filter reps without mismatch: $(grep ' E rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -c ' E rep' /tmp/probe_$lib.log); F/H clean: $(grep -E ' (F|H) rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -cE ' (F|H) rep' /tmp/probe_$lib.log)  $(tail -1 /tmp/probe_$lib.log | grep -i error | cut -c1-200)"
done | tee gpurun_out/r02_${T}_efilter_variants.txt
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -8 gpurun_out/r02_${T}_pytest.log | cut -c1-250
rm -rf /tmp/prof_f; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -- python tools/gpu_perf.py --images 200 --reps 1 --all-filters > gpurun_out/r02_${T}_filters_c2.json 2>&1
grep "^{" gpurun_out/r02_${T}_filters_c2.json | cut -c1-400
db=$(find /tmp/prof_f -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_filters_c2_kernel_stats.txt 2>&1
