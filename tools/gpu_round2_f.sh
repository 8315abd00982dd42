#!/bin/bash
# Round 2, GPU session F: essential-matrix filter mitigations (solver called from uniform control flow / SGPR spills to memory),
# detector probe without the graph replay.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-f}
for lib in olddbg uniform nosgprvgpr both; do
  arg=regard3d_amd/libr3dm_bisect_$lib.so
  timeout 120 python tools/efilter_probe2.py $arg 3 > /tmp/probe_$lib.log 2>&1
  echo "[$lib] E-filter reps without mismatch: $(grep ' E rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -c ' E rep' /tmp/probe_$lib.log); F/H clean: $(grep -E ' (F|H) rep' /tmp/probe_$lib.log | grep -c 'pairs: \[\]') of $(grep -cE ' (F|H) rep' /tmp/probe_$lib.log)  $(tail -1 /tmp/probe_$lib.log | grep -i error | cut -c1-200)"
done | tee gpurun_out/r02_${T}_efilter_variants.txt
timeout 200 python tools/akaze_perf.py 2>&1 | grep "^{" | tee gpurun_out/r02_${T}_akaze_perf.txt | cut -c1-300
