#!/bin/bash
# Round 2, GPU session N: state after the AC-RANSAC kernel work (sort-skipping bound, register sort, batched residual pass):
# all GPU tests, the default bench under rocprofv3, the phase split.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-n}
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r02_${T}_pytest_summary.txt
timeout 200 python tools/filter_phase_split.py 24 8192 2>&1 | tail -8 | tee gpurun_out/r02_${T}_filter_phase_split.txt
rm -rf /tmp/prof_n; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -- python bench.py --no-cpu-baseline > gpurun_out/r02_${T}_bench_c2_nocpu.json 2> gpurun_out/r02_${T}_bench_c2_nocpu.err
db=$(find /tmp/prof_n -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_bench_c2_nocpu_kernel_stats.txt 2>&1
head -c 1500 gpurun_out/r02_${T}_bench_c2_nocpu.json; echo; head -12 gpurun_out/r02_${T}_bench_c2_nocpu_kernel_stats.txt
