#!/bin/bash
# Round 2, GPU session S: detector stencils with four pixels per thread -- parity tests of the detector, timing, kernel stats.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-s}
timeout 900 python -m pytest tests -m gpu -q -k "akaze or features or liop or stage" 2>&1 | tail -3 | tee gpurun_out/r02_${T}_pytest.txt
rm -rf /tmp/prof_ak; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ak -- python tools/akaze_perf.py > gpurun_out/r02_${T}_akaze_perf.txt 2>&1
grep "^{" gpurun_out/r02_${T}_akaze_perf.txt | cut -c1-300
db=$(find /tmp/prof_ak -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_akaze_kernel_stats.txt 2>&1
head -16 gpurun_out/r02_${T}_akaze_kernel_stats.txt | cut -c1-130
timeout 200 python tools/akaze_perf.py 2>&1 | grep "^{" | cut -c1-300 | tee gpurun_out/r02_${T}_akaze_perf_noprof.txt
