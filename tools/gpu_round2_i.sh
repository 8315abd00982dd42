#!/bin/bash
# Round 2, GPU session I: detector with fused FED steps + fused Hessian (bit-exactness tests, timings), the two new full-size tests.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-i}
( time timeout 900 python -m pytest tests/test_gpu_akaze.py tests/test_golden_fixtures.py tests/test_cpp_host.py "tests/test_gpu_fullsize.py::test_c5_graph_matcher_at_16384_rows" "tests/test_gpu_fullsize.py::test_c3_binary_collection_of_24_views" -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -15 gpurun_out/r02_${T}_pytest.log | cut -c1-300
rm -rf /tmp/prof_ak; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ak -- python tools/akaze_perf.py > gpurun_out/r02_${T}_akaze_perf.txt 2>&1
grep "^{" gpurun_out/r02_${T}_akaze_perf.txt | cut -c1-300
db=$(find /tmp/prof_ak -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_akaze_kernel_stats.txt 2>&1
head -16 gpurun_out/r02_${T}_akaze_kernel_stats.txt
