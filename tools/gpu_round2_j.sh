#!/bin/bash
# Round 2, GPU session J: integer-kernel variant with two tiles per barrier; staging of pageable / pinned / device buffers;
# detector after the staging change; the whole -m gpu suite.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-j}
for v in 2 5 7 79; do
  echo "## R3DM_L2_INT_VARIANT=$v"; R3DM_L2_INT_VARIANT=$v timeout 120 python tools/gpu_perf.py --images 40 --reps 0 --integer-mfma 2>&1 | grep integer_mfma_identical
done > gpurun_out/r02_${T}_int_variants.txt 2>&1
cat gpurun_out/r02_${T}_int_variants.txt
timeout 200 python tools/gpu_perf.py --images 48 --reps 1 2>&1 | grep -E "set_image|staging" | tee gpurun_out/r02_${T}_staging.txt
timeout 200 python tools/akaze_perf.py 2>&1 | grep "^{" | tee gpurun_out/r02_${T}_akaze_perf.txt | cut -c1-300
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -8 gpurun_out/r02_${T}_pytest.log | cut -c1-250
