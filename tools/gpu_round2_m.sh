#!/bin/bash
# Round 2, GPU session M: register-resident sort + atomic compaction in the AC-RANSAC kernels -- parity / determinism of the three
# filters against the oracle, the phase split again, the filter tests, the C2 bench.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-m}
timeout 300 python tools/efilter_probe2.py product 3 2>&1 | tail -20 | tee gpurun_out/r02_${T}_filter_probe.txt
timeout 200 python tools/filter_phase_split.py 24 8192 2>&1 | tail -8 | tee gpurun_out/r02_${T}_filter_phase_split.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "filter or stage or fullsize or parity or multi" 2>&1 | tail -5 | tee gpurun_out/r02_${T}_pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_${T}_bench_c2.json 2> gpurun_out/r02_${T}_bench_c2.err; tail -c 900 gpurun_out/r02_${T}_bench_c2.json
