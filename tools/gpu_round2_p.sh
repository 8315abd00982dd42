#!/bin/bash
# Round 2, GPU session P: bf16 row copy for the graph search (ImgDev::ann_rows16) -- ANN tests, A/B against the f32 rows
# (developer build, R3DM_ANN_ROWS16=0), C5 bench leg.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-p}
timeout 900 python -m pytest tests -m gpu -x -q -k "ann or kgraph or c5 or config_legs or golden" 2>&1 | tail -4 | tee gpurun_out/r02_${T}_pytest.txt
R3DM_ANN_ROWS16=0 timeout 300 python tools/ann_perf.py --images 16 --presets default 2>&1 | grep "^{" | cut -c1-420 | tee gpurun_out/r02_${T}_ann_perf_f32rows.txt
R3DM_ANN_ROWS16=1 timeout 300 python tools/ann_perf.py --images 16 --presets default 2>&1 | grep "^{" | cut -c1-420 | tee gpurun_out/r02_${T}_ann_perf_rows16.txt
timeout 900 python bench.py --config c5 --images 96 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_${T}_bench_c5.json 2> gpurun_out/r02_${T}_bench_c5.err; echo "c5 rc=$?"; head -c 1600 gpurun_out/r02_${T}_bench_c5.json
