#!/bin/bash
# features-stage concurrency sweep of bench.py --config stage (timed steps only): tools/stage_tune.sh <images> <AxB> [<AxB> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
N=$1; shift
for f in "$@"; do timeout 600 python bench.py --config stage --steps 2 --warmup 1 --images $N --stage-quick --stage-features $f 2>/dev/null | tail -1; done | tee gpurun_out/stage_tune_$N.txt
