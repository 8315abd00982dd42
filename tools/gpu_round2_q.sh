#!/bin/bash
# Round 2, GPU session Q: C5 with the bf16 row copy in the graph search -- shard 0 of 8 of the BASELINE collection (what one GPU of
# the node runs) under rocprofv3, with the CPU leg.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-q}
prof() {
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 3000 gpurun_out/r02_${T}_$tag.json
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
  head -8 gpurun_out/r02_${T}_${tag}_kernel_stats.txt
}
prof bench_c5_emu8 python bench.py --config c5 --emulate-world 8 --steps 1 --warmup 1 --cpu-seconds 2
