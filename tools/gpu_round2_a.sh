#!/bin/bash
# Round 2, GPU session A (run on the MI355X box through gpurun): the -m gpu suite, then every BASELINE config's bench leg under
# rocprofv3 --kernel-trace --stats (JSON line + kernel summary per config), then the E / H / F filters on the whole C2 graph.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-a}
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -5 gpurun_out/r02_${T}_pytest.log
prof() {   # $1 = tag, rest = command
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 600 gpurun_out/r02_${T}_$tag.json | cut -c1-400
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
}
prof bench_c2 python bench.py
prof bench_c3 python bench.py --config c3
prof bench_liop144 python bench.py --config liop144
prof bench_c5 python bench.py --config c5 --images 96 --steps 2
prof filters_c2 python tools/gpu_perf.py --images 200 --reps 1 --all-filters
ls -la gpurun_out | tail -20
