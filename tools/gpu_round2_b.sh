#!/bin/bash
# Round 2, GPU session B: filter parity probe on both library builds, the whole -m gpu suite (no -x), liop144 + c5 bench legs.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-b}
( timeout 200 python tools/efilter_probe2.py product 3; timeout 200 python tools/efilter_probe2.py dev 3 ) > gpurun_out/r02_${T}_efilter_probe.txt 2>&1
grep -c "mismatching pairs: \[\]" gpurun_out/r02_${T}_efilter_probe.txt; grep -v "mismatching pairs: \[\]" gpurun_out/r02_${T}_efilter_probe.txt | head -20
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -40 gpurun_out/r02_${T}_pytest.log | cut -c1-300
prof() {
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 1500 gpurun_out/r02_${T}_$tag.json | cut -c1-1500
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
}
prof bench_liop144 python bench.py --config liop144 --steps 2
prof bench_c5 python bench.py --config c5 --images 96 --steps 2
