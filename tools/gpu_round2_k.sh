#!/bin/bash
# Round 2, GPU session K: what the driver runs at round end -- smoke(), the -m gpu suite, the default bench -- plus the kernel
# summary of the bench WITHOUT the CPU leg (whose reference-built index check adds ~200 small launches of the dominant kernel to
# the rocprofv3 average).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-k}
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -6 gpurun_out/r02_${T}_pytest.log | cut -c1-250
timeout 600 python bench.py > gpurun_out/r02_${T}_bench_c2.json 2> gpurun_out/r02_${T}_bench_c2.err; echo "bench rc=$?"; tail -c 400 gpurun_out/r02_${T}_bench_c2.json
rm -rf /tmp/prof_c2; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python bench.py --no-cpu-baseline > gpurun_out/r02_${T}_bench_c2_nocpu.json 2> /dev/null
db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_bench_c2_nocpu_kernel_stats.txt 2>&1
head -8 gpurun_out/r02_${T}_bench_c2_nocpu_kernel_stats.txt
