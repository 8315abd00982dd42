#!/bin/bash
# Round 2, GPU session H: the headline bench at the final kernel source under rocprofv3 + its PMC traffic passes.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-h}
rm -rf /tmp/prof_c2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python bench.py > gpurun_out/r02_${T}_bench_c2.json 2> gpurun_out/r02_${T}_bench_c2.err
echo "rc=$?"; tail -c 600 gpurun_out/r02_${T}_bench_c2.json
db=$(find /tmp/prof_c2 -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_bench_c2_kernel_stats.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_$ctr -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmcb_$ctr.log 2>&1
  echo "## pass: $ctr (rc=$?)"; python tools/pmc_summary.py /tmp/pmcb_$ctr 2>&1 | grep -v "stage_" | head -8
done | tee gpurun_out/r02_${T}_pmc_bench_c2.txt
