# HBM-side traffic of the dominant kernel on the full C2 bench: separate --pmc passes (FETCH_SIZE, then WRITE_SIZE), --kernel-trace only.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  timeout 420 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_$ctr -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmcb_$ctr.log 2>&1
  echo "## pass: $ctr (rc=$?)"; python tools/pmc_summary.py /tmp/pmcb_$ctr 2>&1 | grep -v "stage_" | head -6
done | tee gpurun_out/r01_g_pmc_bench_c2.txt
