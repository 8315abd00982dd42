#!/bin/bash
# Round 2, GPU session C: bisect of the essential-matrix filter nondeterminism, new opt-in paths (MFMA Hamming, staged index,
# LDS-shared integer kernel variants A/B), c3 bench with the opt-in leg, PMC traffic passes of the c2 bench.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-c}
for lib in product regard3d_amd/libr3dm_bisect_dbg.so regard3d_amd/libr3dm_bisect_trace.so regard3d_amd/libr3dm_bisect_vmwait.so; do
  timeout 120 python tools/efilter_probe2.py $lib 2 2>&1 | grep " E rep" | sed "s#^#[$lib] #"
done > gpurun_out/r02_${T}_efilter_bisect.txt 2>&1
cut -c1-260 gpurun_out/r02_${T}_efilter_bisect.txt
( time timeout 600 python -m pytest tests/test_gpu_hamming_mfma.py tests/test_cpp_host.py tests/test_gpu_integer_mfma.py -m gpu -q ) > gpurun_out/r02_${T}_pytest.log 2>&1
tail -25 gpurun_out/r02_${T}_pytest.log | cut -c1-250
for v in 2 5 59 6 9; do
  echo "## R3DM_L2_INT_VARIANT=$v"; R3DM_L2_INT_VARIANT=$v timeout 120 python tools/gpu_perf.py --images 40 --reps 0 --integer-mfma 2>&1 | grep integer_mfma_identical
done > gpurun_out/r02_${T}_int_variants.txt 2>&1
cat gpurun_out/r02_${T}_int_variants.txt
prof() {
  local tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- "$@" > gpurun_out/r02_${T}_$tag.json 2> gpurun_out/r02_${T}_$tag.err
  echo "## $tag rc=$?"; tail -c 1800 gpurun_out/r02_${T}_$tag.json
  local db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/r02_${T}_${tag}_kernel_stats.txt 2>&1
}
prof bench_c3 python bench.py --config c3 --steps 2 --cpu-seconds 4
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcb_$ctr -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmcb_$ctr.log 2>&1
  echo "## pass: $ctr (rc=$?)"; python tools/pmc_summary.py /tmp/pmcb_$ctr 2>&1 | grep -v "stage_" | head -8
done | tee gpurun_out/r02_${T}_pmc_bench_c2.txt
