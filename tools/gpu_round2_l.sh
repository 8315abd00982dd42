#!/bin/bash
# Round 2, GPU session L: PMC look at the two new opt-in kernels (matrix pipe busy, L1 address path, LDS, clock) and BASELINE config
# C4 in full on one GPU (499,500 pairs).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
T=${1:-l}
for kind in liop akaze; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    tag=${kind}_$(echo $grp | tr ' ' '_'); rm -rf /tmp/pm_$tag
    timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pm_$tag -- python tools/fastpath_perf.py $kind 16 > /tmp/pm_$tag.log 2>&1
    echo "## $kind pass: $grp (rc=$?)"; grep "^{" /tmp/pm_$tag.log | cut -c1-200; python tools/pmc_summary.py /tmp/pm_$tag 2>&1 | grep -E "l2_knn2|hamming_knn2" | head -6
  done
done | tee gpurun_out/r02_${T}_pmc_fastpaths.txt
timeout 900 python bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-opt-in > gpurun_out/r02_${T}_bench_c4_full.json 2> gpurun_out/r02_${T}_bench_c4_full.err; echo "c4 rc=$?"; tail -c 1500 gpurun_out/r02_${T}_bench_c4_full.json
