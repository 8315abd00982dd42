#!/bin/bash
# PMC look at the graph-search kernel (tools/ann_perf.py, product library): instruction mix, waits, L2 requests.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $grp | tr ' ' '_'); rm -rf /tmp/pm_$tag
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pm_$tag -- python tools/ann_perf.py --images 12 --presets default > /tmp/pm_$tag.log 2>&1
  echo "## pass: $grp (rc=$?)"; python tools/pmc_summary.py /tmp/pm_$tag 2>&1 | grep -E "ann_search" | head -3
done | tee gpurun_out/r02_r_pmc_ann_search.txt
