# PMC look at the integer fast path kernel (one counter group per pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TD)_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_avail_ta.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_'); rm -rf /tmp/pi_$tag
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pi_$tag -- python tools/gpu_perf.py --images 16 --feat 8192 --reps 0 --integer-mfma > /tmp/pi_$tag.log 2>&1
  echo "## pass: $grp (rc=$?)"; python tools/pmc_summary.py /tmp/pi_$tag 2>&1 | grep -E "l2_knn2" | head -4
done | tee gpurun_out/pmc_int.txt
