# PMC look at the integer fast path kernel (one counter group per pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*(MFMA|VALU|VMEM|WAIT|BUSY|LDS|SALU|ISSUE)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_avail.txt
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | tr ' ' '_'); rm -rf /tmp/pi_$tag
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pi_$tag -- python tools/gpu_perf.py --images 16 --feat 8192 --reps 0 --integer-mfma > /tmp/pi_$tag.log 2>&1
  echo "## pass: $grp (rc=$?)"; python tools/pmc_summary.py /tmp/pi_$tag 2>&1 | grep -E "l2_knn2" | head -4
done | tee gpurun_out/pmc_int.txt
