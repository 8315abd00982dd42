# HBM-side fetch traffic of l2_knn2_mfma_kernel on the 48-view workload of tools/ab_l2.py (1,128 pairs): one --pmc pass
# (FETCH_SIZE, --kernel-trace only) per setting of R3DM_XCD_MAP (1 = all workgroups of a pair on one XCD, the default).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for m in ${1:-0 1}; do
  echo "== R3DM_XCD_MAP=$m"
  rm -rf /tmp/pmc_$m
  R3DM_XCD_MAP=$m timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$m -- python tools/ab_l2.py 48 > /tmp/pmc_$m.log 2>&1
  grep '^{' /tmp/pmc_$m.log | cut -c1-160
  python tools/pmc_summary.py /tmp/pmc_$m 2>&1 | grep "l2_knn2" | head -3
done
