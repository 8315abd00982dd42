# A/B of the XCD-aware workgroup mapping of l2_knn2_mfma_kernel: HBM-side fetch traffic (FETCH_SIZE, one --pmc pass each,
# --kernel-trace only) on the 48-view workload of tools/ab_l2.py (1,128 pairs).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for m in 0 1; do
  echo "== R3DM_XCD_MAP=$m"
  rm -rf /tmp/pmc_$m
  R3DM_XCD_MAP=$m timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_$m -- python tools/ab_l2.py 48 > /tmp/pmc_$m.log 2>&1
  grep '^{' /tmp/pmc_$m.log | cut -c1-160
  python tools/pmc_summary.py /tmp/pmc_$m 2>&1 | grep "l2_knn2" | head -3
done
