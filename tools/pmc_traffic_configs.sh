#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, --kernel-trace only) of the dominant kernels of the other bench
# configs: c3 (popcount Hamming), liop144 (f32 tiles, G = 18), c5 on 96 images (graph search on byte rows).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for cfg in "c3" "liop144" "c5 --images 96"; do
  tag=$(echo $cfg | cut -d' ' -f1)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcc_${tag}_$ctr
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcc_${tag}_$ctr -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-opt-in > /tmp/pmcc_${tag}_$ctr.log 2>&1
    echo "## config $cfg, pass $ctr (rc=$?)"; python tools/pmc_summary.py /tmp/pmcc_${tag}_$ctr 2>&1 | grep -E "knn2|ann_search|ann_knn" | head -4
  done
done | tee gpurun_out/r02_y_pmc_traffic_other_configs.txt
