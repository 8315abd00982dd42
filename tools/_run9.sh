cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -k "alternative_launch or graph_index_refuses or device_graphs or two_real_ranks or comm_wire or via_c_abi or rccl or filter or parity or fullsize or stage or multi" 2>&1 | tail -30
timeout 600 python tools/gpu_perf.py --images 200 --reps 2 --all-filters --integer-mfma 2>&1 | grep "^{" | cut -c1-600
