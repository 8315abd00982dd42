cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "alternative_launch or graph_index_refuses" 2>&1 | tail -60
timeout 900 python -m pytest tests -m gpu -q -k "filter or parity or fullsize or smoke or stage" 2>&1 | tail -15
timeout 600 python tools/gpu_perf.py --images 200 --reps 2 --all-filters --integer-mfma 2>&1 | grep "^{" | cut -c1-600
