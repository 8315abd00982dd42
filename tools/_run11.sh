cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "graph_index_refuses or count_tiles or liop_match or test_stage" 2>&1 | tail -5
bash tools/stage_tune.sh 24 2x8 3x8 2x12 | cut -c1-900
