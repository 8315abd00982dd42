cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "graph_index_refuses or count_tiles or mrpt or liop_match" 2>&1 | tail -5
bash tools/stage_tune.sh 24 1x8 2x8 3x8 2x6 3x6 4x6 2x12 3x4 4x4 | cut -c1-900
