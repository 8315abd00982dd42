cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "akaze or test_stage or features" 2>&1 | tail -4
export AK_STAGE=0 AK_BATCHES=8
GRID_MIN_US=15 GRID_HEAD=70 tools/gpu_session.sh r05_i "gridtrace:akaze_perf.py" | grep -E "ms_kernels|mask|refine|prune"
