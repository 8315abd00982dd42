cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=600
tools/gpu_session.sh r05_e "pytest:akaze or test_stage or features"
export AK_STAGE=0 AK_BATCHES=8
echo "== product"; timeout 300 python tools/akaze_perf.py 2>&1 | grep "^{" | cut -c1-330
for v in "R3DM_AK_HEAD=0" "R3DM_AK_HEAD_WAVES=4000" "R3DM_AK_HEAD_WAVES=16000"; do
  echo "== $v"; env $v timeout 300 python tools/akaze_perf.py 2>&1 | grep "^{" | head -1 | cut -c1-330
done
GRID_MIN_US=20 GRID_HEAD=30 tools/gpu_session.sh r05_e "gridtrace:akaze_perf.py"
