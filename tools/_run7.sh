cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=1500 PYTEST_TAIL=25
tools/gpu_session.sh r05_g pytest "bench:stage:--steps 3 --warmup 1 --images 24 --stage-quick"
export AK_STAGE=0 AK_BATCHES=8
timeout 300 python tools/akaze_perf.py 2>&1 | grep "^{" | cut -c1-330
