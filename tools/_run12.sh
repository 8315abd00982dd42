cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=1500 PYTEST_TAIL=25
tools/gpu_session.sh r05_h pytest smoke
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-leg 2>/dev/null | tail -1 | cut -c1-1500
