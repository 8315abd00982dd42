cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=600
tools/gpu_session.sh r05_c "pytest:liop or test_stage or features" "proftool:liop_fused_perf.py" "bench:stage:--steps 3 --warmup 1 --images 24 --stage-quick"
