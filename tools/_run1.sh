cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=900
tools/gpu_session.sh r05_a "pytest:liop or stage or akaze or filter_product or features or extract" "prof:stage:--steps 3 --warmup 1 --images 24 --stage-quick"
