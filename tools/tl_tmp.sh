cd $GRAFT_REPO_ROOT
cat > /tmp/run_dev.py <<'PY'
import sys
sys.argv = ["bench.py", "--config", "stage", "--steps", "6", "--warmup", "1", "--images", "24", "--stage-quick"]
sys.path.insert(0, ".")
from regard3d_amd import api
api.use_developer_library()
exec(compile(open("bench.py").read(), "bench.py", "exec"))
PY
R3DM_MATCH_TIMING=1 python /tmp/run_dev.py 2>&1 | grep -E "run_match_batch|facade match|stage_features" | cut -c1-300
