#!/bin/bash
# per-dispatch durations of the detector kernels on one 4000 x 3000 image (rocprofv3 kernel trace, CSV): which launches are big
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/one_detect.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from regard3d_amd import api
sys.path.insert(0, "tools")
rng = np.random.default_rng(7)
h, w = 3000, 4000
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
img = (0.5 + 0.25 * np.sin(xx / 37.0) * np.cos(yy / 53.0)).astype(np.float32)
img += rng.normal(0, 0.05, img.shape).astype(np.float32)
from scipy.ndimage import gaussian_filter
img = gaussian_filter(img, 1.5).astype(np.float32)
c = api.Context(0)
for _ in range(3):
    kps = c.detect_akaze(img, 0.001)
print(len(kps))
PY
rm -rf /tmp/tr_ak; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_ak -- python /tmp/one_detect.py > /tmp/tr_ak.log 2>&1
f=$(find /tmp/tr_ak -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if not k.startswith("r3dm::ak") and "ak_" not in k: continue
    v3 = sorted(v, reverse=True)
    n = len(v) // 3
    print(f"{k[:40]:40s} calls/img {n:4d}  total/img {sum(v)/3:8.1f} us   top: " + " ".join(f"{x:.0f}" for x in v3[:24:3]) + "   median %.1f" % v3[len(v3)//2])
PY
