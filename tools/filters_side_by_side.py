"""F / E / H one after the other vs r3dm_filter_FEH on a collection with few, long pairs: tools/filters_side_by_side.py [images] [features]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 28000
sc = synth.make_scene(n_img, n_feat, "sift", seed=2002)
c = api.Context(0)
c.set_integer_mfma(True)
K = synth.intrinsics()
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
cnt = np.diff(g.offsets.astype(np.int64))
print(json.dumps(dict(pairs=int(g.num_pairs), matches=int(g.num_matches), longest=int(cnt.max()), median=int(np.median(cnt)))))
for rep in range(2):
    out = {}
    for name, fn in (("F", c.filter_F), ("E", c.filter_E), ("H", c.filter_H)):
        t = time.time(); fn(g); out[name] = dict(wall_ms=(time.time() - t) * 1e3, kernel_ms=c.stats().ms_filter_kernels)
    t = time.time(); _, msk, msw = c.filter_FEH(g, "FEH"); out["FEH"] = dict(wall_ms=(time.time() - t) * 1e3, kernel_ms=msk.tolist(), call_ms=msw.tolist())
    t = time.time(); _, msk, msw = c.filter_FEH(g, "FH"); out["FH"] = dict(wall_ms=(time.time() - t) * 1e3, kernel_ms=msk.tolist())
    print(json.dumps(out), flush=True)
