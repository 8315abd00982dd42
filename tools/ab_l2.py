"""Within-process A/B of the L2 kernel variants is not possible (variant is latched per process), so this
runs one variant per invocation on a mid-size workload and prints kernel TFLOP/s (HIP events)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api, synth
import os as _os
if any(k.startswith("R3DM_") for k in _os.environ):
    api.use_developer_library()      # R3DM_* knobs / traces exist only in the developer build (build.sh dev); otherwise measure the product
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 48
kind = sys.argv[2] if len(sys.argv) > 2 else "sift"
if kind == "sift":
    descs, xys, _ = synth.make_scene_torch(n_img, 8192, seed=2002, device="cuda")
else:
    sc = synth.make_scene(n_img, 8192, kind, seed=2002)
    descs = [torch.from_numpy(d).cuda() for d in sc.descs]; xys = [torch.from_numpy(x).cuda() for x in sc.xys]
c = api.Context(0)
for i in range(n_img): c.set_image(i, descs[i], xys[i], 4000, 3000)
ii, jj = np.triu_indices(n_img, k=1); pairs = np.stack([ii, jj], 1).astype(np.uint32)
res = []
for rep in range(4):
    g = c.match_pairs(pairs, 0.6, True); s = c.stats()
    res.append(s.algorithmic_flops / (s.ms_match_kernels * 1e-3) / 1e12)
print(json.dumps({"variant": os.environ.get("R3DM_L2_VARIANT", "default"), "pairs": len(pairs), "tflops": [round(x, 2) for x in res], "kind": kind, "matches": g.num_matches, "queries": s.n_queries, "fallback": s.n_exact_fallback}))
