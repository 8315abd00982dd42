"""Where does the essential-matrix filter spend its time?  iterations / models per pair and kernel time vs the iteration cap."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
import os as _os
if any(k.startswith("R3DM_") for k in _os.environ):
    api.use_developer_library()      # R3DM_* knobs / traces exist only in the developer build (build.sh dev); otherwise measure the product
sc = synth.make_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 60, 8192, "sift", seed=2002)
c = api.Context(0)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, synth.intrinsics())
g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
print("putative pairs", g.num_pairs)
only = sys.argv[2] if len(sys.argv) > 2 else "FEH"
for name, fn in (("F", lambda it: c.filter_F(g, 4.0, it)), ("E", lambda it: c.filter_E(g, 4.0, it)), ("H", lambda it: c.filter_H(g, 4.0, it))):
    if name not in only: continue
    for it in (128, 512, 2048):
        fn(it); ms = c.stats().ms_filter_kernels
        rep = [r for r in c.filter_report() if r[2]]
        iters = np.array([r[2] for r in rep]); models = np.array([r[3] for r in rep])
        print(json.dumps(dict(filter=name, max_iter=it, ms_kernel=round(ms, 2), pairs=len(rep), iters_mean=float(iters.mean()), iters_max=int(iters.max()),
                              models_mean=float(models.mean()), models_per_iter=float(models.sum() / iters.sum()))))
