"""Runs the geometric filters repeatedly on the full C2-sized putative graph (790 pairs: more workgroups than CUs, the homography kernel
co-resident two per CU) and prints the distinct outcomes -- each filter must report exactly one.  Usage: filter_determinism.py EHHHFH"""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from regard3d_amd import api, synth
import os as _os
if any(k.startswith("R3DM_") for k in _os.environ):
    api.use_developer_library()      # R3DM_* knobs / traces exist only in the developer build (build.sh dev); otherwise measure the product
sc = synth.make_scene(200, 8192, "sift", seed=2002)
c = api.Context(0)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, synth.intrinsics())
g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
out = collections.Counter()
for ch in sys.argv[1]:
    r = {"F": c.filter_F, "E": c.filter_E, "H": c.filter_H}[ch](g)
    out[(ch, r.num_pairs, r.num_matches, round(c.stats().ms_filter_kernels, 1))] += 1
for k, v in sorted(out.items()): print(v, k, flush=True)
