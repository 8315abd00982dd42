"""Where the wall time of r3dm_match_pairs goes on stage-sized views (24 x 28k LIOP-144 rows): kernels vs host post-processing."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 28000
sc = synth.make_scene(n_img, n_feat, "liop", seed=2002)
c = api.Context(0)
c.set_split_mfma(True)
t = time.time()
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
print(json.dumps(dict(set_image_s=time.time() - t)))
pairs = sc.exhaustive_pairs()
for rep in range(3):
    t = time.time(); g = c.match_pairs(pairs, 0.6, True); dt = time.time() - t
    s = c.stats()
    print(json.dumps(dict(rep=rep, wall_ms=dt * 1e3, ms_wall_match=s.ms_wall_match, ms_match_kernels=s.ms_match_kernels, ms_wall_match_post=s.ms_wall_match_post,
                          launches=s.n_match_launches, fallback=s.n_exact_fallback, split=s.n_split_mfma, matches=g.num_matches, pairs=len(g.pairs))), flush=True)
