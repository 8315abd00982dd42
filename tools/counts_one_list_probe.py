"""Count tiles: the product's one-list kernel (0) against the two-list kernel (1 = R3DM_COUNTS_TWO_LISTS), developer build:
python tools/counts_one_list_probe.py <0|1> [images]"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["R3DM_COUNTS_TWO_LISTS"] = sys.argv[1]
from regard3d_amd import api, synth
api.use_developer_library()
from oracle import pyoracle as O
z = np.load("tests/golden/liop_match_ref.npz")
A = (z["hist0"].astype(np.float32) / z["norm0"][:, None]).astype(np.float32)
B = (z["hist1"].astype(np.float32) / z["norm1"][:, None]).astype(np.float32)
c = api.Context(0); c.set_split_mfma(True)
idx, dist = c.knn2(A, B); s = c.stats()
oi, od = O.knn2(A, B)
print("fixture: counts launches", s.n_counts_mfma, "exact fallback", s.n_exact_fallback, "of", len(B), "equal to oracle", bool(np.array_equal(idx, oi) and np.array_equal(dist, od)))
# a collection of 8192-row LIOP-shaped views
sc = synth.make_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 24, 8192, "liopc", seed=2024)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
pairs = sc.exhaustive_pairs()
c.match_pairs(pairs, 0.6, True)
t = time.time(); g = c.match_pairs(pairs, 0.6, True); el = time.time() - t
s = c.stats()
c.set_split_mfma(False)
g0 = c.match_pairs(pairs, 0.6, True)
print("collection: pairs", len(pairs), "kernel ms", round(s.ms_match_kernels, 2), "wall ms", round(el * 1e3, 2), "fallback", s.n_exact_fallback, "of", s.n_queries,
      "identical to f32 tiles", bool(np.array_equal(g.matches, g0.matches) and np.array_equal(g.pairs, g0.pairs)))
