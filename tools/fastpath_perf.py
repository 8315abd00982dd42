"""One pass of the default matcher and one of its opt-in fast path on a synthetic collection (PMC / rocprofv3 target):
    python tools/fastpath_perf.py sift|liop|akaze [images] [features]
sift -> r3dm_set_integer_mfma, liop -> r3dm_set_split_mfma, akaze -> r3dm_set_hamming_mfma; prints kernel times and whether the
graphs are identical."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()

kind = sys.argv[1] if len(sys.argv) > 1 else "liop"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n_feat = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
sc = synth.make_scene(n_img, n_feat, kind, seed=2002)
c = api.Context(0)
binary = kind == "akaze"
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000, binary=binary)
pairs = sc.exhaustive_pairs()
ratio, sq = (0.8, False) if binary else (0.6, True)
setter = {"sift": c.set_integer_mfma, "liop": c.set_split_mfma, "akaze": c.set_hamming_mfma}[kind]
c.match_pairs(pairs[:8], ratio, sq)
g0 = c.match_pairs(pairs, ratio, sq); s0 = c.stats()
setter(True)
c.match_pairs(pairs[:8], ratio, sq)
g1 = c.match_pairs(pairs, ratio, sq); s1 = c.stats()
same = np.array_equal(g0.pairs, g1.pairs) and np.array_equal(g0.offsets, g1.offsets) and np.array_equal(g0.matches, g1.matches)
print(json.dumps(dict(kind=kind, pairs=len(pairs), default_ms_kernel=s0.ms_match_kernels, fast_ms_kernel=s1.ms_match_kernels,
                      speedup=s0.ms_match_kernels / s1.ms_match_kernels, identical=bool(same), fallback_default=int(s0.n_exact_fallback),
                      fallback_fast=int(s1.n_exact_fallback), launches=dict(int=int(s1.n_integer_mfma), split=int(s1.n_split_mfma), hamming=int(s1.n_hamming_mfma)))))
