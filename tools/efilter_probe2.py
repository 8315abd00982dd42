"""Determinism / parity probe of the three AC-RANSAC filters against the oracle, run once per library build:
    python tools/efilter_probe2.py product|dev [reps]
(scene of tests/test_cpp_host.py::test_stage_facade_writes_the_reference_files plus a larger one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
if len(sys.argv) > 1 and sys.argv[1] == "dev":
    api.use_developer_library()
elif len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    api.use_library(os.path.abspath(sys.argv[1]))
from oracle import pyoracle as O
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
O.build()
c = api.Context(0)
for (n_img, n_feat, kind, seed) in ((5, 900, "liop", 23), (8, 3000, "sift", 77)):
    sc = synth.make_scene(n_img, n_feat, kind, seed=seed)
    c.clear_images()
    K = synth.intrinsics()
    for i in range(sc.n_images):
        c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
    pairs = sc.exhaustive_pairs()
    g = c.match_pairs(pairs, 0.6, True)
    counts, matches = O.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    Ks = np.stack([K] * sc.n_images)
    exp = {"F": O.filter_F_collection(sc.xys, sc.widths, sc.heights, pairs, counts, matches, 4.0, 2048, 5489),
           "H": O.filter_H_collection(sc.xys, sc.widths, sc.heights, pairs, counts, matches, 4.0, 2048, 5489),
           "E": O.filter_E_collection(sc.xys, sc.widths, sc.heights, Ks, pairs, counts, matches, 4.0, 2048, 5489)}
    for name, fn in (("F", c.filter_F), ("H", c.filter_H), ("E", c.filter_E)):
        oc, om = exp[name]
        for rep in range(reps):
            d = fn(g).as_dict(); rpt = c.filter_report()
            off = 0; bad = []
            for p, (I, J) in enumerate(pairs):
                e = om[off:off + oc[p]]; off += oc[p]
                got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
                if set(map(tuple, got.tolist())) != set(map(tuple, e.tolist())):
                    bad.append(((int(I), int(J)), len(e), len(got)))
            print(sys.argv[1] if len(sys.argv) > 1 else "product", kind, name, "rep", rep, "mismatching pairs:", bad, flush=True)
