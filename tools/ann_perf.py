"""Config C5 probe: graph-based approximate matching vs the exhaustive matcher on the same views
(throughput, distance evaluations, recall of the exhaustive putative matches).  Not the bench contract."""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
import os as _os
if any(k.startswith("R3DM_") for k in _os.environ):
    api.use_developer_library()      # R3DM_* knobs / traces exist only in the developer build (build.sh dev); otherwise measure the product

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=16)
ap.add_argument("--feat", type=int, default=16384)
ap.add_argument("--kind", default="sift")
ap.add_argument("--presets", default="fast,medium,precise,default")
ap.add_argument("--hnsw", default="fast,medium,precise", help="HNSW presets to time as well (matchingAlgorithm 6 / 7 / 8); '' = none")
ap.add_argument("--mrpt", type=int, default=1, help="time the MRPT arm (matchingAlgorithm 5) as well")
ap.add_argument("--S", type=int, default=0, help="override search_S (0 = preset)")
ap.add_argument("--K", type=int, default=0, help="override index_K (0 = preset)")
a = ap.parse_args()

t = time.time(); sc = synth.make_scene(a.images, a.feat, a.kind, seed=2005); print("gen %.1fs" % (time.time() - t), flush=True)
c = api.Context(0)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
pairs = sc.exhaustive_pairs()
c.match_pairs(pairs[:4], 0.6, True)
t = time.time(); gb = c.match_pairs(pairs, 0.6, True); tb = time.time() - t
sb = c.stats()
bd = gb.as_dict()
nb = gb.num_matches
print(json.dumps(dict(method="exhaustive", pairs=len(pairs), s=tb, pairs_per_s=len(pairs) / tb, ms_kernel=sb.ms_match_kernels,
                      matches=nb)), flush=True)
# the same exhaustive matcher on the opt-in integer fast path (bf16-exact MFMA tiles, identical matches)
c.set_integer_mfma(True)
c.match_pairs(pairs[:4], 0.6, True)
t = time.time(); gi = c.match_pairs(pairs, 0.6, True); ti = time.time() - t
si = c.stats()
c.set_integer_mfma(False)
print(json.dumps(dict(method="exhaustive-integer-mfma", pairs=len(pairs), s=ti, pairs_per_s=len(pairs) / ti, ms_kernel=si.ms_match_kernels,
                      ms_kernel_per_pair=si.ms_match_kernels / len(pairs), launches_on_bf16_tiles=si.n_integer_mfma,
                      identical=bool(np.array_equal(gi.matches, gb.matches) and np.array_equal(gi.offsets, gb.offsets)))), flush=True)
for name in a.presets.split(","):
    kp = api.KGraphParams.preset(name)
    if a.S: kp.search_S = a.S
    if a.K: kp.index_K = a.K
    for i in range(sc.n_images):          # drop cached indices so the build is timed
        c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    t = time.time(); g = c.match_pairs_kgraph(pairs, 0.6, kp); t1 = time.time() - t
    s1 = c.stats(); first = (g.pairs.tobytes(), g.matches.tobytes())
    t = time.time(); g = c.match_pairs_kgraph(pairs, 0.6, kp); t2 = time.time() - t
    assert first == (g.pairs.tobytes(), g.matches.tobytes()), "graph matcher is not deterministic"
    s2 = c.stats()
    d = g.as_dict()
    hit = sum(len(set(map(tuple, d[k].tolist())) & set(map(tuple, bd[k].tolist()))) for k in d if k in bd)
    print(json.dumps(dict(method="kgraph-" + name, K=kp.index_K, P=kp.search_P, S=kp.search_S, s_first=t1, s_cached=t2,
                          pairs_per_s=len(pairs) / t2, ms_build=s1.ms_ann_build, ms_build_per_view=s1.ms_ann_build / max(s1.n_ann_built, 1),
                          ms_search=s2.ms_ann_search, ms_search_per_pair=s2.ms_ann_search / len(pairs),
                          evals_per_query=s2.n_ann_dist / max(s2.n_queries, 1), matches=g.num_matches,
                          match_recall=hit / max(nb, 1), match_precision=hit / max(g.num_matches, 1),
                          speedup_vs_exhaustive=sb.ms_match_kernels / s2.ms_ann_search)), flush=True)
for name in [x for x in a.hnsw.split(",") if x]:
    hp = api.HnswParams.preset(name)
    c.drop_indices()
    t = time.time(); g = c.match_pairs_hnsw(pairs, 0.6, hp); t1 = time.time() - t
    s1 = c.stats(); first = (g.pairs.tobytes(), g.matches.tobytes())
    t = time.time(); g = c.match_pairs_hnsw(pairs, 0.6, hp); t2 = time.time() - t
    assert first == (g.pairs.tobytes(), g.matches.tobytes()), "HNSW matcher is not deterministic"
    s2 = c.stats()
    d = g.as_dict()
    hit = sum(len(set(map(tuple, d[k].tolist())) & set(map(tuple, bd[k].tolist()))) for k in d if k in bd)
    print(json.dumps(dict(method="hnsw-" + name, M=hp.M, ef=hp.ef, s_first=t1, s_cached=t2, pairs_per_s=len(pairs) / t2,
                          ms_build=s1.ms_ann_build, ms_build_per_view=s1.ms_ann_build / max(s1.n_ann_built, 1),
                          ms_search=s2.ms_ann_search, ms_search_per_pair=s2.ms_ann_search / len(pairs),
                          evals_per_query=s2.n_ann_dist / max(s2.n_queries, 1), retries=s2.n_hnsw_retries, matches=g.num_matches,
                          match_recall=hit / max(nb, 1), match_precision=hit / max(g.num_matches, 1),
                          speedup_vs_exhaustive=sb.ms_match_kernels / s2.ms_ann_search)), flush=True)
if a.mrpt:
    mp = api.MrptParams.preset()
    c.drop_indices()
    t = time.time(); g = c.match_pairs_mrpt(pairs, 0.6, mp); t1 = time.time() - t
    s1 = c.stats(); first = (g.pairs.tobytes(), g.matches.tobytes())
    t = time.time(); g = c.match_pairs_mrpt(pairs, 0.6, mp); t2 = time.time() - t
    assert first == (g.pairs.tobytes(), g.matches.tobytes()), "MRPT matcher is not deterministic"
    s2 = c.stats()
    d = g.as_dict()
    hit = sum(len(set(map(tuple, d[k].tolist())) & set(map(tuple, bd[k].tolist()))) for k in d if k in bd)
    print(json.dumps(dict(method="mrpt", n_trees=mp.n_trees, depth=mp.depth, votes=mp.votes, s_first=t1, s_cached=t2, pairs_per_s=len(pairs) / t2,
                          ms_build=s1.ms_ann_build, ms_build_per_view=s1.ms_ann_build / max(s1.n_ann_built, 1),
                          ms_search=s2.ms_ann_search, ms_search_per_pair=s2.ms_ann_search / len(pairs),
                          evals_per_query=s2.n_ann_dist / max(s2.n_queries, 1), matches=g.num_matches,
                          match_recall=hit / max(nb, 1), match_precision=hit / max(g.num_matches, 1),
                          speedup_vs_exhaustive=sb.ms_match_kernels / s2.ms_ann_search)), flush=True)
