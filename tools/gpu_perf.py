"""Quick GPU performance probe (not the bench contract): match + filter on a synthetic scene."""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
import os as _os
if any(k.startswith("R3DM_") for k in _os.environ):
    api.use_developer_library()      # R3DM_* knobs / traces exist only in the developer build (build.sh dev); otherwise measure the product

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=24)
ap.add_argument("--feat", type=int, default=8192)
ap.add_argument("--kind", default="sift")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--check", type=int, default=0, help="verify this many pairs against the oracle")
ap.add_argument("--all-filters", action="store_true", help="also time the essential-matrix and homography filters")
ap.add_argument("--integer-mfma", action="store_true", help="opt into the bf16-exact integer fast path and compare with the f32 path")
a = ap.parse_args()

t = time.time(); sc = synth.make_scene(a.images, a.feat, a.kind, seed=2002); print("gen %.1fs" % (time.time() - t), flush=True)
c = api.Context(0)
print(c.device_info())
binary = a.kind == "akaze"
t = time.time()
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000, binary=binary)
    c.set_intrinsics(i, synth.intrinsics())
print("set_image %.2fs" % (time.time() - t), flush=True)
pairs = sc.exhaustive_pairs()
ratio, sq = (0.8, False) if binary else (0.6, True)
if a.integer_mfma:
    g0 = c.match_pairs(pairs, ratio, sq); s0 = c.stats()
    c.set_integer_mfma(True)
    g1 = c.match_pairs(pairs, ratio, sq); s1 = c.stats()
    same = (np.array_equal(g0.pairs, g1.pairs) and np.array_equal(g0.offsets, g1.offsets)
            and np.array_equal(g0.matches, g1.matches))
    print(json.dumps(dict(integer_mfma_identical=bool(same), f32_ms_kernel=s0.ms_match_kernels, int_ms_kernel=s1.ms_match_kernels,
                          int_launches=s1.n_integer_mfma, f32_fallback=s0.n_exact_fallback, int_fallback=s1.n_exact_fallback)), flush=True)
for rep in range(a.reps):
    t = time.time(); g = c.match_pairs(pairs, ratio, sq); tm = time.time() - t
    s = c.stats()
    t = time.time(); gf = c.filter_F(g); tf = time.time() - t
    s2 = c.stats()
    print(json.dumps(dict(rep=rep, pairs=len(pairs), t_match=tm, t_filter=tf, pairs_per_s=len(pairs) / (tm + tf),
                          ms_kernel=s.ms_match_kernels, tflops=s.algorithmic_flops / (s.ms_match_kernels * 1e-3) / 1e12,
                          fallback=s.n_exact_fallback, queries=s.n_queries, put_pairs=g.num_pairs, put_matches=g.num_matches,
                          f_pairs=gf.num_pairs, f_matches=gf.num_matches, ms_filter_kernel=s2.ms_filter_kernels)), flush=True)
    if a.all_filters:
        t = time.time(); ge = c.filter_E(g); te = time.time() - t; se = c.stats()
        t = time.time(); gh = c.filter_H(g); th = time.time() - t; sh = c.stats()
        print(json.dumps(dict(rep=rep, t_filter_E=te, ms_E_kernel=se.ms_filter_kernels, e_pairs=ge.num_pairs, e_matches=ge.num_matches,
                              t_filter_H=th, ms_H_kernel=sh.ms_filter_kernels, h_pairs=gh.num_pairs)), flush=True)
        t = time.time(); feh, msk, msw = c.filter_FEH(g, "FEH"); tf = time.time() - t
        print(json.dumps(dict(rep=rep, side_by_side="r3dm_filter_FEH", wall_ms=tf * 1e3, kernels_ms_F_E_H=msk.tolist(),
                              same_graphs=bool(np.array_equal(feh["E"].matches, ge.matches) and np.array_equal(feh["H"].matches, gh.matches)))), flush=True)
if a.check:
    from oracle import pyoracle as O
    sub = pairs[: a.check]
    t = time.time(); counts, matches = O.match_collection(sc.descs, sc.xys, sub, ratio, sq, binary=binary); print("oracle %.1fs" % (time.time() - t))
    d = g.as_dict(); off = 0; bad = 0
    for p, (I, J) in enumerate(sub):
        exp = matches[off:off + counts[p]]; off += counts[p]
        got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
        if not np.array_equal(got, exp): bad += 1
    print("checked", len(sub), "pairs, mismatching:", bad)
