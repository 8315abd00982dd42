"""Timing of the AC-RANSAC filters on long pairs, developer build:
    python tools/filter_coop_perf.py one <rows> <matching fraction> [settings...]      ONE pair with a long match list
    python tools/filter_coop_perf.py coll <images> <features> [settings...]            a collection of few, long pairs (F, E, H and FEH)
a setting is "MIN:G:WORKERS" (R3DM_FILTER_COOP_MIN / _G / _WORKERS; empty field = default), e.g. "0::" = the one-workgroup kernel.
R3DM_COOP_PROF=1 in the environment prints the per-pair phase table of the cooperative kernel to stderr."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
if not os.environ.get("USE_PRODUCT"):
    api.use_developer_library()
mode = sys.argv[1]
c = api.Context(0)
c.set_integer_mfma(True)
K = synth.intrinsics()
if mode == "one":
    n, frac = int(sys.argv[2]), float(sys.argv[3])
    rng = np.random.default_rng(n)
    A = np.rint(rng.uniform(0, 255, (n, 16))).astype(np.float32); B = np.rint(rng.uniform(0, 255, (n, 16))).astype(np.float32)
    nm = int(frac * n); src = rng.permutation(n)[:nm]
    B[:nm] = np.clip(A[src] + np.rint(rng.normal(0, 2, (nm, 16))), 0, 255)
    X = np.c_[rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(8, 14, n)]
    if os.environ.get("PLANAR"):          # a planar scene: the homography filter finds a model with (nearly) every match as inlier
        X[:, 2] = 10.0 + 0.2 * X[:, 0]
    f = 4800.0
    xyA = np.c_[f * X[:, 0] / X[:, 2] + 2000, f * X[:, 1] / X[:, 2] + 1500]
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Y = X @ R.T + np.array([0.8, 0.05, 0.1])
    xyB = np.c_[rng.uniform(0, 4000, n), rng.uniform(0, 3000, n)]
    xyB[:nm] = (np.c_[f * Y[:, 0] / Y[:, 2] + 2000, f * Y[:, 1] / Y[:, 2] + 1500] + rng.normal(0, 0.4, (n, 2)))[src]
    Kk = np.array([[f, 0, 2000], [0, f, 1500], [0, 0, 1.0]])
    c.set_image(0, A, xyA.astype(np.float32), 4000, 3000); c.set_image(1, B, xyB.astype(np.float32), 4000, 3000)
    c.set_intrinsics(0, Kk); c.set_intrinsics(1, Kk)
    g = c.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
else:
    n_img, n_feat = int(sys.argv[2]), int(sys.argv[3])
    sc = synth.make_scene(n_img, n_feat, "sift", seed=2002)
    for i in range(sc.n_images):
        c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
    g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
cnt = np.diff(g.offsets.astype(np.int64))
print(json.dumps(dict(pairs=int(g.num_pairs), matches=int(g.num_matches), longest=int(cnt.max()), median=int(np.median(cnt)))), flush=True)
settings = sys.argv[4:] or ["0::", "::"]
for st in settings:
    mn, G, W = (st.split(":") + ["", "", ""])[:3]
    for k, v in (("R3DM_FILTER_COOP_MIN", mn), ("R3DM_FILTER_COOP_G", G), ("R3DM_FILTER_COOP_WORKERS", W)):
        os.environ.pop(k, None)
        if v != "":
            os.environ[k] = v
    out = {}
    for rep in range(2):
        for name, fn in (("F", c.filter_F), ("E", c.filter_E), ("H", c.filter_H)):
            if name not in os.environ.get("KINDS", "FEH"):
                continue
            fn(g); out[name] = round(c.stats().ms_filter_kernels, 2)
        if mode != "one" and os.environ.get("KINDS", "FEH") == "FEH":
            for which in os.environ.get("FEH_WHICH", "FEH").split(","):
                t = time.time()
                try:
                    _, msk, msw = c.filter_FEH(g, which); out[which + "_wall"] = round((time.time() - t) * 1e3, 2); out[which + "_kernels"] = [round(float(x), 2) for x in msk]
                except Exception as e:
                    out[which + "_error"] = str(e)[-330:]; out[which + "_wall"] = round((time.time() - t) * 1e3, 2)
    print(json.dumps(dict(setting=st, ms=out)), flush=True)
