"""The cooperative AC-RANSAC kernel (kernels_filter_coop.hip) against the one-workgroup-per-pair kernel on the same putative graph,
developer build:  python tools/filter_coop_check.py [images] [features] [long-pair rows]

Every setting of (threshold, slices per pair, workers) must give byte-identical filtered graphs, models and per-pair reports
(threshold, NFA, iterations, models, inliers) for F, E and H.  Prints one JSON line per setting and "identical" at the end;
tests/test_gpu_filter_coop.py runs it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
api.use_developer_library()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
n_long = int(sys.argv[3]) if len(sys.argv) > 3 else 9000

sc = synth.make_scene(n_img, n_feat, "sift", seed=411)
K = synth.intrinsics()
c = api.Context(0)
c.set_integer_mfma(True)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
pairs = [tuple(p) for p in sc.exhaustive_pairs().tolist()]
if n_long:
    # two more views that share most of their features: one pair with a long match list
    rng = np.random.default_rng(n_long)
    n = n_long
    A = np.rint(rng.uniform(0, 255, (n, 128))).astype(np.float32)
    B = np.rint(rng.uniform(0, 255, (n, 128))).astype(np.float32)
    nm = int(0.9 * n); src = rng.permutation(n)[:nm]
    B[:nm] = np.clip(A[src] + np.rint(rng.normal(0, 2, (nm, 128))), 0, 255)
    X = np.c_[rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(8, 14, n)]
    f = 4800.0
    xyA = np.c_[f * X[:, 0] / X[:, 2] + 2000, f * X[:, 1] / X[:, 2] + 1500]
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Y = X @ R.T + np.array([0.8, 0.05, 0.1])
    xyB = np.c_[rng.uniform(0, 4000, n), rng.uniform(0, 3000, n)]
    xyB[:nm] = (np.c_[f * Y[:, 0] / Y[:, 2] + 2000, f * Y[:, 1] / Y[:, 2] + 1500] + rng.normal(0, 0.4, (n, 2)))[src]
    a, b = sc.n_images, sc.n_images + 1
    c.set_image(a, A, xyA.astype(np.float32), 4000, 3000); c.set_image(b, B, xyB.astype(np.float32), 4000, 3000)
    c.set_intrinsics(a, K); c.set_intrinsics(b, K)
    pairs.append((a, b))
g = c.match_pairs(np.array(pairs, np.uint32), 0.6, True)
cnt = np.diff(g.offsets.astype(np.int64))
print(json.dumps(dict(pairs=int(g.num_pairs), matches=int(g.num_matches), longest=int(cnt.max()), median=int(np.median(cnt)))), flush=True)


def run(env):
    for k in ("R3DM_FILTER_COOP_MIN", "R3DM_FILTER_COOP_G", "R3DM_FILTER_COOP_WORKERS", "R3DM_FILTER_CHECK"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    out = {}; ms = {}
    for name, fn, kw in (("F", c.filter_F, "want_F"), ("H", c.filter_H, "want_H"), ("E", c.filter_E, "want_E")):
        t = time.time()
        gf, M = fn(g, **{kw: True})
        ms[name] = (round((time.time() - t) * 1e3, 2), round(c.stats().ms_filter_kernels, 2))
        out[name] = (np.array(gf.pairs), np.array(gf.offsets), np.array(gf.matches), np.array(M), np.array(c.filter_report(), np.float64))
    return out, ms


settings = [dict(R3DM_FILTER_COOP_MIN=0),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=1),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=2),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=4),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=8),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=3, R3DM_FILTER_COOP_WORKERS=2),
            dict(R3DM_FILTER_COOP_MIN=300, R3DM_FILTER_COOP_G=5, R3DM_FILTER_CHECK=1),
            dict()]
ref = None
bad = []
for env in settings:
    out, ms = run(env)
    print(json.dumps(dict(env=env, ms_wall_kernel=ms, kept={k: int(len(v[0])) for k, v in out.items()})), flush=True)
    if ref is None:
        ref = out
        continue
    for name in ("F", "H", "E"):
        for what, x, y in zip(("pairs", "offsets", "matches", "models", "report"), out[name], ref[name]):
            if x.shape != y.shape or not np.array_equal(x, y):
                bad.append((str(env), name, what))
print("identical" if not bad else "DIFFERENT: " + json.dumps(bad[:20]))
sys.exit(0 if not bad else 1)
