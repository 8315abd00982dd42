"""Where the AC-RANSAC kernels spend their time: wave-0 cycles in the solve phase (one wave draws and solves 64 minimal samples) and in
the evaluation phase (all four waves: residuals, sort, NFA) of every pair, from the timing build (tools/build_bisect.sh):
    python tools/filter_phase_split.py [images] [features]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
api.use_library(os.path.join(os.path.dirname(os.path.abspath(api.__file__)), "libr3dm_bisect_timing.so"))
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
c = api.Context(0)
if len(sys.argv) > 3:                     # python tools/filter_phase_split.py 2 <rows> <matching fraction>: ONE pair with a long match list
    frac = float(sys.argv[3])
    rng = np.random.default_rng(n_feat)
    n = n_feat
    A = np.rint(rng.uniform(0, 255, (n, 16))).astype(np.float32); B = np.rint(rng.uniform(0, 255, (n, 16))).astype(np.float32)
    nm = int(frac * n); src = rng.permutation(n)[:nm]
    B[:nm] = np.clip(A[src] + np.rint(rng.normal(0, 2, (nm, 16))), 0, 255)
    X = np.c_[rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(8, 14, n)]
    f = 4800.0
    xyA = np.c_[f * X[:, 0] / X[:, 2] + 2000, f * X[:, 1] / X[:, 2] + 1500]
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Y = X @ R.T + np.array([0.8, 0.05, 0.1])
    xyB = np.c_[rng.uniform(0, 4000, n), rng.uniform(0, 3000, n)]
    xyB[:nm] = (np.c_[f * Y[:, 0] / Y[:, 2] + 2000, f * Y[:, 1] / Y[:, 2] + 1500] + rng.normal(0, 0.4, (n, 2)))[src]
    K = np.array([[f, 0, 2000], [0, f, 1500], [0, 0, 1.0]])
    c.set_image(0, A, xyA.astype(np.float32), 4000, 3000); c.set_image(1, B, xyB.astype(np.float32), 4000, 3000)
    c.set_intrinsics(0, K); c.set_intrinsics(1, K)
    g = c.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
    print("one pair,", g.num_matches, "putatives")
else:
    sc = synth.make_scene(n_img, n_feat, "sift", seed=2002)
    K = synth.intrinsics()
    for i in range(sc.n_images):
        c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
    g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
for name, fn in (("F", c.filter_F), ("H", c.filter_H), ("E", c.filter_E)):
    fn(g); fn(g)
    st = c.stats(); rpt = c.filter_report()
    a = np.array([r[0] for r in rpt]); b = np.array([r[1] for r in rpt]); it = np.array([r[2] for r in rpt]); mo = np.array([r[3] for r in rpt])
    solve, ev = np.floor(a), np.floor(b)
    res, srt = (a - solve) * 1e12, (b - ev) * 1e12
    print(f"   of the evaluation: residual passes {res.sum() / ev.sum():.2f}, sort {srt.sum() / ev.sum():.2f}; inliers {np.mean([r[4] for r in rpt]):.0f}")
    print(f"{name}: {len(rpt)} pairs, kernel {st.ms_filter_kernels:.2f} ms; per pair: solve {solve.mean():.3e} cycles, evaluate {ev.mean():.3e} cycles "
          f"(solve share {solve.sum() / (solve.sum() + ev.sum()):.2f}); iterations {it.mean():.0f}, models {mo.mean():.0f}", flush=True)
