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
sc = synth.make_scene(n_img, n_feat, "sift", seed=2002)
c = api.Context(0)
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
