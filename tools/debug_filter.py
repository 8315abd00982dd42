"""Parity hunt for the AC-RANSAC kernel: per-pair summary + per-model trace diff of one pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from regard3d_amd import api, synth
api.use_developer_library()      # the per-model trace (R3DM_TRACE_*) exists only in the developer build (build.sh dev)
from oracle import pyoracle as O

n_img, n_feat, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tI, tJ = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (None, None)
sc = synth.make_scene(n_img, n_feat, "sift", seed=seed)
c = api.Context(0)
for i in range(sc.n_images):
    c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
if tI is not None:
    os.environ["R3DM_TRACE_PAIR"] = f"{tI},{tJ}"; os.environ["R3DM_TRACE_FILE"] = "gpurun_out/trace_gpu.txt"
gf = c.filter_F(g)
rep = c.filter_report()
gp, go, gm = g.pairs, g.offsets, g.matches
for p, (I, J) in enumerate(gp):
    m = gm[int(go[p]):int(go[p + 1])]
    if len(m) <= 7: continue
    xI = sc.xys[I][m[:, 0]].astype(np.float64); xJ = sc.xys[J][m[:, 1]].astype(np.float64)
    inl, fr = O.acransac_F(xI, xJ, 4000, 3000, 4000, 3000, I=int(I), J=int(J))
    r = rep[p]
    same = (r[2] == fr.n_iter and r[3] == fr.n_models and r[4] == fr.n_inliers)
    print(f"pair ({I},{J}) m={len(m)} gpu: it={r[2]} models={r[3]} inl={r[4]} nfa={r[1]:.9f} thr={r[0]:.6f} | cpu: it={fr.n_iter} models={fr.n_models} inl={fr.n_inliers} nfa={fr.nfa:.9f} thr={fr.threshold:.6f} {'OK' if same else 'DIFF'}")
    if tI == I and tJ == J:
        import ctypes
        O.lib().orc_set_debug_iter(int(os.environ.get('R3DM_TRACE_ITER', '-1')))
        _, _, tr = O.acransac_F_traced(xI, xJ, 4000, 3000, 4000, 3000, I=int(I), J=int(J))
        O.lib().orc_debug_sample.restype = ctypes.POINTER(ctypes.c_double)
        dbg = O.lib().orc_debug_sample()
        print('CPU sample idx', [int(dbg[k]) for k in range(7)], 'nm', int(dbg[7]), 'pool_size', int(dbg[8]), 'iter', int(dbg[9]), 'pos', [int(dbg[10+k]) for k in range(7)])
        tg = np.loadtxt("gpurun_out/trace_gpu.txt").reshape(-1, 5)
        for line in open("gpurun_out/trace_gpu.txt"):
            if line.startswith("# sample"):
                v = [int(x) for x in line.split()[2:]]
                print("GPU sample idx", v[:7], "nm", v[7], "pool_size", v[8], "iter", v[9], "pos", v[10:17])
                it_dbg = v[9]
                # CPU: replay the oracle up to that iteration to get its pool -> easiest: sample from the CPU inlier pool of the traced run
        
        print("trace rows cpu", len(tr), "gpu", len(tg))
        n = min(len(tr), len(tg))
        for k in range(n):
            a, b = tr[k], tg[k]
            if not (a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[4] == b[4] and (a[3] == b[3] or abs(a[3] - b[3]) <= 1e-9 * max(1, abs(a[3])))):
                print("first diff at row", k)
                for q in range(max(0, k - 3), min(n, k + 4)):
                    print("  cpu", tr[q].tolist(), "\n  gpu", tg[q].tolist())
                break
        else:
            print("traces agree on the common prefix")
