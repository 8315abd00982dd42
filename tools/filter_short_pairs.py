"""What the one-workgroup AC-RANSAC kernels do on C2's putative graph (790 short pairs): kernel ms of F / E / H, models and
residuals evaluated (r3dm_filter_report), counted f64 flops of the residual passes over the kernel time.
  python tools/filter_short_pairs.py [--images 200] [--reps 3]   (R3DM_* in the environment: the developer build)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=200)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
descs, xys, _ = synth.make_scene_torch(a.images, 8192, seed=2002, device="cuda", kind="sift")
c = api.Context(0)
n = a.images
c.set_images(list(range(n)), [descs[i] for i in range(n)], [xys[i] for i in range(n)], synth.WIDTH, synth.HEIGHT)
for i in range(n):
    c.set_intrinsics(i, synth.intrinsics())
c.set_integer_mfma(True)
ii, jj = np.triu_indices(n, k=1)
g = c.match_pairs(np.stack([ii, jj], 1).astype(np.uint32), 0.6, True)
m = np.diff(np.asarray(g.offsets).astype(np.int64))
print(json.dumps({"putative_pairs": int(g.num_pairs), "putative_matches": int(g.num_matches), "m_mean": float(m.mean()), "m_max": int(m.max()), "m_min": int(m.min())}))
FLOPS = {"F": 36, "E": 21, "H": 19}           # per residual, counted on the source (bench_legs.py: stage_filter_roofline)
for kind in ("F", "E", "H"):
    best = None
    for _ in range(a.reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        gf = getattr(c, "filter_" + kind)(g, 4.0, 2048, seed=5489)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
        s = c.stats(); rep = c.filter_report()
        models = np.array([r[3] for r in rep], np.int64); iters = np.array([r[2] for r in rep], np.int64)
        resid = int((models * m[:len(models)]).sum())
        row = {"kind": kind, "kernel_ms": round(s.ms_filter_kernels, 3), "wall_ms": round(wall, 2), "kept_pairs": int(gf.num_pairs), "models": int(models.sum()), "iterations": int(iters.sum()),
               "models_per_pair_mean": float(models.mean()), "residuals": resid, "us_per_model_and_pair_slot": s.ms_filter_kernels * 1e3 / max(models.mean(), 1),
               "counted_TF": resid * FLOPS[kind] / (s.ms_filter_kernels * 1e-3) / 1e12}
        best = row if best is None or row["kernel_ms"] < best["kernel_ms"] else best
    print(json.dumps(best))
    if os.environ.get("PAIR_TABLE"):
        w = models * m[:len(models)]
        order = np.argsort(-w)[:12]
        print("   top pairs by models x m:", [(int(m[i]), int(iters[i]), int(models[i]), int(rep[i][4])) for i in order])
        full = iters >= 2040
        print("   pairs that ran the whole budget:", int(full.sum()), "their m: mean %.0f max %d" % (m[:len(models)][full].mean() if full.any() else 0, m[:len(models)][full].max() if full.any() else 0),
              "models x m share: %.3f" % (w[full].sum() / max(w.sum(), 1)))
