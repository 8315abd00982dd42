"""tools/devtools/stress_threads.py -- the library's host-threaded paths in a tight loop inside ONE process (product library):
  * r3dm_set_images from pageable numpy (helper threads + the page-locked ring), collections of changing size, replaced views;
  * r3dm_filter_FEH (a collect thread per kind) on graphs of short and long pairs;
  * the whole stage from pixels (features workers, deferred feature files, background writers, the PairWiseMatches maps).
Every round's graphs must equal the first round's.  Written after one unexplained abort of a `pytest -m gpu` run (docs/DESIGN_HISTORY.md R6.8).
  python tools/devtools/stress_threads.py [rounds = 40]"""
import hashlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from regard3d_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(11)


def sha(g):
    return hashlib.sha256(b"".join(np.ascontiguousarray(getattr(g, f)).tobytes() for f in ("pairs", "offsets", "matches"))).hexdigest()[:16]


sc = synth.make_scene(12, 5000, "sift", seed=77)
K = synth.intrinsics()
pairs = sc.exhaustive_pairs()
c = api.Context(0)
ref = None
t0 = time.time()
for r in range(rounds):
    c.clear_images()
    order = rng.permutation(sc.n_images)
    cut = [int(rng.integers(3000, 5001)) for _ in range(sc.n_images)] if r % 3 else [5000] * sc.n_images
    ids = [int(i) for i in order]
    c.set_images(ids, [sc.descs[i][:cut[i]] for i in ids], [sc.xys[i][:cut[i]] for i in ids], 4000, 3000)
    if r % 3 == 0:
        # replace half of the views by themselves, one call each (the single-view path while the ring still holds the batch)
        for i in ids[::2]:
            c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
        for i in range(sc.n_images):
            c.set_intrinsics(i, K)
        g = c.match_pairs(pairs, 0.6, True)
        outs, _, _ = c.filter_FEH(g, "FEH")
        key = (sha(g),) + tuple(sha(outs[k]) for k in "FEH")
        if ref is None:
            ref = key
        assert key == ref, (r, key, ref)
    else:
        g = c.match_pairs(pairs[: 20], 0.6, True)
print(f"registration + match + FEH: {rounds} rounds in {time.time() - t0:.1f} s, graphs {ref}", flush=True)
c.close()

# the stage from pixels, small photographs so that a round is short
imgs, Kc = synth.make_photo_set(6, 900, 1200, seed=7007, device=torch.device("cuda", 0))
views = [dict(id=k, width=1200, height=900, basename=f"img{k:04d}", gray=imgs[k], focal_px=Kc[0, 0], ppx=Kc[0, 2], ppy=Kc[1, 2]) for k in range(6)]
st = api.Stage([0])
ref = None
t0 = time.time()
for r in range(max(4, rounds // 4)):
    d = tempfile.mkdtemp(prefix="r3dm_stress_")
    try:
        rep = st.run(d, views, 0.001, 0.6, 9, True, True, True)
        h = hashlib.sha256(b"".join(open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.startswith("matches."))).hexdigest()[:16]
        if ref is None:
            ref = h
        assert h == ref, (r, h, ref)
    finally:
        shutil.rmtree(d, ignore_errors=True)
st.close()
print(f"stage from pixels: {max(4, rounds // 4)} rounds in {time.time() - t0:.1f} s, match files {ref}", flush=True)
