"""The match phase of the stage on the stage's own descriptors: runs bench's photo set through the stage once, reads the .feat/.desc back and
times r3dm_match_pairs (split-f16) on them with the library's statistics: main kernel, post-processing, exact-scan queries."""
import os, sys, time, json, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
W, H = 4000, 3000
dev = torch.device("cuda", 0)
imgs, K = synth.make_photo_set(N, H, W, seed=7007, device=dev)
views = [dict(id=k, width=W, height=H, basename=f"img{k:04d}", gray=imgs[k], focal_px=K[0, 0], ppx=K[0, 2], ppy=K[1, 2]) for k in range(N)]
d = tempfile.mkdtemp(prefix="r3dm_sm_")
try:
    st = api.Stage([0]); rep = st.run(d, views, 0.001, 0.6, 9, True, False, False); st.close()
    print(json.dumps(dict(stage_ms_match=rep.ms_match, stage_ms_match_kernels=rep.ms_match_kernels)))
    descs, xys = [], []
    for k in range(N):
        raw = np.fromfile(os.path.join(d, f"img{k:04d}.desc"), np.uint8)
        descs.append(np.frombuffer(raw[8:].tobytes(), np.float32).reshape(-1, 144).copy())
        xys.append(np.loadtxt(os.path.join(d, f"img{k:04d}.feat"), dtype=np.float32).reshape(-1, 4)[:, :2].copy())
finally:
    shutil.rmtree(d, ignore_errors=True)
c = api.Context(0); c.set_split_mfma(True)
for k in range(N):
    c.set_image(k, descs[k], xys[k], W, H)
pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], np.uint32)
for rep in range(3):
    t = time.time(); g = c.match_pairs(pairs, 0.6, True); dt = time.time() - t
    s = c.stats()
    print(json.dumps(dict(rep=rep, wall_ms=dt * 1e3, ms_match_kernels=s.ms_match_kernels, ms_wall_match_post=s.ms_wall_match_post, launches=s.n_match_launches,
                          exact_scan_queries=s.n_exact_fallback, queries=s.n_queries, matches=g.num_matches, pairs=len(g.pairs))), flush=True)
