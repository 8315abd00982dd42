"""A/B of the integer (bf16-tile) nominator variants (developer build: R3DM_L2_INT_VARIANT, latched per process) on one workload:
  R3DM_L2_INT_VARIANT=<v> python tools/int_ring_ab.py [n_images] [n_feat] [kind sift|akaze]
prints kernel ms per match call (HIP events on the library's stream), matches and a hash of the graph (equal across variants).
2 = per-wave loads (the product default), 5 = LDS-shared with a barrier per tile, 7 = barrier-free LDS ring, 9 / 59 / 79 = the same
without their epilogues (timing only).  kind akaze: the exact MFMA Hamming (r3dm_set_hamming_mfma; R3DM_HAMMING_RING=1 = the ring)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
kind = sys.argv[3] if len(sys.argv) > 3 else "sift"
descs, xys, _ = synth.make_scene_torch(n_img, n_feat, seed=2002, device="cuda", kind=kind)
c = api.Context(0)
binary = kind == "akaze"
c.set_images(list(range(n_img)), [descs[i] for i in range(n_img)], [xys[i] for i in range(n_img)], 4000, 3000, binary=binary)
if binary:
    c.set_hamming_mfma(True)
else:
    c.set_integer_mfma(True)
ii, jj = np.triu_indices(n_img, k=1); pairs = np.stack([ii, jj], 1).astype(np.uint32)
ms = []
for rep in range(5):
    g = c.match_pairs(pairs, 0.8 if binary else 0.6, not binary); s = c.stats()
    ms.append(round(s.ms_match_kernels, 3))
sha = hashlib.sha256(b"".join(np.ascontiguousarray(getattr(g, f)).tobytes() for f in ("pairs", "offsets", "matches"))).hexdigest()[:16]
print(json.dumps({"variant": os.environ.get("R3DM_L2_INT_VARIANT", os.environ.get("R3DM_HAMMING_RING", "default")), "pairs": len(pairs), "kernel_ms": ms,
                  "int_launches": int(s.n_integer_mfma), "hamming_mfma_launches": int(s.n_hamming_mfma), "matches": g.num_matches, "fallback": int(s.n_exact_fallback), "graph_sha16": sha}))
