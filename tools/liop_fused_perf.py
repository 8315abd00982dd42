"""LIOP from keypoints on a resident image: the fused kernel (warp + blur inside the descriptor's wavefront, the product path of the
features stage) beside the two-kernel form (patches through HBM: r3dm_extract_liop with patches_out).
   python tools/liop_fused_perf.py [keypoints]
Keypoints = what the detector emits on the stage's photographs: the detector itself runs once on the image."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init()
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()
n_want = int(sys.argv[1]) if len(sys.argv) > 1 else 226000
img = synth.make_photo(3000, 4000, seed=100)
c = api.Context(0)
kps, _ = c.detect_akaze(img, 0.001)
reps = (n_want + len(kps) - 1) // len(kps)
K = np.tile(kps, (reps, 1))[:n_want].copy()
print(json.dumps(dict(detected=len(kps), used=len(K), median_size=float(np.median(K[:, 2])))), flush=True)
for rep in range(3):
    t = time.time(); d = c.extract_liop(img, K, 8.0); wall = time.time() - t
    ms = c.stats().ms_liop_kernel
    print(json.dumps(dict(path="fused", rep=rep, kernel_ms=round(ms, 3), patches_per_s=round(len(K) / (ms * 1e-3)), wall_ms=round(wall * 1e3, 1))), flush=True)
n2 = min(len(K), 60000)                              # (the patches come back to the host: 6.7 KB each)
for rep in range(2):
    t = time.time(); d2, p2 = c.extract_liop(img, K[:n2], 8.0, want_patches=True); wall = time.time() - t
    ms = c.stats().ms_liop_kernel
    print(json.dumps(dict(path="extract + describe", rep=rep, n=n2, kernel_ms=round(ms, 3), patches_per_s=round(n2 / (ms * 1e-3)), equal=bool(np.array_equal(d[:n2], d2)))), flush=True)
