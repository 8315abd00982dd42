"""Generates the committed golden fixtures under tests/golden/ (run in the authoring container).

knn2_sift_int.npz : integer-valued SIFT-like descriptors + the 2-NN indices/distances produced by the
    REFERENCE's vendored hnswlib::BruteforceSearch (oracle/_ref/libref_hnsw.so, compiled from
    /root/reference/src/thirdparty/hnswlib by oracle/Makefile).  Data only -- no reference source.
    The fixture is drawn until it has no exact distance ties among the two nearest and the third
    neighbour (hnswlib's `dist <= lastdist` rule and OpenMVG's unstable partial sort both leave tie
    order unspecified).
rng_stream.npz : known answers of the counter-based AC-RANSAC sample stream (pins the integer recipe).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(out, exist_ok=True)
O.build()
assert O.ref_lib() is not None, "oracle/_ref not built: needs /root/reference"

seed = 11
while True:
    rng = np.random.default_rng(seed)
    base = rng.gamma(0.5, 1.0, (260, 128)); base = base / np.linalg.norm(base, axis=1, keepdims=True) * 512
    ds = np.rint(np.clip(base[:200] + rng.normal(0, 6, (200, 128)), 0, 255)).astype(np.float32)
    q = np.rint(np.clip(np.concatenate([base[:60], base[200:]]) + rng.normal(0, 6, (120, 128)), 0, 255)).astype(np.float32)
    idx3, dist3 = O.ref_knn(ds, q, 3)
    if (dist3[:, 0] != dist3[:, 1]).all() and (dist3[:, 1] != dist3[:, 2]).all():
        break
    seed += 1
np.savez_compressed(os.path.join(out, "knn2_sift_int.npz"), dataset=ds.astype(np.uint8), query=q.astype(np.uint8),
                    ref_idx=idx3[:, :2].astype(np.int32), ref_dist=dist3[:, :2].astype(np.float32), seed=seed)
print("knn2_sift_int.npz seed", seed, "matches under ratio 0.6:", int((dist3[:, 0] < 0.36 * dist3[:, 1]).sum()))

vals = np.array([O.lib().orc_rng_u64(5489, I, J, it, at) for (I, J, it, at) in
                 [(0, 1, 0, 0), (0, 1, 0, 1), (0, 1, 1, 0), (3, 5, 2, 0), (199, 7, 2047, 6), (4294967295, 0, 0, 0)]], np.uint64)
smp = np.stack([O.sample7(5489, 3, 5, it, np.arange(100, dtype=np.uint32)) for it in range(4)])
np.savez_compressed(os.path.join(out, "rng_stream.npz"), u64=vals, sample7_pool100=smp)
print("rng", vals, smp.tolist())

# liop_patches.npz : 41x41 patches (smooth, textured, quantised with many equal intensities, half-flat, constant)
# + the descriptors produced by the REFERENCE's own vl_liop.c (oracle/_ref/libref_liop.so).  Data only.
from scipy.ndimage import gaussian_filter
rng = np.random.default_rng(7)
P = np.stack([gaussian_filter(rng.random((41, 41)), s).astype(np.float32) for s in (0.8, 1.2, 2.0, 3.0) for _ in range(6)])
P[3] = np.round(P[3] * 16) / 16            # heavy ties
P[9] = np.round(P[9] * 4) / 4
P[14, :, :18] = 0.25                        # half flat
P[20] = 0.5                                 # constant -> all-zero descriptor
P[21] = np.float32(np.arange(41)[None, :] / 40.0) * np.ones((41, 1), np.float32)   # linear ramp
ref = O.ref_liop(P)
np.savez_compressed(os.path.join(out, "liop_patches.npz"), patches=P, ref_desc=ref)
print("liop_patches.npz", P.shape, "norms", np.round(np.linalg.norm(ref, axis=1), 6)[[0, 3, 14, 20, 21]])
