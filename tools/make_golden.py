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

# ann_hnsw_ref.npz : the REFERENCE's approximate matcher as a recall baseline -- hnswlib::HierarchicalNSW from the reference's
# vendored copy, driven as ArrayMatcher_hnsw does, with the three presets of hnsw_match (src/R3DComputeMatches.cpp:533-565).
# Data only: descriptors, the exact 2-NN (reference BruteforceSearch) and the approximate 2-NN of every preset.
import ctypes as C
from regard3d_amd import synth
sc = synth.make_scene(2, 1000, "sift", seed=77)
A = sc.descs[0].astype(np.float32); B = sc.descs[1].astype(np.float32)
exact_idx, exact_dist = O.ref_knn(A, B, 2)
ann = {}
for name, (M, efc, ef) in {"fast": (5, 112, 5), "medium": (15, 112, 10), "precise": (19, 100, 15)}.items():
    idx = np.zeros((len(B), 2), np.int32); dist = np.zeros((len(B), 2), np.float32)
    assert O.ref_lib().ref_hnsw_ann_l2(A.ctypes.data_as(C.c_void_p), len(A), B.ctypes.data_as(C.c_void_p), len(B), 128, M, efc, ef, 2,
                                       idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p)) == 0
    ann[name] = idx
    print("hnsw", name, "recall@1", float((idx[:, 0] == exact_idx[:, 0]).mean()))
np.savez_compressed(os.path.join(out, "ann_hnsw_ref.npz"), dataset=A.astype(np.uint8), query=B.astype(np.uint8),
                    exact_idx=exact_idx.astype(np.int32), hnsw_fast=ann["fast"], hnsw_medium=ann["medium"], hnsw_precise=ann["precise"])

# regression goldens of the restatements that have NO reference-built counterpart (oracle outputs, frozen so that a change of
# the oracle's arithmetic is noticed): Fast-A-KAZE keypoints + MLDB on a small image, five-point solutions, a graph index
rng = np.random.default_rng(123)
yy, xx = np.mgrid[0:150, 0:200]
img = 0.5 + 0.1 * np.sin(xx / 11.0) * np.cos(yy / 7.0)
for _ in range(12):
    cx, cy, s, a = rng.uniform(30, 170), rng.uniform(30, 120), rng.uniform(2, 6), rng.uniform(0.2, 0.4) * rng.choice([-1, 1])
    img = img + a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
img8 = np.rint(np.clip(img, 0, 1) * 255).astype(np.uint8)
imgf = (img8.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)
kps, mldb, resp = O.akaze_detect_mldb(imgf, 0.001)
np.savez_compressed(os.path.join(out, "akaze_small.npz"), image_u8=img8, keypoints=kps, mldb=mldb, responses=resp)
print("akaze_small.npz", len(kps), "keypoints")
x1 = rng.uniform(-0.5, 0.5, (8, 5, 2)); x2 = x1 + rng.normal(0, 0.05, x1.shape) + np.array([0.1, 0.0])
sols = [O.five_point(a, b) for a, b in zip(x1, x2)]
np.savez_compressed(os.path.join(out, "five_point.npz"), x1=x1, x2=x2, n_solutions=np.array([len(s) for s in sols]),
                    solutions=np.concatenate(sols) if sum(len(s) for s in sols) else np.zeros((0, 3, 3)))
print("five_point.npz", [len(s) for s in sols])
D = np.rint(rng.uniform(0, 255, (300, 32))).astype(np.float32)
g = O.kgraph_build_exact(D, K=8, cap=64)
off, ids, dist = g.csr()
np.savez_compressed(os.path.join(out, "kgraph_index.npz"), data=D.astype(np.uint8), offsets=off, ids=ids, dist=dist)
print("kgraph_index.npz", len(ids), "edges")
