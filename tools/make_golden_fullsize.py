"""Generates tests/golden/knn2_c2_fullsize.npz (run in the authoring container, needs /root/reference).

One image pair at BASELINE config C2's full size -- two views of 8192 x 128 integer-valued SIFT-like descriptors (u8) that
share 45 % of their world points -- and the 3-NN of every query row as computed by the REFERENCE's own vendored
hnswlib::BruteforceSearch + L2Space (oracle/_ref/libref_hnsw.so, compiled from /root/reference/src/thirdparty/hnswlib where it
lies by oracle/Makefile; algorithm: src/thirdparty/hnswlib/hnswlib/bruteforce.h:71-93).  Data only -- no reference source.

ref_idx / ref_dist : [8192, 2] the reference-built 2-NN (ascending distance)
tie_free           : [8192] bool, d1 != d2 and d2 != d3 -- hnswlib's `dist <= lastdist` rule and OpenMVG's unstable partial
                     sort both leave the order of EQUAL distances unspecified, so index equality is asserted on these rows
                     only; the distances must be bit-equal on every row.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from regard3d_amd import synth

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
O.build()
assert O.ref_lib() is not None, "oracle/_ref not built: needs /root/reference"
sc = synth.make_scene(2, 8192, "sift", seed=2002, dtype="u8")
A = sc.descs[0]; B = sc.descs[1]
idx3, dist3 = O.ref_knn(A.astype(np.float32), B.astype(np.float32), 3)
tie_free = (dist3[:, 0] != dist3[:, 1]) & (dist3[:, 1] != dist3[:, 2])
np.savez_compressed(os.path.join(out, "knn2_c2_fullsize.npz"), dataset=A, query=B, ref_idx=idx3[:, :2].astype(np.int32),
                    ref_dist=dist3[:, :2].astype(np.float32), tie_free=tie_free)
print("knn2_c2_fullsize.npz: tie-free rows", int(tie_free.sum()), "of", len(tie_free),
      "| matches under ratio 0.6:", int((dist3[:, 0] < np.float32(0.36) * dist3[:, 1]).sum()),
      "| bytes", os.path.getsize(os.path.join(out, "knn2_c2_fullsize.npz")))
