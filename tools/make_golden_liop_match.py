"""Generates tests/golden/liop_match_ref.npz (run in the authoring container, needs /root/reference).

Pins the REAL-valued matching path -- LIOP-144, the descriptor Regard3D actually matches -- with reference-built code end to end:

  two views x 8,192 synthetic 41x41 patches (55 % of the world textures shared, each observation with its own noise level, gain
  and gamma: LIOP is invariant to monotonic intensity changes, so shared textures give near-duplicate descriptors)
    -> the reference's own r3d_vl_liopdesc_process (src/thirdparty/liop/vl_liop.c:465-580, compiled where it lies into
       oracle/_ref/libref_liop.so)                                  -> unit-norm f32[144] descriptors
    -> the reference's own hnswlib::BruteforceSearch + L2Space (src/thirdparty/hnswlib/hnswlib/bruteforce.h:71-93, AVX L2Sqr,
       oracle/_ref/libref_hnsw.so)                                   -> 3-NN of every row of view 1 among the rows of view 0

Data only -- no reference source.  A LIOP descriptor is an integer histogram divided by its float norm
(vl_liop.c:565-575: desc[i] /= norm), so the fixture stores the histograms as u16 and the norms as f32 and the loader rebuilds
the floats with one IEEE division; this script asserts that the rebuilt rows equal the reference-built rows BIT FOR BIT before
it writes anything (9.4 MB of floats that deflate badly -> 2.4 MB of small integers).

hnswlib's AVX kernel sums the 144 squared differences in another order than OpenMVG's scalar loop, so this is
an index / verdict pin with a stated tolerance, not a bit pin: see tests/test_liop_match_ref.py.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.ndimage import gaussian_filter
from oracle import pyoracle as O

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
O.build()
assert O.ref_lib() is not None and O.ref_liop_lib() is not None, "oracle/_ref not built: needs /root/reference"

N, SHARED = 8192, 4506                                    # 55 % of the rows of each view observe a shared texture
rng = np.random.default_rng(144)
n_world = SHARED + 2 * (N - SHARED)
sig = rng.choice([1.0, 1.5, 2.2, 3.0], n_world)
world = np.stack([gaussian_filter(rng.random((41, 41)), s) for s in sig]).astype(np.float32)
world -= world.min(axis=(1, 2), keepdims=True); world /= world.max(axis=(1, 2), keepdims=True)


def observe(ids, noise):
    p = world[ids].astype(np.float64)
    gain = rng.uniform(0.6, 1.0, (len(ids), 1, 1)); gamma = rng.uniform(0.7, 1.4, (len(ids), 1, 1))
    # per-observation noise from almost none to enough to lose the match: the ratio test sees the whole range of d1 / d2
    sigma = noise * 10.0 ** rng.uniform(0.0, 1.6, (len(ids), 1, 1))
    p = gain * p ** gamma + rng.normal(0, 1.0, p.shape) * sigma
    return np.clip(p, 0, 1).astype(np.float32)


ids0 = np.concatenate([np.arange(SHARED), SHARED + np.arange(N - SHARED)])
ids1 = np.concatenate([np.arange(SHARED), SHARED + (N - SHARED) + np.arange(N - SHARED)])
perm0, perm1 = rng.permutation(N), rng.permutation(N)
P0, P1 = observe(ids0[perm0], 0.004), observe(ids1[perm1], 0.004)
D0, D1 = O.ref_liop(P0), O.ref_liop(P1)                    # the reference's own descriptor routine


def split(D):
    """rows -> (u16 histogram, f32 norm) with hist / norm == row bit for bit"""
    nz = D > 0
    unit = np.where(nz, D, np.inf).min(axis=1)                       # value of a bin holding the smallest non-zero weight
    # the smallest weight need not be 1: find the norm as the float that reproduces every bin
    hist = np.zeros(D.shape, np.uint16); norm = np.zeros(len(D), np.float32)
    for r in range(len(D)):
        if not nz[r].any():
            norm[r] = np.float32(1e-12); continue
        for w in range(1, 7):                                           # smallest non-zero weight is w
            nr = np.float32(w) / unit[r]
            for cand in (nr, np.nextafter(nr, np.float32(0)), np.nextafter(nr, np.float32(np.inf))):
                h = np.rint(D[r].astype(np.float64) * float(cand))
                if np.array_equal((h.astype(np.float32) / np.float32(cand)).astype(np.float32), D[r]):
                    hist[r] = h.astype(np.uint16); norm[r] = cand; break
            if norm[r] != 0: break
        assert norm[r] != 0, f"row {r}: no (histogram, norm) pair reproduces the reference descriptor"
    return hist, norm


h0, n0 = split(D0); h1, n1 = split(D1)
R0 = (h0.astype(np.float32) / n0[:, None]).astype(np.float32); R1 = (h1.astype(np.float32) / n1[:, None]).astype(np.float32)
assert np.array_equal(R0.view(np.uint32), D0.view(np.uint32)) and np.array_equal(R1.view(np.uint32), D1.view(np.uint32))

idx3, dist3 = O.ref_knn(D0, D1, 3)                         # the reference's own brute-force search
np.savez_compressed(os.path.join(out, "liop_match_ref.npz"), hist0=h0, norm0=n0, hist1=h1, norm1=n1,
                    ref_idx=idx3.astype(np.int32), ref_dist=dist3.astype(np.float32))
true_pair = (ids1[perm1][:, None] == ids0[perm0][idx3[:, 0]][:, None]).ravel() & (ids1[perm1] < SHARED)
verdict = dist3[:, 0] < np.float32(0.36) * dist3[:, 1]
print("liop_match_ref.npz:", os.path.getsize(os.path.join(out, "liop_match_ref.npz")), "bytes | matches under ratio 0.6:",
      int(verdict.sum()), "| of which true correspondences:", int((verdict & true_pair).sum()),
      "| shared rows whose nearest row is their texture:", int(true_pair.sum()), "of", SHARED)
