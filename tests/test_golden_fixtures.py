"""Committed fixtures (tests/golden/, generator tools/make_golden.py).

ann_hnsw_ref.npz is REFERENCE-BUILT: the reference's vendored hnswlib (BruteforceSearch for the exact 2-NN, HierarchicalNSW driven
like ArrayMatcher_hnsw with the presets of hnsw_match, src/R3DComputeMatches.cpp:533-565) compiled where it lies.  The other three
freeze outputs of restatements that have no reference-built counterpart, so that a silent change of their arithmetic is noticed.
"""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_exact_2nn_equals_the_reference_bruteforce_and_graph_matcher_beats_the_reference_hnsw(oracle):
    z = np.load(os.path.join(G, "ann_hnsw_ref.npz"))
    A = z["dataset"].astype(np.float32); B = z["query"].astype(np.float32); ex = z["exact_idx"]
    bi, _ = oracle.knn2(A, B)
    assert np.array_equal(bi, ex)                                       # restatement == hnswlib::BruteforceSearch of the reference
    for name in ("fast", "medium", "precise"):
        K, L, rc, P = oracle.KGRAPH_PRESETS[name]
        idx, _, comps = oracle.kgraph_build_exact(A, K=L, cap=64).knn2(B, P=P)
        ours = float((idx[:, 0] == ex[:, 0]).mean())
        ref = float((z["hnsw_" + name][:, 0] == ex[:, 0]).mean())      # the reference's own approximate matcher, same preset name
        assert ours >= ref - 0.01, (name, ours, ref)
        assert comps < 0.5 * len(A) * len(B)


@pytest.mark.gpu
def test_gpu_graph_matcher_against_the_reference_hnsw_fixture(ctx, oracle):
    from regard3d_amd import api
    z = np.load(os.path.join(G, "ann_hnsw_ref.npz"))
    A = z["dataset"].astype(np.float32); B = z["query"].astype(np.float32); ex = z["exact_idx"]
    idx, dist = ctx.knn2(A, B)
    assert np.array_equal(idx, ex)                                       # exhaustive kernel == the reference's BruteforceSearch
    for name in ("fast", "medium", "precise"):
        kp = api.KGraphParams.preset(name)
        gi, gd = ctx.kgraph_knn2(A, B, kp)
        ours = float((gi[:, 0] == ex[:, 0]).mean())
        ref = float((z["hnsw_" + name][:, 0] == ex[:, 0]).mean())
        assert ours >= ref - 0.01, (name, ours, ref)


def test_akaze_regression_golden(oracle):
    z = np.load(os.path.join(G, "akaze_small.npz"))
    img = (z["image_u8"].astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)
    kps, mldb, resp = oracle.akaze_detect_mldb(img, 0.001)
    assert np.array_equal(kps, z["keypoints"]) and np.array_equal(mldb, z["mldb"]) and np.array_equal(resp, z["responses"])


@pytest.mark.gpu
def test_gpu_akaze_regression_golden(ctx):
    z = np.load(os.path.join(G, "akaze_small.npz"))
    img = (z["image_u8"].astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)
    kps, mldb = ctx.detect_akaze_mldb(img, 0.001)
    assert np.array_equal(kps, z["keypoints"]) and np.array_equal(mldb, z["mldb"])


def test_five_point_regression_golden(oracle):
    z = np.load(os.path.join(G, "five_point.npz"))
    off = 0
    for a, b, n in zip(z["x1"], z["x2"], z["n_solutions"]):
        Es = oracle.five_point(a, b)
        assert len(Es) == n
        for E, R in zip(Es, z["solutions"][off:off + n]):
            assert np.allclose(E / np.linalg.norm(E), R / np.linalg.norm(R), rtol=0, atol=1e-9)
        off += n


def test_kgraph_index_regression_golden(oracle):
    z = np.load(os.path.join(G, "kgraph_index.npz"))
    off, ids, dist = oracle.kgraph_build_exact(z["data"].astype(np.float32), K=8, cap=64).csr()
    assert np.array_equal(off, z["offsets"]) and np.array_equal(ids, z["ids"]) and np.array_equal(dist, z["dist"])


# ---- knn2_c2_fullsize.npz: one image pair at BASELINE config C2's full size (8192 x 8192 x 128) with the 2-NN computed by the
# reference's own vendored hnswlib::BruteforceSearch (/root/reference/src/thirdparty/hnswlib/hnswlib/bruteforce.h:71-93),
# generator tools/make_golden_fullsize.py.  Distances bit-equal on every row; indices on the rows without exact distance ties.
def _fullsize():
    z = np.load(os.path.join(G, "knn2_c2_fullsize.npz"))
    return z["dataset"], z["query"], z["ref_idx"], z["ref_dist"], z["tie_free"]


def test_oracle_equals_the_reference_bruteforce_at_full_size(oracle):
    A, B, ridx, rdist, ok = _fullsize()
    assert A.shape == (8192, 128) and B.shape == (8192, 128) and ok.sum() > 0.9 * len(ok)
    idx, dist = oracle.knn2(A.astype(np.float32), B.astype(np.float32))
    assert np.array_equal(dist, rdist)
    assert np.array_equal(idx[ok], ridx[ok])


@pytest.mark.gpu
@pytest.mark.parametrize("as_u8", [False, True])
def test_gpu_knn2_equals_the_reference_bruteforce_at_full_size(ctx, as_u8):
    A, B, ridx, rdist, ok = _fullsize()
    a, b = (A, B) if as_u8 else (A.astype(np.float32), B.astype(np.float32))
    idx, dist = ctx.knn2(a, b)                       # f32 MFMA tiles (the headline path)
    assert ctx.stats().n_integer_mfma == 0
    assert np.array_equal(dist, rdist) and np.array_equal(idx[ok], ridx[ok])
    ctx.set_integer_mfma(True)                       # opt-in bf16-exact tiles
    try:
        idx2, dist2 = ctx.knn2(a, b)
        assert ctx.stats().n_integer_mfma == 1
    finally:
        ctx.set_integer_mfma(False)
    assert np.array_equal(dist2, rdist) and np.array_equal(idx2[ok], ridx[ok])
    assert np.array_equal(idx2, idx)


@pytest.mark.gpu
def test_gpu_match_pair_equals_ratio_test_on_the_reference_bruteforce_at_full_size(ctx):
    """the collection path (r3dm_match_pairs) on the same pair: matches == NNdistanceRatio applied to the reference-built 2-NN"""
    A, B, ridx, rdist, ok = _fullsize()
    ctx.clear_images()
    ctx.set_image(0, A.astype(np.float32)); ctx.set_image(1, B.astype(np.float32))
    g = ctx.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
    keep = rdist[:, 0] < np.float32(0.36) * rdist[:, 1]
    exp = np.stack([ridx[keep, 0].astype(np.uint32), np.nonzero(keep)[0].astype(np.uint32)], 1)
    exp = exp[np.lexsort((exp[:, 1], exp[:, 0]))]
    assert np.array_equal(g.matches, exp)
    ctx.clear_images()
