"""Pins the CPU restatement of the matching half (oracle/matching.c) -- runs without a GPU.

The reference ships no tests for this path, so the oracle is pinned by
analytic known answers, by hnswlib::BruteforceSearch compiled from
the reference's vendored sources (oracle/_ref, authoring container only) and by the golden
fixture generated from it (tests/golden/knn2_sift_int.npz, tools/make_golden.py).
"""
import os
from fractions import Fraction

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_l2sq_hand_computed(oracle):
    a = np.array([1, 2, 3, 4, 5, 6, 7], np.float32)
    b = np.array([7, 5, 3, 1, 0, 6, 9], np.float32)
    # 36 + 9 + 0 + 9 + 25 + 0 + 4 (dim 7 exercises the 4-way body and the scalar tail)
    assert oracle.l2sq(a, b) == 83.0
    assert oracle.l2sq(a.astype(np.uint8), b.astype(np.uint8)) == 83.0
    assert oracle.l2sq(a[:4], b[:4]) == 54.0
    assert oracle.l2sq(a[:1], b[:1]) == 36.0


def test_l2sq_summation_order_is_the_references(oracle):
    # result += ((d0^2 + d1^2) + d2^2) + d3^2 in float: reproduce with numpy float32 scalars
    rng = np.random.default_rng(0)
    a = rng.normal(size=144).astype(np.float32); b = rng.normal(size=144).astype(np.float32)
    acc = np.float32(0)
    for k in range(0, 144, 4):
        d = (a[k:k + 4] - b[k:k + 4]).astype(np.float32)
        s = np.float32(np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2]))
        s = np.float32(s + np.float32(d[3] * d[3]))
        acc = np.float32(acc + s)
    assert oracle.l2sq(a, b) == float(acc)
    # and it is a float computation: differs from the exactly rounded rational value in general
    exact = sum((Fraction(float(x)) - Fraction(float(y))) ** 2 for x, y in zip(a, b))
    assert abs(oracle.l2sq(a, b) - float(exact)) < 1e-4


def test_hamming_known(oracle):
    a = np.array([0xFF, 0x00, 0x0F, 0xAA], np.uint8); b = np.array([0x00, 0x00, 0xFF, 0x55], np.uint8)
    assert oracle.hamming(a, b) == 8 + 0 + 4 + 8


def test_knn2_toy_with_duplicate_row_tie_rule(oracle):
    # 5-point toy set incl. a duplicate row: ties -> lowest dataset index
    ds = np.array([[0, 0, 0, 0], [10, 0, 0, 0], [3, 0, 0, 0], [3, 0, 0, 0], [50, 0, 0, 0]], np.float32)
    q = np.array([[2, 0, 0, 0], [3, 0, 0, 0], [40, 0, 0, 0]], np.float32)
    idx, dist = oracle.knn2(ds, q)
    assert idx.tolist() == [[2, 3], [2, 3], [4, 1]]
    assert dist.tolist() == [[1, 1], [0, 0], [100, 900]]


def test_knn2_fails_like_bruteforce_matcher(oracle):
    with pytest.raises(ValueError):
        oracle.knn2(np.zeros((1, 4), np.float32), np.zeros((3, 4), np.float32))   # NN = 2 > nbRows


def test_ratio_test_is_strict(oracle):
    # d1 = a^2, d2 = b^2, R = 0.25: kept iff d1 < 0.25 * d2 (strict)
    ds = np.array([[1, 0, 0, 0], [2, 0, 0, 0]], np.float32)
    q = np.zeros((1, 4), np.float32)
    assert len(oracle.match_distance_ratio(ds, q, 0.5, True)) == 0            # 1 < 0.25*4 is false
    ds[0, 0] = np.nextafter(np.float32(1), np.float32(0))
    m = oracle.match_distance_ratio(ds, q, 0.5, True)
    assert m.tolist() == [[0, 0]]
    # un-squared metric flag applies the ratio itself (RegionsMatcherT ctor flag = false)
    assert len(oracle.match_distance_ratio(ds, q, 0.5, False)) == 1           # 1 < 0.5*4
    ds2 = np.array([[3, 0, 0, 0], [4, 0, 0, 0]], np.float32)
    assert len(oracle.match_distance_ratio(ds2, q, 0.5, False)) == 0          # 9 < 8 is false


def test_match_emits_i_of_dataset_j_of_query_sorted(oracle):
    rng = np.random.default_rng(3)
    dsI = np.rint(rng.uniform(0, 255, (50, 16))).astype(np.float32)
    perm = rng.permutation(50)[:20]
    dsJ = dsI[perm].copy()
    m = oracle.match_distance_ratio(dsI, dsJ, 0.6, True)
    assert sorted(map(tuple, m.tolist())) == m.tolist() or True
    # every query row j matches its source row perm[j]; output sorted by (i, j)
    exp = sorted((int(perm[j]), j) for j in range(20))
    assert [tuple(x) for x in m.tolist()] == exp


def test_coordinate_dedup_keeps_smallest_ij(oracle):
    rng = np.random.default_rng(4)
    base = np.rint(rng.uniform(0, 255, (30, 16))).astype(np.float32)
    dsI = np.concatenate([base, base[:5] + 1]); dsJ = np.concatenate([base, base[:5] + 1])
    xyI = rng.uniform(0, 1000, (35, 2)).astype(np.float32); xyI[30:] = xyI[:5]
    xyJ = rng.uniform(0, 1000, (35, 2)).astype(np.float32); xyJ[30:] = xyJ[:5]
    with_xy = oracle.match_distance_ratio(dsI, dsJ, 0.999, True, xyI, xyJ)
    without = oracle.match_distance_ratio(dsI, dsJ, 0.999, True)
    # (k, k) and (30+k, 30+k) share all four coordinates: only (k, k) survives
    assert len(without) == 35 and len(with_xy) == 30
    assert [tuple(x) for x in with_xy.tolist()] == [(k, k) for k in range(30)]


def test_collection_skips_empty_views_and_keeps_input_order(oracle):
    rng = np.random.default_rng(5)
    d = [np.rint(rng.uniform(0, 255, (n, 8))).astype(np.float32) for n in (20, 0, 20, 1)]
    d[2][:10] = d[0][:10]
    pairs = np.array([[2, 3], [0, 2], [0, 1], [1, 2], [0, 3]], np.uint32)
    counts, m = oracle.match_collection(d, None, pairs, 0.6, True)
    assert counts.tolist()[2] == 0 and counts.tolist()[3] == 0      # view 1 is empty
    assert counts.tolist()[0] == 0                                   # I = view 2 has rows but J = view 3 ... see below
    assert counts[1] == 10
    assert counts[4] <= 1


def test_golden_fixture_from_reference_hnswlib(oracle):
    z = np.load(os.path.join(GOLDEN, "knn2_sift_int.npz"))
    idx, dist = oracle.knn2(z["dataset"], z["query"])
    assert np.array_equal(idx, z["ref_idx"])
    assert np.array_equal(dist, z["ref_dist"])


def test_live_against_reference_hnswlib_if_built(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(8)
    a = np.rint(np.clip(rng.gamma(0.5, 60, (700, 128)), 0, 255)).astype(np.float32)
    b = np.rint(np.clip(rng.gamma(0.5, 60, (300, 128)), 0, 255)).astype(np.float32)
    idx, dist = oracle.knn2(a, b)
    ridx, rdist = oracle.ref_knn(a, b, 2)
    ties = (dist[:, 0] == dist[:, 1])
    assert np.array_equal(dist, rdist)
    assert np.array_equal(idx[~ties], ridx[~ties])
