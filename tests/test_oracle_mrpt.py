"""CPU model of the MRPT plugin path (oracle/mrpt.c; matchingAlgorithm 5): the index has the shape Mrpt::grow gives it
(/root/reference/src/thirdparty/mrpt/mrpt.h:84-137, 1051-1078, 1664-1690) and the query does what Mrpt::query + ArrayMatcher_mrpt do
(mrpt.h:661-728, src/utils/matcher_mrpt.h:188-245).  PARITY UNPINNED (mrpt.h needs Eigen): these are property tests + the arm's
purpose, recall against the exhaustive matcher."""
import numpy as np
import pytest

from regard3d_amd import synth


def _scene(kind="sift", n=1500, seed=31):
    sc = synth.make_scene(2, n, kind, seed=seed)
    return np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32), sc


def test_depth_clamp_is_the_adapters(oracle):
    # max(2, min(depth, floor(log2 n) - 1))   (matcher_mrpt.h:93)
    assert [oracle.mrpt_depth_for(n, 6) for n in (5, 20, 100, 127, 128, 255, 256, 8192)] == [2, 3, 5, 5, 6, 6, 6, 6]
    assert oracle.mrpt_depth_for(8192, 3) == 3


def test_random_matrix_density_and_moments(oracle):
    R = oracle.mrpt_random_matrix(156, 128, 0.088, seed=0)
    nz = R != 0
    assert abs(nz.mean() - 0.088) < 0.01
    assert abs(R[nz].mean()) < 0.1 and abs(R[nz].std() - 1.0) < 0.1
    assert np.array_equal(R, oracle.mrpt_random_matrix(156, 128, 0.088, seed=0))
    assert not np.array_equal(R, oracle.mrpt_random_matrix(156, 128, 0.088, seed=1))
    # entry (row, col) does not depend on the shape: a shallower forest sees the same vectors row by row
    assert np.array_equal(oracle.mrpt_random_matrix(52, 128, 0.088, seed=0), R[:52])


@pytest.mark.parametrize("n", [128, 1000, 1501])
def test_trees_are_median_splits(oracle, n):
    d0, _, _ = _scene(n=n)
    ix = oracle.mrpt_build(d0)
    e = ix.export()
    depth = ix.depth
    assert depth == oracle.mrpt_depth_for(n, 6)
    # leaf sizes: a node of m rows gives m - m // 2 to the left (mrpt.h:1650-1662)
    def sizes(m, lvl):
        return [m] if lvl == depth else sizes(m - m // 2, lvl + 1) + sizes(m // 2, lvl + 1)
    assert np.array_equal(np.diff(e["leaf_first"]), sizes(n, 0)) and e["leaf_first"][-1] == n
    P = e["R"] @ d0.T.astype(np.float64)                     # projections (double: only compared with tolerances below)
    for t in range(ix.n_trees):
        rows = e["leaves"][t]
        assert np.array_equal(np.sort(rows), np.arange(n))   # every row in exactly one leaf
        # root: the left half's projections are <= the split <= the right half's
        left = rows[: n - n // 2]; right = rows[n - n // 2:]
        s = e["splits"][t, 0]
        p = P[t * depth]
        assert p[left].max() <= s + 1e-3 and p[right].min() >= s - 1e-3


def test_a_dataset_row_finds_itself(oracle):
    d0, _, _ = _scene(n=1200)
    ix = oracle.mrpt_build(d0)
    idx, dist, ne = ix.knn2(d0[:300], 5)
    ok = idx[:, 0] >= 0                                      # (a query is dropped when no SECOND row collects enough votes)
    assert ok.mean() > 0.3
    assert np.array_equal(idx[ok, 0], np.arange(300)[ok]) and np.all(dist[ok, 0] == 0.0)      # all n_trees votes go to the row itself
    assert np.all(idx[ok, 1] >= 0) and np.all(dist[ok, 1] > 0) and np.all(ne[ok] >= 2)


def test_retry_and_dropped_queries(oracle):
    d0, d1, _ = _scene(n=600, seed=5)
    ix = oracle.mrpt_build(d0, n_trees=6, depth=6)
    idx, dist, ne = ix.knn2(d1, 3)                           # half the trees must agree, then a third: a quarter of the queries stay without two rows
    dropped = idx[:, 0] < 0
    assert dropped.any() and (~dropped).any()
    assert np.all(idx[dropped] == -1) and np.all(dist[dropped] == -1.0)
    assert np.all(idx[~dropped, 0] != idx[~dropped, 1]) and np.all(dist[~dropped, 0] <= dist[~dropped, 1])


@pytest.mark.parametrize("kind", ["sift", "liop"])
def test_matches_of_the_exhaustive_matcher_are_mostly_recovered(oracle, kind):
    """the arm's purpose: of the putative matches the exhaustive matcher keeps at ratio 0.8, the forest recovers most"""
    sc = synth.make_scene(2, 2000, kind, seed=77)
    descs = [np.ascontiguousarray(d, np.float32) for d in sc.descs]
    pairs = np.array([[0, 1]], np.uint32)
    c_ex, m_ex = oracle.match_collection(descs, sc.xys, pairs, 0.8)
    c_mr, m_mr = oracle.match_collection_mrpt(descs, sc.xys, pairs, 0.8)
    ex = {tuple(x) for x in m_ex.tolist()}; mr = {tuple(x) for x in m_mr.tolist()}
    assert len(ex) > 200
    assert len(ex & mr) >= 0.6 * len(ex)
