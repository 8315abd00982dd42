"""GPU parity of the MRPT plugin path (matchingAlgorithm 5) -- kernels_mrpt.hip through the C ABI against its CPU model (oracle/mrpt.c):
  (i)   the index: random matrix, split points, leaf arrangement and leaf offsets, bit for bit / element for element;
  (ii)  r3dm_mrpt_knn2: rows and square-root distances of every query, dropped queries included, bit for bit;
  (iii) r3dm_match_pairs_mrpt: the match graph equals the un-squared ratio test / de-duplication applied to the model's neighbours;
  (iv)  the facade's arm 5 "as requested" writes that graph.
"""
import numpy as np
import pytest

from regard3d_amd import api, synth

pytestmark = pytest.mark.gpu


def _views(kind, n, seed, dim=None):
    sc = synth.make_scene(2, n, kind, seed=seed)
    d0, d1 = np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32)
    if dim is not None:
        d0, d1 = np.ascontiguousarray(d0[:, :dim]), np.ascontiguousarray(d1[:, :dim])
    return d0, d1, sc


@pytest.mark.parametrize("kind,n,dim", [("sift", 1000, None), ("liop", 1501, None), ("sift", 128, 64), ("sift", 5000, None), ("sift", 9000, None)])
def test_index_equals_the_cpu_model(ctx, oracle, kind, n, dim):
    d0, _, sc = _views(kind, n, 41, dim)
    ctx.clear_images()
    ctx.set_image(0, d0, sc.xys[0])
    mp = api.MrptParams.preset()
    g = ctx.mrpt_index(0, n, d0.shape[1], mp)
    ix = oracle.mrpt_build(d0, mp.n_trees, mp.depth, float(np.float32(1.0 / np.sqrt(np.float64(d0.shape[1])))), mp.seed)
    e = ix.export()
    assert g["depth"] == ix.depth
    assert np.array_equal(g["R"].view(np.uint32), e["R"].view(np.uint32))
    assert np.array_equal(g["leaf_first"], e["leaf_first"])
    assert np.array_equal(g["splits"].view(np.uint32), e["splits"].view(np.uint32))
    assert np.array_equal(g["leaves"], e["leaves"])


@pytest.mark.parametrize("kind,n,nq", [("sift", 1000, 1000), ("liop", 1501, 700), ("sift", 4096, 63), ("sift", 777, 1), ("sift", 9000, 300)])
def test_knn2_equals_the_cpu_model(ctx, oracle, kind, n, nq):
    d0, d1, _ = _views(kind, n, 43)
    q = d1[:nq]
    mp = api.MrptParams.preset()
    gi, gd = ctx.mrpt_knn2(d0, q, mp)
    ix = oracle.mrpt_build(d0, mp.n_trees, mp.depth, float(np.float32(1.0 / np.sqrt(np.float64(d0.shape[1])))), mp.seed)
    ei, ed, ne = ix.knn2(q, mp.votes)
    assert np.array_equal(gi, ei)
    assert np.array_equal(gd.view(np.uint32), ed.view(np.uint32))
    assert ctx.stats().n_ann_dist >= int(ne.sum())           # (the evaluations of first attempts that were retried count too)


def test_knn2_other_parameters(ctx, oracle):
    d0, d1, _ = _views("sift", 2000, 47)
    for n_trees, depth, votes, density in ((6, 6, 3, -1.0), (40, 4, 8, 0.2), (3, 2, 1, 1.0), (255, 3, 100, 0.05)):
        mp = api.MrptParams(n_trees, depth, votes, density, 7)
        gi, gd = ctx.mrpt_knn2(d0, d1[:400], mp)
        dens = density if density > 0 else float(np.float32(1.0 / np.sqrt(np.float64(d0.shape[1]))))
        ix = oracle.mrpt_build(d0, n_trees, depth, dens, 7)
        ei, ed, _ = ix.knn2(d1[:400], votes)
        assert np.array_equal(gi, ei), (n_trees, depth, votes)
        assert np.array_equal(gd.view(np.uint32), ed.view(np.uint32)), (n_trees, depth, votes)
    with pytest.raises(api.R3dmError):
        ctx.mrpt_knn2(d0, d1[:4], api.MrptParams(26, 7, 5, -1.0, 0))
    with pytest.raises(api.R3dmError):
        ctx.mrpt_knn2(d0, d1[:4], api.MrptParams(4, 6, 5, -1.0, 0))      # votes > n_trees
    with pytest.raises(api.R3dmError):
        ctx.mrpt_knn2(d0[:100], d1[:4], api.MrptParams.preset())          # fewer than 128 rows


def test_match_pairs_mrpt_equals_model_neighbours_plus_ratio_rules(ctx, oracle):
    sc = synth.make_scene(4, 1100, "sift", seed=613)
    small = synth.make_scene(1, 60, "sift", seed=614)                 # a view below 128 rows: scanned exactly
    descs = [d.astype(np.float32) for d in sc.descs] + [small.descs[0].astype(np.float32)]
    xys = list(sc.xys) + [small.xys[0]]
    ctx.clear_images()
    for v, (d, xy) in enumerate(zip(descs, xys)):
        ctx.set_image(v, d, xy)
    pairs = np.array([(i, j) for i in range(5) for j in range(i + 1, 5)], np.uint32)
    mp = api.MrptParams.preset()
    g = ctx.match_pairs_mrpt(pairs, 0.8, mp)
    assert ctx.stats().n_ann_built == 4
    counts, matches = oracle.match_collection_mrpt(descs, xys, pairs, 0.8)
    keep = counts > 0
    assert keep.sum() >= 6 and len(matches) > 100
    assert np.array_equal(g.pairs, pairs[keep])
    assert np.array_equal(np.diff(g.offsets.astype(np.int64)), counts[keep])
    assert np.array_equal(g.matches, matches)
    g2 = ctx.match_pairs_mrpt(pairs, 0.8, mp)                          # the same call again is served by the cached forests
    assert ctx.stats().n_ann_built == 0 and np.array_equal(g2.matches, matches)
    ctx.clear_images()
