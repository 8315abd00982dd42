"""Opt-in exact MFMA formulation of the Hamming matcher (r3dm_set_hamming_mfma, include/r3dm.h): bits as 0 / 1 bytes on
v_mfma_i32_32x32x32_i8, dataset tiles shared by a workgroup through LDS-DMA.  Bar: BIT-EXACT -- the same 2-NN indices and
distances as the CPU restatement (openMVG ArrayMatcherBruteForce<uint8, Hamming>) and the same match graphs as the default
popcount kernel, ties to the lowest row."""
import numpy as np
import pytest

from regard3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hctx(ctx):
    ctx.set_hamming_mfma(True)
    yield ctx
    ctx.set_hamming_mfma(False)


@pytest.mark.parametrize("nI,nJ,nbytes", [(1000, 700, 64), (1000, 700, 61), (777, 300, 32), (2, 5, 61), (33, 31, 64), (4100, 130, 61),
                                          (2048, 2048, 61), (97, 1500, 29)])
def test_knn2_hamming_bit_exact(hctx, oracle, nI, nJ, nbytes):
    rng = np.random.default_rng(nbytes * 1000 + nI)
    a = rng.integers(0, 256, (nI, nbytes), dtype=np.uint8)
    b = rng.integers(0, 256, (nJ, nbytes), dtype=np.uint8)
    m = min(100, nI, nJ)
    b[:m] = a[:m] ^ (rng.random((m, nbytes)) < 0.05).astype(np.uint8)
    idx, dist = hctx.knn2(a, b, binary=True)
    assert hctx.stats().n_hamming_mfma == 1
    oidx, odist = oracle.knn2(a, b, binary=True)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(dist.astype(np.uint32), odist)


def test_ties_duplicates_and_extreme_rows(hctx, oracle):
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (640, 61), dtype=np.uint8)
    a[100:200] = a[0:100]                         # duplicated rows: exact ties, lowest row wins
    a[300:364] = a[200]
    a[5] = 0; a[6] = 255; a[7] = 0
    b = np.concatenate([a[50:150], rng.integers(0, 256, (100, 61), dtype=np.uint8), np.zeros((2, 61), np.uint8), np.full((2, 61), 255, np.uint8)])
    idx, dist = hctx.knn2(a, b, binary=True)
    oidx, odist = oracle.knn2(a, b, binary=True)
    assert np.array_equal(idx, oidx) and np.array_equal(dist.astype(np.uint32), odist)


def test_akaze_collection_graph_equals_popcount_path_and_oracle(ctx, oracle):
    sc = synth.make_scene(6, 2200, "akaze", seed=33)
    pairs = sc.exhaustive_pairs()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000, binary=True)
    g0 = ctx.match_pairs(pairs, 0.8, False)
    assert ctx.stats().n_hamming_mfma == 0
    ctx.set_hamming_mfma(True)
    try:
        g1 = ctx.match_pairs(pairs, 0.8, False)
        assert ctx.stats().n_hamming_mfma == 1
    finally:
        ctx.set_hamming_mfma(False)
    assert np.array_equal(g0.pairs, g1.pairs) and np.array_equal(g0.offsets, g1.offsets) and np.array_equal(g0.matches, g1.matches)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.8, False, binary=True)
    assert np.array_equal(g1.pairs, pairs[counts > 0]) and np.array_equal(g1.matches, matches)
    ctx.clear_images()


def test_ragged_and_tiny_views(hctx, oracle):
    rng = np.random.default_rng(3)
    views = [rng.integers(0, 256, (n, 61), dtype=np.uint8) for n in (1, 2, 31, 32, 33, 500, 0)]
    hctx.clear_images()
    for i, v in enumerate(views):
        hctx.set_image(i, v if len(v) else np.zeros((0, 61), np.uint8), binary=True)
    ii, jj = np.triu_indices(len(views), k=1)
    pairs = np.stack([ii, jj], 1).astype(np.uint32)
    g = hctx.match_pairs(pairs, 0.9, False)
    counts, matches = oracle.match_collection(views, None, pairs, 0.9, False, binary=True)
    assert np.array_equal(g.pairs, pairs[counts > 0]) and np.array_equal(g.matches, matches)
    hctx.clear_images()
