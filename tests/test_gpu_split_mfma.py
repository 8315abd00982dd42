"""Opt-in split-f16 nominator of the L2 matcher for real-valued descriptors (r3dm_set_split_mfma, include/r3dm.h): the all-pairs
contraction on v_mfma_f32_32x32x16_f16 with every value split into two f16 pieces; the nominees are re-scored in the reference
arithmetic and certified (second chance with four nominees, then the exact scan), so the bar is the one of the default path:

    BIT-EXACT 2-NN indices and float distances against the CPU restatement of the reference (openMVG ArrayMatcherBruteForce +
    L2_Vectorized), identical match graphs against the default f32-tile path.
"""
import numpy as np
import pytest

from regard3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sctx(ctx):
    ctx.set_split_mfma(True)
    yield ctx
    ctx.set_split_mfma(False)


def _unit(rng, n, dim):
    a = rng.gamma(0.5, 1.0, (n, dim)).astype(np.float32)
    return a / np.linalg.norm(a, axis=1, keepdims=True)


@pytest.mark.parametrize("nI,nJ,dim", [(700, 900, 144), (1500, 1200, 128), (300, 100, 37), (2100, 2050, 144), (97, 33, 64),
                                       (640, 500, 256), (2, 9, 144), (4100, 130, 100)])
def test_knn2_l2_real_valued_bit_exact_on_the_split_tiles(sctx, oracle, nI, nJ, dim):
    rng = np.random.default_rng(dim * 7919 + nI)
    a = _unit(rng, nI, dim); b = _unit(rng, nJ, dim)
    m = min(50, nI, nJ)
    b[:m] = a[:m] + rng.normal(0, 0.01, (m, dim)).astype(np.float32)
    idx, dist = sctx.knn2(a, b)
    s = sctx.stats()
    assert s.n_split_mfma == 1 and s.n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_value_ranges_scales_and_signs(sctx, oracle):
    """the per-view power-of-two scale: tiny, huge, signed and mixed magnitudes; rows of zeros; values below the f16 range of
    their view (the absolute part of the slack)"""
    rng = np.random.default_rng(5)
    for sa, sb in ((1.0, 1.0), (1e-6, 1e-6), (3e4, 3e4), (1.0, 37.0), (1e3, 1e-2), (255.0, 255.0)):
        a = (rng.normal(0, 1, (500, 128)) * sa).astype(np.float32); b = (rng.normal(0, 1, (400, 128)) * sb).astype(np.float32)
        b[:40] = (a[:40] / sa * sb * (1 + 0.01 * rng.normal(size=(40, 128)))).astype(np.float32)
        a[7] = 0.0; b[3] = 0.0
        a[9, :64] *= 1e-7                                  # pieces far below the view's f16 range
        idx, dist = sctx.knn2(a, b)
        assert sctx.stats().n_split_mfma == 1, (sa, sb)
        oidx, odist = oracle.knn2(a, b)
        assert np.array_equal(dist, odist) and np.array_equal(idx, oidx), (sa, sb)


def test_views_the_split_cannot_serve_keep_the_f32_tiles(sctx, oracle):
    rng = np.random.default_rng(6)
    # integer-valued batches are exact on the f32 tiles already
    a = np.rint(rng.uniform(0, 255, (300, 128))).astype(np.float32); b = np.rint(rng.uniform(0, 255, (200, 128))).astype(np.float32)
    idx, dist = sctx.knn2(a, b)
    assert sctx.stats().n_split_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # magnitudes 60 binary orders apart; a view of zeros; non-finite values
    for a, b in ((_unit(rng, 200, 64) * np.float32(1e9), _unit(rng, 100, 64) * np.float32(1e-9)),
                 (np.zeros((50, 32), np.float32), _unit(rng, 40, 32))):
        idx, dist = sctx.knn2(a, b)
        assert sctx.stats().n_split_mfma == 0
        oidx, odist = oracle.knn2(a, b)
        assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # descriptor lengths without a tensor kernel
    a = _unit(rng, 100, 300); b = _unit(rng, 60, 300)
    idx, dist = sctx.knn2(a, b)
    assert sctx.stats().n_split_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


def test_uncertified_queries_take_the_second_chance_and_the_exact_paths(sctx, oracle):
    """Near-duplicate rows: nothing can be certified (every gap is below the slack), so all queries end in the exact scans --
    the per-pair batched one (first 256 of a pair) and the overflow rescan."""
    rng = np.random.default_rng(21)
    base = rng.gamma(0.5, 1.0, 128).astype(np.float32); base /= np.linalg.norm(base)
    a = (base[None, :] * (1 + 1e-6 * rng.normal(size=(600, 128)))).astype(np.float32)
    b = (base[None, :] * (1 + 1e-6 * rng.normal(size=(333, 128)))).astype(np.float32)
    idx, dist = sctx.knn2(a, b)
    assert sctx.stats().n_split_mfma == 1 and sctx.stats().n_exact_fallback > 300
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # clusters of 3..5 near-identical rows: the merged top-2 is uncertifiable, the four-nominee stage decides most of them
    cent = _unit(rng, 400, 144)
    a = np.repeat(cent, 4, axis=0) * (1 + 2e-5 * rng.normal(size=(1600, 144))).astype(np.float32)
    a = a[rng.permutation(1600)].astype(np.float32)
    b = (cent * (1 + 2e-5 * rng.normal(size=(400, 144)))).astype(np.float32)
    idx, dist = sctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


def test_random_shapes_sweep(sctx, oracle):
    rng = np.random.default_rng(4048)
    ran = 0
    for trial in range(30):
        dim = int(rng.choice([8, 31, 32, 61, 64, 65, 100, 127, 128, 129, 140, 144, 200, 256]))
        nI = int(rng.integers(2, 900)); nJ = int(rng.integers(1, 600))
        if trial % 2 == 0:
            a = rng.normal(0, 1, (nI, dim)).astype(np.float32); b = rng.normal(0, 1, (nJ, dim)).astype(np.float32)
        else:
            a = (rng.random((nI, dim)) * 1000).astype(np.float32); b = (rng.random((nJ, dim)) * 1000).astype(np.float32)
        k = min(nI, nJ) // 3
        if k:
            b[:k] = a[rng.integers(0, nI, k)]                       # exact copies -> distance 0 and likely ties
        idx, dist = sctx.knn2(a, b)
        ran += sctx.stats().n_split_mfma
        oidx, odist = oracle.knn2(a, b)
        assert np.array_equal(dist, odist), (trial, dim, nI, nJ)
        assert np.array_equal(idx, oidx), (trial, dim, nI, nJ)
    assert ran >= 25


def test_liop_collection_graph_equals_the_default_path_and_the_oracle(ctx, oracle):
    sc = synth.make_scene(6, 2500, "liop", seed=91)
    pairs = sc.exhaustive_pairs()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    g0 = ctx.match_pairs(pairs, 0.6, True)
    s0 = ctx.stats()
    assert s0.n_split_mfma == 0
    ctx.set_split_mfma(True)
    try:
        g1 = ctx.match_pairs(pairs, 0.6, True)
        s1 = ctx.stats()
    finally:
        ctx.set_split_mfma(False)
    assert s1.n_split_mfma == 1
    assert s1.n_exact_fallback <= s1.n_queries // 100                     # < 1 % of the queries need the exact scan
    assert np.array_equal(g0.pairs, g1.pairs) and np.array_equal(g0.offsets, g1.offsets) and np.array_equal(g0.matches, g1.matches)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    assert np.array_equal(g1.pairs, pairs[counts > 0]) and np.array_equal(g1.matches, matches)
    f0 = ctx.filter_F(g0); f1 = ctx.filter_F(g1)
    assert np.array_equal(f0.matches, f1.matches) and f0.num_pairs > 0
    ctx.clear_images()
