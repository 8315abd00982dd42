"""Count tiles (kernels_match.hip, l2_knn2_counts_kernel): the split nominator's cheaper form for rows that are small integers times
a per-row scale -- what a LIOP descriptor is (vl_liop.c:553-575: integer votes divided by their norm).  Nomination runs on ONE f16
MFMA per 16 dimensions of the exact integer counts; everything behind it (f32 re-score in the reference's order, certification,
second chance, exact scan) is the split path's, so the bar is the same:

    BIT-EXACT 2-NN indices and float distances against the CPU restatement of the reference, identical graphs against the f32 tiles;
    a view with a single row that is not of the form keeps the split planes (n_counts_mfma tells which tiles ran).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def sctx(ctx):
    ctx.set_split_mfma(True)
    yield ctx
    ctx.set_split_mfma(False)
    ctx.clear_images()


def _liop_like(rng, n, dim, top=40, heavy=0.02):
    """integer vote vectors like vl_liop's, divided by their norm in f32 exactly as vl_liop.c does (float sum of squares, sqrt in
    double, float division)"""
    c = rng.poisson(rng.gamma(0.6, top / 0.6, (n, dim))).astype(np.float32)
    big = rng.random(n) < heavy
    c[big, rng.integers(0, dim, big.sum())] += rng.integers(300, 1900, big.sum())       # a few rows with one very full bin
    c = np.minimum(c, 2047.0)
    nz = c.sum(axis=1) == 0
    c[nz, 0] = 1.0
    norm = np.zeros(n, np.float32)
    for i in range(dim):                                    # float accumulation in index order
        norm = (norm + c[:, i] * c[:, i]).astype(np.float32)
    norm = np.maximum(np.sqrt(norm.astype(np.float64)), 1e-12).astype(np.float32)
    return (c / norm[:, None]).astype(np.float32), c


@pytest.mark.parametrize("nI,nJ,dim", [(700, 900, 144), (1500, 1200, 128), (300, 100, 37), (2100, 2050, 144), (97, 33, 64),
                                       (640, 500, 256), (2, 9, 144), (4100, 130, 100),
                                       (66000, 70, 144)])       # a dataset view beyond 65,536 rows: the two-list kernel (float keys)
def test_knn2_bit_exact_on_the_count_tiles(sctx, oracle, nI, nJ, dim):
    rng = np.random.default_rng(dim * 104729 + nI)
    a, ca = _liop_like(rng, nI, dim)
    b, cb = _liop_like(rng, nJ, dim)
    m = min(60, nI, nJ)
    cb[:m] = np.clip(ca[:m] + rng.integers(-2, 3, (m, dim)), 0, 2047)            # true correspondences: a few votes moved
    nb = np.sqrt((cb[:m].astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
    b[:m] = (cb[:m] / np.maximum(nb, 1e-12)[:, None]).astype(np.float32)
    idx, dist = sctx.knn2(a, b)
    s = sctx.stats()
    assert s.n_counts_mfma == 1 and s.n_split_mfma == 1 and s.n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_reference_built_liop_rows_run_on_the_count_tiles(sctx, oracle):
    """the rows vl_liop.c itself produced (tests/golden/liop_match_ref.npz): eligible, and bit for bit the restatement's 2-NN"""
    z = np.load(os.path.join(G, "liop_match_ref.npz"))
    A = (z["hist0"].astype(np.float32) / z["norm0"][:, None]).astype(np.float32)
    B = (z["hist1"].astype(np.float32) / z["norm1"][:, None]).astype(np.float32)
    idx, dist = sctx.knn2(A, B)
    assert sctx.stats().n_counts_mfma == 1
    oi, od = oracle.knn2(A, B)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert sctx.stats().n_exact_fallback < 0.01 * len(B)         # the certificate holds for nearly every query


def test_rows_that_are_not_counts_keep_the_split_planes(sctx, oracle):
    rng = np.random.default_rng(3)
    a, _ = _liop_like(rng, 800, 144)
    b, _ = _liop_like(rng, 600, 144)
    cases = {}
    r = a.copy(); r[411] = rng.gamma(0.5, 1.0, 144).astype(np.float32); r[411] /= np.linalg.norm(r[411]); cases["one real-valued row"] = (r, b)
    r = b.copy(); r[17, 5] = -r[17, 5] - 0.01; cases["a negative element"] = (a, r)
    r = a.copy(); r[3] = r[3] * np.float32(1.0 + 3e-5 * rng.random()) + np.float32(1e-4) * rng.random(144).astype(np.float32); cases["off the lattice"] = (r, b)
    for name, (x, y) in cases.items():
        idx, dist = sctx.knn2(x, y)
        s = sctx.stats()
        assert s.n_counts_mfma == 0 and s.n_split_mfma == 1, name
        oi, od = oracle.knn2(x, y)
        assert np.array_equal(dist, od) and np.array_equal(idx, oi), name
    # scaled count rows (any positive row scale, zero rows) ARE of the form
    x = a * rng.uniform(0.01, 300.0, (len(a), 1)).astype(np.float32); x[9] = 0.0
    y = b.copy(); y[4] = 0.0
    idx, dist = sctx.knn2(x, y)
    assert sctx.stats().n_counts_mfma == 1
    oi, od = oracle.knn2(x, y)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)


def test_collection_graph_equals_the_f32_tiles(ctx):
    rng = np.random.default_rng(11)
    views = []
    base, cbase = _liop_like(rng, 3000, 144)
    for v in range(5):
        c = cbase.copy()
        moved = rng.random(len(c)) < 0.6
        c[moved] = np.clip(c[moved] + rng.integers(-3, 4, (int(moved.sum()), 144)), 0, 2047)
        c[~moved] = _liop_like(rng, int((~moved).sum()), 144)[1]
        n = np.sqrt((c.astype(np.float64) ** 2).sum(axis=1)).astype(np.float32)
        views.append((c / np.maximum(n, 1e-12)[:, None]).astype(np.float32))
    xy = np.c_[np.arange(3000), np.zeros(3000)].astype(np.float32)
    pairs = np.array([(i, j) for i in range(5) for j in range(i + 1, 5)], np.uint32)
    out = {}
    for split in (False, True):
        ctx.clear_images()
        ctx.set_split_mfma(split)
        for i, v in enumerate(views):
            ctx.set_image(i, v, xy, 4000, 3000)
        g = ctx.match_pairs(pairs, 0.6, True)
        out[split] = (np.array(g.pairs), np.array(g.offsets), np.array(g.matches), ctx.stats().n_counts_mfma)
    ctx.set_split_mfma(False); ctx.clear_images()
    assert out[True][3] >= 1 and out[False][3] == 0
    for k in range(3):
        assert np.array_equal(out[True][k], out[False][k])
    assert out[True][2].shape[0] > 1000
