"""GPU parity tests: the HIP path (through the C ABI) against the CPU restatement (oracle/).

Bars: bit-exact 2-NN indices AND distances for L2 (the kernel
re-scores its candidates with the reference arithmetic) and Hamming; identical match graphs;
AC-RANSAC inlier sets identical for a shared sample stream (tolerance: see test body).
"""
import numpy as np
import pytest

from regard3d_amd import synth

pytestmark = pytest.mark.gpu


def _graph_equal(g, pairs, counts, matches):
    d = g.as_dict()
    off = 0
    exp = {}
    for p, (I, J) in enumerate(pairs):
        if counts[p]:
            exp[(int(I), int(J))] = matches[off:off + counts[p]]
        off += counts[p]
    assert set(d.keys()) == set(exp.keys())
    for k in exp:
        assert np.array_equal(d[k], exp[k]), f"pair {k}"


@pytest.mark.parametrize("nI,nJ,dim", [(100, 70, 128), (257, 1000, 128), (2048, 2048, 128), (33, 31, 64),
                                       (500, 300, 144), (300, 200, 100), (64, 64, 256), (2, 5, 128)])
def test_knn2_l2_integer_sift(ctx, oracle, nI, nJ, dim):
    rng = np.random.default_rng(nI * 7919 + nJ)
    a = np.rint(np.clip(rng.gamma(0.5, 60.0, (nI, dim)), 0, 255)).astype(np.float32)
    b = np.rint(np.clip(rng.gamma(0.5, 60.0, (nJ, dim)), 0, 255)).astype(np.float32)
    b[: min(nJ, nI) // 2] = np.clip(a[: min(nJ, nI) // 2] + np.rint(rng.normal(0, 4, (min(nJ, nI) // 2, dim))), 0, 255)
    idx, dist = ctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


@pytest.mark.parametrize("nI,nJ,dim", [(700, 900, 144), (1500, 1200, 128), (300, 100, 37)])
def test_knn2_l2_real_valued(ctx, oracle, nI, nJ, dim):
    rng = np.random.default_rng(dim)
    a = rng.gamma(0.5, 1.0, (nI, dim)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.gamma(0.5, 1.0, (nJ, dim)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[:50] = a[:50] + rng.normal(0, 0.01, (50, dim)).astype(np.float32)
    idx, dist = ctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_knn2_l2_exact_ties_and_duplicates(ctx, oracle):
    # duplicated dataset rows: lowest row index must win, runner-up is the duplicate (distance tie)
    rng = np.random.default_rng(5)
    a = np.rint(rng.uniform(0, 255, (200, 128))).astype(np.float32)
    a[150] = a[3]; a[77] = a[3]; a[199] = a[10]
    b = a[[3, 10, 50, 77]].copy()
    idx, dist = ctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    assert idx[0].tolist() == [3, 77]


def test_knn2_u8(ctx, oracle):
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (600, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (400, 128), dtype=np.uint8)
    idx, dist = ctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)


@pytest.mark.parametrize("nbytes", [64, 61, 32])
def test_knn2_hamming(ctx, oracle, nbytes):
    rng = np.random.default_rng(nbytes)
    a = rng.integers(0, 256, (1000, nbytes), dtype=np.uint8)
    b = rng.integers(0, 256, (700, nbytes), dtype=np.uint8)
    b[:100] = a[:100] ^ (rng.random((100, nbytes)) < 0.05).astype(np.uint8)
    idx, dist = ctx.knn2(a, b, binary=True)
    oidx, odist = oracle.knn2(a, b, binary=True)
    assert np.array_equal(idx, oidx)
    assert np.array_equal(dist.astype(np.uint32), odist)


def test_knn2_rejects_degenerate_sizes(ctx):
    from regard3d_amd.api import R3dmError
    a = np.zeros((1, 128), np.float32); b = np.zeros((4, 128), np.float32)
    with pytest.raises(R3dmError):
        ctx.knn2(a, b)           # NN = 2 > nbRows, like ArrayMatcherBruteForce


@pytest.mark.parametrize("kind,n_img,n_feat", [("sift", 6, 1024), ("liop", 5, 700), ("akaze", 6, 1000)])
def test_match_collection_graph(ctx, oracle, kind, n_img, n_feat):
    sc = synth.make_scene(n_img, n_feat, kind, seed=1001)
    binary = kind == "akaze"
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], int(sc.widths[i]), int(sc.heights[i]), binary=binary)
    pairs = sc.exhaustive_pairs()
    ratio, squared = (0.8, False) if binary else (0.6, True)
    g = ctx.match_pairs(pairs, ratio, squared)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, ratio, squared, binary=binary)
    _graph_equal(g, pairs, counts, matches)
    assert g.num_matches > 0


def test_match_ragged_and_empty_views(ctx, oracle):
    sc = synth.make_scene(5, 600, "sift", seed=77)
    sc.descs[1] = sc.descs[1][:37]; sc.xys[1] = sc.xys[1][:37]
    sc.descs[2] = sc.descs[2][:0]; sc.xys[2] = sc.xys[2][:0]
    sc.descs[3] = sc.descs[3][:1]; sc.xys[3] = sc.xys[3][:1]
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.6, True)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    _graph_equal(g, pairs, counts, matches)


def test_match_coordinate_dedup(ctx, oracle):
    # two features of I and of J at identical positions with near-identical descriptors:
    # IndMatchDecorator keeps one match per (xI,yI,xJ,yJ)
    sc = synth.make_scene(2, 512, "sift", seed=9)
    for im in (0, 1):
        sc.descs[im] = np.concatenate([sc.descs[im], sc.descs[im][:20]])      # duplicates of the first 20 features
        sc.xys[im] = np.concatenate([sc.xys[im], sc.xys[im][:20]])
    ctx.clear_images()
    for i in range(2):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.99, True)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.99, True)
    _graph_equal(g, pairs, counts, matches)


def test_filter_F_matches_oracle(ctx, oracle):
    sc = synth.make_scene(6, 1500, "sift", seed=2002)
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], int(sc.widths[i]), int(sc.heights[i]))
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.6, True)
    gf, F = ctx.filter_F(g, 4.0, 2048, seed=5489, want_F=True)
    gp, go, gm = g.pairs, g.offsets, g.matches
    counts = np.diff(go.astype(np.int64)).astype(np.uint32)
    oc, om, oF = oracle.filter_F_collection(sc.xys, sc.widths, sc.heights, gp, counts, gm, 4.0, 2048, 5489, want_F=True)
    d = gf.as_dict()
    off = 0
    n_kept = 0
    for p, (I, J) in enumerate(gp):
        key = (int(I), int(J))
        if oc[p]:
            exp = om[off:off + oc[p]]
            assert key in d, f"pair {key} rejected on the GPU, kept by the oracle"
            # identical inlier SET; order (ascending residual) may differ only between residuals that tie to 1e-12
            assert set(map(tuple, d[key].tolist())) == set(map(tuple, exp.tolist())), f"pair {key}"
            Fg = F[n_kept] / np.linalg.norm(F[n_kept]); Fo = oF[p].reshape(9) / np.linalg.norm(oF[p])
            if np.dot(Fg, Fo) < 0: Fg = -Fg
            assert np.linalg.norm(Fg - Fo) < 1e-9
            n_kept += 1
        else:
            assert key not in d
        off += oc[p]
    assert n_kept >= 9


def test_save_load_roundtrip(ctx, oracle, tmp_path):
    sc = synth.make_scene(4, 512, "sift", seed=3)
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    g = ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
    from regard3d_amd.api import Graph
    for ext in ("txt", "bin"):
        path = str(tmp_path / f"matches.putative.{ext}")
        g.save(path)
        g2 = Graph.load(path)
        assert np.array_equal(g.pairs, g2.pairs) and np.array_equal(g.matches, g2.matches)
        # the oracle's reader agrees with the library's writer
        p, c, m = oracle.load_matches(path)
        assert np.array_equal(p, g.pairs) and np.array_equal(m, g.matches)


def test_uncertified_queries_take_the_exact_paths(ctx, oracle):
    """Near-duplicate real-valued rows: the MFMA nomination cannot be certified, so every query goes
    through the per-pair batched exact scan (first 128) and the overflow rescan (the rest)."""
    rng = np.random.default_rng(21)
    base = rng.gamma(0.5, 1.0, 128).astype(np.float32); base /= np.linalg.norm(base)
    a = (base[None, :] * (1 + 1e-6 * rng.normal(size=(600, 128)))).astype(np.float32)
    b = (base[None, :] * (1 + 1e-6 * rng.normal(size=(333, 128)))).astype(np.float32)
    idx, dist = ctx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # same through the collection API (stats expose how many queries needed the exact scan)
    ctx.clear_images()
    ctx.set_image(0, a); ctx.set_image(1, b); ctx.set_image(2, a[:100])
    g = ctx.match_pairs(np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.999, True)
    assert ctx.stats().n_exact_fallback > 300
    counts, matches = oracle.match_collection([a, b, a[:100]], None, np.array([[0, 1], [0, 2], [1, 2]], np.uint32), 0.999, True)
    _graph_equal(g, np.array([[0, 1], [0, 2], [1, 2]]), counts, matches)


def test_integer_descriptors_need_no_rounding_slack(ctx):
    """SIFT-like integer bins: the exactness proof applies, (almost) nothing falls back."""
    sc = synth.make_scene(3, 2048, "sift", seed=5)
    ctx.clear_images()
    for i in range(3):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
    s = ctx.stats()
    assert s.n_exact_fallback <= s.n_queries // 1000


def test_filter_H_matches_oracle(ctx, oracle):
    """Homography AC-RANSAC (GeometricFilter_HMatrix_AC): planar scene so that H has real support."""
    rng = np.random.default_rng(12)
    n = 900
    Ht = np.array([[0.97, 0.05, 60.0], [-0.04, 1.01, 30.0], [2e-5, 1e-5, 1.0]])
    base = np.rint(np.clip(rng.gamma(0.5, 60.0, (n, 128)), 0, 255)).astype(np.float32)
    xy0 = np.stack([rng.uniform(200, 3800, n), rng.uniform(200, 2800, n)], 1)
    descs, xys = [], []
    for v in range(3):
        Hv = np.linalg.matrix_power(Ht, v)
        p = (Hv @ np.c_[xy0, np.ones(n)].T).T; p = p[:, :2] / p[:, 2:] + rng.normal(0, 0.4, (n, 2))
        bad = rng.random(n) < 0.2
        p[bad] = np.stack([rng.uniform(0, 4000, bad.sum()), rng.uniform(0, 3000, bad.sum())], 1)
        d = np.rint(np.clip(base + rng.normal(0, 4, base.shape), 0, 255)).astype(np.float32)
        perm = rng.permutation(n)
        descs.append(d[perm]); xys.append(p[perm].astype(np.float32))
    ctx.clear_images()
    for i in range(3):
        ctx.set_image(i, descs[i], xys[i], 4000, 3000)
    pairs = np.array([[0, 1], [0, 2], [1, 2]], np.uint32)
    g = ctx.match_pairs(pairs, 0.6, True)
    gh, Hm = ctx.filter_H(g, 4.0, 2048, seed=5489, want_H=True)
    counts = np.diff(g.offsets.astype(np.int64)).astype(np.uint32)
    oc, om, oH = oracle.filter_H_collection(xys, [4000] * 3, [3000] * 3, g.pairs, counts, g.matches, 4.0, 2048, 5489, want_F=True)
    d = gh.as_dict(); off = 0; kept = 0
    for p, (I, J) in enumerate(g.pairs):
        exp = om[off:off + oc[p]]; off += oc[p]
        got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
        assert set(map(tuple, got.tolist())) == set(map(tuple, exp.tolist())), (I, J)
        if oc[p]:
            a = Hm[kept] / np.linalg.norm(Hm[kept]); b = oH[p] / np.linalg.norm(oH[p])
            if np.dot(a, b) < 0: a = -a
            assert np.linalg.norm(a - b) < 1e-9
            kept += 1
    assert kept == 3 and gh.num_matches > 1200


def test_liop_kernel_bit_exact(ctx, oracle):
    """LIOP on the GPU vs the reference routine (committed golden = output of the reference's own vl_liop.c;
    live reference build when oracle/_ref travelled; restatement otherwise)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "liop_patches.npz"))
    d, n_resorted = ctx.liop_describe_patches(z["patches"])
    assert np.array_equal(d, z["ref_desc"])
    assert n_resorted >= 3                                       # the quantised / half-flat patches took the exact re-sort
    rng = np.random.default_rng(9)
    from scipy.ndimage import gaussian_filter
    P = np.stack([gaussian_filter(rng.random((41, 41)), 1.2).astype(np.float32) for _ in range(2000)])
    P[:64] = np.round(P[:64] * 64) / 64
    d, _ = ctx.liop_describe_patches(P)
    exp = oracle.ref_liop(P) if oracle.ref_liop_lib() is not None else oracle.liop_describe(P)
    assert np.array_equal(d, exp)
    assert ctx.liop_describe_patches(P[:0])[0].shape == (0, 144)


def test_extract_liop_patches_and_descriptors(ctx, oracle):
    """keypoints -> 41x41 warped + blurred patches -> LIOP: GPU vs the restatement (patch stage) and vs the
    reference routine (LIOP stage), bit-exact, incl. keypoints whose patch leaves the image."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(31)
    img = gaussian_filter(rng.random((600, 800)), 1.5).astype(np.float32)
    img[200:260, 300:380] = 1.0                                             # saturated region -> equal intensities
    n = 1500
    kps = np.stack([rng.uniform(-5, 805, n), rng.uniform(-5, 605, n), rng.uniform(1.5, 12, n), rng.uniform(0, 360, n)], 1).astype(np.float32)
    kps[:50, 0] = rng.uniform(310, 370, 50); kps[:50, 1] = rng.uniform(210, 250, 50); kps[:50, 2] = 2.0
    desc, patches = ctx.extract_liop(img, kps, 8.0, want_patches=True)
    exp_p = oracle.liop_extract_patches(img, kps, 8.0)
    assert np.array_equal(patches, exp_p)
    exp_d = oracle.ref_liop(exp_p) if oracle.ref_liop_lib() is not None else oracle.liop_describe(exp_p)
    assert np.array_equal(desc, exp_d)
    # without patches_out the warp + blur runs inside the descriptor kernel (the features stage's path): the same descriptors
    assert np.array_equal(ctx.extract_liop(img, kps, 8.0), exp_d)
    nrm = np.linalg.norm(desc.astype(np.float64), axis=1)
    assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))


def test_knn2_random_shapes_sweep(ctx, oracle):
    """Randomised sweep over sizes / descriptor lengths / value ranges (all tensor-kernel variants, the scalar-tail
    exact path, integer and real-valued data, exact duplicates)."""
    rng = np.random.default_rng(2024)
    for trial in range(40):
        dim = int(rng.choice([3, 8, 31, 32, 61, 64, 65, 100, 127, 128, 129, 140, 144, 150, 200, 256, 300]))
        nI = int(rng.integers(2, 700)); nJ = int(rng.integers(1, 500))
        if trial % 3 == 0:
            a = np.rint(rng.uniform(0, 255, (nI, dim))).astype(np.float32); b = np.rint(rng.uniform(0, 255, (nJ, dim))).astype(np.float32)
        elif trial % 3 == 1:
            a = rng.normal(0, 1, (nI, dim)).astype(np.float32); b = rng.normal(0, 1, (nJ, dim)).astype(np.float32)
        else:
            a = (rng.random((nI, dim)) * 1000).astype(np.float32); b = (rng.random((nJ, dim)) * 1000).astype(np.float32)
        k = min(nI, nJ) // 3
        if k:
            b[:k] = a[rng.integers(0, nI, k)]                       # exact copies -> distance 0 and likely ties
        idx, dist = ctx.knn2(a, b)
        oidx, odist = oracle.knn2(a, b)
        assert np.array_equal(dist, odist), (trial, dim, nI, nJ)
        assert np.array_equal(idx, oidx), (trial, dim, nI, nJ)


def test_filter_E_matches_oracle(ctx, oracle):
    """Essential-matrix AC-RANSAC (GeometricFilter_EMatrix_AC + the reference's overlap rule): same sample stream, the
    5-point solver uses only + - * / after its QR, so inlier sets AND matrices agree with the CPU restatement."""
    sc = synth.make_scene(5, 1400, "liop", seed=47)
    ctx.clear_images()
    K = synth.intrinsics()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
        if i != 4:
            ctx.set_intrinsics(i, K)                                  # view 4 has no intrinsics: its pairs are not estimated
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.6, True)
    counts = np.diff(g.offsets.astype(np.int64)).astype(np.uint32)
    Ks = np.stack([K] * 5); Ks[4] = 0
    for mc, mr in ((50, 0.3), (0, 0.0)):
        ge, Em = ctx.filter_E(g, 4.0, 2048, seed=5489, min_count=mc, min_ratio=mr, want_E=True)
        oc, om, oE = oracle.filter_E_collection(sc.xys, sc.widths, sc.heights, Ks, g.pairs, counts, g.matches, 4.0, 2048, 5489,
                                                prune_min_count=mc, prune_min_ratio=mr, want_E=True)
        d = ge.as_dict(); off = 0; kept = 0
        for p, (I, J) in enumerate(g.pairs):
            exp = om[off:off + oc[p]]; off += oc[p]
            got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
            assert set(map(tuple, got.tolist())) == set(map(tuple, exp.tolist())), (I, J, len(got), len(exp))
            if oc[p]:
                a = Em[kept] / np.linalg.norm(Em[kept]); b = oE[p] / np.linalg.norm(oE[p])
                if np.dot(a, b) < 0: a = -a
                assert np.linalg.norm(a - b) < 1e-9
                sv = np.linalg.svd(Em[kept].reshape(3, 3), compute_uv=False)
                assert abs(sv[0] - sv[1]) < 1e-6 * sv[0] and sv[2] < 1e-6 * sv[0]
                kept += 1
        assert kept >= 3 and all(4 not in k for k in d)
    rep = ctx.filter_report()
    assert any(r[3] > r[2] for r in rep if r[2])                          # (.., iterations, models, ..): several E per minimal sample


def _two_view_scene(rng, n, dim, frac_match=1.0):
    """n features per view; the first frac_match*n of view 1 are noisy copies of view-0 rows (shuffled), seen by a second camera"""
    A = np.rint(rng.uniform(0, 255, (n, dim))).astype(np.float32)
    nm = int(frac_match * n)
    B = np.rint(rng.uniform(0, 255, (n, dim))).astype(np.float32)
    src = rng.permutation(n)[:nm]
    B[:nm] = np.clip(A[src] + np.rint(rng.normal(0, 2, (nm, dim))), 0, 255)
    X = np.c_[rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(8, 14, n)]
    f = 4800.0
    xyA = np.c_[f * X[:, 0] / X[:, 2] + 2000, f * X[:, 1] / X[:, 2] + 1500]
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Y = X @ R.T + np.array([0.8, 0.05, 0.1])
    xyB_all = np.c_[f * Y[:, 0] / Y[:, 2] + 2000, f * Y[:, 1] / Y[:, 2] + 1500] + rng.normal(0, 0.4, (n, 2))
    xyB = np.c_[rng.uniform(0, 4000, n), rng.uniform(0, 3000, n)]
    xyB[:nm] = xyB_all[src]
    perm = rng.permutation(n)
    return A, xyA.astype(np.float32), B[perm], xyB[perm].astype(np.float32)


@pytest.mark.parametrize("n,frac", [(20000, 1.0), (18000, 0.3)])
def test_views_beyond_the_lds_sort_budget(ctx, oracle, n, frac):
    """nFeatures_ defaults to 20000 (src/Regard3DFeatures.cpp:128): more rows than the 16384-key LDS sort of the finalisation
    kernel -> pairs that keep more than 16384 matches sort in global scratch; more than 8192 putatives -> the filter spills too."""
    rng = np.random.default_rng(n)
    A, xyA, B, xyB = _two_view_scene(rng, n, 16, frac)
    ctx.clear_images()
    ctx.set_image(0, A, xyA, 4000, 3000); ctx.set_image(1, B, xyB, 4000, 3000)
    pairs = np.array([[0, 1]], np.uint32)
    g = ctx.match_pairs(pairs, 0.6, True)
    counts, matches = oracle.match_collection([A, B], [xyA, xyB], pairs, 0.6, True)
    assert counts[0] > (16384 if frac == 1.0 else 4000)
    _graph_equal(g, pairs, counts, matches)
    gf, F = ctx.filter_F(g, 4.0, 2048, seed=5489, want_F=True)
    oc, om, oF = oracle.filter_F_collection([xyA, xyB], [4000, 4000], [3000, 3000], pairs, counts, matches, 4.0, 2048, 5489, want_F=True)
    assert oc[0] > 0.9 * counts[0] and gf.num_pairs == 1
    assert set(map(tuple, gf.matches.tolist())) == set(map(tuple, om.tolist()))
    a = F[0] / np.linalg.norm(F[0]); b = oF[0] / np.linalg.norm(oF[0])
    assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-9


@pytest.mark.parametrize("n,frac", [(11500, 0.95), (7000, 0.9), (22000, 0.95)])
def test_filters_on_long_match_lists(ctx, oracle, n, frac):
    """The wide (512-thread) variant of the AC-RANSAC kernel, chosen when a pair has more than 4096 putatives: ~10 k matches
    (beyond the 8192-entry LDS list: global list, register sort), ~6 k (LDS list) and ~20 k (global list, plain network) --
    F, E and H inlier sets and models equal to the CPU restatement's, as on the 256-thread variant."""
    rng = np.random.default_rng(n + 5)
    A, xyA, B, xyB = _two_view_scene(rng, n, 16, frac)
    K = np.array([[4800.0, 0, 2000], [0, 4800.0, 1500], [0, 0, 1]])          # the cameras of _two_view_scene
    ctx.clear_images()
    ctx.set_image(0, A, xyA, 4000, 3000); ctx.set_image(1, B, xyB, 4000, 3000)
    ctx.set_intrinsics(0, K); ctx.set_intrinsics(1, K)
    pairs = np.array([[0, 1]], np.uint32)
    g = ctx.match_pairs(pairs, 0.6, True)
    counts = np.diff(g.offsets.astype(np.int64)).astype(np.uint32)
    assert counts[0] > 0.8 * n * frac
    xys, W, H = [xyA, xyB], [4000, 4000], [3000, 3000]
    gf, F = ctx.filter_F(g, 4.0, 2048, seed=5489, want_F=True)
    oc, om, oF = oracle.filter_F_collection(xys, W, H, pairs, counts, g.matches, 4.0, 2048, 5489, want_F=True)
    assert oc[0] > 0.5 * counts[0] and set(map(tuple, gf.matches.tolist())) == set(map(tuple, om.tolist()))
    a = F[0] / np.linalg.norm(F[0]); b = oF[0] / np.linalg.norm(oF[0])
    assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-9
    ge, Em = ctx.filter_E(g, 4.0, 2048, seed=5489, min_count=50, min_ratio=0.3, want_E=True)
    ec, em, oE = oracle.filter_E_collection(xys, W, H, np.stack([K, K]), pairs, counts, g.matches, 4.0, 2048, 5489,
                                            prune_min_count=50, prune_min_ratio=0.3, want_E=True)
    assert set(map(tuple, ge.matches.tolist())) == set(map(tuple, em.tolist())) and ec[0] > 0
    a = Em[0] / np.linalg.norm(Em[0]); b = oE[0] / np.linalg.norm(oE[0])
    assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-9
    gh, Hm = ctx.filter_H(g, 4.0, 2048, seed=5489, want_H=True)
    hc, hm, oH = oracle.filter_H_collection(xys, W, H, pairs, counts, g.matches, 4.0, 2048, 5489, want_F=True)
    assert set(map(tuple, gh.matches.tolist())) == set(map(tuple, hm.tolist()))
    if hc[0]:
        a = Hm[0] / np.linalg.norm(Hm[0]); b = oH[0] / np.linalg.norm(oH[0])
        assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-9


def test_filters_side_by_side_equal_the_single_calls(ctx):
    """r3dm_filter_FEH: F, E and H on three streams at once -> the graphs of r3dm_filter_F / _E / _H, whatever the subset"""
    sc = synth.make_scene(6, 1500, "liop", seed=91)
    K = synth.intrinsics()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); ctx.set_intrinsics(i, K)
    g = ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
    one = {"F": ctx.filter_F(g), "E": ctx.filter_E(g), "H": ctx.filter_H(g)}
    for which in ("FEH", "FH", "E", "EH"):
        for rep in range(2):
            got, msk, msw = ctx.filter_FEH(g, which)
            for k in which:
                assert np.array_equal(got[k].pairs, one[k].pairs) and np.array_equal(got[k].offsets, one[k].offsets), (which, k)
                assert np.array_equal(got[k].matches, one[k].matches), (which, k)
            assert all(msk["FEH".index(k)] > 0 for k in which)
    assert one["F"].num_pairs >= 5


def test_filters_on_degenerate_and_tiny_pairs(ctx, oracle):
    """Hand-made putative graphs that push the solvers into their degenerate branches (NaN / inf residuals, no real roots,
    singular systems, samples as large as the list): collinear points, one repeated point, pure noise, lists of SS+1 .. SS+6
    matches, a pure translation.  The three filters must agree with the CPU restatement pair by pair."""
    from regard3d_amd import api
    rng = np.random.default_rng(77)
    n = 400
    base = np.c_[rng.uniform(100, 3900, n), rng.uniform(100, 2900, n)]
    line = np.c_[np.linspace(200, 3800, n), np.linspace(300, 2700, n)]
    same = np.tile([[1234.5, 987.25]], (n, 1))
    views = [base, base + [35.0, -12.0], line, line[::-1].copy(), same, rng.uniform(0, 3000, (n, 2)), rng.uniform(0, 3000, (n, 2))]
    views = [v.astype(np.float32) for v in views]
    ctx.clear_images()
    K = synth.intrinsics()
    dummy = np.zeros((n, 128), np.float32)
    for i, v in enumerate(views):
        ctx.set_image(i, dummy, v, 4000, 3000); ctx.set_intrinsics(i, K)
    ident = np.c_[np.arange(n), np.arange(n)].astype(np.uint32)
    spec = [((0, 1), ident),                                   # pure translation: F / E degenerate-ish, H exact
            ((2, 3), ident),                                   # collinear in both views
            ((4, 5), ident), ((0, 4), ident),                  # one repeated point on one side
            ((5, 6), ident)]                                   # pure noise
    for k, m in enumerate(range(5, 14)):                       # tiny lists around the minimal sample sizes
        spec.append(((0, 2 + (k % 5)), ident[rng.permutation(n)[:m]]))
    spec.sort(key=lambda s: s[0])
    seen = set(); spec = [s for s in spec if not (s[0] in seen or seen.add(s[0]))]
    pairs = np.array([s[0] for s in spec], np.uint32)
    counts = np.array([len(s[1]) for s in spec], np.uint32)
    matches = np.concatenate([s[1][np.lexsort((s[1][:, 1], s[1][:, 0]))] for s in spec]).astype(np.uint32)
    g = api.Graph.from_csr(pairs, np.r_[0, np.cumsum(counts)].astype(np.uint64), matches)
    W = [4000] * len(views); Hh = [3000] * len(views)
    Ks = np.stack([K] * len(views))
    bad = []
    for name, gpu, cpu in (("F", lambda: ctx.filter_F(g), lambda: oracle.filter_F_collection(views, W, Hh, pairs, counts, matches)),
                           ("H", lambda: ctx.filter_H(g), lambda: oracle.filter_H_collection(views, W, Hh, pairs, counts, matches)),
                           ("E", lambda: ctx.filter_E(g), lambda: oracle.filter_E_collection(views, W, Hh, Ks, pairs, counts, matches))):
        d = gpu().as_dict()
        oc, om = cpu()[:2]
        off = 0
        for p, (I, J) in enumerate(pairs):
            exp = om[off:off + oc[p]]; off += oc[p]
            got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
            if name == "F" and 4 in (int(I), int(J)):
                # every correspondence shares one image point: the 7-point system has rank 3, its "null space" is rounding
                # noise and the cubic is ill-conditioned, so the 1-ulp differences between glibc's and the device's
                # acos / cos / cbrt pick different models.  Both sides must still return a valid subset.
                assert len(got) <= counts[p] and len(exp) <= counts[p]
                continue
            if set(map(tuple, got.tolist())) != set(map(tuple, exp.tolist())):
                bad.append((name, int(I), int(J), int(counts[p]), len(got), len(exp)))
    assert not bad, bad
    assert (0, 1) in ctx.filter_H(g).as_dict()                 # the translation is a perfect homography


@pytest.mark.gpu
@pytest.mark.parametrize("scout", ["0", "3"])
def test_filters_skip_no_model_that_could_win(ctx, scout):
    """The AC-RANSAC kernels skip the sort of a model whose histogram bound on the NFA stays above the best NFA so far
    (kernels_filter.hip).  The developer build with R3DM_FILTER_CHECK=1 skips nothing and checks `bound <= NFA` on every model it
    evaluates (invariant 8 -> error); its inlier sets and models must equal the product's.  R3DM_FILTER_SCOUT=3 beside it: the scout
    pass runs too (reciprocal intervals instead of divisions, the NFA bound from the two ends of every bin's count range less the
    tables' drift) and every model's exact count and NFA are checked against what the scout promised (invariants 9, 10)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = synth.make_scene(7, 4000, "sift", seed=612)
    K = synth.intrinsics()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); ctx.set_intrinsics(i, K)
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.6, True)
    prod = {}
    for name, fn, kw in (("F", ctx.filter_F, "want_F"), ("H", ctx.filter_H, "want_H"), ("E", ctx.filter_E, "want_E")):
        out, M = fn(g, **{kw: True})
        prod[name] = (np.array(out.pairs), np.array(out.offsets), np.array(out.matches), M, np.array(ctx.filter_report(), np.float64))
    assert len(prod["F"][0]) > 0 and len(prod["E"][0]) > 0
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r})
        import numpy as np
        from regard3d_amd import api, synth
        api.use_developer_library()
        sc = synth.make_scene(7, 4000, "sift", seed=612)
        K = synth.intrinsics()
        c = api.Context(0)
        for i in range(sc.n_images):
            c.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); c.set_intrinsics(i, K)
        g = c.match_pairs(sc.exhaustive_pairs(), 0.6, True)
        for name, fn, kw in (("F", c.filter_F, "want_F"), ("H", c.filter_H, "want_H"), ("E", c.filter_E, "want_E")):
            out, M = fn(g, **{{kw: True}})     # raises on a violated invariant
            np.savez(sys.argv[1] + name + ".npz", pairs=np.array(out.pairs), offsets=np.array(out.offsets), matches=np.array(out.matches), models=M,
                     report=np.array(c.filter_report(), np.float64))
        print("checked")
    """)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([sys.executable, "-c", code, d + "/"], env=dict(os.environ, R3DM_FILTER_CHECK="1", R3DM_FILTER_SCOUT=scout), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "checked" in r.stdout, r.stdout[-800:] + r.stderr[-2500:]
        for name in ("F", "H", "E"):
            z = np.load(d + "/" + name + ".npz")
            for k, ref in zip(("pairs", "offsets", "matches", "models", "report"), prod[name]):
                assert np.array_equal(z[k], ref), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["F", "H"])
def test_filters_without_a_residual_bound(ctx, oracle, kind):
    """max_residual_px = inf (ACRANSAC's default precision): a contrario mode from the first model, every residual is kept and
    sorted -- the residual histogram of the sort-skipping bound then works with its clamped top bin.  Same inlier sets as the oracle."""
    sc = synth.make_scene(5, 1200, "sift", seed=909)
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], int(sc.widths[i]), int(sc.heights[i]))
    g = ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
    gp, go, gm = g.pairs, g.offsets, g.matches
    counts = np.diff(go.astype(np.int64)).astype(np.uint32)
    inf = float("inf")
    if kind == "F":
        got = ctx.filter_F(g, inf, 512, seed=77).as_dict()
        oc, om = oracle.filter_F_collection(sc.xys, sc.widths, sc.heights, gp, counts, gm, inf, 512, 77)
    else:
        got = ctx.filter_H(g, inf, 512, seed=77).as_dict()
        oc, om = oracle.filter_H_collection(sc.xys, sc.widths, sc.heights, gp, counts, gm, inf, 512, 77)
    off = 0; kept = 0
    for p, (I, J) in enumerate(gp):
        key = (int(I), int(J))
        exp = om[off:off + oc[p]]; off += oc[p]
        if oc[p]:
            assert key in got and set(map(tuple, got[key].tolist())) == set(map(tuple, exp.tolist())), key
            kept += 1
        else:
            assert key not in got
    assert kept >= (6 if kind == "F" else 0)
