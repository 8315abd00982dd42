"""GPU parity of the graph-based approximate matcher (config C5) against oracle/kgraph.c.

The HIP path builds a deterministic index (exact K-NN graph + reverse edges) and runs the reference's pool search in
the reference's float arithmetic, so everything is compared BIT-EXACTLY with the CPU restatement run on the same index
and start rows; against the reference's own (irreproducible) NN-descent index the bar is recall.
"""
import numpy as np
import pytest

from regard3d_amd import api, synth

pytestmark = pytest.mark.gpu


def _csr_rows(off, ids, n):
    adj = np.full((n, 64), 0xFFFFFFFF, np.uint32)
    deg = np.diff(off.astype(np.int64)).astype(np.uint32)
    for i in range(n):
        adj[i, :deg[i]] = ids[off[i]:off[i + 1]]
    return adj, deg


@pytest.mark.parametrize("n,dim,K,kind", [(1500, 128, 16, "sift"), (1000, 128, 24, "sift"), (900, 144, 24, "liop"),
                                          (700, 64, 8, "rand"), (400, 40, 32, "rand"), (130, 128, 2, "sift"),
                                          (600, 128, 20, "dups")])
def test_index_equals_the_cpu_model(ctx, oracle, n, dim, K, kind):
    rng = np.random.default_rng(n + dim + K)
    if kind == "sift":
        A = synth.make_scene(1, n, "sift", seed=n).descs[0].astype(np.float32)
    elif kind == "liop":
        A = synth.make_scene(1, n, "liop", seed=n).descs[0].astype(np.float32)
    elif kind == "dups":          # many identical rows: heavy ties and hub rows with more than 64 incoming edges
        A = np.rint(rng.uniform(0, 40, (n, dim))).astype(np.float32)
        A[100:400] = A[7]
        A[400:450] = A[8]
    else:
        A = rng.normal(0, 1, (n, dim)).astype(np.float32)
    ctx.clear_images()
    ctx.set_image(5, A)
    adj, deg = ctx.kgraph_index(5, n, K)
    g = oracle.kgraph_build_exact(A, K=K, cap=64)
    off, ids, _ = g.csr()
    eadj, edeg = _csr_rows(off, ids, n)
    assert np.array_equal(deg, edeg)
    assert np.array_equal(adj, eadj)


@pytest.mark.parametrize("nI,nJ,dim,K,P,S", [(1500, 1200, 128, 24, 10, 10), (1500, 700, 128, 20, 2, 10), (900, 901, 144, 24, 12, 16),
                                             (2000, 333, 128, 16, 6, 7), (700, 500, 64, 8, 33, 10), (400, 300, 40, 32, 61, 1)])
def test_knn2_equals_the_cpu_search_on_the_same_index(ctx, oracle, nI, nJ, dim, K, P, S):
    rng = np.random.default_rng(nI + nJ + P)
    if dim == 128:
        sc = synth.make_scene(2, max(nI, nJ), "sift", seed=nI)
        A, B = sc.descs[0][:nI].astype(np.float32), sc.descs[1][:nJ].astype(np.float32)
    elif dim == 144:
        sc = synth.make_scene(2, max(nI, nJ), "liop", seed=nI)
        A, B = sc.descs[0][:nI].astype(np.float32), sc.descs[1][:nJ].astype(np.float32)
    else:
        A = rng.normal(0, 1, (nI, dim)).astype(np.float32)
        B = (A[rng.integers(0, nI, nJ)] + rng.normal(0, 0.3, (nJ, dim))).astype(np.float32)
    kp = api.KGraphParams(index_K=K, search_P=P, search_S=S, seed=77)
    idx, dist = ctx.kgraph_knn2(A, B, kp, pair=(3, 9))
    g = oracle.kgraph_build_exact(A, K=K, cap=64)
    oidx, odist, _ = g.knn2(B, P=P, S=S, seed=77, I=3, J=9, min_rows=128)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_small_views_are_scanned(ctx, oracle):
    sc = synth.make_scene(2, 120, "sift", seed=3)
    A, B = sc.descs[0].astype(np.float32), sc.descs[1].astype(np.float32)
    idx, dist = ctx.kgraph_knn2(A, B, api.KGraphParams.preset("precise"))
    bi, bd = oracle.knn2(A, B)
    assert np.array_equal(idx, bi) and np.array_equal(dist, bd)


def _check_graph(g, pairs, counts, matches):
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


@pytest.mark.parametrize("kind,preset", [("sift", "default"), ("liop", "precise"), ("sift", "fast")])
def test_collection_equals_the_cpu_model_and_recovers_the_exhaustive_matches(ctx, oracle, kind, preset):
    sc = synth.make_scene(5, 1300, kind, seed=31)
    sc.descs[3] = sc.descs[3][:90]; sc.xys[3] = sc.xys[3][:90]         # one small view: exhaustive route, both as I and as J
    sc.descs[4] = sc.descs[4][:0]; sc.xys[4] = sc.xys[4][:0]           # one empty view: its pairs never enter the map
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    pairs = sc.exhaustive_pairs()
    kp = api.KGraphParams.preset(preset)
    g = ctx.match_pairs_kgraph(pairs, 0.6, kp)
    counts, matches, comps = oracle.match_collection_kgraph(sc.descs, sc.xys, pairs, 0.6, builder="exact", K=kp.index_K,
                                                            P=kp.search_P, S=kp.search_S, seed=kp.seed, min_rows=128)
    _check_graph(g, pairs, counts, matches)
    st = ctx.stats()
    assert st.n_ann_dist > 0 and st.n_ann_built == 3                  # views 0, 1, 2 are first views of an indexed pair
    # rows of integers 0 .. 255 are gathered from their u8 copy (same f32 values, a quarter of the bytes), real-valued ones from the f32 rows
    assert (st.n_ann_rows8 > 0) == (kind == "sift") and st.n_ann_rows8 <= st.n_match_launches and st.n_ann_rows16 == 0
    assert st.n_ann_dot8 == st.n_ann_rows8                              # ... and the query views are bytes too: integer dot products
    assert st.n_ann_dist < 0.5 * sum(len(sc.descs[i]) * len(sc.descs[j]) for i, j in pairs)
    # against the exhaustive matcher: most matches recovered, few spurious
    bc, bm = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    rec = counts.sum() / max(bc.sum(), 1)
    assert rec >= (0.5 if preset == "fast" else 0.8), rec
    # and not worse than the reference's own index builder searched the same way
    K, L, rc, P = oracle.KGRAPH_PRESETS[preset]
    nc, nm, _ = oracle.match_collection_kgraph(sc.descs, sc.xys, pairs, 0.6, builder="nndescent", K=K, L=L, recall=rc, P=P)
    assert counts.sum() >= 0.95 * nc.sum(), (counts.sum(), nc.sum(), bc.sum())


def test_index_is_rebuilt_when_the_view_or_K_changes(ctx, oracle):
    sc = synth.make_scene(2, 800, "sift", seed=5)
    ctx.clear_images()
    ctx.set_image(0, sc.descs[0], sc.xys[0]); ctx.set_image(1, sc.descs[1], sc.xys[1])
    a16, _ = ctx.kgraph_index(0, 800, 16)
    a24, _ = ctx.kgraph_index(0, 800, 24)
    assert not np.array_equal(a16, a24)
    ctx.set_image(0, sc.descs[1], sc.xys[1])                          # re-staging drops the cached index
    b24, _ = ctx.kgraph_index(0, 800, 24)
    c24, _ = ctx.kgraph_index(1, 800, 24)
    assert np.array_equal(b24, c24)


def test_kgraph_rejects_what_it_cannot_do(ctx):
    rng = np.random.default_rng(0)
    ctx.clear_images()
    ctx.set_image(0, rng.integers(0, 255, (300, 64), dtype=np.uint8), binary=True)
    ctx.set_image(1, rng.integers(0, 255, (300, 64), dtype=np.uint8), binary=True)
    with pytest.raises(api.R3dmError):
        ctx.match_pairs_kgraph(np.array([[0, 1]], np.uint32), 0.6)
    with pytest.raises(api.R3dmError):
        ctx.kgraph_knn2(rng.normal(size=(300, 30)).astype(np.float32), rng.normal(size=(10, 30)).astype(np.float32))
    with pytest.raises(api.R3dmError):
        ctx.kgraph_knn2(rng.normal(size=(300, 32)).astype(np.float32), rng.normal(size=(10, 32)).astype(np.float32),
                        api.KGraphParams(index_K=64, search_P=10, search_S=10, seed=1))


@pytest.mark.parametrize("variant", ["bytes", "negative", "real_queries"])
def test_compact_row_copies_change_nothing(ctx, variant):
    """The graph search on the compact row copies of integer-valued views -- ImgDev::ann_rows8 (integers 0 .. 255; with byte queries
    the distances are integer dot products), ann_rows16 (other integers of magnitude <= 256) -- against the developer build told to
    keep the f32 rows (R3DM_ANN_ROWS16=0): identical 2-NN indices, distances and evaluation counts."""
    import os, subprocess, sys, tempfile, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prep = {"bytes": "", "negative": "A[5, 7] = -3.0", "real_queries": "B = B + np.float32(0.25)"}[variant]
    setup = textwrap.dedent(f"""
        sc = synth.make_scene(2, 3000, "sift", seed=88)
        A, B = sc.descs[0].astype(np.float32), sc.descs[1].astype(np.float32)
        {prep}
    """)
    ns = {"synth": synth, "np": np}
    exec(setup, ns)
    A, B = ns["A"], ns["B"]
    kp = api.KGraphParams.preset("default")
    idx, dist = ctx.kgraph_knn2(A, B, kp, pair=(1, 2))
    st = ctx.stats()
    assert (st.n_ann_rows16, st.n_ann_rows8, st.n_ann_dot8) == {"bytes": (0, 1, 1), "negative": (1, 0, 0), "real_queries": (0, 1, 0)}[variant]
    code = "import sys; sys.path.insert(0, %r)\nimport numpy as np\nfrom regard3d_amd import api, synth\napi.use_developer_library()\n" % root + setup + textwrap.dedent("""
        c = api.Context(0)
        idx, dist = c.kgraph_knn2(A, B, api.KGraphParams.preset("default"), pair=(1, 2))
        s = c.stats()
        assert s.n_ann_rows16 == 0 and s.n_ann_rows8 == 0 and s.n_ann_dot8 == 0
        np.savez(sys.argv[1], idx=idx, dist=dist, evals=np.array([s.n_ann_dist]))
    """)
    n_evals = st.n_ann_dist
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([sys.executable, "-c", code, d + "/f32.npz"], env=dict(os.environ, R3DM_ANN_ROWS16="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        z = np.load(d + "/f32.npz")
    assert np.array_equal(z["idx"], idx) and np.array_equal(z["dist"], dist) and int(z["evals"][0]) == n_evals


@pytest.mark.parametrize("dim,lo,hi,expect", [(40, -20, 200, "rows16"), (64, 0, 255, "dot8"), (96, 0, 180, "dot8"), (256, 0, 255, "dot8"),
                                               (48, 0, 100, "dot8"), (72, 0, 100, "rows16"), (128, 0, 256, "rows16"),
                                               (144, 0, 255, "f32"), (136, -5, 250, "f32")])
def test_compact_rows_at_other_lengths(ctx, oracle, dim, lo, hi, expect):
    """Integer-valued descriptors of other lengths and ranges through the compact-row searches (partial 16- / 8-element blocks, the
    2^24 bound of the integer dot products at D = 256, 256 as the one bf16 integer that is not a byte) against oracle/kgraph.c.
    Lengths 132 .. 144 (LIOP's 144: nine float4 per lane) have no compact-row kernel: integer-valued views of that length must
    run on the f32 rows, not fail."""
    rng = np.random.default_rng(dim + hi)
    nI, nJ = 900, 400
    A = np.rint(rng.uniform(lo, hi, (nI, dim))).astype(np.float32)
    B = np.clip(np.rint(A[rng.integers(0, nI, nJ)] + rng.normal(0, 6, (nJ, dim))), lo, hi).astype(np.float32)
    A[0, 0], B[0, 0] = hi, hi
    A[1, 1] = lo
    kp = api.KGraphParams(index_K=16, search_P=8, search_S=10, seed=5)
    idx, dist = ctx.kgraph_knn2(A, B, kp, pair=(4, 6))
    st = ctx.stats()
    assert (st.n_ann_rows16, st.n_ann_rows8, st.n_ann_dot8) == {"rows16": (1, 0, 0), "dot8": (0, 1, 1), "f32": (0, 0, 0)}[expect]
    g = oracle.kgraph_build_exact(A, K=16, cap=64)
    oidx, odist, _ = g.knn2(B, P=8, S=10, seed=5, I=4, J=6, min_rows=128)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


def test_drop_indices_forces_a_rebuild_with_the_same_result(ctx):
    sc = synth.make_scene(4, 1200, "sift", seed=17)
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
    pairs = sc.exhaustive_pairs()
    kp = api.KGraphParams.preset("default")
    g0 = ctx.match_pairs_kgraph(pairs, 0.6, kp); built0 = ctx.stats().n_ann_built
    g1 = ctx.match_pairs_kgraph(pairs, 0.6, kp); built1 = ctx.stats().n_ann_built
    ctx.drop_indices()
    g2 = ctx.match_pairs_kgraph(pairs, 0.6, kp); built2 = ctx.stats().n_ann_built
    assert (built0, built1, built2) == (3, 0, 3)
    for g in (g1, g2):
        assert np.array_equal(g.pairs, g0.pairs) and np.array_equal(g.offsets, g0.offsets) and np.array_equal(g.matches, g0.matches)


def test_graph_index_refuses_views_beyond_its_row_bound(ctx):
    """The index is the exact K-NN graph, quadratic in the rows of a view: beyond R3DM_KGRAPH_MAX_ROWS (131,072) the library says
    so instead of indexing silently (the reference's NN-descent builder, src/thirdparty/kgraph/kgraph.cpp:703-999, is not built);
    the exhaustive matcher still serves such a view."""
    n = 131072 + 32
    rng = np.random.default_rng(3)
    big = rng.integers(0, 255, (n, 8)).astype(np.float32)
    small = big[:256].copy()
    xy = rng.uniform(0, 3000, (n, 2)).astype(np.float32)      # (distinct positions: matches at one position are one match to the reference)
    ctx.clear_images()
    ctx.set_image(0, big, xy, 4000, 3000); ctx.set_image(1, small, xy[:256], 4000, 3000)
    with pytest.raises(api.R3dmError, match="R3DM_KGRAPH_MAX_ROWS"):
        ctx.kgraph_index(0, n)
    with pytest.raises(api.R3dmError, match="R3DM_KGRAPH_MAX_ROWS"):
        ctx.match_pairs_kgraph(np.array([[0, 1]], np.uint32), 0.6, api.KGraphParams.preset("default"))
    adj, deg = ctx.kgraph_index(1, 256)                    # a view inside the bound next to it is indexed as ever
    assert deg.min() >= 1
    g = ctx.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
    assert g.num_pairs == 1 and g.num_matches >= 200      # rows 0..255 of the big view are the small view's own rows
    ctx.clear_images()
