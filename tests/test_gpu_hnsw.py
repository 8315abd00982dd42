"""GPU parity of the HNSW plugin path (matchingAlgorithm 6 / 7 / 8) -- kernels_hnsw.hip through the C ABI.

  (i)   SEARCH against the reference-built library: tests/golden/hnsw_ref_index.npz holds indices written by the reference's own
        hnswlib (tools/make_golden_hnsw.py) and its searchKnn(ef, 2) results; r3dm_hnsw_knn2_on_index must return the same rows and
        distances BIT FOR BIT (heap tie order and AVX summation order included).
  (ii)  BUILD against its CPU model (oracle/hnsw.c: orc_hnsw_build_batch): every link list, entry point and level, bit-exact.
  (iii) the two together against the model's search, and the recall bar: at the reference's ef the batch-built index finds the true
        nearest row at least as often as the reference-built one.
  (iv)  r3dm_match_pairs_hnsw: the match graph equals the ratio test / de-duplication applied to the model's neighbours.
"""
import numpy as np
import pytest

from regard3d_amd import api, synth
from test_oracle_hnsw import PRESETS, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["sift", "liop"])
@pytest.mark.parametrize("preset", PRESETS)
def test_search_on_the_reference_built_index_is_hnswlibs(ctx, oracle, scene, preset):
    d0, d1, ix, idx, dist = load_case(scene, preset)
    M, _, ef = oracle.HNSW_PRESETS[preset]
    gi, gd = ctx.hnsw_knn2_on_index(d0, ix, M, d1, ef)
    assert np.array_equal(gi, idx)
    assert np.array_equal(gd.view(np.uint32), dist.view(np.uint32))


@pytest.mark.parametrize("ef", [2, 3, 40, 200])
def test_search_other_beams_equal_the_restatement(ctx, oracle, ef):
    """ef the presets do not use (searchKnn clamps ef below k = 2; a wide beam grows the candidate heap)"""
    d0, d1, ix, _, _ = load_case("sift", "medium")
    M = oracle.HNSW_PRESETS["medium"][0]
    ei, ed = oracle.hnsw_from_arrays(d0, M, ix).knn2(d1, ef)
    gi, gd = ctx.hnsw_knn2_on_index(d0, ix, M, d1, ef)
    assert np.array_equal(gi, ei)
    assert np.array_equal(gd.view(np.uint32), ed.view(np.uint32))


def test_malformed_index_arrays_are_refused(ctx, oracle):
    d0, d1, ix, _, _ = load_case("sift", "fast")
    M = oracle.HNSW_PRESETS["fast"][0]
    bad = dict(ix); bad["links0"] = ix["links0"].copy(); bad["links0"][5, 1] = len(d0)            # a link past the last row
    with pytest.raises(api.R3dmError):
        ctx.hnsw_knn2_on_index(d0, bad, M, d1, 5)
    bad = dict(ix); bad["enterpoint"] = 0 if ix["levels"][0] < ix["maxlevel"] else 1                # an entry point that does not own the top layer
    with pytest.raises(api.R3dmError):
        ctx.hnsw_knn2_on_index(d0, bad, M, d1, 5)


def _scene(n, dim, kind, seed):
    if kind in ("sift", "liop"):
        sc = synth.make_scene(2, n, kind, seed=seed)
        return sc.descs[0].astype(np.float32), sc.descs[1].astype(np.float32)
    rng = np.random.default_rng(seed)
    if kind == "dups":            # many identical rows: equal distances everywhere
        A = np.rint(rng.uniform(0, 40, (n, dim))).astype(np.float32)
        A[100:300] = A[7]
        B = A[rng.integers(0, n, n)].copy()
        return A, B
    A = rng.normal(0, 1, (n, dim)).astype(np.float32)
    B = (A[rng.integers(0, n, n)] + rng.normal(0, 0.3, (n, dim))).astype(np.float32)
    return A, B


@pytest.mark.parametrize("n,dim,kind,preset", [(2000, 128, "sift", "fast"), (2000, 128, "sift", "medium"), (2000, 128, "sift", "precise"),
                                               (1200, 144, "liop", "medium"), (1200, 144, "liop", "precise"), (900, 64, "rand", "medium"),
                                               (700, 256, "rand", "precise"), (600, 128, "dups", "medium"), (130, 128, "sift", "fast")])
def test_batch_built_index_equals_the_cpu_model(ctx, oracle, n, dim, kind, preset):
    A, B = _scene(n, dim, kind, seed=n + dim)
    M, _, ef = oracle.HNSW_PRESETS[preset]
    ctx.clear_images()
    ctx.set_image(3, A)
    got = ctx.hnsw_index(3, n, api.HnswParams.preset(preset))
    model = oracle.hnsw_build_batch(A, M)
    want = model.export()
    assert np.array_equal(got["levels"], want["levels"])
    assert (got["enterpoint"], got["maxlevel"]) == (want["enterpoint"], want["maxlevel"])
    assert np.array_equal(got["up_off"], want["up_off"])
    assert np.array_equal(got["links0"], want["links0"])
    assert np.array_equal(got["up_links"], want["up_links"])
    # ... and the search on it
    ei, ed = model.knn2(B, ef)
    gi, gd = ctx.hnsw_knn2(A, B, api.HnswParams.preset(preset))
    assert np.array_equal(gi, ei)
    assert np.array_equal(gd.view(np.uint32), ed.view(np.uint32))


@pytest.mark.parametrize("scene", ["sift", "liop"])
@pytest.mark.parametrize("preset", PRESETS)
def test_recall_at_least_the_reference_built_index(ctx, oracle, scene, preset):
    d0, d1, ix, idx, dist = load_case(scene, preset)
    D = ((d1[:, None, :].astype(np.float64) - d0[None, :, :]) ** 2).sum(-1)
    o = np.argsort(D, 1)[:, :2]
    gi, gd = ctx.hnsw_knn2(d0, d1, api.HnswParams.preset(preset))
    for k in (0, 1):
        assert (gi[:, k] == o[:, k]).mean() >= (idx[:, k] == o[:, k]).mean()


@pytest.mark.parametrize("preset", ["fast", "precise"])
def test_match_pairs_hnsw_equals_model_neighbours_plus_ratio_rules(ctx, oracle, preset):
    sc = synth.make_scene(4, 900, "sift", seed=611)
    small = synth.make_scene(1, 60, "sift", seed=612)                 # a view below 128 rows: scanned exactly
    descs = [d.astype(np.float32) for d in sc.descs] + [small.descs[0].astype(np.float32)]
    xys = list(sc.xys) + [small.xys[0]]
    ctx.clear_images()
    for v, (d, xy) in enumerate(zip(descs, xys)):
        ctx.set_image(v, d, xy)
    pairs = np.array([(i, j) for i in range(5) for j in range(i + 1, 5)], np.uint32)
    hp = api.HnswParams.preset(preset)
    g = ctx.match_pairs_hnsw(pairs, 0.8, hp)
    st = ctx.stats()
    assert st.n_hnsw_launches >= 1 and st.n_ann_built == 4
    counts, matches = oracle.match_collection_hnsw(descs, xys, pairs, 0.8, preset)
    keep = counts > 0
    assert np.array_equal(g.pairs, pairs[keep])
    assert np.array_equal(np.diff(g.offsets.astype(np.int64)), counts[keep])
    assert np.array_equal(g.matches, matches)
    # the same call again is served by the cached indices
    g2 = ctx.match_pairs_hnsw(pairs, 0.8, hp)
    assert ctx.stats().n_ann_built == 0 and np.array_equal(g2.matches, matches)


@pytest.mark.parametrize("scene", ["sift", "liop"])
def test_recall_at_8k_rows_at_least_the_reference_built_index(ctx, oracle, scene):
    """VERDICT r2 item 7 (i) at the size of BASELINE's views; the reference-built rows come from tests/golden/hnsw_ref_recall_8k.npz"""
    from test_oracle_hnsw import load_case_8k
    d0, d1, exact, ref = load_case_8k(scene)
    for preset in PRESETS:
        gi, _ = ctx.hnsw_knn2(d0, d1, api.HnswParams.preset(preset))
        for k in (0, 1):
            ours, theirs = (gi[:, k] == exact[:, k]).mean(), (ref[preset][:, k] == exact[:, k]).mean()
            print(scene, preset, "recall@%d" % (k + 1), "batch-built %.4f" % ours, "reference-built %.4f" % theirs)
            assert ours >= theirs


@pytest.mark.parametrize("nq", [1, 7, 9, 63])
def test_search_ragged_query_counts(ctx, oracle, nq):
    """four queries share a workgroup (a wavefront each): counts that leave wavefronts of the last workgroup empty"""
    d0, d1, ix, _, _ = load_case("sift", "precise")
    M, _, ef = oracle.HNSW_PRESETS["precise"]
    ei, ed = oracle.hnsw_from_arrays(d0, M, ix).knn2(d1[:nq], ef)
    gi, gd = ctx.hnsw_knn2_on_index(d0, ix, M, d1[:nq], ef)
    assert np.array_equal(gi, ei)
    assert np.array_equal(gd.view(np.uint32), ed.view(np.uint32))


@pytest.mark.parametrize("knob", ["R3DM_HNSW_DENSE_STEPS=1", "R3DM_HNSW_QW=1", "R3DM_HNSW_QW=2", "R3DM_HNSW_QW=4", "R3DM_HNSW_QW=8"])
def test_search_kernel_variants_agree(ctx, oracle, tmp_path, knob):
    """developer build: R3DM_HNSW_DENSE_STEPS=1 measures every link of a hop eight links at a time, as round 3 did (the product packs
    the unvisited links first); R3DM_HNSW_QW = queries per wavefront (1: a wavefront per query; 2 / 4 / 8: a group of 32 / 16 / 8 lanes
    per query).  Every variant returns the product's rows and distances and counts the same distance evaluations"""
    import os, subprocess, sys
    d0, d1, ix, _, _ = load_case("liop", "precise")
    M, _, ef = oracle.HNSW_PRESETS["precise"]
    gi, gd = ctx.hnsw_knn2_on_index(d0, ix, M, d1, ef)
    evals = ctx.stats().n_ann_dist
    np.savez(str(tmp_path / "case.npz"), d0=d0, d1=d1, **{"ix_" + k: v for k, v in ix.items()})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import numpy as np; from regard3d_amd import api; api.use_developer_library(); "
            f"z = np.load({str(tmp_path / 'case.npz')!r}); ix = {{k[3:]: z[k] for k in z.files if k.startswith('ix_')}}; "
            f"c = api.Context(0); i, d = c.hnsw_knn2_on_index(z['d0'], ix, {M}, z['d1'], {ef}); "
            f"np.save({str(tmp_path / 'i.npy')!r}, i); np.save({str(tmp_path / 'd.npy')!r}, d); print(c.stats().n_ann_dist)")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **dict([knob.split("=")])), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(str(tmp_path / "i.npy")), gi)
    assert np.array_equal(np.load(str(tmp_path / "d.npy")).view(np.uint32), gd.view(np.uint32))
    assert int(r.stdout.strip().splitlines()[-1]) == evals
