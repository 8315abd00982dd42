"""The HNSW plugin path (matchingAlgorithm 6..8): oracle/hnsw.c against the reference-built hnswlib.

tests/golden/hnsw_ref_index.npz holds, for the three presets of src/R3DComputeMatches.cpp:533-565 and two seeded scenes, what
the reference's own HierarchicalNSW (src/thirdparty/hnswlib/hnswlib/hnswalg.h, single-thread insertion in row order) produced:
levels, every link list, entry point and searchKnn(ef, 2).  The restatement must reproduce ALL of it bit for bit -- that is what
lets the GPU search kernel be tested against it anywhere.  tools/make_golden_hnsw.py wrote the fixture.
"""
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hnsw_ref_index.npz")
PRESETS = ["fast", "medium", "precise"]


def load_case(scene, preset):
    from regard3d_amd import synth
    g = np.load(GOLD)
    n, seed = (int(v) for v in g[f"{scene}_scene"])
    sc = synth.make_scene(2, n, scene, seed=seed)
    d0, d1 = np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32)
    assert [zlib.crc32(d0.tobytes()), zlib.crc32(d1.tobytes())] == g[f"{scene}_crc"].tolist(), "synth.make_scene drifted"
    p = f"{scene}_{preset}_"
    ix = dict(levels=g[p + "levels"].astype(np.int32), links0=g[p + "links0"].astype(np.int32), up_off=g[p + "up_off"],
              up_links=g[p + "up_links"].astype(np.int32), enterpoint=int(g[p + "entry"][0]), maxlevel=int(g[p + "entry"][1]))
    return d0, d1, ix, g[p + "idx"].astype(np.int32), g[p + "dist"]


def test_levels_are_the_minstd_stream(oracle):
    """getRandomLevel (hnswalg.h:146-151): -log(U) / log(M) with minstd_rand0(100) and generate_canonical<double, 53>"""
    for preset in PRESETS:
        M = oracle.HNSW_PRESETS[preset][0]
        _, _, ix, _, _ = load_case("sift", preset)
        assert np.array_equal(oracle.hnsw_levels(len(ix["levels"]), M), ix["levels"])
    lv = oracle.hnsw_levels(100000, 16)
    frac = [(lv >= k).mean() for k in (1, 2)]
    assert abs(frac[0] - 1 / 16) < 0.004 and abs(frac[1] - 1 / 256) < 0.001


@pytest.mark.parametrize("scene", ["sift", "liop"])
@pytest.mark.parametrize("preset", PRESETS)
def test_build_equals_reference_built_index(oracle, scene, preset):
    d0, d1, ix, idx, dist = load_case(scene, preset)
    M, efc, ef = oracle.HNSW_PRESETS[preset]
    mine = oracle.hnsw_build(d0, M, efc)
    ex = mine.export()
    for k in ("levels", "links0", "up_off", "up_links"):
        assert np.array_equal(ex[k], ix[k]), k
    assert (ex["enterpoint"], ex["maxlevel"]) == (ix["enterpoint"], ix["maxlevel"])
    mi, md = mine.knn2(d1, ef)
    assert np.array_equal(mi, idx)
    assert np.array_equal(md.view(np.uint32), dist.view(np.uint32))


@pytest.mark.parametrize("scene", ["sift", "liop"])
@pytest.mark.parametrize("preset", PRESETS)
def test_search_on_the_exported_reference_index(oracle, scene, preset):
    """searchKnn restated, run on the arrays the reference wrote (no build of ours involved)"""
    d0, d1, ix, idx, dist = load_case(scene, preset)
    M, _, ef = oracle.HNSW_PRESETS[preset]
    g = oracle.hnsw_from_arrays(d0, M, ix)
    mi, md = g.knn2(d1, ef)
    assert np.array_equal(mi, idx)
    assert np.array_equal(md.view(np.uint32), dist.view(np.uint32))
    # what the approximate search costs: recall of the first neighbour against the exact scan
    exact = ((d1[:, None, :].astype(np.float64) - d0[None, :, :]) ** 2).sum(-1).argmin(1) if len(d0) <= 1200 else None
    if exact is not None:
        assert (mi[:, 0] == exact).mean() > {"fast": 0.5, "medium": 0.85, "precise": 0.9}[preset]


def test_live_reference_if_built(oracle):
    """in the authoring container the reference library itself is asked again (another seed than the fixture's)"""
    if oracle.ref_lib() is None or not hasattr(oracle.ref_lib(), "ref_hnsw_export"):
        pytest.skip("oracle/_ref/libref_hnsw.so not built (needs /root/reference)")
    from regard3d_amd import synth
    sc = synth.make_scene(2, 700, "sift", seed=4242)
    d0, d1 = sc.descs[0].astype(np.float32), sc.descs[1].astype(np.float32)
    for preset in PRESETS:
        M, efc, ef = oracle.HNSW_PRESETS[preset]
        ix, idx, dist = oracle.ref_hnsw_export(d0, d1, M, efc, ef)
        mine = oracle.hnsw_build(d0, M, efc)
        ex = mine.export()
        assert all(np.array_equal(ex[k], ix[k]) for k in ("levels", "links0", "up_off", "up_links"))
        mi, md = mine.knn2(d1, ef)
        assert np.array_equal(mi, idx) and np.array_equal(md.view(np.uint32), dist.view(np.uint32))


@pytest.mark.parametrize("scene", ["sift", "liop"])
@pytest.mark.parametrize("preset", PRESETS)
def test_batch_built_index_recall_at_least_the_reference_index(oracle, scene, preset):
    """the index the HIP path builds (orc_hnsw_build_batch: exact candidates + hnswlib's heuristic, one pass) searched with hnswlib's
    searchKnn at the reference's ef finds the true nearest row at least as often as the reference-built index does"""
    d0, d1, ix, idx, dist = load_case(scene, preset)
    M, _, ef = oracle.HNSW_PRESETS[preset]
    D = ((d1[:, None, :].astype(np.float64) - d0[None, :, :]) ** 2).sum(-1)
    o = np.argsort(D, 1)[:, :2]
    g = oracle.hnsw_build_batch(d0, M)
    ex = g.export()
    assert np.array_equal(ex["levels"], ix["levels"]) and ex["maxlevel"] == ix["maxlevel"]          # the same level draw
    assert ex["links0"][:, 0].max() <= 2 * M and (ex["up_links"][:, 0].max() if len(ex["up_links"]) else 0) <= M
    mi, md = g.knn2(d1, ef)
    for k in (0, 1):
        ours, ref = (mi[:, k] == o[:, k]).mean(), (idx[:, k] == o[:, k]).mean()
        assert ours >= ref, (k, ours, ref)


GOLD8 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hnsw_ref_recall_8k.npz")


def load_case_8k(scene):
    """two views x 8,192 rows (the size of BASELINE's views) + the exact 2-NN and the reference-built HNSW's rows per preset"""
    from regard3d_amd import synth
    g = np.load(GOLD8)
    n, seed = (int(v) for v in g[f"{scene}_scene"])
    sc = synth.make_scene(2, n, scene, seed=seed)
    d0, d1 = np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32)
    assert [zlib.crc32(d0.tobytes()), zlib.crc32(d1.tobytes())] == g[f"{scene}_crc"].tolist(), "synth.make_scene drifted"
    return d0, d1, g[f"{scene}_exact"].astype(np.int32), {p: g[f"{scene}_{p}_idx"].astype(np.int32) for p in PRESETS}


@pytest.mark.parametrize("scene", ["sift", "liop"])
def test_batch_built_index_recall_at_8k_rows(oracle, scene):
    """recall >= the reference-built HierarchicalNSW per preset at 8k rows"""
    d0, d1, exact, ref = load_case_8k(scene)
    for preset in PRESETS:
        M, _, ef = oracle.HNSW_PRESETS[preset]
        mi, _ = oracle.hnsw_build_batch(d0, M).knn2(d1, ef)
        for k in (0, 1):
            ours, theirs = (mi[:, k] == exact[:, k]).mean(), (ref[preset][:, k] == exact[:, k]).mean()
            assert ours >= theirs, (preset, k, ours, theirs)
