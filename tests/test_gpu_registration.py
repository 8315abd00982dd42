"""GPU tests of view registration (r3dm_set_image / r3dm_set_images): what Regions_Provider::load + Features_Provider::load are to the
reference (/root/reference/src/R3DComputeMatches.cpp:2040,2094-2095).

The registration path decides nothing about results -- it only moves rows -- so every test here holds the graphs to the CPU restatement
(oracle/) whatever the source memory (pageable numpy through the page-locked ring, page-locked host, device tensors), whatever the entry
(per view, whole collection), and checks that the layouts a path does not read are not staged (the view's footprint).
"""
import numpy as np
import pytest

from regard3d_amd import api, synth

pytestmark = pytest.mark.gpu


def _expected(pairs, counts, matches):
    off = 0
    exp = {}
    for p, (I, J) in enumerate(pairs):
        if counts[p]:
            exp[(int(I), int(J))] = matches[off:off + counts[p]]
        off += counts[p]
    return exp


def _graph_equal(g, pairs, counts, matches):
    d = g.as_dict()
    exp = _expected(pairs, counts, matches)
    assert set(d.keys()) == set(exp.keys())
    for k in exp:
        assert np.array_equal(d[k], exp[k]), f"pair {k}"


def _scene(kind, n_img=7, n_feat=900, seed=4242):
    sc = synth.make_scene(n_img, n_feat, kind, seed=seed)
    # ragged, tiny and empty views, and repeated positions in two of them (coordinate de-duplication reads the position classes
    # the registration computes on the device)
    sc.descs[1] = sc.descs[1][:37]; sc.xys[1] = sc.xys[1][:37]
    sc.descs[2] = sc.descs[2][:0]; sc.xys[2] = sc.xys[2][:0]
    sc.descs[3] = sc.descs[3][:1]; sc.xys[3] = sc.xys[3][:1]
    for im in (4, 5):
        sc.descs[im] = np.concatenate([sc.descs[im], sc.descs[im][:25]])
        sc.xys[im] = np.concatenate([sc.xys[im], sc.xys[im][:25]])
    return sc


@pytest.mark.parametrize("kind", ["sift", "liop", "akaze"])
@pytest.mark.parametrize("source", ["numpy_batch", "numpy_single", "pinned_batch", "device_batch", "device_single",
                                    "mixed_batch", "mixed_single"])
def test_every_source_and_entry_registers_the_same_collection(oracle, kind, source):
    import torch
    sc = _scene(kind)
    binary = kind == "akaze"
    ratio, squared = (0.9, False) if binary else (0.95, True)
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, ratio, squared, binary=binary)
    ids = list(range(sc.n_images))
    if source.startswith("numpy"):
        D, X = sc.descs, sc.xys
    elif source.startswith("pinned"):
        D = [torch.from_numpy(np.ascontiguousarray(d)).pin_memory() for d in sc.descs]
        X = [torch.from_numpy(np.ascontiguousarray(x)).pin_memory() for x in sc.xys]
    elif source.startswith("mixed"):
        # rows and positions of one view in different kinds of memory, changing from view to view: pageable rows + device positions,
        # device rows + pageable positions, page-locked rows + device positions
        pick = lambda a, how: (a if how == 0 else torch.from_numpy(np.ascontiguousarray(a)).pin_memory() if how == 1
                               else torch.from_numpy(np.ascontiguousarray(a)).cuda())
        D = [pick(d, (0, 2, 1)[i % 3]) for i, d in enumerate(sc.descs)]
        X = [pick(x, (2, 0, 2)[i % 3]) for i, x in enumerate(sc.xys)]
    else:
        D = [torch.from_numpy(np.ascontiguousarray(d)).cuda() for d in sc.descs]
        X = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in sc.xys]
    c = api.Context(0)
    try:
        for rep in range(2):          # the second round recycles the first one's buffers (r3dm_clear_images keeps them)
            c.clear_images()
            if source.endswith("batch"):
                c.set_images(ids, D, X, 4000, 3000, binary=binary)
            else:
                for i in ids:
                    c.set_image(i, D[i], X[i], 4000, 3000, binary=binary)
            g = c.match_pairs(pairs, ratio, squared)
            _graph_equal(g, pairs, counts, matches)
            assert g.num_matches > 0
        _, _, ring, direct = c.view_info(0)
        if source.startswith("numpy"):
            assert ring == 2 * 6 and direct == 2              # (the empty view has nothing to send through the ring)
        elif source.startswith("mixed"):
            assert ring + direct == 2 * 7 and ring > 0 and direct > 0
        else:
            assert ring == 0 and direct == 2 * 7
    finally:
        c.close()


def test_a_collection_longer_than_the_ring_and_views_of_unequal_sizes(oracle):
    # 40 views through an 8-slot ring, sizes from 0 to 3000 rows in no order: slots are reused while the DMA of earlier views runs
    rng = np.random.default_rng(7)
    base = synth.make_scene(4, 3000, "sift", seed=31)
    descs, xys = [], []
    for k in range(40):
        n = int(rng.integers(0, 3001)) if k % 7 else 0
        src = k % 4
        sel = rng.permutation(3000)[:n]
        descs.append(np.ascontiguousarray(base.descs[src][sel])); xys.append(np.ascontiguousarray(base.xys[src][sel]))
    pairs = np.array([(a, b) for a in range(40) for b in range(a + 1, 40) if (a * 7 + b) % 11 == 0], np.uint32)
    counts, matches = oracle.match_collection(descs, xys, pairs, 0.8, True)
    c = api.Context(0)
    try:
        c.set_images(list(range(40)), descs, xys, 4000, 3000)
        g = c.match_pairs(pairs, 0.8, True)
        _graph_equal(g, pairs, counts, matches)
    finally:
        c.close()


def test_layouts_are_staged_by_the_path_that_reads_them():
    sift = synth.make_scene(3, 1024, "sift", seed=5)
    liop = synth.make_scene(3, 1024, "liop", seed=6)
    pairs = sift.exhaustive_pairs()
    raw = 1024 * 128 * 4
    c = api.Context(0)
    try:
        c.set_images([0, 1, 2], sift.descs, sift.xys, 4000, 3000, wait=True)
        lay, b, _, _ = c.view_info(0)
        assert lay == 0 and b <= 1.3 * raw, (lay, b)              # f32 tiles + norms + positions: nothing else
        g0 = c.match_pairs(pairs, 0.6, True)
        lay, b, _, _ = c.view_info(0)
        assert lay == 0 and b <= 1.3 * raw, (lay, b)              # integer-valued views: the default path never re-reads a row
        c.set_integer_mfma(True)
        g1 = c.match_pairs(pairs, 0.6, True)
        c.set_integer_mfma(False)
        lay, _, _, _ = c.view_info(0)
        assert lay == api.LAYOUT_BF16
        for f in ("pairs", "offsets", "matches"):
            assert np.array_equal(getattr(g0, f), getattr(g1, f))
        # real-valued views: the default path re-scores its nominees from the row-major rows, the split switch adds its own layouts
        raw = 1024 * 144 * 4
        c.clear_images()
        c.set_images([0, 1, 2], liop.descs, liop.xys, 4000, 3000, wait=True)
        assert c.view_info(1)[0] == 0
        g0 = c.match_pairs(pairs, 0.6, True)
        assert c.view_info(1)[0] == api.LAYOUT_ROWS
        c.set_split_mfma(True)
        g1 = c.match_pairs(pairs, 0.6, True)
        c.set_split_mfma(False)
        lay = c.view_info(1)[0]
        assert lay & api.LAYOUT_ROWS and lay & (api.LAYOUT_SPLIT | api.LAYOUT_COUNTS) and not lay & api.LAYOUT_BF16
        for f in ("pairs", "offsets", "matches"):
            assert np.array_equal(getattr(g0, f), getattr(g1, f))
    finally:
        c.close()


def test_a_switch_that_is_on_at_registration_stages_its_layout_there(oracle):
    liopc = synth.make_scene(4, 800, "liopc", seed=8)
    pairs = liopc.exhaustive_pairs()
    counts, matches = oracle.match_collection(liopc.descs, liopc.xys, pairs, 0.8, True)
    c = api.Context(0)
    try:
        c.set_split_mfma(True)
        c.set_images(list(range(4)), liopc.descs, liopc.xys, 4000, 3000, wait=True)
        lay = c.view_info(2)[0]
        assert lay == api.LAYOUT_ROWS | api.LAYOUT_COUNTS
        g = c.match_pairs(pairs, 0.8, True)
        assert c.stats().n_counts_mfma > 0                        # votes x scale rows: the count tiles ran
        assert c.view_info(2)[0] == api.LAYOUT_ROWS | api.LAYOUT_COUNTS      # ... and nothing was added for it
        _graph_equal(g, pairs, counts, matches)
    finally:
        c.close()


def test_replacing_a_view_replaces_its_statistics_and_layouts(oracle):
    sift = synth.make_scene(3, 700, "sift", seed=15)
    liop = synth.make_scene(3, 700, "liop", seed=16)
    pairs = sift.exhaustive_pairs()
    c = api.Context(0)
    try:
        c.set_integer_mfma(True)
        c.set_images([0, 1, 2], sift.descs, sift.xys, 4000, 3000)
        c.match_pairs(pairs, 0.6, True)
        # the same ids, now real-valued rows of another length: the integer path must not be taken on stale statistics
        liop128 = [np.ascontiguousarray(d[:, :128]) for d in liop.descs]
        for i in range(3):
            c.set_image(i, liop128[i], liop.xys[i], 4000, 3000)
        g = c.match_pairs(pairs, 0.7, True)
        assert c.stats().n_integer_mfma == 0
        counts, matches = oracle.match_collection(liop128, liop.xys, pairs, 0.7, True)
        _graph_equal(g, pairs, counts, matches)
    finally:
        c.close()


def test_an_index_gains_the_layout_of_a_switch_that_comes_on_after_build(oracle):
    rng = np.random.default_rng(3)
    a = np.rint(np.clip(rng.gamma(0.5, 60.0, (1500, 128)), 0, 255)).astype(np.float32)
    b = np.rint(np.clip(rng.gamma(0.5, 60.0, (900, 128)), 0, 255)).astype(np.float32)
    b[:300] = np.clip(a[:300] + np.rint(rng.normal(0, 4, (300, 128))), 0, 255)
    oidx, odist = oracle.knn2(a, b)
    c = api.Context(0)
    c2 = api.Context(0)
    try:
        ix = c.index_create(a)
        idx, dist = c.index_knn2(ix, b)
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        c2.set_integer_mfma(True)                                 # another context of the device, the switch on: bf16 tiles are added to the index
        idx, dist = c2.index_knn2(ix, b)
        assert c2.stats().n_integer_mfma == 1
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        idx, dist = c.index_knn2(ix, b)                           # ... and the first context still searches it on the f32 tiles
        assert c.stats().n_integer_mfma == 0
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
        ix.close()
    finally:
        c.close(); c2.close()
