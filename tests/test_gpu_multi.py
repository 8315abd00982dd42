"""Single-process multi-GPU entry of the C ABI (r3dm_multi_*, include/r3dm.h): one context + one host thread per device,
pairs dealt by rows of I in snake order, graphs merged in host memory.  The box has one GPU, so the two contexts share it --
the host-side logic (deal, threads, merge, model re-ordering) is exactly what an 8-GPU node runs.

Bar: the merged graphs and models are IDENTICAL to a single-context run (which the other suites hold against the oracle)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from regard3d_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return np.array_equal(a.pairs, b.pairs) and np.array_equal(a.offsets, b.offsets) and np.array_equal(a.matches, b.matches)


@pytest.mark.parametrize("n_ctx", [2, 3, 8])
def test_two_contexts_reassemble_the_single_context_graphs(ctx, oracle, n_ctx):
    sc = synth.make_scene(9, 1500, "sift", seed=404)
    pairs = sc.exhaustive_pairs()
    K = synth.intrinsics()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], synth.WIDTH, synth.HEIGHT)
        ctx.set_intrinsics(i, K)
    g1 = ctx.match_pairs(pairs, 0.6, True)
    f1, F1 = ctx.filter_F(g1, want_F=True)
    h1, H1 = ctx.filter_H(g1, want_H=True)
    e1, E1 = ctx.filter_E(g1, want_E=True)
    m = api.MultiContext([0] * n_ctx)
    try:
        assert m.num_devices == n_ctx
        for i in range(sc.n_images):
            m.set_image(i, sc.descs[i], sc.xys[i], synth.WIDTH, synth.HEIGHT)
            m.set_intrinsics(i, K)
        # a view crosses PCIe once, whatever the number of contexts; the others get it device to device
        assert m.transfer_counts() == (sc.n_images, sc.n_images * (n_ctx - 1))
        g2 = m.match_pairs(pairs, 0.6, True)
        assert _same(g1, g2)
        # every context really took its share of the rows of I (with 8 contexts and 9 views: one row each)
        owner = api.shard_owner(pairs, n_ctx)
        for k in range(n_ctx):
            assert m.device_stats(k).n_pairs == int((owner == k).sum())
        f2, F2 = m.filter_F(g2, want_F=True)
        h2, H2 = m.filter_H(g2, want_H=True)
        e2, E2 = m.filter_E(g2, want_E=True)
        assert f1.num_pairs > 3
        assert _same(f1, f2) and np.array_equal(F1, F2)
        assert _same(h1, h2) and np.array_equal(H1, H2)
        assert _same(e1, e2) and np.array_equal(E1, E2)
        # the oracle on the same inputs (putative graph)
        counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
        assert np.array_equal(g2.pairs, pairs[counts > 0]) and np.array_equal(g2.matches, matches)
        m.set_integer_mfma(True)
        g3 = m.match_pairs(pairs, 0.6, True)
        assert _same(g1, g3) and all(m.device_stats(k).n_integer_mfma >= 1 for k in range(n_ctx))
        # the graph matcher over the same deal: every context builds the indices of its own rows of I only
        kp = api.KGraphParams.preset("default")
        a1 = ctx.match_pairs_kgraph(pairs, 0.6, kp)
        a2 = m.match_pairs_kgraph(pairs, 0.6, kp)
        assert _same(a1, a2) and a1.num_matches > 0
        assert sum(m.device_stats(k).n_ann_built for k in range(n_ctx)) == sc.n_images - 1
        # ... and the HNSW matcher (hnsw_match): indices per owner of the row, graphs merged
        hp = api.HnswParams.preset("medium")
        b1 = ctx.match_pairs_hnsw(pairs, 0.6, hp)
        b2 = m.match_pairs_hnsw(pairs, 0.6, hp)
        assert _same(b1, b2) and b1.num_matches > 0
    finally:
        m.close()
        ctx.clear_images()


def test_multi_create_rejects_bad_device_lists():
    with pytest.raises(api.R3dmError):
        api.MultiContext([0, 99])
    with pytest.raises(api.R3dmError):
        api.MultiContext([])


def test_filter_refuses_views_without_an_image_size(ctx):
    """with width = height = 0 the AC normalisation is inf/NaN and the filter used to return an empty graph with
    R3DM_OK; now it is an error."""
    sc = synth.make_scene(2, 600, "sift", seed=12)
    ctx.clear_images()
    ctx.set_image(0, sc.descs[0], sc.xys[0], synth.WIDTH, synth.HEIGHT)
    ctx.set_image(1, sc.descs[1], sc.xys[1])                       # no size
    g = ctx.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
    assert g.num_matches > 20
    for f in (ctx.filter_F, ctx.filter_H, ctx.filter_E):
        with pytest.raises(api.R3dmError, match="image size"):
            f(g)
    ctx.clear_images()


def test_stray_developer_environment_variables_do_not_change_product_results(oracle, tmp_path):
    """R3DM_L2_VARIANT=9 / R3DM_L2_INT_VARIANT=9 used to select timing-only kernels whose results are
    meaningless.  The product library has no such kernels and never reads the environment."""
    sc = synth.make_scene(3, 900, "sift", seed=31)
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    np.savez(str(tmp_path / "in.npz"), **{f"d{i}": sc.descs[i] for i in range(3)}, **{f"x{i}": sc.xys[i] for i in range(3)}, pairs=pairs)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import numpy as np; from regard3d_amd import api; "
            f"z = np.load({str(tmp_path / 'in.npz')!r}); c = api.Context(0); "
            "[c.set_image(i, z[f'd{i}'], z[f'x{i}'], 4000, 3000) for i in range(3)]; "
            "g = c.match_pairs(z['pairs'], 0.6, True); c.set_integer_mfma(True); g2 = c.match_pairs(z['pairs'], 0.6, True); "
            f"np.savez({str(tmp_path / 'out.npz')!r}, p=g.pairs, m=g.matches, p2=g2.pairs, m2=g2.matches)")
    env = dict(os.environ, R3DM_L2_VARIANT="9", R3DM_L2_INT_VARIANT="9", R3DM_XCD_MAP="0", R3DM_FILTER_LPT="0", R3DM_AK_LIVE_CAP="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(str(tmp_path / "out.npz"))
    for p, m in ((z["p"], z["m"]), (z["p2"], z["m2"])):
        assert np.array_equal(p, pairs[counts > 0]) and np.array_equal(m, matches)
