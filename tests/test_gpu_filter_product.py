"""The AC-RANSAC filters of the PRODUCT library (libr3dm.so, no developer knobs) on a collection that mixes short pairs (one
workgroup each, acransac_kernel) with long ones (the cooperative kernel, > 4096 putatives), through r3dm_filter_FEH, against the CPU
restatement pair by pair -- OpenMVG's ACRANSAC as restated in SURVEY.md App. A.5, call site
/root/reference/src/R3DComputeMatches.cpp:2099-2233 (F, E + overlap rule, H).  Also: the same call repeated in mixed orders and from two
contexts at once gives the same bytes, and a collection WITHOUT intrinsics (E has no work item at all while F and H lead long pairs:
the case in which r3dm_filter_FEH once filed the empty E block over F's) still filters."""
import threading

import numpy as np
import pytest

from regard3d_amd import api, synth

pytestmark = pytest.mark.gpu

N_VIEWS, N_PTS = 12, 16000
W, H = 4000, 3000


def _collection(seed=2025):
    """12 views of one rigid cloud (feature k of every view = point k + 0.4 px noise), 18 pairs with 300 .. 15000 putatives of which
    ~70 % are true correspondences and the rest point at random other features"""
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-4, 4, N_PTS), rng.uniform(-3, 3, N_PTS), rng.uniform(8, 14, N_PTS)]
    f = 4800.0
    xys = []
    for v in range(N_VIEWS):
        th = 0.02 * v - 0.1
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        Y = X @ R.T + np.array([0.35 * v - 1.5, 0.03 * v, 0.05 * v])
        xy = np.c_[f * Y[:, 0] / Y[:, 2] + W / 2, f * Y[:, 1] / Y[:, 2] + H / 2] + rng.normal(0, 0.4, (N_PTS, 2))
        xys.append(xy.astype(np.float32))
    lengths = [300, 650, 1200, 1900, 2600, 3500, 4090, 4100, 4700, 5600, 6900, 8200, 9000, 10500, 12000, 13500, 15000, 7]
    pairs, counts, matches = [], [], []
    k = 0
    for i in range(N_VIEWS):
        for j in range(i + 1, N_VIEWS):
            if (i + 2 * j) % 3 != 0 or k >= len(lengths):
                continue
            m = lengths[k]; k += 1
            ii = np.sort(rng.permutation(N_PTS)[:m])
            jj = ii.copy()
            out = rng.random(m) < 0.3
            jj[out] = rng.integers(0, N_PTS, int(out.sum()))
            pairs.append((i, j)); counts.append(m); matches.append(np.c_[ii, jj])
    assert k == len(lengths)
    pairs = np.array(pairs, np.uint32); counts = np.array(counts, np.uint32)
    matches = np.concatenate(matches).astype(np.uint32)
    return xys, pairs, counts, matches


def _register(c, xys, with_K):
    c.clear_images()
    dummy = np.zeros((N_PTS, 128), np.float32)
    K = synth.intrinsics()
    for v, xy in enumerate(xys):
        c.set_image(v, dummy, xy, W, H)
        if with_K:
            c.set_intrinsics(v, K)


def _same(a, b):
    return np.array_equal(a.pairs, b.pairs) and np.array_equal(a.offsets, b.offsets) and np.array_equal(a.matches, b.matches)


def test_mixed_short_and_long_pairs_against_the_oracle(ctx, oracle):
    assert api.LIB_PATH.endswith("libr3dm.so"), "this test is about the product library"
    xys, pairs, counts, matches = _collection()
    _register(ctx, xys, True)
    g = api.Graph.from_csr(pairs, np.r_[0, np.cumsum(counts)].astype(np.uint64), matches)
    got, msk, _ = ctx.filter_FEH(g, "FEH")
    st = ctx.stats()
    assert st.n_filter_coop_pairs == 3 * int((counts > 4096).sum()), "the long pairs of all three filters run on the cooperative kernel"
    Ws, Hs = [W] * N_VIEWS, [H] * N_VIEWS
    Ks = np.stack([synth.intrinsics()] * N_VIEWS)
    exp = {"F": oracle.filter_F_collection(xys, Ws, Hs, pairs, counts, matches)[:2],
           "H": oracle.filter_H_collection(xys, Ws, Hs, pairs, counts, matches)[:2],
           "E": oracle.filter_E_collection(xys, Ws, Hs, Ks, pairs, counts, matches)[:2]}
    bad = []
    for name in "FEH":
        d = got[name].as_dict()
        oc, om = exp[name]
        off = 0
        for p, (I, J) in enumerate(pairs):
            e = om[off:off + oc[p]]; off += oc[p]
            gm = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
            if set(map(tuple, gm.tolist())) != set(map(tuple, e.tolist())):
                bad.append((name, int(I), int(J), int(counts[p]), len(gm), len(e)))
        assert sum(1 for c_ in oc if c_) >= 12, (name, oc)
    assert not bad, bad
    # the same graph in other orders and subsets, three times over: the same bytes (a persistent-worker kernel with hand-rolled hand-offs
    # must not depend on who was idle when)
    for which in ("HF", "E", "FEH", "EH", "FEH"):
        again, _, _ = ctx.filter_FEH(g, which)
        for k_ in which:
            assert _same(again[k_], got[k_]), (which, k_)
    for k_, fn in (("F", ctx.filter_F), ("E", ctx.filter_E), ("H", ctx.filter_H)):
        assert _same(fn(g), got[k_]), k_
    # two contexts of the device at once
    other = api.Context(0)
    try:
        _register(other, xys, True)
        res = {}

        def run(name, c):
            res[name] = c.filter_FEH(g, "FEH")[0]
        ts = [threading.Thread(target=run, args=("a", ctx)), threading.Thread(target=run, args=("b", other))]
        for t in ts: t.start()
        for t in ts: t.join()
        for k_ in "FEH":
            assert _same(res["a"][k_], got[k_]) and _same(res["b"][k_], got[k_]), k_
    finally:
        other.close()


def test_long_pairs_without_intrinsics(ctx):
    """No view has a K: the E filter has no work item (E_ACRobust skips pairs without valid pinhole intrinsics) while F and H lead long
    pairs on the cooperative kernel -- F + E + H, F + E and E alone must all work, E empty, F and H as with intrinsics."""
    xys, pairs, counts, matches = _collection(seed=7)
    g = api.Graph.from_csr(pairs, np.r_[0, np.cumsum(counts)].astype(np.uint64), matches)
    _register(ctx, xys, True)
    ref, _, _ = ctx.filter_FEH(g, "FH")
    _register(ctx, xys, False)
    for which in ("FEH", "FE", "EH", "E", "FH"):
        got, _, _ = ctx.filter_FEH(g, which)
        if "E" in which:
            assert got["E"].num_pairs == 0 and got["E"].num_matches == 0
        for k_ in which:
            if k_ != "E":
                assert _same(got[k_], ref[k_]), (which, k_)
    assert ctx.filter_E(g).num_pairs == 0
    # only ONE view of a long pair has intrinsics: still no E item for it
    ctx.set_intrinsics(int(pairs[-2][0]), synth.intrinsics())
    got, _, _ = ctx.filter_FEH(g, "FEH")
    assert got["E"].num_pairs == 0 and _same(got["F"], ref["F"]) and _same(got["H"], ref["H"])
    assert ref["F"].num_pairs >= 12
