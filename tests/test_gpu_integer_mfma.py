"""Opt-in integer fast path of the L2 matcher (r3dm_set_integer_mfma, include/r3dm.h): the all-pairs contraction on
v_mfma_f32_32x32x16_bf16 for integer-valued descriptors of magnitude <= 256.

Bar: BIT-EXACT -- the same 2-NN indices and float distances as the CPU restatement of the reference
(openMVG ArrayMatcherBruteForce + L2_Vectorized) and as the default f32-MFMA path; every input the proof does not
cover must keep running on the f32 tiles / the exact scan with unchanged results.
"""
import numpy as np
import pytest

from regard3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ictx(ctx):
    ctx.set_integer_mfma(True)
    yield ctx
    ctx.set_integer_mfma(False)


def _sift_like(rng, n, dim, lo=0, hi=255):
    return np.rint(np.clip(rng.gamma(0.5, 60.0, (n, dim)) + lo, lo, hi)).astype(np.float32)


@pytest.mark.parametrize("nI,nJ,dim", [(100, 70, 128), (257, 1000, 128), (2048, 2048, 128), (33, 31, 64), (1000, 777, 61),
                                       (300, 200, 100), (300, 260, 256), (2, 5, 128), (4100, 130, 128)])
def test_knn2_bit_exact_on_the_bf16_tiles(ictx, oracle, nI, nJ, dim):
    rng = np.random.default_rng(nI * 104729 + nJ)
    hi = 127 if dim > 128 else 255           # 2 * D * max^2 must stay below 2^24 for the proof (else: f32 tiles)
    a = _sift_like(rng, nI, dim, hi=hi)
    b = _sift_like(rng, nJ, dim, hi=hi)
    m = min(nJ, nI) // 2
    b[:m] = np.clip(a[:m] + np.rint(rng.normal(0, 4, (m, dim))), 0, hi)
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 1
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_knn2_signed_and_extreme_bins(ictx, oracle):
    """negative integers, the magnitude bound itself (256) and rows of all-255 bins (largest partial sums:
    2 * 128 * 255 * 255 < 2^24) stay exact"""
    rng = np.random.default_rng(8)
    a = rng.integers(-128, 128, (700, 128)).astype(np.float32)
    b = rng.integers(-128, 128, (500, 128)).astype(np.float32)
    a[0, :] = -128; b[0, :] = 127
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 1
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    a = _sift_like(rng, 300, 128); b = _sift_like(rng, 200, 128)
    a[:5] = 255; b[:3] = 255; a[5:8] = 0; b[3] = 0
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 1
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    a = rng.integers(0, 257, (400, 64)).astype(np.float32); b = rng.integers(0, 257, (300, 64)).astype(np.float32)
    a[0, 0] = 256; b[0, 0] = 256
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 1
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


def test_exact_ties_and_duplicate_rows(ictx, oracle):
    rng = np.random.default_rng(3)
    a = _sift_like(rng, 640, 128)
    a[100:200] = a[0:100]                    # duplicated dataset rows: ties between nominated and un-nominated rows
    a[300:364] = a[200]
    b = np.concatenate([a[50:150], _sift_like(rng, 100, 128)])
    idx, dist = ictx.knn2(a, b)
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


def test_u8_views_and_match_graph_equal_f32_path_and_oracle(ctx, oracle):
    sc = synth.make_scene(6, 1500, "sift", seed=77)
    descs = [d.astype(np.uint8) for d in sc.descs]                  # R3DM_U8 storage (features of type unsigned char)
    assert all(np.array_equal(d.astype(np.float32), s) for d, s in zip(descs, sc.descs))
    pairs = sc.exhaustive_pairs()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, descs[i], sc.xys[i], 4000, 3000)
    g0 = ctx.match_pairs(pairs, 0.6, True)
    assert ctx.stats().n_integer_mfma == 0
    ctx.set_integer_mfma(True)
    try:
        g1 = ctx.match_pairs(pairs, 0.6, True)
        s = ctx.stats()
    finally:
        ctx.set_integer_mfma(False)
    assert s.n_integer_mfma == 1
    assert s.n_exact_fallback <= s.n_queries // 1000
    assert np.array_equal(g0.pairs, g1.pairs) and np.array_equal(g0.offsets, g1.offsets)
    assert np.array_equal(g0.matches, g1.matches)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    d = g1.as_dict(); off = 0
    for p, (I, J) in enumerate(pairs):
        exp = matches[off:off + counts[p]]; off += counts[p]
        got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
        assert np.array_equal(got, exp)


def test_inputs_outside_the_proof_keep_the_f32_tiles(ictx, oracle):
    rng = np.random.default_rng(12)
    # real-valued (normalised) descriptors
    a = rng.gamma(0.5, 1.0, (700, 128)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.gamma(0.5, 1.0, (400, 128)).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # integers beyond the bf16-exact range
    a = rng.integers(0, 600, (500, 32)).astype(np.float32); b = rng.integers(0, 600, (300, 32)).astype(np.float32)
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # one integer view against one real-valued view
    a = _sift_like(rng, 500, 128); b = a[:300] + rng.normal(0, 0.25, (300, 128)).astype(np.float32)
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # 256-D rows of full-range bins: partial sums may pass 2^24
    a = _sift_like(rng, 300, 256); b = _sift_like(rng, 200, 256); a[0] = 255; b[0] = 255
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    # LIOP-length rows (144) have no bf16 kernel
    a = _sift_like(rng, 300, 144); b = _sift_like(rng, 200, 144)
    idx, dist = ictx.knn2(a, b)
    assert ictx.stats().n_integer_mfma == 0
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)


# ---- the collection-level edge cases of test_gpu_parity.py, repeated with the fast path switched on ------------------
def _parity_module():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("_gpu_parity_cases", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ragged_and_empty_views(ictx, oracle):
    _parity_module().test_match_ragged_and_empty_views(ictx, oracle)          # views of 37, 0 and 1 rows
    assert ictx.stats().n_integer_mfma >= 1


def test_coordinate_dedup(ictx, oracle):
    _parity_module().test_match_coordinate_dedup(ictx, oracle)
    assert ictx.stats().n_integer_mfma >= 1


def test_exact_ties_duplicates_and_random_shapes(ictx, oracle):
    P = _parity_module()
    P.test_knn2_l2_exact_ties_and_duplicates(ictx, oracle)
    P.test_knn2_random_shapes_sweep(ictx, oracle)       # a third of the trials are integer-valued: those take the bf16 tiles


def test_collection_graphs_of_every_descriptor_kind(ictx, oracle):
    P = _parity_module()
    P.test_match_collection_graph(ictx, oracle, "sift", 6, 1024)
    assert ictx.stats().n_integer_mfma >= 1
    P.test_match_collection_graph(ictx, oracle, "liop", 5, 700)               # real-valued: f32 tiles
    assert ictx.stats().n_integer_mfma == 0
    P.test_match_collection_graph(ictx, oracle, "akaze", 6, 1000)             # binary: Hamming kernel
    assert ictx.stats().n_integer_mfma == 0


def test_views_beyond_the_lds_sort_budget(ictx, oracle):
    _parity_module().test_views_beyond_the_lds_sort_budget(ictx, oracle, 20000, 1.0)


@pytest.mark.parametrize("env,kind", [({"R3DM_L2_INT_VARIANT": "7"}, "sift"), ({"R3DM_L2_INT_VARIANT": "5"}, "sift"), ({"R3DM_HAMMING_RING": "1"}, "akaze")])
def test_lds_shared_variants_of_the_developer_build_give_the_products_graph(env, kind):
    """The workgroup-shared forms of the bf16 / i8 nominators -- dataset tiles through LDS-DMA, with a barrier per tile (5) or through the
    barrier-free ring of slots with sequence words and release counters (7; measured on a par with the per-wave loads and kept in
    the developer build) -- must produce the graph the product's kernels produce: views of ragged sizes (workgroups with
    waves that hold no queries, fewer tiles than ring slots, more tiles than ring slots)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, json, hashlib; sys.path.insert(0, {root!r})
import numpy as np
from regard3d_amd import api, synth
if {bool(env)!r} and sys.argv[1] == "dev":
    api.use_developer_library()
sc = synth.make_scene(6, 2300, {kind!r}, seed=77)
for k, n in ((1, 37), (2, 200), (3, 1), (4, 1025)):
    sc.descs[k] = sc.descs[k][:n]; sc.xys[k] = sc.xys[k][:n]
binary = {kind!r} == "akaze"
c = api.Context(0)
c.set_images(list(range(6)), sc.descs, sc.xys, 4000, 3000, binary=binary)
(c.set_hamming_mfma if binary else c.set_integer_mfma)(True)
g = c.match_pairs(sc.exhaustive_pairs(), 0.8 if binary else 0.7, not binary)
s = c.stats()
print(json.dumps(dict(sha=hashlib.sha256(b"".join(np.ascontiguousarray(getattr(g, f)).tobytes() for f in ("pairs", "offsets", "matches"))).hexdigest(),
                      matches=int(g.num_matches), fast=int(s.n_hamming_mfma if binary else s.n_integer_mfma))))
"""
    clean = {k: v for k, v in os.environ.items() if not k.startswith("R3DM_")}
    out = {}
    for which, e in (("product", clean), ("dev", dict(clean, **env))):
        r = subprocess.run([sys.executable, "-c", code, which], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[which] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["product"]["fast"] >= 1 and out["dev"]["fast"] >= 1 and out["product"]["matches"] > 0
    assert out["dev"] == out["product"]
