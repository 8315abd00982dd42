"""The C++ faces of the boundary (include/r3dm_array_matcher.hpp, include/r3d_compute_matches.hpp):
compile check on CPU; on the GPU box the small C++ host program is run and compared with the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

from regard3d_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "adapter_main")
    lib = os.path.join(ROOT, "regard3d_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fopenmp", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp"), "-o", out,
                           "-L" + lib, "-lr3dm", "-Wl,-rpath," + lib])
    return out


def _write_views(oracle, d, sc):
    import ctypes
    names = []
    for i in range(sc.n_images):
        name = f"img{i:03d}"
        desc = np.ascontiguousarray(sc.descs[i], np.float32)
        assert oracle.lib().orc_save_desc(os.path.join(d, name + ".desc").encode(), ctypes.c_uint64(desc.shape[0]),
                                          ctypes.c_size_t(desc.shape[1] * 4), desc.ctypes.data_as(ctypes.c_void_p)) == 0
        xyso = np.zeros((desc.shape[0], 4), np.float32); xyso[:, :2] = sc.xys[i]; xyso[:, 2] = 1.0
        with open(os.path.join(d, name + ".feat"), "w") as f:       # full-precision text so positions round-trip exactly
            for r in xyso:
                f.write("%.9g %.9g %.9g %.9g\n" % tuple(r))
        names.append(name)
    return names


def test_cpp_headers_compile_and_link(host_exe):
    assert os.path.exists(host_exe)
    # without arguments the program only reports usage (exit code 2); it must at least load libr3dm.so
    assert subprocess.run([host_exe]).returncode == 2


def test_facade_reports_failure_without_gpu_or_files(host_exe, tmp_path):
    import torch
    r = subprocess.run([host_exe, "stage", str(tmp_path), "128", "missing_view"], capture_output=True, text=True)
    assert r.returncode == 7                                   # computeMatches() returned false, did not crash
    if not torch.cuda.is_available():
        assert "computeMatches failed" in r.stderr


@pytest.mark.gpu
def test_array_matcher_adapter_matches_oracle(host_exe, oracle, tmp_path):
    sc = synth.make_scene(2, 700, "liop", seed=17)
    names = _write_views(oracle, str(tmp_path), sc)
    out = str(tmp_path / "knn.txt")
    subprocess.check_call([host_exe, "knn", str(tmp_path / (names[0] + ".desc")), str(tmp_path / (names[1] + ".desc")), "144", out])
    got = np.loadtxt(out)
    oidx, odist = oracle.knn2(sc.descs[0], sc.descs[1])
    assert np.array_equal(got[:, 0].astype(int), np.arange(700))           # IndMatch(i_ = query row, j_ = dataset row)
    assert np.array_equal(got[:, 1].astype(int), oidx[:, 0]) and np.array_equal(got[:, 3].astype(int), oidx[:, 1])
    assert np.array_equal(got[:, 2].astype(np.float32), odist[:, 0]) and np.array_equal(got[:, 4].astype(np.float32), odist[:, 1])


@pytest.mark.gpu
def test_array_matcher_builds_once_and_searches_concurrently(host_exe, oracle, tmp_path):
    """the plugin contract's amortisation: Build stages the dataset once, each of 50 SearchNeighbours
    calls -- issued from 8 OpenMP threads like the reference's loop over J, src/R3DComputeMatches.cpp:465 -- uploads only its
    queries, on one of the pool's contexts"""
    sc = synth.make_scene(2, 1500, "sift", seed=19)
    names = _write_views(oracle, str(tmp_path), sc)
    out = str(tmp_path / "loop.txt")
    r = subprocess.run([host_exe, "loop", str(tmp_path / (names[0] + ".desc")), str(tmp_path / (names[1] + ".desc")), "128", "50", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    staged, contexts, same = map(int, r.stdout.split())
    assert staged == 1 + 50                                   # one dataset + fifty query sets: no dataset re-staging
    assert 1 <= contexts <= 4 and same == 1
    got = np.loadtxt(out)
    oidx, odist = oracle.knn2(sc.descs[0], sc.descs[1])
    assert np.array_equal(got[:, 1].astype(int), oidx[:, 0]) and np.array_equal(got[:, 3].astype(int), oidx[:, 1])
    assert np.array_equal(got[:, 2].astype(np.float32), odist[:, 0]) and np.array_equal(got[:, 4].astype(np.float32), odist[:, 1])


@pytest.mark.gpu
def test_index_api_equals_knn2_and_serves_any_context_of_the_device(ctx, oracle):
    from regard3d_amd import api
    rng = np.random.default_rng(44)
    a = np.rint(rng.uniform(0, 255, (900, 128))).astype(np.float32)
    b = np.rint(rng.uniform(0, 255, (400, 128))).astype(np.float32)
    ix = ctx.index_create(a)
    staged0 = ctx.stats().n_views_staged
    i1, d1 = ctx.index_knn2(ix, b)
    other = api.Context(0)                                      # a second context on the same device searches the same index
    i2, d2 = other.index_knn2(ix, b[:100])
    other.set_integer_mfma(True)
    i3, d3 = other.index_knn2(ix, b)
    assert other.stats().n_integer_mfma == 1
    other.close()
    oidx, odist = oracle.knn2(a, b)
    assert np.array_equal(i1, oidx) and np.array_equal(d1, odist) and np.array_equal(i3, oidx) and np.array_equal(d3, odist)
    assert np.array_equal(i2, oidx[:100]) and np.array_equal(d2, odist[:100])
    assert ctx.stats().n_views_staged == staged0 + 1            # the query set only
    ix.close()
    # binary rows and real-valued rows through the same entry points
    a8 = rng.integers(0, 256, (300, 61), dtype=np.uint8); b8 = rng.integers(0, 256, (200, 61), dtype=np.uint8)
    ix = ctx.index_create(a8, binary=True)
    i, d = ctx.index_knn2(ix, b8)
    oi, od = oracle.knn2(a8, b8, binary=True)
    assert np.array_equal(i, oi) and np.array_equal(d, od.astype(np.float32))
    ix.close()
    ar = rng.normal(0, 1, (500, 144)).astype(np.float32); br = rng.normal(0, 1, (300, 144)).astype(np.float32)
    ix = ctx.index_create(ar)
    for split in (False, True):
        ctx.set_split_mfma(split)
        i, d = ctx.index_knn2(ix, br)
        assert ctx.stats().n_split_mfma == int(split)
        oi, od = oracle.knn2(ar, br)
        assert np.array_equal(i, oi) and np.array_equal(d, od)
    ctx.set_split_mfma(False)
    ix.close()


@pytest.mark.gpu
def test_stage_facade_writes_the_reference_files(host_exe, oracle, tmp_path):
    sc = synth.make_scene(5, 900, "liop", seed=23)
    names = _write_views(oracle, str(tmp_path), sc)
    r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n_put, n_f = map(int, r.stdout.split())
    prog = [l for l in r.stderr.splitlines() if l.startswith("progress")]          # updateProgress hook, the reference's messages
    assert prog == ["progress 0.70 Find putative matches", "progress 0.80 Calculate fundamental matrix",
                    "progress 0.90 Calculate essential matrix", "progress 0.95 Calculate homography matrix"]
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    for ext in ("txt", "bin"):
        p, c, m = oracle.load_matches(str(tmp_path / f"matches.putative.{ext}"))
        assert np.array_equal(p, pairs[counts > 0]) and np.array_equal(c, counts[counts > 0]) and np.array_equal(m, matches)
    assert n_put == int((counts > 0).sum())
    p_h, c_h, m_h = oracle.load_matches(str(tmp_path / "matches.h.txt"))          # homography filter ran too
    oh, omh = oracle.filter_H_collection(sc.xys, sc.widths, sc.heights, pairs, counts, matches, 4.0, 2048, 5489)
    assert np.array_equal(p_h, pairs[oh > 0]) and np.array_equal(c_h, oh[oh > 0])
    p_e, c_e, m_e = oracle.load_matches(str(tmp_path / "matches.e.txt"))          # essential-matrix filter + overlap rule
    Ks = np.stack([synth.intrinsics()] * sc.n_images)
    oe, ome = oracle.filter_E_collection(sc.xys, sc.widths, sc.heights, Ks, pairs, counts, matches, 4.0, 2048, 5489)
    assert np.array_equal(p_e, pairs[oe > 0]) and np.array_equal(c_e, oe[oe > 0]) and (oe > 0).sum() >= 2
    # adjacency SVGs: putative always; the geometric one shows the LAST filter (H) and, like upstream, is not
    # written for an empty map (a relief scene has no dominant plane, so H may keep nothing)
    svgs = ["PutativeAdjacencyMatrix.svg"] + (["GeometricAdjacencyMatrix.svg"] if (oh > 0).any() else [])
    for svg in svgs:
        txt = open(str(tmp_path / svg)).read()
        assert txt.startswith("<?xml") and txt.count("<rect") >= 1 and txt.rstrip().endswith("</svg>")
    assert os.path.exists(str(tmp_path / "GeometricAdjacencyMatrix.svg")) == bool((oh > 0).any())
    oc, om = oracle.filter_F_collection(sc.xys, sc.widths, sc.heights, pairs, counts, matches, 4.0, 2048, 5489)
    p, c, m = oracle.load_matches(str(tmp_path / "matches.f.txt"))
    assert np.array_equal(p, pairs[oc > 0]) and np.array_equal(c, oc[oc > 0]) and n_f == int((oc > 0).sum())
    off = 0
    for k, cnt in enumerate(c):                                   # same inlier set per pair (order: ascending residual)
        seg = m[off:off + cnt]; off += cnt
        exp = om[int(oc[:np.flatnonzero(oc > 0)[k]].sum()):][:cnt]
        assert set(map(tuple, seg.tolist())) == set(map(tuple, exp.tolist()))


@pytest.mark.gpu
def test_stage_facade_kgraph_dispatch(host_exe, oracle, tmp_path):
    """matchingAlgorithm 3 = "KGraph precise" (src/R3DComputeMatches.cpp:2051-2054): putative file == the CPU model"""
    sc = synth.make_scene(4, 900, "liop", seed=29)
    names = _write_views(oracle, str(tmp_path), sc)
    env = dict(os.environ, R3DM_TEST_ALGO="3", R3DM_TEST_ARMS="requested")
    r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "matcher graph" in r.stderr
    pairs = sc.exhaustive_pairs()
    counts, matches, _ = oracle.match_collection_kgraph(sc.descs, sc.xys, pairs, 0.6, builder="exact", K=24, P=12, S=10,
                                                        seed=1998, min_rows=128)
    p, c, m = oracle.load_matches(str(tmp_path / "matches.putative.txt"))
    assert np.array_equal(p, pairs[counts > 0]) and np.array_equal(c, counts[counts > 0]) and np.array_equal(m, matches)
    # MRPT / FLANN arms (no such index here) are served by the same matcher with the preset of matching recall: 0 ("FLANN") == 3
    r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True,
                       env=dict(os.environ, R3DM_TEST_ALGO="0", R3DM_TEST_ARMS="requested"))
    assert r.returncode == 0 and "matcher graph" in r.stderr, r.stderr
    p8, c8, m8 = oracle.load_matches(str(tmp_path / "matches.putative.txt"))
    assert np.array_equal(p8, p) and np.array_equal(c8, c) and np.array_equal(m8, m)
    # arms 6 / 7 / 8 run hnsw_match itself: hnswlib's search on the batch-built index; putative file == the CPU model
    for algo, preset in ((6, "fast"), (8, "precise")):
        r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True,
                           env=dict(os.environ, R3DM_TEST_ALGO=str(algo), R3DM_TEST_ARMS="requested"))
        assert r.returncode == 0 and "matcher hnsw" in r.stderr, r.stderr
        hc, hm = oracle.match_collection_hnsw(sc.descs, sc.xys, pairs, 0.6, preset)
        ph, ch, mh = oracle.load_matches(str(tmp_path / "matches.putative.txt"))
        assert np.array_equal(ph, pairs[hc > 0]) and np.array_equal(ch, hc[hc > 0]) and np.array_equal(mh, hm)
    # the default policy serves an approximate arm with whichever matcher is faster for the views: LIOP-144 (real-valued) -> the
    # exhaustive matcher, i.e. arm 0 (the GUI's default, FLANN in the reference) writes exactly what arm 9 writes
    r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True, env=dict(os.environ, R3DM_TEST_ALGO="0"))
    assert r.returncode == 0 and "matcher exhaustive" in r.stderr, r.stderr
    pe, ce, me = oracle.load_matches(str(tmp_path / "matches.putative.txt"))
    cx, mx = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    assert np.array_equal(pe, pairs[cx > 0]) and np.array_equal(ce, cx[cx > 0]) and np.array_equal(me, mx)
    r = subprocess.run([host_exe, "stage", str(tmp_path), "144"] + names, capture_output=True, text=True,
                       env=dict(os.environ, R3DM_TEST_ALGO="11"))
    assert r.returncode == 7 and "not served" in r.stderr               # unknown arm: refused


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dim", [("sift", "128"), ("liop", "144")])
def test_stage_facade_exact_fast_paths_write_the_same_files(host_exe, oracle, tmp_path, kind, dim):
    """the facade's exact fast paths (default: on -- bf16 tiles for SIFT-like integer descriptors, split-f16 nomination for
    real-valued LIOP-144) against R3DComputeMatches::setExactFastPaths(false): byte-identical matches.putative / matches.f"""
    sc = synth.make_scene(5, 1200, kind, seed=31)
    names = _write_views(oracle, str(tmp_path), sc)
    r = subprocess.run([host_exe, "stage", str(tmp_path), dim] + names, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = {f: open(str(tmp_path / f), "rb").read() for f in ("matches.putative.bin", "matches.putative.txt", "matches.f.bin")}
    r2 = subprocess.run([host_exe, "stage", str(tmp_path), dim] + names, capture_output=True, text=True,
                        env=dict(os.environ, R3DM_TEST_F32_TILES="1"))
    assert r2.returncode == 0, r2.stderr
    assert r2.stdout == r.stdout
    for f, blob in ref.items():
        assert open(str(tmp_path / f), "rb").read() == blob, f
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    p, c, m = oracle.load_matches(str(tmp_path / "matches.putative.bin"))
    assert np.array_equal(p, pairs[counts > 0]) and np.array_equal(c, counts[counts > 0]) and np.array_equal(m, matches)


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["9", "2"])
def test_stage_facade_on_a_device_list_writes_the_same_files(host_exe, oracle, tmp_path, algo):
    """R3DComputeMatches(device_ids): the stage dealt to several contexts (r3dm_multi_*; here three on the box's one GPU) writes
    byte-identical putative / F / E / H files -- exhaustive arm 9 and the KGraph arm 2."""
    sc = synth.make_scene(6, 1000, "sift", seed=37)
    names = _write_views(oracle, str(tmp_path), sc)
    files = ("matches.putative.bin", "matches.putative.txt", "matches.f.bin", "matches.e.bin", "matches.h.bin")
    r = subprocess.run([host_exe, "stage", str(tmp_path), "128"] + names, capture_output=True, text=True, env=dict(os.environ, R3DM_TEST_ALGO=algo))
    assert r.returncode == 0, r.stderr
    ref = {f: open(str(tmp_path / f), "rb").read() for f in files}
    for f in files: os.remove(str(tmp_path / f))
    r2 = subprocess.run([host_exe, "stage", str(tmp_path), "128"] + names, capture_output=True, text=True,
                        env=dict(os.environ, R3DM_TEST_ALGO=algo, R3DM_TEST_DEVICES="3"))
    assert r2.returncode == 0, r2.stderr
    assert r2.stdout == r.stdout and int(r.stdout.split()[0]) > 3
    for f, blob in ref.items():
        assert open(str(tmp_path / f), "rb").read() == blob, f


@pytest.mark.gpu
def test_features_facade_matches_oracle(host_exe, oracle, tmp_path):
    """Regard3DFeatures::detectAndExtract (include/regard3d_features.hpp) == CPU restatement of detect + LIOP"""
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:400, 0:560]
    img = 0.5 + 0.1 * np.sin(xx / 13.0) * np.cos(yy / 9.0)
    for _ in range(40):
        cx, cy, s, a = rng.uniform(40, 520), rng.uniform(40, 360), rng.uniform(2, 9), rng.uniform(0.2, 0.4) * rng.choice([-1, 1])
        img = img + a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img = np.clip(img, 0, 1).astype(np.float32)
    raw = str(tmp_path / "img.f32"); out = str(tmp_path / "feats.txt")
    img.tofile(raw)
    # the reference's signatures, from 5 worker threads at once under initAKAZESemaphore(1) (tests/cpp/adapter_main.cpp)
    r = subprocess.run([host_exe, "features", raw, "560", "400", out, "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["5", "1", "1"]             # 5 threads, identical results, one detector call in flight at a time
    got = np.loadtxt(out, dtype=np.float64).astype(np.float32)
    okp = oracle.akaze_detect(img, 0.001)["kps"]
    odesc = oracle.liop_describe(oracle.liop_extract_patches(img, okp, 8.0))
    want = okp.copy(); want[:, 2] /= 2.0
    assert got.shape == (len(okp), 148) and len(okp) > 30
    assert np.array_equal(got[:, :4], want) and np.array_equal(got[:, 4:], odesc)
