"""GPU parity of the Fast-A-KAZE detector (r3dm_detect_akaze) against oracle/akaze.c: the device uses only + - * / sqrt in
float without contraction and the host library shares libm with the oracle, so keypoints, sizes, angles and responses are
compared BIT-EXACTLY.  (The oracle itself is parity-unpinned against the reference: see its header.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(h, w, seed, n_blobs=40, noise=0.01):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = 0.5 + 0.1 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
    for _ in range(n_blobs):
        mg = min(40, h // 4)
        cx, cy = rng.uniform(mg, w - mg), rng.uniform(mg, h - mg)
        s = rng.uniform(2, 12); a = rng.uniform(0.15, 0.45) * rng.choice([-1, 1])
        th = rng.uniform(0, np.pi); e = rng.uniform(1.0, 2.5)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        img = img + a * np.exp(-(u * u / (2 * s * s * e) + v * v / (2 * s * s / e)))
    img = img + rng.normal(0, noise, img.shape)
    return np.clip(img, 0, 1).astype(np.float32)


@pytest.mark.parametrize("h,w,thr,seed", [(480, 640, 0.001, 1), (757, 999, 0.001, 2), (600, 800, 0.0001, 3), (1500, 2000, 0.001, 4),
                                           (120, 160, 0.001, 5), (60, 90, 0.001, 6)])
def test_detector_equals_the_cpu_restatement(ctx, oracle, h, w, thr, seed):
    img = _scene(h, w, seed, n_blobs=max(4, h * w // 8000))
    kps, resp = ctx.detect_akaze(img, thr)
    ref = oracle.akaze_detect(img, thr)
    assert len(kps) == len(ref["kps"])
    assert np.array_equal(resp, ref["responses"])
    assert np.array_equal(kps[:, :3], ref["kps"][:, :3])
    assert np.array_equal(kps[:, 3], ref["kps"][:, 3])
    if h >= 400:
        assert len(kps) > 50


def test_detect_then_describe_is_the_reference_pipeline(ctx, oracle):
    """detectKeypoints -> extractLIOPFeatures (src/Regard3DFeatures.cpp:206-222) entirely on the GPU == the CPU restatement"""
    img = _scene(600, 800, 11)
    kps, _ = ctx.detect_akaze(img, 0.001)
    desc = ctx.extract_liop(img, kps, 8.0)
    okp = oracle.akaze_detect(img, 0.001)["kps"]
    patches = oracle.liop_extract_patches(img, okp, 8.0)
    odesc = oracle.liop_describe(patches)
    assert np.array_equal(kps, okp) and np.array_equal(desc, odesc) and len(kps) > 50


def test_many_extrema_take_the_scratch_path(ctx, oracle):
    """white noise at a tiny threshold: far more candidates per level than the 3072 live-set slots (the live set itself stays small)"""
    rng = np.random.default_rng(8)
    img = np.clip(0.5 + rng.normal(0, 0.2, (700, 900)), 0, 1).astype(np.float32)
    kps, resp = ctx.detect_akaze(img, 1e-7)
    ref = oracle.akaze_detect(img, 1e-7)
    assert len(kps) == len(ref["kps"]) and len(kps) > 4000
    assert np.array_equal(kps, ref["kps"]) and np.array_equal(resp, ref["responses"])


def test_live_set_overflow_falls_back_to_scratch(ctx, tmp_path):
    """R3DM_AK_LIVE_CAP=8 (test hook of the DEVELOPER build, libr3dm_dev.so: the product library ignores the environment) makes
    the LDS live set of the in-level pruning overflow on every level, so each level is redone with the live set in global
    scratch: the keypoints must not change."""
    import os, subprocess, sys
    rng = np.random.default_rng(8)
    img = np.clip(0.5 + rng.normal(0, 0.2, (500, 700)), 0, 1).astype(np.float32)
    kps, resp = ctx.detect_akaze(img, 1e-6)
    assert len(kps) > 2000
    np.save(str(tmp_path / "img.npy"), img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import numpy as np; from regard3d_amd import api; api.use_developer_library(); "
            f"c = api.Context(0); k, r = c.detect_akaze(np.load({str(tmp_path / 'img.npy')!r}), 1e-6); "
            f"np.save({str(tmp_path / 'k.npy')!r}, k); np.save({str(tmp_path / 'r.npy')!r}, r)")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, R3DM_AK_LIVE_CAP="8"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(str(tmp_path / "k.npy")), kps) and np.array_equal(np.load(str(tmp_path / "r.npy")), resp)


@pytest.mark.parametrize("env", [{"R3DM_AK_PRUNE": "0"}, {"R3DM_AK_LIVE_CAP": "8"}, {"R3DM_AK_PRUNE": "0", "R3DM_AK_LIVE_CAP": "8"}])
@pytest.mark.parametrize("kind", ["scene", "noise", "dense"])
def test_in_level_pruning_in_parallel_equals_the_one_wavefront_form(ctx, tmp_path, env, kind):
    """Round 6: the in-level pruning resolves the connected components of `within size of each other` side by side (lone candidates on
    the spot, components of <= 64 by a wavefront with the slots in registers, larger ones with the live set in LDS, a live set that
    outgrows LDS handing the level back).  R3DM_AK_PRUNE=0 (developer build) is the one-wavefront form of rounds 2-5, R3DM_AK_LIVE_CAP=8
    forces the hand-back: keypoints, their order and responses must not move.  noise / dense: thousands of candidates per level,
    components of hundreds of candidates."""
    import os, subprocess, sys
    rng = np.random.default_rng(21)
    if kind == "scene":
        img, thr = _scene(700, 900, 9, n_blobs=120, noise=0.02), 1e-5
    elif kind == "noise":
        img, thr = np.clip(0.5 + rng.normal(0, 0.2, (500, 700)), 0, 1).astype(np.float32), 1e-7
    else:
        # smooth noise: extrema a few pixels apart everywhere -> chains of candidates within each other's radius
        from scipy.ndimage import gaussian_filter
        img, thr = np.clip(0.5 + 4.0 * gaussian_filter(rng.normal(0, 0.2, (600, 800)), 1.2), 0, 1).astype(np.float32), 1e-8
    kps, resp = ctx.detect_akaze(img, thr)
    assert len(kps) > 1000
    np.save(str(tmp_path / "img.npy"), img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import numpy as np; from regard3d_amd import api; api.use_developer_library(); "
            f"c = api.Context(0); k, r = c.detect_akaze(np.load({str(tmp_path / 'img.npy')!r}), {thr!r}); "
            f"np.save({str(tmp_path / 'k.npy')!r}, k); np.save({str(tmp_path / 'r.npy')!r}, r)")
    r = subprocess.run([sys.executable, "-c", code], env=dict({k: v for k, v in os.environ.items() if not k.startswith("R3DM_")}, **env),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(str(tmp_path / "k.npy")), kps) and np.array_equal(np.load(str(tmp_path / "r.npy")), resp)


@pytest.mark.parametrize("env", [{"R3DM_AK_HEAD": "1"}, {"R3DM_AK_FED_MARCH": "0"}, {"R3DM_AK_FED_MARCH": "0", "R3DM_AK_FED_MULTI": "0"},
                                 {"R3DM_AK_FED_KMAX": "2", "R3DM_AK_FED_WAVES": "300"}])
def test_alternative_launch_forms_of_the_scale_space_are_bit_identical(ctx, tmp_path, env):
    """The scale space has several launch forms of the same arithmetic: FED steps riding one pass in registers (the product), one step per
    launch, four steps through LDS; the level head as four launches (the product) or as one marching pass with LDS rings
    (R3DM_AK_HEAD=1, measured slower).  The developer build selects them by environment; keypoints and responses must not move."""
    import os, subprocess, sys
    img = _scene(700, 900, 5)
    kps, resp = ctx.detect_akaze(img, 1e-5)
    assert len(kps) > 1000
    np.save(str(tmp_path / "img.npy"), img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import numpy as np; from regard3d_amd import api; api.use_developer_library(); "
            f"c = api.Context(0); k, r = c.detect_akaze(np.load({str(tmp_path / 'img.npy')!r}), 1e-5); "
            f"np.save({str(tmp_path / 'k.npy')!r}, k); np.save({str(tmp_path / 'r.npy')!r}, r)")
    r = subprocess.run([sys.executable, "-c", code], env=dict({k: v for k, v in os.environ.items() if not k.startswith("R3DM_")}, **env),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(str(tmp_path / "k.npy")), kps) and np.array_equal(np.load(str(tmp_path / "r.npy")), resp)


def test_blank_and_tiny_images(ctx):
    kps, _ = ctx.detect_akaze(np.full((300, 400), 0.3, np.float32))
    assert len(kps) == 0
    kps, _ = ctx.detect_akaze(np.random.default_rng(0).random((20, 30)).astype(np.float32))   # too small for one evolution level
    assert len(kps) == 0


def test_features_stage_work_item_writes_the_reference_files(ctx, oracle, tmp_path):
    """R3DFeaturesThread::processWorkItem after imread: BGR8 -> gray float -> Fast-AKAZE + LIOP -> <name>.feat / <name>.desc"""
    import ctypes
    rng = np.random.default_rng(21)
    g = (_scene(480, 640, 21) * 255).astype(np.uint8)
    bgr = np.stack([np.clip(g.astype(np.int32) + rng.integers(-20, 20, g.shape), 0, 255).astype(np.uint8) for _ in range(3)], axis=2)
    gray = ctx.gray_from_bgr8(bgr)
    sc = np.float32(1.0 / 255.0)
    exp = (bgr[..., 0].astype(np.float32) * sc) * np.float32(0.114) + (bgr[..., 1].astype(np.float32) * sc) * np.float32(0.587) \
        + (bgr[..., 2].astype(np.float32) * sc) * np.float32(0.299)
    assert np.array_equal(gray, exp.astype(np.float32))
    feat, desc = str(tmp_path / "img.feat"), str(tmp_path / "img.desc")
    n = ctx.extract_features_to_files(gray, feat, desc, 0.001)
    okp = oracle.akaze_detect(gray, 0.001)["kps"]
    odesc = oracle.liop_describe(oracle.liop_extract_patches(gray, okp, 8.0))
    assert n == len(okp) > 30
    xyso = np.zeros((n, 4), np.float32); cnt = ctypes.c_int(0)
    assert oracle.lib().orc_load_feat(feat.encode(), ctypes.byref(cnt), xyso.ctypes.data_as(ctypes.c_void_p), n) == 0 and cnt.value == n
    want = okp.copy(); want[:, 2] /= 2.0                               # SIOPointFeature.scale = keypoint size / 2
    assert np.allclose(xyso, want, rtol=1e-5, atol=0)                  # the text format keeps 6 significant digits
    raw = np.fromfile(desc, np.uint8)
    assert int(np.frombuffer(raw[:8].tobytes(), np.uint64)[0]) == n
    assert np.array_equal(np.frombuffer(raw[8:].tobytes(), np.float32).reshape(n, 144), odesc)


def test_features_stage_over_an_image_list_runs_images_concurrently_and_skips_existing_files(ctx, tmp_path):
    """r3dm_multi_extract_features = R3DFeaturesThread::extractFeaturesAndDescriptors: K images in flight on K contexts give
    byte-identical .feat / .desc files to the one-at-a-time work item; images whose two files exist are skipped
    (src/threads/R3DFeaturesThread.cpp:139-142)"""
    from regard3d_amd import api
    imgs = [_scene(300 + 40 * k, 400 + 30 * k, 50 + k) for k in range(7)]
    seq_dir = tmp_path / "seq"; par_dir = tmp_path / "par"; seq_dir.mkdir(); par_dir.mkdir()
    n_seq = [ctx.extract_features_to_files(im, str(seq_dir / f"i{k}.feat"), str(seq_dir / f"i{k}.desc"), 0.001) for k, im in enumerate(imgs)]
    feats = [str(par_dir / f"i{k}.feat") for k in range(7)]; descs = [str(par_dir / f"i{k}.desc") for k in range(7)]
    m = api.MultiContext([0, 0, 0, 0])
    nf, sk = m.extract_features(imgs, feats, descs, 0.001)
    assert nf.tolist() == n_seq and not sk.any() and min(n_seq) > 10
    for k in range(7):
        assert open(feats[k], "rb").read() == open(str(seq_dir / f"i{k}.feat"), "rb").read()
        assert open(descs[k], "rb").read() == open(str(seq_dir / f"i{k}.desc"), "rb").read()
    # second run: everything exists -> skipped, files untouched; remove one .desc -> only that image is recomputed
    os_mtime = [__import__("os").path.getmtime(p) for p in feats]
    __import__("os").remove(descs[3])
    nf2, sk2 = m.extract_features(imgs, feats, descs, 0.001)
    m.close()
    assert sk2.tolist() == [True, True, True, False, True, True, True] and nf2.tolist() == n_seq
    assert open(descs[3], "rb").read() == open(str(seq_dir / "i3.desc"), "rb").read()
    assert [__import__("os").path.getmtime(p) for p in feats[:3]] == os_mtime[:3]


def test_deferred_feature_files_are_the_same_files(ctx, tmp_path):
    """r3dm_set_deferred_feature_files: the batch call returns with the images computed and the fwrites on the context's writer thread;
    after r3dm_features_files_wait the files are byte for byte those of the immediate mode -- also when a second batch follows at once
    (the context joins its writer before it reuses the landing buffer) -- and an unwritable path is reported by the wait"""
    from regard3d_amd import api
    imgs = [_scene(420, 560, 70 + k) for k in range(6)]
    now = tmp_path / "now"; later = tmp_path / "later"; now.mkdir(); later.mkdir()
    def paths(d, ks): return [str(d / f"i{k}.feat") for k in ks], [str(d / f"i{k}.desc") for k in ks]
    n_now = list(ctx.extract_features_batch(imgs[:3], *paths(now, range(3)))) + list(ctx.extract_features_batch(imgs[3:], *paths(now, range(3, 6))))
    ctx.set_deferred_feature_files(True)
    try:
        n_later = list(ctx.extract_features_batch(imgs[:3], *paths(later, range(3)))) + list(ctx.extract_features_batch(imgs[3:], *paths(later, range(3, 6))))
        ctx.features_files_wait()
        assert n_later == n_now and min(n_now) > 10
        for k in range(6):
            for ext in ("feat", "desc"):
                assert open(str(later / f"i{k}.{ext}"), "rb").read() == open(str(now / f"i{k}.{ext}"), "rb").read(), (k, ext)
        ctx.extract_features_batch(imgs[:1], [str(tmp_path / "no_such_dir" / "a.feat")], [str(tmp_path / "no_such_dir" / "a.desc")])
        with pytest.raises(api.R3dmError):
            ctx.features_files_wait()
        ctx.features_files_wait()                                      # the error was consumed
    finally:
        ctx.set_deferred_feature_files(False)


def test_mldb_descriptors_equal_the_cpu_restatement_and_match_across_noise(ctx, oracle):
    """detectAndCompute(DESCRIPTOR_MLDB): 61-byte descriptors bit-equal; then the config-C3 chain detect -> MLDB -> Hamming matcher"""
    img = _scene(600, 800, 31)
    kps, desc = ctx.detect_akaze_mldb(img, 0.001)
    okp, odesc, _ = oracle.akaze_detect_mldb(img, 0.001)
    assert np.array_equal(kps, okp) and np.array_equal(desc, odesc) and len(kps) > 50
    assert np.all((desc[:, 60] >> 6) == 0)                              # 486 bits: the top two bits of byte 60 stay clear
    rng = np.random.default_rng(5)
    img2 = np.clip(img + rng.normal(0, 0.003, img.shape), 0, 1).astype(np.float32)
    kps2, desc2 = ctx.detect_akaze_mldb(img2, 0.001)
    ctx.clear_images()
    ctx.set_image(0, desc, kps[:, :2].copy(), 800, 600, binary=True)
    ctx.set_image(1, desc2, kps2[:, :2].copy(), 800, 600, binary=True)
    g = ctx.match_pairs(np.array([[0, 1]], np.uint32), 0.8, False)      # Hamming: plain ratio
    m = g.matches
    assert len(m) > 0.5 * len(kps)
    d = np.hypot(*(kps[m[:, 0], :2] - kps2[m[:, 1], :2]).T)
    assert (d < 1.5).mean() > 0.95                                      # matched descriptors belong to the same image point
