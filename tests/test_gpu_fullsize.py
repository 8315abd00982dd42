"""GPU tests at BASELINE.json sizes.  C1 (10 x 2048) is compared with the oracle in full; at 8192
features per view (C2/C3 size) the oracle would take minutes per pair, so size-independent
properties are checked instead: ground-truth recall of the synthetic scene, permutation equivariance,
idempotence, spot-checked exact distances, and independence of the F filter from batching/sharding.
"""
import numpy as np
import pytest

from regard3d_amd import synth

pytestmark = pytest.mark.gpu


def _load(ctx, sc, binary=False):
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], int(sc.widths[i]), int(sc.heights[i]), binary=binary)


def test_config1_10_images_2k_sift_vs_oracle(ctx, oracle):
    sc = synth.make_scene(10, 2048, "sift", seed=1001)                 # BASELINE configs[0]
    _load(ctx, sc)
    pairs = sc.exhaustive_pairs()
    assert len(pairs) == 45
    g = ctx.match_pairs(pairs, 0.6, True)
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    keep = counts > 0
    assert np.array_equal(g.pairs, pairs[keep]) and np.array_equal(g.matches, matches)
    gf = ctx.filter_F(g)
    oc, om = oracle.filter_F_collection(sc.xys, sc.widths, sc.heights, pairs, counts, matches)
    d = gf.as_dict(); off = 0
    for p, (I, J) in enumerate(pairs):
        exp = om[off:off + oc[p]]; off += oc[p]
        got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
        assert set(map(tuple, got.tolist())) == set(map(tuple, exp.tolist()))


@pytest.mark.parametrize("kind", ["sift", "akaze"])
def test_fullsize_pair_properties(ctx, oracle, kind):
    sc = synth.make_scene(4, 8192, kind, seed=2002 if kind == "sift" else 3003)
    binary = kind == "akaze"
    _load(ctx, sc, binary)
    ratio, sq = (0.8, False) if binary else (0.6, True)
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, ratio, sq)
    d = g.as_dict()
    # (1) ground truth of the scene: neighbours share 45/30/15 % of their features
    for (I, J), m in d.items():
        wi, wj = sc.world_ids[I][m[:, 0]], sc.world_ids[J][m[:, 1]]
        true = (wi == wj) & (wi >= 0)
        shared = len(set(sc.world_ids[I][sc.world_ids[I] >= 0]) & set(sc.world_ids[J][sc.world_ids[J] >= 0]))
        assert true.mean() > 0.99                                        # precision
        assert true.sum() >= 0.98 * shared                               # recall
        assert np.all(np.diff(m[:, 0].astype(np.int64) * 2**32 + m[:, 1]) > 0)   # ordered by (i_, j_), unique
    # (2) idempotence
    g2 = ctx.match_pairs(pairs, ratio, sq)
    assert np.array_equal(g.pairs, g2.pairs) and np.array_equal(g.matches, g2.matches)
    # (3) permutation equivariance: shuffling the rows of view 0 relabels i_ and nothing else
    rng = np.random.default_rng(5)
    perm = rng.permutation(8192)
    ctx.set_image(0, sc.descs[0][perm], sc.xys[0][perm], 4000, 3000, binary=binary)
    gp = ctx.match_pairs(np.array([[0, 1]], np.uint32), ratio, sq).as_dict()[(0, 1)]
    back = np.stack([perm[gp[:, 0]], gp[:, 1]], 1)
    order = np.lexsort((back[:, 1], back[:, 0]))
    assert np.array_equal(back[order], d[(0, 1)])
    # (4) exact distances of a sample of queries against the reference arithmetic
    idx, dist = ctx.knn2(sc.descs[0], sc.descs[1][:64], binary=binary)
    for q in range(0, 64, 7):
        for k in range(2):
            exp = oracle.hamming(sc.descs[0][idx[q, k]], sc.descs[1][q]) if binary else oracle.l2sq(sc.descs[0][idx[q, k]], sc.descs[1][q])
            assert float(dist[q, k]) == float(exp)
        assert dist[q, 0] <= dist[q, 1]


def test_filter_is_independent_of_batching_and_sharding(ctx):
    """The sample stream is keyed by (seed, I, J): filtering a shard gives the same per-pair result."""
    from regard3d_amd import api, dist
    sc = synth.make_scene(7, 3000, "sift", seed=44)
    _load(ctx, sc)
    pairs = sc.exhaustive_pairs()
    g_all = ctx.match_pairs(pairs, 0.6, True)
    f_all = ctx.filter_F(g_all).as_dict()
    parts = []
    for r in range(3):
        mine = dist.shard_pairs(pairs, r, 3)
        parts.append(ctx.filter_F(ctx.match_pairs(mine, 0.6, True)))
    merged = api.Graph.merge(parts).as_dict()
    assert merged.keys() == f_all.keys()
    for k in f_all:
        assert np.array_equal(merged[k], f_all[k])
    # a different seed is a different stream (and still a valid filter)
    f_other = ctx.filter_F(g_all, seed=777).as_dict()
    assert f_other.keys() == f_all.keys()
    assert any(not np.array_equal(f_other[k], f_all[k]) for k in f_all)


def test_filters_are_deterministic_when_workgroups_share_a_cu(ctx, oracle):
    """Regression: the homography kernel (two workgroups per CU) returned different pair counts from run to run, up to memory
    faults, because the barrier at the top of its chunk loop was emitted without the LDS wait (kernels_filter.hip).  Enough
    pairs to oversubscribe the CUs, every filter several times in mixed order, identical graphs required; a sample of the H
    results is checked against the CPU restatement."""
    sc = synth.make_scene(110, 4096, "sift", seed=2002)
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000); ctx.set_intrinsics(i, synth.intrinsics())
    g = ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
    assert g.num_pairs > 400
    seen = {}
    for ch in "EHHFHEHHHFHH":
        r = {"F": ctx.filter_F, "E": ctx.filter_E, "H": ctx.filter_H}[ch](g)
        key = (r.num_pairs, r.num_matches, hash(r.matches.tobytes()))
        assert seen.setdefault(ch, key) == key, (ch, seen[ch], key)
    counts = np.diff(g.offsets.astype(np.int64)).astype(np.uint32)
    sub = np.arange(0, g.num_pairs, 9)                                 # every 9th putative pair through the oracle
    off = g.offsets.astype(np.int64)
    sp = g.pairs[sub]; scnt = counts[sub]; sm = np.concatenate([g.matches[off[k]:off[k + 1]] for k in sub])
    oh, omh = oracle.filter_H_collection(sc.xys, sc.widths, sc.heights, sp, scnt, sm, 4.0, 2048, 5489)
    d = ctx.filter_H(g).as_dict(); o = 0
    for k, (I, J) in enumerate(sp):
        exp = omh[o:o + oh[k]]; o += oh[k]
        got = d.get((int(I), int(J)), np.zeros((0, 2), np.uint32))
        assert set(map(tuple, got.tolist())) == set(map(tuple, exp.tolist())), (I, J)


def test_bench_contract_one_and_two_ranks(tmp_path):
    """bench.py as the driver launches it: N = 1 directly, N = 2 through torch.distributed.run (two ranks on the one GPU
    of this box, graphs exchanged through gloo -- the RCCL branch differs only in the tensors' device).  Checks the JSON
    contract and that the sharded run reassembles the graph the single-rank run of the same collection produces."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "1", "--feat", "1024", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--images", "17"] + common,
                         capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in j1, key
    assert j1["n_gpus"] == 1 and j1["config"]["pairs"] == 17 * 16 // 2 and j1["roofline"]["achieved"] > 0
    assert j1["opt_in_integer_mfma"]["identical_to_headline_graphs"] is True and j1["opt_in_integer_mfma"]["integer_mfma_launches"] == 1
    env = dict(os.environ, R3DM_SHARE_GPU="1", R3DM_DIST_BACKEND="gloo")
    port = 29600 + os.getpid() % 300
    # default N > 1 = STRONG scaling of the same collection (the metric's "200 img x 8k, 1/2/4/8 GPU"): 17 images, 136 pairs
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--images", "17"] + common, capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["images"] == 17 and j2["config"]["pairs"] == 136 and j2["scaling"] == "strong"
    assert 0 < j2["config"]["pairs_this_rank"] < 136
    for k in ("putative_pairs", "putative_matches", "F_pairs", "F_matches"):
        assert j2["detail"][k] == j1["detail"][k], k
    # --scaling weak: 2 ranks x C(12,2) = 132 pairs -> 17 images (136 pairs), the same collection again
    wk = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port + 1), os.path.join(root, "bench.py"),
                         "--gpus", "2", "--images", "12", "--scaling", "weak"] + common, capture_output=True, text=True, timeout=300, env=env)
    assert wk.returncode == 0, wk.stderr[-2000:]
    j3 = json.loads([l for l in wk.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j3["config"]["images"] == 17 and j3["config"]["pairs"] == 136 and j3["scaling"] == "weak"
    for k in ("putative_pairs", "putative_matches", "F_pairs", "F_matches"):
        assert j3["detail"][k] == j1["detail"][k], k


def test_bench_c5_two_ranks_reassemble_the_single_rank_graph():
    """config C5 through the rank shard: every rank builds the graph indices of its own rows of I (dropped and rebuilt inside the
    step) and searches its pairs; the gathered graphs equal the single-rank run's."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "c5", "--steps", "1", "--warmup", "1", "--images", "9", "--feat", "2048", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert j1["detail"]["ann_index_build_ms_per_step"] > 0 and j1["roofline"]["row_bytes"] == 128 and j1["roofline"]["bound"] == "valu"
    env = dict(os.environ, R3DM_SHARE_GPU="1", R3DM_DIST_BACKEND="gloo")
    port = 29900 + os.getpid() % 90
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=300, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j2["config"]["pairs"] == 36 and 0 < j2["config"]["pairs_this_rank"] < 36
    for k in ("putative_pairs", "putative_matches", "F_pairs", "F_matches"):
        assert j2["detail"][k] == j1["detail"][k], k


@pytest.mark.parametrize("config,images,feat", [("c2", 40, 512), ("c2", 6, 512), ("c5", 20, 1024), ("c5", 6, 1024)])
def test_bench_eight_ranks_reassemble_the_single_rank_graphs(config, images, feat):
    """N = 8 -- the size of the driver's scaling run -- before an 8-GPU node shows up: eight torch.distributed ranks share this
    box's GPU (R3DM_SHARE_GPU=1, graphs through gloo; the RCCL branch differs in the tensors' device only).  40 / 20 images:
    every rank owns rows of I; 6 images: five rows for eight ranks, three ranks run EMPTY shards through match, filter and the
    exchange.  The reassembled graphs must be those of the single-rank run, byte for byte (detail.graphs_sha16)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", config, "--steps", "1", "--warmup", "0", "--images", str(images), "--feat", str(feat), "--no-cpu-baseline", "--no-opt-in"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    env = dict(os.environ, R3DM_SHARE_GPU="1", R3DM_DIST_BACKEND="gloo")
    port = 28100 + os.getpid() % 900
    eight = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8"] + common,
                           capture_output=True, text=True, timeout=600, env=env)
    assert eight.returncode == 0, eight.stderr[-3000:]
    j8 = json.loads([l for l in eight.stdout.strip().splitlines() if l.startswith("{")][-1])
    n_pairs = images * (images - 1) // 2
    assert j8["n_gpus"] == 8 and j8["config"]["pairs"] == n_pairs and j8["scaling"] == "strong"
    assert j8["detail"]["graphs_sha16"] == j1["detail"]["graphs_sha16"]
    for k in ("putative_pairs", "putative_matches", "F_pairs", "F_matches"):
        assert j8["detail"][k] == j1["detail"][k], k
    assert j1["detail"]["putative_matches"] > 0


@pytest.mark.parametrize("config,extra", [("c3", ["--images", "10", "--feat", "2048"]), ("liop144", ["--images", "10", "--feat", "2048"]),
                                          ("c5", ["--images", "6", "--feat", "4096"]), ("c4", ["--images", "24", "--feat", "1024", "--emulate-world", "8"])])
def test_bench_config_legs(config, extra):
    """every BASELINE config has a bench leg with its own roofline and a CPU baseline that doubles as a parity check"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", "1", "--warmup", "1",
                        "--cpu-seconds", "0.5"] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["config"]["name"] == config and j["roofline"]["achieved"] > 0 and j["value"] > 0
    assert j["roofline"]["bound"] == {"c3": "valu", "c5": "valu"}.get(config, "mfma")
    cb = j["cpu_baseline"]
    assert cb["parity_pairs_checked"] >= 1 and cb["putative_mismatches"] == 0 and cb["F_inlier_set_mismatches"] == 0, cb
    assert j["detail"]["putative_matches"] > 0
    # the clock of SURVEY 8(d) starts at host memory: the line carries what registration costs and one measured pass from there
    assert j["detail"]["register_ms"] > 0 and 0 < j["detail"]["value_from_host"] <= j["value"] * 1.05
    if config == "c5":
        # what the config exists to report: ANN vs brute force, recall and throughput, and both bounds of the search
        d = j["detail"]
        assert 0.5 < d["recall_at_1"] <= 1.0 and 0.5 < d["recall_at_2"] <= 1.0                    # (over all queries, strangers included)
        assert 0.99 < d["recall_at_1_of_queries_with_a_match"] <= 1.0 and 0.99 < d["match_set_f1"] <= 1.0 and 0.99 < d["match_set_recall"] <= 1.0
        assert d["exhaustive_pairs_per_s"]["f32_tiles"] > 0 and d["exhaustive_pairs_per_s"]["integer_tiles_opt_in"] > 0
        assert d["exhaustive_pairs_per_s"]["integer_graphs_identical_to_f32"] is True
        assert set(j["roofline"]["bounds"]) == {"valu_issue", "gathered_bytes_over_hbm"}
    if config == "c4":
        assert "shard 0 of 8" in j["config"]["workload"] and j["config"]["pairs"] < 24 * 23 // 2
        assert cb["optimised_cpu"]["reference_built_index_mismatches"] == 0 and cb["optimised_cpu"]["reference_built_distance_mismatches"] == 0
        assert cb["optimised_cpu"]["reference_built_rows_checked"] > 0


def test_graph_exchange_over_rccl_single_rank(tmp_path):
    """The exchange of regard3d_amd/dist.py with backend "nccl" (= RCCL) and device tensors, in a one-rank group on this
    box's GPU: the same torch.distributed calls, dtypes and devices as the N > 1 bench path (the world_size-2 run of the
    same code is tests/test_dist_gloo.py on CPU)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {root!r})
        import numpy as np, torch, torch.distributed as td
        from regard3d_amd import api, dist, synth
        torch.cuda.set_device(0)
        td.init_process_group("nccl", init_method="tcp://127.0.0.1:{29700 + os.getpid() % 200}", rank=0, world_size=1,
                              device_id=torch.device("cuda", 0))
        sc = synth.make_scene(6, 700, "sift", seed=1001)
        ctx = api.Context(0)
        for i in range(sc.n_images):
            ctx.set_image(i, sc.descs[i], sc.xys[i], 4000, 3000)
        g = ctx.match_pairs(sc.exhaustive_pairs(), 0.6, True)
        gf = ctx.filter_F(g, 4.0, 2048, seed=5489)
        out = dist.all_gather_graphs([g, gf], device=torch.device("cuda", 0), force_collective=True)
        for a, b in zip((g, gf), out):
            assert np.array_equal(a.pairs, b.pairs) and np.array_equal(a.offsets, b.offsets) and np.array_equal(a.matches, b.matches)
        assert g.num_matches > 0 and gf.num_pairs > 0
        td.barrier(); td.destroy_process_group()
        print("rccl exchange ok", g.num_pairs, gf.num_pairs)
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl exchange ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_c5_graph_matcher_at_16384_rows(ctx, oracle):
    """BASELINE config C5 at its full per-view size (16,384 x 128) on a 4-view collection: the GPU graph matcher equals the CPU
    model (exact index + pool search, same start-row stream) on a sampled pair bit for bit, recovers the exhaustive matcher's
    putative matches, is idempotent, and evaluates two orders of magnitude fewer distances than brute force."""
    from regard3d_amd import api
    sc = synth.make_scene(4, 16384, "sift", seed=5005)
    _load(ctx, sc, False)
    pairs = sc.exhaustive_pairs()
    kp = api.KGraphParams.preset(3)
    g = ctx.match_pairs_kgraph(pairs, 0.6, kp)
    s = ctx.stats()
    assert s.n_ann_built == 3 and 100 < s.n_ann_dist / s.n_queries < 2000          # ~490 evaluations per query, not 16,384
    gb = ctx.match_pairs(pairs, 0.6, True)                                          # exhaustive on the same views
    d, db = g.as_dict(), gb.as_dict()
    hit = sum(len(set(map(tuple, d[k].tolist())) & set(map(tuple, db[k].tolist()))) for k in d if k in db)
    assert hit >= 0.99 * gb.num_matches and hit >= 0.999 * g.num_matches             # recall / precision of the putative matches
    g2 = ctx.match_pairs_kgraph(pairs, 0.6, kp)
    assert np.array_equal(g.pairs, g2.pairs) and np.array_equal(g.matches, g2.matches) and ctx.stats().n_ann_built == 0
    # oracle model on one pair at full size (the exact K-NN graph of a 16k view on the CPU takes a few seconds)
    sub = np.array([[0, 1]], np.uint32)
    counts, matches, _ = oracle.match_collection_kgraph(sc.descs[:2], sc.xys[:2], sub, 0.6, builder="exact", K=kp.index_K, L=kp.index_K,
                                                        cap=64, P=kp.search_P, S=kp.search_S, seed=kp.seed, min_rows=128)
    assert np.array_equal(d[(0, 1)], matches) and counts[0] == len(matches) > 1000
    ctx.clear_images()


def test_c3_binary_collection_of_24_views(ctx, oracle):
    """BASELINE config C3 as a collection: 24 views x 4,096 x 486-bit rows, 276 pairs, popcount and MFMA formulations against the
    oracle on a sample of pairs and against each other on all of them; F filter on the result."""
    sc = synth.make_scene(24, 4096, "akaze", seed=3003)
    _load(ctx, sc, True)
    pairs = sc.exhaustive_pairs()
    g = ctx.match_pairs(pairs, 0.8, False)
    ctx.set_hamming_mfma(True)
    try:
        g2 = ctx.match_pairs(pairs, 0.8, False)
    finally:
        ctx.set_hamming_mfma(False)
    assert np.array_equal(g.pairs, g2.pairs) and np.array_equal(g.offsets, g2.offsets) and np.array_equal(g.matches, g2.matches)
    sample = pairs[[0, 1, 2, 3, 22, 23, 45, 100, 200, 275]]
    counts, matches = oracle.match_collection(sc.descs, sc.xys, sample, 0.8, False, binary=True)
    d = g.as_dict(); off = 0
    for p, (I, J) in enumerate(sample):
        exp = matches[off:off + counts[p]]; off += counts[p]
        assert np.array_equal(d.get((int(I), int(J)), np.zeros((0, 2), np.uint32)), exp), (I, J)
    gf = ctx.filter_F(g)
    # images more than three steps apart share nothing: every kept pair is a neighbouring one
    assert gf.num_pairs >= 60 and all(int(J) - int(I) <= 3 for I, J in gf.pairs)
    ctx.clear_images()


def _collection_1000(feat, seed):
    """the 1000-view collections of BASELINE configs C4 / C5 as host arrays (generated on the device in blocks of 100 views)"""
    import torch
    descs, xys = [], []
    d, x, _ = synth.make_scene_torch(1000, feat, seed=seed, device="cuda", kind="sift")
    for i in range(1000):
        descs.append(d[i].cpu().numpy()); xys.append(x[i].cpu().numpy())
    del d, x
    torch.cuda.empty_cache()
    return descs, xys


def _sample_pairs(rng, n_views, n):
    pairs = {(0, 1), (0, n_views - 1), (n_views - 2, n_views - 1), (499, 500)}
    while len(pairs) < n:
        a, b = sorted(rng.choice(n_views, 2, replace=False).tolist())
        pairs.add((a, b))
    return np.array(sorted(pairs), np.uint32)


def test_c4_resident_collection_of_1000_views(oracle):
    """BASELINE config C4's collection -- 1000 views x 8,192 SIFT-128 f32, registered from host memory in one r3dm_set_images call --
    stays within 1.3 x its raw rows in HBM on the default path; pairs sampled across the whole index range match exactly as they do in
    a context that holds only their two views (nothing a view's 999 neighbours do to the table, the slabs or the ring leaks into a pair's
    result), two of them equal the CPU restatement, and the F filter keeps the overlapping ones."""
    from regard3d_amd import api
    descs, xys = _collection_1000(8192, 4004)
    raw = sum(d.nbytes for d in descs)
    c = api.Context(0)
    try:
        c.set_images(list(range(1000)), descs, xys, synth.WIDTH, synth.HEIGHT, wait=True)
        views_bytes, ring_bytes, _ = c.memory_info()
        assert raw <= views_bytes <= 1.3 * raw, (views_bytes / raw)
        assert all(c.view_info(v)[0] == 0 for v in (0, 499, 999))
        pairs = _sample_pairs(np.random.default_rng(44), 1000, 160)
        g = c.match_pairs(pairs, 0.6, True)
        assert c.memory_info()[0] <= 1.3 * raw                     # integer-valued views: the default path added no layout
        d = g.as_dict()
        assert len(d) > 0
        gf = c.filter_F(g, 4.0, 2048, seed=5489)
        assert 0 < gf.num_pairs <= g.num_pairs
        c2 = api.Context(0)
        try:
            for (I, J) in pairs[::16].tolist():
                c2.clear_images()
                c2.set_image(I, descs[I], xys[I], synth.WIDTH, synth.HEIGHT); c2.set_image(J, descs[J], xys[J], synth.WIDTH, synth.HEIGHT)
                one = c2.match_pairs(np.array([[I, J]], np.uint32), 0.6, True).as_dict()
                assert np.array_equal(one.get((I, J), np.zeros((0, 2), np.uint32)), d.get((I, J), np.zeros((0, 2), np.uint32))), (I, J)
        finally:
            c2.close()
        sub = np.array([[0, 999], [499, 500]], np.uint32)
        counts, matches = oracle.match_collection([descs[0], descs[999], descs[499], descs[500]], [xys[0], xys[999], xys[499], xys[500]],
                                                  np.array([[0, 1], [2, 3]], np.uint32), 0.6, True)
        off = 0
        for p, key in enumerate(((0, 999), (499, 500))):
            assert np.array_equal(d.get(key, np.zeros((0, 2), np.uint32)), matches[off:off + counts[p]]), key
            off += counts[p]
    finally:
        c.close()


def test_c5_resident_collection_of_1000_views_graph_matcher():
    """BASELINE config C5's collection -- 1000 views x 16,384 SIFT-128 -- resident at once: the graph matcher (index of every sampled
    first view built on first use, rows staged on first use) gives every sampled pair the graph it gives in a context holding only that
    pair's views; the views the sample never touched hold nothing beyond their tiles."""
    from regard3d_amd import api
    descs, xys = _collection_1000(16384, 5005)
    raw = sum(d.nbytes for d in descs)
    kp = api.KGraphParams.preset(3)
    c = api.Context(0)
    try:
        c.set_images(list(range(1000)), descs, xys, synth.WIDTH, synth.HEIGHT, wait=True)
        assert c.memory_info()[0] <= 1.3 * raw
        pairs = _sample_pairs(np.random.default_rng(55), 1000, 48)
        g = c.match_pairs_kgraph(pairs, 0.6, kp)
        d = g.as_dict()
        assert len(d) > 0 and c.stats().n_ann_built == len(set(pairs[:, 0].tolist()))
        used = set(pairs.reshape(-1).tolist())
        untouched = next(v for v in range(1000) if v not in used)
        assert c.view_info(untouched)[0] == 0 and c.view_info(int(pairs[0, 0]))[0] == api.LAYOUT_ROWS
        c2 = api.Context(0)
        try:
            for (I, J) in pairs[::8].tolist():
                c2.clear_images()
                c2.set_image(I, descs[I], xys[I], synth.WIDTH, synth.HEIGHT); c2.set_image(J, descs[J], xys[J], synth.WIDTH, synth.HEIGHT)
                one = c2.match_pairs_kgraph(np.array([[I, J]], np.uint32), 0.6, kp).as_dict()
                assert np.array_equal(one.get((I, J), np.zeros((0, 2), np.uint32)), d.get((I, J), np.zeros((0, 2), np.uint32))), (I, J)
        finally:
            c2.close()
    finally:
        c.close()
