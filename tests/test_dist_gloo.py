"""The N > 1 path on CPU: world_size-2 `gloo` run of the sharding + single all-gather reassembly
(regard3d_amd/dist.py).  The per-rank match graphs are produced by the oracle here (no GPU in this
container); the collective, packing and merge code is the production code.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_pairs_partition_and_balance():
    from regard3d_amd import dist
    n = 37
    i, j = np.triu_indices(n, k=1)
    pairs = np.stack([i, j], 1).astype(np.uint32)
    for world in (1, 2, 3, 8):
        parts = [dist.shard_pairs(pairs, r, world) for r in range(world)]
        allp = np.concatenate(parts)
        assert allp.shape[0] == pairs.shape[0]
        assert set(map(tuple, allp.tolist())) == set(map(tuple, pairs.tolist()))       # a partition
        sizes = [p.shape[0] for p in parts]
        assert max(sizes) - min(sizes) <= n                                             # snake deal balances rows
        for p in parts:                                                                  # rows of one I stay together
            for I in np.unique(p[:, 0]):
                assert (p[:, 0] == I).sum() == (pairs[:, 0] == I).sum()


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    from oracle import pyoracle as O
    from regard3d_amd import api, dist, synth
    td.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(6, 300, "sift", seed=1001)
    pairs = sc.exhaustive_pairs()
    mine = dist.shard_pairs(pairs, rank, world)
    counts, matches = O.match_collection(sc.descs, sc.xys, mine, 0.6, True)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    local = api.Graph.from_csr(mine, offs, matches)
    full = dist.all_gather_graph(local, device="cpu")
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), pairs=full.pairs, offsets=full.offsets, matches=full.matches)
    td.barrier()
    td.destroy_process_group()


def _worker8(rank, world, port, tmpdir, n_img):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as td
    from oracle import pyoracle as O
    from regard3d_amd import api, dist, synth
    td.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(n_img, 200, "sift", seed=1001)
    pairs = sc.exhaustive_pairs()
    mine = dist.shard_pairs(pairs, rank, world)
    if mine.shape[0]:
        counts, matches = O.match_collection(sc.descs, sc.xys, mine, 0.6, True)
    else:                                                 # a rank without rows of I: an empty graph goes through the same exchange
        counts, matches = np.zeros(0, np.uint32), np.zeros((0, 2), np.uint32)
    keep = counts > 0
    offs = np.concatenate([[0], np.cumsum(counts[keep])]).astype(np.uint64)
    put = api.Graph.from_csr(mine[keep], offs, matches)
    # a second graph per rank (stands in for the F-filtered one): every other kept pair
    sel = np.flatnonzero(keep)[::2]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    m2 = np.concatenate([matches[off[k]:off[k + 1]] for k in sel]) if len(sel) else np.zeros((0, 2), np.uint32)
    second = api.Graph.from_csr(mine[sel], np.concatenate([[0], np.cumsum(counts[sel])]).astype(np.uint64), m2)
    full, full2 = dist.all_gather_graphs([put, second], device="cpu")
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), pairs=full.pairs, offsets=full.offsets, matches=full.matches,
             pairs2=full2.pairs, matches2=full2.matches, mine=np.array([mine.shape[0]]))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("n_img", [6, 13])
def test_all_gather_graphs_world8_gloo_with_empty_shards(tmp_path, oracle, n_img):
    """world = 8, the size the driver's scaling run uses: 6 images give 5 rows of I, so three ranks own NO pairs and ship empty
    graphs; 13 images give every rank one or two rows.  Two graphs per rank in the one exchange, as bench.py sends them."""
    import torch.multiprocessing as mp
    from regard3d_amd import synth
    port = 23500 + (os.getpid() % 2000) + n_img
    mp.spawn(_worker8, args=(8, port, str(tmp_path), n_img), nprocs=8, join=True)
    sc = synth.make_scene(n_img, 200, "sift", seed=1001)
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    keep = counts > 0
    ranks = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(8)]
    assert sum(int(r["mine"][0]) for r in ranks) == len(pairs)
    assert sum(int(r["mine"][0]) == 0 for r in ranks) == (3 if n_img == 6 else 0)
    for r in ranks:                                       # every rank owns the whole graph, ordered by (I, J)
        assert np.array_equal(r["pairs"], pairs[keep])
        assert np.array_equal(np.diff(r["offsets"].astype(np.int64)), counts[keep])
        assert np.array_equal(r["matches"], matches)
        assert np.array_equal(r["pairs2"], ranks[0]["pairs2"]) and np.array_equal(r["matches2"], ranks[0]["matches2"])
    assert len(ranks[0]["pairs2"]) >= keep.sum() // 2 - 8


def test_all_gather_graph_world2_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    from regard3d_amd import synth
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc = synth.make_scene(6, 300, "sift", seed=1001)
    pairs = sc.exhaustive_pairs()
    counts, matches = oracle.match_collection(sc.descs, sc.xys, pairs, 0.6, True)
    keep = counts > 0
    r0 = np.load(os.path.join(tmp_path, "rank0.npz")); r1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    for r in (r0, r1):                                    # every rank owns the whole graph, ordered by (I, J)
        assert np.array_equal(r["pairs"], pairs[keep])
        assert np.array_equal(np.diff(r["offsets"].astype(np.int64)), counts[keep])
        assert np.array_equal(r["matches"], matches)
