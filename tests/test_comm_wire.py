"""The exchange of the one-process-per-GPU route inside the product library (regard3d_amd/csrc/api_comm.cpp, include/r3dm.h):
the wire format on CPU (r3dm_graphs_pack / r3dm_graphs_unpack_merge: any transport), the RCCL entry on the GPU box."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from regard3d_amd import api, dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(rng, pairs, lo, hi):
    pairs = np.asarray(pairs, np.uint32).reshape(-1, 2)
    counts = rng.integers(lo, hi, len(pairs))
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    m = rng.integers(0, 5000, (int(counts.sum()), 2)).astype(np.uint32)
    return api.Graph.from_csr(pairs, offs, m)


def _same(a, b):
    return all(np.array_equal(getattr(a, f), getattr(b, f)) for f in ("pairs", "offsets", "matches"))


def test_wire_format_round_trip_and_merge_of_ranks():
    rng = np.random.default_rng(5)
    i, j = np.triu_indices(9, k=1)
    allp = np.stack([i, j], 1).astype(np.uint32)
    for world in (1, 2, 3, 8):
        owner = api.shard_owner(allp, world)
        # two graphs per rank, as bench.py ships them (putative + filtered); some ranks own nothing
        ranks = [[_graph(rng, allp[owner == r], 1, 40), _graph(rng, allp[owner == r][::2], 1, 9)] for r in range(world)]
        words = [api.graphs_pack(gs) for gs in ranks]
        merged = api.graphs_unpack_merge(words, 2)
        for k in range(2):
            assert _same(merged[k], api.Graph.merge([gs[k] for gs in ranks]))
        assert np.array_equal(merged[0].pairs, allp)                # ordered by (I, J), every pair once
        # the python harness's own format is the same words
        for r in range(world):
            head = np.array([2] + [dist._pack(g).size for g in ranks[r]], np.uint32)
            assert np.array_equal(words[r], np.concatenate([head] + [dist._pack(g) for g in ranks[r]]))


def test_malformed_buffers_are_refused():
    rng = np.random.default_rng(6)
    g = _graph(rng, [[0, 1], [0, 2]], 1, 5)
    w = api.graphs_pack([g])
    for bad in (w[:-1], np.r_[w, 7].astype(np.uint32), np.r_[np.uint32(2), w[1:]], np.zeros(0, np.uint32)):
        with pytest.raises(api.R3dmError):
            api.graphs_unpack_merge([bad], 1)
    with pytest.raises(api.R3dmError):
        api.graphs_unpack_merge([w], 2)                             # a rank that ships fewer graphs than the caller expects


@pytest.mark.gpu
def test_rccl_exchange_in_a_one_rank_communicator():
    """RCCL refuses a device twice in one communicator, so the 1-GPU box runs the entry with world 1: packing, both ncclAllGathers on
    device buffers, unpacking, merge."""
    rng = np.random.default_rng(7)
    i, j = np.triu_indices(30, k=1)
    allp = np.stack([i, j], 1).astype(np.uint32)
    g0, g1, empty = _graph(rng, allp, 0, 300), _graph(rng, allp[::3], 1, 50), _graph(rng, np.zeros((0, 2)), 0, 1)
    comm = api.Comm(api.Comm.unique_id(), 0, 1, 0)
    assert comm.rank == 0 and comm.world == 1
    for rep in range(3):
        out = comm.allgather_graphs([g0, g1, empty])
        assert _same(out[0], g0) and _same(out[1], g1) and out[2].num_pairs == 0
    assert comm.allgather_graphs([]) == []


@pytest.mark.gpu
def test_bench_via_c_abi_reassembles_the_same_graphs():
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = {}
    for name, extra in (("torch", []), ("c_abi", ["--via-c-abi"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--images", "24", "--feat", "1024", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline", "--no-opt-in", "--no-stage-leg"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])      # (RCCL prints a version banner on stdout)
    assert out["torch"]["detail"]["graphs_sha16"] == out["c_abi"]["detail"]["graphs_sha16"]
    assert "r3dm_allgather_graphs" in out["c_abi"]["detail"]["exchange"]
