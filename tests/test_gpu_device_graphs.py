"""Device-resident match graphs and the one collective of the path (the reference's counterpart is the std::map its
OpenMP threads fill under `omp critical`, /root/reference/src/R3DComputeMatches.cpp:465,481-487).

* r3dm_set_device_graphs: match and filter results keep, beside their host vectors, the same CSR in device memory (gather kernels of
  kernels_graph.hip); r3dm_allgather_graphs then sends them from the device -- here in a one-rank RCCL communicator (RCCL refuses the same
  device twice), against the host-packed exchange of the very same graphs.
* Two REAL ranks over RCCL whenever the box has two GPUs (the driver's scaling node): bench.py through regard3d_amd/dist.py and through
  --via-c-abi must reassemble the single-rank graphs (graphs_sha16).  Skipped, with the reason printed, on a one-GPU box."""
import json
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


def test_graphs_go_on_the_wire_from_device_memory(ctx):
    sc = synth.make_scene(9, 1400, "sift", seed=77)
    K = synth.intrinsics()
    ctx.clear_images()
    for i in range(sc.n_images):
        ctx.set_image(i, sc.descs[i], sc.xys[i], synth.WIDTH, synth.HEIGHT); ctx.set_intrinsics(i, K)
    pairs = sc.exhaustive_pairs()
    comm = api.Comm(api.Comm.unique_id(), 0, 1, 0)
    try:
        # without mirrors: the host-packed exchange
        g0 = ctx.match_pairs(pairs, 0.6, True)
        f0 = ctx.filter_F(g0)
        assert g0.on_device == -1 and f0.on_device == -1
        ref = comm.allgather_graphs([g0, f0])
        assert comm.last_device_graphs == 0 and _same(ref[0], g0) and _same(ref[1], f0)
        # with mirrors: the same graphs, the same merged result, every payload from device memory
        ctx.set_device_graphs(True)
        g1 = ctx.match_pairs(pairs, 0.6, True)
        feh, _, _ = ctx.filter_FEH(g1, "FEH")
        f1, e1, h1 = feh["F"], feh["E"], feh["H"]
        assert _same(g1, g0) and _same(f1, f0)
        assert all(x.on_device == 0 for x in (g1, f1, e1, h1))
        out = comm.allgather_graphs([g1, f1, e1, h1])
        assert comm.last_device_graphs == 4
        for a, b in zip(out, (g1, f1, e1, h1)):
            assert _same(a, b)
        assert g1.num_pairs > 10 and f1.num_pairs > 5
        # a graph built on the host (no mirror) next to mirrored ones, an empty one, and the single-filter entries
        host = api.Graph.from_csr(g1.pairs, g1.offsets, g1.matches)
        fe = ctx.filter_E(g1); fh = ctx.filter_H(g1)
        assert host.on_device == -1 and fe.on_device == 0 and _same(fe, e1) and _same(fh, h1)
        out = comm.allgather_graphs([host, fe, fh])
        assert comm.last_device_graphs == 2 and _same(out[0], g1) and _same(out[1], e1) and _same(out[2], h1)
        # the matcher in several batches (appended mirror) and the integer fast path
        ctx.set_integer_mfma(True)
        g2 = ctx.match_pairs(pairs, 0.6, True)
        ctx.set_integer_mfma(False)
        assert g2.on_device == 0 and _same(comm.allgather_graphs([g2])[0], g0)
    finally:
        ctx.set_device_graphs(False)
        ctx.clear_images()
        del comm


def test_two_real_ranks_over_rccl_reassemble_the_single_rank_graphs():
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"two-rank RCCL run needs 2 GPUs, this box has {n}: covered by the world-2 gloo test on CPU (tests/test_dist_gloo.py) "
                    "and the one-rank RCCL tests; the driver's scaling node runs this test for real")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--config", "c2", "--images", "16", "--feat", "2048", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-opt-in", "--no-stage-leg"]
    sha = {}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    sha["1"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["detail"]["graphs_sha16"]
    for world in sorted({2, min(n, 8)}):
        for name, extra in (("dist.py", []), ("c-abi", ["--via-c-abi"])):
            port = 29500 + (os.getpid() + 7 * world + len(extra)) % 400
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args + extra
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
            assert r.returncode == 0, (name, world, r.stdout[-1500:], r.stderr[-3000:])
            j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            assert j["n_gpus"] == world
            sha[f"{world}/{name}"] = j["detail"]["graphs_sha16"]
    assert len(set(sha.values())) == 1, sha
