#!/usr/bin/env python
"""bench.py -- image-pairs matched/sec (+ F-inlier filter) on MI355X, BASELINE.json's metric.

A "step" = one pass of the hot path over the whole pair list of the workload: 2-NN matching
(fused MFMA squared-L2 + exact re-scoring + ratio test), per-pair finalisation, the AC-RANSAC
fundamental-matrix filter, the result graphs back in host RAM and -- for N > 1 -- the single
all-gather that reassembles the pairwise match graph on every rank.  Descriptors are resident in
HBM before the timed region starts (r3dm_set_image copied and re-laid them out).

N = 1 : BASELINE.json configs[1]: 200 images x 8192 SIFT-128 f32, exhaustive 19,900 pairs.
N > 1 : weak scaling of the same job: one image collection whose exhaustive pair count is ~N x 19,900
        (283 / 400 / 565 images for N = 2 / 4 / 8), descriptors replicated on every GPU, pairs sharded
        by rows of I (regard3d_amd/dist.py), no collective on the matching data path.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (l2_knn2_mfma_kernel): achieved
TFLOP/s = algorithmic 2*nI*nJ*D flops of the launches / their HIP-event time, against the dense FP32
MFMA peak.  `cpu_baseline` times the CPU restatement (oracle/, OpenMP over J like the reference) on a
bounded sample of the same workload on the host cores of this box, and doubles as a parity check.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from regard3d_amd import api, dist as r3dist, synth

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16, dense (opt-in integer fast path only)


def images_for(n_gpus: int, base_images: int) -> int:
    if n_gpus <= 1:
        return base_images
    target = n_gpus * base_images * (base_images - 1) // 2
    return int(round((1 + math.sqrt(1 + 8 * target)) / 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=200, help="images at N=1 (BASELINE config: 200)")
    ap.add_argument("--feat", type=int, default=8192)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="rough budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the extra (untimed-region) pass with r3dm_set_integer_mfma")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1) and rank == 0:
        print(f"# note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # test hooks (CI on a 1-GPU box): R3DM_SHARE_GPU=1 maps every rank to cuda:0, R3DM_DIST_BACKEND=gloo exchanges
    # the graphs through host tensors; the defaults are one GPU per rank and RCCL
    if os.environ.get("R3DM_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("R3DM_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as td
        if backend == "nccl":
            td.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
        else:
            td.init_process_group(backend)

    n_images = images_for(world, a.images)
    descs, xys, _ = synth.make_scene_torch(n_images, a.feat, seed=2002, device=dev)
    torch.cuda.synchronize()
    ctx = api.Context(local_rank)
    for i in range(n_images):
        ctx.set_image(i, descs[i], xys[i], synth.WIDTH, synth.HEIGHT)
    ii, jj = np.triu_indices(n_images, k=1)
    pairs = np.stack([ii, jj], 1).astype(np.uint32)
    mine = r3dist.shard_pairs(pairs, rank, world)

    def step():
        g = ctx.match_pairs(mine, 0.6, True)
        s_match = ctx.stats()
        gf = ctx.filter_F(g, 4.0, 2048, seed=5489)
        s_all = ctx.stats()
        full = r3dist.all_gather_graphs([g, gf], device=xdev)
        return g, gf, full, s_match, s_all

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = 0.0; kernel_flops = 0.0; launches = 0; filter_ms = 0.0; fallback = 0; queries = 0
    wall = {"match": 0.0, "match_post": 0.0, "filter": 0.0}
    for _ in range(a.steps):
        g, gf, full, s_match, s_all = step()
        kernel_ms += s_match.ms_match_kernels; kernel_flops += s_match.algorithmic_flops
        launches += s_match.n_match_launches; filter_ms += s_all.ms_filter_kernels
        fallback += s_match.n_exact_fallback; queries += s_match.n_queries
        wall["match"] += s_all.ms_wall_match; wall["match_post"] += s_all.ms_wall_match_post; wall["filter"] += s_all.ms_wall_filter
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    total_pairs = pairs.shape[0]
    value = total_pairs * a.steps / elapsed
    achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "metric": "image-pairs matched/sec (+ F-inlier filter)",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n_images} images x {a.feat} SIFT-128 f32 descriptors, exhaustive {total_pairs} pairs, "
                               "brute-force L2 2-NN + ratio 0.6 + F-matrix AC-RANSAC (4 px, 2048 it)"
                               + ("" if world == 1 else f", pairs sharded over {world} GPUs + 1 all-gather"),
                   "images": n_images, "features_per_image": a.feat, "dim": 128, "pairs": int(total_pairs),
                   "pairs_this_rank": int(mine.shape[0]), "parallelism": f"pair-shard x{world}"},
        "roofline": {"bound": "mfma", "kernel": "l2_knn2_mfma_kernel<G=16,NJ=2>", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": None, "avg_launch_ms": kernel_ms / max(launches, 1),
                     "flops_per_launch": kernel_flops / max(launches, 1), "launches": int(launches)},
        "detail": {"filter_kernel_ms_per_step": filter_ms / a.steps, "match_kernel_ms_per_step": kernel_ms / a.steps,
                   "wall_ms_per_step": {k: v / a.steps for k, v in wall.items()},
                   "match_only_pairs_per_s_this_rank": (mine.shape[0] * a.steps / (wall["match"] * 1e-3)) if wall["match"] > 0 else None,
                   "exact_fallback_queries_per_step": fallback / a.steps, "queries_per_step": queries / a.steps,
                   "putative_pairs": int(full[0].num_pairs), "putative_matches": int(full[0].num_matches),
                   "F_pairs": int(full[1].num_pairs), "F_matches": int(full[1].num_matches)},
    }

    # HBM traffic of one launch of the dominant kernel comes from separate rocprofv3 --pmc passes of this same
    # command (a process cannot profile itself): profiles/r01_pmc_traffic.json, valid for the default workload only
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if world == 1 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("workload_pairs") == int(total_pairs) and a.feat == 8192:
            out["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes/launch (PMC FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; algorithmic %.4g)" % tj["algorithmic_bytes_per_launch"]
    # Outside the timed region, N = 1 only: the same step with the opt-in integer fast path (r3dm_set_integer_mfma:
    # bf16-exact MFMA for integer-valued descriptors, bit-identical results -- DESIGN.md section 4.9).  Reported beside
    # the headline, never as `value`: the headline stays on the f32 MFMA tiles the north star names.
    if world == 1 and not a.no_opt_in:
        ctx.set_integer_mfma(True)
        try:
            step(); fence()
            t1 = time.perf_counter()
            g2, gf2, _, sm2, sa2 = step()
            fence()
            el2 = time.perf_counter() - t1
        finally:
            ctx.set_integer_mfma(False)
        same = all(np.array_equal(getattr(x, f), getattr(y, f)) for x, y in ((g, g2), (gf, gf2)) for f in ("pairs", "offsets", "matches"))
        ach2 = sm2.algorithmic_flops / (sm2.ms_match_kernels * 1e-3) / 1e12 if sm2.ms_match_kernels > 0 else 0.0
        out["opt_in_integer_mfma"] = {
            "value": total_pairs / el2, "unit": "pairs/s", "ms_per_step": el2 * 1e3, "steps": 1,
            "identical_to_headline_graphs": bool(same), "integer_mfma_launches": int(sm2.n_integer_mfma),
            "dtype": "bf16 operands holding exact integers, f32 accumulate (exact below 2^24)",
            "roofline": {"bound": "mfma", "kernel": "l2_knn2_int_kernel<GB=8,NJ=2>", "achieved": ach2, "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach2 / BF16_MFMA_PEAK_TFLOPS, "avg_launch_ms": sm2.ms_match_kernels / max(sm2.n_match_launches, 1)},
            "filter_kernel_ms": sa2.ms_filter_kernels,
        }
    if world == 1 and os.path.exists(tpath) and "opt_in_integer_mfma" in out:
        ti = json.load(open(tpath)).get("integer_fast_path")
        if ti and json.load(open(tpath)).get("workload_pairs") == int(total_pairs) and a.feat == 8192:
            out["opt_in_integer_mfma"]["roofline"]["traffic"] = ti["traffic_bytes_per_launch"]
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(descs, xys, g, gf, a.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        td.barrier()
        td.destroy_process_group()


def cpu_baseline(descs, xys, g, gf, budget_s):
    """CPU restatement (oracle/) timed on this box's host cores on a bounded sample: image 0 against
    images 1..S (the reference's loop: I fixed, `omp parallel for schedule(dynamic)` over J,
    /root/reference/src/R3DComputeMatches.cpp:465) + the AC-RANSAC F filter of those pairs.
    The same pairs are then compared with the GPU result (parity check for free)."""
    from oracle import pyoracle as O
    cores = os.cpu_count() or 1
    n_images = descs.shape[0]
    # one pair costs ~9 s on one core (8192^2 x 128, scalar f32 like OpenMVG's L2<float>): size the sample to the budget
    S = int(min(n_images - 1, max(cores, int(cores * budget_s / 9.0))))
    hd = [descs[i].cpu().numpy() for i in range(S + 1)]
    hx = [xys[i].cpu().numpy() for i in range(S + 1)]
    sub = np.stack([np.zeros(S, np.uint32), np.arange(1, S + 1, dtype=np.uint32)], 1)
    O.build()
    t0 = time.perf_counter()
    counts, matches = O.match_collection(hd, hx, sub, 0.6, True)
    t_match = time.perf_counter() - t0
    t1 = time.perf_counter()
    W = np.full(S + 1, synth.WIDTH, np.uint32); H = np.full(S + 1, synth.HEIGHT, np.uint32)
    oc, om = O.filter_F_collection(hx, W, H, sub, counts, matches, 4.0, 2048, 5489)
    t_filter = time.perf_counter() - t1
    # parity of the sampled pairs: GPU putative graph == oracle, GPU F-inlier sets == oracle
    dg, dgf = g.as_dict(), gf.as_dict()
    bad_put = bad_f = 0; off = offf = 0
    for p in range(S):
        key = (0, p + 1)
        exp = matches[off:off + counts[p]]; off += counts[p]
        got = dg.get(key, np.zeros((0, 2), np.uint32))
        bad_put += int(not np.array_equal(got, exp))
        expf = om[offf:offf + oc[p]]; offf += oc[p]
        gotf = dgf.get(key, np.zeros((0, 2), np.uint32))
        bad_f += int(set(map(tuple, gotf.tolist())) != set(map(tuple, expf.tolist())))
    out = {"value": S / (t_match + t_filter), "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"pairs (0,1..{S}) of the same workload: brute-force L2 2-NN + ratio ({t_match:.1f} s) + "
                     f"AC-RANSAC F filter ({t_filter:.2f} s), OpenMP over J on {cores} threads",
           "parity_pairs_checked": S, "putative_mismatches": bad_put, "F_inlier_set_mismatches": bad_f}
    out.update(cpu_baseline_extras(O, hd, hx, counts, cores, S))
    return out


def cpu_baseline_extras(O, hd, hx, counts, cores, S):
    """the same restatement on ONE thread, and an "optimised CPU" figure so that the speed-up is
    not quoted against a strawman: the reference's own vendored hnswlib::BruteforceSearch + L2Space (AVX L2Sqr loop,
    oracle/_ref, built from /root/reference/src/thirdparty/hnswlib), one image pair per host thread, 2-NN only."""
    import ctypes
    extra = {}
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        try:
            t0 = time.perf_counter()
            O.match_collection(hd[:2], hx[:2], np.array([[0, 1]], np.uint32), 0.6, True)
            t1 = time.perf_counter() - t0
        finally:
            gomp.omp_set_num_threads(cores)
        extra["one_thread"] = {"value": 1.0 / t1, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"pair (0,1): brute-force L2 2-NN + ratio, {t1:.1f} s"}
    except OSError:
        pass
    if O.ref_lib() is not None:
        from concurrent.futures import ThreadPoolExecutor
        n = int(min(cores, S))

        def one(p):                                    # ctypes drops the GIL: the pairs run concurrently
            idx, dist = O.ref_knn(hd[0], hd[p + 1], 2)
            return int(np.count_nonzero(dist[:, 0] < np.float32(0.36) * dist[:, 1]))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(n) as ex:
            passed = list(ex.map(one, range(n)))
        t = time.perf_counter() - t0
        # sanity: the ratio test on the reference-built 2-NN keeps at least the matches the oracle kept (it de-duplicates)
        ok = all(passed[p] >= int(counts[p]) for p in range(n))
        extra["optimised_cpu"] = {"value": n / t, "unit": "pairs/s", "cores": n, "kind": "reference",
                                  "sample": f"pairs (0,1..{n}): hnswlib::BruteforceSearch + L2Space (AVX L2Sqr) built from the "
                                            f"reference's vendored source, one pair per thread, 2-NN only (no ratio / filter), {t:.1f} s",
                                  "consistent_with_port": bool(ok)}
    return extra


if __name__ == "__main__":
    main()
