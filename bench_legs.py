"""bench_legs.py -- the legs of bench.py OUTSIDE its timed region: roofline arithmetic, the counter-traffic lookup, the opt-in fast paths
re-run beside the headline, and the stage leg (pixels -> matches.* through the facade).  bench.py keeps the contract: arguments, the
timed steps between barriers, the JSON line, and every call into oracle/ (the `cpu_baseline` objects) -- nothing in this module imports
or calls the CPU restatement.  Imported by bench.py only."""
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from regard3d_amd import api, synth

# /opt/skills/guides/MI355X_MICROARCH.md
FP32_MFMA_PEAK_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_bf16, dense (opt-in paths only)
HBM_PEAK_GBS = 8000.0
# The roof of the popcount Hamming kernel (2 lane-ops per 32-bit word: v_xor_b32, v_bcnt_u32_b32 with its accumulate).  SURVEY 8d priced
# it at 78.6 T lane-op/s = 256 CU x 128 lanes/clk x 2.4 GHz, which is the rate of PACKED f32 (v_pk_fma_f32: two values per lane) -- the
# 157.3 TFLOP/s vector peak.  Issue rates by instruction, tools/ubench/valu_issue.hip on this chip (profiles/r04_ubench_valu_issue.txt;
# s_memtime = the 2.4 GHz shader clock; one wavefront alone: 4 cycles per wave64 instruction for all of them):
#     whole chip, wave64 instructions per SIMD and ns:  v_xor_b32 0.99   v_fma_f32 0.97   v_pk_fma_f32 0.55   v_bcnt_u32_b32 0.57
#     the alternating pair v_xor_b32 + v_bcnt_u32_b32:  0.64  = 41.7-42.2 T lane-op/s
# v_bcnt_u32_b32 (VOP3) does not get the second issue slot v_xor_b32 / v_fma_f32 get from other wavefronts; the pair the kernel is made
# of sustains 42.2 T lane-op/s in registers, with nothing else to do.  That is the roof; the stated 78.6 T stays in the line beside it.
VALU_LANE_OPS_STATED_T = 78.6      # SURVEY 8d's figure (packed-f32 lane rate)
VALU_LANE_OPS_PEAK_T = 42.2        # measured ceiling of the v_xor_b32 + v_bcnt_u32_b32 pair (profiles/r04_ubench_valu_issue.txt)
HAMMING_ALGORITHMIC_VALU_SHARE = 256.0 / 284.0   # hamming_knn2_kernel<16,4> inner loop: 128 xor + 128 bcnt of 284 VALU instructions (ISA)



def stage_main(a, embed=None, cpu_baseline_fn=None):
    """--config stage: the reference's default Compute-matches stage from pixels, through the one facade call.
    embed = {"images", "steps", "warmup"}: the timed steps only, returned as a dict (the `stage_leg` object of the default bench line)."""
    import shutil
    import tempfile
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--config stage is a single-GPU leg (the features and filter phases deal to devices inside the facade)")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    W, H = (int(x) for x in a.stage_size.lower().split("x"))
    N = (embed["images"] if embed else a.images) or 32
    n_steps = embed["steps"] if embed else a.steps
    n_warm = embed["warmup"] if embed else max(a.warmup, 1)
    imgs, K = synth.make_photo_set(N, H, W, seed=7007, device=dev)          # gray / 255 floats, resident in HBM
    torch.cuda.synchronize()
    views = [dict(id=k, width=W, height=H, basename=f"img{k:04d}", gray=imgs[k], focal_px=K[0, 0], ppx=K[0, 2], ppy=K[1, 2]) for k in range(N)]
    bare = [dict(v, gray=None) for v in views]
    n_pairs = N * (N - 1) // 2
    d = tempfile.mkdtemp(prefix="r3dm_stage_")
    conc, batch = (int(x) for x in a.stage_features.lower().split("x"))

    def wipe(all_files=True):
        for f in os.listdir(d):
            if all_files or f.startswith("matches."):
                os.remove(os.path.join(d, f))

    stage = api.Stage([0])             # the facade object of a long-lived host: contexts and work buffers survive between steps

    timed = [0.0]

    def step(algo=9):
        wipe()                                  # bench housekeeping (deleting the previous step's 400 MB of files): outside the step's clock
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = stage.run(d, views, 0.001, 0.6, algo, True, True, True, 5489, conc, batch, background_nice=True)
        torch.cuda.synchronize()
        timed[0] += time.perf_counter() - t
        return r

    try:
        for _ in range(n_warm):
            step()
        torch.cuda.synchronize()
        timed[0] = 0.0
        t0 = time.perf_counter()
        reps = [step().as_dict() for _ in range(n_steps)]
        torch.cuda.synchronize()
        housekeeping = time.perf_counter() - t0 - timed[0]
        elapsed = timed[0]                      # the n_steps facade calls, each bracketed by a device synchronisation
        mean = lambda k: sum(r[k] for r in reps) / len(reps)
        last = reps[-1]
        if a.stage_quick or embed:
            quick = {"stage_features": a.stage_features, "images": N, "image_size": [W, H], "pairs": n_pairs, "steps": n_steps,
                     "ms_per_step": elapsed / n_steps * 1e3, "pairs_per_s": n_pairs * n_steps / elapsed,
                     "housekeeping_ms_per_step": housekeeping / n_steps * 1e3,
                     "step_ms": [r["ms_total"] for r in reps], "step_features_ms": [r["ms_features"] for r in reps],
                     "keypoints_per_image": last["n_keypoints"] / N,
                     "putative_pairs": int(last["n_putative_pairs"]), "putative_matches": int(last["n_putative_matches"]), "F_matches": int(last["n_F_matches"]),
                     "phases_ms": {k[3:]: mean(k) for k in ("ms_features", "ms_load", "ms_match", "ms_match_kernels", "ms_match_post", "ms_filter_F", "ms_filter_E", "ms_filter_H", "ms_filters_wall", "ms_files", "ms_total")},
                     # in situ = inside the timed step, with the other batches in flight beside them: HIP-event time per image, summed over the features contexts
                     "features_in_situ": {"detector_kernels_ms_per_image": last["features"]["ms_detect_kernels"] / N, "liop_kernels_ms_per_image": last["features"]["ms_liop_kernels"] / N,
                                          "liop_kernels_ms_per_28k_keypoints": last["features"]["ms_liop_kernels"] / max(last["n_keypoints"], 1) * 28000.0,
                                          "kernel_sum_ms": last["features"]["ms_detect_kernels"] + last["features"]["ms_liop_kernels"], "wall_ms": mean("ms_features")}}
            if embed:
                return quick
            print(json.dumps(quick))
            return
        # the GUI's default arm (matchingAlgorithm 0 = FLANN kd-trees in the reference, src/Regard3DMainFrame.cpp:2405) on the files
        # the step left: no extraction, matching + filters only -- under the facade's default policy (an approximate arm is served by
        # whichever matcher is faster on the views: exhaustive for LIOP-144), as requested (the graph matcher), and arm 9 the same
        # way for a like-for-like match phase: as the facade runs it (exact fast paths: split-f16 nomination for LIOP) and on plain f32 tiles
        def rerun(algo, **kw):
            wipe(False)
            r = stage.run(d, bare, 0.001, 0.6, algo, **kw).as_dict()
            return r, {x: open(os.path.join(d, f"matches.{x}.bin"), "rb").read() for x in ("putative", "f", "e", "h")}
        r0g, f0g = rerun(0, arms_as_requested=True)
        r9t, f9t = rerun(9, f32_tiles=True)
        r0, f0 = rerun(0)
        r9, f9 = rerun(9)                   # last: its files are what the CPU leg below compares with
        # detector roofline: a dedicated pass of the batch entry on B resident images, one context, nothing else on the GPU
        ctx = api.Context(0)
        B = min(8, N)
        ctx.detect_akaze_batch(imgs[:B], 0.001)
        ctx.detect_akaze_batch(imgs[:B], 0.001)
        sd = ctx.stats()
        det_gbs = sd.detect_algorithmic_bytes / (sd.ms_detect_kernels * 1e-3) / 1e9
        det_cmp_gbs = sd.detect_compulsory_bytes / (sd.ms_detect_kernels * 1e-3) / 1e9
        out = {
            "metric": "image-pairs matched/sec (+ F-inlier filter)", "value": n_pairs * a.steps / elapsed, "unit": "pairs/s", "n_gpus": 1,
            "steps": a.steps, "warmup": max(a.warmup, 1), "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "housekeeping_ms_per_step": housekeeping / a.steps * 1e3,
            "vs_baseline": None, "dtype": "f32 (detector, LIOP, L2) / f64 (AC-RANSAC)", "data": "synthetic",
            "config": {"workload": f"stage: {N} synthetic {W}x{H} photographs (one textured plane, 78 % overlap between neighbours) resident in HBM -> "
                                   f"R3DComputeMatches::computeMatches: Fast-A-KAZE + LIOP-144 ({conc} batches of {batch} in flight) -> .feat/.desc -> exhaustive {n_pairs} pairs, "
                                   "brute-force L2 2-NN + ratio 0.6 (matchingAlgorithm 9; split-f16 nomination + f32 re-score, bit-identical to f32 tiles) -> F + E + H AC-RANSAC (4 px, 2048 it) -> matches.*.txt/.bin",
                       "name": "stage", "images": N, "pairs": n_pairs, "image_size": [W, H], "parallelism": "1 GPU",
                       "host_options": "R3DM_STAGE_BACKGROUND_NICE (R3DComputeMatches::setBackgroundThreadsNice(10)): this host asks the stage's background "
                                       "writer threads to stand back -- the box's container owns 16 cores; off by default in the library"},
            "phases_ms": {k[3:]: mean(k) for k in ("ms_features", "ms_load", "ms_match", "ms_filter_F", "ms_filter_E", "ms_filter_H", "ms_filters_wall", "ms_files", "ms_total")},
            "phases_note": "filter_F / _E / _H run side by side on the one device (r3dm_filter_FEH): they overlap, filters_wall is their sum in the total",
            "kernels_ms": {"match": mean("ms_match_kernels"), "match_post_wall": mean("ms_match_post"), "F": mean("ms_F_kernels"), "E": mean("ms_E_kernels"), "H": mean("ms_H_kernels"),
                           "detector_sum_over_contexts": last["features"]["ms_detect_kernels"], "liop_sum_over_contexts": last["features"]["ms_liop_kernels"]},
            "features_in_situ": {"detector_kernels_ms_per_image": last["features"]["ms_detect_kernels"] / N, "liop_kernels_ms_per_image": last["features"]["ms_liop_kernels"] / N,
                                 "liop_kernels_ms_per_28k_keypoints": last["features"]["ms_liop_kernels"] / max(last["n_keypoints"], 1) * 28000.0,
                                 "kernel_sum_ms": last["features"]["ms_detect_kernels"] + last["features"]["ms_liop_kernels"], "wall_ms": mean("ms_features"),
                                 "note": "HIP-event time inside the timed step with the other batches in flight beside them, summed over the features contexts"},
            "features": {"images_per_s": N / (mean("ms_features") * 1e-3), "ms_per_image": mean("ms_features") / N, "keypoints": int(last["n_keypoints"]),
                         "keypoints_per_image": last["n_keypoints"] / N, "file_ms_sum_over_contexts": last["features"]["ms_files"],
                         "detector_passes": int(last["features"]["n_passes"]), "regrows": int(last["features"]["n_regrows"])},
            "graphs": {k: int(last[k]) for k in ("n_putative_pairs", "n_putative_matches", "n_F_pairs", "n_F_matches", "n_E_pairs", "n_E_matches", "n_H_pairs", "n_H_matches")},
            "arm_9_on_existing_files": {"ms_match": r9["ms_match"], "ms_match_kernels": r9["ms_match_kernels"], "ms_total": r9["ms_total"], "putative_matches": int(r9["n_putative_matches"])},
            "arm_9_on_plain_f32_tiles": {"ms_match": r9t["ms_match"], "ms_match_kernels": r9t["ms_match_kernels"], "ms_total": r9t["ms_total"],
                                         "all_match_files_identical_to_arm_9": bool(f9t == f9),
                                         "note": "R3DM_STAGE_F32_TILES: the arithmetic BASELINE's configurations name; the facade's default nominates on split-f16 "
                                                 "tiles and re-scores in f32 in the reference's order -- bit-identical files (include/r3d_compute_matches.hpp)"},
            "arm_0_gui_default": {"ms_match": r0["ms_match"], "ms_total": r0["ms_total"], "putative_matches": int(r0["n_putative_matches"]),
                                  "served_by": "exhaustive matcher" if r0["match_was_exhaustive"] else "graph matcher",
                                  "all_match_files_identical_to_arm_9": bool(f0 == f9),
                                  "note": "the reference's arm 0 is FLANN kd-trees (approximate); the facade serves an approximate arm with the exhaustive matcher "
                                          "when r3dm_exhaustive_is_faster says so for the registered views (LIOP-144: real-valued rows), DESIGN.md section 4.7"},
            "arm_0_as_requested_graph_matcher": {"ms_match": r0g["ms_match"], "ms_total": r0g["ms_total"], "putative_matches": int(r0g["n_putative_matches"]),
                                                 "served_by": "exhaustive matcher" if r0g["match_was_exhaustive"] else "graph matcher",
                                                 "putative_matches_recovered_vs_arm_9": r0g["n_putative_matches"] / max(r9["n_putative_matches"], 1)},
            "roofline": {"bound": "hbm", "kernel": f"Fast-A-KAZE detector pass, B = {B} images (ak_* kernels, first scale-space launch .. keypoint compaction)",
                         "achieved": det_cmp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": det_cmp_gbs / HBM_PEAK_GBS, "traffic": None,
                         "bytes_counted": "compulsory",
                         "compulsory_bytes_per_image": sd.detect_compulsory_bytes / B, "as_structured_bytes_per_image": sd.detect_algorithmic_bytes / B,
                         "achieved_as_structured": det_gbs, "frac_as_structured": det_gbs / HBM_PEAK_GBS,
                         "ms_per_image": sd.ms_detect_kernels / B,
                         "note": "compulsory bytes = what a perfectly fused level would still move (smoothed plane in + out, determinant out, conductivity out, "
                                 "8 B per pixel and FED step: round 2's count); as-structured bytes = every stencil pass of the launch sequence reads / writes "
                                 "whole planes once (round 3's count; PMC FETCH_SIZE agrees with it).  `frac` is on the compulsory count -- fusing passes raises it, "
                                 "the as-structured fraction only says how fast the passes that exist run (DESIGN.md section 4.8).  Measured in a dedicated pass "
                                 f"(one context, HIP events on the library's stream); the stage itself keeps {conc} such passes in flight"},
        }
        out["roofline_liop"] = stage_liop_roofline(ctx, dev, last, N, imgs[0])
        # the AC-RANSAC kernels: counted f64 flops of the residual passes over the HIP-event time of the side-by-side call + CU occupancy
        out["roofline_filters"] = stage_filter_roofline(d, views)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_fn(ctx, imgs, d, K, W, H, a.cpu_seconds)      # (bench.py: the only caller of oracle/)
        print(json.dumps(out))
    finally:
        stage.close()
        shutil.rmtree(d, ignore_errors=True)


# liop_kernel<true> (the product path of the features stage since round 5: warp + blur of the 41 x 41 patch inside the descriptor's
# wavefront, one wavefront per keypoint): VALU instructions per keypoint = SQ_INSTS_VALU of a rocprofv3 --pmc pass of
# tools/liop_fused_perf.py 120000 over its 120,000 keypoints (profiles/r05_final_pmc_liop.txt: 1.37918e9 per launch; + 1,914 SALU and
# 1,675 LDS instructions per keypoint; the two-kernel form it replaced: 8,351 + 3,319).  A wave64 VALU instruction occupies its SIMD16
# for 4 cycles: the chip issues at most 256 CU x 4 SIMD x 2.4 GHz / 4 = 614.4 G wave-instructions/s.
LIOP_VALU_PER_PATCH = 11493.0
VALU_ISSUE_PEAK_G = 256 * 4 * 2.4e9 / 4 / 1e9


def stage_liop_roofline(ctx, dev, last, n_images, image):
    """The fused LIOP kernel in a dedicated pass: the keypoints the detector finds on one of the stage's photographs (resident in HBM),
    repeated to 65,536, through r3dm_extract_liop -- bound by VALU issue (the bilinear warp, the 1,024-key bitonic network on
    (intensity, position) keys and the f64 bilinear samples); its HBM traffic is a few per cent of the HBM roof.  The stage's own LIOP
    time is reported beside it."""
    n = 65536
    kps, _ = ctx.detect_akaze(image, 0.001)
    if len(kps) == 0:
        return None
    K = np.tile(kps, ((n + len(kps) - 1) // len(kps), 1))[:n].copy()
    ms = []
    for _ in range(3):
        ctx.extract_liop(image, K, 8.0)
        ms.append(ctx.stats().ms_liop_kernel)
    m = sorted(ms)[1]
    rate = n / (m * 1e-3)
    ach = rate * LIOP_VALU_PER_PATCH / 1e9
    kp = float(last["n_keypoints"])
    return {"bound": "valu", "kernel": "liop_kernel<true> (patch warp + blur + descriptor, one wavefront per keypoint; 65,536 keypoints of one photograph, dedicated pass)",
            "achieved": ach, "peak": VALU_ISSUE_PEAK_G,
            "unit": "G wave-instructions/s (VALU issue)", "frac": ach / VALU_ISSUE_PEAK_G, "traffic": None,
            "valu_instructions_per_patch": LIOP_VALU_PER_PATCH, "patches_per_s": rate, "kernel_ms": m, "ms_per_28k_keypoints": m / n * 28000.0,
            "in_stage": {"liop_kernels_ms_per_image_sum_over_contexts": last["features"]["ms_liop_kernels"] / n_images,
                         "keypoints_per_image": kp / n_images,
                         "patches_per_s": kp / (last["features"]["ms_liop_kernels"] * 1e-3) if last["features"]["ms_liop_kernels"] > 0 else None},
            "note": "instructions per keypoint from PMC (profiles/r05_final_pmc_liop.txt: SQ_INSTS_VALU of this kernel over its keypoints, same source) x keypoints / "
                    "HIP-event time.  in_stage: the same kernel over the keypoints of all images, as the contexts' events time it with the other "
                    "batches in flight beside it (their sum exceeds the wall time of the features phase)"}


# f64 operations of ONE residual (one model applied to one putative match), counted on the source (kernels_filter.hip; + - * / one each):
#   F  sym_epipolar_err:   F x1 12, F^T x2 8, x2.(F x1) 4, squares + sums + two reciprocals + the product 12          = 36
#   E  epipolar_dist_err:  l = F x1 12, l.x2 4, d^2 1, l0^2 + l1^2 3, the quotient 1                                 = 21
#   H  h_asym_err:         w 4, two numerators 8, two quotients 2, two differences 2, squares + sum 3               = 19
RESIDUAL_F64_FLOPS = {"F": 36.0, "E": 21.0, "H": 19.0}
F64_VECTOR_PEAK_T = 78.6          # MI355X_MICROARCH.md: FP64 vector = 256 CU x 64 FMA lanes/clk x 2 x 2.4 GHz


def stage_filter_roofline(d, views):
    """The AC-RANSAC kernels of the stage (acransac_coop_kernel for pairs of >= 4096 putatives + acransac_kernel<kind> for the rest), F, E
    and H side by side as the facade runs them: counted f64 flops of the residual passes = sum over pairs of
    (models evaluated, r3dm_filter_report) x (putative matches of the pair) x RESIDUAL_F64_FLOPS[kind], over the HIP-event time of
    the side-by-side call, against the f64 vector peak; CU occupancy = workgroups of the call / 256.  Every input of the fraction is
    in the object: flops = sum(per_kind[k].model_match_evaluations x flops_per_residual[k])."""
    g = api.Graph.load(os.path.join(d, "matches.putative.bin"))
    ctx = api.Context(0)
    import numpy as _np
    for v in views:
        raw = _np.fromfile(os.path.join(d, v["basename"] + ".desc"), _np.uint8)
        n = int(_np.frombuffer(raw[:8].tobytes(), _np.uint64)[0])
        desc = _np.frombuffer(raw[8:].tobytes(), _np.float32).reshape(n, 144)
        xy = _np.loadtxt(os.path.join(d, v["basename"] + ".feat"), dtype=_np.float32).reshape(-1, 4)[:, :2].copy()
        ctx.set_image(v["id"], desc, xy, v["width"], v["height"])
        ctx.set_intrinsics(v["id"], _np.array([[v["focal_px"], 0, v["ppx"]], [0, v["focal_px"], v["ppy"]], [0, 0, 1.0]]))
    counts = _np.diff(g.offsets.astype(_np.int64)).astype(_np.float64)
    ctx.filter_FEH(g, "FEH", 4.0, 2048, seed=5489)                       # warm: buffers, streams
    ms3 = []
    for _ in range(3):
        _, msk, _ = ctx.filter_FEH(g, "FEH", 4.0, 2048, seed=5489)
        ms3.append(float(max(msk)))
    st = ctx.stats()
    wgs, coop_items = int(st.n_filter_workgroups), int(st.n_filter_coop_pairs)
    ms = sorted(ms3)[1]
    per_kind, flops = {}, 0.0
    for kind, call in (("F", ctx.filter_F), ("E", ctx.filter_E), ("H", ctx.filter_H)):
        call(g, 4.0, 2048, seed=5489)
        alone_ms = ctx.stats().ms_filter_kernels
        rep = ctx.filter_report()
        ev = float(sum(r[3] * m for r, m in zip(rep, counts)))
        per_kind[kind] = {"models_evaluated": int(sum(r[3] for r in rep)), "iterations": int(sum(r[2] for r in rep)),
                          "model_match_evaluations": ev, "flops_per_residual": RESIDUAL_F64_FLOPS[kind], "kernel_ms_alone": alone_ms}
        flops += ev * RESIDUAL_F64_FLOPS[kind]
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    ctx.close()
    return {"bound": "valu", "kernel": "acransac_coop_kernel (pairs of >= 4096 putatives, F + E + H in one pool of workgroups) + acransac_kernel<kind> (shorter pairs)",
            "achieved": ach, "peak": F64_VECTOR_PEAK_T, "unit": "TFLOP/s (f64 vector)", "frac": ach / F64_VECTOR_PEAK_T, "traffic": None,
            "kernel_ms": ms, "kernel_ms_runs": ms3, "flops": flops, "per_kind": per_kind,
            "pairs": int(g.num_pairs), "putative_matches": int(counts.sum()),
            "workgroups": wgs, "items_on_cooperative_kernel": coop_items, "cu_occupancy": min(wgs, 256) / 256.0,
            "note": "residual passes only (the minimal solvers, the NFA walk and the rare sorts are not counted: this is a lower bound of the f64 work). "
                    "The algorithm is a chain of dependent iterations per pair (sample -> models -> residuals of all matches -> NFA -> pool), "
                    "so the f64 roof is an upper bound no schedule reaches; round 3's one-workgroup-per-pair shape ran the same stage at "
                    "75 ms with 94 of 256 CUs busy (profiles/r03_end_bench_stage_kernel_stats.txt)"}


def roofline(name, cfg, acc, dim, world):
    """the dominant kernel of the config: algorithmic work of its launches / their HIP-event time (events recorded on the
    library's own stream inside r3dm_match_pairs*)"""
    L = max(acc["launches"], 1)
    if cfg["matcher"] == "kgraph":
        ms = acc["ann_ms"]
        # The search is VALU-ISSUE-bound, not gather-bound (PMC: profiles/r05_final_pmc_c5.txt):
        # one wavefront per query executes 5.6 k VALU (+ 5.3 k SALU) instructions on the byte-row / v_dot4 path, and a wave64 VALU
        # instruction occupies its SIMD16 for 4 cycles -> the chip issues at most 256 CU x 4 SIMD x 2.4 GHz / 4 = 614.4 G wave-instructions/s.
        # 99.3 % of the row gathers are served by L1 / L2; what reaches the fabric is reported as `traffic` (PMC, scaled per pair).
        rows8 = acc.get("ann_rows8", 0) > 0 and acc.get("ann_rows8", 0) == acc.get("ann_launches", -1)
        rows16 = acc.get("ann_rows16", 0) > 0 and acc.get("ann_rows16", 0) == acc.get("ann_launches", -1)
        dot8 = acc.get("ann_dot8", 0) == acc.get("ann_launches", -1)
        row_bytes = dim * (1.0 if rows8 else 2.0 if rows16 else 4.0)
        # instructions per query and HBM-side bytes per pair: PMC passes of `bench.py --config c5 --images 96`, reported only while the
        # library holds the very machine code of ann_search_kernel<8, 3> that was profiled (profiles/pmc_traffic.json, tools/pmc_traffic_json.py)
        ent, ent_why = None, "no PMC entry for this kernel"
        ent_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if rows8 and dot8 and os.path.exists(ent_path):
            e = json.load(open(ent_path)).get("c5:ann_search_kernel")
            if e and e.get("code_sha16"):
                cur = _kernel_code_sha16(e["kernel"])
                if cur == e["code_sha16"]:
                    ent = e
                else:
                    ent_why = f"profiles/pmc_traffic.json: the entry for {e['kernel']} was measured on other machine code (entry {e['code_sha16']}, this library {cur})"
        VALU_PER_QUERY = ent["valu_instructions_per_query"] if ent else None
        peak = 256 * 4 * 2.4e9 / 4 / 1e9
        ach = acc["queries"] * VALU_PER_QUERY / (ms * 1e-3) / 1e9 if (ms > 0 and VALU_PER_QUERY) else 0.0
        out = {"bound": "valu", "kernel": ("ann_search_kernel<u8 rows, v_dot4>" if dot8 else "ann_search_kernel<u8 rows>") if rows8 else "ann_search_kernel<bf16 rows>" if rows16 else "ann_search_kernel",
               "achieved": ach, "peak": peak, "unit": "G wave-instructions/s (VALU issue)", "frac": ach / peak, "traffic": None,
               "valu_instructions_per_query": VALU_PER_QUERY, "salu_instructions_per_query": ent["salu_instructions_per_query"] if ent else None,
               "note": ("VALU-issue-bound: instructions per query from PMC (%s: SQ_INSTS_VALU / SQ_WAVES of this kernel, one wavefront per query; same machine code, "
                        "code_sha16 %s) x queries / HIP-event time of the launches; the gathers (evaluations x %d B rows) are 99 %% cache hits"
                        % (ent["from"], ent["code_sha16"], int(row_bytes))) if ent else ("instructions per query not reported: " + ent_why),
               "evaluations": int(acc["ann_dist"]), "evaluations_per_query": acc["ann_dist"] / max(acc["queries"], 1), "row_bytes": int(row_bytes),
               "gathered_GB_per_s": acc["ann_dist"] * row_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "search_ms_total": ms}
        if ent:
            out["traffic"] = ent["traffic_bytes_per_pair"] * acc.get("pairs", 0) / max(acc["launches"], 1)
            out["traffic_source"] = (f"{ent['from']}: FETCH_SIZE + WRITE_SIZE of the search launch of a 96-image step, per pair "
                                     f"({ent['traffic_bytes_per_pair'] / 1e6:.2f} MB) x the pairs of a launch")
        return out
    ms = acc["kernel_ms"]
    if cfg["kind"] == "akaze":
        t_ops = acc["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0     # lane-ops: 2 per 32-bit word (xor, popcount-accumulate)
        return {"bound": "valu", "kernel": "hamming_knn2_kernel<W=16,QL=4>", "achieved": t_ops, "peak": VALU_LANE_OPS_PEAK_T,
                "unit": "T lane-op/s", "frac": t_ops / VALU_LANE_OPS_PEAK_T, "traffic": None,
                "peak_source": "measured issue ceiling of the v_xor_b32 + v_bcnt_u32_b32 pair on this chip (tools/ubench/valu_issue.hip, profiles/r04_ubench_valu_issue.txt)",
                "stated_roof_survey_8d": VALU_LANE_OPS_STATED_T, "frac_of_stated_roof": t_ops / VALU_LANE_OPS_STATED_T,
                "algorithmic_share_of_valu_instructions": HAMMING_ALGORITHMIC_VALU_SHARE,
                "frac_counting_every_valu_instruction": t_ops / HAMMING_ALGORITHMIC_VALU_SHARE / VALU_LANE_OPS_PEAK_T,
                "note": "integer VALU issue bound (neither MFMA nor HBM).  78.6 T lane-op/s is the packed-f32 lane rate; v_bcnt_u32_b32 issues once per "
                        "~4.2 cycles per SIMD (37.9 T lane-op/s alone) and the xor + popcount pair sustains 42.2 T in registers.  The kernel's inner loop "
                        "spends 28 of 284 VALU instructions on the two-smallest tracking (ISA count), so 0.90 x 42.2 = 38.0 T is what this loop can reach",
                "avg_launch_ms": ms / L, "lane_ops_per_launch": acc["flops"] / L, "launches": int(acc["launches"])}
    tf = acc["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    G = 18 if dim == 144 else dim // 8
    return {"bound": "mfma", "kernel": f"l2_knn2_mfma_kernel<G={G},NJ=2>", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms / L,
            "flops_per_launch": acc["flops"] / L, "launches": int(acc["launches"])}


def _sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def _kernel_code_sha16(kernel):
    from regard3d_amd.codeobj import kernel_hash, mangled_needle
    return kernel_hash(os.path.join(ROOT, "regard3d_amd", "libr3dm.so"), mangled_needle(kernel))


def attach_traffic(roof, config, kernel, at_named_size):
    """HBM traffic of one launch of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command (a process
    cannot profile itself): profiles/pmc_traffic.json (tools/pmc_traffic_json.py), keyed by config + kernel and by the fingerprint
    of the kernel's MACHINE CODE in the library that was profiled (regard3d_amd/codeobj.py).  An entry is printed only while the
    library this process runs holds that very code -- edits elsewhere in the source file do not stale it, and the same library
    state gives the same answer in every run; otherwise traffic stays null and traffic_source says why."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath) or not at_named_size:          # (counters were taken on the config as BASELINE names it: whole job, full size)
        return
    ent = json.load(open(tpath)).get(f"{config}:{kernel}")
    if not ent:
        return
    cur = _kernel_code_sha16(ent["kernel"])
    if not ent.get("code_sha16") or ent.get("code_sha16") != cur:
        roof["traffic_source"] = (f"profiles/pmc_traffic.json: the entry for {ent['kernel']} was measured on other machine code "
                                  f"(entry {ent.get('code_sha16')}, this library {cur}): not reported")
        return
    if "traffic_bytes_per_launch" not in ent:                   # (an entry counted per pair -- c5's graph search -- is reported by its own leg)
        return
    roof["traffic"] = ent["traffic_bytes_per_launch"]
    roof["traffic_source"] = (f"profiles/pmc_traffic.json <- {ent.get('from', '?')}: separate rocprofv3 --pmc passes of this command on the same "
                              f"kernel machine code (code_sha16 {cur}; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), not measured inside this process"
                              + ("; algorithmic %.4g bytes/launch" % ent["algorithmic_bytes_per_launch"] if "algorithmic_bytes_per_launch" in ent else ""))


def opt_in_integer(ctx, step, fence, g, gf, job_pairs):
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
    return {"value": job_pairs / el2, "unit": "pairs/s", "ms_per_step": el2 * 1e3, "steps": 1,
            "identical_to_headline_graphs": bool(same), "integer_mfma_launches": int(sm2.n_integer_mfma),
            "dtype": "bf16 operands holding exact integers, f32 accumulate (exact below 2^24)",
            "roofline": {"bound": "mfma", "kernel": "l2_knn2_int_kernel<GB=8,NJ=3>", "achieved": ach2, "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach2 / BF16_MFMA_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_ms": sm2.ms_match_kernels / max(sm2.n_match_launches, 1)},
            "filter_kernel_ms": sa2.ms_filter_kernels,
            "wall_ms": {"match": sa2.ms_wall_match, "match_post": sa2.ms_wall_match_post, "filter": sa2.ms_wall_filter}}


def opt_in_split(ctx, step, fence, g, gf, job_pairs):
    """the same step with the split-f16 nominator (r3dm_set_split_mfma): nomination on v_mfma_f32_32x32x16_f16, distances and
    certification in the reference's f32 arithmetic as before -- bit-identical graphs, reported beside the headline"""
    ctx.set_split_mfma(True)
    try:
        step(); fence()
        t1 = time.perf_counter()
        g2, gf2, _, sm2, sa2 = step()
        fence()
        el2 = time.perf_counter() - t1
    finally:
        ctx.set_split_mfma(False)
    same = all(np.array_equal(getattr(x, f), getattr(y, f)) for x, y in ((g, g2), (gf, gf2)) for f in ("pairs", "offsets", "matches"))
    # matrix work: the split kernel runs 3 f16 MFMAs per 16 dims (3 x the algorithmic 2 n^2 D flops); on COUNT tiles (rows = integer votes
    # x a row scale: what LIOP is) the nominator runs ONE (l2_knn2_counts2_kernel: executed = algorithmic)
    counts = int(getattr(sm2, "n_counts_mfma", 0)) > 0
    mult = 1.0 if counts else 3.0
    ach = mult * sm2.algorithmic_flops / (sm2.ms_match_kernels * 1e-3) / 1e12 if sm2.ms_match_kernels > 0 else 0.0
    return {"value": job_pairs / el2, "unit": "pairs/s", "ms_per_step": el2 * 1e3, "steps": 1,
            "identical_to_headline_graphs": bool(same), "split_mfma_launches": int(sm2.n_split_mfma), "count_tile_launches": int(getattr(sm2, "n_counts_mfma", 0)),
            "dtype": ("f16 integer votes nominate (1 MFMA per 16 dims, f32 accumulate), row scales in the epilogue" if counts else
                      "f16 hi/lo pieces nominate (3 MFMAs per 16 dims, f32 accumulate)") + "; distances re-scored in f32 as in the headline",
            "exact_fallback_fraction": sm2.n_exact_fallback / max(sm2.n_queries, 1),
            "roofline": {"bound": "mfma", "kernel": "l2_knn2_counts2_kernel<GB=9,PF=9>" if counts else "l2_knn2_split_kernel<GB=9,NJ=2>", "achieved": ach,
                         "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s (executed f16 matrix flops = %d x algorithmic)" % int(mult), "frac": ach / BF16_MFMA_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_tflops": ach / mult, "avg_launch_ms": sm2.ms_match_kernels / max(sm2.n_match_launches, 1),
                         "algorithmic_bytes_per_launch": sm2.algorithmic_bytes / max(sm2.n_match_launches, 1)},
            "filter_kernel_ms": sa2.ms_filter_kernels,
            "wall_ms": {"match": sa2.ms_wall_match, "match_post": sa2.ms_wall_match_post, "filter": sa2.ms_wall_filter}}


def opt_in_hamming(ctx, step, fence, g, gf, job_pairs):
    """the same step with the exact MFMA formulation of the Hamming matcher (r3dm_set_hamming_mfma): bits as 0 / 1 bytes on
    v_mfma_i32_32x32x32_i8, d = popcount(a) + popcount(b) - 2 a.b -- bit-identical graphs, reported beside the popcount headline"""
    ctx.set_hamming_mfma(True)
    try:
        step(); fence()
        t1 = time.perf_counter()
        g2, gf2, _, sm2, sa2 = step()
        fence()
        el2 = time.perf_counter() - t1
    finally:
        ctx.set_hamming_mfma(False)
    same = all(np.array_equal(getattr(x, f), getattr(y, f)) for x, y in ((g, g2), (gf, gf2)) for f in ("pairs", "offsets", "matches"))
    # executed matrix work: 2 n^2 x 512 bit-products per pair (486 bits padded to 16 blocks of 32); algorithmic_flops counts 2 n^2 x 16 words
    ach = 32.0 * sm2.algorithmic_flops / (sm2.ms_match_kernels * 1e-3) / 1e12 if sm2.ms_match_kernels > 0 else 0.0
    return {"value": job_pairs / el2, "unit": "pairs/s", "ms_per_step": el2 * 1e3, "steps": 1,
            "identical_to_headline_graphs": bool(same), "hamming_mfma_launches": int(sm2.n_hamming_mfma),
            "dtype": "i8 operands holding bits 0/1, i32 accumulate (exact)",
            "roofline": {"bound": "mfma", "kernel": "l2_knn2_int_lds_kernel<GB=16,NJ=2,OPS=i8>", "achieved": ach, "peak": 5000.0,
                         "unit": "TOP/s (executed i8 matrix ops, 512 bits per row)", "frac": ach / 5000.0, "traffic": None,
                         "avg_launch_ms": sm2.ms_match_kernels / max(sm2.n_match_launches, 1)},
            "filter_kernel_ms": sa2.ms_filter_kernels,
            "wall_ms": {"match": sa2.ms_wall_match, "match_post": sa2.ms_wall_match_post, "filter": sa2.ms_wall_filter}}


