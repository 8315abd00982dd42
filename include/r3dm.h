/*
 * r3dm.h -- C ABI of the MI355X-native compute-matches hot path (libr3dm.so).
 *
 * Drop-in boundary for Regard3D's "Compute matches" stage.  Each entry point replaces one
 * interface of the reference (rhiestan/Regard3D, paths relative to /root/reference):
 *
 *   r3dm_set_image          <- Regions_Provider::load + regions_provider->get(I)
 *                              (src/R3DComputeMatches.cpp:2040, 443-472): one call per view with the
 *                              view's DescriptorRawData() / RegionCount() / DescriptorLength() and its
 *                              feature positions (Features_Provider::load, :2094-2095)
 *   r3dm_match_pairs        <- Matcher_Regions(fDistRatio, BRUTE_FORCE_L2).Match(regions_provider,
 *                              pairs, map_PutativesMatches) (src/R3DComputeMatches.cpp:2037-2039,2048)
 *                              and its in-tree clones kgraph_match / hnsw_match / mrpt_match
 *                              (:808-902, :502-597, :423-491): a new "matchingAlgorithm == 9" arm
 *   r3dm_filter_F           <- ImageCollectionGeometricFilter::Robust_model_estimation(
 *                              GeometricFilter_FMatrix_AC(4.0, 2048), putative, false) +
 *                              Get_geometric_matches() (src/R3DComputeMatches.cpp:2099,2113-2115)
 *   r3dm_knn2               <- openMVG::matching::ArrayMatcher<Scalar,Metric>::Build +
 *                              SearchNeighbours(query, nbQuery, &idx, &dist, NN=2)
 *                              (plugin contract: src/utils/matcher_kgraph.h:120-125,205-211)
 *   r3dm_save_matches /     <- openMVG::matching::Save / Load(PairWiseMatches, "matches.*.txt|.bin")
 *   r3dm_load_matches          (src/R3DComputeMatches.cpp:2064,2120; names src/R3DProject.cpp:858-871)
 *   r3dm_graph_*            <- openMVG::matching::PairWiseMatches accessors
 *                              (R3DComputeMatchesStatistics, src/R3DComputeMatches.h:59-66)
 *
 * Conventions: every function returns 0 on success or a negative r3dm_status; nothing throws;
 * all state lives in the opaque context; buffers passed in stay owned by the caller (they are
 * copied -- host or device pointers are both accepted, the copy is hipMemcpyDefault); objects
 * returned by the library are released only with the matching *_free.  One context per host
 * thread (or serialise externally); one context drives one GPU (one process per GPU, or r3dm_multi_* below for one
 * process driving several GPUs).
 * There is NO CPU fallback: if no gfx950 device is visible r3dm_create fails with
 * R3DM_ERR_NO_DEVICE.
 */
#ifndef R3DM_H
#define R3DM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct r3dm_ctx r3dm_ctx;
typedef struct r3dm_graph r3dm_graph;

typedef enum {
    R3DM_OK = 0,
    R3DM_ERR_INVALID = -1,     /* bad argument                                   */
    R3DM_ERR_NO_DEVICE = -2,   /* no usable gfx950 GPU / HIP runtime failure      */
    R3DM_ERR_HIP = -3,         /* a HIP call failed (see r3dm_last_error)         */
    R3DM_ERR_IO = -4,          /* file could not be read / written                */
    R3DM_ERR_UNSUPPORTED = -5, /* descriptor type / metric combination            */
    R3DM_ERR_NOMEM = -6
} r3dm_status;

typedef enum {
    R3DM_F32 = 0,   /* Scalar_Regions<.., float, D>: D floats per row (SIFT-128 f32, LIOP-144) */
    R3DM_U8 = 1,    /* Scalar_Regions<.., unsigned char, D>: D bytes per row, L2 metric        */
    R3DM_BIN = 2    /* Binary_Regions<.., NB>: NB bytes per row, Hamming metric                 */
} r3dm_dtype;

typedef struct { uint32_t i, j; } r3dm_match;   /* IndMatch{i_, j_}: i_ = row in I, j_ = row in J */

/* which squared epipolar error the AC kernel binds (default symmetric) */
typedef enum { R3DM_ERR_SYMMETRIC_EPIPOLAR = 0, R3DM_ERR_SAMPSON = 1, R3DM_ERR_EPIPOLAR_ONE_SIDED = 2 } r3dm_ferror;

/* ---- context ---- */
int  r3dm_create(int device_id, r3dm_ctx** out);
void r3dm_destroy(r3dm_ctx* ctx);
const char* r3dm_last_error(const r3dm_ctx* ctx);
/* "gfx950", CU count, bytes of HBM -- for reports */
int  r3dm_device_info(const r3dm_ctx* ctx, char* arch, size_t arch_cap, int* n_cu, uint64_t* hbm_bytes);

/* ---- views ----
 * Register (or replace) view `view_id`: n descriptors of `dim` elements (floats for F32, bytes for
 * U8/BIN), row-major, plus optional feature positions xy (n x 2 floats, pixel coordinates; NULL if
 * no coordinate de-duplication / geometric filter is wanted).  The data is copied to HBM and
 * re-laid-out there (MFMA fragment-order tiles + row norms).
 * What the reference does before it matches (Regions_Provider::load / Features_Provider::load, src/R3DComputeMatches.cpp:2040,2094-2095).
 * The call returns when the caller's buffers are consumed, NOT when the view is laid out: rows in pageable host memory are copied
 * into a ring of page-locked slots and travel from there (one DMA + one kernel per view, queued on the context's streams); rows
 * behind device pointers (this device's or a peer's) and page-locked host pointers are copied by the copy engine into the ring's
 * device slot, and only that copy is waited for.  A device buffer must be COMPLETE when the call is made: the library's streams are
 * not ordered behind the caller's, so the stream that produced the buffer has to be synchronised first (the features workers of
 * r3dm_multi_extract_features hand their batches over that way).  Every later call of the context is ordered behind
 * the registration; r3dm_images_wait waits for it explicitly.  A view costs its f32 tiles + norms in HBM (1.0 x its f32 size); the
 * layouts only some paths read (row-major rows for real-valued views and the approximate matchers, bf16 / split-f16 / count /
 * byte tiles of the opt-in paths) are made by the first call that needs them, or here when the path's switch is already on. */
int r3dm_set_image(r3dm_ctx* ctx, uint32_t view_id, uint32_t width, uint32_t height,
                   const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy);
/* ... a whole collection in one call (the reference loads all regions in one Regions_Provider::load): helper threads of the host
 * fill the ring with the next views while the DMA and the kernels of the previous ones run.  dtype: r3dm_dtype. */
typedef struct r3dm_view_desc {
    uint32_t view_id, width, height;
    uint32_t n, dim;
    int32_t dtype;
    const void* desc;
    const float* xy;
} r3dm_view_desc;
int r3dm_set_images(r3dm_ctx* ctx, const r3dm_view_desc* views, uint32_t n_views);
int r3dm_images_wait(r3dm_ctx* ctx);
/* What a registered view holds in HBM (reports, tests): *layouts = the on-demand layouts staged so far, bits R3DM_LAYOUT_*;
 * *bytes = device memory of every layout and index of the view; *ring_uploads / *direct_uploads (context-wide, may be NULL) = views that
 * travelled through the page-locked ring / were copied straight from the caller's device or page-locked buffers (and empty views),
 * since r3dm_create. */
#define R3DM_LAYOUT_ROWS   1u   /* row-major f32 rows */
#define R3DM_LAYOUT_BF16   2u   /* bf16 tiles (r3dm_set_integer_mfma) */
#define R3DM_LAYOUT_SPLIT  4u   /* split-f16 planes (r3dm_set_split_mfma) */
#define R3DM_LAYOUT_COUNTS 8u   /* count tiles + scale order (r3dm_set_split_mfma, votes x scale rows) */
#define R3DM_LAYOUT_BIN8   16u  /* byte-per-bit tiles (r3dm_set_hamming_mfma) */
int r3dm_view_info(r3dm_ctx* ctx, uint32_t view_id, uint32_t* layouts, uint64_t* bytes, uint64_t* ring_uploads, uint64_t* direct_uploads);
/* device memory the context holds for its registered views (the slabs their layouts are cut from, live and recycled blocks alike),
 * for the upload ring (device slots + position-class tables), and the ring's page-locked host memory; any pointer may be NULL */
int r3dm_memory_info(const r3dm_ctx* ctx, uint64_t* views_device_bytes, uint64_t* ring_device_bytes, uint64_t* ring_host_bytes);
int r3dm_clear_images(r3dm_ctx* ctx);
/* r3dm_clear_images keeps the staging buffers of up to 256 cleared views for the next collection (the index buffers of a view are
 * always released); r3dm_trim gives those spares back to the device as well. */
int r3dm_trim(r3dm_ctx* ctx);

/* Opt-in integer fast path of the L2 matcher (default off; no reference counterpart -- the reference has one L2 loop,
 * openMVG/matching/metric.hpp L2_Vectorized via ArrayMatcherBruteForce).  When every view of a batch holds
 * integer-valued descriptors of magnitude <= 256 (SIFT bins 0..255 stored as float or u8), the all-pairs contraction
 * runs on v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_32x32x2_f32: every such value is a bf16, every product and
 * partial sum an integer below 2^24, so the f32 accumulators -- and therefore the matches -- are bit-identical to the
 * f32 path and to the reference; the kernel re-checks the condition per pair and sends anything else to the exact
 * scan.  Batches with any other view (LIOP, normalised SIFT) keep the f32 tiles.  r3dm_stats.n_integer_mfma reports
 * which path ran. */
int r3dm_set_integer_mfma(r3dm_ctx* ctx, int enable);

/* Opt-in split-f16 nominator of the L2 matcher for REAL-valued descriptors (default off; no reference counterpart) -- LIOP-144,
 * normalised SIFT: what Regard3D actually matches (src/Regard3DFeatures.h:44-48).  Their matching is always "nominate on
 * MFMA keys, re-score the nominees with the reference's own summation (L2_Vectorized order, no FMA), certify against a
 * rounding slack, exact scan for what cannot be certified".  With this switch the nomination runs on
 * v_mfma_f32_32x32x16_f16 with every value split into two f16 pieces (a.b ~ ah.bh + al.bh + ah.bl, 3/16 of the matrix
 * cycles of the f32 tiles) and a slack that covers the split; the distances that are compared, ratio-tested and returned
 * are still the reference's f32 sums, so results are bit-identical to the default path and to the CPU restatement.
 * Batches of integer-valued views keep the f32 tiles (or r3dm_set_integer_mfma).  r3dm_stats.n_split_mfma reports which path ran. */
int r3dm_set_split_mfma(r3dm_ctx* ctx, int enable);

/* Opt-in exact MFMA formulation of the Hamming matcher for binary descriptors (default off: the default is the plain
 * xor + popcount kernel BASELINE config C3 names; no reference counterpart -- OpenMVG has one Hamming loop).  The bits of a
 * row become bytes 0 / 1, Hamming(a, b) = popcount(a) + popcount(b) - 2 a.b, and a.b runs on v_mfma_i32_32x32x32_i8 with
 * the dataset tiles shared by a workgroup through LDS; every quantity is a small integer, so indices, distances and
 * matches are bit-identical to the popcount kernel and to the CPU restatement (ties -> lowest row).
 * r3dm_stats.n_hamming_mfma reports which path ran. */
int r3dm_set_hamming_mfma(r3dm_ctx* ctx, int enable);

/* ---- putative matching ----
 * pairs_ij: n_pairs x 2 view ids (I, J); J's rows are the queries, I's rows the dataset.
 * Descriptor lengths and speed: the tensor kernels serve float / byte rows of up to 64, 128, 144 and 256 elements (rows are padded
 * to the next of these: SIFT-128, LIOP-144, SURF-64, 256-D learned descriptors) and binary rows of 29..32 / 61..64 bytes.  Any other
 * L2 length (> 256 elements) is matched by the exact scan alone -- one workgroup per query row streaming the dataset: the same
 * results, two to three orders of magnitude slower; r3dm_stats.n_exact_fallback then equals n_queries.  Lengths that are not a
 * multiple of 4 (37, 61 ...) run on the tensor kernels like any other; only their uncertified queries (a handful per million) are
 * re-done by that per-query scan instead of the batched one, because the reference's 4-way unrolled sum ends in a scalar tail.
 * dist_ratio: Lowe ratio (0.6 default in the reference, src/Regard3DFeatures.cpp:129);
 * squared_metric != 0 applies ratio^2 (RegionsMatcherT ctor flag, true for L2).
 * The result holds only non-empty pairs, ordered by (I, J), matches ordered by (i_, j_). */
int r3dm_match_pairs(r3dm_ctx* ctx, const uint32_t* pairs_ij, uint64_t n_pairs,
                     float dist_ratio, int squared_metric, r3dm_graph** out);

/* ---- geometric filter ----
 * AC-RANSAC fundamental-matrix filter over every pair of `putative`.  F_out (may be NULL) receives
 * 9 doubles (row-major F) per KEPT pair, in the order of the returned graph. */
int r3dm_filter_F(r3dm_ctx* ctx, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                  uint64_t seed, r3dm_ferror err_kind, r3dm_graph** out, double* F_out);

/* Homography AC-RANSAC filter: ImageCollectionGeometricFilter::Robust_model_estimation(
 * GeometricFilter_HMatrix_AC(4.0, 2048), putative, false) (src/R3DComputeMatches.cpp:2216-2221; run by default,
 * src/Regard3DFeatures.cpp:129-131) -- 4-point DLT, asymmetric transfer error, point-to-point NFA scale,
 * accept iff #inliers > 2.5 * 4.  H_out: 9 doubles (row-major H, x_J ~ H x_I) per KEPT pair. */
int r3dm_filter_H(r3dm_ctx* ctx, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                  uint64_t seed, r3dm_graph** out, double* H_out);

/* Essential-matrix variant (GeometricFilter_EMatrix_AC(4.0, imax_iteration), src/R3DComputeMatches.cpp:2169, default on
 * via computeEssentialMatrix_) -- 5-point solver on K^-1 x, epipolar distance in pixels through F = K2^-T E K1^-1,
 * accept iff #inliers > 2.5 * 5 -- followed by Regard3D's own overlap rule (:2175-2192): a pair is dropped when it keeps
 * fewer than min_count (50) matches or less than min_ratio (0.3) of its putative matches; pass 0 / 0 for the bare
 * OpenMVG filter.  Needs r3dm_set_intrinsics for both views of a pair; pairs without are not estimated, as in
 * E_ACRobust.hpp.  E_out: 9 doubles (row-major E, x_J^T E x_I = 0 in camera coordinates) per KEPT pair.
 * K: row-major 3x3 pinhole matrix of the view (Pinhole_Intrinsic::K()), NULL to remove; call after r3dm_set_image. */
int r3dm_set_intrinsics(r3dm_ctx* ctx, uint32_t view_id, const double* K);
int r3dm_filter_E(r3dm_ctx* ctx, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                  uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out);

/* Per-pair outcome of the last r3dm_filter_F / r3dm_filter_H call -- what OpenMVG's ACRANSAC returns besides the inliers
 * (std::pair<errorMax, minNFA>) plus work counters.  One entry per pair of the putative graph, in its
 * order; pairs with too few putatives (<= 7 for F, <= 4 for H) are all-zero.  Returns the number of entries available. */
typedef struct {
    double   threshold_px;   /* AC-RANSAC inlier threshold (pixels); 0 when no meaningful model     */
    double   nfa;            /* minimum log10 NFA (+inf when nothing was evaluated)                  */
    uint32_t iterations;     /* iterations executed                                                  */
    uint32_t models;         /* models evaluated                                                     */
    uint32_t inliers;        /* inliers found (before the > 2.5*7 acceptance rule)                   */
    uint32_t reserved;
} r3dm_pair_report;
int r3dm_filter_report(const r3dm_ctx* ctx, r3dm_pair_report* out, uint64_t cap);
/* The three filters of one putative graph side by side -- what the reference runs one after the other at src/R3DComputeMatches.cpp:
 * 2113-2120 (F), :2130-2204 (E + overlap rule) and :2216-2233 (H); they only read the putative graph.  which: bit 0 F, bit 1 E, bit 2 H
 * (the out_* of a requested filter must not be NULL).  Pairs with >= 4096 putatives of all requested filters share ONE cooperative
 * kernel (a pool of persistent workgroups; a pair's residual passes are row slices run by idle workers), shorter pairs run one
 * workgroup per pair on a stream per filter; one filter alone is served the same way (r3dm_filter_F / _E / _H).  Results are those of
 * r3dm_filter_F / _E / _H whatever the split (inlier sets, models, iteration counts: tests/test_gpu_filter_coop.py).  ms_kernels3 /
 * ms_wall3 (optional): HIP-event time of the kernels and wall time of each call in the order F, E, H (they overlap; a filter with
 * long pairs ends when the cooperative kernel does).  r3dm_filter_report afterwards: the E call's, else F's.  r3dm_stats:
 * n_filter_workgroups / n_filter_coop_pairs. */
int r3dm_filter_FEH(r3dm_ctx* ctx, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter, uint64_t seed, int which,
                    uint32_t e_min_count, float e_min_ratio, r3dm_graph** out_F, r3dm_graph** out_E, r3dm_graph** out_H,
                    double* ms_kernels3, double* ms_wall3);

/* ---- ArrayMatcher-shaped low level call: 2-NN of each query row among the dataset rows ----
 * out_idx / out_dist: 2 entries per query, ascending distance (dist: float squared L2 for F32/U8,
 * Hamming distance converted to float for BIN).  Fails (R3DM_ERR_INVALID) when n_query < 1 or
 * n_dataset < 2, like ArrayMatcherBruteForce::SearchNeighbours with NN = 2. */
int r3dm_knn2(r3dm_ctx* ctx, const void* dataset, uint32_t n_dataset, const void* query, uint32_t n_query,
              uint32_t dim, r3dm_dtype dtype, int32_t* out_idx, float* out_dist);

/* The same with the dataset staged ONCE, the way the reference uses its plugins: Build per first view I, SearchNeighbours per
 * J from an OpenMP loop (src/R3DComputeMatches.cpp:462-479; ArrayMatcher_kgraph::Build / SearchNeighbours,
 * src/utils/matcher_kgraph.h:120-166,205-251).  r3dm_index_create copies and re-lays-out `dataset` (it need not outlive the
 * call); the index belongs to the context's DEVICE, not to the context: r3dm_index_knn2 may be called with any context of
 * that device, so a host can keep a small pool of contexts and run its searches concurrently (include/r3dm_array_matcher.hpp
 * does).  Each search uploads the query rows only; r3dm_stats.n_views_staged counts the copies + re-layouts a context made. */
typedef struct r3dm_index r3dm_index;
int  r3dm_index_create(r3dm_ctx* ctx, const void* dataset, uint32_t n_dataset, uint32_t dim, r3dm_dtype dtype, r3dm_index** out);
int  r3dm_index_knn2(r3dm_ctx* ctx, const r3dm_index* index, const void* query, uint32_t n_query, int32_t* out_idx, float* out_dist);
void r3dm_index_destroy(r3dm_index* index);

/* ---- approximate matching: the KGraph plugin path (BASELINE config C5) ----
 * Replaces kgraph_match (src/R3DComputeMatches.cpp:808-902): per first view I an index over its descriptors
 * (ArrayMatcher_kgraph::Build, src/utils/matcher_kgraph.h:138-153), per query row of J a graph search for its 2
 * nearest rows (SearchNeighbours, :204-251 -> KGraphImpl::search, src/thirdparty/kgraph/kgraph.cpp:411-552), then
 * the same ratio test / de-duplication / pair rules as r3dm_match_pairs.  F32 (and U8) descriptors with
 * dim % 4 == 0 only, squared-L2 metric (kgraph_match constructs its matcher with b_squared_metric = true, :842).
 * The index is the exact index_K-nearest-neighbour graph completed with reverse edges (DESIGN.md "ANN"); it is
 * built on first use and cached with the view.  Views with fewer than 128 rows are matched exhaustively.
 * Results are deterministic: they depend on (descriptors, parameters, view ids) only. */
typedef struct {
    uint32_t index_K;    /* forward neighbours per row, 1..32  (reference: IndexParams K/L, presets 2/20 .. 16/24) */
    uint32_t search_P;   /* random start rows per query, 2..61 (reference: SearchParams P, presets 2 / 6 / 12 / 10) */
    uint32_t search_S;   /* neighbours expanded per step, 1..16 (reference: SearchParams S, default 10)              */
    uint32_t reserved;
    uint64_t seed;       /* start-row stream (reference: SearchParams seed, 1998)                                    */
} r3dm_kgraph_params;
/* preset = matchingAlgorithm of the reference: 0 "KGraph fast", 1 "medium", 2 "precise", anything else its default
 * block (src/R3DComputeMatches.cpp:844-873) */
int r3dm_kgraph_preset(int preset, r3dm_kgraph_params* out);
/* parameters with which the graph matcher serves an approximate arm of the reference's dispatch (matchingAlgorithm,
 * src/R3DComputeMatches.cpp:2035-2062: 0 FLANN kd-trees, 1..3 KGraph, 5 MRPT, 6..8 HNSW) with a recall at least that of the arm;
 * R3DM_ERR_INVALID for the exhaustive arms 4 / 9 and unknown values.  The KGraph arms are the reference's algorithm.  The HNSW
 * arms have their own entry points (r3dm_match_pairs_hnsw below: hnswlib's search on an HNSW index) -- this mapping is what a host
 * uses for them only when it wants the FASTEST matcher of at least the arm's recall, or for a descriptor length hnswlib's SIMD16
 * distance does not serve.  The MRPT arm (5) has its own entry points too since round 4 (r3dm_match_pairs_mrpt below: random
 * projection trees built and queried on the device); this mapping serves it only under the fastest-matcher policy.  FLANN (arm 0)
 * is PERMANENTLY SUBSTITUTED: no kd-tree index is built here, the deterministic graph matcher answers for it and its matches are not
 * those the reference's arm would return (both are approximate) -- FLANN is an external library that is not in the reference tree,
 * so there is nothing to restate or pin, and every approximate arm measured on this GPU is slower than the exact fast paths, under
 * which the facade's default policy serves these arms at recall 1.  The arm numbers are accepted and mapped, never rejected.
 * See the table in api_match.cpp and DESIGN.md sections 4.7 and 7. */
int r3dm_ann_params_for_algorithm(int matching_algorithm, r3dm_kgraph_params* out);
int r3dm_match_pairs_kgraph(r3dm_ctx* ctx, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                            const r3dm_kgraph_params* params, r3dm_graph** out);
/* 1 when, for the views registered in ctx, the EXHAUSTIVE matcher (r3dm_match_pairs) is expected to be at least as fast as the
 * graph matcher -- and it is exact: the views hold real-valued descriptors (LIOP-144, the descriptor Regard3D matches: the graph
 * search then gathers f32 rows and runs at 937 pairs/s on 16,384-row views where the exhaustive f32 kernel runs at 2,017 and the
 * split-f16 path at ~5,000, profiles/r02_p_ann_perf_f32rows.txt) of a length the tensor kernels serve (64 / 128 / 144 / 256) and no
 * view has more than 32,768 rows.  0 otherwise (integer-valued byte rows, where the v_dot4 graph search beats the f32 tiles; very
 * large views, where n log n beats n^2).  A host that only wants the arm's RESULT QUALITY (the reference's approximate arms 0-3,
 * 5-8 exist to save CPU time, not to lose matches) can use this to serve them with the exact matcher: the facade's default. */
int r3dm_exhaustive_is_faster(const r3dm_ctx* ctx);
/* ArrayMatcher_kgraph-shaped call: index `dataset`, 2 approximate nearest rows of every query row.  pair_i / pair_j
 * key the start-row stream (view ids in r3dm_match_pairs_kgraph).  out_idx -1 / out_dist +inf where the search
 * found fewer than two rows. */
int r3dm_kgraph_knn2(r3dm_ctx* ctx, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                     uint32_t dim, const r3dm_kgraph_params* params, uint32_t pair_i, uint32_t pair_j,
                     int32_t* out_idx, float* out_dist);
/* Rows per view the graph matcher indexes: its index is the exact K-NN graph, built by an all-pairs scan of the view (quadratic in the
 * rows: 1-4 ms at 16 k rows, ~0.3 s at this bound).  r3dm_kgraph_index / r3dm_match_pairs_kgraph / r3dm_kgraph_knn2 return
 * R3DM_ERR_UNSUPPORTED for a larger view instead of indexing it in quadratic time (the reference's NN-descent builder,
 * src/thirdparty/kgraph/kgraph.cpp:703-999, is not built); the exhaustive matcher has no such bound. */
#define R3DM_KGRAPH_MAX_ROWS 131072u
/* the index of a registered view (built if necessary): adj_out = n x 64 rows (0xFFFFFFFF padded), deg_out = n */
int r3dm_kgraph_index(r3dm_ctx* ctx, uint32_t view_id, uint32_t index_K, uint32_t* adj_out, uint32_t* deg_out);
/* forget the graph indices of all registered views (they are rebuilt on the next r3dm_match_pairs_kgraph); the reference builds
 * the index of image I inside every kgraph_match call (src/R3DComputeMatches.cpp:826-841), so a measurement that wants the same
 * accounting calls this between passes.  The registered descriptors stay. */
int r3dm_drop_indices(r3dm_ctx* ctx);

/* ---- approximate matching: the HNSW plugin path (matchingAlgorithm 6 / 7 / 8) ----
 * Replaces hnsw_match (src/R3DComputeMatches.cpp:497-593): per first view I an HNSW index over its descriptors
 * (ArrayMatcher_hnsw::Build, src/utils/matcher_hnsw.h:53-83 -> hnswlib::HierarchicalNSW, src/thirdparty/hnswlib/hnswlib/hnswalg.h),
 * per query row of J hnswlib's searchKnn(row, 2) with setEf(ef) (SearchNeighbours, matcher_hnsw.h:150-190), then the same ratio
 * test / de-duplication / pair rules as r3dm_match_pairs.  F32 / U8 descriptors of length 64, 128, 144 or 256 (hnswlib's
 * L2SqrSIMD16Ext, dim % 16 == 0), squared-L2 metric, at most 262,144 rows per view; views with fewer than 128 rows are scanned.
 *
 * SEARCH: hnswlib's algorithm step for step -- greedy descent through the upper layers, searchBaseLayerST on layer 0 with the two
 * priority queues moved as libstdc++'s heaps move them (ties between equally distant rows included), distances summed in the order
 * of hnswlib's AVX kernel.  On an index written by the reference-built library r3dm_hnsw_knn2_on_index returns hnswlib's own
 * rows and distances BIT FOR BIT (tests/test_gpu_hnsw.py against tests/golden/hnsw_ref_index.npz).
 * BUILD: hnswlib inserts rows one after another, each insertion searching the graph so far, and the reference adds rows 1 .. n-1
 * from an OpenMP loop -- its index differs from run to run.  Here the index is built in one batch, deterministically: hnswlib's
 * level draw (same engine and seed), per layer exact candidates (layer 0: the 64 closest of the exact 32-NN graph with reverse
 * edges; above: the 32 nearest members), hnswlib's neighbour-selection heuristic over them (at most 2M links on layer 0, M above),
 * free places refilled with the closest rejected candidates.  ef_construction has no role in it (accepted, ignored).  At the
 * reference's ef the batch-built index finds the true nearest row at least as often as the reference-built one (same tests;
 * CPU model: oracle/hnsw.c orc_hnsw_build_batch, GPU parity with it bit-exact). */
typedef struct {
    uint32_t M;                /* links per row above layer 0 (2M on layer 0), 2..32   (reference presets 5 / 15 / 19)        */
    uint32_t ef_construction;  /* the reference's build beam (112 / 112 / 100): unused by the batch construction             */
    uint32_t ef;               /* search beam, 1..512 (searchKnn uses max(ef, 2))      (reference presets 5 / 10 / 15)        */
    uint32_t seed;             /* level draw (hnswlib: random_seed = 100)                                                    */
} r3dm_hnsw_params;
/* preset = matchingAlgorithm - 6 of the reference: 0 "HNSW fast", 1 "medium", anything else "precise" */
int r3dm_hnsw_preset(int preset, r3dm_hnsw_params* out);
int r3dm_match_pairs_hnsw(r3dm_ctx* ctx, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                          const r3dm_hnsw_params* params, r3dm_graph** out);
/* ArrayMatcher_hnsw-shaped call: index `dataset` (>= 128 rows), searchKnn(row, 2) of every query row.  out_idx -1 / out_dist +inf
 * where the search found fewer than two rows. */
int r3dm_hnsw_knn2(r3dm_ctx* ctx, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                   uint32_t dim, const r3dm_hnsw_params* params, int32_t* out_idx, float* out_dist);
/* An HNSW index as arrays in hnswlib's own shape: links0 = n x (1 + 2M) ints (count, then links, -1 padded), up_off = n + 1 ints
 * (first upper-layer row of a node; layer L >= 1 of node i is row up_off[i] + L - 1), up_links = up_rows x (1 + M) ints. */
typedef struct {
    uint32_t M;
    const int32_t* links0;
    const int32_t* up_off;
    const int32_t* up_links;
    uint32_t up_rows;
    int32_t enterpoint, maxlevel;
} r3dm_hnsw_arrays;
/* searchKnn(row, 2) with setEf(ef) on an index handed over as arrays -- e.g. one written by hnswlib itself (any n_dataset >= 2).
 * The arrays are validated (R3DM_ERR_INVALID when a link or an offset is out of range). */
int r3dm_hnsw_knn2_on_index(r3dm_ctx* ctx, const float* dataset, uint32_t n_dataset, uint32_t dim, const r3dm_hnsw_arrays* index,
                            const float* query, uint32_t n_query, uint32_t ef, int32_t* out_idx, float* out_dist);
/* the index of a registered view (built if necessary) in that shape; up_links holds up_cap rows, *up_rows receives the number needed */
int r3dm_hnsw_index(r3dm_ctx* ctx, uint32_t view_id, const r3dm_hnsw_params* params, int32_t* links0, int32_t* up_off,
                    int32_t* up_links, uint32_t up_cap, uint32_t* up_rows, int32_t* enterpoint, int32_t* maxlevel);

/* ---- MRPT plugin path (matchingAlgorithm 5) ----
 * mrpt_match (src/R3DComputeMatches.cpp:423-491): per first view I an index of n_trees random projection trees of depth `depth` over
 * its descriptors (ArrayMatcher_mrpt::Build, src/utils/matcher_mrpt.h:76-128 -> Mrpt::grow, src/thirdparty/mrpt/mrpt.h:84-137: sparse
 * random vectors of density 1 / sqrt(dim), every tree level splits its nodes at the median projection), per query row of J
 * Mrpt::query(row, 2, votes) (mrpt.h:661-728: one leaf per tree, the rows that collect >= votes votes are measured exactly, the two
 * nearest kept; once more with votes - 1 when fewer than two rows were elected, matcher_mrpt.h:224-232; a query that still has none is
 * dropped), then the ratio test ON THE SQUARE ROOTS the index returns with the un-squared ratio (RegionsMatcherT(regions, false),
 * src/R3DComputeMatches.cpp:461) and the de-duplication / pair rules every arm shares.  The index is built on the device
 * (kernels_mrpt.hip); parity is with the CPU model oracle/mrpt.c bit for bit.  What cannot match a reference build -- and could not
 * be pinned to one, mrpt.h being Eigen code: the random vectors are drawn from a counter-based stream instead of std::mt19937 with
 * implementation-defined distributions (same density, same N(0, 1) values), rows whose projection EQUALS a node's median go left in
 * (projection, row) order where std::nth_element leaves them anywhere, candidates are measured with the reference's brute-force
 * metric in its summation order.  Views with fewer than 128 rows are scanned exhaustively; descriptor lengths: any multiple of 4
 * up to 512; at most 131,072 rows per indexed view. */
typedef struct {
    uint32_t n_trees;          /* 1..255                                   (reference: 26)                                         */
    uint32_t depth;            /* 1..6; clamped per view to max(2, min(depth, floor(log2 n) - 1)) as ArrayMatcher_mrpt::Build does   (6) */
    uint32_t votes;            /* rows with at least this many of the n_trees votes are candidates, 1..n_trees   (5)               */
    float    density;          /* share of non-zero entries of a random vector; <= 0: 1 / sqrt(dim), mrpt.h's default   (0.088: set but never passed on by the reference) */
    uint64_t seed;             /* of the counter-based stream the random vectors are drawn from                                     */
} r3dm_mrpt_params;
int r3dm_mrpt_preset(r3dm_mrpt_params* out);               /* 26 / 6 / 5 / default density / seed 0 */
int r3dm_match_pairs_mrpt(r3dm_ctx* ctx, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                          const r3dm_mrpt_params* params, r3dm_graph** out);
/* ArrayMatcher_mrpt-shaped call: index `dataset` (>= 128 rows), the two nearest elected rows of every query row.  out_dist = the
 * SQUARE ROOTS of the squared L2 distances, as Mrpt::query returns them; out_idx -1 / out_dist -1 for a query the reference drops. */
int r3dm_mrpt_knn2(r3dm_ctx* ctx, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                   uint32_t dim, const r3dm_mrpt_params* params, int32_t* out_idx, float* out_dist);
/* the index of a registered view (built if necessary): R = n_trees * depth x dim (zeros where the sparse matrix has no entry),
 * splits = n_trees x (2^depth - 1) in heap order, leaves = n_trees x n rows leaf after leaf, leaf_first = 2^depth + 1 offsets;
 * *depth_out = the clamped depth.  Any output may be NULL.  The arrays must hold the sizes of depth = params->depth. */
int r3dm_mrpt_index(r3dm_ctx* ctx, uint32_t view_id, const r3dm_mrpt_params* params, float* R, float* splits, int32_t* leaves,
                    int32_t* leaf_first, uint32_t* depth_out);

/* ---- keypoint detection: Fast-A-KAZE ----
 * The "Fast-AKAZE" arm of Regard3DFeatures::detectKeypoints (src/Regard3DFeatures.cpp:596-617): cv::AKAZE2::create() with its
 * defaults (src/thirdparty/fast-akaze/AKAZEConfig.h:18-43: 4 octaves x 4 sublevels, PM_G2 diffusivity), setThreshold(threshold),
 * detect() = nonlinear scale space by fast explicit diffusion + determinant-of-Hessian extrema + sub-pixel refinement + main
 * orientation (src/thirdparty/fast-akaze/AKAZEFeatures.cpp:245-382), then the angle conversion of :604-613.
 * image: height x width floats in [0, 1] (gray / 255, as R3DFeaturesThread.cpp:163-191 prepares it), host or device pointer.
 * keypoints_out: cap x 4 floats (x, y, size, angle in degrees) -- exactly what r3dm_extract_liop takes; responses_out optional.
 * *n_out = number detected (may exceed cap; only the first cap are written, in the reference's order: by evolution level, then
 * by detection order).  The reference serialises this stage with a semaphore (one image at a time). */
int r3dm_detect_akaze(r3dm_ctx* ctx, const float* image, uint32_t width, uint32_t height, float threshold,
                      float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out);

/* The same over a BATCH of n_images same-size images in ONE pass of the detector (no reference counterpart: the reference admits one
 * image at a time into this stage, src/Regard3DFeatures.cpp:71-125).  The ~550 dependent launches of the scale space depend on the
 * image size only, so they serve all images at once (one image alone waits for launch latency below the first octave), and nothing
 * in the chain visits the host.  images[b]: height x width floats, host or device; keypoints_out[b]: cap x 4
 * floats; responses_out: NULL, or n_images pointers (each may be NULL); n_out[b] as above.  Results per image are bit-identical
 * to r3dm_detect_akaze on that image. */
int r3dm_detect_akaze_batch(r3dm_ctx* ctx, uint32_t n_images, const float* const* images, uint32_t width, uint32_t height,
                            float threshold, float* const* keypoints_out, float* const* responses_out, uint32_t cap, uint32_t* n_out);

/* cv::AKAZE2::detectAndCompute with DESCRIPTOR_MLDB (src/thirdparty/fast-akaze/akaze.cpp:171-221; the binary descriptors of
 * BASELINE config C3): keypoints as above plus the full MLDB descriptor of each -- 3 grids (2x2, 3x3, 4x4) x 3 channels, all
 * pairwise comparisons = 486 bits packed LSB first into 61 bytes (AKAZEFeatures.cpp:1790-1909).  descriptors_out: cap x 61
 * bytes, ready for r3dm_set_image(..., dim 61, R3DM_BIN). */
int r3dm_detect_akaze_mldb(r3dm_ctx* ctx, const float* image, uint32_t width, uint32_t height, float threshold,
                           float* keypoints_out, unsigned char* descriptors_out, uint32_t cap, uint32_t* n_out);

/* ---- the per-image work item of the features stage ----
 * R3DFeaturesThread::processWorkItem (src/threads/R3DFeaturesThread.cpp:123-210) after cv::imread: 8-bit BGR -> float / 255 ->
 * BGR2GRAY (r3dm_gray_from_bgr8; bgr = height x width x 3 bytes, gray_out = height x width floats, host or device), then
 * Regard3DFeatures::detectAndExtract with the "Fast-AKAZE" detector + LIOP and KeypointSet::saveToBinFile
 * (src/keypointSet.hpp:61-67): <feat_path> gets one "x y scale orientation" line per feature (scale = size / 2), <desc_path>
 * an 8-byte count followed by count x 144 floats -- the files r3dm_compute_matches_dir / the facade read back.
 * As in the reference (:139-142) the work item does nothing when BOTH files already exist (n_features = rows of the existing .desc).
 * Image decoding (cv::imread) stays with the caller. */
int r3dm_gray_from_bgr8(r3dm_ctx* ctx, const unsigned char* bgr, uint32_t width, uint32_t height, float* gray_out);
int r3dm_extract_features_to_files(r3dm_ctx* ctx, const float* gray, uint32_t width, uint32_t height, float threshold,
                                   const char* feat_path, const char* desc_path, uint32_t* n_features);

/* n_images same-size images through the work item in one pass: detector batch (above) -> the angle and LIOP patch map of every
 * keypoint on the host (atan2f / cos / sin of the host libm, as the reference; 32 bytes per keypoint up, 24 down) -> ONE patch
 * extraction launch and ONE LIOP launch over the keypoints of all images -> files.  grays: n_images pointers to height x width
 * floats, or NULL and bgrs: n_images pointers to height x width x 3 bytes (BGR as cv::imread decodes; converted on the device as
 * r3dm_gray_from_bgr8 does); host or device.  No skip rule here: every image is computed and its two files (re)written. */
int r3dm_extract_features_batch(r3dm_ctx* ctx, uint32_t n_images, const float* const* grays, const unsigned char* const* bgrs,
                                uint32_t width, uint32_t height, float threshold, const char* const* feat_paths,
                                const char* const* desc_paths, uint32_t* n_features);

/* ---- descriptor extraction: LIOP on pre-extracted patches ----
 * r3d_vl_liopdesc_process of the vendored VLFeat copy (src/thirdparty/liop/vl_liop.c:465-580) as Regard3D
 * calls it per keypoint (src/Regard3DFeatures.cpp:727-752,827: new_basic(41) -> 4 neighbours, 6 bins, radius 6):
 * patches = n x 41 x 41 floats (the warped + blurred patch of every keypoint, row-major), desc_out = n x 144
 * floats, non-negative, unit L2 norm (all-zero for a constant patch).  Host or device pointers.
 * n_resorted (optional) receives the number of patches with equal intensities that went through the exact
 * re-sort.  Results are bit-identical to the reference routine. */
int r3dm_liop_describe_patches(r3dm_ctx* ctx, const float* patches, uint32_t n, uint32_t side, float* desc_out,
                               uint32_t* n_resorted);

/* ---- descriptor extraction: keypoints -> LIOP descriptors ----
 * The per-keypoint loop of Regard3DFeatures::extractLIOPFeatures (src/Regard3DFeatures.cpp:719-861; serial in the
 * reference): for every keypoint a 41x41 patch by inverse affine warp (scale = size/41 * kpSizeFactor,
 * angle = -90 - kp.angle, :768-803) + Gaussian blur sigma 1.2 (:807), then LIOP.  image: h x w floats (gray / 255,
 * as R3DFeaturesThread.cpp:163-191 prepares it); keypoints: n x 4 floats (x, y, size, angle in degrees) as left by
 * detectKeypoints (:574-684); kp_size_factor: getKpSizeFactor(detector) (:691-717, 8.0 for A-KAZE).
 * desc_out: n x 144 floats.  patches_out (optional): n x 41 x 41 floats.  Host or device pointers. */
int r3dm_extract_liop(r3dm_ctx* ctx, const float* image, uint32_t width, uint32_t height,
                      const float* keypoints, uint32_t n, float kp_size_factor, float* desc_out, float* patches_out);

/* ---- match graph (PairWiseMatches) ---- */
uint64_t          r3dm_graph_num_pairs(const r3dm_graph* g);
uint64_t          r3dm_graph_num_matches(const r3dm_graph* g);
const uint32_t*   r3dm_graph_pairs(const r3dm_graph* g);     /* n_pairs x 2                        */
const uint64_t*   r3dm_graph_offsets(const r3dm_graph* g);   /* n_pairs + 1 (CSR into matches)     */
const r3dm_match* r3dm_graph_matches(const r3dm_graph* g);
void              r3dm_graph_free(r3dm_graph* g);
/* build a graph from host CSR arrays (copied); used to re-assemble shards after the all-gather */
int r3dm_graph_from_csr(const uint32_t* pairs_ij, uint64_t n_pairs, const uint64_t* offsets,
                        const r3dm_match* matches, r3dm_graph** out);
/* merge several graphs (e.g. one per rank) into one ordered by (I, J) */
int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);

/* ---- multi-GPU, single process: one context per device, one host thread per device ----
 * For a C++ host like Regard3D itself (one process, /root/reference/src/R3DComputeMatches.cpp:437-489 treats the pairs of the
 * collection as independent OpenMP iterations; so does OpenMVG's filter loop, :2099).  r3dm_multi_create opens one context per
 * entry of device_ids (an id may repeat: two contexts on one GPU); r3dm_multi_set_image replicates a view on every device;
 * r3dm_multi_match_pairs deals the pair list to the devices by rows of I in snake order (r3dm_shard_pairs: rows sorted by
 * decreasing pair count, dealt 0..W-1, W-1..0, ... -- cost-balanced, the pairs of one I stay on one device), runs
 * r3dm_match_pairs on every device from its own host thread and merges the graphs; the filters deal the putative pairs
 * longest list first.  No collective is involved (the graph is reassembled in host memory, where PairWiseMatches lives);
 * the multi-PROCESS route -- one rank per GPU, one RCCL all-gather -- is regard3d_amd/dist.py.  Results are identical to a
 * single-device run (every per-pair computation is independent of the device that runs it). */
typedef struct r3dm_multi r3dm_multi;
int  r3dm_multi_create(const int* device_ids, int n_dev, r3dm_multi** out);
void r3dm_multi_destroy(r3dm_multi* m);
int  r3dm_multi_num_devices(const r3dm_multi* m);
r3dm_ctx* r3dm_multi_ctx(r3dm_multi* m, int k);            /* the k-th device's context (statistics, reports) */
const char* r3dm_multi_last_error(const r3dm_multi* m);
int r3dm_multi_set_image(r3dm_multi* m, uint32_t view_id, uint32_t width, uint32_t height,
                         const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy);
/* how the views registered so far travelled: a view crosses PCIe once (host -> the first context's device; not at all when the
 * caller's buffers are device memory) and reaches the other devices by hipMemcpyPeerAsync (xGMI where peer access exists) */
int r3dm_multi_transfer_counts(const r3dm_multi* m, uint64_t* host_uploads, uint64_t* peer_copies);
int r3dm_multi_set_intrinsics(r3dm_multi* m, uint32_t view_id, const double* K);
int r3dm_multi_clear_images(r3dm_multi* m);
int r3dm_multi_set_integer_mfma(r3dm_multi* m, int enable);
int r3dm_multi_match_pairs(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs,
                           float dist_ratio, int squared_metric, r3dm_graph** out);
/* the same deal for the graph matcher (kgraph_match, config C5): a device builds the index of every image I whose row it owns */
int r3dm_multi_match_pairs_kgraph(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                  const r3dm_kgraph_params* params, r3dm_graph** out);
/* ... and for the HNSW matcher (hnsw_match, matchingAlgorithm 6..8) */
int r3dm_multi_match_pairs_hnsw(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                const r3dm_hnsw_params* params, r3dm_graph** out);
/* ... and for the MRPT matcher (mrpt_match, matchingAlgorithm 5) */
int r3dm_multi_match_pairs_mrpt(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                const r3dm_mrpt_params* params, r3dm_graph** out);
int r3dm_multi_filter_F(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                        uint64_t seed, r3dm_graph** out, double* F_out);
int r3dm_multi_filter_H(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                        uint64_t seed, r3dm_graph** out, double* H_out);
int r3dm_multi_filter_E(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                        uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out);
/* The features stage over an image list: R3DFeaturesThread::extractFeaturesAndDescriptors (src/threads/R3DFeaturesThread.cpp:38-121),
 * whose worker pool pulls images off a work list and runs processWorkItem on each.  Here every context of `m` is a worker that pulls
 * BATCHES of same-size images (r3dm_extract_features_batch: up to 8 per detector pass): create
 * the r3dm_multi with one device id repeated K times to keep K batches in flight on that GPU (K streams + K sets of work buffers;
 * the reference admits one image at a time into the A-KAZE scale space, src/Regard3DFeatures.cpp:71-125 -- HBM does not need that),
 * or with several device ids to spread the list over the GPUs of a node.  Images whose <feat> AND <desc> files already exist are
 * skipped like processWorkItem does (:139-142); skipped[i] (optional) says so and n_features[i] (optional) then carries the row
 * count of the existing .desc.  grays[i]: height x width floats, host or device.  err (optional) receives the first failure. */
int r3dm_multi_extract_features(r3dm_multi* m, uint32_t n_images, const float* const* grays, const uint32_t* widths,
                                const uint32_t* heights, float threshold, const char* const* feat_paths,
                                const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped, char* err, size_t err_cap);
/* the general form: per image either grays[i] (gray / 255 floats) or bgrs[i] (decoded 8-bit BGR, what cv::imread hands
 * processWorkItem, src/threads/R3DFeaturesThread.cpp:163-165: the float / 255 -> BGR2GRAY conversion then runs on the device);
 * either array may be NULL as a whole.  batch: images per detector pass of a worker, 0 = the default (8; fewer when HBM or the list
 * is short).  A worker's batch is a run of images of one size and one kind. */
int r3dm_multi_extract_features_ex(r3dm_multi* m, uint32_t n_images, const float* const* grays, const unsigned char* const* bgrs,
                                   const uint32_t* widths, const uint32_t* heights, float threshold, const char* const* feat_paths,
                                   const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped, uint32_t batch,
                                   char* err, size_t err_cap);
/* owner_out[p] = the device / rank (0..world-1) that the snake deal gives pair p.  Pure host code (no GPU needed). */
int r3dm_shard_pairs(const uint32_t* pairs_ij, uint64_t n_pairs, uint32_t world, uint32_t* owner_out);

/* ---- multi-GPU, one process per GPU: the ONE collective of the path ----
 * Every rank matches and filters its share of the pair list (r3dm_shard_pairs) and then calls r3dm_allgather_graphs: one
 * ncclAllGather of the per-rank sizes (8 bytes each) and one of the packed graphs padded to the largest rank, over RCCL (xGMI between
 * the GPUs of a node); every rank ends up with the graphs of the whole collection, ordered by (I, J) -- the std::map the reference's
 * OpenMP threads fill under `omp critical` (/root/reference/src/R3DComputeMatches.cpp:465,481-487).  The communicator is RCCL's own:
 * rank 0 draws an id (r3dm_comm_unique_id, 128 bytes) and hands it to the other ranks by whatever the host has (MPI_Bcast, a file,
 * torch.distributed), every rank calls r3dm_comm_create(id, rank, world, device).  RCCL is bound at run time (dlopen): without
 * librccl.so these calls return R3DM_ERR_UNSUPPORTED, they never fall back to another transport.
 * r3dm_graphs_pack / r3dm_graphs_unpack_merge expose the wire format (uint32 words: [n_graphs, len_k ..] then per graph
 * [P, M_lo, M_hi, pairs, counts, matches]) for hosts that bring their own transport; words from r3dm_graphs_pack are released with
 * r3dm_words_free. */
typedef struct r3dm_comm r3dm_comm;
int  r3dm_comm_unique_id(void* id_out_128);
int  r3dm_comm_create(const void* id_128, int rank, int world, int device_id, r3dm_comm** out);
void r3dm_comm_destroy(r3dm_comm* comm);
int  r3dm_comm_rank(const r3dm_comm* comm);
int  r3dm_comm_world(const r3dm_comm* comm);
const char* r3dm_comm_last_error(const r3dm_comm* comm);
int  r3dm_allgather_graphs(r3dm_comm* comm, const r3dm_graph* const* local, uint32_t n_graphs, r3dm_graph** merged_out /* [n_graphs] */);
/* Device-resident graphs for that exchange.  With r3dm_set_device_graphs(ctx, 1) every graph the context produces from then on
 * (r3dm_match_pairs*, r3dm_filter_F / _E / _H / _FEH) keeps, beside its host vectors, the same CSR in the memory of the context's GPU,
 * laid out by gather kernels where the matches already are (the finalisation's output array; the putative matches through the filters'
 * inlier indices).  r3dm_allgather_graphs sends such a graph from device memory -- its payload does not cross PCIe on the way out --
 * when the mirror lives on the communicator's device; other graphs are packed on the host as before (r3dm_graph_from_csr, loaded or
 * merged graphs have no mirror).  r3dm_graph_on_device: the device id of a graph's mirror, -1 without one;
 * r3dm_comm_last_device_graphs: how many local graphs of the last exchange went out that way.  The approximate matchers
 * (r3dm_match_pairs_kgraph / _hnsw / _mrpt) merge two part graphs on the host -- graph-searched pairs and exhaustively scanned small
 * ones: their result keeps a mirror when all of its pairs are of one kind (the usual case), none otherwise.
 * A rank takes part in every collective of an exchange whatever happened while it prepared its share -- graphs it cannot pack, buffers
 * it cannot allocate, copies into its send buffer that fail are reported in the words it contributes -- so all ranks return together;
 * not covered: a device that is gone (hipSetDevice, the size collective's own 16 (W + 1) bytes, a failing collective). */
int  r3dm_set_device_graphs(r3dm_ctx* ctx, int enable);
int  r3dm_graph_on_device(const r3dm_graph* g);
int  r3dm_comm_last_device_graphs(const r3dm_comm* comm);
int  r3dm_graphs_pack(const r3dm_graph* const* local, uint32_t n_graphs, uint32_t** words_out, uint64_t* n_words_out);
void r3dm_words_free(uint32_t* words);
int  r3dm_graphs_unpack_merge(const uint32_t* const* rank_words, const uint64_t* rank_n_words, uint32_t world, uint32_t n_graphs,
                              r3dm_graph** merged_out /* [n_graphs] */);

/* ---- files: ".txt" (what Regard3D's consumers read) or ".bin" (cereal portable binary) ---- */
int r3dm_save_matches(const r3dm_graph* g, const char* path);
int r3dm_load_matches(const char* path, r3dm_graph** out);

/* ---- run statistics of the last r3dm_match_pairs / r3dm_filter_F call ---- */
typedef struct {
    double   ms_match_kernels;     /* HIP-event time of the 2-NN kernels (dominant kernel)        */
    uint64_t n_match_launches;     /* launches of the dominant kernel                              */
    double   ms_filter_kernels;    /* HIP-event time of the AC-RANSAC kernel                       */
    uint64_t n_pairs;              /* pairs processed                                              */
    uint64_t n_queries;            /* query rows processed                                         */
    uint64_t n_exact_fallback;     /* queries re-done by the exact scan (uncertified top-2)        */
    double   algorithmic_flops;    /* 2 * nI * nJ * D summed over pairs (L2) / lane-ops (Hamming)   */
    double   algorithmic_bytes;    /* compulsory HBM bytes: both descriptor sets once + results     */
    /* host wall-clock breakdown of the last calls (milliseconds) */
    double   ms_wall_match;        /* whole r3dm_match_pairs call                                   */
    double   ms_wall_match_post;   /* of which: exact scans + finalisation + copies back + assembly */
    double   ms_wall_filter;       /* whole r3dm_filter_F call                                      */
    double   ms_liop_kernel;       /* HIP-event time of the last r3dm_liop_describe_patches kernel  */
    /* r3dm_match_pairs_kgraph */
    double   ms_ann_build;         /* HIP-event time of the index builds triggered by the call      */
    double   ms_ann_search;        /* HIP-event time of the search kernel                           */
    uint64_t n_ann_built;          /* indices built by the call                                     */
    uint64_t n_ann_dist;           /* descriptor distances evaluated by the searches                */
    double   ms_detect;            /* wall time of the last r3dm_detect_akaze call                  */
    uint64_t n_integer_mfma;       /* launches of the dominant kernel that ran as the integer fast path  */
    uint64_t n_split_mfma;         /* launches of the dominant kernel that ran as the split-f16 nominator */
    uint64_t n_views_staged;       /* views / datasets / query sets copied + re-laid-out by this context since r3dm_create */
    uint64_t n_hamming_mfma;       /* launches of the Hamming matcher that ran as the MFMA formulation */
    uint64_t n_detect_images;      /* images of the last detector pass (r3dm_detect_akaze: 1; the batch entries: B)                   */
    uint64_t n_ann_rows16;         /* graph-search launches that gathered the bf16 row copy (integer-valued views: same distances, half the bytes) */
    uint64_t n_ann_rows8;          /* ... the u8 row copy (integers 0 .. 255: a quarter of the bytes)                                          */
    uint64_t n_ann_dot8;           /* ... of those, launches whose query views are bytes too: distances as exact integer dot products (v_dot4_u32_u8) */
    /* the last detector pass (r3dm_detect_akaze*, r3dm_extract_features_*) */
    double   ms_detect_kernels;    /* HIP-event time from the first scale-space launch to the keypoint compaction, all images of the pass */
    double   detect_algorithmic_bytes; /* HBM bytes the pass structure implies: every stencil pass reads / writes whole image planes once */
    double   ms_liop_wall;         /* host + device time of the LIOP half of the last features pass (patch maps, two launches, copy back) */
    double   ms_feature_files;     /* time spent writing the .feat / .desc files of the last features pass                              */
    /* r3dm_match_pairs_hnsw (ms_ann_build / ms_ann_search / n_ann_built / n_ann_dist are shared with the KGraph path) */
    uint64_t n_hnsw_launches;      /* launches of the HNSW search kernel                                                                */
    uint64_t n_hnsw_retries;       /* ... of those, repeats because a query's candidate heap outgrew its LDS room                       */
    uint64_t n_counts_mfma;        /* launches of the split nominator that ran on COUNT tiles (rows = small integers x a row scale: LIOP; one f16 MFMA per 16 dimensions instead of three) */
    double   detect_compulsory_bytes; /* the last detector pass: bytes a perfectly fused level would still move (smoothed plane in + out, determinant out,
                                       * conductivity out, 8 bytes per pixel and FED step); detect_algorithmic_bytes is the as-structured count            */
    uint64_t n_filter_workgroups;  /* workgroups of the last AC-RANSAC call (all its filters): pool workers of the cooperative kernel + one per short pair */
    uint64_t n_filter_coop_pairs;  /* ... (pair, filter) items that ran on the cooperative kernel                                                          */
} r3dm_stats;
int r3dm_get_stats(const r3dm_ctx* ctx, r3dm_stats* out);

/* ---- totals of the features work this context has done since r3dm_create (never reset: a stage that runs many batches on several
 * contexts adds them up; R3DFeaturesThread keeps doneCount_ / numberOfKeypoints_ the same way, src/threads/R3DFeaturesThread.cpp:32-36) ---- */
typedef struct {
    uint64_t n_images;                 /* images through the detector                                                         */
    uint64_t n_passes;                 /* detector passes (batches)                                                            */
    uint64_t n_keypoints;              /* keypoints found                                                                      */
    uint64_t n_regrows;                /* detection phases repeated because an image had more candidates than slots            */
    double   ms_detect_kernels;        /* HIP-event time, first scale-space launch .. keypoint compaction                      */
    double   detect_algorithmic_bytes; /* HBM bytes the pass structure implies (every stencil pass moves whole planes once)    */
    double   ms_liop_kernels;          /* HIP-event time of the LIOP patch extraction + descriptor launches                     */
    double   ms_wall;                  /* wall time inside the features entry points (detector + LIOP + copies + files)        */
    double   ms_files;                 /* of which: writing .feat / .desc                                                      */
} r3dm_features_totals;
int r3dm_get_features_totals(const r3dm_ctx* ctx, r3dm_features_totals* out);
/* How many helper threads a host should start beside this library: half of the cores the process may really use -- the affinity mask
 * AND the cgroup CPU quota (a container can show 256 processors and own 16; more runnable threads than that are throttled for the rest
 * of the scheduler period, which looks like random 60-100 ms stalls) -- at most `want`.  The library sizes its own teams with it. */
int r3dm_host_threads(int want);

/* A sink for the features entry points (r3dm_extract_features_batch, r3dm_multi_extract_features*): called once per COMPUTED image
 * (not for skipped ones), from a helper thread of the library (several images of a batch may arrive concurrently: the sink must be
 * thread-safe), after the image's two files are written.  desc_device = n_features x 144 floats
 * in DEVICE memory of the computing context (valid until the sink returns); xy_as_written = n_features x 2 floats exactly as a reader
 * of the .feat file parses them (the file holds 6 significant digits).  What Regions_Provider::load would read back from the files
 * (src/R3DComputeMatches.cpp:2040) is thus handed over without the round trip through the file system and the PCIe bus: the facade
 * registers the view with the matcher straight from it (r3dm_set_image accepts device pointers).  image_index = the index in the
 * arrays of the call.  A non-zero return fails the features call with R3DM_ERR_INVALID.  NULL removes the sink. */
typedef int (*r3dm_features_sink)(void* user, uint32_t image_index, uint32_t n_features, const float* desc_device, const float* xy_as_written);
int r3dm_set_features_sink(r3dm_ctx* ctx, r3dm_features_sink sink, void* user);
int r3dm_multi_set_features_sink(r3dm_multi* m, r3dm_features_sink sink, void* user);

/* Deferred feature files.  The reference's feature thread returns when KeypointSet::saveToBinFile has written the image's .feat /
 * .desc (src/threads/R3DFeaturesThread.cpp:139-170, src/keypointSet.hpp:61-67); with on != 0 a features call of this context returns
 * when the images are COMPUTED and handed to the sink -- the two fwrites of every image of a batch (16 MB of descriptors per 28 k
 * keypoints) run on a writer thread of the context, beside the caller's next step (the facade: the match phase) and beside the
 * context's next batch.  r3dm_features_files_wait joins the writer and reports its I/O error, if any (R3DM_ERR_IO; the message in
 * r3dm_last_error); a context joins its writer by itself before it needs the landing buffer again, when the mode is switched
 * off, and in r3dm_destroy.  The files are complete when the wait returns R3DM_OK -- call it before anything reads them.
 * Off by default: without it every features entry point returns with its files written, as before. */
int r3dm_set_deferred_feature_files(r3dm_ctx* ctx, int on);
/* The library's background host threads (the deferred feature-file writers of a context; the facade's match-file writers and map
 * builders) run at the host's default priority unless asked: nice_value 1 .. 19 makes them stand back behind the caller's own threads
 * (Linux per-thread nice; useful when the process lives under a CPU quota it can exhaust), 0 (default) leaves priorities alone. */
int r3dm_set_background_nice(r3dm_ctx* ctx, int nice_value);
int r3dm_multi_set_background_nice(r3dm_multi* m, int nice_value);
int r3dm_features_files_wait(r3dm_ctx* ctx);
int r3dm_multi_set_deferred_feature_files(r3dm_multi* m, int on);
int r3dm_multi_features_files_wait(r3dm_multi* m, char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif
