/*
 * r3d_oracle.h -- CPU restatement ("oracle") of Regard3D's compute-matches hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / CPU baseline -- never as a fallback for the HIP path.
 *
 * What it restates (reference = /root/reference, rhiestan/Regard3D):
 *   - the per-I / OpenMP-over-J / critical-insert loop nest of the matcher drivers
 *     (src/R3DComputeMatches.cpp:423-491 mrpt_match, :502-597 hnsw_match, :808-902
 *     kgraph_match -- all share the skeleton) and the dispatch + save + F-filter calls
 *     (src/R3DComputeMatches.cpp:2035-2129);
 *   - the ArrayMatcher plugin contract (src/utils/matcher_kgraph.h:120-251):
 *     Build / SearchNeighbours, NN entries per query in ascending distance order;
 *   - the file formats named in src/R3DProject.cpp:854-871 and src/keypointSet.hpp:49-67.
 *
 * The arithmetic itself (squared-L2 metric, brute-force 2-NN, distance-ratio test,
 * AC-RANSAC with the 7-point fundamental solver, matches.*.txt/.bin writers) lives in
 * OpenMVG 1.4, an EXTERNAL dependency that is NOT vendored under /root/reference
 * (FIND_PACKAGE(OpenMVG REQUIRED), src/CMakeLists.txt:485; "OpenMVG 1.4", README.md:19).
 * Those parts restate OpenMVG's published algorithm and anchor on
 * the reference's own call sites.  The reference ships NO tests or golden vectors for this
 * path, so:
 *
 *      >>>  PARITY UNPINNED by reference tests.  <<<
 *
 * The oracle is instead pinned by (a) analytic known-answer tests (tests/test_oracle_*.py),
 * (b) hnswlib::BruteforceSearch compiled from the reference's vendored copy
 * (oracle/_ref, built by oracle/Makefile; fixtures in tests/golden/), and
 * (c) numpy/scipy second opinions (SVD null space, polynomial roots, exact rational L2).
 *
 * Restatement decisions where OpenMVG leaves behaviour unspecified (documented in DESIGN.md):
 *   - equal distances: the lowest dataset row index wins (OpenMVG's partial sort is unstable);
 *   - matches of a pair are emitted sorted by (i_, j_); coordinate de-duplication keeps the
 *     smallest (i_, j_) of every group of matches with identical (xI,yI,xJ,yJ);
 *   - squared L2 is evaluated without FMA contraction (build with -ffp-contract=off);
 *   - AC-RANSAC draws its minimal samples from a counter-based generator keyed by
 *     (seed, I, J, iteration, attempt) instead of std::mt19937, so the sample stream is
 *     identical on any host or device and independent of how pairs are batched/sharded;
 *   - the 7-point solver takes the 2-D null space from a Householder QR of A^T.
 */
#ifndef R3D_ORACLE_H
#define R3D_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t i, j; } orc_match;

/* ---- metrics (OpenMVG matching/metric.hpp L2<T>, metric_hamming.hpp) ---- */
float    orc_l2sq_f32(const float* a, const float* b, size_t n);
float    orc_l2sq_u8(const uint8_t* a, const uint8_t* b, size_t n);
uint32_t orc_hamming(const uint8_t* a, const uint8_t* b, size_t nbytes);

/* ---- brute-force 2-NN (OpenMVG ArrayMatcherBruteForce::SearchNeighbours) ----
 * dataset: nI rows, query: nJ rows, row-major.  idx/dist hold 2 entries per query, ascending.
 * Returns 0, or -1 when the call would fail in OpenMVG (nJ < 1 or nI < 2). */
int orc_knn2_l2_f32(const float* dataset, int nI, const float* query, int nJ, int dim,
                    int32_t* idx, float* dist);
int orc_knn2_l2_u8(const uint8_t* dataset, int nI, const uint8_t* query, int nJ, int dim,
                   int32_t* idx, float* dist);
int orc_knn2_hamming(const uint8_t* dataset, int nI, const uint8_t* query, int nJ, int nbytes,
                     int32_t* idx, uint32_t* dist);

/* ---- MatchDistanceRatio (OpenMVG RegionsMatcherT; call sites
 * src/R3DComputeMatches.cpp:479,585,890).  dtype: 0 = f32 L2, 1 = u8 L2, 2 = binary Hamming.
 * dim = floats / bytes per row.  xyI/xyJ (n x 2 floats) may be NULL (no coordinate dedup).
 * out must hold nJ entries; returns the number of matches written (sorted by (i,j)). */
int orc_match_distance_ratio(int dtype, const void* descI, int nI, const float* xyI,
                             const void* descJ, int nJ, const float* xyJ, int dim,
                             float dist_ratio, int squared_metric, orc_match* out);

/* The ratio test + de-duplications alone, on a 2-NN table computed elsewhere (the ANN drivers). */
int orc_ratio_dedup_f32(const int32_t* idx, const float* dist, int nJ, const float* xyI, const float* xyJ,
                        float dist_ratio, int squared_metric, orc_match* out);

/* ---- collection matcher: the reference loop nest (src/R3DComputeMatches.cpp:428-489).
 * images: arrays of length n_images.  pairs: n_pairs x 2 (I,J).  Results are returned in a
 * CSR over the INPUT pair order: counts[p] matches for pair p, concatenated in `out`
 * (capacity out_cap entries).  Returns total matches or -1 if out_cap is too small.
 * OpenMP: serial over I, parallel-for schedule(dynamic) over the J's of one I. */
int64_t orc_match_collection(int dtype, int n_images, const void* const* desc, const int* n_rows,
                             const float* const* xy, int dim, const uint32_t* pairs, int64_t n_pairs,
                             float dist_ratio, int squared_metric,
                             uint32_t* counts, orc_match* out, int64_t out_cap);

/* ---- AC-RANSAC fundamental-matrix filter (OpenMVG GeometricFilter_FMatrix_AC + ACRANSAC +
 * SevenPointSolver + SymmetricEpipolarDistanceError; call site
 * src/R3DComputeMatches.cpp:2113-2115 with precision 4.0, 2048 iterations). */
typedef struct {
    double F[9];        /* row-major, un-normalised (pixel) fundamental matrix             */
    double threshold;   /* AC-RANSAC inlier threshold in pixels (unormalizeError(errorMax)) */
    double nfa;         /* minimum log10 NFA                                                */
    uint32_t n_inliers; /* inliers found by AC-RANSAC (before the > 2.5*7 acceptance test)   */
    uint32_t n_iter;    /* iterations actually executed                                      */
    uint32_t n_models;  /* models evaluated                                                  */
    int accepted;       /* 1 iff n_inliers > 2.5 * 7                                         */
} orc_fresult;

/* xI, xJ: m x 2 pixel coordinates (row-major doubles) of the putative matches, in putative
 * order.  inliers: capacity m; receives indices into the putative list in AC-RANSAC order
 * (ascending residual).  Returns the number of inliers (0 when rejected by AC-RANSAC itself;
 * check res->accepted for the 2.5*7 rule). */
int orc_acransac_F(const double* xI, const double* xJ, int m,
                   int wI, int hI, int wJ, int hJ,
                   double precision_px, uint32_t max_iter,
                   uint64_t seed, uint32_t I, uint32_t J,
                   uint32_t* inliers, orc_fresult* res);

/* Filter a whole putative graph (CSR as produced by orc_match_collection, pairs with count 0
 * skipped).  OpenMP over pairs (OpenMVG Robust_model_estimation).  out_counts[p] = number of
 * geometric matches kept for pair p (0 if rejected); matches concatenated in out. */
int64_t orc_filter_F_collection(int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t* out_counts, orc_match* out, double* F_out /* 9 per pair or NULL */);

/* Homography variant (GeometricFilter_HMatrix_AC, src/R3DComputeMatches.cpp:2216-2221): same result struct,
 * F[] holds the un-normalised H (x_J ~ H x_I), accepted iff n_inliers > 2.5 * 4. */
int orc_acransac_H(const double* xI, const double* xJ, int m, int wI, int hI, int wJ, int hJ,
                   double precision_px, uint32_t max_iter, uint64_t seed, uint32_t I, uint32_t J,
                   uint32_t* inliers, orc_fresult* res);
int64_t orc_filter_H_collection(int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t* out_counts, orc_match* out, double* H_out);
int      orc_four_point_h(const double* x1 /*4x2*/, const double* x2, double* H /*9*/);
double   orc_h_asym_err(const double* H, double x1, double y1, double x2, double y2);
void     orc_sample_n(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter,
                      const uint32_t* pool, uint32_t pool_size, uint32_t n, uint32_t* sample);

/* Debug trace of the next orc_acransac_F call(s): rows of (iter, model, #<=bound, NFA, improved). */
void orc_set_trace(double* buf, int cap_rows);
int  orc_trace_rows(void);

/* Pieces exposed for known-answer tests. */
uint64_t orc_rng_u64(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter, uint32_t attempt);
void     orc_sample7(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter,
                     const uint32_t* pool, uint32_t pool_size, uint32_t* sample7);
int      orc_seven_point(const double* x1 /*7x2 normalised*/, const double* x2, double* Fs /*3x9*/);
int      orc_solve_cubic(const double* coeffs /*c0..c3*/, double* roots);
double   orc_sym_epipolar_err(const double* F, double x1, double y1, double x2, double y2);
void     orc_logcombi_tables(uint32_t n, uint32_t k_sample, float* logc_n /*n+1*/, float* logc_k /*n+1*/);

/* ---- LIOP descriptor (vendored VLFeat copy src/thirdparty/liop/vl_liop.c; call site
 * src/Regard3DFeatures.cpp:827).  patches: n x side x side floats, desc: n x 144 floats. */
int orc_liop_describe(const float* patches, int n, int side, float* desc);
int orc_liop_geometry(int side, int* n_pix, int* pix, double* sx, double* sy);
/* patch extraction of extractLIOPFeatures (src/Regard3DFeatures.cpp:768-808): kps = n x (x, y, size, angle_deg) */
void orc_liop_affine(const float* kp, float kp_size_factor, float* M6);
int orc_liop_extract_patches(const float* image, int w, int h, const float* kps, int n, float kp_size_factor, float* patches);

/* ---- file formats (src/keypointSet.hpp:49-67, src/R3DProject.cpp:854-871) ---- */
int orc_save_matches(const char* path, int64_t n_pairs, const uint32_t* pairs,
                     const uint32_t* counts, const orc_match* matches);   /* .txt or .bin by extension */
/* Two-call load: first with pairs/counts/matches == NULL to get sizes. */
int orc_load_matches(const char* path, int64_t* n_pairs, int64_t* n_matches,
                     uint32_t* pairs, uint32_t* counts, orc_match* matches);
int orc_save_feat(const char* path, int n, const float* xyso /* n x 4: x y scale orientation */);
int orc_load_feat(const char* path, int* n, float* xyso, int cap);
int orc_save_desc(const char* path, uint64_t n, size_t row_bytes, const void* data);
int orc_load_desc(const char* path, uint64_t* n, size_t row_bytes, void* data, uint64_t cap);

/* ---- essential-matrix variant (GeometricFilter_EMatrix_AC, src/R3DComputeMatches.cpp:2169; oracle/essential.c):
 * 5-point solver on K^-1 x, EpipolarDistanceError in pixels through F = K2^-T E K1^-1, accepted iff n_inliers > 2.5 * 5.
 * res->F holds E, res->threshold the squared pixel bound (ACKernelAdaptorEssential::unormalizeError is the identity). */
int      orc_five_point(const double* x1 /*5x2*/, const double* x2, double* Es /*<= 10 x 9*/);
int      orc_real_roots10(const double* p, int deg, double* roots);
double   orc_epipolar_dist_err(const double* F, double x1, double y1, double x2, double y2);
void     orc_inv3(const double* K, double* Ki);
void     orc_f_from_e(const double* E, const double* K1i, const double* K2i, double* F);
int      orc_acransac_E(const double* xI, const double* xJ, int m, int wI, int hI, int wJ, int hJ,
                        const double* K1, const double* K2,
                        double precision_px, uint32_t max_iter, uint64_t seed, uint32_t I, uint32_t J,
                        uint32_t* inliers, orc_fresult* res);
int64_t  orc_filter_E_collection(int n_images, const int* n_rows, const float* const* xy,
                                 const uint32_t* widths, const uint32_t* heights, const double* Ks,
                                 const uint32_t* pairs, int64_t n_pairs,
                                 const uint32_t* counts, const orc_match* matches,
                                 double precision_px, uint32_t max_iter, uint64_t seed,
                                 uint32_t prune_min_count, float prune_min_ratio,
                                 uint32_t* out_counts, orc_match* out, double* E_out);

/* ---- Fast-A-KAZE keypoint detector (oracle/akaze.c; src/Regard3DFeatures.cpp:596-617, src/thirdparty/fast-akaze/).
 * PARITY UNPINNED: the reference detector sits on OpenCV primitives (absent here); see the header of akaze.c. */
int   orc_akaze_gauss_ksize(float sigma);
void  orc_akaze_gaussian(const float* src, int w, int h, float sigma, float* dst);
void  orc_akaze_scharr(const float* src, int w, int h, float* Lx, float* Ly);
void  orc_akaze_scaled_deriv(const float* src, int w, int h, int s, int dx, float* dst);
float orc_akaze_kcontrast(const float* Lx, const float* Ly, int w, int h, float perc, int nbins);
void  orc_akaze_halfsample(const float* src, int w, int h, float* dst);
int   orc_akaze_fed_tau(float T, float* tau);
void  orc_akaze_orientation_vec(const float* Lx, const float* Ly, int cols, int x0, int y0, int scale, float* out_xy);
void  orc_akaze_mldb(const float* Lt, const float* Lx, const float* Ly, int cols, float xf, float yf, float co, float si,
                     float scale, unsigned char* desc);
int   orc_akaze_detect_mldb(const float* image, int w, int h, float dthreshold, float* kps, unsigned char* desc, int cap, float* responses);
int   orc_akaze_detect(const float* image, int w, int h, float dthreshold, float* kps, int cap, float* responses, int* levels,
                       int dbg_level, float* dbg_ldet, float* dbg_lt, float* dbg_info);

/* ---- KGraph plugin path (config C5): oracle/kgraph.c.  src/thirdparty/kgraph/kgraph.cpp:411-552 (search),
 * :703-997 (NN-descent), :660-700 (reverse), src/utils/matcher_kgraph.h, src/R3DComputeMatches.cpp:808-902. */
/* ---- hnsw.c: the HNSW plugin path (hnswlib restated: levels, single-thread build, searchKnn; pinned by oracle/_ref) ---- */
typedef struct orc_hnsw orc_hnsw;
float     orc_hnsw_l2(const float* a, const float* b, uint32_t dim);
void      orc_hnsw_levels(uint32_t n, uint32_t M, uint32_t seed, int32_t* out);
orc_hnsw* orc_hnsw_build(const float* data, uint32_t n, uint32_t dim, uint32_t M, uint32_t ef_construction, uint32_t seed);
orc_hnsw* orc_hnsw_build_batch(const float* data, uint32_t n, uint32_t dim, uint32_t M, uint32_t seed);
orc_hnsw* orc_hnsw_from_arrays(const float* data, uint32_t n, uint32_t dim, uint32_t M, const int32_t* levels, const int32_t* links0,
                               const int32_t* up_off, const int32_t* up_links, int32_t enterpoint, int32_t maxlevel);
void      orc_hnsw_free(orc_hnsw* g);
uint32_t  orc_hnsw_up_rows(const orc_hnsw* g);
void      orc_hnsw_export(const orc_hnsw* g, int32_t* levels, int32_t* links0, int32_t* up_off, int32_t* up_links, int32_t* enterpoint, int32_t* maxlevel);
int64_t   orc_match_collection_hnsw(int n_images, const float* const* desc, const int* n_rows, const float* const* xy, int dim,
                                    const uint32_t* pairs, int64_t n_pairs, float dist_ratio, int builder, uint32_t M,
                                    uint32_t ef_construction, uint32_t ef, uint32_t seed, uint32_t min_rows,
                                    uint32_t* counts, orc_match* out, int64_t out_cap);
int       orc_hnsw_knn2(const orc_hnsw* g, const float* query, uint32_t nq, uint32_t ef, int32_t* idx, float* dist, uint64_t* n_dist);

typedef struct orc_kgraph orc_kgraph;
orc_kgraph* orc_kgraph_build_nndescent(const float* data, uint32_t n, uint32_t dim,
                                       uint32_t K, uint32_t L, uint32_t S, uint32_t R, uint32_t iterations,
                                       float recall_target, float delta_target, uint32_t n_controls, uint32_t seed,
                                       float* info);
orc_kgraph* orc_kgraph_build_exact(const float* data, uint32_t n, uint32_t dim, uint32_t K, uint32_t cap);
void     orc_kgraph_free(orc_kgraph* g);
uint32_t orc_kgraph_size(const orc_kgraph* g);
uint64_t orc_kgraph_edges(const orc_kgraph* g);
void     orc_kgraph_export(const orc_kgraph* g, uint64_t* off, uint32_t* ids, float* dist);
void     orc_kgraph_seeds(uint64_t seed, uint32_t I, uint32_t J, uint32_t q, uint32_t n, uint32_t P, uint32_t* out);
uint32_t orc_kgraph_search(const orc_kgraph* g, const float* data, uint32_t dim, const float* query,
                           uint32_t K, uint32_t P, uint32_t S, const uint32_t* seeds, uint32_t min_rows,
                           uint32_t* ids, float* dists, uint32_t* n_comps_out);
int      orc_kgraph_knn2(const orc_kgraph* g, const float* data, uint32_t dim, const float* query, uint32_t nq,
                         uint32_t P, uint32_t S, uint64_t seed, uint32_t I, uint32_t J, uint32_t min_rows,
                         int32_t* idx, float* dist, uint64_t* n_comps);
/* kgraph_match (src/R3DComputeMatches.cpp:808-902): f32 descriptors; builder 0 = exact index (cap closest edges),
 * 1 = NN-descent with (K, L, recall) of a preset.  Same CSR convention as orc_match_collection. */
int64_t  orc_match_collection_kgraph(int n_images, const float* const* desc, const int* n_rows,
                                     const float* const* xy, int dim, const uint32_t* pairs, int64_t n_pairs,
                                     float dist_ratio, int builder, uint32_t K, uint32_t L, float recall, uint32_t cap,
                                     uint32_t P, uint32_t S, uint64_t seed, uint32_t min_rows,
                                     uint32_t* counts, orc_match* out, int64_t out_cap, uint64_t* n_comps);

#ifdef __cplusplus
}
#endif
#endif
