// r3dm_internal.hpp -- structures shared by the host library (api_*.cpp, r3dm_ctx.hpp) and the HIP kernels
// (kernels_*.hip).  Not part of the public ABI (include/r3dm.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/r3dm.h"

// Developer knobs -- A/B kernel variants (two of them ablations whose results are meaningless), traces, invariant checks, the
// launch-order switches -- exist only in the developer build: `build.sh dev` compiles the same sources with -DR3DM_DEVTOOLS
// plus tools/devtools/dev_knobs.cpp (the one place that calls getenv) into regard3d_amd/libr3dm_dev.so for tools/ and the
// fallback-path tests.  The product library (libr3dm.so) resolves every knob to its default at compile time and never reads
// the environment: a stray variable cannot change what a product call computes.
#ifdef R3DM_DEVTOOLS
int r3dm_dev_knob(const char* name, int dflt);
const char* r3dm_dev_str(const char* name);
#else
constexpr int r3dm_dev_knob(const char*, int dflt) { return dflt; }
constexpr const char* r3dm_dev_str(const char*) { return nullptr; }
#endif

namespace r3dm {

constexpr uint32_t kNone = 0xFFFFFFFFu;       // "no match" / invalid row
constexpr uint32_t kTileRows = 32;            // rows per MFMA tile (32x32x2 f32)
constexpr uint32_t kSlackBytes = 32768;
// A dispatch carries its global size (workgroups x threads per workgroup) as a 32-bit number of work-items: a launch of more
// than 2^32 - 1 threads is not rejected by the runtime, it WRAPS (found on the first C5-sized graph search: 4560 pairs x 4096
// workgroups x 256 threads ran only the first 464 pairs).  Launchers refuse such grids; callers batch below the limit.
constexpr uint64_t kMaxBlocksOf256 = 0xFFFFFFFFull / 256ull;       // zero slack behind every tiled / norm array (prefetch runs past the end)

// One registered view, resident in HBM.  Layouts (DESIGN.md "Data layout in HBM"):
//   rows  : row-major f32 [n][dim]            -- exact re-scoring in the reference's summation order
//   tiled : [n_tiles][G][2][32][4] f32        -- MFMA fragment order: float4 (g, h, r) = row 32t+r,
//                                                dims 8g+4h .. 8g+4h+3; one 1 KiB line per wave load
//   norms : f32 [n_tiles*32], ||row||^2, +inf for padding rows
//   bin   : row-major u32 [n_pad][words]      -- binary descriptors, zero padded
struct ImgDev {
    const float*    rows;
    const float*    tiled;
    const float*    norms;
    const uint32_t* bin;
    const float*    xy;        // [n][2] pixel coordinates or nullptr
    const uint32_t* canon;     // position-class id per feature (smallest index with identical xy) or nullptr
    uint32_t n;
    uint32_t n_tiles;
    uint32_t dim;              // floats per row (F32/U8) or bytes per row (BIN)
    uint32_t G;                // dim padded to 8, / 8   (F32/U8)
    uint32_t words;            // u32 words per row (BIN)
    uint32_t width, height;
    uint32_t max_norm_bits;    // float bits of max ||row||^2 (filled by the staging kernel)
    uint32_t max_abs_bits;     // float bits of max |element|   (filled by the staging kernel)
    uint32_t not_integer;      // bit 0: some element is not an integer, bit 1: some element is negative (filled by the staging kernel)
    // graph index of the view (kernels_ann.hip), nullptr until r3dm_match_pairs_kgraph builds it
    const uint32_t* ann_adj;   // [n][kAnnDeg] neighbour rows ordered by (distance, id), kNone padded
    const uint32_t* ann_deg;   // [n] valid entries of each adjacency row
    // part of the index: the rows once more as bf16 ([n][dim], row-major) when every element is a bf16 (integers of magnitude
    // <= 256: SIFT bins), else nullptr.  The graph search is bound by its row gathers; this halves their bytes and converts back to
    // the SAME f32 values, so the search computes exactly what it computes on `rows`.  (Keep the three pointers adjacent: one copy sets them.)
    const uint16_t* ann_rows16;
    // ... or as bytes ([n][dim] u8) when every element is an integer in 0 .. 255 (what SIFT descriptors are on disk): one 128-byte
    // line per 128-dimensional row.  A view has at most one of the two copies.
    const uint8_t* ann_rows8;
    // bf16 fragment-order tiles for the integer fast path (r3dm_set_integer_mfma): [n_tiles][G/2][2][32][8] bf16,
    // lane half h of 16-dim block kb holds dims 16 kb + 8 h .. + 7 of row 32 t + r.  Exact iff the view is
    // integer-valued with |x| <= 256 (every such value is a bf16); the kernel checks that itself.
    const uint16_t* tiled16;
    // f16 hi / lo fragment-order tiles for the split nominator (r3dm_set_split_mfma): [n_tiles][G/2][hi | lo][2][32][8] f16 of the
    // values scaled by 2^split_k (max|x| 2^split_k in [2^13, 2^14); both filled by stage_split_kernel)
    const uint16_t* tiledh;
    int32_t split_k;
    // (in the 4 bytes of padding that followed split_k: the size of the entry and the offsets of its other fields stay what the
    // profiled machine code of the matching kernels was compiled with, profiles/pmc_traffic.json)
    uint32_t has_dup;          // != 0: some feature of the view shares its position with an earlier one (filled by the staging kernel)
    // count tiles (kernels_match.hip, l2_knn2_counts_kernel): rows that are small integers times a per-row scale (LIOP) as f16 integers
    // in fragment order [n_tiles][G/2][2][32][8], the row scales, and whether EVERY row of the view is of that form
    const uint16_t* tiledc;    // keypoint order: the QUERY side of the nominator
    const float* cscale;       // [n] row scales, keypoint order
    const uint16_t* tiledp;    // the same tiles with the rows in the order of their scales: the DATASET side
    const float* cquad;        // [n_tiles][64] the row line of an ordered tile: ||a||^2 of its 32 rows, then their negated scales
    const uint32_t* cperm;     // [n_tiles * 32] row of the ordered image -> keypoint index (kNone on padding)
    uint32_t counts_fail;      // != 0: some row is not (filled by the staging kernel)
    // binary views, opt-in MFMA Hamming (r3dm_set_hamming_mfma): one byte (0 / 1) per bit in i8 fragment order,
    // [n_tiles][words][2][32][16]; `norms` then holds popcount + 0x3F800000 as float bits
    const uint8_t* tiled8;
};
#ifndef R3DM_INF
#define R3DM_INF __builtin_huge_valf()
#endif
constexpr uint32_t kAnnDeg = 64;        // adjacency slots per row
constexpr uint32_t kAnnMaxK = 32;       // forward (exact nearest) neighbours per row
constexpr uint32_t kAnnMinRows = 128;   // views with fewer rows are matched exhaustively

struct AnnBuildJob {
    uint32_t slot;
    unsigned long long* fwd;      // [n][K] keys (distance bits << 32 | row), ascending, ~0 padded
    uint32_t* rev_cnt;            // [n]   reverse edges received
    uint32_t* rev_cur;            // [n]   fill cursor
    uint32_t* rev_off;            // [n+1] exclusive scan of rev_cnt
    unsigned long long* rev;      // [<= n*K] reverse keys, grouped by receiving row
    uint32_t* adj;                // -> ImgDev::ann_adj
    uint32_t* deg;                // -> ImgDev::ann_deg
    const uint8_t* rows8;         // the view's byte rows (ImgDev::ann_rows8 once the index is published) or nullptr
};
struct AnnBuildParams {
    const ImgDev* imgs;
    const AnnBuildJob* jobs;
    uint32_t K;
};
struct AnnSearchParams {
    const ImgDev* imgs;
    const uint2*  pairs;          // slot indices (I, J)
    const uint2*  pair_ids;       // view ids (I, J): keys of the start-row stream
    uint32_t      n_pairs;
    uint32_t      qb_per_pair;    // workgroups (4 queries each) per pair
    uint32_t      q_stride;
    uint32_t      flag_words;     // u32 words of the visited bitset of one query
    uint32_t      P, S, pool_cap; // start rows, neighbours expanded per step, K + P
    uint64_t      seed;
    float         ratio_R;
    uint32_t*     nn_idx;
    int32_t*      knn_idx;        // optional
    float*        knn_dist;       // optional
    unsigned long long* n_comps;  // distance evaluations (atomic)
};

// ---- HNSW plugin path (kernels_hnsw.hip): an index is three arrays in hnswlib's own shape
struct HnswView {
    const float*   rows;          // [n][dim] f32, row-major
    const uint8_t* rows8;         // the same rows as bytes when every element is an integer 0 .. 255 (ImgDev::ann_rows8), else nullptr
    const int32_t* l0;            // [n][1 + 2M]: count, links (farthest first, -1 padded)
    const int32_t* up_off;        // [n + 1]: first upper-layer row of a node (layer L of node i: row up_off[i] + L - 1)
    const int32_t* up;            // [rows][1 + M]
    uint32_t n, dim, M;
    int32_t enter, maxlevel;
};
struct HnswSearchJob {
    HnswView     ix;
    const float* query;           // [nq][dim]
    uint32_t     nq;
    uint32_t     out_base;        // first slot of the pair in nn_idx / knn_*
};
struct HnswSearchParams {
    const HnswSearchJob* jobs;
    uint32_t n_jobs, ef, cand_cap, flag_words;
    uint32_t rows8;               // != 0: every job's index view holds its byte rows; the search gathers those (a quarter of the bytes, the same float values)
    float    ratio_R;
    uint32_t* nn_idx;
    int32_t*  knn_idx;            // optional
    float*    knn_dist;           // optional
    unsigned long long* n_comps;  // distance evaluations (atomic)
    uint32_t* n_overflow;         // queries whose candidate heap did not fit cand_cap (their nn_idx slot holds 0xFFFFFFFE)
    uint32_t dense_steps;         // developer build: != 0 measures every link of a hop, visited or not, eight LINKS per step (the round-3 stepping; wavefront-per-query kernel only)
    uint32_t queries_per_wave;    // developer build: 2 / 4 / 8 run hnsw_search_group_kernel (measured, not adopted); else a wavefront per query
    uint32_t per_query;           // set by the launcher: LDS bytes of one query's state
};
// ---- MRPT plugin path (kernels_mrpt.hip): random projection trees of a view in the shape of /root/reference/src/thirdparty/mrpt/mrpt.h
struct MrptView {
    const float*   rows;          // [n][dim] f32, row-major
    const float*   RT;            // [dim][n_trees * depth]: the random matrix, transposed (zeros where the sparse matrix has no entry)
    const float*   splits;        // [n_trees][2^depth - 1], heap order
    const int32_t* leaves;        // [n_trees][n]: rows of a tree, leaf after leaf
    const int32_t* leaf_first;    // [2^depth + 1]
    uint32_t n, dim, n_trees, depth;
};
struct MrptQueryJob {
    MrptView     ix;
    const float* query;           // [nq][dim]
    uint32_t     nq;
    uint32_t     out_base;        // first slot of the pair in nn_idx / knn_*
};
struct MrptQueryParams {
    const MrptQueryJob* jobs;
    uint32_t n_jobs, votes, elected_cap;
    uint32_t max_n, pool_pad, per_wave, waves;   // set by the launcher
    float    ratio;               // UN-squared: the test runs on sqrt distances (RegionsMatcherT(regions, false))
    uint32_t* nn_idx;
    int32_t*  knn_idx;            // optional
    float*    knn_dist;           // optional: sqrtf of the squared L2 distances, -1 for a query without two elected rows
    unsigned long long* n_comps;  // distance evaluations (atomic)
};
hipError_t launch_mrpt_project(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, const float* R, uint32_t n_trees, uint32_t depth, float* proj);
hipError_t launch_mrpt_trees(hipStream_t st, const float* proj, uint32_t n, uint32_t n_trees, uint32_t depth, uint32_t cap, unsigned long long* keys,
                             int32_t* leaves, float* splits);
hipError_t launch_mrpt_query(hipStream_t st, const MrptQueryParams& P, uint32_t max_nq, uint32_t max_n, uint32_t max_pool);

struct HnswBuildJob {
    const float*    rows;
    const uint32_t* adj;          // ImgDev::ann_adj of the view (exact 32-NN graph + reverse edges)
    const uint32_t* adj_deg;
    const uint32_t* up_node;      // [up_rows] node of an upper-layer row
    const uint32_t* up_level;     // [up_rows] its layer (1 ..)
    const uint32_t* members;      // rows of layer 1, then layer 2, ... each ascending
    const uint32_t* mem_off;      // [maxlevel + 1] offsets into members
    int32_t* l0;
    int32_t* up;
    uint32_t n, dim, M, up_rows;
};
struct HnswBuildParams {
    const HnswBuildJob* jobs;
};

struct MatchParams {
    const ImgDev* imgs;
    const uint2*  pairs;          // slot indices (I, J)
    uint32_t      n_pairs;
    uint32_t      qb_per_pair;    // workgroups per pair
    uint32_t      xcd_map;        // != 0: all workgroups of a pair on one XCD (blockIdx & 7)
    uint32_t      q_stride;       // entries per pair in nn_idx / knn_* (>= max n_J)
    float         ratio_R;        // ratio^2 (squared metric) or ratio
    float         err_scale;      // certification slack factor: 8 * Dpad * 2^-24
    uint32_t*     nn_idx;         // [n_pairs][q_stride] matched I row, kNone, or kFallback
    int32_t*      knn_idx;        // optional [n_pairs][q_stride][2]
    float*        knn_dist;       // optional [n_pairs][q_stride][2]
    // queries whose nominated pair could not be certified are redone exactly; they are collected per
    // pair (fb_q[pair][0..kFbPerPair)) so one workgroup can stream image I once for all of them
    uint32_t*     fb_q;           // [n_pairs][kFbPerPair] query rows
    uint32_t*     fb_cnt;         // [n_pairs] number collected (may exceed kFbPerPair: overflow -> global rescan)
    uint32_t*     fb_total;       // [2]: total uncertified queries, number that overflowed their pair's list
    // the exact scan of a pair's uncertified queries split over fb_slices workgroups (row ranges of image I): per-slice results in
    // fb_part, the last workgroup of a pair to finish merges them (fb_done: tickets, zeroed by the host)
    float4*       fb_part;        // [n_pairs][kFbPerPair][fb_slices] (d0, bits of i0, d1, bits of i1)
    uint32_t*     fb_done;        // [n_pairs]
    uint32_t      fb_slices;      // 1: one workgroup per pair, results emitted directly
};
constexpr uint32_t kFbPerPair = 256;
constexpr uint32_t kFallback = 0xFFFFFFFEu;

struct FinalizeParams {
    const ImgDev* imgs;
    const uint2*  pairs;
    uint32_t      n_pairs;
    uint32_t      q_stride;
    const uint32_t* nn_idx;
    uint32_t      sort_cap;       // LDS sort capacity: power of two, >= q_stride unless the spill buffers are set
    unsigned long long* spill_keys;   // optional [n_pairs][spill_stride]: sort buffer of pairs keeping more than sort_cap matches
    unsigned char*      spill_drop;   // optional [n_pairs][spill_stride]
    uint32_t      spill_stride;   // power of two >= q_stride
    r3dm_match*   out;            // compacted matches
    uint64_t      out_cap;
    unsigned long long* total;    // running total (atomic)
    uint64_t*     pair_off;       // [n_pairs]
    uint32_t*     pair_cnt;       // [n_pairs]
};

constexpr int kCoopB = 32;                  // models per batch of the cooperative AC-RANSAC kernel, at most
constexpr uint32_t kCoopMaxG = 30;          // slices per pair, at most (5-bit slice code of a task, 31 = start-up)

struct FilterParams {
    const ImgDev* imgs;
    const uint2*  pairs;          // slot indices, one per work item
    const uint2*  pair_ids;       // view ids (I, J): keys of the sample stream
    const uint64_t* offsets;      // per work item: [2k] = begin, [2k+1] = end of its putative list inside `matches`
    const uint32_t* order;        // launch order: workgroup b runs item order[b] (longest putative lists first), nullptr = identity
    const uint64_t* soff;         // per work item: start of its slice in the work arrays (multiples of 32 elements, m + 1 <= slice)
    const r3dm_match* matches;
    uint32_t      n_items;
    uint32_t      m_cap;          // LDS sort capacity (power of two); items with more putatives use the spill buffers
    uint32_t      wide;           // != 0: the 512-thread variant of the kernel (long match lists: one workgroup per CU, half the trips per pass)
    unsigned long long* spill_keys;  // optional global sort buffers for those items
    uint32_t*     spill_idx;
    const uint64_t* spill_off;    // [n_items] element offset of the item's slice (next_pow2(m) elements)
    double        precision_px;
    uint32_t      max_iter;
    uint64_t      seed;
    int           err_kind;
    int           model_kind;     // 0 = fundamental matrix (7-point), 1 = homography (4-point), 2 = essential matrix (5-point)
    uint32_t*     dbg;            // optional [4]: first violated invariant (code, item, a, b); R3DM_FILTER_CHECK=1
    const double* kinv;           // [slots][9] inverse intrinsics K^-1 of every view (essential matrix only)
    const float*  log10_tab;      // log10f(k), k = 0..max_m  (host-computed: same libm as the reference build)
    const float*  logc_k;         // logcombi(sample size, n), n = 0..max_m (host-computed)
    // outputs
    uint32_t*     inl_count;      // [n_items] inliers kept (0 if rejected)
    uint32_t*     inl_idx;        // [slices] inlier positions into the pair's putative list, AC-RANSAC order (indexed by soff)
    double*       F_out;          // [n_items][9]
    double*       thr_nfa;        // [n_items][2]  threshold px, nfa
    uint32_t*     iters;          // [n_items][2]  iterations, models
    // scratch (global memory, sliced per item by its `begin` offset)
    double*       pts_scratch;    // [slices][4] normalised (x1, y1, x2, y2)
    uint32_t*     pool_scratch;   // [n_matches]    current sampling pool
    float*        scratch_logc;   // [n_matches + n_items + 1] logcombi(k, m) table of each item
    double*       la_tab;         // [n_items][1024] NFA slope of every residual-histogram bin of the item (the one-workgroup kernel's scout pass)
    uint32_t      scout;          // != 0: the one-workgroup kernel scouts its models a wavefront each ahead of the walk (kernels_filter.hip)
    // ---- long pairs: the cooperative kernel (kernels_filter_coop.hip).  Items order[0 .. n_short) run on the one-workgroup-per-pair
    // kernel, the n_coop items of coop_items on a pool of workers over a task queue.
    uint32_t      n_short;
    uint32_t      n_coop;
    uint32_t      coop_workers;   // workgroups of the cooperative launch
    const uint32_t* coop_items;   // [n_coop] work item of every cooperative pair
    const uint32_t* coop_G;       // [n_coop] slices (workgroups per batch) of the pair, 1 .. kCoopMaxG
    const uint32_t* coop_slice;   // [n_coop] matches per slice (< 65536: the slice histograms count in 16 bits)
    const uint32_t* coop_hoff;    // [n_coop] first histogram slot of the pair
    unsigned char* coop_pub;      // [n_coop] 64-byte records a slice task reads about its pair (kernels_filter_coop.hip: CoopPub)
    double*       coop_models;    // [n_coop][chunk x 9 x MAX_MODELS] models of the current chunk of minimal samples
    double*       coop_bm;        // [n_coop][kCoopB][9] matrices the residuals of the batch in flight are taken with
    uint32_t*     coop_hist;      // [slots][kCoopB][512] residual histograms of a slice (1024 u16 bins per model)
    uint32_t*     coop_cnt;       // [slots][kCoopB] matches within the bound
    double*       coop_tstar;     // [slices] T*(k) of every item (indexed by soff), k = 0 .. m
    double*       coop_la;        // [n_coop][1024] NFA slope of every histogram bin
    unsigned long long* coop_prof; // developer build: [n_coop][16] wall-clock ticks per phase (kernels_filter_coop.hip), or nullptr
    uint32_t*     coop_q;         // task queue: [head, tail, done, potential, active, n_pairs, cap - 1, next pair to start] + seq[cap] + data[cap]
    // debug trace (R3DM_TRACE_PAIR): rows of (iter, model, #<=bound, NFA, improved) for one item
    double*       trace;
    uint32_t      trace_item, trace_cap, trace_iter;
    uint32_t*     trace_rows;
};

// Workgroup barrier for waves that exchange data through LDS, with an EXPLICIT `s_waitcnt lgkmcnt(0)`.  hipcc (ROCm 7.2) was
// seen to emit a bare s_barrier -- dropping the wait that __syncthreads' release fence implies -- when the barrier opens a loop
// header and the pending ds_write sits in the latch block (kernels_filter.hip, chunk loop: other waves then read a stale loop
// counter).  The builtin cannot be optimised away and costs nothing when nothing is pending.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void r3dm_syncthreads()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);          // vmcnt = max, expcnt = max, lgkmcnt = 0
    __syncthreads();
}
#endif

// ---- Fast-A-KAZE detector (kernels_akaze.hip) ----
struct AkTaps { int n; float k[31]; };                  // Gaussian taps (host: getGaussianKernel restated), n <= 31
struct AkAreaTab { int si; float alpha; };              // one source cell of an INTER_AREA destination cell
struct AkLevelDev {
    int w, h, border;
    float ratio, psize;                                  // octave ratio; keypoint size before the final doubling (esigma * 1.5)
    const float* Ldet; const float* Lx; const float* Ly; const float* Lt;
    uint32_t* row_cnt; uint32_t* row_off;                // [h - 2 border] extrema per image row / exclusive scan
    unsigned long long* mask; uint32_t mask_words;       // [h - 2 border][mask_words] one bit per interior pixel: is a scale-space extremum (64-pixel ballots)
    uint32_t* counts;                                    // [0] candidates, [1] list entries after the in-level pruning
    float4* cand;                                        // raster-ordered candidates (x, y, response, -)
    float4* list;                                        // kept points (x, y, response, -), in insertion order
    float*  live;                                        // scratch for the live set of large levels (4 x capacity)
    unsigned char* dead_lower; unsigned char* dead_upper;
    float4* out0; float2* out1; uint32_t* out_valid;     // refined (x, y, size, response), dominant gradient vector, kept?
};
struct AkMldbItem { uint32_t level; float xf, yf, co, si, scale; };   // keypoint in level coordinates (level = image * n_levels + level), cos / sin of its angle, sigma_size
// per image of a batch, written by the device: candidates the count pass found, whether they exceeded the slot capacity (then the
// image's lists were not built and the host repeats the detection phase with larger slot arrays), keypoints compacted
struct AkBatchMeta { uint32_t need, overflow, n_kp, pad; };
// one surviving keypoint as it crosses to the host: refined position, size (diameter), response, dominant gradient vector, level
struct AkKpRec { float x, y, size, response, max_x, max_y; uint32_t level, pad; };
// All launchers: B = images of the batch (same size); image buffers hold B planes back to back; levels = [B][n_levels].
hipError_t ak_mldb(hipStream_t st, const AkLevelDev* levels, const AkMldbItem* items, uint32_t n, const unsigned char* pairs, unsigned char* out);
hipError_t ak_bgr_to_gray(hipStream_t st, const unsigned char* bgr, float* gray, size_t n);
hipError_t ak_gaussian(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, int B, const AkTaps& kf);
hipError_t ak_scharr(hipStream_t st, const float* src, float* rd, float* rs, float* Lx, float* Ly, int w, int h, int B);
hipError_t ak_scharr_g2(hipStream_t st, const float* src, float* dst, int w, int h, int B, const float* inv_k2);
hipError_t ak_kcontrast(hipStream_t st, const uint32_t* hmax_bits, const uint32_t* hist, int nbins, uint32_t total, int have_hist, float* inv_k2, int B);
hipError_t ak_scaled_deriv_xy(hipStream_t st, const float* src, float* dst_x, float* dst_y, int w, int h, int B, int s);
hipError_t ak_scaled_deriv_det(hipStream_t st, const float* lx, const float* ly, float* ldet, int w, int h, int B, int s);
hipError_t ak_modg_max(hipStream_t st, const float* src, int w, int h, int B, uint32_t* out_max);
hipError_t ak_modg_hist(hipStream_t st, const float* src, int w, int h, int B, const uint32_t* hmax_bits, int nbins, uint32_t* hist);
hipError_t ak_fed_step(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, float step_size);
hipError_t ak_fed_multi(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, const float* tau, int n_steps);
hipError_t ak_level_head(hipStream_t st, const float* src, float* Lx, float* Ly, float* Ldet, float* flow, int w, int h, int B, const AkTaps& kf, int s,
                         const float* inv_k2, int rows_per_band);
hipError_t ak_fed_march(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, const float* tau, int n_steps, int rows_per_band);
hipError_t ak_halfsample(hipStream_t st, const float* src, float* dst, int w, int h, int B, const AkAreaTab* xt, const int* xb,
                         const AkAreaTab* yt, const int* yb);
hipError_t ak_extrema(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, int max_rows, float thr, int pass);
struct AkTileTable { uint32_t begin[17]; };            // first tile of every level in the extremum count pass (0xFFFFFFFF beyond the last level)
hipError_t ak_extrema_mask(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, const AkTileTable& tt, uint32_t n_tiles, float thr);
hipError_t ak_scan_rows(hipStream_t st, const AkLevelDev* levels, int n_levels, int B);
hipError_t ak_layout(hipStream_t st, AkLevelDev* levels, int n_levels, int B, unsigned char* slots, uint32_t cap, AkBatchMeta* meta);
hipError_t ak_prune_levels(hipStream_t st, const AkLevelDev* levels, int n_levels, int B);
hipError_t ak_list_ranges(hipStream_t st, const AkLevelDev* levels, int n_levels, int B);
hipError_t ak_cross(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, int mode);
hipError_t ak_refine(hipStream_t st, const AkLevelDev* levels, int n_levels, int B);
hipError_t ak_compact(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, AkKpRec* recs, uint32_t cap, AkBatchMeta* meta);
constexpr uint32_t kAkSlotBytes = 80;                  // per candidate slot: cand 16 + list 16 + live 16 + out0 16 + out1 8 + valid 4 + dead 2 (+ 2 spare)

// ---- launchers implemented in the .hip files (host side) ----
// registration of one float view (kernels_match.hip: stage_view_kernel): raw descriptors -> fragment-order tiles + norms + statistics
// (+ the row-major f32 copy when `rows` is given), the positions copied / entered into the position-class table by role blocks of the
// same launch, the classes read back by a second small launch
struct StageViewArgs {
    const void* raw;               // [n][dim] f32 or u8, row-major: device memory, or page-locked host memory read over the link
    int raw_is_u8;
    uint32_t n, dim, G, n_tiles;
    float* tiled; float* norms;
    float* rows;                   // optional
    uint32_t* img_stats;           // &ImgDev::max_norm_bits: 3 consecutive words (zeroed by the table entry's upload)
    const float* xy_src; float* xy_dst;         // optional positions (xy_src == xy_dst: already in place)
    unsigned long long* canon_keys; uint32_t* canon_vals; uint32_t canon_bits;     // hash table of 2^bits slots, all bytes 0xFF
    uint32_t* canon_dst; uint32_t* has_dup;     // -> ImgDev::canon's array, &ImgDev::has_dup
};
hipError_t launch_stage_view(hipStream_t st, const StageViewArgs& A);
hipError_t launch_stage_positions(hipStream_t st, const StageViewArgs& A);      // the position part alone (binary views)
hipError_t launch_untile_rows(hipStream_t st, const float* tiled, uint32_t n, uint32_t dim, uint32_t G, uint32_t n_tiles, float* rows);
hipError_t launch_stage_bf16(hipStream_t st, const float* tiled, uint32_t G, uint32_t n_tiles, uint16_t* tiled16);
hipError_t launch_stage_bin(hipStream_t st, const uint8_t* raw, uint32_t n, uint32_t nbytes,
                            uint32_t* bin, uint32_t words, uint32_t n_pad);
// returns hipErrorInvalidValue when (G, dtype) has no tensor kernel; caller falls back to the exact scan
hipError_t launch_l2_knn2(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles, bool integer_mfma = false);
// (between the translation units of the matcher: the bf16 launcher, hipErrorNotSupported = no such kernel for this G; the LDS-shared developer variants)
hipError_t launch_l2_knn2_int(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles);
hipError_t launch_l2_int_lds_variant(hipStream_t st, const MatchParams& P, uint32_t max_nj_tiles, int iv);
hipError_t launch_hamming_mfma(hipStream_t st, const MatchParams& P, uint32_t words, uint32_t max_nj_tiles);
hipError_t launch_stage_bin8(hipStream_t st, const uint32_t* bin, uint32_t n, uint32_t words, uint32_t n_tiles, uint8_t* tiled8, float* norms);
hipError_t launch_l2_knn2_split(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles);
// ImgDev::cquad holds [n_tiles][64] row-line floats, slack, then [n_tiles + 1][16] quad summaries (the extra line: +inf / scale 1, the
// "tile before the first"): offset of the summaries in floats
__host__ __device__ inline size_t counts_summary_offset(uint32_t n_tiles) { return (size_t)n_tiles * 64u + 8192u; }
hipError_t launch_l2_knn2_counts(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles, int variant = 0);   // variant 1: the two-list kernel for every G (l2_knn2_counts_kernel: float keys -- dataset views beyond 65,536 rows, and R3DM_COUNTS_TWO_LISTS in the developer build)
hipError_t launch_stage_counts(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, uint32_t GB, uint32_t n_tiles,
                               uint16_t* tiledc, float* cscale, const float* norms, uint16_t* tiledp, float* crow, uint32_t* cperm,
                               uint32_t* fail_dev);
hipError_t launch_stage_split(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, uint32_t GB, uint32_t n_tiles,
                              uint16_t* tiledh, const uint32_t* img_stats_dev, int32_t* split_k_dev);
hipError_t launch_l2_exact_items(hipStream_t st, const MatchParams& P, uint32_t count, int scan_all);
// exact scan of the per-pair fallback lists (one workgroup per pair); false return -> no kernel for this G
hipError_t launch_l2_exact_batch(hipStream_t st, const MatchParams& P, uint32_t G);
hipError_t launch_hamming_knn2(hipStream_t st, const MatchParams& P, uint32_t words, uint32_t max_n);
hipError_t launch_finalize(hipStream_t st, const FinalizeParams& P);
hipError_t launch_filter_F(hipStream_t st, const FilterParams& P);
// one pool of workers for the long pairs of up to three filters: dev_params = FilterParams[3] in device memory indexed by model kind,
// q = the scheduling words, start = kind << 30 | pair in the order pairs are started
hipError_t launch_filter_coop(hipStream_t st, const FilterParams* dev_params, uint32_t* q, const uint32_t* start, uint32_t n_workers);
size_t     filter_coop_lds_bytes();
inline size_t filter_coop_model_doubles(int model_kind) { return model_kind == 2 ? 32u * 90u : 64u * 27u; }   // chunk x 9 x MAX_MODELS
size_t     filter_F_lds_bytes(uint32_t m_cap, int model_kind);
// rows8: every job carries byte rows -> the all-pairs scan runs on integer dot products (same keys)
hipError_t launch_ann_build(hipStream_t st, const AnnBuildParams& P, uint32_t n_jobs, uint32_t max_n, uint32_t dim, bool rows8);
// rows_mode: 0 = f32 rows, 1 = ImgDev::ann_rows16 (bf16), 2 = ImgDev::ann_rows8 (u8) -- every indexed view of the batch must hold that copy;
// 3 = u8 rows on both sides (the query views hold ann_rows8 too): distances as integer dot products
hipError_t launch_ann_search(hipStream_t st, const AnnSearchParams& P, uint32_t max_nJ, uint32_t max_nI, uint32_t dim, int rows_mode);
hipError_t launch_hnsw_search(hipStream_t st, const HnswSearchParams& P, uint32_t max_nq, uint32_t max_n, uint32_t dim);
hipError_t launch_hnsw_link(hipStream_t st, const HnswBuildParams& P, uint32_t n_jobs, uint32_t max_items, uint32_t dim);
hipError_t launch_ann_rows16(hipStream_t st, const float* rows, uint16_t* rows16, size_t n_elems);
hipError_t launch_ann_rows8(hipStream_t st, const float* rows, uint8_t* rows8, size_t n_elems);
// img_of (optional): image of the batch each keypoint belongs to -- its pixels start img_of[k] * w * h floats into `image`
hipError_t launch_liop_extract(hipStream_t st, const float* image, int w, int h, const float* M6, const float* kern,
                               uint32_t n, float* patches, const uint32_t* img_of = nullptr);
// one contiguous run of matches of the graph gather kernel (kernels_graph.hip): element offsets into src / idx / dst
struct GraphSeg { uint64_t src, idx, dst; uint32_t cnt, pad; };
hipError_t launch_graph_gather(hipStream_t st, const r3dm_match* src, const uint32_t* idx, const GraphSeg* segs, uint32_t n_segs, r3dm_match* dst);

// geometry tables of the 41 x 41 LIOP patch (api_features.cpp: liop_prepare), all in device memory
struct LiopTables {
    const int* pix;            // [n_pix] support pixels in scan order, as offsets into the zero-ringed 43 x 43 patch
    const double* samp_w;      // [n_pix][4][2] fractional parts (wx, wy) of the four sample positions
    const int* samp_off;       // [n_pix][4] ringed-patch offset of every sample's top-left tap
    uint32_t n_pix;
};
hipError_t launch_liop(hipStream_t st, const LiopTables& T, const float* patches, uint32_t n, float* desc, uint32_t* n_tie_patches, uint32_t* tie_list);
// keypoints -> descriptors without the patches ever reaching HBM (the warp + blur runs inside the descriptor's wavefront)
hipError_t launch_liop_fused(hipStream_t st, const LiopTables& T, const float* image, int w, int h, const float* M6, const float* kern,
                             const uint32_t* img_of, uint32_t n, float* desc, uint32_t* n_tie_patches, uint32_t* tie_list);

}  // namespace r3dm
