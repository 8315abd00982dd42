// r3d_compute_matches.hpp -- C++ facade with the call surface of the reference's stage driver,
// implemented purely on top of the C ABI (include/r3dm.h).
//
// Mirrors /root/reference/src/R3DComputeMatches.h:47-67:
//     bool R3DComputeMatches::computeMatches(Regard3DFeatures::R3DFParams &params, bool svgOutput,
//          const R3DProjectPaths &paths, int cameraModel, int matchingAlgorithm);
//     const R3DComputeMatchesStatistics &getStatistics();
// and the parameter structs it takes (/root/reference/src/Regard3DFeatures.h:52-69,
// /root/reference/src/R3DProject.h:39-65).  wxWidgets / OpenMVG types are replaced by std types:
// the SfM_Data the stage reads is reduced to the fields it actually uses -- view id, image size and
// the basename of the .feat/.desc files (/root/reference/src/R3DComputeMatches.cpp:1763-1777).
// computeMatches runs the whole stage as the reference does (/root/reference/src/R3DComputeMatches.cpp:1994-2240): first the
// features stage -- R3DFeaturesThread::extractFeaturesAndDescriptors, src/threads/R3DFeaturesThread.cpp:38-210 -- for every view
// whose <basename>.feat / <basename>.desc are not BOTH in the matches directory (views that have both are reused as they are,
// :139-142), then matching, the F / E / H filters and the match files.  Image DECODING (cv::imread) stays with the caller: a
// view carries its decoded pixels (View::bgr8 as cv::imread returns them, or View::gray = gray / 255 floats), or the caller sets an
// image provider that is asked for the pixels of exactly the views that need extraction.
#pragma once

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "r3dm.h"

namespace r3d_amd {

// Regard3DFeatures::R3DFParams (same member names, same defaults: src/Regard3DFeatures.cpp:128-135)
struct R3DFParams {
    std::vector<std::string> keypointDetectorList_{"Fast-AKAZE"};     // R3DFParams(): src/Regard3DFeatures.cpp:133
    float threshold_ = 0.001f;
    int nFeatures_ = 20000;
    float distRatio_ = 0.6f;
    bool computeHomographyMatrix_ = true;
    bool computeFundalmentalMatrix_ = true;
    bool computeEssentialMatrix_ = true;
};

// the subset of R3DProjectPaths the stage touches (same member names)
struct R3DProjectPaths {
    std::string relativeMatchesPath_;
    std::string matchesSfmDataFilename_;
    std::string matchesPutitativeFilename_;   // [sic] -- spelling of the reference
    std::string matchesFFilename_;
    std::string matchesEFilename_;
    std::string matchesHFilename_;
};

struct View {
    uint32_t id_view = 0;
    uint32_t ui_width = 0, ui_height = 0;
    std::string basename;                      // <matches dir>/<basename>.feat|.desc
    // pinhole intrinsics of the view (sfm_data.bin: Pinhole_Intrinsic focal / principal point); focal_px <= 0 = unknown,
    // such views are skipped by the essential-matrix filter exactly as E_ACRobust.hpp skips views without intrinsics
    double focal_px = -1.0, ppx = 0.0, ppy = 0.0;
    // decoded pixels for the features stage (host or device memory, borrowed until computeMatches returns; either may be null):
    // bgr8 = ui_height x ui_width x 3 bytes in cv::imread's BGR order, gray = ui_height x ui_width floats (gray / 255)
    const unsigned char* bgr8 = nullptr;
    const float* gray = nullptr;
};

using MatchList = std::vector<r3dm_match>;          // IndMatches (IndMatch{i_, j_} == r3dm_match{i, j})
using PairWiseMatches = std::map<std::pair<uint32_t, uint32_t>, MatchList>;

class R3DComputeMatches {
public:
    // matchingAlgorithm value of the new dispatch arm next to src/R3DComputeMatches.cpp:2054-2062;
    // 4 ("Brute Force", src/Regard3DMainFrameBase.cpp:1020) is served by the same GPU path; the approximate arms (0 FLANN,
    // 1..3 KGraph, 5 MRPT, 6..8 HNSW) run the graph-based approximate matcher with a preset of at least the arm's recall
    // (FLANN is permanently substituted by it, not reimplemented: r3dm.h, r3dm_ann_params_for_algorithm, DESIGN.md section 7;
    // the HNSW arms run hnswlib's own search on a batch-built index, r3dm_match_pairs_hnsw; the MRPT arm its own trees, r3dm_match_pairs_mrpt).
    static constexpr int kMatchingAlgorithmGPU = 9;

    explicit R3DComputeMatches(int device_id = 0);
    // all GPUs of a node from this one process: the pair loop and the filter loop of the stage are dealt to the listed devices
    // (r3dm_multi_*: one context + one host thread per device, results merged in the reference's map order)
    explicit R3DComputeMatches(const std::vector<int>& device_ids);
    ~R3DComputeMatches();
    R3DComputeMatches(const R3DComputeMatches&) = delete;
    R3DComputeMatches& operator=(const R3DComputeMatches&) = delete;

    void addViews(const std::vector<View>& views);                 // stands in for addImages + sfm_data.bin
    void clearViews() { views_.clear(); }                            // a kept-alive stage object between two collections
    void setRegionsType(r3dm_dtype dtype, uint32_t dim);           // default: float x 144 (R3D_AKAZE_LIOP_Regions)
    void setSeed(uint64_t seed) { seed_ = seed; }
    // The exhaustive matcher's exact MFMA fast paths: split-f16 nomination for real-valued descriptors (LIOP-144, what Regard3D matches:
    // 2.9x), bf16 tiles for integer-valued ones (SIFT bins: 7.9x) and i8 tiles for binary ones (A-KAZE MLDB: 3.0x, exact integers).  Both nominate candidates on narrow tiles and then compute the
    // distances of the candidates in f32 in the reference's order, with a certificate per query (a query whose certificate fails goes
    // through the exact scan), so indices, distances and match files are BIT-IDENTICAL to the plain f32-tile path (tests/
    // test_gpu_split_mfma.py, test_gpu_integer_mfma.py, test_liop_match_ref.py against reference-built code; bench.py --config stage
    // compares the files on every run).  ON by default in this facade since round 3; the C ABI's r3dm_match_pairs keeps f32 tiles
    // unless r3dm_set_split_mfma / r3dm_set_integer_mfma are called (BASELINE's configurations name the f32 arithmetic).
    void setExactFastPaths(bool on) { setIntegerFastPath(on); setSplitFastPath(on); setHammingFastPath(on); }
    // no reference counterpart: forwards r3dm_set_integer_mfma (bit-identical results, integer-valued descriptors only)
    void setIntegerFastPath(bool on);
    // no reference counterpart: forwards r3dm_set_split_mfma (bit-identical results, real-valued descriptors: LIOP)
    void setSplitFastPath(bool on);
    // no reference counterpart: forwards r3dm_set_hamming_mfma (bit-identical results, binary descriptors: A-KAZE MLDB on i8 MFMA tiles, 3.0x)
    void setHammingFastPath(bool on);
    // How the approximate arms of the dispatch (0 FLANN, 1-3 KGraph, 5 MRPT, 6-8 HNSW) are served.  kArmsFastest (default): by the
    // EXHAUSTIVE matcher whenever r3dm_exhaustive_is_faster says it is not slower on the registered views -- on LIOP-144 every
    // approximate arm is then exact and >= 2x faster than the graph search (the GUI's default arm 0 included); kArmsAsRequested:
    // arms 6-8 by hnsw_match itself (hnswlib's searchKnn on a batch-built HNSW index, r3dm_match_pairs_hnsw; descriptor lengths 64 /
    // 128 / 144 / 256), arm 5 by mrpt_match itself (random projection trees on the device, r3dm_match_pairs_mrpt), arms 1-3 by
    // kgraph_match, and the arms whose index is not built here (0 FLANN kd-trees, HNSW on other lengths) by the graph matcher with a
    // preset of at least the arm's recall (r3dm_ann_params_for_algorithm).
    enum ArmsPolicy { kArmsFastest = 0, kArmsAsRequested = 1 };
    void setApproximateArmsPolicy(ArmsPolicy p) { arms_policy_ = p; }
    // which matcher the last computeMatches call ran: true = exhaustive (arm 4 / 9, or an approximate arm routed to it)
    bool lastMatchWasExhaustive() const { return last_exhaustive_; }
    // ... true = hnsw_match (arms 6-8 under kArmsAsRequested)
    bool lastMatchWasHnsw() const { return last_hnsw_; }
    // ... true = mrpt_match (arm 5 under kArmsAsRequested)
    bool lastMatchWasMrpt() const { return last_mrpt_; }
    // updateProgress(float, const wxString&) (src/R3DComputeMatches.cpp:2664): the GUI hook, called with the reference's own
    // fractions and messages (0.7 "Find putative matches", 0.8 / 0.9 / 0.95 "Calculate ... matrix", :2000,2107,2133,2209)
    using ProgressFn = void (*)(float progress, const char* msg, void* user);
    void setProgressCallback(ProgressFn fn, void* user) { progress_ = fn; progress_user_ = user; }
    // Pixels on demand: called (from the thread that runs computeMatches) for each view that needs the features stage and
    // carries no pixels; fills bgr8 or gray of `out` (valid until the matching release call) and returns false if the image
    // cannot be provided.  Lets a host decode 16 images at a time instead of holding the whole collection in memory.
    struct Pixels { const unsigned char* bgr8 = nullptr; const float* gray = nullptr; };
    using ImageProviderFn = bool (*)(const View& view, Pixels* out, void* user);
    using ImageReleaseFn = void (*)(const View& view, void* user);
    void setImageProvider(ImageProviderFn get, ImageReleaseFn release, void* user) { provider_ = get; provider_release_ = release; provider_user_ = user; }
    // false: every view goes through its .feat / .desc files, as in the reference (default true: views computed in this call are
    // registered with the matcher from device memory -- the same values, without the file system and PCIe round trip)
    void setDirectRegistration(bool on) { direct_registration_ = on; }
    // detector batches in flight per device and images per batch of the features stage (defaults 3 x 8)
    void setFeaturesConcurrency(int batches_in_flight, int images_per_batch) { feat_conc_ = batches_in_flight; feat_batch_ = images_per_batch; }
    // nice value (1 .. 19) of the stage's background host threads -- match-file writers, PairWiseMatches map builders, the deferred
    // feature-file writers of its contexts -- so that they stand back behind the caller's threads when the process lives under a CPU
    // quota; 0 (default): the library leaves thread priorities alone
    void setBackgroundThreadsNice(int nice_value) { background_nice_ = nice_value < 0 ? 0 : (nice_value > 19 ? 19 : nice_value); }

    bool computeMatches(R3DFParams& params, bool svgOutput, const R3DProjectPaths& paths,
                        int cameraModel, int matchingAlgorithm);

    // wall time of the phases of the last computeMatches call, milliseconds (no reference counterpart: the GUI shows progress only)
    struct PhaseTimes {
        double features = 0, load = 0, match = 0, filter_F = 0, filter_E = 0, filter_H = 0, files = 0, total = 0;
        double match_kernels = 0, F_kernels = 0, E_kernels = 0, H_kernels = 0;     // HIP-event time of the dominant kernel of each phase
        double match_post = 0;                                                      // of `match`: everything behind the nomination kernel (exact scan of uncertified queries, finalisation, copy back)
        double filters_wall = 0;                                                    // F + E + H together: they run side by side on one device (r3dm_filter_FEH), so filter_F / _E / _H overlap
        uint64_t images_extracted = 0;
        r3dm_features_totals features_totals{};                                      // summed over the contexts of the features stage
    };
    const PhaseTimes& getPhaseTimes() const { return phases_; }

    struct R3DComputeMatchesStatistics {
        std::vector<int> numberOfKeypoints_;
        PairWiseMatches putativeMatches_;
        PairWiseMatches fundamentalMatches_;
        PairWiseMatches essentialMatches_;     // GeometricFilter_EMatrix_AC + overlap rule (r3dm_filter_E)
        PairWiseMatches homographyMatches_;    // GeometricFilter_HMatrix_AC (r3dm_filter_H)
    };
    const R3DComputeMatchesStatistics& getStatistics() const { return statistics_; }
    const std::string& errorMessage() const { return errorMessage_; }

private:
    r3dm_ctx* ctx_ = nullptr;
    r3dm_multi* multi_ = nullptr;          // set instead of ctx_ by the device-list constructor
    std::vector<View> views_;
    r3dm_dtype dtype_ = R3DM_F32;
    uint32_t dim_ = 144;
    ProgressFn progress_ = nullptr;
    void* progress_user_ = nullptr;
    uint64_t seed_ = 5489;
    std::vector<int> devices_;             // the device list this facade was built with
    r3dm_multi* feat_multi_ = nullptr;     // contexts of the features stage (feat_conc_ per device), created on first use
    int feat_conc_ = 3, feat_batch_ = 8;      // (round 5: with the detector and LIOP kernels at 59 ms per 24 images the host part of a batch is no longer hidden by two)
    ArmsPolicy arms_policy_ = kArmsFastest;
    bool last_exhaustive_ = true, last_hnsw_ = false, last_mrpt_ = false;
    // views registered with the matcher straight from the features stage (r3dm_set_features_sink): descriptors device to device,
    // positions as a reader of the .feat file would parse them; the load step reads the files of the other views only
    std::mutex sink_mu_;
    std::vector<char> registered_;         // per view of views_
    std::vector<uint32_t> registered_n_;
    const size_t* sink_need_ = nullptr;    // views of the running features chunk
    bool direct_registration_ = true;
    int background_nice_ = 0;
    static int features_sink(void* self, uint32_t image_index, uint32_t n_features, const float* desc_device, const float* xy_as_written);
    ImageProviderFn provider_ = nullptr;
    ImageReleaseFn provider_release_ = nullptr;
    void* provider_user_ = nullptr;
    PhaseTimes phases_;
    bool runFeaturesStage(const R3DFParams& params, const std::string& dir);
    R3DComputeMatchesStatistics statistics_;
    std::string errorMessage_;
};

}  // namespace r3d_amd

// C entry point over the facade (lets non-C++ hosts and the tests drive the directory-level stage)
extern "C" {
// the whole stage from pixels for non-C++ hosts, the tests and bench.py: views with decoded pixels (either pointer may be null
// when the view's .feat / .desc exist), the device list, the reference's parameters, and a report of the phases
typedef struct {
    uint32_t id, width, height;
    const char* basename;
    const unsigned char* bgr8;       /* height x width x 3, BGR, host or device; or NULL */
    const float* gray;               /* height x width floats, gray / 255, host or device; or NULL */
    double focal_px, ppx, ppy;       /* focal_px <= 0: intrinsics unknown */
} r3dm_view_image;
typedef struct {
    double ms_features, ms_load, ms_match, ms_filter_F, ms_filter_E, ms_filter_H, ms_files, ms_total;
    double ms_match_kernels, ms_F_kernels, ms_E_kernels, ms_H_kernels;
    double ms_filters_wall;          /* F + E + H together (side by side on one device: ms_filter_F / _E / _H overlap) */
    double ms_match_post;            /* of ms_match: behind the nomination kernel (exact scan of uncertified queries, finalisation, copy back) */
    uint64_t images_extracted, n_keypoints;
    uint64_t n_putative_pairs, n_putative_matches, n_F_pairs, n_F_matches, n_E_pairs, n_E_matches, n_H_pairs, n_H_matches;
    uint64_t match_was_exhaustive;   /* 1: the exhaustive matcher ran (arm 4 / 9, or an approximate arm routed to it) */
    r3dm_features_totals features;
} r3dm_stage_report;
int r3dm_compute_matches_stage(const int* device_ids, int n_devices, const char* matches_dir, const r3dm_view_image* views, uint32_t n_views,
                               float threshold, float dist_ratio, int matching_algorithm, int compute_F, int compute_E, int compute_H,
                               uint64_t seed, int features_batches_in_flight, int features_images_per_batch, uint32_t flags,
                               r3dm_stage_report* report, char* err, size_t err_cap);
/* The same with the stage object kept alive between calls, as a long-lived host (the Regard3D GUI process) would keep it: device
 * contexts, work buffers of the detector and page-locked staging memory are allocated once, not per call (~12 GB of hipMalloc
 * per features context at 12 Mpx and batches of 8). */
typedef struct r3dm_stage r3dm_stage;
int  r3dm_stage_create(const int* device_ids, int n_devices, r3dm_stage** out);
void r3dm_stage_destroy(r3dm_stage* s);
int  r3dm_stage_run(r3dm_stage* s, const char* matches_dir, const r3dm_view_image* views, uint32_t n_views,
                    float threshold, float dist_ratio, int matching_algorithm, int compute_F, int compute_E, int compute_H,
                    uint64_t seed, int features_batches_in_flight, int features_images_per_batch, uint32_t flags,
                    r3dm_stage_report* report, char* err, size_t err_cap);
/* flags of r3dm_compute_matches_stage / r3dm_stage_run */
#define R3DM_STAGE_ARMS_AS_REQUESTED 1u   /* approximate arms always on the graph matcher (default: the faster matcher, R3DComputeMatches::setApproximateArmsPolicy) */
#define R3DM_STAGE_SPLIT_MFMA        2u   /* (implied since round 3: the facade's exact fast paths are on by default) */
#define R3DM_STAGE_INTEGER_MFMA      4u   /* (implied since round 3) */
#define R3DM_STAGE_BACKGROUND_NICE  16u   /* background writer threads at nice 10 (R3DComputeMatches::setBackgroundThreadsNice(10)); default: priorities untouched */
#define R3DM_STAGE_F32_TILES         8u   /* plain f32 MFMA tiles for the exhaustive matcher: R3DComputeMatches::setExactFastPaths(false); same files, slower */
typedef struct { uint32_t id, width, height; const char* basename; } r3dm_view;
int r3dm_compute_matches_dir(int device_id, const char* matches_dir, const r3dm_view* views, uint32_t n_views,
                             r3dm_dtype dtype, uint32_t dim, float dist_ratio, int compute_F, uint64_t seed,
                             uint64_t* n_putative_pairs, uint64_t* n_geometric_pairs, char* err, size_t err_cap);
}
