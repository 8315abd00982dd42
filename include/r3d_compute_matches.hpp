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
// The facade matches views whose <basename>.feat / <basename>.desc exist in the matches directory, which is the
// reference's behaviour when both files exist (/root/reference/src/threads/R3DFeaturesThread.cpp:139-142); they are
// produced by the feature stage (include/regard3d_features.hpp, r3dm_extract_features_to_files).
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "r3dm.h"

namespace r3d_amd {

// Regard3DFeatures::R3DFParams (same member names, same defaults: src/Regard3DFeatures.cpp:128-135)
struct R3DFParams {
    std::vector<std::string> keypointDetectorList_;
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
};

using MatchList = std::vector<r3dm_match>;          // IndMatches (IndMatch{i_, j_} == r3dm_match{i, j})
using PairWiseMatches = std::map<std::pair<uint32_t, uint32_t>, MatchList>;

class R3DComputeMatches {
public:
    // matchingAlgorithm value of the new dispatch arm next to src/R3DComputeMatches.cpp:2054-2062;
    // 4 ("Brute Force", src/Regard3DMainFrameBase.cpp:1020) is served by the same GPU path; the approximate arms (0 FLANN,
    // 1..3 KGraph, 5 MRPT, 6..8 HNSW) run the graph-based approximate matcher with a preset of at least the arm's recall
    // (FLANN / MRPT / HNSW are substituted by it, not reimplemented: r3dm.h, r3dm_ann_params_for_algorithm).
    static constexpr int kMatchingAlgorithmGPU = 9;

    explicit R3DComputeMatches(int device_id = 0);
    // all GPUs of a node from this one process: the pair loop and the filter loop of the stage are dealt to the listed devices
    // (r3dm_multi_*: one context + one host thread per device, results merged in the reference's map order)
    explicit R3DComputeMatches(const std::vector<int>& device_ids);
    ~R3DComputeMatches();
    R3DComputeMatches(const R3DComputeMatches&) = delete;
    R3DComputeMatches& operator=(const R3DComputeMatches&) = delete;

    void addViews(const std::vector<View>& views);                 // stands in for addImages + sfm_data.bin
    void setRegionsType(r3dm_dtype dtype, uint32_t dim);           // default: float x 144 (R3D_AKAZE_LIOP_Regions)
    void setSeed(uint64_t seed) { seed_ = seed; }
    // no reference counterpart: forwards r3dm_set_integer_mfma (bit-identical results, integer-valued descriptors only)
    void setIntegerFastPath(bool on);
    // updateProgress(float, const wxString&) (src/R3DComputeMatches.cpp:2664): the GUI hook, called with the reference's own
    // fractions and messages (0.7 "Find putative matches", 0.8 / 0.9 / 0.95 "Calculate ... matrix", :2000,2107,2133,2209)
    using ProgressFn = void (*)(float progress, const char* msg, void* user);
    void setProgressCallback(ProgressFn fn, void* user) { progress_ = fn; progress_user_ = user; }

    bool computeMatches(R3DFParams& params, bool svgOutput, const R3DProjectPaths& paths,
                        int cameraModel, int matchingAlgorithm);

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
    R3DComputeMatchesStatistics statistics_;
    std::string errorMessage_;
};

}  // namespace r3d_amd

// C entry point over the facade (lets non-C++ hosts and the tests drive the directory-level stage)
extern "C" {
typedef struct { uint32_t id, width, height; const char* basename; } r3dm_view;
int r3dm_compute_matches_dir(int device_id, const char* matches_dir, const r3dm_view* views, uint32_t n_views,
                             r3dm_dtype dtype, uint32_t dim, float dist_ratio, int compute_F, uint64_t seed,
                             uint64_t* n_putative_pairs, uint64_t* n_geometric_pairs, char* err, size_t err_cap);
}
