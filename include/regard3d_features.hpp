// regard3d_features.hpp -- C++ facade with the call surface of the reference's feature stage, header-only over the C ABI
// (include/r3dm.h).
//
// Mirrors /root/reference/src/Regard3DFeatures.h:36-110:
//     static std::vector<std::string> getKeypointDetectors();   static std::vector<std::string> getFeatureExtractors();
//     static void detectAndExtract(const Image<float>& img, FeatsR3D& feats, DescsR3D& descs, const R3DFParams& params);
//     (private) detectKeypoints(img, vec_keypoints, fdname, params), extractLIOPFeatures(img, vec_keypoints, kpSizeFactor, ...),
//     getKpSizeFactor(fdname)
// with FeatureR3D = openMVG::features::SIOPointFeature {x, y, scale, orientation}, DescriptorR3D = Descriptor<float, 144>
// (:41-45) replaced by plain structs of the same content, cv::KeyPoint by KeyPointR3D {x, y, size, angle}, and
// openMVG::image::Image<float> by a (pointer, width, height) view of the same row-major gray / 255 data.
// The only detector of the default list ("Fast-AKAZE", src/Regard3DFeatures.cpp:132) is served; the others of
// getKeypointDetectors() need OpenCV / VLFeat detectors that are outside the hot path.
#pragma once

#include <array>
#include <string>
#include <vector>

#include "r3d_compute_matches.hpp"      // R3DFParams
#include "r3dm.h"

namespace r3d_amd {

struct FeatureR3D { float x, y, scale, orientation; };          // SIOPointFeature
using FeatsR3D = std::vector<FeatureR3D>;
using DescriptorR3D = std::array<float, 144>;                   // LIOP
using DescsR3D = std::vector<DescriptorR3D>;
struct KeyPointR3D { float x, y, size, angle; };                // the cv::KeyPoint fields the stage uses
struct ImageViewF { const float* data; uint32_t width, height; };

class Regard3DFeatures {
public:
    using R3DFParams = r3d_amd::R3DFParams;

    static std::vector<std::string> getKeypointDetectors() { return {"Fast-AKAZE"}; }
    static std::vector<std::string> getFeatureExtractors() { return {"LIOP"}; }            // src/Regard3DFeatures.cpp:198-204

    // src/Regard3DFeatures.cpp:691-717 (only the detectors served here)
    static float getKpSizeFactor(const std::string& fdname) { return (fdname == "AKAZE" || fdname == "Fast-AKAZE") ? 8.0f : 1.0f; }

    // src/Regard3DFeatures.cpp:574-617; false (with r3dm_last_error set) instead of an OpenCV exception
    static bool detectKeypoints(r3dm_ctx* ctx, const ImageViewF& img, std::vector<KeyPointR3D>& vec_keypoints,
                                const std::string& fdname, const R3DFParams& params)
    {
        vec_keypoints.clear();
        if (fdname != "Fast-AKAZE") return false;
        uint32_t n = 0, cap = 65536;
        for (int attempt = 0; attempt < 2; ++attempt) {
            vec_keypoints.resize(cap);
            if (r3dm_detect_akaze(ctx, img.data, img.width, img.height, params.threshold_,
                                  reinterpret_cast<float*>(vec_keypoints.data()), nullptr, cap, &n) != R3DM_OK) { vec_keypoints.clear(); return false; }
            if (n <= cap) break;
            cap = n;
        }
        vec_keypoints.resize(n);
        return true;
    }

    // src/Regard3DFeatures.cpp:719-861: appends to feats / descs like the reference (scale = size / 2, :835-836)
    static bool extractLIOPFeatures(r3dm_ctx* ctx, const ImageViewF& img, const std::vector<KeyPointR3D>& vec_keypoints,
                                    float kpSizeFactor, FeatsR3D& feats, DescsR3D& descs)
    {
        if (vec_keypoints.empty()) return true;
        const size_t n = vec_keypoints.size(), base = descs.size();
        descs.resize(base + n);
        if (r3dm_extract_liop(ctx, img.data, img.width, img.height, reinterpret_cast<const float*>(vec_keypoints.data()), (uint32_t)n,
                              kpSizeFactor, reinterpret_cast<float*>(descs.data() + base), nullptr) != R3DM_OK) { descs.resize(base); return false; }
        for (const KeyPointR3D& kp : vec_keypoints) feats.push_back({kp.x, kp.y, kp.size / 2.0f, kp.angle});
        return true;
    }

    // src/Regard3DFeatures.cpp:206-222
    static bool detectAndExtract(r3dm_ctx* ctx, const ImageViewF& img, FeatsR3D& feats, DescsR3D& descs, const R3DFParams& params)
    {
        std::vector<std::string> detectors = params.keypointDetectorList_;
        if (detectors.empty()) detectors.push_back("Fast-AKAZE");                          // R3DFParams() default (:132)
        for (const std::string& keypointDetector : detectors) {
            std::vector<KeyPointR3D> vec_keypoints;
            if (!detectKeypoints(ctx, img, vec_keypoints, keypointDetector, params)) return false;
            if (!extractLIOPFeatures(ctx, img, vec_keypoints, getKpSizeFactor(keypointDetector), feats, descs)) return false;
        }
        return true;
    }
};

static_assert(sizeof(KeyPointR3D) == 16 && sizeof(DescriptorR3D) == 576, "r3dm_detect_akaze / r3dm_extract_liop exchange packed rows");

}  // namespace r3d_amd
