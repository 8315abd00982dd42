// regard3d_features.hpp -- C++ facade with the call surface of the reference's feature stage, header-only over the C ABI
// (include/r3dm.h).
//
// Mirrors /root/reference/src/Regard3DFeatures.h:36-110 signature for signature:
//     static bool initAKAZESemaphore(int count = 1);   static void uninitializeAKAZESemaphore();              (:71-72)
//     static std::vector<std::string> getKeypointDetectors();   static std::vector<std::string> getFeatureExtractors();
//     static void detectAndExtract(const Image<float>& img, FeatsR3D& feats, DescsR3D& descs, const R3DFParams& params);  (:87-88)
//     (private) detectKeypoints(img, vec_keypoints, fdname, params), extractLIOPFeatures(img, vec_keypoints, kpSizeFactor, ...),
//     getKpSizeFactor(fdname)
// No handle in any signature: like the reference, the functions are static and are called concurrently from CPUs+1 worker
// threads (/root/reference/src/threads/R3DFeaturesThread.cpp:58-77, processWorkItem :123-210); each call leases a library
// context (one HIP stream + scratch) from the process-wide pool of r3dm_context_pool.hpp for its duration.
//
// Types: FeatureR3D = openMVG::features::SIOPointFeature {x, y, scale, orientation}, DescriptorR3D = Descriptor<float, 144>
// (:41-45) are plain structs of the same content here, cv::KeyPoint is KeyPointR3D {x, y, size, angle}, and
// openMVG::image::Image<float> is ImageViewF: a borrowed view of the same row-major gray / 255 data with Image's accessors
// (Width(), Height(), data()).  With OpenMVG on the include path define R3DM_WITH_OPENMVG: detectAndExtract then also
// takes the real openMVG::image::Image<float> (an Eigen row-major matrix: the view is taken without a copy).
//
// Detector arms.  getKeypointDetectors() lists what the reference lists (src/Regard3DFeatures.cpp:182-196).  Only the
// default arm "Fast-AKAZE" (R3DFParams(), :133) is computed by this library; the others live in OpenCV / OpenMVG / VLFeat
// code outside the hot path and FAIL LOUDLY: detectAndExtract throws std::runtime_error
// naming the detector (the reference throws a cv::Exception out of the same call when a detector cannot be created).
// isDetectorServed(name) tells a caller beforehand.
//
// initAKAZESemaphore(count): the reference serialises the scale-space construction with a counting semaphore because
// cv::AKAZE2 is itself multi-threaded (src/Regard3DFeatures.cpp:71-125, called with 1 from
// src/R3DComputeMatches.cpp:1847 and released at :2253).  Same meaning here: at most `count` detector calls are in flight
// on the GPU at once, the rest wait; uninitializeAKAZESemaphore() lifts the limit (then the pool size bounds it).
#pragma once

#include <array>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "r3d_compute_matches.hpp"      // R3DFParams
#include "r3dm.h"
#include "r3dm_context_pool.hpp"

#ifdef R3DM_WITH_OPENMVG
#include "openMVG/image/image_container.hpp"
#endif

namespace r3d_amd {

struct FeatureR3D { float x, y, scale, orientation; };          // SIOPointFeature
using FeatsR3D = std::vector<FeatureR3D>;
using DescriptorR3D = std::array<float, 144>;                   // LIOP
using DescsR3D = std::vector<DescriptorR3D>;
struct KeyPointR3D { float x, y, size, angle; };                // the cv::KeyPoint fields the stage uses

// stand-in for openMVG::image::Image<float>: row-major, values gray / 255 (src/threads/R3DFeaturesThread.cpp:160-175)
struct ImageViewF {
    const float* data_; uint32_t width_, height_;
    ImageViewF(const float* d, uint32_t w, uint32_t h) : data_(d), width_(w), height_(h) {}
#ifdef R3DM_WITH_OPENMVG
    ImageViewF(const openMVG::image::Image<float>& img)          // NOLINT: implicit on purpose, Image<float> call sites compile unchanged
        : data_(img.data()), width_((uint32_t)img.Width()), height_((uint32_t)img.Height()) {}
#endif
    int Width() const { return (int)width_; }
    int Height() const { return (int)height_; }
    const float* data() const { return data_; }
};

class Regard3DFeatures {
public:
    using R3DFParams = r3d_amd::R3DFParams;
    using FeatureR3D = r3d_amd::FeatureR3D;
    using FeatsR3D = r3d_amd::FeatsR3D;
    using DescriptorR3D = r3d_amd::DescriptorR3D;
    using DescsR3D = r3d_amd::DescsR3D;

    // src/Regard3DFeatures.h:71-72, src/Regard3DFeatures.cpp:172-180
    static bool initAKAZESemaphore(int count = 1)
    {
        if (count < 1) return false;                             // wxSemaphore(count < 0) is not IsOk()
        Sema& s = sema();
        std::lock_guard<std::mutex> lock(s.mu);
        s.limit = count; s.enabled = true;
        s.cv.notify_all();
        return true;
    }
    static void uninitializeAKAZESemaphore()
    {
        Sema& s = sema();
        std::lock_guard<std::mutex> lock(s.mu);
        s.enabled = false;
        s.cv.notify_all();
    }

    // device the static calls run on (not in the reference: it has no devices); default 0
    static void setDevice(int device_id) { device() = device_id; }
    // contexts the static calls may hold at once (default ContextPool::kPoolSize)
    static void setMaxConcurrentCalls(int n) { detail::ContextPool::of(device()).setLimit(n); }

    static std::vector<std::string> getKeypointDetectors()        // src/Regard3DFeatures.cpp:182-196 (no VLFeat: no "DOG")
    {
        return {"AKAZE", "Fast-AKAZE", "MSER", "ORB", "BRISK", "GFTT"};
    }
    static bool isDetectorServed(const std::string& fdname) { return fdname == "Fast-AKAZE"; }
    static std::vector<std::string> getFeatureExtractors() { return {"LIOP"}; }            // src/Regard3DFeatures.cpp:198-204

    // src/Regard3DFeatures.cpp:206-222.  Appends to feats / descs (the reference never clears them either).
    static void detectAndExtract(const ImageViewF& img, FeatsR3D& feats, DescsR3D& descs, const R3DFParams& params)
    {
        for (const std::string& keypointDetector : params.keypointDetectorList_) {
            std::vector<KeyPointR3D> vec_keypoints;
            detectKeypoints(img, vec_keypoints, keypointDetector, params);
            const float kpSizeFactor = getKpSizeFactor(keypointDetector);
            extractLIOPFeatures(img, vec_keypoints, kpSizeFactor, feats, descs);
        }
    }

    // src/Regard3DFeatures.cpp:691-717 (the nlopt-determined factors, all of them)
    static float getKpSizeFactor(const std::string& fdname)
    {
        if (fdname == "AKAZE" || fdname == "Fast-AKAZE") return 8.0f;
        if (fdname == "DOG") return 0.25f;
        if (fdname == "MSER") return 0.08f;
        if (fdname == "ORB") return 0.025f;
        if (fdname == "BRISK") return 0.15f;
        if (fdname == "GFTT") return 0.13f;
        if (fdname == "HARRIS") return 0.25f;
        return 1.0f;                                             // "SimpleBlob", "TBMR", anything else
    }

    // src/Regard3DFeatures.cpp:574-684 (private in the reference; public here for the tests)
    static void detectKeypoints(const ImageViewF& img, std::vector<KeyPointR3D>& vec_keypoints,
                                const std::string& fdname, const R3DFParams& params)
    {
        vec_keypoints.clear();
        if (!isDetectorServed(fdname))
            throw std::runtime_error("Regard3DFeatures::detectKeypoints: detector \"" + fdname +
                                     "\" is not computed by the GPU library (only \"Fast-AKAZE\" is)");
        SemaLocker locker;                                       // AKAZESemaLocker (:580, :592)
        detail::ContextLease lease(detail::ContextPool::of(device()));
        if (!lease.ctx) throw std::runtime_error("Regard3DFeatures::detectKeypoints: no usable gfx950 device");
        uint32_t n = 0, cap = 65536;
        for (int attempt = 0; attempt < 2; ++attempt) {
            vec_keypoints.resize(cap);
            if (r3dm_detect_akaze(lease.ctx, img.data(), (uint32_t)img.Width(), (uint32_t)img.Height(), params.threshold_,
                                  reinterpret_cast<float*>(vec_keypoints.data()), nullptr, cap, &n) != R3DM_OK) {
                vec_keypoints.clear();
                throw std::runtime_error(std::string("Regard3DFeatures::detectKeypoints: ") + r3dm_last_error(lease.ctx));
            }
            if (n <= cap) break;
            cap = n;
        }
        vec_keypoints.resize(n);
    }

    // src/Regard3DFeatures.cpp:719-861: appends to feats / descs like the reference (scale = size / 2, :835-836)
    static void extractLIOPFeatures(const ImageViewF& img, const std::vector<KeyPointR3D>& vec_keypoints,
                                    float kpSizeFactor, FeatsR3D& feats, DescsR3D& descs)
    {
        if (vec_keypoints.empty()) return;
        detail::ContextLease lease(detail::ContextPool::of(device()));
        if (!lease.ctx) throw std::runtime_error("Regard3DFeatures::extractLIOPFeatures: no usable gfx950 device");
        const size_t n = vec_keypoints.size(), base = descs.size();
        descs.resize(base + n);
        if (r3dm_extract_liop(lease.ctx, img.data(), (uint32_t)img.Width(), (uint32_t)img.Height(),
                              reinterpret_cast<const float*>(vec_keypoints.data()), (uint32_t)n,
                              kpSizeFactor, reinterpret_cast<float*>(descs.data() + base), nullptr) != R3DM_OK) {
            descs.resize(base);
            throw std::runtime_error(std::string("Regard3DFeatures::extractLIOPFeatures: ") + r3dm_last_error(lease.ctx));
        }
        feats.reserve(feats.size() + n);
        for (const KeyPointR3D& kp : vec_keypoints) feats.push_back({kp.x, kp.y, kp.size / 2.0f, kp.angle});
    }

    // diagnostics for the tests: the largest number of detector calls that were inside the semaphore at once
    static int maxDetectorsInFlight() { Sema& s = sema(); std::lock_guard<std::mutex> lock(s.mu); return s.high_water; }
    static void resetMaxDetectorsInFlight() { Sema& s = sema(); std::lock_guard<std::mutex> lock(s.mu); s.high_water = s.in_flight; }

private:
    struct Sema { std::mutex mu; std::condition_variable cv; bool enabled = false; int limit = 1, in_flight = 0, high_water = 0; };
    static Sema& sema() { static Sema* s = new Sema; return *s; }         // never destroyed (worker threads may outlive main's statics)
    static int& device() { static int d = 0; return d; }
    struct SemaLocker {                                                   // AKAZESemaLocker, src/Regard3DFeatures.cpp:71-123
        SemaLocker()
        {
            Sema& s = sema();
            std::unique_lock<std::mutex> lock(s.mu);
            s.cv.wait(lock, [&] { return !s.enabled || s.in_flight < s.limit; });
            if (++s.in_flight > s.high_water) s.high_water = s.in_flight;
        }
        ~SemaLocker()
        {
            Sema& s = sema();
            { std::lock_guard<std::mutex> lock(s.mu); --s.in_flight; }
            s.cv.notify_all();
        }
        SemaLocker(const SemaLocker&) = delete;
        SemaLocker& operator=(const SemaLocker&) = delete;
    };
};

static_assert(sizeof(KeyPointR3D) == 16 && sizeof(DescriptorR3D) == 576, "r3dm_detect_akaze / r3dm_extract_liop exchange packed rows");

}  // namespace r3d_amd
