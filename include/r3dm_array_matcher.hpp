// r3dm_array_matcher.hpp -- the reference's matcher-plugin slot, served by the GPU library.
//
// Shape of openMVG::matching::ArrayMatcher<Scalar, Metric> as the reference's own plugins implement it
// (/root/reference/src/utils/matcher_kgraph.h:34-260, matcher_hnsw.h:34-206, matcher_mrpt.h:45-259):
//     bool Build(const Scalar* dataset, int nbRows, int dimension);
//     bool SearchNeighbour(const Scalar* query, int* indice, DistanceType* distance);
//     bool SearchNeighbours(const Scalar* query, int nbQuery, IndMatches* pvec_indices,
//                           std::vector<DistanceType>* pvec_distances, size_t NN);
// Contract kept: the dataset pointer is BORROWED and must outlive the matcher
// (matcher_kgraph.h:134-136); SearchNeighbours emits nbQuery*NN entries IndMatch(queryRow, datasetRow)
// in ascending distance order per query (matcher_kgraph.h:222-244); `false` = failure and the caller
// (RegionsMatcherT::MatchDistanceRatio) then emits no matches for the pair.  Distances are squared L2
// (pass b_squared_metric = true to RegionsMatcherT, like the kgraph/hnsw plugins,
// /root/reference/src/R3DComputeMatches.cpp:569,842).
//
// With OpenMVG on the include path define R3DM_WITH_OPENMVG before including this header: the class
// then derives from openMVG::matching::ArrayMatcher<Scalar, Metric> and uses its IndMatch type, so it
// drops into `RegionsMatcherT<ArrayMatcher_r3dm<float>>` exactly like ArrayMatcher_kgraph
// (/root/reference/src/R3DComputeMatches.cpp:838-842).  Without it (this repository: no OpenMVG in
// the image) equivalent stand-in types are used so the adapter can be compiled and tested.
//
// Only NN <= 2 is served (MatchDistanceRatio asks for exactly 2).  Thread-safety: the reference calls
// SearchNeighbours from many OpenMP threads; calls on one adapter are serialised by a mutex because
// one r3dm context drives one GPU stream.  For whole-collection throughput use r3dm_match_pairs
//  -- this adapter re-stages the query set on every call.
#pragma once

#include <cstdint>
#include <mutex>
#include <vector>

#include "r3dm.h"

#ifdef R3DM_WITH_OPENMVG
#include "openMVG/matching/indMatch.hpp"
#include "openMVG/matching/matching_interface.hpp"
#include "openMVG/matching/metric.hpp"
#endif

namespace r3d_amd {

#ifdef R3DM_WITH_OPENMVG
using IndMatch = openMVG::matching::IndMatch;
using IndMatches = openMVG::matching::IndMatches;
template <typename Scalar> using DefaultMetric = openMVG::matching::L2<Scalar>;
#define R3DM_ARRAY_MATCHER_BASE(Scalar, Metric) : public openMVG::matching::ArrayMatcher<Scalar, Metric>
#define R3DM_OVERRIDE override
#else
struct IndMatch {
    IndMatch(uint32_t i = 0, uint32_t j = 0) : i_(i), j_(j) {}
    uint32_t i_, j_;
};
using IndMatches = std::vector<IndMatch>;
template <typename Scalar> struct DefaultMetric { using ResultType = float; };
#define R3DM_ARRAY_MATCHER_BASE(Scalar, Metric)
#define R3DM_OVERRIDE
#endif

template <typename Scalar = float, typename Metric = DefaultMetric<Scalar>>
class ArrayMatcher_r3dm R3DM_ARRAY_MATCHER_BASE(Scalar, Metric) {
public:
    using DistanceType = typename Metric::ResultType;

    explicit ArrayMatcher_r3dm(int device_id = 0) { if (r3dm_create(device_id, &ctx_) != R3DM_OK) ctx_ = nullptr; }
    virtual ~ArrayMatcher_r3dm() { if (ctx_) r3dm_destroy(ctx_); }
    // not part of the ArrayMatcher interface: forwards r3dm_set_integer_mfma (same results, integer-valued float rows only)
    void setIntegerFastPath(bool on) { if (ctx_) (void)r3dm_set_integer_mfma(ctx_, on ? 1 : 0); }
    ArrayMatcher_r3dm(const ArrayMatcher_r3dm&) = delete;
    ArrayMatcher_r3dm& operator=(const ArrayMatcher_r3dm&) = delete;

    bool Build(const Scalar* dataset, int nbRows, int dimension) R3DM_OVERRIDE
    {
        if (nbRows < 1 || !ctx_) return false;              // matcher_kgraph.h:126-130
        dataset_ = dataset; nbRows_ = nbRows; dimension_ = dimension;
        return true;
    }

    bool SearchNeighbour(const Scalar* query, int* indice, DistanceType* distance) R3DM_OVERRIDE
    {
        IndMatches idx; std::vector<DistanceType> dist;
        if (!SearchNeighbours(query, 1, &idx, &dist, 1)) return false;
        indice[0] = static_cast<int>(idx[0].j_); distance[0] = dist[0];
        return true;
    }

    bool SearchNeighbours(const Scalar* query, int nbQuery, IndMatches* pvec_indices,
                          std::vector<DistanceType>* pvec_distances, size_t NN) R3DM_OVERRIDE
    {
        if (!ctx_ || !dataset_ || nbQuery < 1 || NN < 1 || NN > 2 || nbRows_ < 2) return false;
        std::vector<int32_t> idx(2 * static_cast<size_t>(nbQuery));
        std::vector<float> dist(2 * static_cast<size_t>(nbQuery));
        {
            std::lock_guard<std::mutex> lock(mu_);
            const r3dm_dtype dt = sizeof(Scalar) == 1 ? R3DM_U8 : R3DM_F32;
            if (r3dm_knn2(ctx_, dataset_, static_cast<uint32_t>(nbRows_), query, static_cast<uint32_t>(nbQuery),
                          static_cast<uint32_t>(dimension_), dt, idx.data(), dist.data()) != R3DM_OK)
                return false;
        }
        pvec_indices->reserve(pvec_indices->size() + nbQuery * NN);
        pvec_distances->reserve(pvec_distances->size() + nbQuery * NN);
        for (int q = 0; q < nbQuery; ++q)
            for (size_t k = 0; k < NN; ++k) {
                pvec_indices->emplace_back(static_cast<uint32_t>(q), static_cast<uint32_t>(idx[2 * q + k]));
                pvec_distances->emplace_back(static_cast<DistanceType>(dist[2 * q + k]));
            }
        return true;
    }

private:
    r3dm_ctx* ctx_ = nullptr;
    const Scalar* dataset_ = nullptr;
    int nbRows_ = 0, dimension_ = 0;
    std::mutex mu_;
};

}  // namespace r3d_amd
