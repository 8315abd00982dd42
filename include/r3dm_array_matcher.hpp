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
// Only NN <= 2 is served (MatchDistanceRatio asks for exactly 2).
// Build() stages the dataset ONCE (r3dm_index_create: copy to HBM + MFMA fragment tiles + norms); every SearchNeighbours
// uploads only its query rows (r3dm_index_knn2) -- the amortisation the plugin contract is built around: the reference
// builds per first view I and searches once per J (/root/reference/src/R3DComputeMatches.cpp:462-479).
// Thread-safety: the reference calls SearchNeighbours from many OpenMP threads (`omp parallel for schedule(dynamic)` over J,
// :465).  A context drives one HIP stream and owns its scratch, so the adapter leases a context per call from a small
// process-wide pool (kPoolSize per device): concurrent searches run on different streams and overlap on the GPU; callers
// beyond the pool size wait for a free context.  For whole-collection throughput use r3dm_match_pairs.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <vector>

#include "r3dm.h"
#include "r3dm_context_pool.hpp"     // detail::ContextPool / ContextLease (shared with regard3d_features.hpp)

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

    explicit ArrayMatcher_r3dm(int device_id = 0) : pool_(detail::ContextPool::of(device_id)) {}
    virtual ~ArrayMatcher_r3dm() { if (index_) r3dm_index_destroy(index_); }
    // not part of the ArrayMatcher interface: forwards r3dm_set_integer_mfma (same results, integer-valued float rows only)
    void setIntegerFastPath(bool on) { integer_fast_path_ = on; }
    // not part of the ArrayMatcher interface: forwards r3dm_set_split_mfma (same results, real-valued rows)
    void setSplitFastPath(bool on) { split_fast_path_ = on; }
    // copies + re-layouts made on this adapter's device by all adapters of the process (test / diagnostics hook)
    uint64_t viewsStaged() const { return pool_.viewsStaged(); }
    int contextsInUse() const { return pool_.created(); }
    ArrayMatcher_r3dm(const ArrayMatcher_r3dm&) = delete;
    ArrayMatcher_r3dm& operator=(const ArrayMatcher_r3dm&) = delete;

    bool Build(const Scalar* dataset, int nbRows, int dimension) R3DM_OVERRIDE
    {
        if (nbRows < 1 || dimension < 1 || !dataset) return false;     // matcher_kgraph.h:126-130
        detail::ContextLease lease(pool_);
        if (!lease.ctx) return false;
        if (index_) { r3dm_index_destroy(index_); index_ = nullptr; }
        const r3dm_dtype dt = sizeof(Scalar) == 1 ? R3DM_U8 : R3DM_F32;
        nbRows_ = nbRows; dimension_ = dimension;
        return r3dm_index_create(lease.ctx, dataset, static_cast<uint32_t>(nbRows), static_cast<uint32_t>(dimension), dt, &index_) == R3DM_OK;
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
        if (!index_ || !query || nbQuery < 1 || NN < 1 || NN > 2 || nbRows_ < 2) return false;
        std::vector<int32_t> idx(2 * static_cast<size_t>(nbQuery));
        std::vector<float> dist(2 * static_cast<size_t>(nbQuery));
        {
            detail::ContextLease lease(pool_);                 // one stream + scratch per concurrent search
            if (!lease.ctx) return false;
            (void)r3dm_set_integer_mfma(lease.ctx, integer_fast_path_ ? 1 : 0);
            (void)r3dm_set_split_mfma(lease.ctx, split_fast_path_ ? 1 : 0);
            if (r3dm_index_knn2(lease.ctx, index_, query, static_cast<uint32_t>(nbQuery), idx.data(), dist.data()) != R3DM_OK)
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
    detail::ContextPool& pool_;
    r3dm_index* index_ = nullptr;
    int nbRows_ = 0, dimension_ = 0;
    bool integer_fast_path_ = false, split_fast_path_ = false;
};

}  // namespace r3d_amd
