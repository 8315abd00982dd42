// oracle/ref_hnsw_bruteforce.cpp -- thin C driver around the REFERENCE's own vendored
// hnswlib::BruteforceSearch<float> + L2Space (squared L2), compiled from the sources where
// they lie under /root/reference/src/thirdparty/hnswlib (see oracle/Makefile, target _ref).
// TEST INFRASTRUCTURE ONLY.  It is the independent second opinion that pins the oracle's
// brute-force 2-NN indices on integer-valued descriptors (sums are exact there, so the SIMD
// summation order and hnswlib's `dist <= lastdist` tie rule do not matter as long as the
// fixture has no exact ties) -- SURVEY.md section 8(c)(i).  Never shipped, never copied.
#include <iostream>
#include <fstream>
#include <queue>
#include <vector>
#include <cstring>
#include <cstdint>
#include <stdexcept>

#include "hnswlib/hnswlib.h"

extern "C" int ref_hnsw_knn_l2(const float* dataset, int nI, const float* query, int nJ, int dim,
                               int k, int32_t* idx, float* dist)
{
    try {
        hnswlib::L2Space space(dim);
        hnswlib::BruteforceSearch<float> bf(&space, (size_t)nI);
        for (int r = 0; r < nI; ++r)
            bf.addPoint((void*)(dataset + (size_t)r * dim), (hnswlib::labeltype)r);
        for (int q = 0; q < nJ; ++q) {
            auto res = bf.searchKnn(query + (size_t)q * dim, (size_t)k);
            // max-heap: pop gives descending distance
            int pos = (int)res.size();
            while (!res.empty()) {
                --pos;
                idx[(size_t)q * k + pos] = (int32_t)res.top().second;
                dist[(size_t)q * k + pos] = res.top().first;
                res.pop();
            }
        }
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}

// The reference's approximate matcher built on the same vendored library: hnswlib::HierarchicalNSW as
// ArrayMatcher_hnsw::Build / SearchNeighbours drive it (/root/reference/src/utils/matcher_hnsw.h:53-83,150-170:
// HierarchicalNSW(space, n, M, efConstruction), addPoint row by row, setEf(ef), searchKnn(query, NN)).  Gives the tests a
// REFERENCE-BUILT recall baseline for the graph matcher (presets: src/R3DComputeMatches.cpp:533-565).
extern "C" int ref_hnsw_ann_l2(const float* dataset, int nI, const float* query, int nJ, int dim,
                               int M, int efConstruction, int ef, int k, int32_t* idx, float* dist)
{
    try {
        hnswlib::L2Space space(dim);
        hnswlib::HierarchicalNSW<float> alg(&space, (size_t)nI, (size_t)M, (size_t)efConstruction);
        for (int r = 0; r < nI; ++r) alg.addPoint((void*)(dataset + (size_t)r * dim), (hnswlib::labeltype)r);
        alg.setEf((size_t)ef);
        for (int q = 0; q < nJ; ++q) {
            auto res = alg.searchKnn((void*)(query + (size_t)q * dim), (size_t)k);
            for (int p = 0; p < k; ++p) { idx[(size_t)q * k + p] = -1; dist[(size_t)q * k + p] = 0.f; }
            int pos = (int)res.size();
            while (!res.empty()) {
                --pos;
                if (pos < k) { idx[(size_t)q * k + pos] = (int32_t)res.top().second; dist[(size_t)q * k + pos] = res.top().first; }
                res.pop();
            }
        }
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}
