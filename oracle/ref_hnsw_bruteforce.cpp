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

// The reference-built HierarchicalNSW exported AS DATA (arrays only): built single-threaded (addPoint in row order -- the reference
// adds row 0 first and the rest from an OpenMP loop, which makes its index depend on thread timing; the single-thread order is the
// one reproducible instance of it), then read out of hnswlib's own tables: element_levels_, the level-0 link lists
// (data_level0_memory_: count + maxM0 ids per element), the upper-level lists (linkLists_[i]: per level count + maxM ids),
// enterpoint_node_, maxlevel_.  With the 2-NN of `query` by the reference's own searchKnn(ef, k = 2) as ArrayMatcher_hnsw drives it
// (/root/reference/src/utils/matcher_hnsw.h:150-170): the fixture that pins the search restatement and the GPU search kernel.
//   levels   [n]                      int32
//   links0   [n][1 + 2 M]             int32 (count, ids)
//   up_off   [n + 1]                  int32 (prefix of the elements' level counts)
//   up_links [up_off[n]][1 + M]       int32 (count, ids), element i's level L list at row up_off[i] + L - 1
extern "C" int ref_hnsw_export(const float* dataset, int nI, const float* query, int nJ, int dim, int M, int efConstruction, int ef,
                               int32_t* levels, int32_t* links0, int32_t* up_off, int32_t* up_links, int up_cap,
                               int32_t* enterpoint, int32_t* maxlevel, int32_t* idx /* nJ x 2 */, float* dist /* nJ x 2 */)
{
    try {
        hnswlib::L2Space space(dim);
        hnswlib::HierarchicalNSW<float> alg(&space, (size_t)nI, (size_t)M, (size_t)efConstruction);
        for (int r = 0; r < nI; ++r) alg.addPoint((void*)(dataset + (size_t)r * dim), (hnswlib::labeltype)r);
        const int m0 = 2 * M;
        int rows = 0;
        for (int i = 0; i < nI; ++i) {
            levels[i] = alg.element_levels_[i];
            const unsigned int* l0 = alg.get_linklist0((hnswlib::tableint)i);
            links0[(size_t)i * (1 + m0)] = (int32_t)l0[0];
            for (int k = 0; k < m0; ++k) links0[(size_t)i * (1 + m0) + 1 + k] = k < (int)l0[0] ? (int32_t)l0[1 + k] : -1;
            up_off[i] = rows;
            for (int L = 1; L <= alg.element_levels_[i]; ++L) {
                if (rows >= up_cap) return -2;
                const unsigned int* ll = alg.get_linklist((hnswlib::tableint)i, L);
                up_links[(size_t)rows * (1 + M)] = (int32_t)ll[0];
                for (int k = 0; k < M; ++k) up_links[(size_t)rows * (1 + M) + 1 + k] = k < (int)ll[0] ? (int32_t)ll[1 + k] : -1;
                ++rows;
            }
        }
        up_off[nI] = rows;
        *enterpoint = (int32_t)alg.enterpoint_node_;
        *maxlevel = (int32_t)alg.maxlevel_;
        alg.setEf((size_t)ef);
        for (int q = 0; q < nJ; ++q) {
            auto res = alg.searchKnn((void*)(query + (size_t)q * dim), 2);
            for (int p = 0; p < 2; ++p) { idx[(size_t)q * 2 + p] = -1; dist[(size_t)q * 2 + p] = 0.f; }
            int pos = (int)res.size();
            while (!res.empty()) {               // max-heap on (distance, label): pop gives descending order, as ArrayMatcher_hnsw reverses it
                --pos;
                if (pos < 2) { idx[(size_t)q * 2 + pos] = (int32_t)res.top().second; dist[(size_t)q * 2 + pos] = res.top().first; }
                res.pop();
            }
        }
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}
