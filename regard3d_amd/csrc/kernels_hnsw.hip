// kernels_hnsw.hip -- the HNSW plugin path (matchingAlgorithm 6..8 of /root/reference/src/R3DComputeMatches.cpp:2035-2062;
// hnsw_match :497-593, ArrayMatcher_hnsw src/utils/matcher_hnsw.h:53-83,150-190, hnswlib src/thirdparty/hnswlib/hnswlib/hnswalg.h).
//
//   hnsw_search_kernel   hnswlib's searchKnn(query, 2) with setEf(ef), one wavefront per query: greedy descent through the upper
//                        layers (hnswalg.h:745-768), then searchBaseLayerST on layer 0 (:214-280).  The two std::priority_queues
//                        are binary heaps in LDS moved exactly as libstdc++'s push_heap / pop_heap move them (hnswlib compares heap
//                        entries by distance only, so WHICH of two equally distant rows leaves first is decided by the sift order;
//                        integer-valued SIFT bins produce such ties all the time).  Distances are L2SqrSIMD16Ext's AVX arm
//                        (space_l2.h:40-75): eight interleaved partial sums, added left to right, no FMA -- lane l of an 8-lane
//                        group IS accumulator l, eight rows per wavefront step.  On an index exported from the reference-built
//                        library the results equal hnswlib's bit for bit (tests/test_gpu_hnsw.py).
//   hnsw_link_kernel     the batch construction (DESIGN.md "HNSW", CPU model oracle/hnsw.c: orc_hnsw_build_batch): one wavefront
//                        per (row, layer): candidates (layer 0: the row's list in the exact 32-NN graph of kernels_ann.hip; above:
//                        its 32 nearest members of the layer by an exact scan), hnswlib's getNeighborsByHeuristic2 (:282-322) over
//                        them in ascending (distance, row) order, refill with the closest rejected, list written farthest first.
//
// The distance evaluations are the same instruction sequence everywhere (hn_dist8), compiled with -ffp-contract=off.
#include "r3dm_internal.hpp"

namespace r3dm {
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
struct HnPair { float d; uint32_t id; };

// lanes of one wavefront meet here: compiler fence + scheduling barrier (LDS operations of one wavefront execute in program order)
#define HN_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// bits/stl_heap.h __push_heap with CompareByFirst (a max-heap on d)
__device__ __forceinline__ void hn_sift_up(HnPair* v, uint32_t hole, float d, uint32_t id)
{
    while (hole > 0) {
        const uint32_t parent = (hole - 1) >> 1;
        const HnPair p = v[parent];
        if (!(p.d < d)) break;
        v[hole] = p; hole = parent;
    }
    v[hole] = HnPair{d, id};
}
__device__ __forceinline__ void hn_push(HnPair* v, uint32_t& n, float d, uint32_t id) { hn_sift_up(v, n, d, id); n += 1; }
// pop_heap + pop_back: __pop_heap -> __adjust_heap
__device__ __forceinline__ void hn_pop(HnPair* v, uint32_t& n)
{
    if (n > 1) {
        const uint32_t len = n - 1;
        const HnPair value = v[len];
        uint32_t hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            const HnPair a = v[second], b = v[second - 1];
            if (a.d < b.d) { second--; v[hole] = b; } else v[hole] = a;
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            v[hole] = v[second - 1]; hole = second - 1;
        }
        hn_sift_up(v, hole, value.d, value.id);
    }
    n -= 1;
}

// accumulator l (= lane & 7) of L2SqrSIMD16Ext over one row: acc_l = sum_i (a[8i + l] - b[8i + l])^2 in increasing i
// ROW = float or uint8_t (byte rows of integer-valued views: the conversion is exact, so the sum is the same float)
template <int NB, typename ROW>
__device__ __forceinline__ float hn_acc(const float (&q)[NB], const ROW* __restrict__ row, uint32_t l)
{
    float x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = (float)row[8 * i + l];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) { const float t = q[i] - x[i]; acc = acc + t * t; }
    return acc;
}
// ... and the horizontal sum acc0 + acc1 + ... + acc7, left to right; valid in every lane of the 8-lane group
__device__ __forceinline__ float hn_hsum8(float acc, uint32_t lane)
{
    const int base = (int)(lane & ~7u);
    float r = __shfl(acc, base);
#pragma unroll
    for (int k = 1; k < 8; ++k) r = r + __shfl(acc, base + k);
    return r;
}
template <int NB>
__device__ __forceinline__ void hn_load_q(float (&q)[NB], const float* __restrict__ row, uint32_t l)
{
#pragma unroll
    for (int i = 0; i < NB; ++i) q[i] = row[8 * i + l];
}

// distances from the row in q to the `size` rows ids[0 .. size): dist[j] (LDS) for every j; skip[j] != 0 leaves dist[j] unset
template <int NB, typename ROW, typename SKIP>
__device__ __forceinline__ void hn_dist_list(const float (&q)[NB], const ROW* __restrict__ rows, uint32_t dim, const int32_t* ids, uint32_t size,
                                             float* dist, uint32_t lane, SKIP skip)
{
    const uint32_t g = lane >> 3, l = lane & 7u;
    for (uint32_t j0 = 0; j0 < size; j0 += 8) {
        const uint32_t j = j0 + g;
        const bool on = j < size;
        const uint32_t c = on ? (uint32_t)ids[j] : 0u;
        const bool work = on && !skip(c);
        float acc = 0.f;
        if (work) acc = hn_acc<NB, ROW>(q, rows + (size_t)c * dim, l);
        const float d = hn_hsum8(acc, lane);
        if (work && l == 0) dist[j] = d;
    }
}

// The base layer's variant: only the links that are not marked visited get a distance, and they are packed first -- eight unvisited
// rows per wavefront step instead of eight LINKS of which most are already visited (a hop of the precise preset has 38 links and
// ~6 new rows: one or two steps instead of five; the kernel is bound by the instructions it issues, and the distance steps were
// four fifths of them).  dist[j] still belongs to link j, so the sequential pass below reads what it read before.
// `dense_steps` (developer build, R3DM_HNSW_DENSE_STEPS): the round-3 stepping over all links, for A/B runs.
template <int NB, typename ROW>
__device__ __forceinline__ uint32_t hn_dist_list_unvisited(const float (&q)[NB], const ROW* __restrict__ rows, uint32_t dim, const int32_t* ids, uint32_t size,
                                                       float* dist, int32_t* todo, const uint32_t* visited, uint32_t lane)
{
    const uint32_t g = lane >> 3, l = lane & 7u;
    const bool on = lane < size;                                          // size <= 2M <= 64: a lane per link
    const uint32_t mine = on ? (uint32_t)ids[lane] : 0u;
    const bool need = on && ((visited[mine >> 5] >> (mine & 31u)) & 1u) == 0u;
    const unsigned long long m = __ballot(need);
    const uint32_t n_need = (uint32_t)__popcll(m);
    if (need) todo[__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)lane;
    HN_SYNC();
    for (uint32_t k0 = 0; k0 < n_need; k0 += 8) {
        const uint32_t k = k0 + g;
        const bool work = k < n_need;
        const uint32_t j = work ? (uint32_t)todo[k] : 0u;
        float acc = 0.f;
        if (work) acc = hn_acc<NB, ROW>(q, rows + (size_t)(uint32_t)ids[j] * dim, l);
        const float d = hn_hsum8(acc, lane);
        if (work && l == 0) dist[j] = d;
    }
    return n_need;
}

// ------------------------------------------------------------------------------------------------------------ search
template <int NB, typename ROW>
__global__ void __launch_bounds__(256) hnsw_search_kernel(const HnswSearchParams P)
{
    extern __shared__ unsigned char hn_smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const HnswSearchJob job = P.jobs[blockIdx.y];
    const uint32_t qi = blockIdx.x * 4 + wave;
    // per-wave LDS: visited bits | top heap (ef + 1) | candidate heap (cand_cap) | dist[64] | ids[64] | todo[64]
    const uint32_t per_wave = P.flag_words * 4 + (P.ef + 1) * 8 + P.cand_cap * 8 + 64 * 4 + 64 * 4 + 64 * 4;
    unsigned char* base = hn_smem + (size_t)wave * per_wave;
    uint32_t* visited = reinterpret_cast<uint32_t*>(base);
    HnPair* top = reinterpret_cast<HnPair*>(base + P.flag_words * 4);
    HnPair* cand = top + (P.ef + 1);
    float* dist = reinterpret_cast<float*>(cand + P.cand_cap);
    int32_t* ids = reinterpret_cast<int32_t*>(dist + 64);
    int32_t* todo = ids + 64;
    if (qi >= job.nq) return;                                            // (no workgroup barrier below)
    const HnswView ix = job.ix;
    const uint32_t dim = ix.dim, M = ix.M, l = lane & 7u;
    const ROW* __restrict__ xrows = sizeof(ROW) == 1 ? reinterpret_cast<const ROW*>(ix.rows8) : reinterpret_cast<const ROW*>(ix.rows);
    const size_t o = (size_t)job.out_base + qi;

    float q[NB];
    hn_load_q<NB>(q, job.query + (size_t)qi * dim, l);
    for (uint32_t w = lane; w < P.flag_words; w += 64) visited[w] = 0;
    unsigned long long evals = 1;

    uint32_t cur = (uint32_t)ix.enter;
    float curdist = hn_hsum8(hn_acc<NB, ROW>(q, xrows + (size_t)cur * dim, l), lane);
    curdist = __shfl(curdist, 0);
    // greedy descent (hnswalg.h:745-768): the list is walked in order, every strictly closer row takes over
    for (int level = ix.maxlevel; level > 0; --level) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int32_t* L = ix.up + ((size_t)ix.up_off[cur] + (uint32_t)(level - 1)) * (1 + M);
            const uint32_t size = (uint32_t)L[0];
            hn_dist_list<NB, ROW>(q, xrows, dim, L + 1, size, dist, lane, [](uint32_t) { return false; });
            HN_SYNC();
            evals += size;
            for (uint32_t j = 0; j < size; ++j) {
                const float d = dist[j];
                if (d < curdist) { curdist = d; cur = (uint32_t)L[1 + j]; changed = true; }
            }
            HN_SYNC();
        }
    }
    // searchBaseLayerST (hnswalg.h:214-280)
    const uint32_t ef = P.ef;
    uint32_t n_top = 0, n_cand = 0;
    bool overflow = false;
    HN_SYNC();
    if (lane == 0) visited[cur >> 5] |= 1u << (cur & 31u);
    hn_push(top, n_top, curdist, cur);
    hn_push(cand, n_cand, -curdist, cur);
    float lower = curdist;
    HN_SYNC();
    while (n_cand) {
        const HnPair c0 = cand[0];
        if ((-c0.d) > lower) break;
        hn_pop(cand, n_cand);
        const int32_t* L = ix.l0 + (size_t)c0.id * (1 + 2 * M);
        const uint32_t size = (uint32_t)L[0];
        HN_SYNC();
        if (lane < size) ids[lane] = L[1 + lane];                       // 2M <= 64 links
        HN_SYNC();
        uint32_t n_walk = size;
        if (P.dense_steps) hn_dist_list<NB, ROW>(q, xrows, dim, ids, size, dist, lane, [&](uint32_t c) { return ((visited[c >> 5] >> (c & 31u)) & 1u) != 0; });
        else n_walk = hn_dist_list_unvisited<NB, ROW>(q, xrows, dim, ids, size, dist, todo, visited, lane);
        HN_SYNC();
        // the walk over the list in link order; links that were visited before the hop are no-ops in hnswlib's loop, so only the others
        // (todo, ascending) are walked -- a row that occurs twice in a list is in todo twice and the second visit finds it marked
        for (uint32_t k = 0; k < n_walk; ++k) {
            const uint32_t j = P.dense_steps ? k : (uint32_t)todo[k];
            const uint32_t c = (uint32_t)ids[j];
            const uint32_t w = visited[c >> 5], bit = 1u << (c & 31u);
            if (w & bit) continue;                                     // seen before this list, or earlier in this list
            visited[c >> 5] = w | bit;
            evals += 1;
            const float d = dist[j];
            if (top[0].d > d || n_top < ef) {
                if (n_cand >= P.cand_cap) { overflow = true; break; }
                hn_push(cand, n_cand, -d, c);
                hn_push(top, n_top, d, c);
                if (n_top > ef) hn_pop(top, n_top);
                lower = top[0].d;
            }
        }
        if (overflow) break;
    }
    if (overflow) {                                                     // the host repeats the query with a larger heap
        if (lane == 0) { atomicAdd(P.n_overflow, 1u); P.nn_idx[o] = kNone - 1u; }
        return;
    }
    while (n_top > 2) hn_pop(top, n_top);
    // results as ArrayMatcher_hnsw::SearchNeighbours orders them: ascending (distance, row)
    HnPair r0{0.f, kNone}, r1{0.f, kNone};
    const uint32_t nr = n_top;
    if (nr >= 1) { r0 = top[0]; hn_pop(top, n_top); }
    if (nr == 2) {
        r1 = top[0];
        if (r1.d < r0.d || (!(r0.d < r1.d) && r1.id < r0.id)) { const HnPair t = r0; r0 = r1; r1 = t; }
    }
    if (lane == 0) {
        const bool two = nr == 2;
        P.nn_idx[o] = (two && r0.d < P.ratio_R * r1.d) ? r0.id : kNone;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = nr >= 1 ? (int32_t)r0.id : -1; P.knn_idx[2 * o + 1] = two ? (int32_t)r1.id : -1;
            P.knn_dist[2 * o] = nr >= 1 ? r0.d : R3DM_INF;   P.knn_dist[2 * o + 1] = two ? r1.d : R3DM_INF;
        }
        atomicAdd(P.n_comps, evals);
    }
}

// ------------------------------------------------------------------------------------------------------------ search, QW queries per wavefront
// The same procedure with a GROUP of 64 / QW lanes per query (QW = 2, 4, 8): lane l of an 8-lane subgroup is accumulator l of
// L2SqrSIMD16Ext as before, a group measures 8 / QW rows per step, and the sequential part -- the walk over the unvisited links, the
// two heaps -- is issued once per WAVEFRONT for QW queries instead of once per query.  Every query performs exactly the operations
// it performs in hnsw_search_kernel, in the same order, so the results are the same bits; the groups of a wavefront diverge (different
// list lengths and hop counts) and the hardware masks them.  State per query in LDS as above; the stride between queries is 8 bytes
// off a multiple of 128 so that the groups' accesses to the same offset (the heap roots) fall into different banks.  Eight queries
// per workgroup whatever QW (8 / QW wavefronts): the LDS of a workgroup bounds the queries in flight on a CU either way.
//
// DEVELOPER BUILD ONLY (R3DM_HNSW_QW): measured and not adopted.  ms of search per pair, precise preset (M 19, ef 15), round 4:
//                                        QW = 1     2      4      8
//     12 views x 8,192 LIOP-144 rows      0.455   0.429  0.577  0.829
//     16 views x 16,384 SIFT byte rows    0.910   1.008  1.360  1.923
// A query is a chain of dependent steps (pop -> link list -> rows -> heap moves in LDS) and the LDS state (3.9 KB) caps the queries in
// flight on a CU at ~40 however they are spread over wavefronts; packing them into fewer wavefronts removes the redundant issue slots
// but also the wavefronts that hid each other's latency.  The search is latency-bound per query, not issue-bound.
#ifdef R3DM_DEVTOOLS
template <int NB, typename ROW, int QW>
__global__ void __launch_bounds__(64 * (8 / QW)) hnsw_search_group_kernel(const HnswSearchParams P)
{
    constexpr uint32_t GL = 64 / QW, RS = GL / 8;                        // lanes per query, rows per step of a group
    extern __shared__ unsigned char hn_smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t grp = lane / GL, gl = lane % GL, sub = gl >> 3, l = gl & 7u;
    const HnswSearchJob job = P.jobs[blockIdx.y];
    const uint32_t slot = wave * QW + grp;
    const uint32_t qi = blockIdx.x * 8 + slot;
    unsigned char* base = hn_smem + (size_t)slot * P.per_query;
    uint32_t* visited = reinterpret_cast<uint32_t*>(base);
    HnPair* top = reinterpret_cast<HnPair*>(base + P.flag_words * 4);
    HnPair* cand = top + (P.ef + 1);
    float* dist = reinterpret_cast<float*>(cand + P.cand_cap);
    int32_t* ids = reinterpret_cast<int32_t*>(dist + 64);
    int32_t* todo = ids + 64;
    if (qi >= job.nq) return;                                            // (no workgroup barrier below; a group leaves as a whole)
    const HnswView ix = job.ix;
    const uint32_t dim = ix.dim, M = ix.M;
    const ROW* __restrict__ xrows = sizeof(ROW) == 1 ? reinterpret_cast<const ROW*>(ix.rows8) : reinterpret_cast<const ROW*>(ix.rows);
    const size_t o = (size_t)job.out_base + qi;
    const unsigned long long gmask = GL == 64 ? ~0ull : ((1ull << GL) - 1ull);

    float q[NB];
    hn_load_q<NB>(q, job.query + (size_t)qi * dim, l);
    for (uint32_t w = gl; w < P.flag_words; w += GL) visited[w] = 0;
    unsigned long long evals = 1;

    uint32_t cur = (uint32_t)ix.enter;
    float curdist = hn_hsum8(hn_acc<NB, ROW>(q, xrows + (size_t)cur * dim, l), lane);
    // greedy descent (hnswalg.h:745-768)
    for (int level = ix.maxlevel; level > 0; --level) {
        bool changed = true;
        while (changed) {
            changed = false;
            const int32_t* L = ix.up + ((size_t)ix.up_off[cur] + (uint32_t)(level - 1)) * (1 + M);
            const uint32_t size = (uint32_t)L[0];
            HN_SYNC();
            for (uint32_t j = gl; j < size; j += GL) ids[j] = L[1 + j];
            HN_SYNC();
            for (uint32_t j0 = 0; j0 < size; j0 += RS) {
                const uint32_t j = j0 + sub;
                const bool on = j < size;
                float acc = 0.f;
                if (on) acc = hn_acc<NB, ROW>(q, xrows + (size_t)(uint32_t)ids[j] * dim, l);
                const float d = hn_hsum8(acc, lane);
                if (on && l == 0) dist[j] = d;
            }
            HN_SYNC();
            evals += size;
            for (uint32_t j = 0; j < size; ++j) {
                const float d = dist[j];
                if (d < curdist) { curdist = d; cur = (uint32_t)ids[j]; changed = true; }
            }
        }
    }
    // searchBaseLayerST (hnswalg.h:214-280)
    const uint32_t ef = P.ef;
    uint32_t n_top = 0, n_cand = 0;
    bool overflow = false;
    HN_SYNC();
    visited[cur >> 5] |= 1u << (cur & 31u);                              // (the group's lanes write the same word)
    hn_push(top, n_top, curdist, cur);
    hn_push(cand, n_cand, -curdist, cur);
    float lower = curdist;
    HN_SYNC();
    while (n_cand) {
        const HnPair c0 = cand[0];
        if ((-c0.d) > lower) break;
        hn_pop(cand, n_cand);
        const int32_t* L = ix.l0 + (size_t)c0.id * (1 + 2 * M);
        const uint32_t size = (uint32_t)L[0];
        HN_SYNC();
        for (uint32_t j = gl; j < size; j += GL) ids[j] = L[1 + j];       // 2M <= 64 links
        HN_SYNC();
        // the unvisited links, packed (hn_dist_list_unvisited, per group)
        uint32_t n_walk = 0;
        for (uint32_t j0 = 0; j0 < size; j0 += GL) {
            const uint32_t j = j0 + gl;
            const bool on = j < size;
            const uint32_t mine = on ? (uint32_t)ids[j] : 0u;
            const bool need = on && ((visited[mine >> 5] >> (mine & 31u)) & 1u) == 0u;
            const unsigned long long m = (__ballot(need) >> (grp * GL)) & gmask;
            if (need) todo[n_walk + (uint32_t)__popcll(m & ((1ull << gl) - 1ull))] = (int32_t)j;
            n_walk += (uint32_t)__popcll(m);
        }
        HN_SYNC();
        for (uint32_t k0 = 0; k0 < n_walk; k0 += RS) {
            const uint32_t k = k0 + sub;
            const bool work = k < n_walk;
            const uint32_t j = work ? (uint32_t)todo[k] : 0u;
            float acc = 0.f;
            if (work) acc = hn_acc<NB, ROW>(q, xrows + (size_t)(uint32_t)ids[j] * dim, l);
            const float d = hn_hsum8(acc, lane);
            if (work && l == 0) dist[j] = d;
        }
        HN_SYNC();
        for (uint32_t k = 0; k < n_walk; ++k) {
            const uint32_t j = (uint32_t)todo[k];
            const uint32_t c = (uint32_t)ids[j];
            const uint32_t w = visited[c >> 5], bit = 1u << (c & 31u);
            if (w & bit) continue;                                     // earlier in this list
            visited[c >> 5] = w | bit;
            evals += 1;
            const float d = dist[j];
            if (top[0].d > d || n_top < ef) {
                if (n_cand >= P.cand_cap) { overflow = true; break; }
                hn_push(cand, n_cand, -d, c);
                hn_push(top, n_top, d, c);
                if (n_top > ef) hn_pop(top, n_top);
                lower = top[0].d;
            }
        }
        if (overflow) break;
    }
    if (overflow) {                                                     // the host repeats the launch with a larger heap
        if (gl == 0) { atomicAdd(P.n_overflow, 1u); P.nn_idx[o] = kNone - 1u; }
        return;
    }
    while (n_top > 2) hn_pop(top, n_top);
    HnPair r0{0.f, kNone}, r1{0.f, kNone};
    const uint32_t nr = n_top;
    if (nr >= 1) { r0 = top[0]; hn_pop(top, n_top); }
    if (nr == 2) {
        r1 = top[0];
        if (r1.d < r0.d || (!(r0.d < r1.d) && r1.id < r0.id)) { const HnPair t = r0; r0 = r1; r1 = t; }
    }
    if (gl == 0) {
        const bool two = nr == 2;
        P.nn_idx[o] = (two && r0.d < P.ratio_R * r1.d) ? r0.id : kNone;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = nr >= 1 ? (int32_t)r0.id : -1; P.knn_idx[2 * o + 1] = two ? (int32_t)r1.id : -1;
            P.knn_dist[2 * o] = nr >= 1 ? r0.d : R3DM_INF;   P.knn_dist[2 * o + 1] = two ? r1.d : R3DM_INF;
        }
        atomicAdd(P.n_comps, evals);
    }
}
#endif  // R3DM_DEVTOOLS

// ------------------------------------------------------------------------------------------------------------ construction
__device__ __forceinline__ unsigned long long hn_key(float d, uint32_t id) { return ((unsigned long long)__float_as_uint(d) << 32) | id; }
__device__ __forceinline__ float hn_key_d(unsigned long long k) { return __uint_as_float((uint32_t)(k >> 32)); }

// one row against one row, all eight accumulators in ONE lane (the heuristic compares a candidate with every kept row: a lane per kept row)
__device__ __forceinline__ float hn_l2_lane(const float* __restrict__ a, const float* __restrict__ b, uint32_t dim)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = 0; i < dim; i += 8) {
        const float4 a0 = *reinterpret_cast<const float4*>(a + i), a1 = *reinterpret_cast<const float4*>(a + i + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + i), b1 = *reinterpret_cast<const float4*>(b + i + 4);
        float t;
        t = a0.x - b0.x; acc[0] = acc[0] + t * t;  t = a0.y - b0.y; acc[1] = acc[1] + t * t;
        t = a0.z - b0.z; acc[2] = acc[2] + t * t;  t = a0.w - b0.w; acc[3] = acc[3] + t * t;
        t = a1.x - b1.x; acc[4] = acc[4] + t * t;  t = a1.y - b1.y; acc[5] = acc[5] + t * t;
        t = a1.z - b1.z; acc[6] = acc[6] + t * t;  t = a1.w - b1.w; acc[7] = acc[7] + t * t;
    }
    return acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
}

template <int NB>
__global__ void __launch_bounds__(256) hnsw_link_kernel(const HnswBuildParams P)
{
    __shared__ unsigned long long s_key[4][64];     // candidates (ascending), later the kept rows
    __shared__ unsigned long long s_ret[4][64];
    __shared__ unsigned long long s_pruned[4][64];
    __shared__ float s_dist[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, l = lane & 7u, g = lane >> 3;
    const HnswBuildJob job = P.jobs[blockIdx.y];
    const uint32_t t = blockIdx.x * 4 + wave;
    if (t >= job.n + job.up_rows) return;
    const uint32_t dim = job.dim, M = job.M;
    unsigned long long* key = s_key[wave];
    unsigned long long* ret = s_ret[wave];
    unsigned long long* pruned = s_pruned[wave];
    float* dist = s_dist[wave];

    uint32_t node, limit, nc;
    int32_t* out;
    float q[NB];
    if (t < job.n) {
        // layer 0: the row's list in the exact 32-NN graph (+ reverse edges, 64 closest), re-measured with hnswlib's distance
        node = t; limit = 2 * M; out = job.l0 + (size_t)node * (1 + 2 * M);
        hn_load_q<NB>(q, job.rows + (size_t)node * dim, l);
        nc = min(job.adj_deg[node], 64u);
        const int32_t* ids = reinterpret_cast<const int32_t*>(job.adj + (size_t)node * kAnnDeg);
        hn_dist_list<NB, float>(q, job.rows, dim, ids, nc, dist, lane, [](uint32_t) { return false; });
        HN_SYNC();
        const unsigned long long mine = lane < nc ? hn_key(dist[lane], (uint32_t)ids[lane]) : ~0ull;
        key[lane] = mine;
        HN_SYNC();
        uint32_t rank = 0;
        for (uint32_t j = 0; j < 64; ++j) rank += key[j] < mine ? 1u : 0u;       // rows are unique, so are the keys
        HN_SYNC();
        if (lane < nc) key[rank] = mine;
        HN_SYNC();
    } else {
        // layer L > 0: the 32 nearest members of the layer by (distance, row), exact scan; lanes 0 .. 31 hold the list in order
        const uint32_t r = t - job.n;
        node = job.up_node[r];
        const uint32_t L = job.up_level[r];
        limit = M; out = job.up + (size_t)r * (1 + M);
        hn_load_q<NB>(q, job.rows + (size_t)node * dim, l);
        const uint32_t* mem = job.members + job.mem_off[L - 1];
        const uint32_t nm = job.mem_off[L] - job.mem_off[L - 1];
        const uint32_t K = min(32u, nm - 1);
        unsigned long long mine = ~0ull;
        for (uint32_t b0 = 0; b0 < nm; b0 += 8) {
            const uint32_t b = b0 + g;
            const bool on = b < nm;
            const uint32_t c = on ? mem[b] : node;
            const bool work = on && c != node;
            float acc = 0.f;
            if (work) acc = hn_acc<NB, float>(q, job.rows + (size_t)c * dim, l);
            const float d = hn_hsum8(acc, lane);
            const unsigned long long e_mine = work ? hn_key(d, c) : ~0ull;
            for (uint32_t gg = 0; gg < 8; ++gg) {
                const unsigned long long e = __shfl(e_mine, (int)(gg * 8));
                const unsigned long long worst = __shfl(mine, (int)(K ? K - 1 : 0));
                if (K == 0 || e >= worst) continue;                               // wave-uniform
                const uint32_t pos = (uint32_t)__popcll(__ballot(lane < K && mine < e));
                const unsigned long long up = __shfl_up(mine, 1);
                if (lane < K && lane > pos) mine = up;
                if (lane == pos) mine = e;
            }
        }
        nc = (uint32_t)__popcll(__ballot(lane < K && mine != ~0ull));
        key[lane] = lane < nc ? mine : ~0ull;
        HN_SYNC();
    }

    // getNeighborsByHeuristic2 over key[0 .. nc) (ascending); a list shorter than the limit is kept whole
    uint32_t nret = 0, npr = 0;
    if (nc < limit) { if (lane < nc) ret[lane] = key[lane]; nret = nc; }
    else {
        for (uint32_t k = 0; k < nc && nret < limit; ++k) {
            const unsigned long long ck = key[k];
            const uint32_t cid = __builtin_amdgcn_readfirstlane((uint32_t)ck);
            const float dq = hn_key_d(ck);
            bool bad = false;
            if (lane < nret) bad = hn_l2_lane(job.rows + (size_t)(uint32_t)ret[lane] * dim, job.rows + (size_t)cid * dim, dim) < dq;
            const bool good = __ballot(bad) == 0ull;
            HN_SYNC();
            if (lane == 0) { if (good) ret[nret] = ck; else pruned[npr] = ck; }
            if (good) ++nret; else ++npr;
            HN_SYNC();
        }
        // free places go to the closest rejected candidates; then ascending order again
        const uint32_t fill = min(npr, limit - nret);
        if (lane < fill) ret[nret + lane] = pruned[lane];
        nret += fill;
        HN_SYNC();
        const unsigned long long mine = lane < nret ? ret[lane] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nret; ++j) rank += ret[j] < mine ? 1u : 0u;
        HN_SYNC();
        if (lane < nret) ret[rank] = mine;
        HN_SYNC();
    }
    HN_SYNC();
    if (lane == 0) out[0] = (int32_t)nret;
    if (lane < nret) out[1 + lane] = (int32_t)(uint32_t)ret[nret - 1 - lane];      // farthest first
    else if (lane < limit) out[1 + lane] = -1;
}

}  // namespace

hipError_t launch_hnsw_search(hipStream_t st, const HnswSearchParams& Pin, uint32_t max_nq, uint32_t max_n, uint32_t dim)
{
    HnswSearchParams P = Pin;
    if (P.n_jobs == 0 || max_nq == 0) return hipSuccess;
    if (P.ef < 2 || P.ef > 512 || P.cand_cap < 16) return hipErrorInvalidValue;
    if (P.n_jobs > 65535u) return hipErrorInvalidValue;
    P.flag_words = (max_n + 31) / 32;
    const size_t state = (size_t)P.flag_words * 4 + (size_t)(P.ef + 1) * 8 + (size_t)P.cand_cap * 8 + 768;
    const size_t per_query = (state + 127) / 128 * 128 + 8;
    uint32_t qw = 1;                                                     // a wavefront per query, four per workgroup
#ifdef R3DM_DEVTOOLS
    if (P.queries_per_wave == 2 || P.queries_per_wave == 4 || P.queries_per_wave == 8) qw = P.queries_per_wave;   // groups, eight queries per workgroup
    if (qw > 1 && per_query * 8 > 160 * 1024) qw = 1;
#endif
    P.per_query = (uint32_t)per_query;
    const size_t lds = qw > 1 ? per_query * 8 : state * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid(qw > 1 ? (max_nq + 7) / 8 : (max_nq + 3) / 4, P.n_jobs);
#define R3DM_HNSW_LAUNCH(KERNEL, THREADS)                                                                              \
    do {                                                                                                               \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), lds, st, P);                                                    \
    } while (0)
#ifdef R3DM_DEVTOOLS
#define R3DM_HNSW_SEARCH_T(NB, ROW)                                                                                     \
    do {                                                                                                               \
        if (qw == 2) R3DM_HNSW_LAUNCH((hnsw_search_group_kernel<NB, ROW, 2>), 256);                                     \
        else if (qw == 4) R3DM_HNSW_LAUNCH((hnsw_search_group_kernel<NB, ROW, 4>), 128);                                \
        else if (qw == 8) R3DM_HNSW_LAUNCH((hnsw_search_group_kernel<NB, ROW, 8>), 64);                                 \
        else R3DM_HNSW_LAUNCH((hnsw_search_kernel<NB, ROW>), 256);                                                      \
    } while (0)
#else
#define R3DM_HNSW_SEARCH_T(NB, ROW) R3DM_HNSW_LAUNCH((hnsw_search_kernel<NB, ROW>), 256)
#endif
#define R3DM_HNSW_SEARCH(NB) do { if (P.rows8) R3DM_HNSW_SEARCH_T(NB, uint8_t); else R3DM_HNSW_SEARCH_T(NB, float); } while (0)
    switch (dim) {
        case 64:  R3DM_HNSW_SEARCH(8); break;
        case 128: R3DM_HNSW_SEARCH(16); break;
        case 144: R3DM_HNSW_SEARCH(18); break;
        case 256: R3DM_HNSW_SEARCH(32); break;
        default: return hipErrorInvalidValue;
    }
#undef R3DM_HNSW_SEARCH
#undef R3DM_HNSW_SEARCH_T
#undef R3DM_HNSW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_hnsw_link(hipStream_t st, const HnswBuildParams& P, uint32_t n_jobs, uint32_t max_items, uint32_t dim)
{
    if (n_jobs == 0 || max_items == 0) return hipSuccess;
    if (n_jobs > 65535u) return hipErrorInvalidValue;
    const dim3 grid((max_items + 3) / 4, n_jobs);
    switch (dim) {
        case 64:  hipLaunchKernelGGL((hnsw_link_kernel<8>), grid, dim3(256), 0, st, P); break;
        case 128: hipLaunchKernelGGL((hnsw_link_kernel<16>), grid, dim3(256), 0, st, P); break;
        case 144: hipLaunchKernelGGL((hnsw_link_kernel<18>), grid, dim3(256), 0, st, P); break;
        case 256: hipLaunchKernelGGL((hnsw_link_kernel<32>), grid, dim3(256), 0, st, P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace r3dm
