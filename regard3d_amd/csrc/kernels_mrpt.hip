// kernels_mrpt.hip -- the MRPT plugin path (matchingAlgorithm 5 of /root/reference/src/R3DComputeMatches.cpp:2035-2062;
// mrpt_match :423-491, ArrayMatcher_mrpt src/utils/matcher_mrpt.h:45-259, index src/thirdparty/mrpt/mrpt.h).
//
//   mrpt_project_kernel   the rows of a view projected on the n_trees x depth random vectors (Mrpt::grow, mrpt.h:124-131): one thread
//                         per (row, tree), the non-zero terms of a vector in ascending column order, float, no FMA -- the dense
//                         loop adds x * 0 for the entries the sparse matrix does not have, which changes nothing
//   mrpt_tree_kernel      grow_subtree (mrpt.h:1051-1078) for one tree per workgroup, level by level: ONE sort of all rows per level
//                         by (node, projection, row) -- the nodes of a level are contiguous ranges whose sizes depend on n only
//                         (n - n/2 left, n/2 right), so the sorted sequence IS the next level's arrangement: the left child is the
//                         first half of its node's range.  Split point: the median row's projection (odd count) or the mean of the
//                         two rows either side of the cut, as the reference computes it
//   mrpt_query_kernel     Mrpt::query (mrpt.h:661-728) with k = 2 and ArrayMatcher_mrpt's retry (matcher_mrpt.h:224-232), one
//                         wavefront per query: project, route to one leaf per tree, count votes in LDS (a byte per row), collect the
//                         rows that reach the threshold, measure them with the reference's brute-force metric, keep the two
//                         nearest by (distance, row); distances leave as sqrtf, the ratio test runs on them un-squared
//                         (RegionsMatcherT(regions, false): b_squared_metric = false, src/R3DComputeMatches.cpp:461)
// CPU model, bit for bit: oracle/mrpt.c (its header lists where the restatement departs from a reference build and why).
#include "r3dm_internal.hpp"

namespace r3dm {
namespace {

constexpr uint32_t kNoneM = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t mr_order_bits(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float mr_from_order_bits(uint32_t b)
{
    return __uint_as_float((b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b);
}
// global-memory exchange INSIDE a workgroup (same recipe as kernels_filter.hip: wg_fence): the waves of a workgroup share one CU
// and its L1, a workgroup-scope fence + an explicit wait for outstanding stores and loads is enough (no -mtgsplit: build.sh)
__device__ __forceinline__ void mr_sync_global()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the brute-force metric of the reference in its own summation order (kernels_match.hip: exact_l2sq; oracle: orc_l2sq_f32)
__device__ __forceinline__ float mr_l2sq(const float* __restrict__ a, const float* __restrict__ b, uint32_t dim)
{
    float result = 0.0f;
    uint32_t k = 0;
    for (; k + 3 < dim; k += 4) {
        const float d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; k < dim; ++k) { const float d0 = a[k] - b[k]; result += d0 * d0; }
    return result;
}

// ------------------------------------------------------------------------------------------------ projections
// grid (ceil(n / 256), n_trees); proj [n_trees][depth][n]
__global__ __launch_bounds__(256) void mrpt_project_kernel(const float* __restrict__ rows, uint32_t n, uint32_t dim, const float* __restrict__ R,
                                                           uint32_t depth, float* __restrict__ proj)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, t = blockIdx.y;
    if (i >= n) return;
    const float* x = rows + (size_t)i * dim;
    const float* Rt = R + (size_t)t * depth * dim;
    float acc[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) acc[l] = 0.0f;
    for (uint32_t c = 0; c < dim; ++c) {
        const float xv = x[c];
#pragma unroll
        for (int l = 0; l < 8; ++l)
            if ((uint32_t)l < depth) { const float tt = Rt[(size_t)l * dim + c] * xv; acc[l] = acc[l] + tt; }
    }
#pragma unroll
    for (int l = 0; l < 8; ++l)
        if ((uint32_t)l < depth) proj[((size_t)t * depth + (uint32_t)l) * n + i] = acc[l];
}

// ------------------------------------------------------------------------------------------------ trees
constexpr uint32_t kMrNT = 1024;          // threads of a tree's workgroup
constexpr uint32_t kMrChunk = 4096;       // keys sorted in LDS at a time (32 KB)

// node of position p at `level` (0 .. 2^level - 1) and its range [lo, lo + len)
__device__ __forceinline__ void mr_node_of(uint32_t p, uint32_t n, uint32_t level, uint32_t& node, uint32_t& lo, uint32_t& len)
{
    node = 0; lo = 0; len = n;
    for (uint32_t d = 0; d < level; ++d) {
        const uint32_t left = len - len / 2u;
        if (p - lo < left) { len = left; node = 2u * node; }
        else { lo += left; len = len / 2u; node = 2u * node + 1u; }
    }
}

// compare-exchange passes of the bitonic network on the chunk in LDS: strides first .. 1 of the merge step `size` (global index = base + local)
__device__ __forceinline__ void mr_lds_passes(unsigned long long* lk, uint32_t base, uint32_t size, uint32_t first_stride, uint32_t tid)
{
    for (uint32_t stride = first_stride; stride >= 1u; stride >>= 1) {
        for (uint32_t e = tid; e < kMrChunk / 2u; e += kMrNT) {
            const uint32_t lo = 2u * e - (e & (stride - 1u)), hi = lo + stride;
            const bool up = (((base + lo) & size) == 0u);
            const unsigned long long a = lk[lo], b = lk[hi];
            if ((a > b) == up) { lk[lo] = b; lk[hi] = a; }
        }
        __syncthreads();
    }
}

// ascending sort of keys[0 .. cap) (cap a power of two, padding = ~0) by one workgroup
__device__ void mr_sort(unsigned long long* __restrict__ keys, uint32_t cap, unsigned long long* lk, uint32_t tid)
{
    if (cap <= kMrChunk) {
        for (uint32_t e = tid; e < cap; e += kMrNT) lk[e] = keys[e];
        for (uint32_t e = cap + tid; e < kMrChunk; e += kMrNT) lk[e] = ~0ull;
        __syncthreads();
        for (uint32_t size = 2u; size <= kMrChunk; size <<= 1) mr_lds_passes(lk, 0u, size, size >> 1, tid);
        for (uint32_t e = tid; e < cap; e += kMrNT) keys[e] = lk[e];
        mr_sync_global();
        return;
    }
    for (uint32_t c0 = 0; c0 < cap; c0 += kMrChunk) {                         // every chunk sorted, alternating directions
        for (uint32_t e = tid; e < kMrChunk; e += kMrNT) lk[e] = keys[c0 + e];
        __syncthreads();
        for (uint32_t size = 2u; size <= kMrChunk; size <<= 1) mr_lds_passes(lk, c0, size, size >> 1, tid);
        for (uint32_t e = tid; e < kMrChunk; e += kMrNT) keys[c0 + e] = lk[e];
        __syncthreads();
    }
    mr_sync_global();
    for (uint32_t size = 2u * kMrChunk; size <= cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride >= kMrChunk; stride >>= 1) { // strides that cross chunks: through global memory
            for (uint32_t e = tid; e < cap / 2u; e += kMrNT) {
                const uint32_t lo = 2u * e - (e & (stride - 1u)), hi = lo + stride;
                const bool up = ((lo & size) == 0u);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            mr_sync_global();
        }
        for (uint32_t c0 = 0; c0 < cap; c0 += kMrChunk) {
            for (uint32_t e = tid; e < kMrChunk; e += kMrNT) lk[e] = keys[c0 + e];
            __syncthreads();
            mr_lds_passes(lk, c0, size, kMrChunk >> 1, tid);
            for (uint32_t e = tid; e < kMrChunk; e += kMrNT) keys[c0 + e] = lk[e];
            __syncthreads();
        }
        mr_sync_global();
    }
}

// one workgroup per tree.  keys: [n_trees][cap] scratch; leaves [n_trees][n]; splits [n_trees][2^depth - 1]
__global__ __launch_bounds__(kMrNT) void mrpt_tree_kernel(const float* __restrict__ proj, uint32_t n, uint32_t depth, uint32_t cap,
                                                          unsigned long long* __restrict__ keys_all, int32_t* __restrict__ leaves,
                                                          float* __restrict__ splits)
{
    __shared__ unsigned long long lk[kMrChunk];
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    unsigned long long* keys = keys_all + (size_t)t * cap;
    int32_t* idx = leaves + (size_t)t * n;
    const uint32_t n_nodes_all = (1u << depth) - 1u;
    for (uint32_t p = tid; p < n; p += kMrNT) idx[p] = (int32_t)p;
    mr_sync_global();
    for (uint32_t level = 0; level < depth; ++level) {
        const float* pl = proj + ((size_t)t * depth + level) * n;
        for (uint32_t p = tid; p < cap; p += kMrNT) {
            unsigned long long k = ~0ull;
            if (p < n) {
                uint32_t node, lo, len;
                mr_node_of(p, n, level, node, lo, len);
                const uint32_t r = (uint32_t)idx[p];
                k = ((unsigned long long)node << 58) | ((unsigned long long)mr_order_bits(pl[r]) << 26) | (unsigned long long)r;
            }
            keys[p] = k;
        }
        mr_sync_global();
        mr_sort(keys, cap, lk, tid);
        for (uint32_t p = tid; p < n; p += kMrNT) idx[p] = (int32_t)(uint32_t)(keys[p] & 0x3FFFFFFull);
        // split points of this level's nodes
        for (uint32_t s = tid; s < (1u << level); s += kMrNT) {
            uint32_t lo = 0, len = n;
            for (uint32_t d = 0; d < level; ++d) {                            // node s: its bits from the top say left / right
                const uint32_t left = len - len / 2u;
                if (((s >> (level - 1u - d)) & 1u) == 0u) len = left; else { lo += left; len = len / 2u; }
            }
            const uint32_t n_left = len - len / 2u;
            const float vl = mr_from_order_bits((uint32_t)(keys[lo + n_left - 1u] >> 26));
            float split = vl;
            if ((len & 1u) == 0u) { const float vr = mr_from_order_bits((uint32_t)(keys[lo + n_left] >> 26)); const float sum = vr + vl; split = sum * 0.5f; }
            splits[(size_t)t * n_nodes_all + ((1u << level) - 1u) + s] = split;
        }
        mr_sync_global();
    }
}

// ------------------------------------------------------------------------------------------------ queries
template <typename T> __device__ __forceinline__ T mr_shfl_xor(T v, int m) { return __shfl_xor(v, m); }

// LDS per wave: pq[n_pool_pad] | leaf[256] | cnt[2] | elected[elected_cap] | votes[ceil(n / 4)] words
__global__ __launch_bounds__(256) void mrpt_query_kernel(const MrptQueryParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char mr_smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const MrptQueryJob job = P.jobs[blockIdx.y];
    const uint32_t qi = blockIdx.x * P.waves + wave;
    if (qi >= job.nq) return;                                                 // (no workgroup barrier below)
    const MrptView ix = job.ix;
    const uint32_t n = ix.n, dim = ix.dim, depth = ix.depth, n_trees = ix.n_trees, n_pool = n_trees * depth;
    unsigned char* base = mr_smem + (size_t)wave * P.per_wave;
    float* pq = reinterpret_cast<float*>(base);
    uint32_t* leaf = reinterpret_cast<uint32_t*>(base + P.pool_pad * 4u);
    uint32_t* cnt = leaf + 256;                                               // (n_trees <= 255)
    uint32_t* elected = cnt + 2;
    uint32_t* votes = elected + P.elected_cap;
    const uint32_t vote_words = (P.max_n + 3u) / 4u;
    const float* q = job.query + (size_t)qi * dim;
    const size_t o = (size_t)job.out_base + qi;

    // project: output j = lane, lane + 64, ...; RT [dim][n_pool]: the lanes read consecutive j
    for (uint32_t j = lane; j < n_pool; j += 64u) {
        float acc = 0.0f;
        for (uint32_t c = 0; c < dim; ++c) { const float tt = ix.RT[(size_t)c * n_pool + j] * q[c]; acc = acc + tt; }
        pq[j] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // route: a lane per tree
    const uint32_t n_nodes_all = (1u << depth) - 1u;
    for (uint32_t t = lane; t < n_trees; t += 64u) {
        uint32_t node = 0;
        for (uint32_t d = 0; d < depth; ++d) {
            const float sp = ix.splits[(size_t)t * n_nodes_all + node];
            node = (pq[t * depth + d] <= sp) ? 2u * node + 1u : 2u * node + 2u;
        }
        leaf[t] = node - n_nodes_all;
    }
    uint32_t i0 = kNoneM, i1 = kNoneM; float d0 = 0.f, d1 = 0.f;
    uint32_t need = P.votes;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (uint32_t w = lane; w < vote_words; w += 64u) votes[w] = 0u;
        if (lane == 0) cnt[0] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t t = 0; t < n_trees; ++t) {
            const uint32_t lf = leaf[t];
            const uint32_t b = (uint32_t)ix.leaf_first[lf], e = (uint32_t)ix.leaf_first[lf + 1u];
            const int32_t* rows_t = ix.leaves + (size_t)t * n;
            for (uint32_t a = b + lane; a < e; a += 64u) {
                const uint32_t r = (uint32_t)rows_t[a];
                const uint32_t sh = 8u * (r & 3u);
                const uint32_t old = atomicAdd(&votes[r >> 2], 1u << sh);     // a row occurs once per tree: no two lanes of this step share a byte
                if (((old >> sh) & 255u) + 1u == need) { const uint32_t pos = atomicAdd(&cnt[0], 1u); if (pos < P.elected_cap) elected[pos] = r; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t ne = min(cnt[0], P.elected_cap);
        // the two nearest of the elected rows by (distance, row)
        i0 = kNoneM; i1 = kNoneM; d0 = 0.f; d1 = 0.f;
        for (uint32_t k = lane; k < ne; k += 64u) {
            const uint32_t r = elected[k];
            const float d = mr_l2sq(ix.rows + (size_t)r * dim, q, dim);
            if (i0 == kNoneM || d < d0 || (d == d0 && r < i0)) { i1 = i0; d1 = d0; i0 = r; d0 = d; }
            else if (i1 == kNoneM || d < d1 || (d == d1 && r < i1)) { i1 = r; d1 = d; }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const uint32_t oi0 = mr_shfl_xor(i0, m), oi1 = mr_shfl_xor(i1, m);
            const float od0 = mr_shfl_xor(d0, m), od1 = mr_shfl_xor(d1, m);
            // merge two sorted pairs (mine, other) -> the two smallest by (d, i)
            uint32_t c_i[4] = {i0, i1, oi0, oi1}; float c_d[4] = {d0, d1, od0, od1};
            uint32_t b0 = kNoneM, b1 = kNoneM; float e0 = 0.f, e1 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t r = c_i[u]; const float d = c_d[u];
                if (r == kNoneM) continue;
                if (b0 == kNoneM || d < e0 || (d == e0 && r < b0)) { b1 = b0; e1 = e0; b0 = r; e0 = d; }
                else if (b1 == kNoneM || d < e1 || (d == e1 && r < b1)) { b1 = r; e1 = d; }
            }
            i0 = b0; i1 = b1; d0 = e0; d1 = e1;
        }
        if (lane == 0) atomicAdd(P.n_comps, (unsigned long long)ne);
        if ((i0 != kNoneM && i1 != kNoneM) || need <= 1u) break;               // wave-uniform
        need -= 1u;                                                           // ArrayMatcher_mrpt: "Try again" with votes - 1
    }
    if (lane == 0) {
        const bool two = i0 != kNoneM && i1 != kNoneM;
        const float s0 = two ? sqrtf(d0) : -1.0f, s1 = two ? sqrtf(d1) : -1.0f;
        P.nn_idx[o] = (two && s0 < P.ratio * s1) ? i0 : kNoneM;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = two ? (int32_t)i0 : -1; P.knn_idx[2 * o + 1] = two ? (int32_t)i1 : -1;
            P.knn_dist[2 * o] = s0; P.knn_dist[2 * o + 1] = s1;
        }
    }
}

}  // namespace

hipError_t launch_mrpt_project(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, const float* R, uint32_t n_trees, uint32_t depth, float* proj)
{
    if (n == 0 || n_trees == 0) return hipSuccess;
    if (depth < 1 || depth > 8 || n_trees > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mrpt_project_kernel, dim3((n + 255u) / 256u, n_trees), dim3(256), 0, st, rows, n, dim, R, depth, proj);
    return hipGetLastError();
}

hipError_t launch_mrpt_trees(hipStream_t st, const float* proj, uint32_t n, uint32_t n_trees, uint32_t depth, uint32_t cap, unsigned long long* keys,
                             int32_t* leaves, float* splits)
{
    if (n == 0 || n_trees == 0) return hipSuccess;
    if (n >= (1u << 26) || depth > 6 || cap < n || (cap & (cap - 1u)) != 0u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mrpt_tree_kernel, dim3(n_trees), dim3(kMrNT), 0, st, proj, n, depth, cap, keys, leaves, splits);
    return hipGetLastError();
}

// max_nq: most queries of a job; max_n: most rows of an index view of the batch; max_pool: largest n_trees x depth
hipError_t launch_mrpt_query(hipStream_t st, const MrptQueryParams& Pin, uint32_t max_nq, uint32_t max_n, uint32_t max_pool)
{
    MrptQueryParams P = Pin;
    if (P.n_jobs == 0 || max_nq == 0) return hipSuccess;
    if (P.n_jobs > 65535u || P.votes < 1 || P.votes > 255u) return hipErrorInvalidValue;
    P.max_n = max_n;
    P.pool_pad = (max_pool + 63u) / 64u * 64u;
    const size_t per_wave = ((size_t)P.pool_pad * 4 + 256 * 4 + 8 + (size_t)P.elected_cap * 4 + (size_t)((max_n + 3u) / 4u) * 4 + 15) / 16 * 16;
    uint32_t waves = 4;
    while (waves > 1 && per_wave * waves > 150u * 1024u) waves >>= 1;
    if (per_wave * waves > 160u * 1024u) return hipErrorInvalidValue;
    P.per_wave = (uint32_t)per_wave; P.waves = waves;
    const size_t lds = per_wave * waves;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mrpt_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mrpt_query_kernel, dim3((max_nq + waves - 1u) / waves, P.n_jobs), dim3(64u * waves), lds, st, P);
    return hipGetLastError();
}

}  // namespace r3dm
