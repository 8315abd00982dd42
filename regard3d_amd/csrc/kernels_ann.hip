// kernels_ann.hip -- graph-based approximate 2-NN for gfx950 (BASELINE config C5: the KGraph plugin path).
//
// Replaces, for f32 descriptors, what the reference does per first image I in kgraph_match
// (/root/reference/src/R3DComputeMatches.cpp:808-902): ArrayMatcher_kgraph::Build (NN-descent index,
// src/utils/matcher_kgraph.h:138-153 -> src/thirdparty/kgraph/kgraph.cpp:703-997) and, per query row of J,
// KGraphImpl::search (kgraph.cpp:411-552) behind SearchNeighbours (matcher_kgraph.h:204-251).
//
// MI355X design (DESIGN.md "ANN"):
//   index  = EXACT K-NN graph of the view (an all-pairs scan is ~1 ms per 16k-row view here, cheaper and
//            better than 30 rounds of lock-based NN-descent), completed with reverse edges like
//            KGraph::reverse(-1), every adjacency list ordered by (distance, id), unique, cut to the 64
//            closest -> fixed [n][64] u32 rows, one 256 B line per expanded node;
//   search = the reference's pool expansion, one wavefront per query: the pool (K + P entries) lives one
//            entry per lane, 16 candidates of the expanded node are scored at once by 4-lane groups, the
//            visited set is a bitset in LDS.  Distances are formed in the reference's arithmetic (OpenMVG
//            L2<float>: 4-way groups, sequential float accumulation, no FMA), so the search is bit-identical
//            to the CPU restatement (oracle/kgraph.c) run on the same index and start rows.
#include "r3dm_internal.hpp"

namespace r3dm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cf32p;        // constant address space -> SMEM loads

namespace {

__device__ __forceinline__ uint64_t ann_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ uint64_t ann_rng_u64(uint64_t seed, uint32_t I, uint32_t J, uint32_t a, uint32_t b)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    const uint64_t x = ann_mix64(seed + G * (1ULL + (((uint64_t)I << 32) | (uint64_t)J)));
    return ann_mix64(x + G * (1ULL + (((uint64_t)a << 32) | (uint64_t)b)));
}

// ((d0^2 + d1^2) + d2^2) + d3^2 of one 4-element group: the addend of OpenMVG's unrolled L2 loop
__device__ __forceinline__ float group_sq(const f32x4 x, const f32x4 y)
{
    const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
    return d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
}

__device__ __forceinline__ unsigned long long key_of(float d, uint32_t id)
{
    return ((unsigned long long)__float_as_uint(d) << 32) | id;          // d >= 0: integer order = (distance, id) order
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// index, step 1: exact K nearest rows of every row of one view (self excluded, ties -> lowest row).
// One thread per row; the scanned row is wave-uniform and arrives through the scalar cache, the
// thread's own row sits in VGPRs (G4 > 0) or in LDS (G4 == 0: any dim % 4 == 0), its K-list in LDS.
// ------------------------------------------------------------------------------------------------
template <int G4, int THREADS>
__global__ __launch_bounds__(THREADS)
void ann_knn_rows_kernel(const AnnBuildParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ann_smem[];
    const AnnBuildJob job = P.jobs[blockIdx.y];
    const ImgDev* __restrict__ im = P.imgs + job.slot;
    const uint32_t n = im->n, dim = im->dim, g4 = dim >> 2;
    const uint32_t row = blockIdx.x * THREADS + threadIdx.x;
    if (blockIdx.x * THREADS >= n) return;
    const uint32_t K = P.K;
    unsigned long long* list = (unsigned long long*)ann_smem;            // [K][THREADS]
    float* qs = (float*)(ann_smem + (size_t)K * THREADS * 8);            // [g4*4][THREADS] when G4 == 0
    const uint32_t my = row < n ? row : n - 1;
    const f32x4* src = (const f32x4*)(im->rows + (size_t)my * dim);

    f32x4 q[G4 > 0 ? G4 : 1];
    if constexpr (G4 > 0) {
#pragma unroll
        for (int g = 0; g < G4; ++g) q[g] = src[g];
    } else {
        for (uint32_t g = 0; g < g4; ++g) {
            const f32x4 v = src[g];
            qs[(4 * g + 0) * THREADS + threadIdx.x] = v[0]; qs[(4 * g + 1) * THREADS + threadIdx.x] = v[1];
            qs[(4 * g + 2) * THREADS + threadIdx.x] = v[2]; qs[(4 * g + 3) * THREADS + threadIdx.x] = v[3];
        }
    }
    for (uint32_t k = 0; k < K; ++k) list[k * THREADS + threadIdx.x] = ~0ull;
    unsigned long long worst = ~0ull;

    const cf32p base = (cf32p)(uintptr_t)im->rows;
    for (uint32_t r = 0; r < n; ++r) {
        const cf32p a = base + (size_t)r * dim;
        float acc = 0.0f;
        if constexpr (G4 > 0) {
#pragma unroll
            for (int g = 0; g < G4; ++g) {
                const f32x4 x = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                acc += group_sq(x, q[g]);
            }
        } else {
            for (uint32_t g = 0; g < g4; ++g) {
                const f32x4 x = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                const f32x4 y = {qs[(4 * g + 0) * THREADS + threadIdx.x], qs[(4 * g + 1) * THREADS + threadIdx.x],
                                 qs[(4 * g + 2) * THREADS + threadIdx.x], qs[(4 * g + 3) * THREADS + threadIdx.x]};
                acc += group_sq(x, y);
            }
        }
        const unsigned long long key = (r == my) ? ~0ull : key_of(acc, r);
        if (key < worst) {
            uint32_t k = K - 1;
            while (k > 0) {
                const unsigned long long up = list[(k - 1) * THREADS + threadIdx.x];
                if (up <= key) break;
                list[k * THREADS + threadIdx.x] = up;
                --k;
            }
            list[k * THREADS + threadIdx.x] = key;
            worst = list[(K - 1) * THREADS + threadIdx.x];
        }
    }
    if (row < n) {
        unsigned long long* out = job.fwd + (size_t)row * K;
        for (uint32_t k = 0; k < K; ++k) out[k] = list[k * THREADS + threadIdx.x];
    }
}

// The same scan for views of bytes (AnnBuildJob::rows8: integers 0 .. 255, D <= 256): ||a - q||^2 = ||a||^2 - 2 a.q + ||q||^2 with
// a.q from v_dot4_u32_u8 -- the scanned row is wave-uniform (scalar registers), the thread's own row sits packed in VGPRs; every
// value is an integer below 2^24, so the float key equals the one the f32 scan accumulates, and the lists are identical.  D / 4 + 4
// vector instructions per scanned row instead of 3 D.
typedef const __attribute__((address_space(4))) uint32_t* cu32p;
template <int W /* u32 words per row */>
__global__ __launch_bounds__(256)
void ann_knn_rows8_kernel(const AnnBuildParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ann_smem[];
    const AnnBuildJob job = P.jobs[blockIdx.y];
    const ImgDev* __restrict__ im = P.imgs + job.slot;
    const uint32_t n = im->n;
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const uint32_t K = P.K;
    unsigned long long* list = (unsigned long long*)ann_smem;            // [K][256]
    const uint32_t my = row < n ? row : n - 1;
    uint32_t q[W];
    uint32_t qq = 0;
    {
        const u32x4* src = (const u32x4*)(job.rows8 + (size_t)my * (W * 4));
#pragma unroll
        for (int g = 0; g < W / 4; ++g) {
            const u32x4 v = src[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) { q[4 * g + k] = v[k]; qq = __builtin_amdgcn_udot4(v[k], v[k], qq, false); }
        }
    }
    for (uint32_t k = 0; k < K; ++k) list[k * 256 + threadIdx.x] = ~0ull;
    unsigned long long worst = ~0ull;
    const cu32p base = (cu32p)(uintptr_t)job.rows8;
    const cf32p nrm = (cf32p)(uintptr_t)im->norms;
    for (uint32_t r = 0; r < n; ++r) {
        const cu32p a = base + (size_t)r * W;
        uint32_t aq = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) aq = __builtin_amdgcn_udot4(a[w], q[w], aq, false);
        const uint32_t aa = (uint32_t)nrm[r];                            // ||a||^2, an integer (staging statistics of the view)
        const float acc = (float)(aa + qq - 2u * aq);
        const unsigned long long key = (r == my) ? ~0ull : key_of(acc, r);
        if (key < worst) {
            uint32_t k = K - 1;
            while (k > 0) {
                const unsigned long long up = list[(k - 1) * 256 + threadIdx.x];
                if (up <= key) break;
                list[k * 256 + threadIdx.x] = up;
                --k;
            }
            list[k * 256 + threadIdx.x] = key;
            worst = list[(K - 1) * 256 + threadIdx.x];
        }
    }
    if (row < n) {
        unsigned long long* out = job.fwd + (size_t)row * K;
        for (uint32_t k = 0; k < K; ++k) out[k] = list[k * 256 + threadIdx.x];
    }
}

// is `id` one of the forward neighbours of `node`?
__device__ __forceinline__ bool ann_has_forward(const unsigned long long* fwd, uint32_t K, uint32_t node, uint32_t id)
{
    const unsigned long long* l = fwd + (size_t)node * K;
    bool hit = false;
    for (uint32_t k = 0; k < K; ++k) hit |= ((uint32_t)l[k] == id) && (l[k] != ~0ull);
    return hit;
}

// step 2 (mode 0: count, mode 1: fill): reverse edge j <- i for every forward edge i -> j that j does not
// already hold as a forward edge itself (that copy would be an exact duplicate: same distance, same id)
__global__ __launch_bounds__(256)
void ann_reverse_kernel(const AnnBuildParams P, int mode)
{
    const AnnBuildJob job = P.jobs[blockIdx.y];
    const uint32_t n = P.imgs[job.slot].n, K = P.K;
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (uint64_t)n * K) return;
    const uint32_t i = (uint32_t)(e / K);
    const unsigned long long key = job.fwd[e];
    if (key == ~0ull) return;
    const uint32_t j = (uint32_t)key;
    if (ann_has_forward(job.fwd, K, j, i)) return;
    if (mode == 0) atomicAdd(job.rev_cnt + j, 1u);
    else {
        const uint32_t pos = atomicAdd(job.rev_cur + j, 1u);
        job.rev[job.rev_off[j] + pos] = (key & 0xFFFFFFFF00000000ull) | i;
    }
}

// exclusive scan of rev_cnt -> rev_off (one workgroup per view; n <= 4M)
__global__ __launch_bounds__(1024)
void ann_scan_kernel(const AnnBuildParams P)
{
    __shared__ uint32_t part[1024];
    const AnnBuildJob job = P.jobs[blockIdx.x];
    const uint32_t n = P.imgs[job.slot].n;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = threadIdx.x * per, e = b + per < n ? b + per : n;
    uint32_t s = 0;
    for (uint32_t k = b; k < e; ++k) s += job.rev_cnt[k];
    part[threadIdx.x] = s;
    r3dm_syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t t = 0; t < 1024; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; } }
    r3dm_syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = b; k < e; ++k) { job.rev_off[k] = run; run += job.rev_cnt[k]; }
    if (threadIdx.x == 1023) job.rev_off[n] = run;
}

// ascending bitonic sort of one 64-bit key per lane across the wavefront
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long key, uint32_t lane)
{
#pragma unroll
    for (uint32_t size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, (int)stride);
            const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)stride);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool up = ((lane & size) == 0) || size == 64;
            const bool lower = ((lane & stride) == 0);
            const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}
// `key` is bitonic across the wave -> ascending
__device__ __forceinline__ unsigned long long wave_merge64(unsigned long long key, uint32_t lane)
{
#pragma unroll
    for (uint32_t stride = 32; stride > 0; stride >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, (int)stride);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)stride);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        const bool lower = ((lane & stride) == 0);
        const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
        key = lower ? mn : mx;
    }
    return key;
}

// step 3: adjacency row of every node = the 64 smallest keys of (forward list + reverse list); one wave per node
__global__ __launch_bounds__(256)
void ann_merge_kernel(const AnnBuildParams P)
{
    const AnnBuildJob job = P.jobs[blockIdx.y];
    const uint32_t n = P.imgs[job.slot].n, K = P.K;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n) return;
    const uint32_t rb = job.rev_off[node], re = job.rev_off[node + 1];
    // first 64: forward keys in lanes [0, K), reverse keys behind them
    unsigned long long cur = ~0ull;
    if (lane < K) cur = job.fwd[(size_t)node * K + lane];
    else if (rb + (lane - K) < re) cur = job.rev[rb + (lane - K)];
    cur = wave_sort64(cur, lane);
    for (uint32_t b = rb + (64 - K); b < re; b += 64) {
        unsigned long long nx = (b + lane < re) ? job.rev[b + lane] : ~0ull;
        nx = wave_sort64(nx, lane);
        // keep the 64 smallest of the two ascending runs: min(cur[i], nx[63 - i]) is bitonic
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)nx, (int)(63 - lane));
        const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(nx >> 32), (int)(63 - lane));
        const unsigned long long rv = ((unsigned long long)hi << 32) | lo;
        cur = wave_merge64(cur < rv ? cur : rv, lane);
    }
    const bool live = (cur != ~0ull);
    job.adj[(size_t)node * kAnnDeg + lane] = live ? (uint32_t)cur : kNone;
    const unsigned long long bal = __ballot(live);
    if (lane == 0) job.deg[node] = (uint32_t)__builtin_popcountll(bal);
}

// ------------------------------------------------------------------------------------------------
// search: one wavefront per query row of J against the index of I
// ------------------------------------------------------------------------------------------------
// ROWS 1 / 2: the rows of I are gathered from their bf16 / u8 copy (ImgDev::ann_rows16 / ann_rows8: every element IS a bf16 /
// a byte, so the f32 values -- and with them every distance, in the same operation order -- are those of the f32 rows at a half /
// a quarter of the bytes per gather; a 128-dimensional u8 row is one 128-byte line).
// ROWS 3: the query view holds the byte copy too.  Then every (a - q)^2 and every partial sum of the reference's float accumulation is
// an integer below 2^24 (D <= 256), i.e. exact in ANY order, and the distance is ||a||^2 - 2 a.q + ||q||^2 from two v_dot4_u32_u8
// per four dimensions instead of ~6 f32 instructions per dimension -- the kernel is bound by VALU issue (86 % busy, 41 VALU
// instructions per evaluation on f32 arithmetic: profiles/r02_r_pmc_ann_search.txt), not by its gathers.
template <int NQ, int ROWS>
__global__ __launch_bounds__(256)                          // (capping the registers at 64 for eight waves per SIMD measured no gain: 1,746 vs 1,780 pairs/s on C5)
void ann_search_kernel(const AnnSearchParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ann_smem[];
    const uint32_t pair = blockIdx.x / P.qb_per_pair;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q = (blockIdx.x % P.qb_per_pair) * 4 + wave;
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nI = Ip->n, nJ = Jp->n, dim = Ip->dim, g4 = dim >> 2;
    if (q >= nJ) return;                                          // whole wave; no workgroup barrier below
    uint32_t* flags = (uint32_t*)ann_smem + (size_t)wave * P.flag_words;
    for (uint32_t w = lane; w < P.flag_words; w += 64) flags[w] = 0u;

    const uint32_t sub = lane & 3u, grp = lane >> 2;
    const uint32_t g_lo = sub * NQ;
    const float* __restrict__ rowsI = Ip->rows;
    const uint16_t* __restrict__ rows16 = Ip->ann_rows16;
    const uint8_t* __restrict__ rows8 = Ip->ann_rows8;
    [[maybe_unused]] const float* __restrict__ normsI = Ip->norms;
    const uint32_t* __restrict__ adj = Ip->ann_adj;
    const uint32_t* __restrict__ deg = Ip->ann_deg;

    f32x4 qv[ROWS == 3 ? 1 : NQ];
    [[maybe_unused]] uint32_t q8[ROWS == 3 ? NQ : 1];
    [[maybe_unused]] uint32_t qq_part = 0;                  // ||q||^2 (summed over the four lanes of the group below)
    if constexpr (ROWS == 3) {
        static_assert(ROWS != 3 || NQ % 4 == 0, "byte rows are read 16 elements at a time");
        const u32x4* src = (const u32x4*)(Jp->ann_rows8 + (size_t)q * dim) + (g_lo >> 2);
#pragma unroll
        for (int g4i = 0; g4i < NQ / 4; ++g4i) {
            const u32x4 w = (g_lo + 4 * g4i < g4) ? src[g4i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) { q8[4 * g4i + k] = w[k]; qq_part = __builtin_amdgcn_udot4(w[k], w[k], qq_part, false); }
        }
        qq_part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qq_part, 0xB1, 0xF, 0xF, true);
        qq_part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qq_part, 0x4E, 0xF, 0xF, true);
    } else {
        const f32x4* src = (const f32x4*)(Jp->rows + (size_t)q * dim);
#pragma unroll
        for (int g = 0; g < NQ; ++g) qv[g] = (g_lo + g < g4) ? src[g_lo + g] : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // pool: one entry per lane, ascending distance, `L` valid entries, at most `cap` = K + P
    float pdist = R3DM_INF; uint32_t pid = kNone, pm = 0, pM = 0; bool pflag = false;
    uint32_t L = 0;
    const uint32_t cap = P.pool_cap, S = P.S;
    const uint2 vid = P.pair_ids[pair];
    uint32_t comps = 0;

    uint32_t e_id = 0, e_m = 0, end_m = 0;
    uint32_t seed_round = 0;
    const uint32_t n_seed_rounds = (P.P + 15) / 16;
    for (;;) {
        // ---- the 16 candidates of this step: start rows first, then neighbours of the best open pool entry
        uint32_t cid = kNone;
        bool valid;
        if (seed_round < n_seed_rounds) {
            const uint32_t s = seed_round * 16 + grp;
            valid = s < P.P;
            if (valid) {
                const uint32_t lo = (uint32_t)(((uint64_t)s * nI) / P.P), hi = (uint32_t)(((uint64_t)(s + 1) * nI) / P.P);
                const uint64_t r = ann_rng_u64(P.seed ^ 0x6b67726170680000ULL, vid.x, vid.y, q, s);
                cid = lo + (uint32_t)(((r >> 32) * (uint64_t)(hi - lo)) >> 32);
            }
            ++seed_round;
        } else {
            const unsigned long long open = __ballot(pflag && lane < L);
            if (open == 0ull) break;
            const uint32_t k = (uint32_t)__builtin_ctzll(open);
            e_id = (uint32_t)__builtin_amdgcn_readlane((int)pid, (int)k);
            e_m = (uint32_t)__builtin_amdgcn_readlane((int)pm, (int)k);
            const uint32_t e_M = (uint32_t)__builtin_amdgcn_readlane((int)pM, (int)k);
            end_m = e_m + S;
            const bool done = end_m > e_M;
            if (done) end_m = e_M;
            if (lane == k) { pm = end_m; if (done) pflag = false; }
            valid = (grp < S) && (e_m + grp < end_m);
            if (valid) cid = adj[(size_t)e_id * kAnnDeg + e_m + grp];
        }
        bool fresh = false;
        if (valid) fresh = ((flags[cid >> 5] >> (cid & 31u)) & 1u) == 0u;
        if (fresh && sub == 0) atomicOr(&flags[cid >> 5], 1u << (cid & 31u));

        // ---- distance of the group's candidate in the reference's summation order
        float r = 0.0f;
        uint32_t cdeg = 0;
        if constexpr (ROWS == 3) {
            if (fresh) {
                const u32x4* a8 = (const u32x4*)(rows8 + (size_t)cid * dim) + (g_lo >> 2);
                const float aa_f = normsI[cid];                                // ||a||^2 from the staging statistics (an integer)
                uint32_t aq = 0;
#pragma unroll
                for (int g4i = 0; g4i < NQ / 4; ++g4i) {
                    if (g_lo + 4 * g4i < g4) {
                        const u32x4 w = a8[g4i];
#pragma unroll
                        for (int k = 0; k < 4; ++k) aq = __builtin_amdgcn_udot4(w[k], q8[4 * g4i + k], aq, false);
                    }
                }
                cdeg = deg[cid];
                aq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)aq, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                aq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)aq, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]: all four lanes hold a.q
                r = (float)((uint32_t)aa_f + qq_part - 2u * aq);                    // sum of (a - q)^2 >= 0, below 2^24
            }
        } else
        if (fresh) {
            float gs[NQ];
            if constexpr (ROWS == 2) {
                static_assert(NQ % 4 == 0, "u8 rows are read 16 elements (four groups) at a time");
                const u32x4* a8 = (const u32x4*)(rows8 + (size_t)cid * dim) + (g_lo >> 2);
#pragma unroll
                for (int g4i = 0; g4i < NQ / 4; ++g4i) {
                    if (g_lo + 4 * g4i < g4) {                                 // dim % 16 == 0: groups come in fours
                        const u32x4 w = a8[g4i];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = {(float)(w[k] & 0xFFu), (float)((w[k] >> 8) & 0xFFu), (float)((w[k] >> 16) & 0xFFu), (float)(w[k] >> 24)};
                            gs[4 * g4i + k] = group_sq(v, qv[4 * g4i + k]);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) gs[4 * g4i + k] = 0.0f;
                    }
                }
            } else if constexpr (ROWS == 1) {
                static_assert(NQ % 2 == 0, "bf16 rows are read 8 elements (two groups) at a time");
                const u32x4* a16 = (const u32x4*)(rows16 + (size_t)cid * dim) + (g_lo >> 1);
#pragma unroll
                for (int g2 = 0; g2 < NQ / 2; ++g2) {
                    if (g_lo + 2 * g2 < g4) {                                  // dim % 8 == 0: groups come in pairs
                        const u32x4 w = a16[g2];
                        const f32x4 lo4 = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xFFFF0000u)};
                        const f32x4 hi4 = {__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xFFFF0000u), __uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xFFFF0000u)};
                        gs[2 * g2] = group_sq(lo4, qv[2 * g2]); gs[2 * g2 + 1] = group_sq(hi4, qv[2 * g2 + 1]);
                    } else { gs[2 * g2] = 0.0f; gs[2 * g2 + 1] = 0.0f; }
                }
            } else {
                const f32x4* a = (const f32x4*)(rowsI + (size_t)cid * dim) + g_lo;
#pragma unroll
                for (int g = 0; g < NQ; ++g) gs[g] = (g_lo + g < g4) ? group_sq(a[g], qv[g]) : 0.0f;
            }
            cdeg = deg[cid];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float t = r;
#pragma unroll
                for (int g = 0; g < NQ; ++g) t = ((uint32_t)(s * NQ + g) < g4) ? t + gs[g] : t;
                r = (sub == (uint32_t)s) ? t : r;
                const float prev = __shfl_up(r, 1);
                if (sub == (uint32_t)s + 1u) r = prev;
            }
        }
        // ---- sequential sorted inserts, candidate order = adjacency order (UpdateKnnList semantics)
        // A full pool rejects every candidate at or beyond its last entry, and that entry only moves down while the step's
        // candidates are inserted: those are dropped here, before the one-at-a-time loop (most of them, once the pool has settled)
        const unsigned long long evaluated = __ballot(fresh && sub == 3u);
        comps += (uint32_t)__builtin_popcountll(evaluated);
        const float worst = (L >= cap) ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(pdist), (int)(cap - 1u))) : R3DM_INF;
        unsigned long long todo = __ballot(fresh && sub == 3u && (r < worst || L < cap));
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const float cd = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r), src));
            const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cid, src);
            const uint32_t cM = (uint32_t)__builtin_amdgcn_readlane((int)cdeg, src);
            const uint32_t rk = (uint32_t)__builtin_popcountll(__ballot(lane < L && pdist <= cd));
            if (rk >= cap) continue;
            const float sd = __shfl_up(pdist, 1);
            const uint32_t si = (uint32_t)__shfl_up((int)pid, 1);
            const uint32_t sm = (uint32_t)__shfl_up((int)pm, 1);
            const uint32_t sM = (uint32_t)__shfl_up((int)pM, 1);
            const int sf = __shfl_up((int)pflag, 1);
            if (lane > rk) { pdist = sd; pid = si; pm = sm; pM = sM; pflag = sf != 0; }
            if (lane == rk) { pdist = cd; pid = ci; pm = 0; pM = cM; pflag = true; }
            if (L < cap) ++L;
        }
    }

    // ---- results: the two best pool entries; distance-ratio test (squared metric: R = ratio^2)
    const float d0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(pdist), 0));
    const float d1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(pdist), 1));
    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)pid, 0);
    const uint32_t i1 = (uint32_t)__builtin_amdgcn_readlane((int)pid, 1);
    if (lane == 0) {
        const size_t o = (size_t)pair * P.q_stride + q;
        const bool two = L >= 2;
        P.nn_idx[o] = (two && d0 < P.ratio_R * d1) ? i0 : kNone;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = L >= 1 ? (int32_t)i0 : -1; P.knn_idx[2 * o + 1] = two ? (int32_t)i1 : -1;
            P.knn_dist[2 * o] = L >= 1 ? d0 : R3DM_INF;   P.knn_dist[2 * o + 1] = two ? d1 : R3DM_INF;
        }
        atomicAdd(P.n_comps, (unsigned long long)comps);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
hipError_t launch_ann_build(hipStream_t st, const AnnBuildParams& P, uint32_t n_jobs, uint32_t max_n, uint32_t dim, bool rows8)
{
    if (n_jobs == 0 || max_n == 0) return hipSuccess;
    const uint32_t K = P.K;
    if ((dim & 3u) || K < 1 || K > kAnnMaxK) return hipErrorInvalidValue;
    const dim3 g256((max_n + 255) / 256, n_jobs);
    const size_t l256 = (size_t)K * 256 * 8;
    if (rows8 && dim == 128) hipLaunchKernelGGL((ann_knn_rows8_kernel<32>), g256, dim3(256), l256, st, P);
    else if (rows8 && dim == 64) hipLaunchKernelGGL((ann_knn_rows8_kernel<16>), g256, dim3(256), l256, st, P);
    else if (rows8 && dim == 256) hipLaunchKernelGGL((ann_knn_rows8_kernel<64>), g256, dim3(256), l256, st, P);
    else if (rows8 && dim == 96) hipLaunchKernelGGL((ann_knn_rows8_kernel<24>), g256, dim3(256), l256, st, P);
    else if (rows8 && dim == 48) hipLaunchKernelGGL((ann_knn_rows8_kernel<12>), g256, dim3(256), l256, st, P);
    else if (dim == 128) {
        hipLaunchKernelGGL((ann_knn_rows_kernel<32, 256>), dim3((max_n + 255) / 256, n_jobs), dim3(256), (size_t)K * 256 * 8, st, P);
    } else if (dim == 144) {
        hipLaunchKernelGGL((ann_knn_rows_kernel<36, 256>), dim3((max_n + 255) / 256, n_jobs), dim3(256), (size_t)K * 256 * 8, st, P);
    } else {
        const size_t lds = (size_t)K * 64 * 8 + (size_t)dim * 64 * 4;
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute((const void*)ann_knn_rows_kernel<0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ann_knn_rows_kernel<0, 64>), dim3((max_n + 63) / 64, n_jobs), dim3(64), lds, st, P);
    }
    const uint32_t eb = (uint32_t)(((uint64_t)max_n * K + 255) / 256);
    hipLaunchKernelGGL(ann_reverse_kernel, dim3(eb, n_jobs), dim3(256), 0, st, P, 0);
    hipLaunchKernelGGL(ann_scan_kernel, dim3(n_jobs), dim3(1024), 0, st, P);
    hipLaunchKernelGGL(ann_reverse_kernel, dim3(eb, n_jobs), dim3(256), 0, st, P, 1);
    hipLaunchKernelGGL(ann_merge_kernel, dim3((max_n + 3) / 4, n_jobs), dim3(256), 0, st, P);
    return hipGetLastError();
}

__global__ __launch_bounds__(256)
void ann_rows16_kernel(const float* __restrict__ rows, uint16_t* __restrict__ rows16, size_t n_elems)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * 256)
        rows16[i] = (uint16_t)(__float_as_uint(rows[i]) >> 16);                // exact: the caller checked that every element is a bf16
}

hipError_t launch_ann_rows16(hipStream_t st, const float* rows, uint16_t* rows16, size_t n_elems)
{
    if (n_elems == 0) return hipSuccess;
    const size_t blocks = std::min<size_t>((n_elems + 255) / 256, 8192);
    hipLaunchKernelGGL(ann_rows16_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, rows, rows16, n_elems);
    return hipGetLastError();
}

__global__ __launch_bounds__(256)
void ann_rows8_kernel(const float* __restrict__ rows, uint8_t* __restrict__ rows8, size_t n_elems)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * 256)
        rows8[i] = (uint8_t)rows[i];                                           // exact: the caller checked that every element is an integer in 0 .. 255
}

hipError_t launch_ann_rows8(hipStream_t st, const float* rows, uint8_t* rows8, size_t n_elems)
{
    if (n_elems == 0) return hipSuccess;
    const size_t blocks = std::min<size_t>((n_elems + 255) / 256, 8192);
    hipLaunchKernelGGL(ann_rows8_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, rows, rows8, n_elems);
    return hipGetLastError();
}

hipError_t launch_ann_search(hipStream_t st, const AnnSearchParams& Pin, uint32_t max_nJ, uint32_t max_nI, uint32_t dim, int rows_mode)
{
    AnnSearchParams P = Pin;
    if ((dim & 3u) || P.pool_cap > 63 || P.S < 1 || P.S > 16 || P.P < 2) return hipErrorInvalidValue;
    P.qb_per_pair = (max_nJ + 3) / 4;
    P.flag_words = (max_nI + 31) / 32;
    const size_t lds = (size_t)P.flag_words * 4 * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const uint64_t grid = (uint64_t)P.n_pairs * P.qb_per_pair;
    if (grid == 0) return hipSuccess;
    if (grid > kMaxBlocksOf256) return hipErrorInvalidValue;
    const uint32_t nq = (dim / 4 + 3) / 4;
#define R3DM_ANN_LAUNCH(NQ, ROWS)                                                                                      \
    do {                                                                                                               \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)ann_search_kernel<NQ, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((ann_search_kernel<NQ, ROWS>), dim3((uint32_t)grid), dim3(256), lds, st, P);                 \
    } while (0)
    if (rows_mode == 3 && (dim & 15u) == 0 && dim <= 256 && nq != 9) {
        if (nq <= 4) R3DM_ANN_LAUNCH(4, 3);
        else if (nq <= 8) R3DM_ANN_LAUNCH(8, 3);
        else if (nq <= 16) R3DM_ANN_LAUNCH(16, 3);
        else return hipErrorInvalidValue;
    }
    else if (rows_mode >= 2 && (dim & 15u) == 0 && nq != 9) {
        if (nq <= 4) R3DM_ANN_LAUNCH(4, 2);
        else if (nq <= 8) R3DM_ANN_LAUNCH(8, 2);
        else if (nq <= 16) R3DM_ANN_LAUNCH(16, 2);
        else if (nq <= 32) R3DM_ANN_LAUNCH(32, 2);
        else return hipErrorInvalidValue;
    }
    else if (rows_mode == 1 && (dim & 7u) == 0 && nq != 9) {
        if (nq <= 4) R3DM_ANN_LAUNCH(4, 1);
        else if (nq <= 8) R3DM_ANN_LAUNCH(8, 1);
        else if (nq <= 16) R3DM_ANN_LAUNCH(16, 1);
        else if (nq <= 32) R3DM_ANN_LAUNCH(32, 1);
        else return hipErrorInvalidValue;
    }
    else if (rows_mode != 0) return hipErrorInvalidValue;
    else if (nq <= 4) R3DM_ANN_LAUNCH(4, 0);
    else if (nq <= 8) R3DM_ANN_LAUNCH(8, 0);
    else if (nq == 9) R3DM_ANN_LAUNCH(9, 0);
    else if (nq <= 16) R3DM_ANN_LAUNCH(16, 0);
    else if (nq <= 32) R3DM_ANN_LAUNCH(32, 0);
    else return hipErrorInvalidValue;
#undef R3DM_ANN_LAUNCH
    return hipGetLastError();
}

}  // namespace r3dm
