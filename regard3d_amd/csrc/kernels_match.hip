// kernels_match.hip -- putative matching on gfx950 (MI355X): staging of a view and the fused squared-L2 2-NN on FP32 MFMA tiles
// (the headline kernel).  Siblings: kernels_match_16bit.hip (bf16 / f16 nominators), kernels_match_hamming.hip (binary descriptors),
// kernels_match_exact.hip (exact scans, per-pair finalisation); shared device helpers in kernels_match_common.hpp.
//
// What it replaces in the reference (rhiestan/Regard3D, /root/reference):
//   Matcher_Regions(fDistRatio, BRUTE_FORCE_L2).Match()    src/R3DComputeMatches.cpp:2037-2039,2048
//   per-I / OpenMP-over-J loop nest                          src/R3DComputeMatches.cpp:437-489
//   ArrayMatcher::SearchNeighbours(NN=2) + MatchDistanceRatio  src/utils/matcher_kgraph.h:205-251,
//                                                            src/R3DComputeMatches.cpp:479
// Arithmetic contract (OpenMVG L2<float>): distances are the f32
// 4-way-unrolled sum of squared differences, NO fused multiply-add; equal distances -> lowest
// dataset row.  The MFMA pass only nominates candidates (||a||^2 - 2 a.b, any summation order); the
// two best candidates are re-scored with the reference arithmetic, and a query whose runner-up is
// not provably separated from every un-nominated row is redone by an exact scan (kFallback).
//
// This file is compiled with -ffp-contract=off; fused operations are spelled fmaf()/MFMA.


#include "kernels_match_common.hpp"

namespace r3dm {

// ------------------------------------------------------------------------------------------------
// staging (registration of a view: what Regions_Provider::load is to the reference, /root/reference/src/R3DComputeMatches.cpp:2040,2094)
// ONE kernel per view on the default path: the raw row-major descriptors (f32, or u8 converted on the way) are read ONCE -- from the
// upload ring's device slot, from the caller's device buffer, or straight from page-locked host memory over the link -- and leave as
// the MFMA fragment-order tiles + norms + the view's statistics; the row-major f32 copy (`rows`) is written only when asked for
// (a view of integer-valued bins never needs it: its MFMA keys ARE the reference distances, and the exact scan reads the tiles).
// Every other layout (bf16 tiles, split-f16 planes, count tiles, byte tiles) is staged on first use by the path that reads it
// (api_core.cpp: ensure_layouts).
// ------------------------------------------------------------------------------------------------

// role blocks behind the n_tiles tile blocks -- one per 256 features, at least 8: they copy the positions, enter them into the
// position-class hash table (IndMatchDecorator's coordinate de-duplication needs to know which features share a position; one
// compare-and-swap + one minimum per feature, a thread each) and zero the slack
__host__ __device__ inline uint32_t stage_aux_blocks(uint32_t n) { const uint32_t b = (n + 255u) / 256u; return b < 8u ? 8u : (b > 1024u ? 1024u : b); }

__device__ __forceinline__ bool canon_key_of(float fx, float fy, unsigned long long& key)
{
    if (fx != fx || fy != fy) return false;                          // NaN: equals nothing, itself included -> a class of its own
    const float zx = fx == 0.0f ? 0.0f : fx, zy = fy == 0.0f ? 0.0f : fy;      // -0 folded into +0: equal floats <-> equal bit patterns
    key = ((unsigned long long)__float_as_uint(zx) << 32) | __float_as_uint(zy);
    return true;
}
__device__ __forceinline__ uint32_t canon_hash(unsigned long long key, uint32_t bits)
{
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

__device__ __forceinline__ void stage_aux_role(const StageViewArgs& A, uint32_t rb, uint32_t n_role_blocks)
{
    const uint32_t n = A.n, G = A.G;
    const uint32_t nthr = n_role_blocks * 256u, tid = rb * 256u + threadIdx.x;
    if (A.xy_src) {
        const unsigned long long kEmpty = ~0ull;
        const uint32_t mask = (1u << A.canon_bits) - 1u;
        for (uint32_t k = tid; k < n; k += nthr) {
            const float fx = A.xy_src[2 * (size_t)k], fy = A.xy_src[2 * (size_t)k + 1];
            if (A.xy_dst != A.xy_src) { A.xy_dst[2 * (size_t)k] = fx; A.xy_dst[2 * (size_t)k + 1] = fy; }
            unsigned long long key;
            if (A.canon_keys && canon_key_of(fx, fy, key)) {
                uint32_t slot = canon_hash(key, A.canon_bits) & mask;
                for (;;) {
                    const unsigned long long old = atomicCAS(A.canon_keys + slot, kEmpty, key);
                    if (old == kEmpty || old == key) { atomicMin(A.canon_vals + slot, k); break; }
                    slot = (slot + 1u) & mask;
                }
            }
        }
    }
    // zero slack behind the tiles and the norms (prefetches run past the end; a recycled buffer holds an older view there)
    if (A.tiled) {
        float* ts = A.tiled + (size_t)A.n_tiles * G * 256u;
        for (uint32_t e = tid; e < kSlackBytes / 4u; e += nthr) ts[e] = 0.0f;
        float* ns = A.norms + (size_t)A.n_tiles * 32u;
        for (uint32_t e = tid; e < kSlackBytes / 4u; e += nthr) ns[e] = 0.0f;
    }
}

// one workgroup per 32-row tile (+ the role blocks)
__global__ __launch_bounds__(256)
void stage_view_kernel(const StageViewArgs A)
{
    __shared__ float sm[32 * 260];                     // the tile's rows, row stride dim + 4 floats (dim <= 256 staged through LDS)
    const uint32_t n = A.n, dim = A.dim, G = A.G;
    // (the role blocks come first: their atomics' round trips run beside the tile blocks' copies)
    const uint32_t n_role = gridDim.x - A.n_tiles;
    if (blockIdx.x < n_role) { stage_aux_role(A, blockIdx.x, n_role); return; }
    const uint32_t t = blockIdx.x - n_role;
    const uint32_t rows_here = (n - t * 32u < 32u) ? n - t * 32u : 32u;
    const uint32_t per_tile = G * 256u;                // floats per tile = G * 2 * 32 * 4
    float* dst = A.tiled + (size_t)t * per_tile;
    const bool lds = dim <= 256u && (dim & 3u) == 0u && (((uintptr_t)A.raw) & (A.raw_is_u8 ? 3u : 15u)) == 0u;
    const uint32_t stride = dim + 4u;
    if (lds) {
        // coalesced read of the tile's rows_here x dim values (contiguous in the raw image), converted, into LDS: thread (ty, tx) walks
        // rows ty, ty + 8, ... and the 4-element chunks tx, tx + 32, ... of a row
        const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5, d4 = dim >> 2;
        if (A.raw_is_u8) {
            const uint32_t* src = (const uint32_t*)((const uint8_t*)A.raw + (size_t)t * 32u * dim);     // dim % 4 == 0: 4-byte aligned
            for (uint32_t r = ty; r < rows_here; r += 8u)
                for (uint32_t k4 = tx; k4 < d4; k4 += 32u) {
                    const uint32_t w = src[r * d4 + k4];
                    *(f32x4*)(sm + r * stride + 4u * k4) = f32x4{(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
                }
        } else {
            const f32x4* src = (const f32x4*)((const float*)A.raw + (size_t)t * 32u * dim);
            for (uint32_t r = ty; r < rows_here; r += 8u)
                for (uint32_t k4 = tx; k4 < d4; k4 += 32u) *(f32x4*)(sm + r * stride + 4u * k4) = src[r * d4 + k4];
        }
        __syncthreads();
        if (A.rows) {
            f32x4* ro = (f32x4*)(A.rows + (size_t)t * 32u * dim);
            for (uint32_t r = ty; r < rows_here; r += 8u)
                for (uint32_t k4 = tx; k4 < d4; k4 += 32u) ro[r * d4 + k4] = *(const f32x4*)(sm + r * stride + 4u * k4);
        }
        // fragment order: float4 (g, h, r) = row 32 t + r, dims 8 g + 4 h .. + 3
        for (uint32_t e4 = threadIdx.x; e4 < per_tile / 4u; e4 += 256u) {
            const uint32_t r = e4 & 31u, k = (e4 >> 5) * 4u;            // (g, h) = e4 >> 5: dims 8 g + 4 h = 4 (e4 >> 5)
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < rows_here && k < dim) v = *(const f32x4*)(sm + r * stride + k);
            ((f32x4*)dst)[e4] = v;
        }
    } else {
        // any other descriptor length: element by element from the raw image
        for (uint32_t e = threadIdx.x; e < per_tile; e += 256u) {
            const uint32_t c = e & 3u, r = (e >> 2) & 31u, h = (e >> 7) & 1u, g = e >> 8;
            const uint32_t row = t * 32u + r, k = 8u * g + 4u * h + c;
            float v = 0.0f;
            if (row < n && k < dim) v = A.raw_is_u8 ? (float)((const uint8_t*)A.raw)[(size_t)row * dim + k] : ((const float*)A.raw)[(size_t)row * dim + k];
            dst[e] = v;
            if (A.rows && row < n && k < dim) A.rows[(size_t)row * dim + k] = v;
        }
    }
    if (threadIdx.x < 32u) {
        const uint32_t row = t * 32u + threadIdx.x;
        float s = R3DM_INF;
        if (row < n) {
            s = 0.0f;
            float mx = 0.0f; bool nonint = false, neg = false;
            // ||row||^2 as ONE fma chain in element order (the accumulators' C operand; what rounds 1-5 computed), four elements per LDS read
            auto take = [&](float v) {
                s = fmaf(v, v, s);
                mx = fmaxf(mx, fabsf(v));
                nonint |= !(v == rintf(v));                  // also true for NaN
                neg |= v < 0.0f;
            };
            if (lds) {
                const float* p = sm + threadIdx.x * stride;
                for (uint32_t k = 0; k < dim; k += 4u) { const f32x4 v = *(const f32x4*)(p + k); take(v[0]); take(v[1]); take(v[2]); take(v[3]); }
            } else {
                for (uint32_t k = 0; k < dim; ++k)
                    take(A.raw_is_u8 ? (float)((const uint8_t*)A.raw)[(size_t)row * dim + k] : ((const float*)A.raw)[(size_t)row * dim + k]);
            }
            // img_stats = &ImgDev::max_norm_bits, max_abs_bits, not_integer (non-negative floats order like uints)
            atomicMax(A.img_stats + 0, __float_as_uint(s));
            atomicMax(A.img_stats + 1, __float_as_uint(mx));
            if (nonint) atomicOr(A.img_stats + 2, 1u);     // ImgDev::not_integer: bit 0 = some non-integer, bit 1 = some negative
            if (neg) atomicOr(A.img_stats + 2, 2u);
        }
        A.norms[(size_t)t * 32u + threadIdx.x] = s;
    }
}

// position classes, second pass (behind stage_view_kernel's inserts): canon[k] = the smallest feature index at k's position;
// *has_dup is set when some feature is not the first of its class
__global__ __launch_bounds__(256)
void canon_lookup_kernel(const float* __restrict__ xy, uint32_t n, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                         uint32_t bits, uint32_t* __restrict__ canon, uint32_t* __restrict__ has_dup)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    uint32_t cls = k;
    unsigned long long key;
    if (canon_key_of(xy[2 * (size_t)k], xy[2 * (size_t)k + 1], key)) {
        const uint32_t mask = (1u << bits) - 1u;
        uint32_t slot = canon_hash(key, bits) & mask;
        while (keys[slot] != key) slot = (slot + 1u) & mask;          // the key was entered by the pass before
        cls = vals[slot];
    }
    canon[k] = cls;
    if (cls != k) atomicOr(has_dup, 1u);
}

__global__ __launch_bounds__(256)
void stage_positions_kernel(const StageViewArgs A) { stage_aux_role(A, blockIdx.x, gridDim.x); }

hipError_t launch_stage_positions(hipStream_t st, const StageViewArgs& A)
{
    if (!A.xy_src || A.n == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_positions_kernel, dim3(stage_aux_blocks(A.n)), dim3(256), 0, st, A);
    if (A.canon_keys)
        hipLaunchKernelGGL(canon_lookup_kernel, dim3((A.n + 255u) / 256u), dim3(256), 0, st, A.xy_dst, A.n, A.canon_keys, A.canon_vals, A.canon_bits,
                           A.canon_dst, A.has_dup);
    return hipGetLastError();
}

hipError_t launch_stage_view(hipStream_t st, const StageViewArgs& A)
{
    hipLaunchKernelGGL(stage_view_kernel, dim3(A.n_tiles + stage_aux_blocks(A.n)), dim3(256), 0, st, A);
    if (A.xy_src && A.canon_keys && A.n)
        hipLaunchKernelGGL(canon_lookup_kernel, dim3((A.n + 255u) / 256u), dim3(256), 0, st, A.xy_dst, A.n, A.canon_keys, A.canon_vals, A.canon_bits,
                           A.canon_dst, A.has_dup);
    return hipGetLastError();
}

// ---- layouts staged on first use, from the fragment-order tiles (which hold the view's f32 values verbatim)
// row-major f32 rows: float4 chunk k4 of row q is tiled float4 ((q >> 5) 2G + k4) 32 + (q & 31)
__global__ __launch_bounds__(256)
void untile_rows_kernel(const float* __restrict__ tiled, uint32_t n, uint32_t dim, uint32_t G, float* __restrict__ rows)
{
    const uint32_t t = blockIdx.x;
    const float* src = tiled + (size_t)t * G * 256u;
    for (uint32_t e = threadIdx.x; e < G * 256u; e += 256u) {
        const uint32_t c = e & 3u, r = (e >> 2) & 31u, h = (e >> 7) & 1u, g = e >> 8;
        const uint32_t row = t * 32u + r, k = 8u * g + 4u * h + c;
        if (row < n && k < dim) rows[(size_t)row * dim + k] = src[e];
    }
}
hipError_t launch_untile_rows(hipStream_t st, const float* tiled, uint32_t n, uint32_t dim, uint32_t G, uint32_t n_tiles, float* rows)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(untile_rows_kernel, dim3(n_tiles), dim3(256), 0, st, tiled, n, dim, G, rows);
    return hipGetLastError();
}

// bf16 tiles of the integer fast path: the upper 16 bits of the float ARE the value when it is an integer of magnitude <= 256 (8
// significant bits); for any other view these tiles are never read (kernel-side check).  [tile][16-dim block][lane half][32 rows][8]
__global__ __launch_bounds__(256)
void stage_bf16_kernel(const float* __restrict__ tiled, uint32_t G, uint16_t* __restrict__ tiled16)
{
    const uint32_t t = blockIdx.x, GB = (G + 1u) / 2u;
    const float* src = tiled + (size_t)t * G * 256u;
    uint16_t* dst16 = tiled16 + (size_t)t * GB * 512u;
    for (uint32_t e = threadIdx.x; e < GB * 512u; e += 256u) {
        const uint32_t c8 = e & 7u, r = (e >> 3) & 31u, h = (e >> 8) & 1u, kb = e >> 9;
        const uint32_t k = 16u * kb + 8u * h + c8;                   // dimension; in the f32 tiles: g = k / 8, half (k / 4) & 1, lane k & 3
        const uint32_t g = k >> 3;
        const float v = g < G ? src[g * 256u + ((k >> 2) & 1u) * 128u + r * 4u + (k & 3u)] : 0.0f;
        dst16[e] = (uint16_t)(__float_as_uint(v) >> 16);
    }
}
hipError_t launch_stage_bf16(hipStream_t st, const float* tiled, uint32_t G, uint32_t n_tiles, uint16_t* tiled16)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_bf16_kernel, dim3(n_tiles), dim3(256), 0, st, tiled, G, tiled16);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fused squared-L2 2-NN: one workgroup = 4 waves; each wave keeps NJ query tiles (32 queries each,
// pre-scaled by -2) in registers as MFMA B fragments and streams every 32-row dataset tile of
// image I as the A fragment, straight from the fragment-ordered HBM/L2 image (1 KiB coalesced
// line per load, rolling PF-deep register window).  D = C + A*B with C initialised to ||a||^2
// gives key = ||a||^2 - 2 a.b per (row, query) in the accumulator; lane (h, c) then owns query
// column c and the 16 rows {e + 8*qd + 4h} of the tile.
//   v_mfma_f32_32x32x2_f32: A lane l = A[i = l&31][k = l>>5], B lane l = B[k = l>>5][j = l&31],
//   D lane l reg r = D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
// The k axis is permuted identically on both operands (lane half h supplies dims 8g+4h+cc at
// step 4g+cc), which a dot product does not notice.
// ------------------------------------------------------------------------------------------------
// One dataset tile: MFMAs of tile t into `cur`, while the VALU folds the finished accumulators of
// tile t-1 (`prev`) into the running top-2 lists -- software pipelining inside the wave, so the
// epilogue issues in the shadow of the 64-cycle MFMAs instead of after them.
template <int G, int NJ, int PF, int PIPE>
__device__ __forceinline__ void l2_tile_step(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rn, uint32_t voffA, uint32_t voffN,
                                             uint32_t soffA, uint32_t soffN, f32x4 (&abuf)[PF], f32x4 (&nrm)[4],
                                             const f32x4 (&bq)[NJ][G], f32x16 (&cur)[NJ], const f32x16 (&prev)[NJ],
                                             Top2 (&st)[NJ], uint32_t prev_rowbase)
{
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[nj][r] = nrm[r >> 2][r & 3];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 a = abuf[g % PF];
        abuf[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 1024u);
        if (g == 2) {   // next tile's norms: early, so the wait at the tile boundary finds them landed
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) nrm[qd] = bload16(rn, voffN, soffN + (uint32_t)qd * 32u);
        }
        if constexpr (PIPE == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
                cur[nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cc], bq[nj][g][cc], cur[nj], 0, 0, 0);
        if constexpr (PIPE == 4) __builtin_amdgcn_s_setprio(0);
        // this group's share of the previous tile's 16 accumulator values per query tile
        if constexpr (PIPE == 9) {
            // ablation (timing only, results meaningless): keep the accumulators alive, skip the epilogue
#pragma unroll
            for (int r = (g * 16) / G; r < ((g + 1) * 16) / G; ++r)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) asm volatile("" ::"v"(prev[nj][r]));
        } else if constexpr (PIPE == 3) {
            // test-and-skip: a value can only change a list if it is below that lane's bound d2; once the
            // lists have warmed up that is rare, so one wave-wide test guards the whole slice
            bool any = false;
#pragma unroll
            for (int r = (g * 16) / G; r < ((g + 1) * 16) / G; ++r)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) any |= prev[nj][r] < st[nj].d2;
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
                for (int r = (g * 16) / G; r < ((g + 1) * 16) / G; ++r)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        top2_push(st[nj], prev[nj][r], prev_rowbase + (uint32_t)((r & 3) + 8 * (r >> 2)));
            }
        } else {
#pragma unroll
            for (int r = (g * 16) / G; r < ((g + 1) * 16) / G; ++r)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
                    top2_push(st[nj], prev[nj][r], prev_rowbase + (uint32_t)((r & 3) + 8 * (r >> 2)));
        }
        if constexpr (PIPE == 2) {
            // issue order inside the step: MFMA, 3 VALU, MFMA, 3 VALU, ... so the epilogue slice hides
            // behind the 64-cycle matrix instructions instead of in front of them
#pragma unroll
            for (int i = 0; i < 4 * NJ; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);                // keep each prefetch / epilogue slice in its own step
    }
}

template <int G, int NJ, int PF, int PIPE, int WPS>
__global__ __launch_bounds__(256, WPS)
void l2_knn2_mfma_kernel(const MatchParams P)
{
    static_assert(G % PF == 0, "prefetch window must divide the group count");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
    // workgroup -> (pair, query block).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2; with
    // xcd_map every workgroup of a pair runs on ONE XCD (pair p on XCD p % 8), so image I and the query tiles are pulled
    // into one L2 instead of eight (pairs are sorted by I: an XCD's consecutive pairs mostly share their dataset image).
    uint32_t pair, qb;
    if (P.xcd_map) {
        const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        pair = (j / P.qb_per_pair) * 8u + xcd;
        qb = j % P.qb_per_pair;
        if (pair >= P.n_pairs) return;
    } else {
        pair = blockIdx.x / P.qb_per_pair;
        qb = blockIdx.x % P.qb_per_pair;
    }
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nI = Ip->n, ntI = Ip->n_tiles, ntJ = Jp->n_tiles;
    const uint32_t qt0 = (qb * 4u + wave) * NJ;
    if (qt0 >= ntJ) return;                           // wave-uniform; no barriers in this kernel

    // ---- query fragments (B operand), scaled by -2
    f32x4 bq[NJ][G];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;          // clamp: results discarded below
        const gf4p src = (gf4p)Jp->tiled + (size_t)qt * (G * 64) + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) bq[nj][g] = src[g * 64] * -2.0f;
    }

    Top2 st[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) top2_init(st[nj]);

    if (nI >= 2) {
        // ---- dataset stream (A operand): float4 index = (t*G + g)*64 + lane.  abase/nbase are
        // wave-uniform (SGPR) bases; only `lane` / `h` are per-lane.
        const gf4p abase = (gf4p)Ip->tiled;
        const gf4p nbase = (gf4p)Ip->norms;               // tile t, quad qd -> float4 index t*8 + 2*qd + h
        f32x4 abuf[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) abuf[s] = abase[s * 64 + lane];
        f32x4 nrm[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) nrm[qd] = nbase[2 * qd + h];

        if constexpr (PIPE != 0) {
            f32x16 accA[NJ], accB[NJ];
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) accB[nj][r] = R3DM_INF;          // "tile -1": keys that never win
            // descriptors from wave-uniform values only (readfirstlane) so no waterfall loop is emitted
            const uint64_t pa = (uint64_t)Ip->tiled, pn = (uint64_t)Ip->norms;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa)),
                0, 0x7FFFFFFF, 0x00020000);
            const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pn >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pn)),
                0, 0x7FFFFFFF, 0x00020000);
            const uint32_t voffA = lane * 16u, voffN = h * 16u;
            const uint32_t tileB = (uint32_t)G * 1024u;                // bytes per tile
            const uint32_t hb = 4u * h;
            uint32_t t = 0;
            for (; t + 1 < ntI; t += 2) {
                l2_tile_step<G, NJ, PF, PIPE>(ra, rn, voffA, voffN, t * tileB + PF * 1024u, (t + 1) * 128u, abuf, nrm, bq, accA, accB, st, (t - 1) * 32u + hb);
                l2_tile_step<G, NJ, PF, PIPE>(ra, rn, voffA, voffN, (t + 1) * tileB + PF * 1024u, (t + 2) * 128u, abuf, nrm, bq, accB, accA, st, t * 32u + hb);
            }
            if (t < ntI) {
                l2_tile_step<G, NJ, PF, PIPE>(ra, rn, voffA, voffN, t * tileB + PF * 1024u, (t + 1) * 128u, abuf, nrm, bq, accA, accB, st, (t - 1) * 32u + hb);
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) top2_push(st[nj], accA[nj][r], t * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
            } else {
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) top2_push(st[nj], accB[nj][r], (ntI - 1) * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
            }
        } else {
        for (uint32_t t = 0; t < ntI; ++t) {
            f32x16 acc[NJ];
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nj][r] = nrm[r >> 2][r & 3];
            const gf4p atile = abase + (size_t)(t * G + PF) * 64;    // PF groups ahead (slack-padded)
            const gf4p ntile = nbase + (size_t)(t + 1) * 8;          // next tile's norms (slack-padded)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 a = abuf[g % PF];
                abuf[g % PF] = atile[g * 64 + lane];
                if (g == 2) {   // next tile's norms: early, so the wait at the tile boundary finds them landed
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) nrm[qd] = ntile[2 * qd + h];
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        acc[nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cc], bq[nj][g][cc], acc[nj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);        // keep each prefetch in its own step
            }
            const uint32_t rowbase = t * 32u + 4u * h;
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    top2_push(st[nj], acc[nj][r], rowbase + (uint32_t)((r & 3) + 8 * (r >> 2)));
        }
        }   // !PIPE
    }

    l2_finish_queries<NJ>(P, pair, Ip, Jp, st, qt0, h, c, (float)(G * 8), false);
}

template <int G, int NJ, int PF, int PIPE, int WPS>
static hipError_t launch_l2_t(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;                 // 4 waves x NJ query tiles x 32 queries
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    static const int xcd_map = r3dm_dev_knob("R3DM_XCD_MAP", 1);
    P.xcd_map = (uint32_t)xcd_map;
    const uint64_t grid64 = (uint64_t)(xcd_map ? (P.n_pairs + 7u) / 8u * 8u : P.n_pairs) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)grid64;
    hipLaunchKernelGGL((l2_knn2_mfma_kernel<G, NJ, PF, PIPE, WPS>), dim3(grid), dim3(256), 0, st, P);
    return hipGetLastError();
}

// The A/B variants below (tools/ab_l2.py) -- including two ablations whose results are meaningless (timing only) -- exist
// only in the developer build (-DR3DM_DEVTOOLS -> regard3d_amd/libr3dm_dev.so, build.sh dev); the product library compiles
// one kernel per descriptor length and never reads the environment.
hipError_t launch_l2_knn2(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles, bool integer_mfma)
{
    if (integer_mfma) {
        const hipError_t e = launch_l2_knn2_int(st, P, G, max_nj_tiles);      // kernels_match_16bit.hip; hipErrorNotSupported: no bf16 kernel for this G
        if (e != hipErrorNotSupported) return e;                              // (G = 18, LIOP, never integer: f32 tiles)
    }
#ifdef R3DM_DEVTOOLS
    // R3DM_L2_VARIANT selects a build of the kernel for A/B measurements (tools/ab_l2.py):
    //   0 epilogue after the MFMAs | 1 software-pipelined epilogue | 3 pipelined + wave-wide test-and-skip
    //   (default) | 9 ablation without epilogue (timing only) | 13 / 43: NJ = 1 x 3 waves/SIMD, NJ = 4 x 1 wave/SIMD
    static const int variant = r3dm_dev_knob("R3DM_L2_VARIANT", 3);
    if (G == 16) {
        switch (variant) {
            case 0:  return launch_l2_t<16, 2, 4, 0, 2>(st, P, max_nj_tiles);
            case 1:  return launch_l2_t<16, 2, 4, 1, 2>(st, P, max_nj_tiles);
            case 9:  return launch_l2_t<16, 2, 4, 9, 2>(st, P, max_nj_tiles);
            case 13: return launch_l2_t<16, 1, 4, 3, 3>(st, P, max_nj_tiles);
            case 43: return launch_l2_t<16, 4, 4, 3, 1>(st, P, max_nj_tiles);
            default: break;
        }
    }
    if (G == 18 && variant == 0) return launch_l2_t<18, 2, 3, 0, 2>(st, P, max_nj_tiles);
    if (G == 18 && variant == 2) return launch_l2_t<18, 2, 2, 3, 2>(st, P, max_nj_tiles);
#endif
    switch (G) {
        case 8:  return launch_l2_t<8, 2, 4, 3, 2>(st, P, max_nj_tiles);
        case 16: return launch_l2_t<16, 2, 4, 3, 2>(st, P, max_nj_tiles);
        case 18: return launch_l2_t<18, 2, 3, 3, 2>(st, P, max_nj_tiles);
        case 32: return launch_l2_t<32, 1, 4, 3, 2>(st, P, max_nj_tiles);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace r3dm
