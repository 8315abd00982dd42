// kernels_match.hip -- putative matching on gfx950 (MI355X): staging, fused squared-L2 2-NN on
// FP32 MFMA tiles with exact re-scoring + ratio test, Hamming 2-NN on integer VALU, and the
// per-pair finalisation (compaction, (i_,j_) ordering, coordinate de-duplication).
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

#include "r3dm_internal.hpp"

namespace r3dm {

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// pointers read out of ImgDev are generic; the hot loops cast them to the global address space so
// the compiler emits global_load (vmcnt only, SGPR base + lane offset) instead of flat_load
typedef const __attribute__((address_space(1))) f32x4* gf4p;
typedef const __attribute__((address_space(1))) float* gf1p;


// ------------------------------------------------------------------------------------------------
// staging: raw row-major descriptors -> rows (f32) + MFMA fragment-order tiles + norms
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256)
void stage_rows_kernel(const void* __restrict__ raw, int raw_is_u8, uint32_t n, uint32_t dim,
                       float* __restrict__ rows)
{
    const size_t total = (size_t)n * dim;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
        rows[e] = raw_is_u8 ? (float)((const uint8_t*)raw)[e] : ((const float*)raw)[e];
}

// one workgroup per 32-row tile
__global__ __launch_bounds__(256)
void stage_tiles_kernel(const float* __restrict__ rows, uint32_t n, uint32_t dim, uint32_t G,
                        float* __restrict__ tiled, uint16_t* __restrict__ tiled16, float* __restrict__ norms,
                        uint32_t* __restrict__ img_stats)
{
    const uint32_t t = blockIdx.x;
    const uint32_t per_tile = G * 256;                 // floats per tile = G * 2 * 32 * 4
    float* dst = tiled + (size_t)t * per_tile;
    for (uint32_t e = threadIdx.x; e < per_tile; e += 256) {
        const uint32_t c = e & 3, r = (e >> 2) & 31, h = (e >> 7) & 1, g = e >> 8;
        const uint32_t row = t * 32 + r, k = 8 * g + 4 * h + c;
        dst[e] = (row < n && k < dim) ? rows[(size_t)row * dim + k] : 0.0f;
    }
    // bf16 tiles of the integer fast path: the upper 16 bits of the float ARE the value when it is an integer of
    // magnitude <= 256 (8 significant bits); for any other view these tiles are never read (kernel-side check)
    const uint32_t GB = (G + 1) / 2, per_tile16 = GB * 512;
    uint16_t* dst16 = tiled16 + (size_t)t * per_tile16;
    for (uint32_t e = threadIdx.x; e < per_tile16; e += 256) {
        const uint32_t c8 = e & 7, r = (e >> 3) & 31, h = (e >> 8) & 1, kb = e >> 9;
        const uint32_t row = t * 32 + r, k = 16 * kb + 8 * h + c8;
        const float v = (row < n && k < dim) ? rows[(size_t)row * dim + k] : 0.0f;
        dst16[e] = (uint16_t)(__float_as_uint(v) >> 16);
    }
    if (threadIdx.x < 32) {
        const uint32_t row = t * 32 + threadIdx.x;
        float s = R3DM_INF;
        if (row < n) {
            s = 0.0f;
            const float* p = rows + (size_t)row * dim;
            float mx = 0.0f; bool nonint = false, neg = false;
            for (uint32_t k = 0; k < dim; ++k) {
                const float v = p[k];
                s = fmaf(v, v, s);
                mx = fmaxf(mx, fabsf(v));
                nonint |= !(v == rintf(v));                  // also true for NaN
                neg |= v < 0.0f;
            }
            // img_stats = &ImgDev::max_norm_bits, max_abs_bits, not_integer (non-negative floats order like uints)
            atomicMax(img_stats + 0, __float_as_uint(s));
            atomicMax(img_stats + 1, __float_as_uint(mx));
            if (nonint) atomicOr(img_stats + 2, 1u);       // ImgDev::not_integer: bit 0 = some non-integer, bit 1 = some negative
            if (neg) atomicOr(img_stats + 2, 2u);
        }
        norms[(size_t)t * 32 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256)
void stage_bin_kernel(const uint8_t* __restrict__ raw, uint32_t n, uint32_t nbytes,
                      uint32_t* __restrict__ bin, uint32_t words, uint32_t n_pad)
{
    const size_t total = (size_t)n_pad * words;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const uint32_t row = (uint32_t)(e / words), w = (uint32_t)(e % words);
        uint32_t v = 0;
        if (row < n)
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t byte = 4 * w + b;
                if (byte < nbytes) v |= (uint32_t)raw[(size_t)row * nbytes + byte] << (8 * b);
            }
        bin[e] = v;
    }
}

hipError_t launch_stage_f32(hipStream_t st, const void* raw, int raw_is_u8, uint32_t n, uint32_t dim,
                            float* rows, float* tiled, uint16_t* tiled16, float* norms, uint32_t G, uint32_t n_tiles,
                            uint32_t* img_stats_dev)
{
    if (n == 0) return hipSuccess;
    const size_t total = (size_t)n * dim;
    uint32_t grid = (uint32_t)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(stage_rows_kernel, dim3(grid), dim3(256), 0, st, raw, raw_is_u8, n, dim, rows);
    hipLaunchKernelGGL(stage_tiles_kernel, dim3(n_tiles), dim3(256), 0, st, rows, n, dim, G, tiled, tiled16, norms, img_stats_dev);
    return hipGetLastError();
}

hipError_t launch_stage_bin(hipStream_t st, const uint8_t* raw, uint32_t n, uint32_t nbytes,
                            uint32_t* bin, uint32_t words, uint32_t n_pad)
{
    if (n_pad == 0) return hipSuccess;
    const size_t total = (size_t)n_pad * words;
    uint32_t grid = (uint32_t)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(stage_bin_kernel, dim3(grid), dim3(256), 0, st, raw, n, nbytes, bin, words, n_pad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exact squared L2 in the reference's arithmetic (OpenMVG L2<float>): 4-way unrolled, float
// accumulator, ((d0^2 + d1^2) + d2^2) + d3^2 added to the running result, scalar tail, no FMA.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float exact_l2sq(const float* __restrict__ a, const float* __restrict__ b, uint32_t dim)
{
    float result = 0.0f;
    uint32_t k = 0;
    if ((dim & 3u) == 0) {
        const f32x4* a4 = (const f32x4*)a;
        const f32x4* b4 = (const f32x4*)b;
        for (; k < dim; k += 4) {
            const f32x4 x = a4[k >> 2], y = b4[k >> 2];
            const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
            result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        return result;
    }
    for (; k + 3 < dim; k += 4) {
        const float d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; k < dim; ++k) { const float d0 = a[k] - b[k]; result += d0 * d0; }
    return result;
}

// ------------------------------------------------------------------------------------------------
// running (best, runner-up, bound) list of one query column held by one lane
// ------------------------------------------------------------------------------------------------
struct Top2 {
    float d0, d1, d2;      // d0 <= d1 <= d2 ; d2 = smallest key NOT nominated (certification bound)
    uint32_t i0, i1;
};

__device__ __forceinline__ void top2_init(Top2& s)
{
    s.d0 = s.d1 = s.d2 = R3DM_INF; s.i0 = s.i1 = kNone;
}

__device__ __forceinline__ void top2_push(Top2& s, float key, uint32_t idx)
{
    // branch-free: locals first so every ?: is a plain select (v_cndmask), never control flow
    const float od0 = s.d0, od1 = s.d1, od2 = s.d2;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0;
    const bool c1 = key < od1;
    const uint32_t t1 = c1 ? idx : oi1;
    s.d2 = __builtin_amdgcn_fmed3f(od1, od2, key);     // min(d2, max(d1, key))
    s.d1 = __builtin_amdgcn_fmed3f(od0, od1, key);     // min(d1, max(d0, key))
    s.d0 = __builtin_amdgcn_fmed3f(-R3DM_INF, od0, key);   // min(d0, key) as one v_med3_f32 (no canonicalising v_max)
    s.i1 = c0 ? oi0 : t1;
    s.i0 = c0 ? idx : oi0;
}

// write the verdict for one query: ratio test, optional 2-NN dump
__device__ __forceinline__ void emit_result(const MatchParams& P, uint32_t pair, uint32_t q,
                                            float ea, uint32_t ia, float eb, uint32_t ib)
{
    const size_t o = (size_t)pair * P.q_stride + q;
    P.nn_idx[o] = (ib != kNone && ea < P.ratio_R * eb) ? ia : kNone;
    if (P.knn_idx) {
        P.knn_idx[2 * o] = (int32_t)ia; P.knn_idx[2 * o + 1] = (int32_t)ib;
        P.knn_dist[2 * o] = ea;         P.knn_dist[2 * o + 1] = eb;
    }
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
// 16-byte buffer load: wave-uniform descriptor + SGPR byte offset + per-lane 32-bit offset -- no 64-bit
// per-lane address registers in the hot loop (the pointer form spilled at 256 VGPRs)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

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

// ---- per query: merge the two lane halves, re-score exactly, certify, ratio-test (tail of both L2 kernels).
// dpad = padded descriptor length, bf16_tiles = the keys come from the integer fast path
__device__ __forceinline__ void lex_push(Top2& s, float key, uint32_t idx)
{
    const float od0 = s.d0, od1 = s.d1;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0 || (key == od0 && idx < oi0);
    const bool c1 = key < od1 || (key == od1 && idx < oi1);
    s.d1 = c0 ? od0 : (c1 ? key : od1);
    s.i1 = c0 ? oi0 : (c1 ? idx : oi1);
    s.d0 = c0 ? key : od0;
    s.i0 = c0 ? idx : oi0;
}

// LEX: the lists are exact lexicographic (distance, index) top-2 lists without a bound (l2_knn2_int_kernel)
// SPLIT: the keys come from the split-f16 nominator (l2_knn2_split_kernel) in units of key_inv^-1; a query whose merged
//        top-2 cannot be certified gets a second chance with all four nominees of its two lane halves before it is sent to
//        the exact scan
// lane_key_inv (count tiles, l2_knn2_counts_kernel): the keys of query tile nj are in units of lane_key_inv[nj]^-1, a value per QUERY
//        (both lane halves of a column hold the same one)
template <int NJ, bool LEX = false, bool SPLIT = false>
__device__ __forceinline__ void l2_finish_queries(const MatchParams& P, uint32_t pair, const ImgDev* __restrict__ Ip,
                                                  const ImgDev* __restrict__ Jp, const Top2 (&st)[NJ], uint32_t qt0,
                                                  uint32_t h, uint32_t c, float dpad, bool bf16_tiles,
                                                  float key_inv = 1.0f, float slack_abs = 0.0f, const float* lane_key_inv = nullptr)
{
    const uint32_t nI = Ip->n, nJ = Jp->n, ntJ = Jp->n_tiles;
    const float maxnorm = __uint_as_float(Ip->max_norm_bits);
    const uint32_t dim = Ip->dim;
    // Exactness proof for integer-valued descriptors (e.g. SIFT bins 0..255): when every element of
    // both views is an integer and all partial sums stay below 2^24, the MFMA pass (norm init, fma
    // chain, + ||q||^2) and the reference's sum of squared differences are BOTH exact, hence equal:
    // no rounding slack is needed and only true ties with an un-nominated row need the exact scan.
    // Non-negative data: ||a||^2 <= D mI^2 and the running ||a||^2 - 2 sum(a q) stays within [-2 D mI mJ, D mI^2]; the distance
    // itself is at most D max(mI, mJ)^2.  With negative elements the partial sums reach D mI^2 + 2 D mI mJ and the distance
    // D (mI + mJ)^2, where the reference's own sum starts to round: one bound on the latter covers both.
    const float mI = __uint_as_float(Ip->max_abs_bits), mJ = __uint_as_float(Jp->max_abs_bits);
    const uint32_t fl = Ip->not_integer | Jp->not_integer;              // bit 0: non-integer, bit 1: negative elements
    const bool exact_pair = !SPLIT && (fl & 1u) == 0u &&
                            ((fl & 2u) ? dpad * (mI + mJ) * (mI + mJ) < 16777216.0f
                                       : (2.0f * dpad * mI * mJ < 16777216.0f && dpad * mI * mI < 16777216.0f && dpad * mJ * mJ < 16777216.0f)) &&
                            (!bf16_tiles || (mI <= 256.0f && mJ <= 256.0f));      // bf16 tiles hold the values exactly
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        Top2 s = st[nj];
        if constexpr (SPLIT) {
            const float ki = lane_key_inv ? lane_key_inv[nj] : key_inv;                 // positive scale: order unchanged
            s.d0 *= ki; s.d1 *= ki; s.d2 *= ki;
        }
        const Top2 own = s;                              // this lane half's list (rows 8 qd + 4 h + k of every tile)
        // partner half (same query column, the other 16 rows of every tile)
        const float pd0 = __shfl_xor(s.d0, 32), pd1 = __shfl_xor(s.d1, 32), pd2 = __shfl_xor(s.d2, 32);
        const uint32_t pi0 = __shfl_xor(s.i0, 32), pi1 = __shfl_xor(s.i1, 32);
        if constexpr (LEX) {
            lex_push(s, pd0, pi0);
            lex_push(s, pd1, pi1);
            s.d2 = R3DM_INF;                               // nothing un-nominated can tie or beat an exact top-2
        } else {
            top2_push(s, pd0, pi0);
            top2_push(s, pd1, pi1);
            s.d2 = fminf(s.d2, pd2);
        }
        // make both halves agree on the nominated pair (lane c's view)
        const uint32_t ci0 = __shfl(s.i0, (int)c), ci1 = __shfl(s.i1, (int)c);
        const float bound = __shfl(s.d2, (int)c);

        const uint32_t qt = qt0 + nj;
        const uint32_t q = qt * 32u + c;
        const bool valid = (qt < ntJ) && (q < nJ);
        const uint32_t cand = h ? ci1 : ci0;
        const float cd0 = __shfl(s.d0, (int)c), cd1 = __shfl(s.d1, (int)c);     // (both shuffles outside the lane-dependent select)
        const float ck = h ? cd1 : cd0;                                         // MFMA key ||a||^2 - 2 a.b of this lane's nominee
        float e = R3DM_INF;
        if (valid && cand != kNone) {
            // exact pairs (proof above): key + ||q||^2 IS the reference distance, bit for bit -- no need to fetch the two
            // nominated rows again (that re-read was 60 % of the kernel's HBM-side traffic: 3 x 512 B per query).
            // Otherwise re-score in the reference's summation order.
            if (exact_pair) e = ck + Jp->norms[q];
            else e = exact_l2sq(Ip->rows + (size_t)cand * dim, Jp->rows + (size_t)q * dim, dim);
        }
        const float eo = __shfl_xor(e, 32);
        float ea = h ? eo : e, eb = h ? e : eo;          // ea <-> ci0, eb <-> ci1
        uint32_t ia = ci0, ib = ci1;
        if (eb < ea || (eb == ea && ib < ia)) { const float tf = ea; ea = eb; eb = tf; const uint32_t tu = ia; ia = ib; ib = tu; }
        // certification (evaluated identically by both lane halves of a query)
        const float nb = valid ? Jp->norms[q] : 0.0f;
        const float slack = exact_pair ? 0.0f : P.err_scale * (maxnorm + nb) + slack_abs;
        const float A3 = bound + nb;                        // distance of the best un-nominated row (exact if exact_pair)
        // certified: every un-nominated row is strictly farther than the runner-up.  With exact
        // arithmetic a runner-up that merely TIES an un-nominated row still fixes the best row (ea < eb)
        // and the runner-up DISTANCE, which is all the ratio test needs; only the raw 2-NN dump
        // (r3dm_knn2) needs the tie's index resolved by the exact scan.
        bool certified = (eb < A3 - slack) || (exact_pair && P.knn_idx == nullptr && ea < eb && eb <= A3);
        if (bf16_tiles && !exact_pair) certified = false;      // bf16 keys of a non-exact pair mean nothing: exact scan
        // Match mode only needs the VERDICT of the ratio test.  The two re-scored nominees bound the true runner-up distance from
        // above (d2 <= eb: two rows are no farther than eb) and, with the un-nominated rows' lower bound L = bound + ||q||^2 -
        // slack, the true best distance from below (d1 >= min(ea, L)).  If min(ea, L) >= R eb then d1 >= R d2 whatever the exact
        // top-2 is: the query has no match, exactly as the exact scan would find -- and that is the fate of nearly every
        // uncertifiable query (descriptors without a counterpart sit at almost equal distances from their nearest rows).
        bool no_match = false;
        if (!certified && !exact_pair && !bf16_tiles && P.knn_idx == nullptr && valid && nI >= 2 && ib != kNone) {
            float L = A3 - slack;
            L -= fabsf(L) * 9.5367431640625e-07f;            // 2^-20: the float evaluation of L itself
            no_match = fminf(ea, L) >= P.ratio_R * eb;
        }
        if constexpr (SPLIT) {
            // second chance: the two lane halves of a query nominated up to four rows between them.  Re-score all four in the
            // reference arithmetic and certify against the smallest key that NONE of them holds (each half's third key):
            // the gap from the runner-up to the fifth-best row is what has to exceed the slack now, not the gap to the third.
            const bool need = valid && nI >= 2 && !certified && !no_match;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                float f0 = R3DM_INF, f1 = R3DM_INF;
                if (need) {
                    const float* qrow = Jp->rows + (size_t)q * dim;
                    if (own.i0 != kNone) f0 = exact_l2sq(Ip->rows + (size_t)own.i0 * dim, qrow, dim);
                    if (own.i1 != kNone) f1 = exact_l2sq(Ip->rows + (size_t)own.i1 * dim, qrow, dim);
                }
                Top2 m4; top2_init(m4);
                lex_push(m4, f0, own.i0); lex_push(m4, f1, own.i1);
                const float g0 = __shfl_xor(f0, 32), g1 = __shfl_xor(f1, 32);
                const uint32_t j0 = __shfl_xor(own.i0, 32), j1 = __shfl_xor(own.i1, 32);
                lex_push(m4, g0, j0); lex_push(m4, g1, j1);
                const float bound4 = fminf(own.d2, __shfl_xor(own.d2, 32));
                if (need && m4.i1 != kNone && m4.d1 < (bound4 + nb) - slack) {
                    ea = m4.d0; ia = m4.i0; eb = m4.d1; ib = m4.i1; certified = true;
                }
            }
        }
        if (valid && h == 0) {
            if (nI < 2) {
                emit_result(P, pair, q, R3DM_INF, kNone, R3DM_INF, kNone);
            } else if (certified) {
                emit_result(P, pair, q, ea, ia, eb, ib);
            } else if (no_match) {
                P.nn_idx[(size_t)pair * P.q_stride + q] = kNone;
            } else {
                P.nn_idx[(size_t)pair * P.q_stride + q] = kFallback;
                const uint32_t pos = atomicAdd(P.fb_cnt + pair, 1u);
                atomicAdd(P.fb_total, 1u);
                if (pos < kFbPerPair) P.fb_q[(size_t)pair * kFbPerPair + pos] = q;
                else atomicAdd(P.fb_total + 1, 1u);
            }
        }
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

// ------------------------------------------------------------------------------------------------
// integer fast path (r3dm_set_integer_mfma): the same contraction on v_mfma_f32_32x32x16_bf16.
// Views whose descriptors are integers of magnitude <= 256 (SIFT bins) are staged a second time as bf16 tiles
// (ImgDev::tiled16, [tile][16-dim block][lane half][32 rows][8 bf16] -- 16 bytes per lane and step like the f32
// tiles, half as many steps).  Every value is a bf16, every product and partial sum an integer below 2^24, so the f32
// accumulators hold exactly the values of the f32 path and of the reference's sum of squared differences
// (l2_finish_queries re-checks the condition per pair; anything else goes to the exact scan).
// At 32 cycles per MFMA (16x fewer matrix cycles) the VALU side of l2_tile_step -- 10.7 VALU instructions per MFMA:
// accumulator init, one compare per key, 8-instruction pushes into (best, runner-up, bound) lists -- would hold the
// issue port longer than the matrix pipe runs.  Exact keys allow less:
//   * lists hold (best, runner-up) only.  Keys are exact and every lane sees its rows in increasing index order, so
//     strict '<' keeps the lexicographic (distance, index) top-2 of the lane's rows, and a lexicographic merge of the
//     two lane halves IS the exact top-2 -- no certification bound, a third fewer list updates;
//   * one wave-wide test per FOUR keys of a list (v_min3 + v_min + v_cmp instead of four v_cmp);
//   * the accumulators start from the norm vector through the MFMA's C operand (8 v_mov_b64 per tile instead of 32 v_mov).
// Measured (780 pairs of 8192 x 8192 rows): f32 tiles 95.0 ms; this kernel 12.4 ms (12.96 before the per-key tests in the
// update path) (the f32 kernel's structure on bf16
// tiles: 14.97 ms; without any epilogue: 11.4 ms).  The shader clock drops from 2.32 GHz (f32 kernel) to 1.84 GHz under
// the bf16 matrix load (GRBM_GUI_ACTIVE / duration), so 12.4 ms is 56 % of the clocked bf16 peak.  Sharing the dataset
// tiles of a workgroup through LDS (a quarter of the L1 traffic) measured 16.0 ms against 15.0 ms and was dropped.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__device__ __forceinline__ void tope_push(Top2& s, float key, uint32_t idx)
{
    const float od0 = s.d0, od1 = s.d1;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0;
    const bool c1 = key < od1;
    const uint32_t t1 = c1 ? idx : oi1;
    s.d1 = __builtin_amdgcn_fmed3f(od0, od1, key);
    s.d0 = vmin2(od0, key);
    s.i1 = c0 ? oi0 : t1;
    s.i0 = c0 ? idx : oi0;
}

template <int GB, int NJ, int PF, int ABL>
__device__ __forceinline__ void int_tile_step(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rn, uint32_t voffA, uint32_t voffN,
                                              uint32_t soffA, uint32_t soffN, f32x4 (&abuf)[PF], const f32x16& nrm_cur, f32x16& nrm_next,
                                              const f32x4 (&bq)[NJ][GB], f32x16 (&cur)[NJ], const f32x16 (&prev)[NJ],
                                              Top2 (&st)[NJ], uint32_t prev_rowbase)
{
    constexpr int NG = 4 * NJ;                             // (list, quad) groups of four keys per tile
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f32x4 a = abuf[g % PF];
        abuf[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 1024u);
        if (g == (GB > 2 ? 2 : GB - 1)) {   // next tile's norms, element 4 qd + k = row 8 qd + 4 h + k: the accumulator layout
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 v = bload16(rn, voffN, soffN + (uint32_t)qd * 32u);
#pragma unroll
                for (int k = 0; k < 4; ++k) nrm_next[4 * qd + k] = v[k];
            }
        }
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq[nj][g]),
                                                              g == 0 ? nrm_cur : cur[nj], 0, 0, 0);
#pragma unroll
        for (int gi = (g * NG) / GB; gi < ((g + 1) * NG) / GB; ++gi) {
            const int nj = gi % NJ, qd = gi / NJ;
            const float p0 = prev[nj][4 * qd], p1 = prev[nj][4 * qd + 1], p2 = prev[nj][4 * qd + 2], p3 = prev[nj][4 * qd + 3];
            if constexpr (ABL != 0) {
                asm volatile("" ::"v"(p0), "v"(p1), "v"(p2), "v"(p3));
            } else {
                // Step 0 may follow the previous tile's last MFMAs (the writers of p0..p3) closely: its minimum goes through
                // ordinary fminf so that the compiler's MFMA -> VALU hazard pass sees the read; from step 1 on at least NJ
                // MFMAs and a sched_barrier lie in between and the two-instruction asm form is safe.
                const float m = g == 0 ? __builtin_fminf(__builtin_fminf(p0, p1), __builtin_fminf(p2, p3)) : vmin2(vmin3(p0, p1, p2), p3);
                if (__builtin_amdgcn_ballot_w64(m < st[nj].d1) != 0ull) {
                    // some lane improves on one of the four keys: usually ONE key does, so test each before its 7-instruction push
                    // (the 16-step body of D = 256 stays with unconditional pushes: the compiler gives up unrolling the larger one)
                    const uint32_t rb = prev_rowbase + 8u * (uint32_t)qd;
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p0 < st[nj].d1) != 0ull) tope_push(st[nj], p0, rb);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p1 < st[nj].d1) != 0ull) tope_push(st[nj], p1, rb + 1u);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p2 < st[nj].d1) != 0ull) tope_push(st[nj], p2, rb + 2u);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p3 < st[nj].d1) != 0ull) tope_push(st[nj], p3, rb + 3u);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0>
__global__ __launch_bounds__(256, WPS)
void l2_knn2_int_kernel(const MatchParams P)
{
    static_assert(GB % PF == 0, "prefetch window must divide the block count");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
    uint32_t pair, qb;
    if (P.xcd_map) {                                       // pair p on XCD p % 8 (see l2_knn2_mfma_kernel)
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
    if (qt0 >= ntJ) return;                                // wave-uniform; no barriers in this kernel

    f32x4 bq[NJ][GB];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;
        const gf4p src = (gf4p)(const void*)Jp->tiled16 + (size_t)qt * (GB * 64) + lane;
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const u32x4 w = __builtin_bit_cast(u32x4, src[g * 64]);     // -2 x (integer, |x| <= 256) is a bf16 again
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(w[k] << 16) * -2.0f, hi = __uint_as_float(w[k] & 0xFFFF0000u) * -2.0f;
                o[k] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
            }
            bq[nj][g] = __builtin_bit_cast(f32x4, o);
        }
    }
    Top2 st[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) top2_init(st[nj]);     // d2 stays +inf: these lists carry no bound

    if (nI >= 2) {
        const uint64_t pa = (uint64_t)Ip->tiled16, pn = (uint64_t)Ip->norms;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa)),
            0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pn >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pn)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffA = lane * 16u, voffN = h * 16u;
        constexpr uint32_t tileB = (uint32_t)GB * 1024u;
        const uint32_t hb = 4u * h;
        f32x4 abuf[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) abuf[s] = bload16(ra, voffA, (uint32_t)s * 1024u);
        f32x16 nrmA, nrmB;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = bload16(rn, voffN, (uint32_t)qd * 32u);
#pragma unroll
            for (int k = 0; k < 4; ++k) nrmA[4 * qd + k] = v[k];
        }
        f32x16 accA[NJ], accB[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[nj][r] = R3DM_INF;              // "tile -1": keys that never win
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            int_tile_step<GB, NJ, PF, ABL>(ra, rn, voffA, voffN, t * tileB + PF * 1024u, (t + 1) * 128u, abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
            int_tile_step<GB, NJ, PF, ABL>(ra, rn, voffA, voffN, (t + 1) * tileB + PF * 1024u, (t + 2) * 128u, abuf, nrmB, nrmA, bq, accB, accA, st, t * 32u + hb);
        }
        if (t < ntI) {
            int_tile_step<GB, NJ, PF, ABL>(ra, rn, voffA, voffN, t * tileB + PF * 1024u, (t + 1) * 128u, abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) tope_push(st[nj], accA[nj][r], t * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
        } else {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) tope_push(st[nj], accB[nj][r], (ntI - 1) * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
        }
    }
    l2_finish_queries<NJ, true>(P, pair, Ip, Jp, st, qt0, h, c, (float)(GB * 16), true);
}

// ------------------------------------------------------------------------------------------------
// split-f16 nomination for real-valued descriptors (r3dm_set_split_mfma): LIOP-144, normalised SIFT -- what Regard3D
// actually matches (/root/reference/src/Regard3DFeatures.h:44-48).  Their path through l2_knn2_mfma_kernel is already
// "nominate on MFMA keys -> re-score the nominees in the reference arithmetic -> certify against a rounding slack", so the
// nominator need not run on f32 tiles.  Every value x of a view is scaled by the view's power of two s (max|x| s in
// [2^13, 2^14)) and split into two f16 pieces x s = hi + lo + r with |r| <= 2^-22 |x s| (f16 carries 11 significant bits;
// pieces below the f16 normal range lose at most 2^-25 absolutely), and
//     a.b  ~  ah.bh + al.bh + ah.bl            (the dropped al.bl term is <= 2^-22 |a||b| too)
// runs as three v_mfma_f32_32x32x16_f16 per 16 dimensions: 96 matrix cycles against 512 on the f32 tiles.  Products of f16
// values are exact in f32, so the key differs from the exact one by the split residue (3 x 2^-22 ||a|| ||b||) plus the
// f32 accumulation of 3 D products (bounded with a one-sided 2^-23 per addition, i.e. without assuming round-to-nearest
// inside the matrix unit); host: MatchParams::err_scale = (3 Dpad + 34) 2^-22.  That is 3x the slack of the f32 tiles, which
// is why the tail (l2_finish_queries<SPLIT>) gives an uncertified query a second chance with the four nominees of its two
// lane halves.  Results stay bit-identical to the oracle: certification or exact scan, as on the f32 tiles.
// Layout: ImgDev::tiledh = [tile][16-dim block][hi | lo][lane half][32 rows][8 f16] -- 2 KiB per block, one contiguous
// stream per view.  The wave keeps the hi fragments of its NJ query tiles in registers and their lo fragments in LDS
// (written once, read by the same wave only: no barrier in the loop); dataset hi / lo fragments stream through a PF-deep
// register window like the f32 kernel's.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }    // -126 <= k <= 127

// one workgroup per 32-row tile: the two f16 planes of the view, scaled by 2^split_k (read from the image table: the
// statistics kernel ahead of this one on the stream produced max|x|)
__global__ __launch_bounds__(256)
void stage_split_kernel(const float* __restrict__ rows, uint32_t n, uint32_t dim, uint32_t GB, uint16_t* __restrict__ tiledh,
                        const uint32_t* __restrict__ img_stats, int32_t* __restrict__ split_k_out)
{
    const float mx = __uint_as_float(img_stats[1]);
    int k = 0;
    if (mx > 0.0f && mx < R3DM_INF) {
        k = 13 - ((int)((__float_as_uint(mx) >> 23) & 0xFFu) - 127);           // max|x| 2^k in [2^13, 2^14)
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *split_k_out = k;
    const float sc = pow2f(k);
    const uint32_t t = blockIdx.x;
    uint16_t* dst = tiledh + (size_t)t * GB * 1024;                              // halves per tile = GB * 2 planes * 512
    for (uint32_t e = threadIdx.x; e < GB * 512; e += 256) {
        const uint32_t c8 = e & 7, r = (e >> 3) & 31, h = (e >> 8) & 1, kb = e >> 9;
        const uint32_t row = t * 32 + r, kk = 16 * kb + 8 * h + c8;
        const float v = (row < n && kk < dim) ? rows[(size_t)row * dim + kk] * sc : 0.0f;
        const _Float16 hi = (_Float16)v;                                         // round to nearest even
        const _Float16 lo = (_Float16)(v - (float)hi);                           // the subtraction is exact in f32
        const uint32_t o = kb * 1024 + (h * 32 + r) * 8 + c8;
        dst[o] = __builtin_bit_cast(uint16_t, hi);
        dst[o + 512] = __builtin_bit_cast(uint16_t, lo);
    }
}

hipError_t launch_stage_split(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, uint32_t GB, uint32_t n_tiles,
                              uint16_t* tiledh, const uint32_t* img_stats_dev, int32_t* split_k_dev)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_split_kernel, dim3(n_tiles), dim3(256), 0, st, rows, n, dim, GB, tiledh, img_stats_dev, split_k_dev);
    return hipGetLastError();
}

template <int GB, int NJ, int PF>
__device__ __forceinline__ void split_tile_step(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rn, uint32_t voffA, uint32_t voffN,
                                                uint32_t soffA, uint32_t soffN, f32x4 (&ah)[PF], f32x4 (&al)[PF], f32x4 (&nrm)[4], float cscale,
                                                const f32x4 (&bqh)[NJ][GB], const f32x4* __restrict__ bl_lds, f32x16 (&cur)[NJ],
                                                const f32x16 (&prev)[NJ], Top2 (&st)[NJ], uint32_t prev_rowbase)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = nrm[r >> 2][r & 3] * cscale;       // ||a||^2 in key units (sI sJ); +inf for padding rows
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) cur[nj][r] = v;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f16x8 a_hi = __builtin_bit_cast(f16x8, ah[g % PF]);
        const f16x8 a_lo = __builtin_bit_cast(f16x8, al[g % PF]);
        ah[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 2048u);
        al[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 2048u + 1024u);
        if (g == 1) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) nrm[qd] = bload16(rn, voffN, soffN + (uint32_t)qd * 32u);
        }
        f32x4 bl[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) bl[nj] = bl_lds[(nj * GB + g) * 64];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, __builtin_bit_cast(f16x8, bqh[nj][g]), cur[nj], 0, 0, 0);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, __builtin_bit_cast(f16x8, bl[nj]), cur[nj], 0, 0, 0);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, __builtin_bit_cast(f16x8, bqh[nj][g]), cur[nj], 0, 0, 0);
        // this block's share of the previous tile's keys: wave-wide test-and-skip, as in l2_tile_step<PIPE 3>
        bool any = false;
#pragma unroll
        for (int r = (g * 16) / GB; r < ((g + 1) * 16) / GB; ++r)
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) any |= prev[nj][r] < st[nj].d2;
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
            for (int r = (g * 16) / GB; r < ((g + 1) * 16) / GB; ++r)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
                    top2_push(st[nj], prev[nj][r], prev_rowbase + (uint32_t)((r & 3) + 8 * (r >> 2)));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int GB, int NJ, int PF>
__global__ __launch_bounds__(256, 2)
void l2_knn2_split_kernel(const MatchParams P)
{
    static_assert(GB % PF == 0, "prefetch window must divide the block count");
    extern __shared__ __attribute__((aligned(16))) unsigned char split_smem[];     // [wave][NJ][GB][64 lanes] x 16 B: query lo fragments
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
    uint32_t pair, qb;
    if (P.xcd_map) {                                       // pair p on XCD p % 8 (see l2_knn2_mfma_kernel)
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
    if (qt0 >= ntJ) return;                                // wave-uniform; no workgroup barriers in this kernel
    const int kI = Ip->split_k, kJ = Jp->split_k;
    const float cscale = pow2f(kI + kJ);                   // key units: sI sJ (||a||^2 - 2 a.b)
    const float key_inv = pow2f(-(kI + kJ));

    // ---- query fragments (B operand), scaled by -2 (exact in f16): hi in registers, lo in this wave's LDS slice
    f32x4* bl_lds = reinterpret_cast<f32x4*>(split_smem) + (size_t)wave * (NJ * GB * 64) + lane;
    f32x4 bqh[NJ][GB];
    const f16x8 m2 = {(_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f, (_Float16)-2.0f};
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;          // clamp: results discarded below
        const gf4p src = (gf4p)(const void*)Jp->tiledh + (size_t)qt * (GB * 128) + lane;     // 128 float4 per block (hi | lo)
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            bqh[nj][g] = __builtin_bit_cast(f32x4, __builtin_bit_cast(f16x8, src[g * 128]) * m2);
            bl_lds[(nj * GB + g) * 64] = __builtin_bit_cast(f32x4, __builtin_bit_cast(f16x8, src[g * 128 + 64]) * m2);
        }
    }
    Top2 st[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) top2_init(st[nj]);

    if (nI >= 2) {
        const uint64_t pa = (uint64_t)Ip->tiledh, pn = (uint64_t)Ip->norms;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa)),
            0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pn >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pn)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffA = lane * 16u, voffN = h * 16u;
        constexpr uint32_t tileB = (uint32_t)GB * 2048u;
        const uint32_t hb = 4u * h;
        f32x4 ah[PF], al[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) { ah[s] = bload16(ra, voffA, (uint32_t)s * 2048u); al[s] = bload16(ra, voffA, (uint32_t)s * 2048u + 1024u); }
        f32x4 nrm[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) nrm[qd] = bload16(rn, voffN, (uint32_t)qd * 32u);
        f32x16 accA[NJ], accB[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[nj][r] = R3DM_INF;              // "tile -1": keys that never win
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            split_tile_step<GB, NJ, PF>(ra, rn, voffA, voffN, t * tileB + PF * 2048u, (t + 1) * 128u, ah, al, nrm, cscale, bqh, bl_lds, accA, accB, st, (t - 1) * 32u + hb);
            split_tile_step<GB, NJ, PF>(ra, rn, voffA, voffN, (t + 1) * tileB + PF * 2048u, (t + 2) * 128u, ah, al, nrm, cscale, bqh, bl_lds, accB, accA, st, t * 32u + hb);
        }
        if (t < ntI) {
            split_tile_step<GB, NJ, PF>(ra, rn, voffA, voffN, t * tileB + PF * 2048u, (t + 1) * 128u, ah, al, nrm, cscale, bqh, bl_lds, accA, accB, st, (t - 1) * 32u + hb);
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
    }
    // absolute part of the slack: pieces below the f16 normal range lose up to 2^-25 each (in scaled units), against an operand
    // of magnitude < 2^14 on the other side, two sides, key = -2 a.b  ->  Dpad 2^-9 in key units
    l2_finish_queries<NJ, false, true>(P, pair, Ip, Jp, st, qt0, h, c, (float)(GB * 16), false, key_inv, (float)(GB * 16) * 0.001953125f * key_inv);
}

template <int GB, int NJ, int PF>
static hipError_t launch_l2_split_t(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    P.xcd_map = 1u;
    const uint64_t grid64 = (uint64_t)((P.n_pairs + 7u) / 8u * 8u) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    const size_t lds = (size_t)4 * NJ * GB * 1024;
    hipError_t e = hipFuncSetAttribute((const void*)l2_knn2_split_kernel<GB, NJ, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((l2_knn2_split_kernel<GB, NJ, PF>), dim3((uint32_t)grid64), dim3(256), lds, st, P);
    return hipGetLastError();
}

// G = padded dim / 8 of the views (8, 16, 18, 32); hipErrorInvalidValue -> no split kernel, caller keeps the f32 tiles
hipError_t launch_l2_knn2_split(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles)
{
    switch (G) {
        case 8:  return launch_l2_split_t<4, 2, 4>(st, P, max_nj_tiles);
        case 16: return launch_l2_split_t<8, 2, 4>(st, P, max_nj_tiles);
        case 18: return launch_l2_split_t<9, 2, 3>(st, P, max_nj_tiles);
        case 32: return launch_l2_split_t<16, 1, 4>(st, P, max_nj_tiles);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// Count tiles: nomination for rows that are SMALL INTEGERS TIMES A PER-ROW SCALE (round 4).  That is what a LIOP descriptor is --
// the vector Regard3D matches (/root/reference/src/Regard3DFeatures.h:44-48): vl_liop accumulates integer votes per bin and divides
// by their norm (/root/reference/src/thirdparty/liop/vl_liop.c:553-575), a_i = c_i / n with c_i an integer of a few hundred at most.
// Integers up to 2048 ARE f16 values, their products are exact in the matrix unit's f32, so
//     a.b = (c_a . c_b) s_a s_b
// needs ONE v_mfma_f32_32x32x16_f16 per 16 dimensions on the count tiles, where the split nominator above spends three on hi / lo
// pieces of the float values: a third of the matrix cycles for the path the product runs by default.  The scales enter afterwards,
// in the test-and-skip epilogue, and only where a key can matter:
//     key_q(a) = ||a||^2 / (2 s_q) - (c_a . c_q) s_a        [ = reference key (||a||^2 - 2 a.q) / (2 s_q): per query a positive scale ]
// with B = -c_q the accumulator holds D' = -(c_a . c_q) <= 0, and for the four keys of a lane's accumulator quad
//     min key >= min(||a||^2) / (2 s_q) + min(D') max(s_a):
// one min3 + min + mul + fma + compare per four keys; the per-key mul + fma run only for a quad that passes (rare once the lists
// have warmed up).  Everything behind the nomination is the split path's: the nominees are re-scored in the reference's own f32
// summation order, certified against the rounding slack (the key error here -- f32 accumulation of exact products, the 2^-21
// representation tolerance checked at staging, three roundings in the epilogue -- is below the split residue the slack was sized
// for), uncertified queries take the four-nominee second chance and then the exact scan.  Results are bit-identical to every other path.
// Eligibility is decided per view at staging (stage_counts_kernel): every row must satisfy |a_i - c_i s| <= 2^-21 max|a| with integers
// 0 <= c_i <= 2047; a view with one row that does not (any descriptor that is not of this form) keeps the split tiles.
// ------------------------------------------------------------------------------------------------
// one workgroup per 32-row tile, a wave per row (eight rows each): recover (c, s) of the row, verify, write the f16 counts in
// fragment order [tile][16-dim block][lane half][32 rows][8 f16] and the row's scale; *fail is set when a row is not of the form
__global__ __launch_bounds__(256)
void stage_counts_kernel(const float* __restrict__ rows, uint32_t n, uint32_t dim, uint32_t GB, uint16_t* __restrict__ tiledc,
                         float* __restrict__ cscale, uint32_t* __restrict__ fail)
{
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint16_t* dst = tiledc + (size_t)t * GB * 512;
    for (uint32_t r = wave; r < 32u; r += 4u) {
        const uint32_t row = t * 32u + r;
        if (row >= n) { if (lane == 0) cscale[row] = 1.0f; continue; }     // (padding rows: counts stay zero, norms are +inf)
        const float* a = rows + (size_t)row * dim;
        float v[4];
        float amax = 0.0f, amin = R3DM_INF;
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = lane + 64u * (uint32_t)e;
            v[e] = k < dim ? a[k] : 0.0f;
            if (!(v[e] >= 0.0f) || !(v[e] < R3DM_INF)) bad = true;              // negative, NaN, inf: not a count row
            amax = fmaxf(amax, v[e]);
            if (v[e] > 0.0f) amin = fminf(amin, v[e]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, off)); amin = fminf(amin, __shfl_xor(amin, off)); }
        bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
        float cnt[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float sc = 1.0f;
        bool ok = !bad;
        if (ok && amax > 0.0f) {
            // the smallest positive element is k x s for a small integer k: try k = 1, 2, ...
            ok = false;
            for (uint32_t k = 1; k <= 64u && !ok; ++k) {
                const float s_try = amin / (float)k;
                if (!(amax / s_try <= 2047.5f)) break;
                bool fits = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = v[e] / s_try;
                    cnt[e] = rintf(q);
                    fits = fits && fabsf(q - cnt[e]) <= 0.0625f;                  // coarse: the fit below is what counts
                }
                if (__builtin_amdgcn_ballot_w64(!fits) != 0ull) continue;
                // least-squares scale of the row, then the tolerance every element must meet
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e] * cnt[e]; s2 += cnt[e] * cnt[e]; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
                sc = s1 / s2;
                bool tol = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) tol = tol && fabsf(v[e] - cnt[e] * sc) <= 4.76837158203125e-07f * amax;    // 2^-21
                ok = __builtin_amdgcn_ballot_w64(!tol) == 0ull;
            }
        } else if (ok) {
            sc = 1.0f;                                       // a zero row: counts 0, any scale
        }
        if (!ok) { if (lane == 0) atomicOr(fail, 1u); continue; }
        if (lane == 0) cscale[row] = sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t kk = lane + 64u * (uint32_t)e;
            if (kk < GB * 16u) {
                const uint32_t c8 = kk & 7u, hh = (kk >> 3) & 1u, kb = kk >> 4;
                dst[kb * 512u + (hh * 32u + r) * 8u + c8] = __builtin_bit_cast(uint16_t, (_Float16)cnt[e]);
            }
        }
    }
}

// The DATASET side of the nominator reads the rows in the order of their scales (32 classes per binary order, i.e. scales within
// 2.2 % of one another inside a class): the quad test of the kernel bounds four keys with the largest scale of their four rows, and
// with rows in keypoint order (scales 30 % apart) that bound let a large share of the quads through to the per-key path.
// One workgroup: counting sort of the rows by scale class -> cperm[position] = row (kNone behind the last row).
// The rows of a class keep their keypoint order (a stable sort): the atomic cursors place them in whatever order the waves arrive, so a
// second kernel ranks every row inside its class segment by row index, one thread per row over the whole chip -- which rows share a
// tile, and with it which queries the epilogue sends to the exact scan, is then the same from run to run (the results are exact
// either way).  scratch: n_pad words (rows of a class, unordered) + 2 n_pad words (every row's class segment).
// (Ranking inside this one-workgroup kernel was tried first: a view's rows fall into a few hundred classes, 14 M serial reads per view,
// +115 ms on the stage's 24 views.)
__global__ __launch_bounds__(1024)
void stage_counts_order_kernel(const float* __restrict__ cscale, uint32_t n, uint32_t n_pad, uint32_t* __restrict__ cperm, uint32_t* __restrict__ scratch)
{
    __shared__ uint32_t hist[8192];
    __shared__ uint32_t start[8192];
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t* __restrict__ tmp = scratch;
    uint2* __restrict__ seg = reinterpret_cast<uint2*>(scratch + n_pad);
    for (uint32_t b = tid; b < 8192u; b += 1024u) hist[b] = 0u;
    __syncthreads();
    for (uint32_t r = tid; r < n; r += 1024u) atomicAdd(&hist[(__float_as_uint(cscale[r]) >> 18) & 8191u], 1u);    // sign 0: exponent + 5 mantissa bits
    __syncthreads();
    uint32_t loc[8], run = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { loc[k] = run; run += hist[tid * 8u + (uint32_t)k]; }
    part[tid] = run;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const uint32_t base = part[tid] - run;
#pragma unroll
    for (int k = 0; k < 8; ++k) { hist[tid * 8u + (uint32_t)k] = base + loc[k]; start[tid * 8u + (uint32_t)k] = base + loc[k]; }          // cursors
    __syncthreads();
    for (uint32_t r = tid; r < n; r += 1024u) tmp[atomicAdd(&hist[(__float_as_uint(cscale[r]) >> 18) & 8191u], 1u)] = r;
    __syncthreads();
    for (uint32_t r = tid; r < n; r += 1024u) { const uint32_t cls = (__float_as_uint(cscale[r]) >> 18) & 8191u; seg[r] = make_uint2(start[cls], hist[cls]); }
    for (uint32_t r = n + tid; r < n_pad; r += 1024u) cperm[r] = kNone;
}
__global__ __launch_bounds__(256)
void stage_counts_rank_kernel(uint32_t n, uint32_t n_pad, const uint32_t* __restrict__ scratch, uint32_t* __restrict__ cperm)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t* __restrict__ tmp = scratch;
    const uint2 sg = reinterpret_cast<const uint2*>(scratch + n_pad)[r];
    uint32_t rank = 0;
    for (uint32_t q = sg.x; q < sg.y; ++q) rank += tmp[q] < r ? 1u : 0u;
    cperm[sg.x + rank] = r;
}

// (the quad summaries of the ordered tiles live behind the row lines in the same allocation: r3dm_internal.hpp counts_summary_offset)
// one workgroup per tile of the ORDERED image: gather the rows cperm names from the keypoint-order count tiles, write their fragments
// and the tile's 256-byte row line (||a||^2 of the 32 rows, then their negated scales)
__global__ __launch_bounds__(256)
void stage_counts_gather_kernel(const uint16_t* __restrict__ tiledc, const float* __restrict__ cscale, const float* __restrict__ norms,
                                const uint32_t* __restrict__ cperm, uint32_t GB, uint16_t* __restrict__ tiledp, float* __restrict__ crow,
                                float* __restrict__ csum)
{
    const uint32_t t = blockIdx.x;
    __shared__ uint32_t src[32];
    if (threadIdx.x < 32u) src[threadIdx.x] = cperm[t * 32u + threadIdx.x];
    __syncthreads();
    uint16_t* dst = tiledp + (size_t)t * GB * 512;
    for (uint32_t e = threadIdx.x; e < GB * 512u; e += 256u) {
        const uint32_t c8 = e & 7u, r = (e >> 3) & 31u, hh = (e >> 8) & 1u, kb = e >> 9;
        const uint32_t sr = src[r];
        dst[e] = sr == kNone ? (uint16_t)0 : tiledc[(size_t)(sr >> 5) * GB * 512 + kb * 512u + (hh * 32u + (sr & 31u)) * 8u + c8];
    }
    if (threadIdx.x < 64u) {
        const uint32_t sr = src[threadIdx.x & 31u];
        crow[(size_t)t * 64u + threadIdx.x] = threadIdx.x < 32u ? (sr == kNone ? R3DM_INF : norms[sr]) : -(sr == kNone ? 1.0f : cscale[sr]);
    }
    // the sixteen numbers l2_knn2_counts2_kernel tests a tile's keys with: min ||a||^2 ([m]) and max scale ([8 + m]) of rows 4 m .. 4 m + 3
    if (threadIdx.x < 16u && csum) {
        const uint32_t m = threadIdx.x & 7u;
        float v = threadIdx.x < 8u ? R3DM_INF : 0.0f;
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t sr = src[4u * m + k];
            if (threadIdx.x < 8u) v = fminf(v, sr == kNone ? R3DM_INF : norms[sr]);
            else v = fmaxf(v, sr == kNone ? 1.0f : cscale[sr]);
        }
        csum[(size_t)t * 16u + threadIdx.x] = v;
        if (t == 0) csum[(size_t)gridDim.x * 16u + threadIdx.x] = threadIdx.x < 8u ? R3DM_INF : 1.0f;      // the line of "the tile before the first"
    }
}

hipError_t launch_stage_counts(hipStream_t st, const float* rows, uint32_t n, uint32_t dim, uint32_t GB, uint32_t n_tiles,
                               uint16_t* tiledc, float* cscale, const float* norms, uint16_t* tiledp, float* crow, uint32_t* cperm,
                               uint32_t* fail_dev)
{
    if (n_tiles == 0 || dim > 256u) return hipSuccess;
    hipLaunchKernelGGL(stage_counts_kernel, dim3(n_tiles), dim3(256), 0, st, rows, n, dim, GB, tiledc, cscale, fail_dev);
    // (the ordered tiles are written by the gather kernel behind these two: until then their first 3 n_pad words -- 12 of the >= 128 bytes a
    // row has there -- are the order kernels' scratch)
    hipLaunchKernelGGL(stage_counts_order_kernel, dim3(1), dim3(1024), 0, st, cscale, n, n_tiles * 32u, cperm, reinterpret_cast<uint32_t*>(tiledp));
    hipLaunchKernelGGL(stage_counts_rank_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, n, n_tiles * 32u, reinterpret_cast<const uint32_t*>(tiledp), cperm);
    hipLaunchKernelGGL(stage_counts_gather_kernel, dim3(n_tiles), dim3(256), 0, st, tiledc, cscale, norms, cperm, GB, tiledp, crow,
                       crow + counts_summary_offset(n_tiles));
    return hipGetLastError();
}

// Per tile the wave loads ONE 256-byte line beside the nine fragment loads: lane l < 32 holds ||a||^2 of row l, lane 32 + l the negated
// scale of row l.  A min over aligned groups of four lanes (two DPP steps) turns that into the quad summaries -- min ||a||^2 and
// -max scale of rows 4 g .. 4 g + 3 in every lane of group g -- and a lane picks the eight numbers of its own accumulator quads
// (rows 8 qd + 4 h + k: group 2 qd + h) with v_readlane + v_cndmask.  The per-row values are only looked at, through the lane crossbar,
// for a quad that passes the test.  (Measured dead ends: both arrays as 16 values per lane -- eight more 1-KiB wave loads per tile:
// 91 ms on the stage's 276 pairs; summaries read through `h ? p[a] : p[b]` -- the compiler selects the ADDRESS and emits flat loads
// with a vmcnt(0) behind them that drains the fragment prefetch: 126 ms.)
__device__ __forceinline__ float quad_min4(float v)
{
    // min over the aligned group of four lanes: xor-1 then xor-2 inside the DPP quad
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = __builtin_fminf(v, a);
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return __builtin_fminf(v, b);
}

template <int GB, int NJ, int PF>
__device__ __forceinline__ void counts_tile_step(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rr, uint32_t voffA, uint32_t voffR,
                                                 uint32_t soffA, uint32_t soffR, f32x4 (&abuf)[PF], uint32_t h,
                                                 float& rowv_load, const float (&qs_prev)[8], float rowv_prev,
                                                 const f32x4 (&bq)[NJ][GB], const float (&cq)[NJ], f32x16 (&cur)[NJ], const f32x16 (&prev)[NJ],
                                                 Top2 (&st)[NJ], uint32_t prev_rowbase)
{
    constexpr int NG = 4 * NJ;                             // (list, quad) groups of four keys per tile
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f32x4 a = abuf[g % PF];
        abuf[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 1024u);
        if (g == (GB > 2 ? 2 : GB - 1))                    // THIS tile's row values (its keys are tested in the next step)
            rowv_load = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)voffR, (int)soffR, 0));
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bq[nj][g]),
                                                             g == 0 ? zero : cur[nj], 0, 0, 0);
#pragma unroll
        for (int gi = (g * NG) / GB; gi < ((g + 1) * NG) / GB; ++gi) {
            const int nj = gi % NJ, qd = gi / NJ;
            const float p0 = prev[nj][4 * qd], p1 = prev[nj][4 * qd + 1], p2 = prev[nj][4 * qd + 2], p3 = prev[nj][4 * qd + 3];
            // lower bound of the quad's four keys (padding rows: ||a||^2 = +inf, count 0 -> key +inf, never below a bound)
            const float pmin = g == 0 ? __builtin_fminf(__builtin_fminf(p0, p1), __builtin_fminf(p2, p3)) : vmin2(vmin3(p0, p1, p2), p3);
            const float lb = __builtin_fmaf(qs_prev[4 + qd], cq[nj], pmin * qs_prev[qd]);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(lb < st[nj].d2) != 0ull, 0)) {
                const uint32_t rb = prev_rowbase + 8u * (uint32_t)qd;       // wave-uniform: the lists hold rows WITHOUT the lane half's + 4 h (added once, at the end)
                const uint32_t r0 = 8u * (uint32_t)qd + 4u * h;               // the quad's first row within its tile
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float n2 = __shfl(rowv_prev, (int)(r0 + (uint32_t)k)), sa = -__shfl(rowv_prev, (int)(32u + r0 + (uint32_t)k));
                    const float pk = k == 0 ? p0 : (k == 1 ? p1 : (k == 2 ? p2 : p3));
                    top2_push(st[nj], __builtin_fmaf(n2, cq[nj], pk * sa), rb + (uint32_t)k);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the eight summary numbers of a lane's accumulator quads from the tile's row line: [qd] = max scale, [4 + qd] = min ||a||^2
// (lane exchanges, ds_bpermute with the lane half folded into the address: 8 LDS instructions per tile where sixteen v_readlane +
// sixteen v_mov + eight v_cndmask stood -- the loop is bound by the VALU instructions it issues between the MFMAs)
__device__ __forceinline__ void counts_quad_summaries(float rowv, uint32_t h, float (&qs)[8])
{
    const int g = __builtin_bit_cast(int, quad_min4(rowv));
    const int a0 = (int)(16u * h);                         // byte address of lane 4 h
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        qs[4 + qd] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a0 + 32 * qd, g));
        qs[qd] = -__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a0 + 128 + 32 * qd, g));
    }
}

template <int GB, int NJ, int PF>
__global__ __launch_bounds__(256, 2)
void l2_knn2_counts_kernel(const MatchParams P)
{
    static_assert(GB % PF == 0, "prefetch window must divide the block count");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
    uint32_t pair, qb;
    if (P.xcd_map) {                                       // pair p on XCD p % 8 (see l2_knn2_mfma_kernel)
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
    const uint32_t nI = Ip->n, ntI = Ip->n_tiles, ntJ = Jp->n_tiles, nJ = Jp->n;
    const uint32_t qt0 = (qb * 4u + wave) * NJ;
    if (qt0 >= ntJ) return;                                // wave-uniform; no barriers in this kernel

    // ---- query fragments (B operand): the NEGATED counts (a sign flip of an f16 integer), and per query 1 / (2 s_q), 2 s_q
    f32x4 bq[NJ][GB];
    float cq[NJ], kinv[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;
        const gf4p src = (gf4p)(const void*)Jp->tiledc + (size_t)qt * (GB * 64) + lane;
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            u32x4 w = __builtin_bit_cast(u32x4, src[g * 64]);
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] ^= 0x80008000u;
            bq[nj][g] = __builtin_bit_cast(f32x4, w);
        }
        const uint32_t q = qt * 32u + c;
        const float sq = q < nJ ? Jp->cscale[q] : 1.0f;
        kinv[nj] = 2.0f * sq;
        cq[nj] = 1.0f / kinv[nj];
    }
    Top2 st[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) top2_init(st[nj]);

    if (nI >= 2) {
        const uint64_t pa = (uint64_t)Ip->tiledp;         // rows in the order of their scales (stage_counts_order_kernel)
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffA = lane * 16u;
        constexpr uint32_t tileB = (uint32_t)GB * 1024u;
        const uint32_t hb = 4u * h;
        const uint64_t prw = (uint64_t)Ip->cquad;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(prw >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)prw)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffR = lane * 4u;
        f32x4 abuf[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) abuf[s] = bload16(ra, voffA, (uint32_t)s * 1024u);
        // the row line of tile t is loaded in step t and used in step t + 1 (the keys of tile t are tested while tile t + 1 is multiplied)
        float rvA = 0.0f, rvB = h ? -1.0f : R3DM_INF;      // "tile -1": ||a||^2 = +inf keeps it out of every list
        float qsA[8], qsB[8];
        f32x16 accA[NJ], accB[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[nj][r] = 0.0f;
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            counts_quad_summaries(rvB, h, qsB);
            counts_tile_step<GB, NJ, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, h, rvA, qsB, rvB, bq, cq, accA, accB, st, (t - 1) * 32u);
            counts_quad_summaries(rvA, h, qsA);
            counts_tile_step<GB, NJ, PF>(ra, rr, voffA, voffR, (t + 1) * tileB + PF * 1024u, (t + 1) * 256u, abuf, h, rvB, qsA, rvA, bq, cq, accB, accA, st, t * 32u);
        }
        if (t < ntI) {
            counts_quad_summaries(rvB, h, qsB);
            counts_tile_step<GB, NJ, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, h, rvA, qsB, rvB, bq, cq, accA, accB, st, (t - 1) * 32u);
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (uint32_t)((r & 3) + 8 * (r >> 2));
                    top2_push(st[nj], __builtin_fmaf(__shfl(rvA, (int)(row + hb)), cq[nj], accA[nj][r] * -__shfl(rvA, (int)(32u + row + hb))), t * 32u + row);
                }
        } else {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = (uint32_t)((r & 3) + 8 * (r >> 2));
                    top2_push(st[nj], __builtin_fmaf(__shfl(rvB, (int)(row + hb)), cq[nj], accB[nj][r] * -__shfl(rvB, (int)(32u + row + hb))), (ntI - 1) * 32u + row);
                }
        }
    }
    // the lists name rows of the ordered image: back to keypoint order before the tail re-scores and certifies them
    {
        const uint32_t* __restrict__ perm = Ip->cperm;
        const uint32_t hb2 = 4u * h;                       // the lane half's rows: + 4 within every group of eight
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            if (st[nj].i0 != kNone) st[nj].i0 = perm[st[nj].i0 + hb2];
            if (st[nj].i1 != kNone) st[nj].i1 = perm[st[nj].i1 + hb2];
        }
    }
    l2_finish_queries<NJ, false, true>(P, pair, Ip, Jp, st, qt0, h, c, (float)(GB * 16), false, 1.0f, 0.0f, kinv);
}

// ------------------------------------------------------------------------------------------------
// Count tiles, ONE list per query (round 4, third form; NJ = 2 only).  In the 32 x 32 accumulator layout lane (c, h) holds rows
// 8 j + 4 h + i of query column c, so l2_knn2_counts_kernel keeps TWO lists per query (one per lane half) and per query tile -- 128
// lists per wave, and the wave-wide test-and-skip takes its slow path whenever any of them can change: ~3.3 of the 8 quad tests of
// a tile step (PMC: 238 VALU instructions per step where the fast path has 47).  v_permlane32_swap_b32 exchanges the upper lane half
// of one register with the lower half of another (tools/ubench/permlane_probe.hip): swapping the accumulators of the two query tiles
// register by register leaves lane c < 32 with ALL 32 rows of query (tile 0, c) and lane 32 + c with all rows of (tile 1, c) --
// 64 lists per wave, each over all rows, so half as many list changes -- and makes every row quantity of a quad wave-uniform (scalar
// operands from v_readlane instead of per-lane exchanges).  The keys, and so the results, are those of l2_knn2_counts_kernel.
//
// With one list per query the shared tail's second chance would have two nominees where the two half-lists gave it four (measured:
// 4 x the queries in the exact scan).  So the list here also carries the THIRD-best row and a lower bound d3 of every key that is
// none of the three (Top3m below): the tail re-scores three rows and certifies against d3 -- a handful of exact scans where the
// two-list kernel needs hundreds (reference-built LIOP fixture: 6 of 8,192 queries against 45; 80 views x 8,192 rows: 22 of 25.9 M
// against 850), and the kernel itself is 5 % faster (13 % without the third row's bookkeeping).  R3DM_COUNTS_TWO_LISTS=1 in the
// developer build runs l2_knn2_counts_kernel instead (tools/counts_one_list_probe.py).  (hipcc 7.2 folds repeated
// __builtin_amdgcn_permlane32_swap calls with different operands into one -- wrong code -- hence the inline assembly with its own
// wait states below.)
// ------------------------------------------------------------------------------------------------
// the list of a query in this kernel: the two nominees and the third-best key (d2: the bound of the first certification, as in Top2)
// PLUS the third-best row (i2) and a lower bound of every key that is none of the three (d3): the tail's second chance re-scores
// three rows and certifies against d3.  d3 = min over (a) the keys pushed out of, or never into, the three -- exactly -- and (b) the
// lower bounds of the quads that were skipped (>= d2 at the time, so >= every d2 since).
struct Top3m { float d0, d1, d2, d3; uint32_t i0, i1, i2; };
__device__ __forceinline__ void top3m_init(Top3m& s) { s.d0 = s.d1 = s.d2 = s.d3 = R3DM_INF; s.i0 = s.i1 = s.i2 = kNone; }
__device__ __forceinline__ void top3m_push(Top3m& s, float key, uint32_t idx)
{
    const float od0 = s.d0, od1 = s.d1, od2 = s.d2, od3 = s.d3;
    const uint32_t oi0 = s.i0, oi1 = s.i1, oi2 = s.i2;
    const bool c0 = key < od0, c1 = key < od1, c2 = key < od2;
    s.d3 = __builtin_amdgcn_fmed3f(od2, od3, key);         // min(d3, max(d2, key)): what falls out of the three (d2 <= d3 always)
    s.d2 = __builtin_amdgcn_fmed3f(od1, od2, key);
    s.d1 = __builtin_amdgcn_fmed3f(od0, od1, key);
    s.d0 = __builtin_amdgcn_fmed3f(-R3DM_INF, od0, key);
    const uint32_t t2 = c2 ? idx : oi2, t1 = c1 ? idx : oi1;
    s.i2 = c1 ? oi1 : t2;
    s.i1 = c0 ? oi0 : t1;
    s.i0 = c0 ? idx : oi0;
}

typedef const __attribute__((address_space(4))) float* cf32p;         // constant address space -> SMEM loads (a tile's sixteen quad summaries)

// v_permlane32_swap_b32: the upper lane half of `a` <-> the lower lane half of `b`   (a' = [a.lo | b.lo], b' = [a.hi | b.hi])
// (s_nop 1 first: the instruction needs two wait states behind a VALU write of either operand -- the compiler inserts them for its
//  own builtin and cannot for an asm statement)
__device__ __forceinline__ void swap_lane_halves(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

template <int GB, int PF>
__device__ __forceinline__ void counts_tile_step_m(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rr, uint32_t voffA, uint32_t voffR,
                                                   uint32_t soffA, uint32_t soffR, f32x4 (&abuf)[PF], float& rowv_load, float rowv_prev, cf32p sum_prev,
                                                   const f32x4 (&bq)[2][GB], float cql, f32x16 (&cur)[2], f32x16 (&prev)[2], Top3m& st, uint32_t prev_rowbase)
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int rv = __builtin_bit_cast(int, rowv_prev);
    float sm[16];                                          // one s_load_dwordx16 at the top of the step: in flight behind the first MFMAs
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[k] = sum_prev[k];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f32x4 a = abuf[g % PF];
        abuf[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 1024u);
        if (g == (GB > 2 ? 2 : GB - 1))
            rowv_load = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)voffR, (int)soffR, 0));
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
            cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bq[nj][g]),
                                                             g == 0 ? zero : cur[nj], 0, 0, 0);
        // eight quad tests per tile (slot = 2 qd + hp: rows 8 qd + 4 hp + k), spread over the GB blocks
#pragma unroll
        for (int gi = (g * 8) / GB; gi < ((g + 1) * 8) / GB; ++gi) {
            const int qd = gi >> 1, hp = gi & 1;
            if (hp == 0) {                                  // the quad's four registers of both query tiles: one list per lane from here on
                // (inline assembly: hipcc 7.2 folds several __builtin_amdgcn_permlane32_swap calls into one -- tools/ubench/permlane_probe.hip;
                //  the wait states an MFMA result needs before a VALU reads it are the compiler's to insert, and it cannot see into the asm)
                if (gi == 0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float lo = prev[0][4 * qd + k], hi = prev[1][4 * qd + k];
                    swap_lane_halves(lo, hi);
                    prev[0][4 * qd + k] = lo;               // rows 8 qd + k      (lane half 0 of both tiles)
                    prev[1][4 * qd + k] = hi;               // rows 8 qd + 4 + k  (lane half 1 of both tiles)
                }
            }
            const float p0 = prev[hp][4 * qd], p1 = prev[hp][4 * qd + 1], p2 = prev[hp][4 * qd + 2], p3 = prev[hp][4 * qd + 3];
            const float pmin = vmin2(vmin3(p0, p1, p2), p3);
            const int r0 = 8 * qd + 4 * hp;
            const float n2min = sm[2 * qd + hp], smax = sm[8 + 2 * qd + hp];      // scalars (SMEM): rows r0 .. r0 + 3
            const float lb = __builtin_fmaf(n2min, cql, pmin * smax);
            const bool mine = lb < st.d2;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine) != 0ull, 0)) {
                const uint32_t rb = prev_rowbase + (uint32_t)r0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float n2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rv, r0 + k));
                    const float sa = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(rv, 32 + r0 + k));
                    const float pk = k == 0 ? p0 : (k == 1 ? p1 : (k == 2 ? p2 : p3));
                    top3m_push(st, __builtin_fmaf(n2, cql, pk * sa), rb + (uint32_t)k);
                }
            }
            // a lane whose own bound did not pass: its four keys are >= lb >= d2 (then and since), whether or not the wave pushed them
            st.d3 = vmin2(st.d3, mine ? R3DM_INF : lb);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the keys of the last tile (nothing multiplies behind it): all 32 rows of the lane's query
__device__ __forceinline__ void counts_last_tile_m(f32x16 (&acc)[2], float rowv, float cql, Top3m& st, uint32_t rowbase)
{
    const int rvi = __builtin_bit_cast(int, rowv);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float lo = acc[0][r], hi = acc[1][r];
        swap_lane_halves(lo, hi);
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hp;
            const float n2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rvi, row));
            const float sa = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(rvi, 32 + row));
            top3m_push(st, __builtin_fmaf(n2, cql, (hp ? hi : lo) * sa), rowbase + (uint32_t)row);
        }
    }
}

template <int GB, int PF>
__global__ __launch_bounds__(256, 2)
void l2_knn2_counts2_kernel(const MatchParams P)
{
    static_assert(GB % PF == 0, "prefetch window must divide the block count");
    constexpr int NJ = 2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
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
    const uint32_t nI = Ip->n, ntI = Ip->n_tiles, ntJ = Jp->n_tiles, nJ = Jp->n;
    const uint32_t qt0 = (qb * 4u + wave) * NJ;
    if (qt0 >= ntJ) return;                                // wave-uniform; no barriers in this kernel

    f32x4 bq[NJ][GB];
    float cq[NJ], kinv[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;
        const gf4p src = (gf4p)(const void*)Jp->tiledc + (size_t)qt * (GB * 64) + lane;
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            u32x4 w = __builtin_bit_cast(u32x4, src[g * 64]);
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] ^= 0x80008000u;
            bq[nj][g] = __builtin_bit_cast(f32x4, w);
        }
        const uint32_t q = qt * 32u + c;
        const float sq = q < nJ ? Jp->cscale[q] : 1.0f;
        kinv[nj] = 2.0f * sq;
        cq[nj] = 1.0f / kinv[nj];
    }
    const float cql = h ? cq[1] : cq[0];                   // this lane's query after the swap: (tile h, column c)
    Top3m st;
    top3m_init(st);

    if (nI >= 2) {
        const uint64_t pa = (uint64_t)Ip->tiledp;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffA = lane * 16u;
        constexpr uint32_t tileB = (uint32_t)GB * 1024u;
        const uint64_t prw = (uint64_t)Ip->cquad;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(prw >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)prw)),
            0, 0x7FFFFFFF, 0x00020000);
        const uint32_t voffR = lane * 4u;
        // the quad summaries of tile t: sixteen floats behind the row lines; "tile -1" reads the line one past the last tile (+inf, 1)
        const cf32p sums = (cf32p)(uintptr_t)(Ip->cquad + counts_summary_offset(ntI));
        f32x4 abuf[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) abuf[s] = bload16(ra, voffA, (uint32_t)s * 1024u);
        float rvA = 0.0f, rvB = h ? -1.0f : R3DM_INF;      // "tile -1": ||a||^2 = +inf keeps it out of every list
        f32x16 accA[NJ], accB[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[nj][r] = 0.0f;
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            counts_tile_step_m<GB, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, rvA, rvB, sums + (size_t)(t == 0 ? ntI : t - 1) * 16u, bq, cql, accA, accB, st, (t - 1) * 32u);
            counts_tile_step_m<GB, PF>(ra, rr, voffA, voffR, (t + 1) * tileB + PF * 1024u, (t + 1) * 256u, abuf, rvB, rvA, sums + (size_t)t * 16u, bq, cql, accB, accA, st, t * 32u);
        }
        // the last tile's keys (and one more multiply step when the tile count is odd)
        if (t < ntI) {
            counts_tile_step_m<GB, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, rvA, rvB, sums + (size_t)(t == 0 ? ntI : t - 1) * 16u, bq, cql, accA, accB, st, (t - 1) * 32u);
            counts_last_tile_m(accA, rvA, cql, st, t * 32u);
        } else {
            counts_last_tile_m(accB, rvB, cql, st, (ntI - 1) * 32u);
        }
    }
    // the list names rows of the ordered image: back to keypoint order; then hand it to the shared tail in the layout it expects (a
    // list per lane half and query tile).  This lane's half carries the two nominees with the bound d3; the other half's slot carries
    // the third-best row as a one-row list with the same bound: the tail's merge then sees the third-best key as the smallest
    // un-nominated one (first certification, as before), and its second chance re-scores the three rows against d3.
    {
        const uint32_t* __restrict__ perm = Ip->cperm;
        if (st.i0 != kNone) st.i0 = perm[st.i0];
        if (st.i1 != kNone) st.i1 = perm[st.i1];
        if (st.i2 != kNone) st.i2 = perm[st.i2];
    }
    Top2 st2[NJ];
    {
        // the partner lane (c, 1 - h) holds the list of the OTHER query tile: fetch what it has for my tile's partner slot
        const float pd2 = __shfl_xor(st.d2, 32), pd3 = __shfl_xor(st.d3, 32);
        const uint32_t pi2 = __shfl_xor(st.i2, 32);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            if ((uint32_t)nj == h) { st2[nj].d0 = st.d0; st2[nj].d1 = st.d1; st2[nj].d2 = st.i2 != kNone ? st.d3 : st.d2; st2[nj].i0 = st.i0; st2[nj].i1 = st.i1; }
            else { st2[nj].d0 = pd2; st2[nj].d1 = R3DM_INF; st2[nj].d2 = pd3; st2[nj].i0 = pi2; st2[nj].i1 = kNone; if (pi2 == kNone) { st2[nj].d0 = R3DM_INF; st2[nj].d2 = pd2; } }
        }
    }
    l2_finish_queries<NJ, false, true>(P, pair, Ip, Jp, st2, qt0, h, c, (float)(GB * 16), false, 1.0f, 0.0f, kinv);
}

template <int GB, int PF>
static hipError_t launch_l2_counts2_t(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * 2u;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    P.xcd_map = 1u;
    const uint64_t grid64 = (uint64_t)((P.n_pairs + 7u) / 8u * 8u) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((l2_knn2_counts2_kernel<GB, PF>), dim3((uint32_t)grid64), dim3(256), 0, st, P);
    return hipGetLastError();
}
template <int GB, int NJ, int PF>
static hipError_t launch_l2_counts_t(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    P.xcd_map = 1u;
    const uint64_t grid64 = (uint64_t)((P.n_pairs + 7u) / 8u * 8u) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((l2_knn2_counts_kernel<GB, NJ, PF>), dim3((uint32_t)grid64), dim3(256), 0, st, P);
    return hipGetLastError();
}

// G = padded dim / 8 of the views (8, 16, 18, 32); hipErrorInvalidValue -> no count kernel, caller keeps the split tiles
hipError_t launch_l2_knn2_counts(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles, int variant)
{
    switch (G) {
        // two query tiles per wave: one list per query (l2_knn2_counts2_kernel); variant != 0 (developer build): the two-list kernel
        case 8:  return variant ? launch_l2_counts_t<4, 2, 4>(st, P, max_nj_tiles) : launch_l2_counts2_t<4, 4>(st, P, max_nj_tiles);
        case 16: return variant ? launch_l2_counts_t<8, 2, 8>(st, P, max_nj_tiles) : launch_l2_counts2_t<8, 8>(st, P, max_nj_tiles);
        case 18: return variant ? launch_l2_counts_t<9, 2, 9>(st, P, max_nj_tiles) : launch_l2_counts2_t<9, 9>(st, P, max_nj_tiles);
        // (256 dimensions stay on the two-list kernel, one query tile per wave: l2_knn2_counts2_kernel<16, 8> -- 128 registers of query
        //  fragments -- compiles with its fragment array indexed through scratch memory, 528 bytes per lane, round 5)
        case 32: return launch_l2_counts_t<16, 1, 8>(st, P, max_nj_tiles);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// The same integer fast path with the dataset tiles SHARED by the four waves of a workgroup.  PMC on l2_knn2_int_kernel
// (profiles/r01_pmc_int_kernel.txt): the L1 address path is 0.92 busy -- every wave pulls every 1 KiB dataset fragment
// through the texture addresser itself, one 64-lane x 16 B buffer_load per two 32-cycle MFMAs -- while the matrix pipe is
// 0.55 busy.  Here a tile (GB KiB) is fetched ONCE per workgroup, straight into LDS (global_load_lds_dwordx4: no VGPR round
// trip, no ds_write; the fragment-ordered image is lane-linear, which is exactly what the LDS-DMA writes), each wave issuing
// GB/4 of its blocks plus its own copy of the tile's 32 norms; all four waves then read the fragments with ds_read_b128
// (conflict-free: lane-linear 16 B).  Three LDS buffers, one barrier per tile:
//     step t:  s_waitcnt vmcnt(0)      this wave's loads of tiles t and t+1 have landed
//              s_barrier               ... everybody's have, and everybody has finished reading tile t-1
//              issue the loads of tile t+2 into the buffer tile t-1 occupied
//              MFMAs of tile t (fragments through a PF-deep register window that runs on into tile t+1),
//              list updates of tile t-1 in their shadow (same lean lexicographic epilogue as l2_knn2_int_kernel)
// (ordering rules of LDS-DMA: cdna_hip_programming.md -- data is ordered for a ds_read only by the issuing wave's vmcnt
// wait followed by a barrier the reader has passed; all LDS in ONE array; no VGPR-destination loads inside the loop.)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// OPS 0: bf16 operands (16 dims per block, v_mfma_f32_32x32x16_bf16).  OPS 1: i8 operands holding the BITS of binary
// descriptors as 0 / 1 (32 bits per block, v_mfma_i32_32x32x32_i8): with C = popcount(a) and B = -2 b the accumulator is
// popcount(a) - 2 a.b = Hamming(a, b) - popcount(b), an exact integer.  The accumulators are biased by 0x3F800000 (the bits of
// 1.0f) through the C operand: the int32 key k and the float with the bits k + 0x3F800000 order identically (normal positive
// floats, |k| <= 2048 steps of one ulp), so the float list machinery below runs on them unchanged.
template <int GB, int NJ, int PF, int ABL, int OPS = 0>
__device__ __forceinline__ void int_tile_step_lds(const unsigned char* __restrict__ lds_cur, const unsigned char* __restrict__ lds_nxt,
                                                  const unsigned char* __restrict__ nrm_nxt, f32x4 (&abuf)[PF], const f32x16& nrm_cur,
                                                  f32x16& nrm_next, const f32x4 (&bq)[NJ][GB], f32x16 (&cur)[NJ], const f32x16 (&prev)[NJ],
                                                  Top2 (&st)[NJ], uint32_t prev_rowbase)
{
    constexpr int NG = 4 * NJ;
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f32x4 a = abuf[g % PF];
        // block g + PF of the tile stream: this tile's, or the first blocks of the next one (landed with this step's barrier)
        abuf[g % PF] = (g + PF < GB) ? *reinterpret_cast<const f32x4*>(lds_cur + (g + PF) * 1024)
                                     : *reinterpret_cast<const f32x4*>(lds_nxt + (g + PF - GB) * 1024);
        if (g == (GB > 2 ? 2 : GB - 1)) {   // next tile's norms, element 4 qd + k = row 8 qd + 4 h + k: the accumulator layout
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(nrm_nxt + qd * 32);
#pragma unroll
                for (int k = 0; k < 4; ++k) nrm_next[4 * qd + k] = v[k];
            }
        }
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            if constexpr (OPS == 0)
                cur[nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bq[nj][g]),
                                                                  g == 0 ? nrm_cur : cur[nj], 0, 0, 0);
            else
                cur[nj] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(
                              __builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, bq[nj][g]),
                              __builtin_bit_cast(i32x16, g == 0 ? nrm_cur : cur[nj]), 0, 0, 0));
        }
#pragma unroll
        for (int gi = (g * NG) / GB; gi < ((g + 1) * NG) / GB; ++gi) {
            const int nj = gi % NJ, qd = gi / NJ;
            const float p0 = prev[nj][4 * qd], p1 = prev[nj][4 * qd + 1], p2 = prev[nj][4 * qd + 2], p3 = prev[nj][4 * qd + 3];
            if constexpr (ABL != 0) {
                asm volatile("" ::"v"(p0), "v"(p1), "v"(p2), "v"(p3));
            } else {
                const float m = g == 0 ? __builtin_fminf(__builtin_fminf(p0, p1), __builtin_fminf(p2, p3)) : vmin2(vmin3(p0, p1, p2), p3);
                if (__builtin_amdgcn_ballot_w64(m < st[nj].d1) != 0ull) {
                    const uint32_t rb = prev_rowbase + 8u * (uint32_t)qd;
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p0 < st[nj].d1) != 0ull) tope_push(st[nj], p0, rb);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p1 < st[nj].d1) != 0ull) tope_push(st[nj], p1, rb + 1u);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p2 < st[nj].d1) != 0ull) tope_push(st[nj], p2, rb + 2u);
                    if (GB > 8 || __builtin_amdgcn_ballot_w64(p3 < st[nj].d1) != 0ull) tope_push(st[nj], p3, rb + 3u);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// tail of the OPS 1 kernel: exact lexicographic (key, row) lists of the two lane halves -> Hamming distances, ratio test on the
// float-converted distances (NNdistanceRatio on Hamming<unsigned char>::ResultType, as hamming_knn2_kernel does)
constexpr uint32_t kHamBias = 0x3F800000u;
template <int NJ>
__device__ __forceinline__ void hamming_finish_queries(const MatchParams& P, uint32_t pair, const ImgDev* __restrict__ Ip,
                                                       const ImgDev* __restrict__ Jp, const Top2 (&st)[NJ], uint32_t qt0, uint32_t h, uint32_t c)
{
    const uint32_t nI = Ip->n, nJ = Jp->n, ntJ = Jp->n_tiles;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        Top2 s = st[nj];
        const float pd0 = __shfl_xor(s.d0, 32), pd1 = __shfl_xor(s.d1, 32);
        const uint32_t pi0 = __shfl_xor(s.i0, 32), pi1 = __shfl_xor(s.i1, 32);
        lex_push(s, pd0, pi0);
        lex_push(s, pd1, pi1);
        const uint32_t qt = qt0 + nj, q = qt * 32u + c;
        if (!(qt < ntJ && q < nJ) || h != 0) continue;
        const size_t o = (size_t)pair * P.q_stride + q;
        if (nI < 2 || s.i1 == kNone) {
            P.nn_idx[o] = kNone;
            if (P.knn_idx) { P.knn_idx[2 * o] = -1; P.knn_idx[2 * o + 1] = -1; P.knn_dist[2 * o] = R3DM_INF; P.knn_dist[2 * o + 1] = R3DM_INF; }
            continue;
        }
        const int pq = (int)(__float_as_uint(Jp->norms[q]) - kHamBias);                 // popcount of the query row
        const uint32_t d0 = (uint32_t)((int)(__float_as_uint(s.d0) - kHamBias) + pq);
        const uint32_t d1 = (uint32_t)((int)(__float_as_uint(s.d1) - kHamBias) + pq);
        P.nn_idx[o] = ((float)d0 < P.ratio_R * (float)d1) ? s.i0 : kNone;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = (int32_t)s.i0; P.knn_idx[2 * o + 1] = (int32_t)s.i1;
            P.knn_dist[2 * o] = (float)d0;    P.knn_dist[2 * o + 1] = (float)d1;
        }
    }
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0, int OPS = 0>
__global__ __launch_bounds__(256, WPS)
void l2_knn2_int_lds_kernel(const MatchParams P)
{
    static_assert(GB % 4 == 0 && PF <= GB, "a tile is dealt to four waves in whole 1 KiB blocks");
    // ONE LDS array: [3 buffers][GB KiB tile] then [3 buffers][4 waves][256 B norms]
    extern __shared__ __attribute__((aligned(16))) unsigned char int_smem[];
    constexpr uint32_t tileB = (uint32_t)GB * 1024u;
    constexpr uint32_t nrm0 = 3u * tileB;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t h = lane >> 5, c = lane & 31u;
    uint32_t pair, qb;
    if (P.xcd_map) {                                       // pair p on XCD p % 8 (see l2_knn2_mfma_kernel)
        const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        pair = (j / P.qb_per_pair) * 8u + xcd;
        qb = j % P.qb_per_pair;
        if (pair >= P.n_pairs) return;                     // whole workgroup
    } else {
        pair = blockIdx.x / P.qb_per_pair;
        qb = blockIdx.x % P.qb_per_pair;
    }
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nI = Ip->n, ntI = Ip->n_tiles, ntJ = Jp->n_tiles;
    const uint32_t qt0 = (qb * 4u + wave) * NJ;
    // a wave without query tiles still takes part in the loads and barriers of its workgroup; its results are discarded
    const bool has_queries = qt0 < ntJ;

    const void* tilesJ = OPS == 0 ? (const void*)Jp->tiled16 : (const void*)Jp->tiled8;
    const void* tilesI = OPS == 0 ? (const void*)Ip->tiled16 : (const void*)Ip->tiled8;
    f32x4 bq[NJ][GB];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;
        const gf4p src = (gf4p)tilesJ + (size_t)qt * (GB * 64) + lane;
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const u32x4 w = __builtin_bit_cast(u32x4, src[g * 64]);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (OPS == 0) {                  // -2 x (integer, |x| <= 256) is a bf16 again
                    const float lo = __uint_as_float(w[k] << 16) * -2.0f, hi = __uint_as_float(w[k] & 0xFFFF0000u) * -2.0f;
                    o[k] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xFFFF0000u);
                } else o[k] = w[k] * 0xFEu;                // bytes 0 / 1 -> 0 / -2 as i8 (no carries between bytes)
            }
            bq[nj][g] = __builtin_bit_cast(f32x4, o);
        }
    }
    Top2 st[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) top2_init(st[nj]);

    if (nI >= 2) {                                         // workgroup-uniform
        // per-lane global sources of this wave's share of a tile: blocks wave * GB/4 + i, and the tile's norms (row l & 31)
        const unsigned char* gA = reinterpret_cast<const unsigned char*>(tilesI) + (size_t)wave * (GB / 4) * 1024u + lane * 16u;
        const unsigned char* gN = reinterpret_cast<const unsigned char*>(Ip->norms) + (lane & 31u) * 4u;
        const uint32_t ldsA = wave * (GB / 4) * 1024u;     // + buffer * tileB + i * 1024   (the DMA adds lane * 16 itself)
        const uint32_t ldsN = nrm0 + wave * 256u;          // + buffer * 1024
        auto issue = [&](uint32_t tile, uint32_t buf) {
#pragma unroll
            for (int i = 0; i < GB / 4; ++i)
                __builtin_amdgcn_global_load_lds((glb_vp)(gA + (size_t)tile * tileB + i * 1024u), (lds_vp)(int_smem + buf * tileB + ldsA + i * 1024u), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gN + (size_t)tile * 128u), (lds_vp)(int_smem + buf * 1024u + ldsN), 4, 0, 0);
        };
        issue(0, 0);
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(2, 2);
        const unsigned char* lane_lds = int_smem + lane * 16u;            // fragment of block g of buffer b: + b * tileB + g * 1024
        const unsigned char* lane_nrm = int_smem + nrm0 + wave * 256u + h * 16u;   // quad qd of buffer b: + b * 1024 + qd * 32
        f32x4 abuf[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) abuf[s] = *reinterpret_cast<const f32x4*>(lane_lds + s * 1024);
        f32x16 nrmA, nrmB;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lane_nrm + qd * 32);
#pragma unroll
            for (int k = 0; k < 4; ++k) nrmA[4 * qd + k] = v[k];
        }
        f32x16 accA[NJ], accB[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[nj][r] = R3DM_INF;              // "tile -1": keys that never win
        const uint32_t hb = 4u * h;
        uint32_t bc = 0, bn = 1;                                             // buffers of tile t and tile t + 1
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            if (t != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(t + 2, bc == 0 ? 2u : bc - 1u);                        // the buffer tile t - 1 occupied
            }
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + bc * tileB, lane_lds + bn * tileB, lane_nrm + bn * 1024u, abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
            bc = bn; bn = bn == 2 ? 0u : bn + 1u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(t + 3, bc == 0 ? 2u : bc - 1u);
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + bc * tileB, lane_lds + bn * tileB, lane_nrm + bn * 1024u, abuf, nrmB, nrmA, bq, accB, accA, st, t * 32u + hb);
            bc = bn; bn = bn == 2 ? 0u : bn + 1u;
        }
        if (t < ntI) {
            if (t != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + bc * tileB, lane_lds + bn * tileB, lane_nrm + bn * 1024u, abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) tope_push(st[nj], accA[nj][r], t * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
        } else {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) tope_push(st[nj], accB[nj][r], (ntI - 1) * 32u + hb + (uint32_t)((r & 3) + 8 * (r >> 2)));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // drain the look-ahead loads before ordinary loads follow
    }
    if (has_queries) {
        if constexpr (OPS == 0) l2_finish_queries<NJ, true>(P, pair, Ip, Jp, st, qt0, h, c, (float)(GB * 16), true);
        else hamming_finish_queries<NJ>(P, pair, Ip, Jp, st, qt0, h, c);
    }
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0, int OPS = 0>
static hipError_t launch_l2_int_lds(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    P.xcd_map = 1u;
    const uint64_t grid64 = (uint64_t)((P.n_pairs + 7u) / 8u * 8u) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    const size_t lds = 3 * (size_t)GB * 1024 + 3 * 1024;
    hipLaunchKernelGGL((l2_knn2_int_lds_kernel<GB, NJ, PF, WPS, ABL, OPS>), dim3((uint32_t)grid64), dim3(256), lds, st, P);
    return hipGetLastError();
}

// Opt-in exact MFMA Hamming (r3dm_set_hamming_mfma): binary rows of `words` u32 staged as 0 / 1 bytes (ImgDev::tiled8) with
// biased popcounts in ImgDev::norms; 8 words = 256 bits = 8 blocks, 16 words = 512 bits = 16 blocks of 32
hipError_t launch_hamming_mfma(hipStream_t st, const MatchParams& P, uint32_t words, uint32_t max_nj_tiles)
{
    switch (words) {
        case 8:  return launch_l2_int_lds<8, 2, 4, 2, 0, 1>(st, P, max_nj_tiles);
        case 16: return launch_l2_int_lds<16, 2, 4, 2, 0, 1>(st, P, max_nj_tiles);
        default: return hipErrorInvalidValue;
    }
}

// binary rows -> i8 fragment tiles [tile][32-bit block kb][lane half h][32 rows][16 bytes] (lane half h of block kb holds bits
// 32 kb + 16 h .. + 15 of row 32 t + r, one bit per byte) + popcount(row) + kHamBias as the bits of a float (0x7F000000 for the
// padding rows: a key that never wins).  One workgroup per 32-row tile.
__global__ __launch_bounds__(256)
void stage_bin8_kernel(const uint32_t* __restrict__ bin, uint32_t n, uint32_t words, uint8_t* __restrict__ tiled8, float* __restrict__ norms)
{
    const uint32_t t = blockIdx.x;
    uint8_t* dst = tiled8 + (size_t)t * words * 1024;                    // words blocks of 1 KiB
    for (uint32_t e = threadIdx.x; e < words * 1024; e += 256) {
        const uint32_t c16 = e & 15, r = (e >> 4) & 31, h = (e >> 9) & 1, kb = e >> 10;
        const uint32_t row = t * 32 + r, bit = 16 * h + c16;
        dst[e] = (row < n) ? (uint8_t)((bin[(size_t)row * words + kb] >> bit) & 1u) : (uint8_t)0;
    }
    if (threadIdx.x < 32) {
        const uint32_t row = t * 32 + threadIdx.x;
        uint32_t v = 0x7F000000u;
        if (row < n) {
            uint32_t pc = 0;
            for (uint32_t w = 0; w < words; ++w) pc += (uint32_t)__builtin_popcount(bin[(size_t)row * words + w]);
            v = kHamBias + pc;
        }
        norms[(size_t)t * 32 + threadIdx.x] = __uint_as_float(v);
    }
}

hipError_t launch_stage_bin8(hipStream_t st, const uint32_t* bin, uint32_t n, uint32_t words, uint32_t n_tiles, uint8_t* tiled8, float* norms)
{
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_bin8_kernel, dim3(n_tiles), dim3(256), 0, st, bin, n, words, tiled8, norms);
    return hipGetLastError();
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0>
static hipError_t launch_l2_int(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    static const int xcd_map = r3dm_dev_knob("R3DM_XCD_MAP", 1);
    P.xcd_map = (uint32_t)xcd_map;
    const uint64_t grid64 = (uint64_t)(xcd_map ? (P.n_pairs + 7u) / 8u * 8u : P.n_pairs) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((l2_knn2_int_kernel<GB, NJ, PF, WPS, ABL>), dim3((uint32_t)grid64), dim3(256), 0, st, P);
    return hipGetLastError();
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
#ifdef R3DM_DEVTOOLS
        // R3DM_L2_INT_VARIANT (A/B measurements on 780 pairs of 8192 x 8192 rows, D = 128; the f32 tiles take 95.0 ms):
        //   2 = NJ 2 x 2 waves/SIMD (default, 12.4 ms) | 8 = same with a whole-tile prefetch window (12.95 ms) |
        //   4 = NJ 4 x 1 wave/SIMD (17.6 ms) | 9 = 2 without the epilogue (timing only, 11.4 ms)
        static const int iv = r3dm_dev_knob("R3DM_L2_INT_VARIANT", 2);
        if (G == 16 && iv == 5) return launch_l2_int_lds<8, 2, 4, 2>(st, P, max_nj_tiles);        // workgroup-shared tiles through LDS-DMA
        if (G == 16 && iv == 59) return launch_l2_int_lds<8, 2, 4, 2, 1>(st, P, max_nj_tiles);    // ... without the epilogue (timing only)
        if (G == 16 && iv == 6) return launch_l2_int_lds<8, 2, 8, 2>(st, P, max_nj_tiles);        // ... whole-tile fragment window
        if (G == 16 && iv == 4) return launch_l2_int<8, 4, 4, 1>(st, P, max_nj_tiles);
        if (G == 16 && iv == 9) return launch_l2_int<8, 2, 4, 2, 1>(st, P, max_nj_tiles);
        if (G == 16 && iv == 8) return launch_l2_int<8, 2, 8, 2>(st, P, max_nj_tiles);
#endif
        switch (G) {
            case 8:  return launch_l2_int<4, 2, 4, 2>(st, P, max_nj_tiles);
            case 16: return launch_l2_int<8, 2, 4, 2>(st, P, max_nj_tiles);
            case 32: return launch_l2_int<16, 2, 4, 2>(st, P, max_nj_tiles);
            default: break;               // G = 18 (LIOP, never integer): f32 tiles
        }
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

// ------------------------------------------------------------------------------------------------
// exact scan of single (pair, query) items: the reference arithmetic over every dataset row.
// Used for un-certified queries (rare), descriptor lengths without a tensor kernel, and as the
// independent on-device cross-check of the MFMA path.  One workgroup per item.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lex_less(float da, uint32_t ia, float db, uint32_t ib)
{
    return da < db || (da == db && ia < ib);
}

__global__ __launch_bounds__(256)
void l2_exact_items_kernel(const MatchParams P, uint32_t count, int scan_all)
{
    __shared__ float sd0[256], sd1[256];
    __shared__ uint32_t si0[256], si1[256];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        uint32_t pair, q;
        pair = it / P.q_stride; q = it % P.q_stride;
        const uint2 pr = P.pairs[pair];
        const ImgDev* __restrict__ Ip = P.imgs + pr.x;
        const ImgDev* __restrict__ Jp = P.imgs + pr.y;
        if (q >= Jp->n) continue;                                 // block-uniform
        if (scan_all == 2 && P.nn_idx[(size_t)pair * P.q_stride + q] != kFallback) continue;
        const uint32_t dim = Ip->dim, nI = Ip->n;
        const float* qv = Jp->rows + (size_t)q * dim;
        float d0 = R3DM_INF, d1 = R3DM_INF; uint32_t i0 = kNone, i1 = kNone;
        for (uint32_t r = threadIdx.x; r < nI; r += 256) {
            const float d = exact_l2sq(Ip->rows + (size_t)r * dim, qv, dim);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = r; }
            else if (d < d1) { d1 = d; i1 = r; }
        }
        sd0[threadIdx.x] = d0; sd1[threadIdx.x] = d1; si0[threadIdx.x] = i0; si1[threadIdx.x] = i1;
        r3dm_syncthreads();
        for (uint32_t s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
                // merge two sorted pairs under the (distance, index) order
                float a0 = sd0[threadIdx.x], a1 = sd1[threadIdx.x]; uint32_t x0 = si0[threadIdx.x], x1 = si1[threadIdx.x];
                const float b0 = sd0[threadIdx.x + s], b1 = sd1[threadIdx.x + s];
                const uint32_t y0 = si0[threadIdx.x + s], y1 = si1[threadIdx.x + s];
                float r0, r1; uint32_t j0, j1;
                if (lex_less(b0, y0, a0, x0)) {
                    r0 = b0; j0 = y0;
                    if (lex_less(b1, y1, a0, x0)) { r1 = b1; j1 = y1; } else { r1 = a0; j1 = x0; }
                } else {
                    r0 = a0; j0 = x0;
                    if (lex_less(b0, y0, a1, x1)) { r1 = b0; j1 = y0; } else { r1 = a1; j1 = x1; }
                }
                sd0[threadIdx.x] = r0; sd1[threadIdx.x] = r1; si0[threadIdx.x] = j0; si1[threadIdx.x] = j1;
            }
            r3dm_syncthreads();
        }
        if (threadIdx.x == 0) {
            if (nI < 2) emit_result(P, pair, q, R3DM_INF, kNone, R3DM_INF, kNone);
            else emit_result(P, pair, q, sd0[0], si0[0], sd1[0], si1[0]);
        }
        r3dm_syncthreads();
    }
}

hipError_t launch_l2_exact_items(hipStream_t st, const MatchParams& P, uint32_t count, int scan_all)
{
    if (count == 0) return hipSuccess;
    uint32_t grid = count < 16384u ? count : 16384u;
    hipLaunchKernelGGL(l2_exact_items_kernel, dim3(grid), dim3(256), 0, st, P, count, scan_all);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exact scan of the per-pair fallback lists: one workgroup per pair, lane = one uncertified query
// (its row in registers), wave w = rows {8w .. 8w+7} of every 32-row tile of image I staged through
// LDS (row reads are wave-uniform -> LDS broadcast).  Reference arithmetic, (distance, row) order.
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256)
void l2_exact_batch_kernel(const MatchParams P)
{
    constexpr int D4 = G * 2;                        // float4 per (padded) row
    __shared__ f32x4 tile[32 * D4];                  // 32 rows x Dpad floats
    __shared__ float md0[256], md1[256];
    __shared__ uint32_t mi0[256], mi1[256];
    __shared__ uint32_t s_ticket;
    const uint32_t pair = blockIdx.x;
    const uint32_t S = P.fb_slices, slice = blockIdx.y;
    const uint32_t cnt_all = P.fb_cnt[pair];
    if (cnt_all == 0) return;
    const uint32_t cnt = cnt_all < kFbPerPair ? cnt_all : kFbPerPair;
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nI = Ip->n, dim = Ip->dim, d4 = dim >> 2;      // dim % 4 == 0 guaranteed by the launcher
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const gf4p irows = (gf4p)Ip->rows;
    // this workgroup's rows of image I: whole 32-row tiles, slice `slice` of S
    const uint32_t tiles_per = ((nI + 31u) / 32u + S - 1u) / S;
    const uint32_t row_beg = slice * tiles_per * 32u;
    const uint32_t row_end = (row_beg + tiles_per * 32u < nI) ? row_beg + tiles_per * 32u : nI;
    for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
        const bool active = b0 + lane < cnt;
        const uint32_t q = P.fb_q[(size_t)pair * kFbPerPair + (active ? b0 + lane : b0)];
        f32x4 qv[D4];
        const gf4p qrow = (gf4p)Jp->rows + (size_t)q * d4;
#pragma unroll
        for (int k = 0; k < D4; ++k) qv[k] = (k < (int)d4) ? qrow[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        float d0 = R3DM_INF, d1 = R3DM_INF; uint32_t i0 = kNone, i1 = kNone;
        for (uint32_t t0 = row_beg; t0 < row_end; t0 += 32) {
            r3dm_syncthreads();
            const uint32_t rows_here = (row_end - t0 < 32u) ? row_end - t0 : 32u;
            for (uint32_t e = threadIdx.x; e < rows_here * d4; e += 256) {
                const uint32_t r = e / d4, k = e % d4;
                tile[r * D4 + k] = irows[(size_t)(t0 + r) * d4 + k];
            }
            r3dm_syncthreads();
            for (uint32_t rr = 0; rr < 8; ++rr) {
                const uint32_t r = wave * 8 + rr;
                if (r >= rows_here) break;                         // wave-uniform
                float result = 0.0f;
#pragma unroll
                for (int k = 0; k < D4; ++k) {
                    if (k < (int)d4) {
                        const f32x4 a = tile[r * D4 + k];
                        const float e0 = a[0] - qv[k][0], e1 = a[1] - qv[k][1], e2 = a[2] - qv[k][2], e3 = a[3] - qv[k][3];
                        result += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
                    }
                }
                const uint32_t row = t0 + r;
                if (result < d0) { d1 = d0; i1 = i0; d0 = result; i0 = row; }
                else if (result < d1) { d1 = result; i1 = row; }
            }
        }
        // merge the four waves' (best, runner-up) per lane under the (distance, row) order
        r3dm_syncthreads();
        md0[threadIdx.x] = d0; md1[threadIdx.x] = d1; mi0[threadIdx.x] = i0; mi1[threadIdx.x] = i1;
        r3dm_syncthreads();
        if (wave == 0 && active) {
            float a0 = md0[lane], a1 = md1[lane]; uint32_t x0 = mi0[lane], x1 = mi1[lane];
            for (uint32_t w = 1; w < 4; ++w) {
                const float b0_ = md0[w * 64 + lane], b1_ = md1[w * 64 + lane];
                const uint32_t y0 = mi0[w * 64 + lane], y1 = mi1[w * 64 + lane];
                float r0, r1; uint32_t j0, j1;
                if (lex_less(b0_, y0, a0, x0)) {
                    r0 = b0_; j0 = y0;
                    if (lex_less(b1_, y1, a0, x0)) { r1 = b1_; j1 = y1; } else { r1 = a0; j1 = x0; }
                } else {
                    r0 = a0; j0 = x0;
                    if (lex_less(b0_, y0, a1, x1)) { r1 = b0_; j1 = y0; } else { r1 = a1; j1 = x1; }
                }
                a0 = r0; a1 = r1; x0 = j0; x1 = j1;
            }
            if (S > 1) P.fb_part[((size_t)pair * kFbPerPair + b0 + lane) * S + slice] = make_float4(a0, __uint_as_float(x0), a1, __uint_as_float(x1));
            else if (nI < 2) emit_result(P, pair, q, R3DM_INF, kNone, R3DM_INF, kNone);
            else emit_result(P, pair, q, a0, x0, a1, x1);
        }
    }
    if (S > 1) {
        // the last slice of the pair to get here merges the S partial (best, runner-up) of every query under the (distance, row) order
        __threadfence();
        r3dm_syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(&P.fb_done[pair], 1u);
        r3dm_syncthreads();
        if (s_ticket != S - 1u) return;
        __threadfence();
        for (uint32_t k = threadIdx.x; k < cnt; k += 256) {
            const uint32_t q = P.fb_q[(size_t)pair * kFbPerPair + k];
            const float4* part = P.fb_part + ((size_t)pair * kFbPerPair + k) * S;
            float4 v = part[0];
            float a0 = v.x, a1 = v.z; uint32_t x0 = __float_as_uint(v.y), x1 = __float_as_uint(v.w);
            for (uint32_t w = 1; w < S; ++w) {
                v = part[w];
                const float b0_ = v.x, b1_ = v.z;
                const uint32_t y0 = __float_as_uint(v.y), y1 = __float_as_uint(v.w);
                float r0, r1; uint32_t j0, j1;
                if (lex_less(b0_, y0, a0, x0)) {
                    r0 = b0_; j0 = y0;
                    if (lex_less(b1_, y1, a0, x0)) { r1 = b1_; j1 = y1; } else { r1 = a0; j1 = x0; }
                } else {
                    r0 = a0; j0 = x0;
                    if (lex_less(b0_, y0, a1, x1)) { r1 = b0_; j1 = y0; } else { r1 = a1; j1 = x1; }
                }
                a0 = r0; a1 = r1; x0 = j0; x1 = j1;
            }
            if (nI < 2) emit_result(P, pair, q, R3DM_INF, kNone, R3DM_INF, kNone);
            else emit_result(P, pair, q, a0, x0, a1, x1);
        }
    }
}

hipError_t launch_l2_exact_batch(hipStream_t st, const MatchParams& P, uint32_t G)
{
    if (P.n_pairs == 0) return hipSuccess;
    if (P.n_pairs > kMaxBlocksOf256 || P.fb_slices < 1 || P.fb_slices > 64 || (uint64_t)P.n_pairs * P.fb_slices > kMaxBlocksOf256) return hipErrorInvalidValue;
    const dim3 grid(P.n_pairs, P.fb_slices);
    switch (G) {
        case 8:  hipLaunchKernelGGL((l2_exact_batch_kernel<8>), grid, dim3(256), 0, st, P); break;
        case 16: hipLaunchKernelGGL((l2_exact_batch_kernel<16>), grid, dim3(256), 0, st, P); break;
        case 18: hipLaunchKernelGGL((l2_exact_batch_kernel<18>), grid, dim3(256), 0, st, P); break;
        case 32: hipLaunchKernelGGL((l2_exact_batch_kernel<32>), grid, dim3(256), 0, st, P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Hamming 2-NN (binary descriptors, e.g. 486-bit A-KAZE MLDB stored in 16 words): integer VALU only.
// Each lane owns QL query rows in registers; dataset rows arrive wave-uniformly through the scalar
// cache (s_load), so a row costs W x (v_xor + v_bcnt-accumulate) per query and no LDS/vector memory.
// The running top-2 is kept on packed keys (distance << 22 | row): unsigned min / med3 then break
// ties towards the lowest row, exactly the oracle's rule.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) uint32_t* cu32p;   // constant address space -> SMEM loads

template <int W, int QL>
__global__ __launch_bounds__(256)
void hamming_knn2_kernel(const MatchParams P)
{
    const uint32_t pair = blockIdx.x / P.qb_per_pair;
    const uint32_t qb = blockIdx.x % P.qb_per_pair;
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nI = Ip->n, nJ = Jp->n;
    const uint32_t q0 = (qb * 256u + threadIdx.x) * QL;
    const uint32_t wave_q0 = (qb * 256u + (threadIdx.x & ~63u)) * QL;
    if (wave_q0 >= nJ) return;

    uint32_t qw[QL][W];
#pragma unroll
    for (int k = 0; k < QL; ++k) {
        uint32_t q = q0 + k; if (q >= nJ) q = nJ - 1;
        const uint32_t* src = Jp->bin + (size_t)q * W;
#pragma unroll
        for (int w = 0; w < W; ++w) qw[k][w] = src[w];
    }
    uint32_t k0[QL], k1[QL];
#pragma unroll
    for (int k = 0; k < QL; ++k) { k0[k] = 0xFFFFFFFFu; k1[k] = 0xFFFFFFFFu; }

    const cu32p base = (cu32p)(uintptr_t)Ip->bin;
#pragma unroll 2
    for (uint32_t r = 0; r < nI; ++r) {
        const cu32p row = base + (size_t)r * W;
        uint32_t a[W];
#pragma unroll
        for (int w = 0; w < W; ++w) a[w] = row[w];
#pragma unroll
        for (int k = 0; k < QL; ++k) {
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) d += (uint32_t)__builtin_popcount(qw[k][w] ^ a[w]);
            const uint32_t key = (d << 22) | r;
            const uint32_t hi = k0[k] > key ? k0[k] : key;       // max(k0, key)
            k1[k] = k1[k] < hi ? k1[k] : hi;                     // min(k1, max(k0, key))  (v_med3_u32)
            k0[k] = k0[k] < key ? k0[k] : key;
        }
    }
#pragma unroll
    for (int k = 0; k < QL; ++k) {
        const uint32_t q = q0 + k;
        if (q >= nJ) continue;
        const size_t o = (size_t)pair * P.q_stride + q;
        if (nI < 2) { P.nn_idx[o] = kNone; if (P.knn_idx) { P.knn_idx[2*o] = -1; P.knn_idx[2*o+1] = -1; P.knn_dist[2*o] = R3DM_INF; P.knn_dist[2*o+1] = R3DM_INF; } continue; }
        const uint32_t d0 = k0[k] >> 22, d1 = k1[k] >> 22;
        const uint32_t i0 = k0[k] & 0x3FFFFFu, i1 = k1[k] & 0x3FFFFFu;
        // NNdistanceRatio on unsigned distances converted to float
        P.nn_idx[o] = ((float)d0 < P.ratio_R * (float)d1) ? i0 : kNone;
        if (P.knn_idx) {
            P.knn_idx[2 * o] = (int32_t)i0; P.knn_idx[2 * o + 1] = (int32_t)i1;
            P.knn_dist[2 * o] = (float)d0;  P.knn_dist[2 * o + 1] = (float)d1;
        }
    }
}

hipError_t launch_hamming_knn2(hipStream_t st, const MatchParams& Pin, uint32_t words, uint32_t max_n)
{
    MatchParams P = Pin;
    constexpr int QL = 4;
    P.qb_per_pair = (max_n + 256u * QL - 1) / (256u * QL);
    const uint64_t grid64 = (uint64_t)P.n_pairs * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    const uint32_t grid = (uint32_t)grid64;
    switch (words) {
        case 8:  hipLaunchKernelGGL((hamming_knn2_kernel<8, QL>), dim3(grid), dim3(256), 0, st, P); break;
        case 16: hipLaunchKernelGGL((hamming_knn2_kernel<16, QL>), dim3(grid), dim3(256), 0, st, P); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// finalisation: one workgroup per pair.  Compacts nn_idx[pair][*] into (i_, j_) keys, sorts them
// (IndMatch::getDeduplicated order), drops matches whose (xI,yI,xJ,yJ) repeat an earlier one
// (IndMatchDecorator), appends the list to the batch output and records (offset, count).
// ------------------------------------------------------------------------------------------------
// body shared by the two storage classes of the sort buffer: `keys` / `drop` point into LDS (fast path) or into a
// per-pair slice of global scratch (pairs that keep more matches than the LDS budget holds: views with > 16k features)
template <bool GLOBAL_BUFFERS, class KeyT, class DropT>
__device__ __forceinline__ void finalize_body(const FinalizeParams& P, KeyT keys, DropT drop, unsigned long long* s_off_p,
                                              uint32_t* wave_cnt, uint32_t* s_total_p, uint32_t pair)
{
    const uint2 pr = P.pairs[pair];
    const ImgDev* __restrict__ Ip = P.imgs + pr.x;
    const ImgDev* __restrict__ Jp = P.imgs + pr.y;
    const uint32_t nJ = Jp->n;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t* src = P.nn_idx + (size_t)pair * P.q_stride;

    uint32_t m = 0;                                  // block-uniform running count
    for (uint32_t base = 0; base < nJ; base += 256) {
        const uint32_t q = base + threadIdx.x;
        const uint32_t v = (q < nJ) ? src[q] : kNone;
        const bool keep = (v < kFallback);
        const unsigned long long bal = __ballot(keep);
        const uint32_t before = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
        r3dm_syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) { const uint32_t cw = wave_cnt[w]; if (w < wave) woff += cw; tot += cw; }
        if (keep) keys[m + woff + before] = ((unsigned long long)v << 32) | q;
        m += tot;
        r3dm_syncthreads();
    }

    if (m > 1) {
        // pad to a power of two and bitonic-sort ascending
        uint32_t cap = 1; while (cap < m) cap <<= 1;
        for (uint32_t k = m + threadIdx.x; k < cap; k += 256) keys[k] = ~0ull;
        if (GLOBAL_BUFFERS) __threadfence();        // agent scope: a workgroup-scope fence emits no vmcnt wait on gfx950
        r3dm_syncthreads();
        for (uint32_t size = 2; size <= cap; size <<= 1) {
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t tId = threadIdx.x; tId < (cap >> 1); tId += 256) {
                    const uint32_t lo = 2 * tId - (tId & (stride - 1));
                    const uint32_t hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const unsigned long long x = keys[lo], y = keys[hi];
                    if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
                }
                if (GLOBAL_BUFFERS) __threadfence();        // agent scope: a workgroup-scope fence emits no vmcnt wait on gfx950
                r3dm_syncthreads();
            }
        }
        // coordinate de-duplication: only possible when both views contain repeated positions.  Element k is dropped when an
        // EARLIER element of the (i, j)-sorted list has the same position classes (ci, cj).  canon[] is the smallest index of a
        // class, so such an element has i >= ci: the scan starts at the first key with i >= ci (binary search) -- for a feature
        // that is its own class representative (the usual case) that is the handful of earlier matches of the same i.
        if (Ip->canon && Jp->canon) {
            for (uint32_t k = threadIdx.x; k < m; k += 256) {
                const uint32_t ci = Ip->canon[(uint32_t)(keys[k] >> 32)], cj = Jp->canon[(uint32_t)keys[k]];
                uint32_t lo = 0, hi = k;
                const unsigned long long want = (unsigned long long)ci << 32;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
                unsigned char d = 0;
                for (uint32_t e = lo; e < k && !d; ++e)
                    d = (Ip->canon[(uint32_t)(keys[e] >> 32)] == ci) && (Jp->canon[(uint32_t)keys[e]] == cj);
                drop[k] = d;
            }
            if (GLOBAL_BUFFERS) __threadfence();        // agent scope: a workgroup-scope fence emits no vmcnt wait on gfx950
            r3dm_syncthreads();
            // stable in-place compaction, 256 elements per round: every element moves to a position <= its own, and a round
            // reads its 256 keys before the barrier that precedes its writes
            uint32_t w = 0;
            for (uint32_t base = 0; base < m; base += 256) {
                const uint32_t k = base + threadIdx.x;
                const bool keep = (k < m) && !drop[k];
                const unsigned long long kk = (k < m) ? keys[k] : 0ull;
                const unsigned long long bal = __ballot(keep);
                const uint32_t before = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
                if (GLOBAL_BUFFERS) __threadfence();
                r3dm_syncthreads();
                uint32_t woff = 0, tot = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) { const uint32_t cw = wave_cnt[q]; if (q < wave) woff += cw; tot += cw; }
                if (keep) keys[w + woff + before] = kk;
                w += tot;
                if (GLOBAL_BUFFERS) __threadfence();
                r3dm_syncthreads();
            }
            m = w;
        }
    }

    if (threadIdx.x == 0) {
        const unsigned long long off = m ? atomicAdd(P.total, (unsigned long long)m) : 0ull;
        *s_off_p = off;
        P.pair_off[pair] = off;
        P.pair_cnt[pair] = m;
    }
    r3dm_syncthreads();
    const unsigned long long off = *s_off_p;
    if (off + m <= P.out_cap)
        for (uint32_t k = threadIdx.x; k < m; k += 256) {
            const unsigned long long kk = keys[k];
            r3dm_match mm; mm.i = (uint32_t)(kk >> 32); mm.j = (uint32_t)kk;
            P.out[off + k] = mm;
        }
}

__global__ __launch_bounds__(256)
void finalize_pairs_kernel(const FinalizeParams P)
{
    // all LDS comes from the dynamic region (keeps the base 16-byte aligned):
    // [keys: sort_cap x u64][drop: sort_cap x u8][s_off u64][wave_cnt 4 x u32][s_total u32]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* keys = (unsigned long long*)smem_raw;
    unsigned char* drop = smem_raw + (size_t)P.sort_cap * 8;
    unsigned long long* s_off_p = (unsigned long long*)(smem_raw + (size_t)P.sort_cap * 9);
    uint32_t* wave_cnt = (uint32_t*)(s_off_p + 1);
    uint32_t* s_total_p = wave_cnt + 4;
    const uint32_t pair = blockIdx.x;

    if (P.spill_keys == nullptr) { finalize_body<false>(P, keys, drop, s_off_p, wave_cnt, s_total_p, pair); return; }

    // views larger than the LDS budget: count what the pair keeps, spill only if it does not fit
    const uint32_t nJ = P.imgs[P.pairs[pair].y].n;
    const uint32_t* src = P.nn_idx + (size_t)pair * P.q_stride;
    uint32_t cnt = 0;
    for (uint32_t q = threadIdx.x; q < nJ; q += 256) cnt += (src[q] < kFallback) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off);
    if ((threadIdx.x & 63u) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    r3dm_syncthreads();
    const uint32_t kept = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    r3dm_syncthreads();
    if (kept <= P.sort_cap) finalize_body<false>(P, keys, drop, s_off_p, wave_cnt, s_total_p, pair);
    else finalize_body<true>(P, P.spill_keys + (size_t)pair * P.spill_stride, P.spill_drop + (size_t)pair * P.spill_stride,
                       s_off_p, wave_cnt, s_total_p, pair);
}

hipError_t launch_finalize(hipStream_t st, const FinalizeParams& P)
{
    if (P.n_pairs == 0) return hipSuccess;
    if (P.n_pairs > kMaxBlocksOf256) return hipErrorInvalidValue;
    const size_t lds = (size_t)P.sort_cap * 9 + 32;                     // keys + drop flags + scalars (sort_cap is a power of two >= 8)
    hipError_t e = hipFuncSetAttribute((const void*)finalize_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(finalize_pairs_kernel, dim3(P.n_pairs), dim3(256), lds, st, P);
    return hipGetLastError();
}

}  // namespace r3dm
