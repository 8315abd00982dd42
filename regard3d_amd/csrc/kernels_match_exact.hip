// kernels_match_exact.hip -- what stands behind the tile kernels: the exact scans in the reference's arithmetic for the (rare) queries
// whose top-2 a tile kernel could not certify, and the per-pair finalisation (compaction of the accepted queries, (i_, j_) ordering,
// coordinate de-duplication -- OpenMVG's IndMatchDecorator as restated in SURVEY.md A.4;
// /root/reference/src/R3DComputeMatches.cpp:479-487).
#include "kernels_match_common.hpp"

namespace r3dm {

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
// Both operands come from the fragment-order tiles (ImgDev::tiled: the f32 values verbatim; float4 chunk k4 of row q is float4
// ((q >> 5) 2G + k4) 32 + (q & 31)): a tile of image I is one contiguous G-KiB copy into LDS, and a view needs no row-major copy
// for this scan (integer-valued views -- SIFT bins -- are registered without one).
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256)
void l2_exact_batch_kernel(const MatchParams P)
{
    constexpr int D4 = G * 2;                        // float4 per (padded) row
    __shared__ f32x4 tile[32 * D4];                  // 32 rows x Dpad floats, fragment order: [chunk k][row r]
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
    const gf4p itiles = (gf4p)Ip->tiled;
    // this workgroup's rows of image I: whole 32-row tiles, slice `slice` of S
    const uint32_t tiles_per = ((nI + 31u) / 32u + S - 1u) / S;
    const uint32_t row_beg = slice * tiles_per * 32u;
    const uint32_t row_end = (row_beg + tiles_per * 32u < nI) ? row_beg + tiles_per * 32u : nI;
    for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
        const bool active = b0 + lane < cnt;
        const uint32_t q = P.fb_q[(size_t)pair * kFbPerPair + (active ? b0 + lane : b0)];
        f32x4 qv[D4];
        const gf4p qrow = (gf4p)Jp->tiled + (size_t)(q >> 5) * (D4 * 32) + (q & 31u);
#pragma unroll
        for (int k = 0; k < D4; ++k) qv[k] = (k < (int)d4) ? qrow[k * 32] : f32x4{0.f, 0.f, 0.f, 0.f};
        float d0 = R3DM_INF, d1 = R3DM_INF; uint32_t i0 = kNone, i1 = kNone;
        for (uint32_t t0 = row_beg; t0 < row_end; t0 += 32) {
            r3dm_syncthreads();
            const uint32_t rows_here = (row_end - t0 < 32u) ? row_end - t0 : 32u;
            {
                const gf4p src = itiles + (size_t)(t0 >> 5) * (D4 * 32);
                for (uint32_t e = threadIdx.x; e < 32u * D4; e += 256) tile[e] = src[e];
            }
            r3dm_syncthreads();
            for (uint32_t rr = 0; rr < 8; ++rr) {
                const uint32_t r = wave * 8 + rr;
                if (r >= rows_here) break;                         // wave-uniform
                float result = 0.0f;
#pragma unroll
                for (int k = 0; k < D4; ++k) {
                    if (k < (int)d4) {
                        const f32x4 a = tile[k * 32 + r];
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
// finalisation: one workgroup per pair.  Compacts nn_idx[pair][*] into (i_, j_) keys, sorts them
// (IndMatch::getDeduplicated order), drops matches whose (xI,yI,xJ,yJ) repeat an earlier one
// (IndMatchDecorator), appends the list to the batch output and records (offset, count).
// ------------------------------------------------------------------------------------------------
// body shared by the two storage classes of the sort buffer: `keys` / `drop` point into LDS (fast path) or into a
// per-pair slice of global scratch (pairs that keep more matches than the LDS budget holds: views with > 16k features)
// a pair that keeps more matches than the LDS sort holds works in its own slice of global scratch: what one wave stored must be
// visible to the other waves of the SAME workgroup after a barrier.  They share one CU and its L1, so a workgroup-scope fence and
// the completion of this wave's own memory operations is all that takes (an agent-scope fence -- what __threadfence() is -- writes
// the L2 back on this eight-L2 part: ~20 us per sorting stage, 2.5 ms for a pair of 20 k matches; same shortcut and the same two
// conditions as wg_fence in kernels_filter.hip: gfx9 s_waitcnt encoding, no threadgroup-split mode -- build.sh refuses -mtgsplit)
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(__gfx942__)
#error "fin_fence: the s_waitcnt encoding and the same-CU L1 argument are written for gfx942 / gfx950"
#endif
#endif
__device__ __forceinline__ void fin_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): this wave's global stores and loads have completed
}

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
        if (GLOBAL_BUFFERS) fin_fence();
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
                if (GLOBAL_BUFFERS) fin_fence();
                r3dm_syncthreads();
            }
        }
        // coordinate de-duplication: only possible when the QUERY view J repeats positions -- two of its features at one position that
        // matched the same row of I, or two rows of I at one position (a query has one match, so repeated positions in I alone never
        // produce two matches with equal coordinates).  Element k is dropped when an
        // EARLIER element of the (i, j)-sorted list has the same position classes (ci, cj).  canon[] is the smallest index of a
        // class, so such an element has i >= ci: the scan starts at the first key with i >= ci (binary search) -- for a feature
        // that is its own class representative (the usual case) that is the handful of earlier matches of the same i.
        // (Until round 6 the test was "both views repeat positions": a pair whose I had none kept both matches of a repeated J position.)
        if (Ip->canon && Jp->canon && Jp->has_dup) {
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
            if (GLOBAL_BUFFERS) fin_fence();
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
                if (GLOBAL_BUFFERS) fin_fence();
                r3dm_syncthreads();
                uint32_t woff = 0, tot = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) { const uint32_t cw = wave_cnt[q]; if (q < wave) woff += cw; tot += cw; }
                if (keep) keys[w + woff + before] = kk;
                w += tot;
                if (GLOBAL_BUFFERS) fin_fence();
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
