// kernels_match_16bit.hip -- the 16-bit nominators of the squared-L2 2-NN (opt-in fast paths, bit-identical results):
//   l2_knn2_int_kernel      integer-valued rows (SIFT bins) on bf16 tiles: the keys are the distances' ranks   (r3dm_set_integer_mfma)
//   l2_knn2_split_kernel    real-valued rows as two f16 pieces, three MFMAs per 16 dimensions               (r3dm_set_split_mfma)
//   l2_knn2_counts2_kernel  rows that are small integers x a row scale (LIOP): one MFMA per 16 dimensions, one list per query
//   l2_knn2_counts_kernel   the same with one list per lane half (256-dimensional views; developer A/B)
// with their staging kernels.  What they replace is what kernels_match.hip replaces (/root/reference/src/R3DComputeMatches.cpp:437-489,
// src/Regard3DFeatures.h:44-48): nominees are re-scored in the reference's f32 arithmetic and certified by the shared tail
// (kernels_match_common.hpp), so the distances that are compared, ratio-tested and returned are the reference's own.
#include "kernels_match_common.hpp"

namespace r3dm {

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
        if (ABL < 2) abuf[g % PF] = bload16(ra, voffA, soffA + (uint32_t)g * 1024u);
        if (ABL < 2 && g == (GB > 2 ? 2 : GB - 1)) {   // next tile's norms, element 4 qd + k = row 8 qd + 4 h + k: the accumulator layout
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
            if constexpr ((ABL & 1) != 0) {
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

// G = padded dim / 8; hipErrorNotSupported: no bf16 kernel for this G (the caller keeps the f32 tiles)
hipError_t launch_l2_knn2_int(hipStream_t st, const MatchParams& P, uint32_t G, uint32_t max_nj_tiles)
{
#ifdef R3DM_DEVTOOLS
    // R3DM_L2_INT_VARIANT (A/B measurements on 780 pairs of 8192 x 8192 rows, D = 128; the f32 tiles take 95.0 ms):
    //   2 = the product | 22 = NJ 2 x 2 waves/SIMD (the product until round 6, 12.4 ms) | 8 = same with a whole-tile prefetch window (12.95 ms) |
    //   4 = NJ 4 x 1 wave/SIMD (17.6 ms) | 9 = 2 without the epilogue (timing only, 11.4 ms) | 5 / 59 / 6: tiles shared through LDS (kernels_match_hamming.hip)
    //   round 5, same workload (2 = 11.67 ms): 9 = no epilogue 10.52 | 10 = no tile / norm loads 7.57 | 11 = neither 7.26 (timing only);
    //   5 = LDS-shared 12.16 | 59 = LDS-shared without the epilogue 9.36.  The loads are the bound: with two query tiles per wave a CU's
    //   four SIMDs consume 4 KiB of fragments per 64 matrix cycles = 64 B/clk, the whole rate of its L1 address path
    static const int iv = r3dm_dev_knob("R3DM_L2_INT_VARIANT", 2);
    if (G == 16 && (iv == 5 || iv == 59 || iv == 6 || iv == 7 || iv == 79)) return launch_l2_int_lds_variant(st, P, max_nj_tiles, iv);
    if (G == 16 && iv == 4) return launch_l2_int<8, 4, 4, 1>(st, P, max_nj_tiles);
    if (G == 16 && iv == 22) return launch_l2_int<8, 2, 4, 2>(st, P, max_nj_tiles);     // two query tiles per wave (the product of rounds 3-5)
    if (G == 16 && iv == 9) return launch_l2_int<8, 2, 4, 2, 1>(st, P, max_nj_tiles);
    if (G == 16 && iv == 8) return launch_l2_int<8, 2, 8, 2>(st, P, max_nj_tiles);
    if (G == 16 && iv == 10) return launch_l2_int<8, 2, 4, 2, 2>(st, P, max_nj_tiles);
    if (G == 16 && iv == 11) return launch_l2_int<8, 2, 4, 2, 3>(st, P, max_nj_tiles);
#endif
    switch (G) {
        case 8:  return launch_l2_int<4, 2, 4, 2>(st, P, max_nj_tiles);
        // (128 dimensions: THREE query tiles per wave still fit two waves per SIMD -- 251 registers, no spills -- and a fragment block
        //  then feeds three MFMAs instead of two: 10.40 ms against 10.80 on 780 pairs of 8,192 rows, same graphs; round 6)
        case 16: return launch_l2_int<8, 3, 4, 2>(st, P, max_nj_tiles);
        case 32: return launch_l2_int<16, 2, 4, 2>(st, P, max_nj_tiles);
        default: return hipErrorNotSupported;
    }
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
// inside the matrix unit); host: MatchParams::err_scale = (3 Dpad + 36) 2^-22.  That is 3x the slack of the f32 tiles, which
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

// The DATASET side of the nominator reads the rows in the order of their scales (512 classes per binary order, i.e. scales within
// 0.2 % of one another inside a class): the quad test of the kernel bounds four keys with the largest scale of their four rows, and
// with rows in keypoint order (scales 30 % apart) that bound let a large share of the quads through to the per-key path.
// cperm[position] = row, rows ordered by (scale class, row index), kNone behind the last row -- a STABLE SORT by class, so the order
// is the same from run to run by construction (which rows share a tile decides which queries the epilogue sends to the exact scan;
// the results are exact either way), at a cost that does not depend on how the scales are distributed.  Two launches for views of up
// to 65,536 rows: (1) every workgroup orders a RUN of 1,024 consecutive rows by (class, row) -- a bitonic network on 32-bit keys
// class << 10 | row-in-run in 4 KiB of LDS; (2) every row finds its place among all runs by counting, in each other run, the keys
// that sort before it (two binary searches per run: rows of earlier runs win ties, rows of later runs lose them) -- the runs are
// 4 KiB each, searched in LDS (32 runs at a time).  8,192 rows: 8 workgroups twice, 7.5 + 8.3 us; 28 k rows: 7 + 23 us.  Larger views take a bitonic network over
// 64-bit keys class << 32 | row, chunks of 16,384 keys in LDS, chunk-crossing strides through global memory.
// (Rounds 4-5: a one-workgroup counting sort on LDS atomics -- 210-265 us per 28 k-row view whose scales fall into a few dozen classes,
// the atomics of a wavefront serialising on them -- plus a ranking kernel of O(rows x class size): 413 us per 8 k-row view of one
// class.  Round 6 first tried the one-workgroup LDS bitonic sort for every size: 16,384 64-bit keys through 105 stages are LDS
// bandwidth, ~300 us.)
// class of a scale: its sign-free exponent and 9 mantissa bits (0.2 % steps); a scale of 0 or a denormal (an all-zero row) is class 0
constexpr uint32_t kOrdNT = 1024;
__device__ __forceinline__ uint32_t order_class(float s);
__device__ __forceinline__ unsigned long long order_key(float s, uint32_t row) { return ((unsigned long long)order_class(s) << 32) | row; }
// compare-exchange passes of the bitonic network on a chunk of C keys in LDS: strides first .. 1 of merge step `size` (global index = base + local)
__device__ __forceinline__ void order_lds_passes(unsigned long long* lk, uint32_t C, uint32_t base, uint32_t size, uint32_t first_stride)
{
    for (uint32_t stride = first_stride; stride >= 1u; stride >>= 1) {
        for (uint32_t e = threadIdx.x; e < C / 2u; e += kOrdNT) {
            const uint32_t lo = 2u * e - (e & (stride - 1u)), hi = lo + stride;
            const bool up = (((base + lo) & size) == 0u);
            const unsigned long long a = lk[lo], b = lk[hi];
            if ((a > b) == up) { lk[lo] = b; lk[hi] = a; }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ uint32_t order_class(float s)
{
    const uint32_t bits = __float_as_uint(s);
    return ((bits >> 23) & 255u) == 0u ? 0u : (bits >> 14) & 0x1FFFFu;
}
constexpr uint32_t kRun = 1024;
// (1) one workgroup of 512 threads per run of 1,024 consecutive rows: keys class << 10 | row-in-run (rows beyond n: all ones), sorted
__global__ __launch_bounds__(512)
void stage_counts_runs_kernel(const float* __restrict__ cscale, uint32_t n, uint32_t* __restrict__ runs)
{
    __shared__ uint32_t lk[kRun];
    const uint32_t base = blockIdx.x * kRun;
    for (uint32_t e = threadIdx.x; e < kRun; e += 512u) { const uint32_t r = base + e; lk[e] = r < n ? (order_class(cscale[r]) << 10) | e : 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t size = 2u; size <= kRun; size <<= 1)
        for (uint32_t stride = size >> 1; stride >= 1u; stride >>= 1) {
            const uint32_t e = threadIdx.x;
            const uint32_t lo = 2u * e - (e & (stride - 1u)), hi = lo + stride;
            const bool up = ((lo & size) == 0u);
            const uint32_t a = lk[lo], b = lk[hi];
            if ((a > b) == up) { lk[lo] = b; lk[hi] = a; }
            __syncthreads();
        }
    for (uint32_t e = threadIdx.x; e < kRun; e += 512u) runs[base + e] = lk[e];
}
// (2) one thread per element of a sorted run: its place = its index in its own run + the keys of the other runs that sort before it.
// The other runs are searched in LDS, 32 runs (128 KiB) at a time: a binary search is ten dependent reads, and 28 runs x 10 reads from
// L2 were 61 us per 28 k-row view.
constexpr uint32_t kPlaceGroup = 32;
__global__ __launch_bounds__(1024)
void stage_counts_place_kernel(const uint32_t* __restrict__ runs, uint32_t n_runs, uint32_t n, uint32_t n_pad, uint32_t* __restrict__ cperm)
{
    extern __shared__ uint32_t place_lds[];                     // [min(n_runs, kPlaceGroup)][kRun]
    const uint32_t r = blockIdx.x, e = threadIdx.x;
    const uint32_t key = runs[r * kRun + e];
    const bool pad = key == 0xFFFFFFFFu;
    const uint32_t cls_lo = key & ~1023u, cls_hi = key | 1023u;   // keys of the same class: [cls_lo, cls_hi]
    uint32_t pos = e;
    for (uint32_t q0 = 0; q0 < n_runs; q0 += kPlaceGroup) {
        const uint32_t qn = n_runs - q0 < kPlaceGroup ? n_runs - q0 : kPlaceGroup;
        __syncthreads();
        {
            const uint4* src = reinterpret_cast<const uint4*>(runs + q0 * kRun);
            uint4* dst = reinterpret_cast<uint4*>(place_lds);
            for (uint32_t i = e; i < qn * (kRun / 4u); i += 1024u) dst[i] = src[i];
        }
        __syncthreads();
        if (pad) continue;
        for (uint32_t qq = 0; qq < qn; ++qq) {
            const uint32_t q = q0 + qq;
            if (q == r) continue;
            const uint32_t* run = place_lds + qq * kRun;
            // earlier runs: keys <= cls_hi sort before mine (same class, smaller row); later runs: keys < cls_lo
            if (q > r && cls_lo == 0u) continue;                   // nothing sorts before class 0
            const uint32_t bound = q < r ? cls_hi : cls_lo - 1u;   // count keys <= bound
            uint32_t lo = 0, hi = kRun;
#pragma unroll
            for (int it = 0; it < 11; ++it)                        // (eleven halvings empty a range of 1,024)
                if (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (run[mid] <= bound) lo = mid + 1u; else hi = mid; }
            pos += lo;
        }
    }
    if (pad) {                                                     // padding: kNone behind the last row (as many padding keys in the run as rows >= n)
        const uint32_t row = r * kRun + e;
        if (row >= n && row < n_pad) cperm[row] = kNone;
    } else cperm[pos] = r * kRun + (key & 1023u);
}

// MODE 0: make the keys of chunk blockIdx.x and sort it (merge steps 2 .. C); MODE 1: the in-chunk strides (C / 2 .. 1) of merge step `size`.
// `last`: the chunk is in its final order -> cperm; else -> keys (global scratch)
template <int MODE>
__global__ __launch_bounds__(kOrdNT)
void stage_counts_sort_kernel(const float* __restrict__ cscale, uint32_t n, uint32_t n_pad, uint32_t C, uint32_t size, int last,
                              unsigned long long* __restrict__ keys, uint32_t* __restrict__ cperm)
{
    extern __shared__ unsigned long long ord_lk[];
    const uint32_t base = blockIdx.x * C;
    if (MODE == 0) {
        for (uint32_t e = threadIdx.x; e < C; e += kOrdNT) { const uint32_t r = base + e; ord_lk[e] = r < n ? order_key(cscale[r], r) : ~0ull; }
        __syncthreads();
        for (uint32_t sz = 2u; sz <= C; sz <<= 1) order_lds_passes(ord_lk, C, base, sz, sz >> 1);
    } else {
        for (uint32_t e = threadIdx.x; e < C; e += kOrdNT) ord_lk[e] = keys[base + e];
        __syncthreads();
        order_lds_passes(ord_lk, C, base, size, C >> 1);
    }
    if (last) { for (uint32_t e = threadIdx.x; e < C; e += kOrdNT) if (base + e < n_pad) cperm[base + e] = (uint32_t)ord_lk[e]; }     // (padding keys: low word = kNone)
    else for (uint32_t e = threadIdx.x; e < C; e += kOrdNT) keys[base + e] = ord_lk[e];
}
// one chunk-crossing stride of merge step `size`
__global__ __launch_bounds__(256)
void stage_counts_sort_global_kernel(unsigned long long* __restrict__ keys, uint32_t half, uint32_t size, uint32_t stride)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= half) return;
    const uint32_t lo = 2u * e - (e & (stride - 1u)), hi = lo + stride;
    const bool up = ((lo & size) == 0u);
    const unsigned long long a = keys[lo], b = keys[hi];
    if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
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
    // (the ordered tiles are written by the gather kernel behind the sort: until then their first 16 of the >= 128 bytes a row has there
    // are the sort's key scratch)
    {
        const uint32_t n_pad = n_tiles * 32u;
        if (n_pad <= 65536u) {
            // runs of 1,024 rows + placement by counting (scratch: 4 bytes per row of the padded runs)
            const uint32_t n_runs = (n_pad + kRun - 1u) / kRun;
            uint32_t* runs = reinterpret_cast<uint32_t*>(tiledp);
            hipLaunchKernelGGL(stage_counts_runs_kernel, dim3(n_runs), dim3(512), 0, st, cscale, n, runs);
            {   // (per launch: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive several)
                const hipError_t e = hipFuncSetAttribute((const void*)stage_counts_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPlaceGroup * kRun * 4u));
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(stage_counts_place_kernel, dim3(n_runs), dim3(1024), (n_runs < kPlaceGroup ? n_runs : kPlaceGroup) * kRun * 4u, st, runs, n_runs, n, n_pad, cperm);
        } else {
        uint32_t N2 = 64u; while (N2 < n_pad) N2 <<= 1;
        const uint32_t C = N2 < 16384u ? N2 : 16384u;
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(tiledp);
        {
            hipError_t e = hipFuncSetAttribute((const void*)stage_counts_sort_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stage_counts_sort_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(stage_counts_sort_kernel<0>, dim3(N2 / C), dim3(kOrdNT), C * 8u, st, cscale, n, n_pad, C, 0u, N2 == C ? 1 : 0, keys, cperm);
        for (uint32_t size = 2u * C; size <= N2 && size > C; size <<= 1) {
            for (uint32_t stride = size >> 1; stride >= C; stride >>= 1)
                hipLaunchKernelGGL(stage_counts_sort_global_kernel, dim3((N2 / 2u + 255u) / 256u), dim3(256), 0, st, keys, N2 / 2u, size, stride);
            hipLaunchKernelGGL(stage_counts_sort_kernel<1>, dim3(N2 / C), dim3(kOrdNT), C * 8u, st, cscale, n, n_pad, C, size, size == N2 ? 1 : 0, keys, cperm);
        }
        }
    }
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
// none of the three: the tail re-scores three rows and certifies against d3 -- a handful of exact scans where the two-list kernel
// needs hundreds.  R3DM_COUNTS_TWO_LISTS=1 in the developer build runs l2_knn2_counts_kernel instead
// (tools/counts_one_list_probe.py).  (hipcc 7.2 folds repeated __builtin_amdgcn_permlane32_swap calls with different operands into
// one -- wrong code -- hence the inline assembly with its own wait states below.)
// d3 = min over (a) the keys pushed out of, or never into, the three -- exactly -- and (b) the lower bounds of the quads that were
// skipped (>= d2 at the time, so >= every d2 since).
//
typedef const __attribute__((address_space(4))) float* cf32p;         // constant address space -> SMEM loads (a tile's sixteen quad summaries)

// v_permlane32_swap_b32: the upper lane half of `a` <-> the lower lane half of `b`   (a' = [a.lo | b.lo], b' = [a.hi | b.hi])
// (s_nop 1 first: the instruction needs two wait states behind a VALU write of either operand -- the compiler inserts them for its
//  own builtin and cannot for an asm statement)
__device__ __forceinline__ void swap_lane_halves(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

// The list is held as PACKED words (round 5; round 4 kept four float keys and three row indices per query: thirteen VALU
// instructions per push -- four medians, three comparisons, six selects -- and 72.3 ms where this form takes 63.6 on 3,160 pairs of
// 8,192 rows).  A wave of 64 lists takes the push path for
// ~3 of the 8 quads of every tile -- 64 queries x a top-3 list change with probability 3 / (rows seen) per row: the share is the
// statistics of the problem, not a loose bound.  Here a key carries its row in its own low bits: key' = key + ||b||^2 / (2 s_b) is
// the squared distance in key units (>= 0 up to rounding, so its float bits order like integers), the low B = ceil(log2(padded
// rows)) mantissa bits are replaced by the row index (v_and_or_b32), and the list is four v_med3_i32 / v_min_i32 on such words:
// five instructions per key instead of thirteen, no index registers.  What the truncation costs is resolution between keys closer
// than 2^(B-23) relative (2^-10 at 8,192 rows): which of two such rows is nominated is then decided by their indices, and the bounds
// handed to the tail are the words with the index bits cleared -- truncated DOWN, so every certification stays a proof; a query
// whose runner-up is that close to its third- and fourth-best rows goes to the exact scan as before, just more often
// (tests/test_gpu_count_tiles.py counts them).
struct Top3p { int d0, d1, d2, d3; };
__device__ __forceinline__ int imed3(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int imin2(int a, int b) { int r; asm("v_min_i32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void top3p_push(Top3p& s, float key, uint32_t idx, int nmask)
{
    int k;                                                  // (key & nmask) | (idx & ~nmask): one v_bfi_b32 ("tile -1" has rows below zero: +inf keys, any index)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(k) : "v"(nmask), "v"(key), "s"(idx));
    s.d3 = imed3(s.d2, s.d3, k);                            // what falls out of the three (d2 <= d3 always)
    s.d2 = imed3(s.d1, s.d2, k);
    s.d1 = imed3(s.d0, s.d1, k);
    s.d0 = imin2(s.d0, k);
}

// the four registers of an accumulator quad of both query tiles -> one query per lane: four v_permlane32_swap_b32 behind ONE
// pair of wait states (the swaps touch disjoint registers; what the wait covers is a VALU write of an operand just before
//  -- as four statements, the first with the wait states: one statement with eight in-out operands makes the register allocator
//  shuffle the accumulator registers through copies; tests/test_build_hazards.py checks in the built code that no swap has a VALU
//  write of one of its operands in the two instructions before it)
__device__ __forceinline__ void swap_lane_halves_first(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_lane_halves_next(float& a, float& b) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// min of four products that are all <= +0 (counts are non-negative and the query fragments carry the sign): on such floats the
// UNSIGNED integer order is the reversed float order (+0 < -0 < ... < -inf), so the minimum is one v_max3_u32 + one v_max_u32 --
// integer instructions, which the compiler emits itself (a float minimum of matrix results costs a canonicalising v_max each, and the
// inline-assembly v_min3_f32 that avoids those is followed by a wait state the hazard recogniser adds behind every asm statement)
__device__ __forceinline__ float min4_nonpositive(float p0, float p1, float p2, float p3)
{
    const uint32_t a = __builtin_bit_cast(uint32_t, p0), b = __builtin_bit_cast(uint32_t, p1), c = __builtin_bit_cast(uint32_t, p2), d = __builtin_bit_cast(uint32_t, p3);
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(a, b), c), d));
}

template <int GB, int PF>
__device__ __forceinline__ void counts_tile_step_p(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rr, uint32_t voffA, uint32_t voffR,
                                                   uint32_t soffA, uint32_t soffR, f32x4 (&abuf)[PF], float& rowv_load, float rowv_prev, cf32p sum_prev,
                                                   const f32x4 (&bq)[2][GB], float cql, float cbl, int nmask,
                                                   f32x16 (&cur)[2], f32x16 (&prev)[2], Top3p& st, uint32_t prev_rowbase)
{
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int rv = __builtin_bit_cast(int, rowv_prev);
    float sm[16];                                          // one s_load_dwordx16 at the top of the step: in flight behind the first MFMAs
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[k] = sum_prev[k];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const f32x4 a = abuf[g % PF];
        // block g + PF of the stream: 1 KiB each; three of four offsets ride in the instruction's immediate field
        abuf[g % PF] = bload16(ra, voffA + (uint32_t)(g & 3) * 1024u, soffA + (uint32_t)(g >> 2) * 4096u);
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
                    if (k == 0) swap_lane_halves_first(lo, hi); else swap_lane_halves_next(lo, hi);
                    prev[0][4 * qd + k] = lo;               // rows 8 qd + k      (lane half 0 of both tiles)
                    prev[1][4 * qd + k] = hi;               // rows 8 qd + 4 + k  (lane half 1 of both tiles)
                }
            }
            const float p0 = prev[hp][4 * qd], p1 = prev[hp][4 * qd + 1], p2 = prev[hp][4 * qd + 2], p3 = prev[hp][4 * qd + 3];
            const float pmin = min4_nonpositive(p0, p1, p2, p3);
            const int r0 = 8 * qd + 4 * hp;
            const float n2min = sm[2 * qd + hp], smax = sm[8 + 2 * qd + hp];      // scalars (SMEM): rows r0 .. r0 + 3
            const int lb = __builtin_bit_cast(int, __builtin_fmaf(n2min, cql, __builtin_fmaf(pmin, smax, cbl)));
            const bool mine = lb < st.d2;                   // (integer order = float order from 0 up; a rounding-negative bound passes)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine) != 0ull, 0)) {
                const uint32_t rb = prev_rowbase + (uint32_t)r0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float n2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rv, r0 + k));
                    const float sa = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(rv, 32 + r0 + k));
                    const float pk = k == 0 ? p0 : (k == 1 ? p1 : (k == 2 ? p2 : p3));
                    top3p_push(st, __builtin_fmaf(n2, cql, __builtin_fmaf(pk, sa, cbl)), rb + (uint32_t)k, nmask);
                }
            }
            // a lane whose own bound did not pass: its four keys are >= lb >= d2 (then and since), whether or not the wave pushed them.
            // (One v_med3_i32(lb, d2, d3) instead -- d3 = d2 for a lane whose bound passed, which is what d3 becomes anyway when one
            //  of the keys enters -- weakens d3 whenever a loose bound passes without an entry: 60 x the exact scans, measured.)
            st.d3 = imin2(st.d3, mine ? 0x7F800000 : lb);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the keys of the last tile (nothing multiplies behind it): all 32 rows of the lane's query
__device__ __forceinline__ void counts_last_tile_p(f32x16 (&acc)[2], float rowv, float cql, float cbl, int nmask, Top3p& st, uint32_t rowbase)
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
            top3p_push(st, __builtin_fmaf(n2, cql, __builtin_fmaf(hp ? hi : lo, sa, cbl)), rowbase + (uint32_t)row, nmask);
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
    float cql, cbl;                                        // this lane's query after the swap: (tile h, column c)
    {
        uint32_t qt = qt0 + h; if (qt >= ntJ) qt = ntJ - 1;
        const uint32_t q = qt * 32u + c;
        cql = 0.5f / (q < nJ ? Jp->cscale[q] : 1.0f);      // keys are in units of 2 s_b
        cbl = (q < nJ ? Jp->norms[q] : 0.0f) * cql;         // ||b||^2 in key units: key + cb = the squared distance / (2 s_b)
    }
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
    }
    // the row index of a key lives in its low B bits; B from the padded row count of the dataset image (wave-uniform)
    const uint32_t idx_mask = (2u << (31u - (uint32_t)__builtin_clz((ntI * 32u - 1u) | 1u))) - 1u;
    const int nmask = (int)~idx_mask;
    Top3p st;
    st.d0 = st.d1 = st.d2 = 0x7FFFFFFF; st.d3 = 0x7F800000;

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
            counts_tile_step_p<GB, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, rvA, rvB, sums + (size_t)(t == 0 ? ntI : t - 1) * 16u, bq, cql, cbl, nmask, accA, accB, st, (t - 1) * 32u);
            counts_tile_step_p<GB, PF>(ra, rr, voffA, voffR, (t + 1) * tileB + PF * 1024u, (t + 1) * 256u, abuf, rvB, rvA, sums + (size_t)t * 16u, bq, cql, cbl, nmask, accB, accA, st, t * 32u);
        }
        // the last tile's keys (and one more multiply step when the tile count is odd)
        if (t < ntI) {
            counts_tile_step_p<GB, PF>(ra, rr, voffA, voffR, t * tileB + PF * 1024u, t * 256u, abuf, rvA, rvB, sums + (size_t)(t == 0 ? ntI : t - 1) * 16u, bq, cql, cbl, nmask, accA, accB, st, (t - 1) * 32u);
            counts_last_tile_p(accA, rvA, cql, cbl, nmask, st, t * 32u);
        } else {
            counts_last_tile_p(accB, rvB, cql, cbl, nmask, st, (ntI - 1) * 32u);
        }
    }
    // unpack: a word below +inf names a row of the ORDERED image in its low bits (back to keypoint order through cperm; a padding
    // row maps to kNone there) and, with those bits cleared, a lower bound of its key'; the tail works in key units without ||b||^2.
    // The tail's layout is a list per lane half and query tile: this lane's half carries the two nominees with the bound d3; the
    // other half's slot carries the third-best row as a one-row list with the same bound -- its merge then sees the third-best key
    // as the smallest un-nominated one (first certification), and its second chance re-scores the three rows against d3.
    const uint32_t* __restrict__ perm = Ip->cperm;
    auto key_of = [&](int w) -> float { return w >= 0x7F800000 ? R3DM_INF : __builtin_bit_cast(float, w & nmask) - cbl; };
    auto row_of = [&](int w) -> uint32_t { return w >= 0x7F800000 ? kNone : perm[(uint32_t)w & idx_mask]; };
    const float f0 = key_of(st.d0), f1 = key_of(st.d1), f2 = key_of(st.d2), f3 = key_of(st.d3);
    const uint32_t i0 = row_of(st.d0), i1 = row_of(st.d1), i2 = row_of(st.d2);
    Top2 st2[NJ];
    float kinv[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        uint32_t qt = qt0 + nj; if (qt >= ntJ) qt = ntJ - 1;
        const uint32_t q = qt * 32u + c;
        kinv[nj] = 2.0f * (q < nJ ? Jp->cscale[q] : 1.0f);
    }
    {
        // the partner lane (c, 1 - h) holds the list of the OTHER query tile: fetch what it has for my tile's partner slot
        const float pd2 = __shfl_xor(f2, 32), pd3 = __shfl_xor(f3, 32);
        const uint32_t pi2 = __shfl_xor(i2, 32);
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            Top2& o = st2[nj];
            if ((uint32_t)nj == h) { o.d0 = f0; o.d1 = f1; o.d2 = i2 != kNone ? f3 : f2; o.i0 = i0; o.i1 = i1; }
            else { o.d0 = pd2; o.d1 = R3DM_INF; o.d2 = pd3; o.i0 = pi2; o.i1 = kNone; if (pi2 == kNone) { o.d0 = R3DM_INF; o.d2 = pd2; } }
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
        // two query tiles per wave: one list per query (l2_knn2_counts2_kernel); variant != 0 (dataset views beyond 65,536 rows; developer knob): the two-list kernel
        case 8:  return variant ? launch_l2_counts_t<4, 2, 4>(st, P, max_nj_tiles) : launch_l2_counts2_t<4, 4>(st, P, max_nj_tiles);
        case 16: return variant ? launch_l2_counts_t<8, 2, 8>(st, P, max_nj_tiles) : launch_l2_counts2_t<8, 8>(st, P, max_nj_tiles);
        case 18: return variant ? launch_l2_counts_t<9, 2, 9>(st, P, max_nj_tiles) : launch_l2_counts2_t<9, 9>(st, P, max_nj_tiles);
        // (256 dimensions stay on the two-list kernel, one query tile per wave: l2_knn2_counts2_kernel<16, 8> -- 128 registers of query
        //  fragments -- compiles with its fragment array indexed through scratch memory, 528 bytes per lane, round 5)
        case 32: return launch_l2_counts_t<16, 1, 8>(st, P, max_nj_tiles);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace r3dm
