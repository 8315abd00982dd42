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
