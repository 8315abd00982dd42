// kernels_match_hamming.hip -- binary descriptors (486-bit A-KAZE MLDB; BASELINE config C3): Hamming 2-NN + ratio test.
//   hamming_knn2_kernel       xor + popcount on the integer VALU, dataset rows through the scalar cache (the default)
//   l2_knn2_int_lds_kernel    bits as 0 / 1 bytes on i8 MFMA tiles shared through LDS (r3dm_set_hamming_mfma): d = |a| + |b| - 2 a.b, exact
// Replaces OpenMVG's ArrayMatcherBruteForce<uchar, Hamming> behind the call sites of /root/reference/src/R3DComputeMatches.cpp:437-489.
#include "kernels_match_common.hpp"

namespace r3dm {


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

// ------------------------------------------------------------------------------------------------
// The same workgroup-shared tiles WITHOUT a barrier in the loop (round 6): a ring of kRingK LDS slots, a sequence word and a release
// counter per slot.  The barrier version lost to the per-wave loads not by its LDS traffic (9.36 ms against 10.52 without the
// epilogue, 780 pairs) but because one s_barrier per tile makes every wave wait for the one whose lists changed (12.16 against
// 11.67 with it): here a wave in its push path stalls nobody until it is kRingK - 1 tiles behind.
//   * tile T lives in slot T % kRingK and is FETCHED by wave T % 4 of the workgroup (all of it: GB LDS-DMA blocks + the 32 norms),
//     as early as the slot is free -- i.e. when all four waves have released tile T - kRingK (rel[slot] == 4 x generation) -- and the
//     fetching wave is within kRingK - 1 tiles of it;
//   * a wave publishes a tile it fetched at the start of a LATER step of its own, behind a counted s_waitcnt vmcnt that covers the
//     tile's DMAs (the only vector-memory operations of the loop), by writing T + 1 into seq[slot]; LDS instructions of a wave execute
//     in order, so a reader that sees the word sees the tile (the rule of the guide: LDS-DMA data is ordered by the issuing wave's
//     vmcnt; the sequence word takes the place of the barrier the reader would pass);
//   * a wave reads tile t (and, through its fragment window, the first blocks of tile t + 1) after it has seen seq of tile t + 1, and
//     releases tile t with one ds_add behind its last read of it.
// Waves without query tiles fetch and release like the others; their results are discarded.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRingK = 8;
__device__ __forceinline__ uint32_t lds_read_u32(const uint32_t* p)
{
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0, int OPS = 0>
__global__ __launch_bounds__(256, WPS)
void l2_knn2_int_ring_kernel(const MatchParams P)
{
    static_assert(PF <= GB, "the fragment window runs at most one tile ahead");
    // ONE LDS array: [kRingK slots][GB KiB tile | 256 B norms], then seq[kRingK], rel[kRingK]
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
    constexpr uint32_t tileB = (uint32_t)GB * 1024u, slotB = tileB + 256u;
    uint32_t* seq = reinterpret_cast<uint32_t*>(ring_smem + kRingK * slotB);
    uint32_t* rel = seq + kRingK;
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
    const bool has_queries = qt0 < ntJ;

    if (threadIdx.x < 2u * kRingK) seq[threadIdx.x] = 0u;  // (seq and rel are adjacent)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the query fragments: from here on this wave's vmcnt counts its LDS-DMAs only
    __syncthreads();                                       // seq / rel are zero for everybody (the only barrier of the kernel)

    if (nI >= 2) {                                         // workgroup-uniform
        const unsigned char* gA = reinterpret_cast<const unsigned char*>(tilesI) + lane * 16u;
        const unsigned char* gN = reinterpret_cast<const unsigned char*>(Ip->norms) + (lane & 31u) * 4u;
        // fetch tile T into its slot: GB + 1 LDS-DMA instructions (the DMA adds lane x 16 / lane x 4 itself)
        auto fetch = [&](uint32_t T) {
            unsigned char* slot = ring_smem + (T % kRingK) * slotB;
#pragma unroll
            for (int i = 0; i < GB; ++i)
                __builtin_amdgcn_global_load_lds((glb_vp)(gA + (size_t)T * tileB + i * 1024u), (lds_vp)(slot + i * 1024u), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gN + (size_t)T * 128u), (lds_vp)(slot + tileB), 4, 0, 0);
        };
        // ---- prologue: the first kRingK - 1 tiles, each by its wave; published when landed
        uint32_t next_fetch = wave;                        // smallest tile = wave (mod 4) this wave has not fetched
        while (next_fetch < kRingK - 1u && next_fetch < ntI) { fetch(next_fetch); next_fetch += 4u; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            for (uint32_t T = wave; T < kRingK - 1u && T < ntI; T += 4u) __atomic_store_n(seq + T % kRingK, T + 1u, __ATOMIC_RELAXED);
        uint32_t pend = 0xFFFFFFFFu;                       // tile fetched in an earlier step and not yet published

        const unsigned char* lane_lds = ring_smem + lane * 16u;            // fragment of block g of slot s: + s * slotB + g * 1024
        const unsigned char* lane_nrm = ring_smem + tileB + h * 16u;       // quad qd of slot s: + s * slotB + qd * 32
        // wait until tile T is published (T < ntI)
        auto wait_tile = [&](uint32_t T) {
            const uint32_t* w = seq + T % kRingK;
            while (lds_read_u32(w) != T + 1u) __builtin_amdgcn_s_sleep(1);
        };
        wait_tile(0);
        if (ntI > 1) wait_tile(1);
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
        // this wave's duties towards the ring, at the head of step t and again in every round of a wait: publish the tile an earlier
        // call fetched (its DMAs are then a step old: the wait is a formality unless HBM served them), fetch the next tile of this
        // wave when its slot has been released by all four waves and it is within kRingK - 1 tiles.  A wave that waits keeps doing
        // both, or two waves could wait for tiles the other one holds back.
        auto service = [&](uint32_t t) {
            if (pend != 0xFFFFFFFFu) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __atomic_store_n(seq + pend % kRingK, pend + 1u, __ATOMIC_RELAXED);
                pend = 0xFFFFFFFFu;
            }
            if (next_fetch < ntI && next_fetch <= t + kRingK - 1u && lds_read_u32(rel + next_fetch % kRingK) == 4u * (next_fetch / kRingK)) {
                fetch(next_fetch);
                pend = next_fetch; next_fetch += 4u;
            }
        };
        // head of step t: tiles t and, for the fragment window, t + 1 are about to be read
        auto step_head = [&](uint32_t t) {
            service(t);
            if (t + 1u < ntI) {
                const uint32_t* w = seq + (t + 1u) % kRingK;
                while (lds_read_u32(w) != t + 2u) { __builtin_amdgcn_s_sleep(1); service(t); }
            }
        };
        auto release = [&](uint32_t t) {                   // behind this wave's last ds_read of tile t (LDS instructions execute in order)
            if (lane == 0) __atomic_fetch_add(rel + t % kRingK, 1u, __ATOMIC_RELAXED);
        };
        uint32_t t = 0;
        for (; t + 1 < ntI; t += 2) {
            step_head(t);
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + (t % kRingK) * slotB, lane_lds + ((t + 1u) % kRingK) * slotB, lane_nrm + ((t + 1u) % kRingK) * slotB,
                                                    abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
            release(t);
            step_head(t + 1u);
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + ((t + 1u) % kRingK) * slotB, lane_lds + ((t + 2u) % kRingK) * slotB, lane_nrm + ((t + 2u) % kRingK) * slotB,
                                                    abuf, nrmB, nrmA, bq, accB, accA, st, t * 32u + hb);
            release(t + 1u);
        }
        if (t < ntI) {
            step_head(t);
            int_tile_step_lds<GB, NJ, PF, ABL, OPS>(lane_lds + (t % kRingK) * slotB, lane_lds + ((t + 1u) % kRingK) * slotB, lane_nrm + ((t + 1u) % kRingK) * slotB,
                                                    abuf, nrmA, nrmB, bq, accA, accB, st, (t - 1) * 32u + hb);
            release(t);
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
        // a tile this wave fetched and nobody waited for yet is still published: other waves may be behind
        if (pend != 0xFFFFFFFFu) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __atomic_store_n(seq + pend % kRingK, pend + 1u, __ATOMIC_RELAXED);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // drain before ordinary loads follow
    }
    if (has_queries) {
        if constexpr (OPS == 0) l2_finish_queries<NJ, true>(P, pair, Ip, Jp, st, qt0, h, c, (float)(GB * 16), true);
        else hamming_finish_queries<NJ>(P, pair, Ip, Jp, st, qt0, h, c);
    }
}

template <int GB, int NJ, int PF, int WPS, int ABL = 0, int OPS = 0>
static hipError_t launch_l2_int_ring(hipStream_t st, const MatchParams& Pin, uint32_t max_nj_tiles)
{
    MatchParams P = Pin;
    const uint32_t tiles_per_wg = 4u * NJ;
    P.qb_per_pair = (max_nj_tiles + tiles_per_wg - 1) / tiles_per_wg;
    P.xcd_map = 1u;
    const uint64_t grid64 = (uint64_t)((P.n_pairs + 7u) / 8u * 8u) * P.qb_per_pair;
    if (grid64 == 0) return hipSuccess;
    if (grid64 > kMaxBlocksOf256) return hipErrorInvalidValue;
    const size_t lds = kRingK * ((size_t)GB * 1024 + 256) + 2 * kRingK * 4;
    {
        const hipError_t e = hipFuncSetAttribute((const void*)l2_knn2_int_ring_kernel<GB, NJ, PF, WPS, ABL, OPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((l2_knn2_int_ring_kernel<GB, NJ, PF, WPS, ABL, OPS>), dim3((uint32_t)grid64), dim3(256), lds, st, P);
    return hipGetLastError();
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
#ifdef R3DM_DEVTOOLS
    // A/B (tools/int_ring_ab.py ... akaze): the barrier-free ring against the barrier per tile (the product)
    static const int ring = r3dm_dev_knob("R3DM_HAMMING_RING", 0);
    if (ring && words == 16) return launch_l2_int_ring<16, 2, 4, 2, 0, 1>(st, P, max_nj_tiles);
    if (ring && words == 8) return launch_l2_int_ring<8, 2, 4, 2, 0, 1>(st, P, max_nj_tiles);
#endif
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

#ifdef R3DM_DEVTOOLS
// the bf16 integer kernel's LDS-shared variants (developer A/B of kernels_match_16bit.hip: R3DM_L2_INT_VARIANT 5 / 59 / 6)
hipError_t launch_l2_int_lds_variant(hipStream_t st, const MatchParams& P, uint32_t max_nj_tiles, int iv)
{
    if (iv == 7) return launch_l2_int_ring<8, 2, 4, 2>(st, P, max_nj_tiles);       // ... through a barrier-free ring of LDS slots
    if (iv == 79) return launch_l2_int_ring<8, 2, 4, 2, 1>(st, P, max_nj_tiles);   // ... without the epilogue (timing only)
    if (iv == 5) return launch_l2_int_lds<8, 2, 4, 2>(st, P, max_nj_tiles);        // workgroup-shared tiles through LDS-DMA
    if (iv == 59) return launch_l2_int_lds<8, 2, 4, 2, 1>(st, P, max_nj_tiles);    // ... without the epilogue (timing only)
    return launch_l2_int_lds<8, 2, 8, 2>(st, P, max_nj_tiles);                     // ... whole-tile fragment window
}
#endif

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

}  // namespace r3dm
