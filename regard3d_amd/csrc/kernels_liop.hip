// kernels_liop.hip -- LIOP descriptor (Regard3D's live descriptor, 144 x f32) on gfx950.
//
// Replaces the per-keypoint loop of Regard3DFeatures::extractLIOPFeatures
// (/root/reference/src/Regard3DFeatures.cpp:719-861, serial in the reference: the OpenMP/TBB pragmas are
// compiled out at :35-37) whose arithmetic is the vendored VLFeat routine r3d_vl_liopdesc_process
// (/root/reference/src/thirdparty/liop/vl_liop.c:465-580) with new_basic(41): 4 neighbours, 6 ordinal
// bins, radius 6, threshold 5/255 of the patch's intensity range.
//
// One wave per 41x41 patch, everything in LDS:
//   1. the 669 pixels of the circular support are ranked by intensity: bitonic sort of (order-preserving
//      float bits << 32 | scan position), 16 keys per lane in registers -- exchanges at distance < 16 are register
//      compare-exchanges, the others lane exchanges (ds_bpermute); no LDS round trips, no barriers (the LDS network this replaces
//      took ~120 k of the ~330 k cycles of a patch and its 8 KiB held the kernel at five waves per CU).  The reference sorts with its own quick sort, whose result differs
//      from any other sort only in the order of EQUAL intensities; patches with ties are therefore re-sorted
//      by one lane with that exact procedure (middle pivot, Lomuto pass, "<= 0") -- rare, and constant
//      patches short-cut to the all-zero descriptor they produce;
//   2. each rank gets its ordinal bin, 4 bilinear samples (f64, positions from host tables computed with the
//      host libm exactly as vl_liopdesc_new does), the permutation index of the sample order and the weight
//      (#pairs differing by more than the threshold); weights are small integers -> integer LDS histogram;
//   3. normalisation with the reference's float running sum (sequential) and float-stored sqrt.
// HBM traffic: 6.7 KB patch in, 576 B out per keypoint -> bound by LDS latency / sort, not HBM.

#include "r3dm_internal.hpp"
#include <type_traits>

namespace r3dm {

constexpr int kLiopSide = 41;
constexpr int kLiopPix = kLiopSide * kLiopSide;   // 1681
constexpr int kLiopSortCap = 1024;
constexpr int kLiopMaxPix = 676;                   // support pixels the exact re-sort's LDS arrays hold (new_basic(41): 669)

__device__ __forceinline__ uint32_t float_order_bits(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// The reference's quick sort (vl_liop.c:84-120: middle pivot, Lomuto pass, "value - pivot <= 0", low part first) decides the order of
// EQUAL intensities, and with it ordinal bins.  It is run by the whole wave, one recursion depth at a time: the partitions of disjoint
// ranges commute, so the segments of a depth are partitioned side by side, a lane each (exactly the reference's pass per segment), and
// their children form the next depth's list.  Same final arrangement as the depth-first original; the serial work drops from
// ~n log n element steps to the longest segment of every depth (~2 n).
// arr[i] = (class << 16) | scan position, class = rank of the pixel's intensity among the DISTINCT intensities of the patch (the sort
// only ever asks "value <= pivot value", which the classes answer exactly; 4 bytes per entry instead of 8).  arr carries 4 entries of
// slack: the Lomuto pass reads four positions ahead -- a swap writes positions `low` <= i and i only, never one still to be read.
// seg: two lists of (begin, end) pairs, 340 pairs each (a depth has at most n / 2 segments of two or more elements); cnt[2]: their lengths.
__device__ void liop_ref_qsort_wave(uint32_t* __restrict__ arr, int n, uint16_t* __restrict__ seg, uint32_t* __restrict__ cnt, uint32_t lane)
{
    constexpr int kListPairs = (kLiopMaxPix + 4) / 2;
    if (lane == 0) { seg[0] = 0; seg[1] = (uint16_t)(n - 1); cnt[0] = n >= 2 ? 1u : 0u; cnt[1] = 0u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int ci = 0;; ci ^= 1) {
        const uint32_t nseg = cnt[ci];
        if (nseg == 0u) break;
        const uint16_t* cur = seg + 2 * kListPairs * ci;
        uint16_t* nxt = seg + 2 * kListPairs * (ci ^ 1);
        for (uint32_t sg = lane; sg < nseg; sg += 64u) {
            const int begin = cur[2 * sg], end = cur[2 * sg + 1];
            const int pivot = (end + begin) / 2;
            uint32_t t = arr[pivot]; arr[pivot] = arr[end]; arr[end] = t;
            const uint32_t pv = arr[end] >> 16;
            int low = begin;
            for (int i = begin; i < end; i += 4) {
                uint32_t e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = arr[i + j];                 // (positions >= end are not used)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i + j < end && (e[j] >> 16) <= pv) {
                        if (low != i + j) { const uint32_t o = arr[low]; arr[i + j] = o; arr[low] = e[j]; }
                        ++low;
                    }
                }
            }
            t = arr[low]; arr[low] = arr[end]; arr[end] = t;
            // children of two or more elements (a one-element range is a no-op in the reference too)
            if (low + 1 < end) { const uint32_t k = atomicAdd(&cnt[ci ^ 1], 1u); nxt[2 * k] = (uint16_t)(low + 1); nxt[2 * k + 1] = (uint16_t)end; }
            if (begin < low - 1) { const uint32_t k = atomicAdd(&cnt[ci ^ 1], 1u); nxt[2 * k] = (uint16_t)begin; nxt[2 * k + 1] = (uint16_t)(low - 1); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) cnt[ci] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// 4-element version of the same quick sort (neighbour samples with equal intensities)
__device__ void liop_ref_qsort4(const float (&v)[4], int (&p)[4])
{
    int stack[8]; int sp = 0;
    stack[sp++] = 0; stack[sp++] = 3;
    while (sp > 0) {
        const int end = stack[--sp], begin = stack[--sp];
        const int pivot = (end + begin) / 2;
        int t = p[pivot]; p[pivot] = p[end]; p[end] = t;
        int low = begin;
        for (int i = begin; i < end; ++i)
            if (v[p[i]] - v[p[end]] <= 0.0f) { t = p[low]; p[low] = p[i]; p[i] = t; ++low; }
        t = p[low]; p[low] = p[end]; p[end] = t;
        if (low < end) { stack[sp++] = low + 1; stack[sp++] = end; }
        if (low > begin) { stack[sp++] = begin; stack[sp++] = low - 1; }
    }
}

// ascending bitonic sort of 1024 distinct u64 keys held 16 per lane (key i in lane i / 16, slot i % 16)
__device__ __forceinline__ void liop_sort1024(unsigned long long (&k)[16], uint32_t lane)
{
    const uint32_t base = lane * 16u;
    for (uint32_t size = 2; size <= 1024u; size <<= 1) {
        for (uint32_t stride = size >> 1; stride >= 16u; stride >>= 1) {           // partner in another lane
            const int lane_xor = (int)(stride >> 4);
            const bool keep_min = ((base & stride) == 0u) == ((base & size) == 0u);   // slot bits are below 16: the lane decides
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)k[s], lane_xor);
                const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(k[s] >> 32), lane_xor);
                const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
                k[s] = keep_min ? (o < k[s] ? o : k[s]) : (o > k[s] ? o : k[s]);
            }
        }
#pragma unroll
        for (int ST = 8; ST >= 1; ST >>= 1) {                                      // partner in this lane
            if ((uint32_t)ST < size) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if ((s & ST) == 0) {
                        const bool up = ((base + (uint32_t)s) & size) == 0u;
                        const unsigned long long a = k[s], b = k[s | ST];
                        const bool gt = a > b;
                        k[s] = (gt == up) ? b : a;
                        k[s | ST] = (gt == up) ? a : b;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// patch extraction: cv::warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_CONSTANT 0) + cv::GaussianBlur(sigma 1.2,
// 11 taps, BORDER_REFLECT_101) of extractLIOPFeatures (/root/reference/src/Regard3DFeatures.cpp:768-808), restated
// from OpenCV 4.0's scalar code paths (OpenCV is external: this sub-stage is parity-unpinned).
// The 2x3 matrices are built on the host (float/double libm arithmetic of the reference, lines 790-799).  Fixed-point source
// coordinates exactly like hal::warpAffine: 10 fractional bits, rounded to 1/32 pixel, float bilinear weights.
//
// One WAVEFRONT per patch, 27 pixels per lane (e = lane + 64 j), no workgroup barrier anywhere:
//   1. the fixed-point coordinate terms that depend on the row or on the column alone (the rint() of a double product each) are
//      computed ONCE per row / column by 41 lanes into two small LDS tables, not once per pixel;
//   2. the bilinear gather (four global loads per pixel, nine pixels' worth in flight) lands in an LDS image whose rows carry the
//      five reflected columns on either side, so the row filter is eleven reads at consecutive addresses -- no border rule per tap;
//   3. the row-filtered values stay in registers until every lane has read its taps, then go back to the same LDS bytes as an
//      image with five reflected ROWS above and below for the column filter.
// Per value exactly the float operations of the reference's two filter passes in their order (no contraction), so the patch is
// bit-identical to the previous one-workgroup-per-patch kernel (which spent most of its ~8 k wave instructions per patch on the
// reflect loops, the per-pixel rint()s and the divisions by 41).
// ------------------------------------------------------------------------------------------------
constexpr int kLiopWS = 52;                        // row stride of the column-padded warped patch: 5 + 41 + 5 (+ 1: odd multiple of 4 banks)
constexpr int kLiopPS = 43;                        // row stride of the zero-ringed patch the descriptor samples (see liop_kernel)
constexpr int kLiopBuf = kLiopSide * kLiopWS;      // 2132 floats >= 51 x 41 (row-padded image) and >= 43 x 43 (ringed patch)
struct LiopWaveLds {
    float buf[kLiopBuf];
    int2 ty[kLiopSide];                            // per row y:    (X0, Y0) = rint((M1 y + M2) 1024) + 16, rint((M4 y + M5) 1024) + 16
    int2 tx[kLiopSide];                            // per column x: rint(M0 x 1024), rint(M3 x 1024)
};
#define LIOP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The pixel of step j of a lane is e = lane + 64 j = (y, x): 64 = 41 + 23, so a step advances (y, x) by (1, 23) with a carry.  Every
// phase starts from a laundered copy of the lane id: the index arithmetic of the 27 steps depends on the lane only, and left visible
// the compiler hoists all of it out of the per-patch loop and keeps ~400 registers of addresses alive (467 spilled VGPRs at three
// waves per SIMD).
__device__ __forceinline__ uint32_t liop_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
#define LIOP_FIRST(y, x, ln) uint32_t y = (ln) >= (uint32_t)kLiopSide ? 1u : 0u, x = (ln) - y * (uint32_t)kLiopSide
#define LIOP_NEXT(y, x) do { x += 23u; y += 1u; if (x >= (uint32_t)kLiopSide) { x -= (uint32_t)kLiopSide; y += 1u; } } while (0)

// out[j] = blurred patch value of pixel e = lane + 64 j (e < 1681).  L is this wavefront's own; on return its buf holds the
// row-filtered image (free to overwrite once the caller has synchronised the wave).
__device__ __forceinline__ void liop_make_patch(const float* __restrict__ image, int w, int h, const float* __restrict__ M6,
                                                const float (&kern)[11], LiopWaveLds& L, uint32_t lane, float (&out)[27])
{
    constexpr int S = kLiopSide;
    if (lane < (uint32_t)S) {
        double M[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = (double)M6[k];
        const int t = (int)lane;
        L.ty[t] = make_int2((int)rint((M[1] * t + M[2]) * 1024) + 16, (int)rint((M[4] * t + M[5]) * 1024) + 16);
        L.tx[t] = make_int2((int)rint(M[0] * t * 1024), (int)rint(M[3] * t * 1024));
    }
    LIOP_WAVE_SYNC();
    // ---- warp: kWarpGroups groups of kWarpPer pixels per lane, the loads of a group (4 per pixel) issued before the first is used.  X and Y are monotone in the
    // row and in the column (sums of two rounded monotone terms), so their extremes over the patch are at its four corners: when all
    // four lie inside the image (almost every keypoint) no tap of the patch needs a border test or a clamp.
    bool inside;
    {
        const int2 y0 = L.ty[0], y1 = L.ty[S - 1], x0 = L.tx[0], x1 = L.tx[S - 1];
        inside = w <= 32768 && h <= 32768;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int2 a = (c & 2) ? y1 : y0, b = (c & 1) ? x1 : x0;
            const int sx = ((a.x + b.x) >> 5) >> 5, sy = ((a.y + b.y) >> 5) >> 5;
            inside = inside && sx >= 0 && sx + 1 < w && sy >= 0 && sy + 1 < h;
        }
    }
    constexpr int kWarpGroups = 2, kWarpPer = 14;          // 28 >= 27 steps; 56 gathers in flight per lane
    auto warp = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const uint32_t ln = liop_opaque(lane);
        LIOP_FIRST(y, x, ln);
#pragma unroll 1
        for (int g = 0; g < kWarpGroups; ++g) {
            float v0[kWarpPer], v1[kWarpPer], v2[kWarpPer], v3[kWarpPer], fxy[kWarpPer][2];
            int at[kWarpPer];
#pragma unroll
            for (int q = 0; q < kWarpPer; ++q) {
                const bool live = y < (uint32_t)S;                      // e < 1681 (the last steps of a lane can leave the patch)
                const uint32_t yc = live ? y : 0u;
                const int2 a = L.ty[yc], b = L.tx[x];
                const int X = (a.x + b.x) >> 5, Y = (a.y + b.y) >> 5;
                int sx = X >> 5, sy = Y >> 5;
                fxy[q][0] = (float)(X & 31) * (1.f / 32); fxy[q][1] = (float)(Y & 31) * (1.f / 32);
                if (FAST) {
                    const float* __restrict__ p0 = image + ((uint32_t)sy * (uint32_t)w + (uint32_t)sx);
                    const float* __restrict__ p1 = p0 + w;
                    v0[q] = p0[0]; v1[q] = p0[1]; v2[q] = p1[0]; v3[q] = p1[1];
                } else {
                    sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
                    sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
                    const bool x0 = (unsigned)sx < (unsigned)w, x1 = (unsigned)(sx + 1) < (unsigned)w;
                    const bool y0 = (unsigned)sy < (unsigned)h, y1 = (unsigned)(sy + 1) < (unsigned)h;
                    const uint32_t cx0 = x0 ? (uint32_t)sx : 0u, cx1 = x1 ? (uint32_t)(sx + 1) : 0u;
                    const uint32_t r0 = (y0 ? (uint32_t)sy : 0u) * (uint32_t)w, r1 = (y1 ? (uint32_t)(sy + 1) : 0u) * (uint32_t)w;
                    const float a0 = image[r0 + cx0], a1 = image[r0 + cx1], a2 = image[r1 + cx0], a3 = image[r1 + cx1];
                    v0[q] = (x0 && y0) ? a0 : 0.f; v1[q] = (x1 && y0) ? a1 : 0.f; v2[q] = (x0 && y1) ? a2 : 0.f; v3[q] = (x1 && y1) ? a3 : 0.f;
                }
                at[q] = live ? (int)(y * (uint32_t)kLiopWS + x) : -1;
                LIOP_NEXT(y, x);
            }
#pragma unroll
            for (int q = 0; q < kWarpPer; ++q) {
                if (at[q] < 0) continue;
                const float fx = fxy[q][0], fy = fxy[q][1];
                const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                const float v = v0[q] * w0 + v1[q] * w1 + v2[q] * w2 + v3[q] * w3;
                L.buf[at[q] + 5] = v;
            }
        }
    };
    if (inside) warp(std::true_type{}); else warp(std::false_type{});
    LIOP_WAVE_SYNC();
    // BORDER_REFLECT_101 of the row filter, materialised: column -k of a row holds its column k, column 40 + k its column 40 - k
    // (41 rows x 10 pad cells, copied inside LDS)
    for (uint32_t c = lane; c < (uint32_t)(S * 10); c += 64u) {
        const uint32_t r = c / 10u, k = c - r * 10u;                                // k = 0..4: left pads, 5..9: right pads
        const uint32_t dst = k < 5u ? 4u - k : 46u + (k - 5u);                      // padded columns 4, 3, .., 0 and 46 .. 50
        const uint32_t src = k < 5u ? 6u + k : 44u - (k - 5u);                      // columns 1 .. 5 (padded 6 .. 10) and 39 .. 35 (padded 44 .. 40)
        L.buf[r * (uint32_t)kLiopWS + dst] = L.buf[r * (uint32_t)kLiopWS + src];
    }
    LIOP_WAVE_SYNC();
    // ---- row filter (RowFilter: s = k0 v[x-5]; s += k_k v[x+k-5], k = 1..10) into registers
    float rv[27];
    {
        const uint32_t ln = liop_opaque(lane);
        LIOP_FIRST(y, x, ln);
#pragma unroll
        for (int j0 = 0; j0 < 27; j0 += 3) {
            float s3[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float* __restrict__ q = L.buf + ((y < (uint32_t)S ? y : 0u) * (uint32_t)kLiopWS + x);
                float s = kern[0] * q[0];
#pragma unroll
                for (int k = 1; k < 11; ++k) s += kern[k] * q[k];
                s3[u] = s;
                LIOP_NEXT(y, x);
            }
            // the three sums are FINISHED here: left free, the compiler issues the 297 reads of the unrolled loop first, sinks the
            // arithmetic behind the next phase and spills the taps in between (467 spilled VGPRs)
            asm volatile("" : "+v"(s3[0]), "+v"(s3[1]), "+v"(s3[2]) :: "memory");
            rv[j0] = s3[0]; rv[j0 + 1] = s3[1]; rv[j0 + 2] = s3[2];
        }
    }
    LIOP_WAVE_SYNC();
    // ---- the row-filtered image with five reflected rows above and below: row 5 + y at (5 + y) * 41
    {
        const uint32_t ln = liop_opaque(lane);
        LIOP_FIRST(y, x, ln);
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            if (y < (uint32_t)S) L.buf[(5u + y) * (uint32_t)S + x] = rv[j];
            LIOP_NEXT(y, x);
        }
    }
    LIOP_WAVE_SYNC();
    for (uint32_t c = lane; c < (uint32_t)(S * 10); c += 64u) {
        const uint32_t k = c / (uint32_t)S, xx = c - k * (uint32_t)S;               // k = 0..4: rows above, 5..9: rows below
        const uint32_t dst = k < 5u ? 4u - k : 46u + (k - 5u);                      // padded rows 4 .. 0 and 46 .. 50
        const uint32_t src = k < 5u ? 6u + k : 44u - (k - 5u);                      // rows 1 .. 5 (padded 6 .. 10) and 39 .. 35 (padded 44 .. 40)
        L.buf[dst * (uint32_t)S + xx] = L.buf[src * (uint32_t)S + xx];
    }
    LIOP_WAVE_SYNC();
    // ---- column filter (SymmColumnFilter: s = k5 r[y]; s += k_{5+j} (r[y+j] + r[y-j]), j = 1..5)
    {
        const uint32_t ln = liop_opaque(lane);
        LIOP_FIRST(y, x, ln);
#pragma unroll
        for (int j0 = 0; j0 < 27; j0 += 3) {
            float s3[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float* __restrict__ q = L.buf + (((y < (uint32_t)S ? y : 0u) + 5u) * (uint32_t)S + x);
                float s = kern[5] * q[0];
#pragma unroll
                for (int t = 1; t <= 5; ++t) s += kern[5 + t] * (q[t * S] + q[-t * S]);
                s3[u] = s;
                LIOP_NEXT(y, x);
            }
            asm volatile("" : "+v"(s3[0]), "+v"(s3[1]), "+v"(s3[2]) :: "memory");
            out[j0] = s3[0]; out[j0 + 1] = s3[1]; out[j0 + 2] = s3[2];
        }
    }
}

// lexicographic index of a permutation of (0, 1, 2, 3) (Lehmer code), as vl_liop.c:541-551 counts it
__device__ __forceinline__ uint32_t liop_lehmer(const int (&np)[4])
{
    const int c0 = (np[1] < np[0]) + (np[2] < np[0]) + (np[3] < np[0]);
    const int c1 = (np[2] < np[1]) + (np[3] < np[1]);
    const int c2 = (np[3] < np[2]);
    return (uint32_t)(c0 * 6 + c1 * 2 + c2);
}
// the permutation index of four DISTINCT samples from their six comparisons: bit 0 = v0 < v1, 1 = v0 < v2, 2 = v0 < v3, 3 = v1 < v2,
// 4 = v1 < v3, 5 = v2 < v3.  rank(k) = number of samples below sample k; np[r] = the sample of rank r (the unique ascending order).
// Patterns no four numbers produce (cycles) map to 0 and are never looked up.
__device__ __forceinline__ uint32_t liop_perm_index(uint32_t key)
{
    const int b01 = key & 1, b02 = (key >> 1) & 1, b03 = (key >> 2) & 1, b12 = (key >> 3) & 1, b13 = (key >> 4) & 1, b23 = (key >> 5) & 1;
    const int rk[4] = {(1 - b01) + (1 - b02) + (1 - b03), b01 + (1 - b12) + (1 - b13), b02 + b12 + (1 - b23), b03 + b13 + b23};
    if ((1 << rk[0] | 1 << rk[1] | 1 << rk[2] | 1 << rk[3]) != 15) return 0u;
    int np[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) np[r] = (rk[0] == r) ? 0 : (rk[1] == r) ? 1 : (rk[2] == r) ? 2 : 3;
    return liop_lehmer(np);
}

struct LiopParams {
    const float* patches;      // [n][41*41] (FUSED = false)
    const int*   pix;          // [n_pix] offsets of the circular support (scan order) in the ZERO-RINGED patch: (x + 1) + (y + 1) * 43
    const double2* samp_w;     // [n_pix][4] (wx, wy): fractional parts of the four sample positions of every support pixel
    const int4*  samp_off;     // [n_pix]    offsets of their top-left taps in the ringed patch: (floor(x) + 1) + (floor(y) + 1) * 43
    uint32_t n, n_pix;
    float* desc;               // [n][144]
    uint32_t* n_tie_patches;   // counts the patches with equal intensities in their support (they took the reference's exact re-sort)
    uint32_t* tie_list;        // (unused since the exact sort runs inside the patch's own wavefront)
    // FUSED = true: the patch is warped + blurred from the image by the wavefront itself (liop_make_patch) and never exists in HBM
    const float* image0; int w, h;
    const float* M6;           // [n][6] inverse maps
    const float* kern;         // [11] blur taps
    const uint32_t* img_of;    // [n] image plane of every keypoint (or nullptr: plane 0)
};

// One wavefront per patch; 13.5 KB of LDS per one-wave workgroup, 11 of them per CU.
// The patch lives in LDS with a ring of zeros around it (43 x 43): the four taps of a bilinear sample are then always readable --
// vl_liop's `if (ix >= 0 && ...)` guards become reads of a 0.0f -- and the floor / fraction of every sample position, which depend
// on the support pixel only, come from host tables (computed with the reference's double operations) instead of f64 -> i64
// conversions per sample.
// A patch with equal intensities in its support is re-ordered by the reference's own quick sort right here, by its own wavefront (a
// wavefront IS a patch: nothing diverges).  Until round 5 such patches went to a second launch that extracted and sorted them again
// (a third of the LIOP time on the stage's photographs); the exact sort now works on intensity CLASSES, 4 bytes per entry, in LDS the
// patch phase has finished with -- no extra LDS, no second pass.
// LDS map in 32-bit words: [0, 2132) warp / filter image, then the ringed patch [0, 1849); [2132, 2296) coordinate tables;
// [2296, 2634) perm (u16); [2634, 3209) spare.  Exact sort (ties only): entries [1849, 2529), segment lists [2529, 3209).
constexpr int kLiopLdsWords = 3209;
constexpr int kLiopPermWord = 2296, kLiopQArrWord = 1849, kLiopQSegWord = 2529;
static_assert(sizeof(LiopWaveLds) == 4 * kLiopPermWord, "LDS map of liop_kernel");
static_assert(kLiopQArrWord >= kLiopPS * kLiopPS && kLiopQArrWord + kLiopMaxPix + 4 <= kLiopQSegWord && kLiopQSegWord + (kLiopMaxPix + 4) <= kLiopLdsWords, "LDS map of liop_kernel");
__device__ __forceinline__ float liop_from_order_bits(uint32_t b) { return __uint_as_float((b & 0x80000000u) ? (b ^ 0x80000000u) : ~b); }

template <bool FUSED>
__global__ __launch_bounds__(64, 3)                  // (three waves per SIMD = what 13.5 KB of LDS per wavefront admits: 168 VGPRs, not the 512 a lone wave may take)
void liop_kernel(const LiopParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLiopLdsWords];
    __shared__ uint32_t hist[144];
    __shared__ float s_norm;
    __shared__ uint32_t qcnt[2];
    __shared__ unsigned char perm_lut[64];
    LiopWaveLds& W = *reinterpret_cast<LiopWaveLds*>(lds);
    float* patch = W.buf;
    uint16_t* perm = reinterpret_cast<uint16_t*>(lds + kLiopPermWord);
    uint32_t* qarr = lds + kLiopQArrWord;
    uint16_t* qseg = reinterpret_cast<uint16_t*>(lds + kLiopQSegWord);
    perm_lut[threadIdx.x] = (unsigned char)liop_perm_index(threadIdx.x);       // (read behind the barriers of the first patch)

    const uint32_t lane0 = threadIdx.x;
    const uint32_t N = P.n_pix;
    float kern[11];
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < 11; ++k) kern[k] = P.kern[k];
    }
    for (uint32_t item = blockIdx.x; item < P.n; item += gridDim.x) {
        // (a laundered lane id per patch: nothing that depends on the lane alone -- a few hundred registers of index arithmetic over
        // the phases below -- is worth keeping alive across a whole patch, and the compiler would)
        const uint32_t lane = liop_opaque(lane0);
        {
            float v[27];
            if (FUSED) {
                const float* image = P.image0 + (P.img_of ? (size_t)P.img_of[item] * ((size_t)P.w * (size_t)P.h) : 0);
                liop_make_patch(image, P.w, P.h, P.M6 + 6 * (size_t)item, kern, W, lane, v);
                LIOP_WAVE_SYNC();                      // every lane is done with the row-filtered image: the patch goes over it
            } else {
                // 1681 floats; a patch starts at a multiple of 4 bytes only, so the vector loads are of single floats, all in flight at once
                const float* src = P.patches + (size_t)item * kLiopPix + liop_opaque(lane);
#pragma unroll
                for (int j = 0; j < 27; ++j) v[j] = (j < 26 || lane < (uint32_t)(kLiopPix - 26 * 64)) ? src[64 * j] : 0.0f;
            }
            // ring of zeros: rows 0 and 42, columns 0 and 42 of the 43 x 43 image
            for (uint32_t e = lane; e < 4u * (uint32_t)kLiopPS; e += 64u) {
                const uint32_t k = e % (uint32_t)kLiopPS, sd = e / (uint32_t)kLiopPS;
                const uint32_t o = sd == 0u ? k : sd == 1u ? 42u * (uint32_t)kLiopPS + k : sd == 2u ? k * (uint32_t)kLiopPS : k * (uint32_t)kLiopPS + 42u;
                patch[o] = 0.0f;
            }
            {
                const uint32_t ln = liop_opaque(lane);
                LIOP_FIRST(y, x, ln);
#pragma unroll
                for (int j = 0; j < 27; ++j) {
                    if (y < (uint32_t)kLiopSide) patch[(y + 1u) * (uint32_t)kLiopPS + x + 1u] = v[j];
                    LIOP_NEXT(y, x);
                }
            }
        }
        for (uint32_t e = lane; e < 144; e += 64) hist[e] = 0;
        r3dm_syncthreads();

        // ---- 1. rank the support pixels by intensity (key i: lane i / 16, slot i % 16).  -0.0f and 0.0f are one intensity to the
        // reference's comparison: v + 0.0f brings both to +0 before the order bits are taken.
        unsigned long long keys[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t i = lane * 16u + (uint32_t)s;
            if (i < N) { const float v = patch[P.pix[i]] + 0.0f; keys[s] = ((unsigned long long)float_order_bits(v) << 32) | i; }
            else keys[s] = ~0ull;
        }
        liop_sort1024(keys, lane);
        bool tie = false;
        uint32_t hi_last = 0u;                           // order bits of the largest intensity, in the lane that holds sorted position N - 1
        // the first key of the next lane, for the pair that straddles two lanes
        const uint32_t nhi = (uint32_t)__shfl_down((int)(uint32_t)(keys[0] >> 32), 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t i = lane * 16u + (uint32_t)s;
            if (i < N) perm[i] = (uint16_t)(keys[s] & 0xFFFFu);
            if (i == N - 1u) hi_last = (uint32_t)(keys[s] >> 32);
            const uint32_t next_hi = (s < 15) ? (uint32_t)(keys[s < 15 ? s + 1 : 15] >> 32) : nhi;
            if (i + 1 < N) tie |= ((uint32_t)(keys[s] >> 32) == next_hi);
        }
        const bool any_tie = __ballot(tie) != 0ull;
        const float vmin = liop_from_order_bits((uint32_t)__shfl((int)(uint32_t)(keys[0] >> 32), 0));
        const float vmax = liop_from_order_bits((uint32_t)__shfl((int)hi_last, (int)((N - 1u) >> 4)));
        r3dm_syncthreads();
        if (vmin == vmax) {
            // constant support: every weight is 0, the descriptor is 0 / max(0, 1e-12) = 0
            for (uint32_t e = lane; e < 144; e += 64) P.desc[(size_t)item * 144 + e] = 0.0f;
            r3dm_syncthreads();
            continue;
        }
        if (any_tie) {
            if (lane == 0) atomicAdd(P.n_tie_patches, 1u);
            // class of sorted position j = number of positions k <= j whose intensity differs from its predecessor's
            uint32_t cls[16];
            uint32_t run = 0;
            const uint32_t phi = (uint32_t)__shfl_up((int)(uint32_t)(keys[15] >> 32), 1);      // last key of the previous lane
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint32_t prev = s ? (uint32_t)(keys[s ? s - 1 : 0] >> 32) : phi;
                const bool first = (lane == 0u && s == 0);
                run += (!first && (uint32_t)(keys[s] >> 32) != prev) ? 1u : 0u;
                cls[s] = run;
            }
            uint32_t incl = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
            const uint32_t base = incl - run;
            // entries in SCAN order: position i holds (class of pixel i, i) -- the array the reference sorts is the scan order
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint32_t j = lane * 16u + (uint32_t)s;
                const uint32_t pos = (uint32_t)(keys[s] & 0xFFFFu);
                if (j < N) qarr[pos] = ((base + cls[s]) << 16) | pos;
            }
            if (lane < 4u) qarr[N + lane] = 0u;
            r3dm_syncthreads();
            liop_ref_qsort_wave(qarr, (int)N, qseg, qcnt, lane);
            r3dm_syncthreads();
            uint32_t q[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) { const uint32_t i = lane + 64u * (uint32_t)k; q[k] = i < N ? qarr[i] : 0u; }
            r3dm_syncthreads();                          // perm lies inside the sort's arrays
#pragma unroll
            for (int k = 0; k < 11; ++k) { const uint32_t i = lane + 64u * (uint32_t)k; if (i < N) perm[i] = (uint16_t)(q[k] & 0xFFFFu); }
            r3dm_syncthreads();
        }
        // threshold = -intensityThreshold * (max - min), all float (vl_liop.c:497-503)
        const float thr = (float)(5.0 / 255) * (vmax - vmin);

        // ---- 2. per rank: bin, 4 bilinear samples, permutation index, weight.  The sample tables of the NEXT rank of a lane are
        // requested before the current one is worked on (the kernel is latency-bound: an L2 round trip per rank otherwise).
        // bin = min(i / area, 5) by five comparisons (a runtime division costs more than the four samples' arithmetic)
        const uint32_t area = N / 6u;
        uint32_t i = lane;
        int4 so = make_int4(0, 0, 0, 0);
        double2 sw[4] = {};
        if (i < N) {
            const uint32_t p = perm[i];
            so = P.samp_off[p];
#pragma unroll
            for (int k = 0; k < 4; ++k) sw[k] = P.samp_w[4 * p + k];
        }
        while (i < N) {
            const uint32_t i2 = i + 64u;
            int4 so2 = make_int4(0, 0, 0, 0);
            double2 sw2[4] = {};
            if (i2 < N) {
                const uint32_t p2 = perm[i2];
                so2 = P.samp_off[p2];
#pragma unroll
                for (int k = 0; k < 4; ++k) sw2[k] = P.samp_w[4 * p2 + k];
            }
            const uint32_t bin = (uint32_t)(i >= area) + (uint32_t)(i >= 2u * area) + (uint32_t)(i >= 3u * area) + (uint32_t)(i >= 4u * area) + (uint32_t)(i >= 5u * area);
            const int offs[4] = {so.x, so.y, so.z, so.w};
            float nv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* __restrict__ q = patch + offs[k];
                const double wx = sw[k].x, wy = sw[k].y;
                const double a = q[0], b = q[1], c = q[kLiopPS], d = q[kLiopPS + 1];
                nv[k] = (float)((1.0 - wy) * (a + (b - a) * wx) + wy * (c + (d - c) * wx));
            }
            // order of the 4 samples: without ties the six comparisons name the permutation; its lexicographic index comes from the table
            // the lanes built at kernel start (liop_perm_index on every pattern).  Equal samples (rare) go through the reference's quick sort.
            const uint32_t key = (uint32_t)(nv[0] < nv[1]) | ((uint32_t)(nv[0] < nv[2]) << 1) | ((uint32_t)(nv[0] < nv[3]) << 2) |
                                 ((uint32_t)(nv[1] < nv[2]) << 3) | ((uint32_t)(nv[1] < nv[3]) << 4) | ((uint32_t)(nv[2] < nv[3]) << 5);
            const bool ntie = (nv[0] == nv[1]) | (nv[0] == nv[2]) | (nv[0] == nv[3]) | (nv[1] == nv[2]) | (nv[1] == nv[3]) | (nv[2] == nv[3]);
            uint32_t index = perm_lut[key];
            if (ntie) {
                int np[4] = {0, 1, 2, 3};
                liop_ref_qsort4(nv, np);
                index = liop_lehmer(np);
            }
            const float t0 = nv[0] + thr, t1 = nv[1] + thr, t2 = nv[2] + thr, t3 = nv[3] + thr;
            const uint32_t weight = (uint32_t)((nv[0] > t1) | (nv[1] > t0)) + (uint32_t)((nv[0] > t2) | (nv[2] > t0)) + (uint32_t)((nv[0] > t3) | (nv[3] > t0)) +
                                    (uint32_t)((nv[1] > t2) | (nv[2] > t1)) + (uint32_t)((nv[1] > t3) | (nv[3] > t1)) + (uint32_t)((nv[2] > t3) | (nv[3] > t2));
            if (weight) atomicAdd(&hist[bin * 24u + index], weight);
            i = i2; so = so2;
#pragma unroll
            for (int k = 0; k < 4; ++k) sw[k] = sw2[k];
        }
        r3dm_syncthreads();

        // ---- 3. normalisation: float running sum in index order, norm stored to float (vl_liop.c:567-575)
        if (lane == 0) {
            float norm = 0.0f;
            for (int e = 0; e < 144; ++e) { const float v = (float)hist[e]; norm += v * v; }
            const double r = sqrt((double)norm);
            s_norm = (float)(r > 1e-12 ? r : 1e-12);
        }
        r3dm_syncthreads();
        const float nrm = s_norm;
        for (uint32_t e = lane; e < 144; e += 64) P.desc[(size_t)item * 144 + e] = (float)hist[e] / nrm;
        r3dm_syncthreads();
    }
}

// the patches alone (r3dm_extract_liop with patches_out, tools): four wavefronts per workgroup, a patch each, no barrier
__global__ __launch_bounds__(256, 3)
void liop_extract_patches_kernel(const float* __restrict__ image0, int w, int h, const float* __restrict__ M6,
                                 const float* __restrict__ kern_g /* 11 taps */, uint32_t n, float* __restrict__ patches,
                                 const uint32_t* __restrict__ img_of)
{
    __shared__ __attribute__((aligned(16))) LiopWaveLds W4[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    LiopWaveLds& W = W4[wave];
    float kern[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) kern[k] = kern_g[k];
    for (uint32_t item = blockIdx.x * 4u + wave; item < n; item += gridDim.x * 4u) {
        // keypoints of a batch of same-size images in one launch: the keypoint's own image plane
        const float* __restrict__ image = image0 + (img_of ? (size_t)img_of[item] * ((size_t)w * (size_t)h) : 0);
        float v[27];
        liop_make_patch(image, w, h, M6 + 6 * (size_t)item, kern, W, lane, v);
        float* out = patches + (size_t)item * kLiopPix + liop_opaque(lane);
#pragma unroll
        for (int j = 0; j < 27; ++j) if (j < 26 || lane < (uint32_t)(kLiopPix - 26 * 64)) out[64 * j] = v[j];
        LIOP_WAVE_SYNC();                              // the next patch's tables and image go over this one's
    }
}

hipError_t launch_liop_extract(hipStream_t st, const float* image, int w, int h, const float* M6, const float* kern,
                               uint32_t n, float* patches, const uint32_t* img_of)
{
    if (n == 0) return hipSuccess;
    if ((uint64_t)w * (uint64_t)h >= (1ull << 32)) return hipErrorInvalidValue;       // 32-bit pixel offsets inside a plane
    const uint32_t wgs = (n + 3u) / 4u;
    const uint32_t grid = wgs < 16384u ? wgs : 16384u;
    hipLaunchKernelGGL(liop_extract_patches_kernel, dim3(grid), dim3(256), 0, st, image, w, h, M6, kern, n, patches, img_of);
    return hipGetLastError();
}

// n_tie_patches: one zeroed word (receives the number of patches that needed the exact re-sort); tie_list: n words of scratch
hipError_t launch_liop(hipStream_t st, const LiopTables& T, const float* patches, uint32_t n, float* desc, uint32_t* n_tie_patches, uint32_t* tie_list)
{
    if (n == 0) return hipSuccess;
    if (T.n_pix < 2 || T.n_pix > (uint32_t)kLiopMaxPix) return hipErrorInvalidValue;
    LiopParams P{};
    P.patches = patches; P.pix = T.pix; P.samp_w = reinterpret_cast<const double2*>(T.samp_w); P.samp_off = reinterpret_cast<const int4*>(T.samp_off);
    P.n = n; P.n_pix = T.n_pix; P.desc = desc; P.n_tie_patches = n_tie_patches; P.tie_list = tie_list;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL((liop_kernel<false>), dim3(grid), dim3(64), 0, st, P);
    return hipGetLastError();
}

// keypoints -> descriptors in one kernel: the warp + blur of extractLIOPFeatures inside the descriptor's wavefront
hipError_t launch_liop_fused(hipStream_t st, const LiopTables& T, const float* image, int w, int h, const float* M6, const float* kern,
                             const uint32_t* img_of, uint32_t n, float* desc, uint32_t* n_tie_patches, uint32_t* tie_list)
{
    if (n == 0) return hipSuccess;
    if (T.n_pix < 2 || T.n_pix > (uint32_t)kLiopMaxPix) return hipErrorInvalidValue;
    if ((uint64_t)w * (uint64_t)h >= (1ull << 32)) return hipErrorInvalidValue;
    LiopParams P{};
    P.pix = T.pix; P.samp_w = reinterpret_cast<const double2*>(T.samp_w); P.samp_off = reinterpret_cast<const int4*>(T.samp_off);
    P.n = n; P.n_pix = T.n_pix; P.desc = desc; P.n_tie_patches = n_tie_patches; P.tie_list = tie_list;
    P.image0 = image; P.w = w; P.h = h; P.M6 = M6; P.kern = kern; P.img_of = img_of;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL((liop_kernel<true>), dim3(grid), dim3(64), 0, st, P);
    return hipGetLastError();
}

}  // namespace r3dm
