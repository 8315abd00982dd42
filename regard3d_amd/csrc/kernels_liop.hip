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

// exact emulation of the reference's quick sort (lane 0 only) over arr[0..n) = (intensity bits, scan position), ordered by
// intensity.  One LDS array of pairs instead of perm[] + val[perm[]] (no dependent second load), and the Lomuto pass reads four
// positions ahead: a swap writes positions `low` <= i and i only, never one that is still to be read, so the look-ahead is exact.
__device__ void liop_ref_qsort(uint2* __restrict__ arr, int n, uint16_t* __restrict__ stack)
{
    int sp = 0;
    stack[sp++] = 0; stack[sp++] = (uint16_t)(n - 1);
    while (sp > 0) {
        const int end = stack[--sp], begin = stack[--sp];
        const int pivot = (end + begin) / 2;
        uint2 t = arr[pivot]; arr[pivot] = arr[end]; arr[end] = t;
        const float pv = __uint_as_float(arr[end].x);
        int low = begin;
        for (int i = begin; i < end; i += 4) {
            uint2 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = arr[i + j];                     // (arr carries 4 entries of slack)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i + j < end && __uint_as_float(e[j].x) - pv <= 0.0f) {
                    if (low != i + j) { const uint2 o = arr[low]; arr[i + j] = o; arr[low] = e[j]; }
                    ++low;
                }
            }
        }
        t = arr[low]; arr[low] = arr[end]; arr[end] = t;
        // the reference recurses into the low part first, then the high part: push high first (LIFO)
        if (low < end) { stack[sp++] = (uint16_t)(low + 1); stack[sp++] = (uint16_t)end; }
        if (low > begin) { stack[sp++] = (uint16_t)begin; stack[sp++] = (uint16_t)(low - 1); }
    }
}

// The same quick sort run by the whole wave, level by level: the partitions of disjoint ranges commute, so the segments of one
// recursion depth are partitioned side by side, a lane each (exactly the reference's Lomuto pass per segment), and their children
// form the next depth's list.  Same final arrangement as the depth-first original; the serial work drops from ~n log n element steps
// to the longest segment of every depth (~2 n): the tie pass of a batch of 226 k keypoints took 5.9 ms with one lane per patch.
// seg: two lists of (begin, end) pairs, 340 pairs each (a depth has at most n / 2 segments of two or more elements); cnt[2]: their lengths.
__device__ void liop_ref_qsort_wave(uint2* __restrict__ arr, int n, uint16_t* __restrict__ seg, uint32_t* __restrict__ cnt, uint32_t lane)
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
            uint2 t = arr[pivot]; arr[pivot] = arr[end]; arr[end] = t;
            const float pv = __uint_as_float(arr[end].x);
            int low = begin;
            for (int i = begin; i < end; i += 4) {
                uint2 e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = arr[i + j];                 // (arr carries 4 entries of slack; positions >= end are not used)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i + j < end && __uint_as_float(e[j].x) - pv <= 0.0f) {
                        if (low != i + j) { const uint2 o = arr[low]; arr[i + j] = o; arr[low] = e[j]; }
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
    // ---- warp: 3 groups of 9 pixels per lane, the 36 loads of a group issued before the first is used
    {
        const uint32_t ln = liop_opaque(lane);
        LIOP_FIRST(y, x, ln);
#pragma unroll 1
        for (int g = 0; g < 3; ++g) {
            float v0[9], v1[9], v2[9], v3[9], wt[9][4];
            int at[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const bool live = y < (uint32_t)S;                                  // e < 1681
                const uint32_t yc = live ? y : 0u;
                const int2 a = L.ty[yc], b = L.tx[x];
                const int X = (a.x + b.x) >> 5, Y = (a.y + b.y) >> 5;
                int sx = X >> 5, sy = Y >> 5;
                sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
                sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
                const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
                wt[q][0] = (1.f - fy) * (1.f - fx); wt[q][1] = (1.f - fy) * fx; wt[q][2] = fy * (1.f - fx); wt[q][3] = fy * fx;
                const bool x0 = (unsigned)sx < (unsigned)w, x1 = (unsigned)(sx + 1) < (unsigned)w;
                const bool y0 = (unsigned)sy < (unsigned)h, y1 = (unsigned)(sy + 1) < (unsigned)h;
                const uint32_t cx0 = x0 ? (uint32_t)sx : 0u, cx1 = x1 ? (uint32_t)(sx + 1) : 0u;
                const uint32_t r0 = (y0 ? (uint32_t)sy : 0u) * (uint32_t)w, r1 = (y1 ? (uint32_t)(sy + 1) : 0u) * (uint32_t)w;
                const float a0 = image[r0 + cx0], a1 = image[r0 + cx1], a2 = image[r1 + cx0], a3 = image[r1 + cx1];
                v0[q] = (x0 && y0) ? a0 : 0.f; v1[q] = (x1 && y0) ? a1 : 0.f; v2[q] = (x0 && y1) ? a2 : 0.f; v3[q] = (x1 && y1) ? a3 : 0.f;
                at[q] = live ? (int)(y * (uint32_t)kLiopWS + x) : -1;
                LIOP_NEXT(y, x);
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                if (at[q] < 0) continue;
                const float v = v0[q] * wt[q][0] + v1[q] * wt[q][1] + v2[q] * wt[q][2] + v3[q] * wt[q][3];
                L.buf[at[q] + 5] = v;
            }
        }
    }
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

struct LiopParams {
    const float* patches;      // [n][41*41] (FUSED = false)
    const int*   pix;          // [n_pix] offsets of the circular support (scan order) in the ZERO-RINGED patch: (x + 1) + (y + 1) * 43
    const double2* samp_w;     // [n_pix][4] (wx, wy): fractional parts of the four sample positions of every support pixel
    const int4*  samp_off;     // [n_pix]    offsets of their top-left taps in the ringed patch: (floor(x) + 1) + (floor(y) + 1) * 43
    uint32_t n, n_pix;
    float* desc;               // [n][144]
    uint32_t* n_tie_patches;   // patches with equal intensities in their support: they need the reference's exact re-sort ...
    uint32_t* tie_list;        // ... and are left to the second pass (their indices, [n])
    // FUSED = true: the patch is warped + blurred from the image by the wavefront itself (liop_make_patch) and never exists in HBM
    const float* image0; int w, h;
    const float* M6;           // [n][6] inverse maps
    const float* kern;         // [11] blur taps
    const uint32_t* img_of;    // [n] image plane of every keypoint (or nullptr: plane 0)
};

// TIE_PASS = false: every patch whose support intensities are all distinct (almost all: blurred float images) -- 16 KB of LDS per
// one-wave workgroup, 10 of them per CU.  A patch with a tie only puts itself on P.tie_list.  TIE_PASS = true: the patches of that
// list, with the arrays of the reference's quick sort (8 KB more) -- launched right behind the first pass, its workgroups read the
// count from device memory (no host round trip; a launch over an empty list costs a few microseconds).
// The patch lives in LDS with a ring of zeros around it (43 x 43): the four taps of a bilinear sample are then always readable --
// vl_liop's `if (ix >= 0 && ...)` guards become reads of a 0.0f -- and the floor / fraction of every sample position, which depend
// on the support pixel only, come from host tables (computed with the reference's double operations) instead of f64 -> i64
// conversions per sample.
template <bool TIE_PASS, bool FUSED>
__global__ __launch_bounds__(64, 3)                  // (three waves per SIMD = what 16 KB of LDS per wavefront admits: 168 VGPRs, not the 512 a lone wave may take)
void liop_kernel(const LiopParams P)
{
    __shared__ __attribute__((aligned(16))) LiopWaveLds W;
    constexpr int kQPairs = TIE_PASS ? (kLiopMaxPix + 4) : 1;
    constexpr int kQStack = TIE_PASS ? (2 * kLiopMaxPix + 8) : 1;
    __shared__ uint2 qarr[kQPairs];                  // exact re-sort: (intensity bits, position)
    __shared__ uint16_t qstack[kQStack];
    __shared__ float inten[kLiopSortCap];            // intensities in scan order (for the exact re-sort)
    __shared__ uint16_t perm[kLiopSortCap];
    __shared__ uint32_t hist[144];
    __shared__ float s_norm;
    __shared__ uint32_t qcnt[2];
    float* patch = W.buf;

    const uint32_t lane0 = threadIdx.x;
    const uint32_t N = P.n_pix;
    const uint32_t n_items = TIE_PASS ? *P.n_tie_patches : P.n;
    float kern[11];
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < 11; ++k) kern[k] = P.kern[k];
    }
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        // (a laundered lane id per patch: nothing that depends on the lane alone -- a few hundred registers of index arithmetic over
        // the phases below -- is worth keeping alive across a whole patch, and the compiler would)
        const uint32_t lane = FUSED ? liop_opaque(lane0) : lane0;
        const uint32_t item = TIE_PASS ? P.tie_list[it] : it;
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

        // ---- 1. rank the support pixels by intensity (key i: lane i / 16, slot i % 16)
        unsigned long long keys[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t i = lane * 16u + (uint32_t)s;
            if (i < N) { const float v = patch[P.pix[i]]; inten[i] = v; keys[s] = ((unsigned long long)float_order_bits(v) << 32) | i; }
            else keys[s] = ~0ull;
        }
        liop_sort1024(keys, lane);
        bool tie = false;
        {
            // the first key of the next lane, for the pair that straddles two lanes
            const uint32_t nhi = (uint32_t)__shfl_down((int)(uint32_t)(keys[0] >> 32), 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint32_t i = lane * 16u + (uint32_t)s;
                if (i < N) perm[i] = (uint16_t)(keys[s] & 0xFFFFu);
                const uint32_t next_hi = (s < 15) ? (uint32_t)(keys[s < 15 ? s + 1 : 15] >> 32) : nhi;
                if (i + 1 < N) tie |= ((uint32_t)(keys[s] >> 32) == next_hi);
            }
        }
        const bool any_tie = __ballot(tie) != 0ull;
        r3dm_syncthreads();
        const float vmin = inten[perm[0]], vmax = inten[perm[N - 1]];
        if (vmin == vmax) {
            // constant support: every weight is 0, the descriptor is 0 / max(0, 1e-12) = 0
            for (uint32_t e = lane; e < 144; e += 64) P.desc[(size_t)item * 144 + e] = 0.0f;
            r3dm_syncthreads();
            continue;
        }
        if (any_tie) {
            if constexpr (!TIE_PASS) {
                if (lane == 0) P.tie_list[atomicAdd(P.n_tie_patches, 1u)] = item;      // the second pass does this patch
                r3dm_syncthreads();
                continue;
            } else {
                for (uint32_t i = lane; i < N + 4u; i += 64) qarr[i] = make_uint2(i < N ? __float_as_uint(inten[i]) : 0u, i);
                r3dm_syncthreads();
                liop_ref_qsort_wave(qarr, (int)N, qstack, qcnt, lane);
                r3dm_syncthreads();
                for (uint32_t i = lane; i < N; i += 64) perm[i] = (uint16_t)qarr[i].y;
                r3dm_syncthreads();
            }
        }
        // threshold = -intensityThreshold * (max - min), all float (vl_liop.c:497-503)
        const float thr = (float)(5.0 / 255) * (inten[perm[N - 1]] - inten[perm[0]]);

        // ---- 2. per rank: bin, 4 bilinear samples, permutation index, weight
        const uint32_t area = N / 6u;
        for (uint32_t i = lane; i < N; i += 64) {
            uint32_t bin = i / area; if (bin > 5u) bin = 5u;
            const uint32_t p = perm[i];
            const int4 so = P.samp_off[p];
            const int offs[4] = {so.x, so.y, so.z, so.w};
            double2 sw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sw[k] = P.samp_w[4 * p + k];
            float nv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* __restrict__ q = patch + offs[k];
                const double wx = sw[k].x, wy = sw[k].y;
                const double a = q[0], b = q[1], c = q[kLiopPS], d = q[kLiopPS + 1];
                nv[k] = (float)((1.0 - wy) * (a + (b - a) * wx) + wy * (c + (d - c) * wx));
            }
            // order of the 4 samples; without ties it is the unique ascending order
            int np[4];
            const bool ntie = nv[0] == nv[1] || nv[0] == nv[2] || nv[0] == nv[3] || nv[1] == nv[2] || nv[1] == nv[3] || nv[2] == nv[3];
            if (!ntie) {
                int rk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) r += (nv[u] < nv[k]);
                    rk[k] = r;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) np[r] = (rk[0] == r) ? 0 : (rk[1] == r) ? 1 : (rk[2] == r) ? 2 : 3;
            } else {
                np[0] = 0; np[1] = 1; np[2] = 2; np[3] = 3;
                liop_ref_qsort4(nv, np);
            }
            // lexicographic index of the permutation (Lehmer code)
            const int c0 = (np[1] < np[0]) + (np[2] < np[0]) + (np[3] < np[0]);
            const int c1 = (np[2] < np[1]) + (np[3] < np[1]);
            const int c2 = (np[3] < np[2]);
            const int index = c0 * 6 + c1 * 2 + c2;
            uint32_t weight = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a + 1; b < 4; ++b) weight += (nv[a] > nv[b] + thr || nv[b] > nv[a] + thr) ? 1u : 0u;
            if (weight) atomicAdd(&hist[bin * 24u + (uint32_t)index], weight);
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
__global__ __launch_bounds__(256, 4)
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
    hipLaunchKernelGGL((liop_kernel<false, false>), dim3(grid), dim3(64), 0, st, P);
    hipLaunchKernelGGL((liop_kernel<true, false>), dim3(grid < 2048u ? grid : 2048u), dim3(64), 0, st, P);
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
    hipLaunchKernelGGL((liop_kernel<false, true>), dim3(grid), dim3(64), 0, st, P);
    hipLaunchKernelGGL((liop_kernel<true, true>), dim3(grid < 2048u ? grid : 2048u), dim3(64), 0, st, P);
    return hipGetLastError();
}

}  // namespace r3dm
