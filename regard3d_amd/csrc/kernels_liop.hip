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

struct LiopParams {
    const float* patches;      // [n][41*41]
    const int*   pix;          // [n_pix] offsets of the circular support (scan order)
    const double* sx;          // [n_pix][4]
    const double* sy;          // [n_pix][4]
    uint32_t n, n_pix;
    float* desc;               // [n][144]
    uint32_t* n_tie_patches;   // patches with equal intensities in their support: they need the reference's exact re-sort ...
    uint32_t* tie_list;        // ... and are left to the second pass (their indices, [n])
};

// TIE_PASS = false: every patch whose support intensities are all distinct (almost all: blurred float images) -- 14 KB of LDS per
// one-wave workgroup, 11 of them per CU.  A patch with a tie only puts itself on P.tie_list.  TIE_PASS = true: the patches of that
// list, with the arrays of the reference's quick sort (8 KB more) -- launched right behind the first pass, its workgroups read the
// count from device memory (no host round trip; a launch over an empty list costs a few microseconds).
template <bool TIE_PASS>
__global__ __launch_bounds__(64)
void liop_kernel(const LiopParams P)
{
    // the patch -- and, in the second pass, over the same bytes, the arrays of the reference's quick sort (the patch is not needed while
    // they are: it is read again from global memory behind the sort)
    constexpr int kPatchFloats = kLiopPix + 3 + 128;                                 // (+ slack: the patch is loaded in float4 pieces)
    constexpr int kQBytes = (kLiopMaxPix + 4) * 8 + (2 * kLiopMaxPix + 8) * 2;
    constexpr int kRegionBytes = (TIE_PASS && kQBytes > kPatchFloats * 4) ? kQBytes : kPatchFloats * 4;
    __shared__ __attribute__((aligned(16))) unsigned char region[kRegionBytes];
    float* patch = reinterpret_cast<float*>(region);
    uint2* qarr = reinterpret_cast<uint2*>(region);                                   // exact re-sort: (intensity bits, position)
    uint16_t* qstack = reinterpret_cast<uint16_t*>(region + (kLiopMaxPix + 4) * 8);
    __shared__ float inten[kLiopSortCap];            // intensities in scan order (for the exact re-sort)
    __shared__ uint16_t perm[kLiopSortCap];
    __shared__ uint32_t hist[144];
    __shared__ float s_norm;
    __shared__ uint32_t qcnt[2];

    const uint32_t lane = threadIdx.x;
    const uint32_t N = P.n_pix;
    const uint32_t n_items = TIE_PASS ? *P.n_tie_patches : P.n;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const uint32_t item = TIE_PASS ? P.tie_list[it] : it;
        const float* src = P.patches + (size_t)item * kLiopPix;
        {
            // 1681 floats; a patch starts at a multiple of 4 bytes only, so the vector loads are of single floats, all in flight at once
            float v[27];
#pragma unroll
            for (int j = 0; j < 27; ++j) { const uint32_t e = lane + 64u * (uint32_t)j; v[j] = e < (uint32_t)kLiopPix ? src[e] : 0.0f; }
#pragma unroll
            for (int j = 0; j < 27; ++j) { const uint32_t e = lane + 64u * (uint32_t)j; if (e < (uint32_t)kLiopPix) patch[e] = v[j]; }
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
            const uint32_t nlo = (uint32_t)__shfl_down((int)(uint32_t)keys[0], 1), nhi = (uint32_t)__shfl_down((int)(uint32_t)(keys[0] >> 32), 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const uint32_t i = lane * 16u + (uint32_t)s;
                if (i < N) perm[i] = (uint16_t)(keys[s] & 0xFFFFu);
                const uint32_t next_hi = (s < 15) ? (uint32_t)(keys[s < 15 ? s + 1 : 15] >> 32) : nhi;
                if (i + 1 < N) tie |= ((uint32_t)(keys[s] >> 32) == next_hi);
            }
            (void)nlo;
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
                for (uint32_t e = lane; e < (uint32_t)kLiopPix; e += 64) patch[e] = src[e];      // the sort's arrays lay over the patch
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
            float nv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double x = P.sx[4 * p + k], y = P.sy[4 * p + k];
                const long xi = (long)x, yi = (long)y;
                const long ix = (x >= 0 || (double)xi == x) ? xi : xi - 1;
                const long iy = (y >= 0 || (double)yi == y) ? yi : yi - 1;
                const double wx = x - ix, wy = y - iy;
                double a = 0, b = 0, c = 0, d = 0;
                const int L = kLiopSide;
                if (ix >= 0 && iy >= 0) a = patch[ix + iy * L];
                if (ix < L - 1 && iy >= 0) b = patch[ix + 1 + iy * L];
                if (ix >= 0 && iy < L - 1) c = patch[ix + (iy + 1) * L];
                if (ix < L - 1 && iy < L - 1) d = patch[ix + 1 + (iy + 1) * L];
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

// ------------------------------------------------------------------------------------------------
// patch extraction: cv::warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_CONSTANT 0) + cv::GaussianBlur(sigma 1.2,
// 11 taps, BORDER_REFLECT_101) of extractLIOPFeatures (/root/reference/src/Regard3DFeatures.cpp:768-808), restated
// from OpenCV 4.0's scalar code paths (OpenCV is external: this sub-stage is parity-unpinned).
// One workgroup per keypoint; the 2x3 matrices are built on the host (float/double libm arithmetic of the
// reference, lines 790-799).  Fixed-point source coordinates exactly like hal::warpAffine: 10 fractional bits,
// rounded to 1/32 pixel, 32x32 table of float bilinear weights.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

__global__ __launch_bounds__(256)
void liop_extract_patches_kernel(const float* __restrict__ image0, int w, int h, const float* __restrict__ M6,
                                 const float* __restrict__ kern /* 11 taps */, uint32_t n, float* __restrict__ patches,
                                 const uint32_t* __restrict__ img_of)
{
    __shared__ float warped[kLiopPix];
    __shared__ float rowp[kLiopPix];
    const int S = kLiopSide;
    for (uint32_t item = blockIdx.x; item < n; item += gridDim.x) {
        // keypoints of a batch of same-size images in one launch: the keypoint's own image plane
        const float* __restrict__ image = image0 + (img_of ? (size_t)img_of[item] * ((size_t)w * (size_t)h) : 0);
        double M[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = (double)M6[6 * (size_t)item + k];
        for (int e = threadIdx.x; e < kLiopPix; e += 256) {
            const int y = e / S, x = e % S;
            const int X0 = (int)rint((M[1] * y + M[2]) * 1024) + 16;
            const int Y0 = (int)rint((M[4] * y + M[5]) * 1024) + 16;
            const int X = (X0 + (int)rint(M[0] * x * 1024)) >> 5;
            const int Y = (Y0 + (int)rint(M[3] * x * 1024)) >> 5;
            int sx = X >> 5, sy = Y >> 5;
            sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);
            sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
            const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
            const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
            const bool x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w, y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h;
            const float v0 = (x0 && y0) ? image[(size_t)sy * w + sx] : 0.f;
            const float v1 = (x1 && y0) ? image[(size_t)sy * w + sx + 1] : 0.f;
            const float v2 = (x0 && y1) ? image[(size_t)(sy + 1) * w + sx] : 0.f;
            const float v3 = (x1 && y1) ? image[(size_t)(sy + 1) * w + sx + 1] : 0.f;
            warped[e] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
        }
        r3dm_syncthreads();
        for (int e = threadIdx.x; e < kLiopPix; e += 256) {
            const int y = e / S, x = e % S;
            float s = kern[0] * warped[y * S + reflect101(x - 5, S)];
#pragma unroll
            for (int k = 1; k < 11; ++k) s += kern[k] * warped[y * S + reflect101(x + k - 5, S)];
            rowp[e] = s;
        }
        r3dm_syncthreads();
        float* out = patches + (size_t)item * kLiopPix;
        for (int e = threadIdx.x; e < kLiopPix; e += 256) {
            const int y = e / S, x = e % S;
            float s = kern[5] * rowp[e];
#pragma unroll
            for (int j = 1; j <= 5; ++j) s += kern[5 + j] * (rowp[reflect101(y + j, S) * S + x] + rowp[reflect101(y - j, S) * S + x]);
            out[e] = s;
        }
        r3dm_syncthreads();
    }
}

hipError_t launch_liop_extract(hipStream_t st, const float* image, int w, int h, const float* M6, const float* kern,
                               uint32_t n, float* patches, const uint32_t* img_of)
{
    if (n == 0) return hipSuccess;
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL(liop_extract_patches_kernel, dim3(grid), dim3(256), 0, st, image, w, h, M6, kern, n, patches, img_of);
    return hipGetLastError();
}

// n_tie_patches: one zeroed word (receives the number of patches that needed the exact re-sort); tie_list: n words of scratch
hipError_t launch_liop(hipStream_t st, const float* patches, const int* pix, const double* sx, const double* sy,
                       uint32_t n, uint32_t n_pix, float* desc, uint32_t* n_tie_patches, uint32_t* tie_list)
{
    if (n == 0) return hipSuccess;
    if (n_pix < 2 || n_pix > (uint32_t)kLiopMaxPix) return hipErrorInvalidValue;
    LiopParams P{patches, pix, sx, sy, n, n_pix, desc, n_tie_patches, tie_list};
    const uint32_t grid = n < 65536u ? n : 65536u;
    hipLaunchKernelGGL(liop_kernel<false>, dim3(grid), dim3(64), 0, st, P);
    hipLaunchKernelGGL(liop_kernel<true>, dim3(grid < 2048u ? grid : 2048u), dim3(64), 0, st, P);
    return hipGetLastError();
}

}  // namespace r3dm
