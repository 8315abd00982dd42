// kernels_akaze.hip -- Fast-A-KAZE keypoint detection for gfx950.
//
// Replaces the "Fast-AKAZE" arm of Regard3DFeatures::detectKeypoints (/root/reference/src/Regard3DFeatures.cpp:596-617):
// cv::AKAZE2::detect = AKAZEFeaturesV2::Create_Nonlinear_Scale_Space + Feature_Detection
// (src/thirdparty/fast-akaze/AKAZEFeatures.cpp:245-382; helpers in nldiffusion_functions.cpp, fed.cpp).  In the reference
// this stage is serialised by a semaphore (one image at a time) and its image-sized stencils run on OpenCV's CPU filters.
//
// Here every stencil is one HBM-bound pass (a thread per pixel, neighbours through L1/L2), the arithmetic is the
// restatement of OpenCV's scalar filter paths documented in oracle/akaze.c -- float, no contraction, only + - * / sqrt on
// the device (Gaussian taps, FED step sizes, the k-contrast scan and atan2 are computed by the host library with libm)
// -- so the keypoints equal the CPU restatement bit for bit.  Scale-space extrema are compacted in raster order and
// pruned by one wavefront per level that keeps the reference's sequential "first neighbour in the list" rule, with the
// live part of the list (rows within one radius of the scan line) held in LDS.
#include "r3dm_internal.hpp"

namespace r3dm {

namespace {

// Batching: one launch serves B same-size images.  blockIdx.z = image; every image plane a launch touches has the launch's own
// w x h, and the planes of image z start z * w * h floats into each buffer (api_features.cpp sizes every buffer for B planes).
#define AK_PLANE(ptr, w, h) ((ptr) + (size_t)blockIdx.z * ((size_t)(w) * (size_t)(h)))
constexpr uint32_t kAkSmallWords = 4096;            // per-image scalars: [0] max |grad| bits, [16..316) histogram, [1024..1032) 1 / k^2 per octave

__device__ __forceinline__ int ak_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int ak_refl101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

}  // namespace

// The stencil kernels of the launch sequence.  A workgroup owns a tile of 64 x (4 R) pixels: every lane a column, every wave R
// consecutive rows.  Two bodies: the INTERIOR body for tiles whose every tap lies inside the image (a workgroup-uniform test
// on blockIdx) loads everything the R rows of a lane need FIRST -- no border rule, addresses = a row pointer + a compile-time
// offset -- and only then computes, so that a wave has R rows' worth of distinct cache lines in flight (with one row per wave
// and dependent tap loops the passes were latency-bound at 2-3.4 TB/s with VALU, TA and HBM all under 60 % busy,
// profiles/r03_d_pmc_akaze_perf.txt); the BORDER body applies the reference's border rule per tap, pixel by pixel.
// Both perform the same float operations in the same order per pixel, so the split changes no bit.
template <bool IN> __device__ __forceinline__ int ak_cl(int v, int hi) { return IN ? v : ak_clamp(v, 0, hi); }
template <bool IN> __device__ __forceinline__ int ak_rf(int p, int len) { return IN ? p : ak_refl101(p, len); }
template <int R> __device__ __forceinline__ bool ak_strip_interior(int w, int h, int rx, int ry)
{
    const int x0 = (int)blockIdx.x * 64, y0 = (int)blockIdx.y * (4 * R);
    return x0 >= rx && x0 + 63 + rx < w && y0 >= ry && y0 + 4 * R - 1 + ry < h;
}
#define AK_STRIP(R) const int x = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63), y0 = (int)blockIdx.y * (4 * (R)) + (int)(threadIdx.x >> 6) * (R)

// ---- GaussianBlur, BORDER_REPLICATE: row pass (SymmRowSmallFilter for 5 taps, RowFilter otherwise), column pass (SymmColumnFilter)
__device__ __forceinline__ void ak_gauss_rows_px(const float* __restrict__ src, float* __restrict__ dst, int w, int x, int y, const AkTaps& kf)
{
    const uint32_t row = (uint32_t)y * (uint32_t)w;
    const float* __restrict__ S = src + row;
    const int n = kf.n, r = n / 2;
    float s;
    if (n == 5) {
        s = S[x] * kf.k[2] + (S[ak_clamp(x - 1, 0, w - 1)] + S[ak_clamp(x + 1, 0, w - 1)]) * kf.k[3]
            + (S[ak_clamp(x - 2, 0, w - 1)] + S[ak_clamp(x + 2, 0, w - 1)]) * kf.k[4];
    } else {
        s = kf.k[0] * S[ak_clamp(x - r, 0, w - 1)];
        for (int k = 1; k < n; ++k) s += kf.k[k] * S[ak_clamp(x + k - r, 0, w - 1)];
    }
    dst[row + (uint32_t)x] = s;
}
__device__ __forceinline__ void ak_gauss_cols_px(const float* __restrict__ src, float* __restrict__ dst, int w, int h, int x, int y, const AkTaps& kf)
{
    const int r = kf.n / 2;
    const uint32_t p = (uint32_t)y * (uint32_t)w + (uint32_t)x;
    float s = kf.k[r] * src[p];
    for (int k = 1; k <= r; ++k)
        s += kf.k[r + k] * (src[(uint32_t)ak_clamp(y + k, 0, h - 1) * (uint32_t)w + (uint32_t)x] + src[(uint32_t)ak_clamp(y - k, 0, h - 1) * (uint32_t)w + (uint32_t)x]);
    dst[p] = s;
}
// Row pass and column pass in ONE kernel: a workgroup row-filters the 16 + 2 r rows its 64 x 16 output tile needs into LDS
// (each wave 5-6 rows, all taps loaded before the first multiply) and column-filters them from there -- the row-filtered image
// never goes to HBM (2 planes of traffic instead of 4; the halo rows are row-filtered twice, by this tile and its neighbour).
// Per value exactly the operations of the two-pass form: row value = ak_gauss_rows_px's expression at (clamped) row y, output =
// ak_gauss_cols_px's expression on those values.  IN = the tile and its halo lie inside the image (no clamps).
template <int N, bool IN> __device__ __forceinline__ void ak_gauss_fused(const float* __restrict__ src, float* __restrict__ dst, int w, int h, const AkTaps& kf,
                                                                         float (*mid)[64])
{
    constexpr int r = N / 2, RI = 16 + 2 * r, PER = (RI + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (int)blockIdx.x * 64 + lane, y0 = (int)blockIdx.y * 16;
    const uint32_t uw = (uint32_t)w;
    float v[PER][N];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int jj = wave + 4 * q;
        const int yy = ak_cl<IN>(y0 - r + (jj < RI ? jj : RI - 1), h - 1);
        const float* __restrict__ S = src + (uint32_t)yy * uw;
        if (IN) {
            const float* __restrict__ P = S + (uint32_t)x;
#pragma unroll
            for (int k = 0; k < N; ++k) v[q][k] = P[k - r];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[q][k] = S[ak_clamp(x + k - r, 0, w - 1)];
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int jj = wave + 4 * q;
        float s;
        if (N == 5) {
            s = v[q][2] * kf.k[2] + (v[q][1] + v[q][3]) * kf.k[3] + (v[q][0] + v[q][4]) * kf.k[4];
        } else {
            s = kf.k[0] * v[q][0];
#pragma unroll
            for (int k = 1; k < N; ++k) s += kf.k[k] * v[q][k];
        }
        if (jj < RI) mid[jj][lane] = s;
    }
    r3dm_syncthreads();
    if (!IN && x >= w) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = wave * 4 + q;
        if (!IN && y0 + j >= h) break;
        float s = kf.k[r] * mid[j + r][lane];
#pragma unroll
        for (int k = 1; k <= r; ++k) s += kf.k[r + k] * (mid[j + r + k][lane] + mid[j + r - k][lane]);
        dst[(uint32_t)(y0 + j) * uw + (uint32_t)x] = s;
    }
}
__global__ __launch_bounds__(256)
void ak_gauss_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, AkTaps kf)
{
    __shared__ float mid[24][64];
    src = AK_PLANE(src, w, h); dst = AK_PLANE(dst, w, h);
    const int r = kf.n / 2;
    const bool in = ak_strip_interior<4>(w, h, r, r);
    if (kf.n == 5) { if (in) ak_gauss_fused<5, true>(src, dst, w, h, kf, mid); else ak_gauss_fused<5, false>(src, dst, w, h, kf, mid); }
    else { if (in) ak_gauss_fused<9, true>(src, dst, w, h, kf, mid); else ak_gauss_fused<9, false>(src, dst, w, h, kf, mid); }
}
// tap counts other than 5 / 9 (no sigma of the detector produces them): the two-pass form, one pixel per thread
__global__ __launch_bounds__(256)
void ak_gauss_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, AkTaps kf)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    ak_gauss_rows_px(AK_PLANE(src, w, h), AK_PLANE(dst, w, h), w, x, y, kf);
}
__global__ __launch_bounds__(256)
void ak_gauss_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, AkTaps kf)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    ak_gauss_cols_px(AK_PLANE(src, w, h), AK_PLANE(dst, w, h), w, h, x, y, kf);
}

// ---- Scharr 3x3, BORDER_REFLECT_101: row pass writes the derivative and the smoothed row, column pass Lx and Ly
__global__ __launch_bounds__(256)
void ak_scharr_rows_kernel(const float* __restrict__ src, float* __restrict__ rd, float* __restrict__ rs, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    src = AK_PLANE(src, w, h); rd = AK_PLANE(rd, w, h); rs = AK_PLANE(rs, w, h);
    const float* S = src + (size_t)y * w;
    const float a = S[ak_refl101(x - 1, w)], b = S[ak_refl101(x + 1, w)];
    rd[(size_t)y * w + x] = b - a;
    rs[(size_t)y * w + x] = S[x] * 10.0f + (a + b) * 3.0f;
}
__global__ __launch_bounds__(256)
void ak_scharr_cols_kernel(const float* __restrict__ rd, const float* __restrict__ rs, float* __restrict__ Lx, float* __restrict__ Ly,
                           int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    rd = AK_PLANE(rd, w, h); rs = AK_PLANE(rs, w, h); Lx = AK_PLANE(Lx, w, h); Ly = AK_PLANE(Ly, w, h);
    const int yu = ak_refl101(y - 1, h), yd = ak_refl101(y + 1, h);
    Lx[(size_t)y * w + x] = (rd[(size_t)yu * w + x] + rd[(size_t)yd * w + x]) * 3.0f + rd[(size_t)y * w + x] * 10.0f;
    Ly[(size_t)y * w + x] = rs[(size_t)yd * w + x] - rs[(size_t)yu * w + x];
}

// ---- multiscale Scharr derivative (taps at -s, 0, +s), BORDER_REFLECT_101: the reference's row pass and column pass
// (sepFilter2D) in ONE kernel.  Every output pixel recomputes the row-filtered values it needs (2 or 3 rows x 2 or 3 taps,
// served by L1/L2) with exactly the operations of the two-pass form, so the result is bit-identical while the intermediate
// image and half of the launches disappear (160 -> 80 launches per 4000 x 3000 image).
__device__ __forceinline__ float ak_sderiv_row(const float* __restrict__ S, int x, int w, int s, int dx)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const float a = S[ak_refl101(x - s, w)], b = S[ak_refl101(x + s, w)];
    if (dx) return (-a) + b;
    if (s == 2) return S[x] * kc + (a + b) * norm;
    return (norm * a + kc * S[x]) + norm * b;
}
__device__ __forceinline__ float ak_sderiv_at(const float* __restrict__ src, int x, int y, int w, int h, int s, int dx)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const float u = ak_sderiv_row(src + (uint32_t)ak_refl101(y - s, h) * (uint32_t)w, x, w, s, dx);
    const float d = ak_sderiv_row(src + (uint32_t)ak_refl101(y + s, h) * (uint32_t)w, x, w, s, dx);
    if (dx) { const float c = ak_sderiv_row(src + (uint32_t)y * (uint32_t)w, x, w, s, dx); return kc * c + norm * (d + u); }
    return d - u;
}
// the interior forms: the taps of one pixel, loaded by the caller.  u / m / d = rows y - S, y, y + S; a / c / b = columns x - S, x, x + S
template <int S> __device__ __forceinline__ float ak_sd_smooth3(float a, float c, float b)      // ak_sderiv_row(dx = 0)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    if (S == 2) return c * kc + (a + b) * norm;
    return (norm * a + kc * c) + norm * b;
}
template <int S> __device__ __forceinline__ float ak_sd_dx(float au, float bu, float am, float bm, float ad, float bd)   // ak_sderiv_at(dx = 1)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const float u = (-au) + bu, d = (-ad) + bd, c = (-am) + bm;
    return kc * c + norm * (d + u);
}
template <int S> __device__ __forceinline__ float ak_sd_dy(float au, float cu, float bu, float ad, float cd, float bd)   // ak_sderiv_at(dx = 0)
{
    const float u = ak_sd_smooth3<S>(au, cu, bu), d = ak_sd_smooth3<S>(ad, cd, bd);
    return d - u;
}
// both derivatives of one source in one pass: d/dx -> dst_x, d/dy -> dst_y (smooth -> Lx, Ly)
template <int S, int R> __device__ __forceinline__ void ak_sderiv_xy_strip(const float* __restrict__ src, float* __restrict__ dst_x, float* __restrict__ dst_y, int w, int x, int y0)
{
    const uint32_t uw = (uint32_t)w;
    const float* __restrict__ P = src + ((uint32_t)y0 * uw + (uint32_t)x);
    float u[R][3], m[R][2], d[R][3];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float* __restrict__ U = P + (uint32_t)(j + 0) * uw - (uint32_t)S * uw;
        const float* __restrict__ M = P + (uint32_t)j * uw;
        const float* __restrict__ D = P + (uint32_t)j * uw + (uint32_t)S * uw;
        u[j][0] = U[-S]; u[j][1] = U[0]; u[j][2] = U[S];
        m[j][0] = M[-S]; m[j][1] = M[S];
        d[j][0] = D[-S]; d[j][1] = D[0]; d[j][2] = D[S];
    }
    const uint32_t p = (uint32_t)y0 * uw + (uint32_t)x;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        dst_x[p + (uint32_t)j * uw] = ak_sd_dx<S>(u[j][0], u[j][2], m[j][0], m[j][1], d[j][0], d[j][2]);
        dst_y[p + (uint32_t)j * uw] = ak_sd_dy<S>(u[j][0], u[j][1], u[j][2], d[j][0], d[j][1], d[j][2]);
    }
}
__global__ __launch_bounds__(256)
void ak_sderiv_xy_kernel(const float* __restrict__ src, float* __restrict__ dst_x, float* __restrict__ dst_y, int w, int h, int s)
{
    constexpr int R = 4;
    AK_STRIP(R);
    src = AK_PLANE(src, w, h); dst_x = AK_PLANE(dst_x, w, h); dst_y = AK_PLANE(dst_y, w, h);
    if (s >= 2 && s <= 4 && ak_strip_interior<R>(w, h, s, s)) {
        if (s == 2) ak_sderiv_xy_strip<2, R>(src, dst_x, dst_y, w, x, y0);
        else if (s == 3) ak_sderiv_xy_strip<3, R>(src, dst_x, dst_y, w, x, y0);
        else ak_sderiv_xy_strip<4, R>(src, dst_x, dst_y, w, x, y0);
        return;
    }
    if (x >= w) return;
    for (int j = 0; j < R; ++j) {
        const int y = y0 + j;
        if (y >= h) break;
        const uint32_t p = (uint32_t)y * (uint32_t)w + (uint32_t)x;
        dst_x[p] = ak_sderiv_at(src, x, y, w, h, s, 1);
        dst_y[p] = ak_sderiv_at(src, x, y, w, h, s, 0);
    }
}
// The second derivatives consumed on the spot: Lxx = d/dx of Lx, Lxy = d/dy of Lx, Lyy = d/dy of Ly, det = Lxx * Lyy - Lxy * Lxy.
// Each is the very expression ak_sderiv_xy_kernel would have stored (same operations, same order), so the determinant is
// bit-identical to the three-image form while two work images and four of the ten plane passes of a level's Hessian disappear
// (reads Lx, Ly, writes Ldet: 3 planes instead of read Lx + write Lxx, Lxy + read Ly, Lxx, Lxy + write Ldet = 7).
template <int S, int R> __device__ __forceinline__ void ak_sderiv_det_strip(const float* __restrict__ lx, const float* __restrict__ ly, float* __restrict__ ldet, int w, int x, int y0)
{
    const uint32_t uw = (uint32_t)w;
    const uint32_t p = (uint32_t)y0 * uw + (uint32_t)x;
    const float* __restrict__ PX = lx + p;
    const float* __restrict__ PY = ly + p;
    float u[R][3], m[R][2], d[R][3], yu[R][3], yd[R][3];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float* __restrict__ U = PX + (uint32_t)j * uw - (uint32_t)S * uw;
        const float* __restrict__ M = PX + (uint32_t)j * uw;
        const float* __restrict__ D = PX + (uint32_t)j * uw + (uint32_t)S * uw;
        const float* __restrict__ YU = PY + (uint32_t)j * uw - (uint32_t)S * uw;
        const float* __restrict__ YD = PY + (uint32_t)j * uw + (uint32_t)S * uw;
        u[j][0] = U[-S]; u[j][1] = U[0]; u[j][2] = U[S];
        m[j][0] = M[-S]; m[j][1] = M[S];
        d[j][0] = D[-S]; d[j][1] = D[0]; d[j][2] = D[S];
        yu[j][0] = YU[-S]; yu[j][1] = YU[0]; yu[j][2] = YU[S];
        yd[j][0] = YD[-S]; yd[j][1] = YD[0]; yd[j][2] = YD[S];
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float lxx = ak_sd_dx<S>(u[j][0], u[j][2], m[j][0], m[j][1], d[j][0], d[j][2]);
        const float lxy = ak_sd_dy<S>(u[j][0], u[j][1], u[j][2], d[j][0], d[j][1], d[j][2]);
        const float lyy = ak_sd_dy<S>(yu[j][0], yu[j][1], yu[j][2], yd[j][0], yd[j][1], yd[j][2]);
        ldet[p + (uint32_t)j * uw] = lxx * lyy - lxy * lxy;
    }
}
__global__ __launch_bounds__(256)
void ak_sderiv_det_kernel(const float* __restrict__ lx, const float* __restrict__ ly, float* __restrict__ ldet, int w, int h, int s)
{
    constexpr int R = 2;
    AK_STRIP(R);
    lx = AK_PLANE(lx, w, h); ly = AK_PLANE(ly, w, h); ldet = AK_PLANE(ldet, w, h);
    if (s >= 2 && s <= 4 && ak_strip_interior<R>(w, h, s, s)) {
        if (s == 2) ak_sderiv_det_strip<2, R>(lx, ly, ldet, w, x, y0);
        else if (s == 3) ak_sderiv_det_strip<3, R>(lx, ly, ldet, w, x, y0);
        else ak_sderiv_det_strip<4, R>(lx, ly, ldet, w, x, y0);
        return;
    }
    if (x >= w) return;
    for (int j = 0; j < R; ++j) {
        const int y = y0 + j;
        if (y >= h) break;
        const float lxx = ak_sderiv_at(lx, x, y, w, h, s, 1);
        const float lxy = ak_sderiv_at(lx, x, y, w, h, s, 0);
        const float lyy = ak_sderiv_at(ly, x, y, w, h, s, 0);
        ldet[(uint32_t)y * (uint32_t)w + (uint32_t)x] = lxx * lyy - lxy * lxy;
    }
}

// ---- k-contrast: maximum of the gradient modulus over the interior, then its histogram.  The modulus comes straight from the
// smoothed image: for an interior pixel the Scharr row + column pass (ak_scharr_rows / _cols_kernel: Lx = (rd[y-1] + rd[y+1]) * 3 +
// rd[y] * 10, Ly = rs[y+1] - rs[y-1]) touches no border, so both kernels recompute exactly those expressions from the 3 x 3
// neighbourhood instead of reading two derivative images that four more plane passes would have to write first.
// Tile = 64 columns x 16 rows of the interior [1, w-2] x [1, h-2]; every lane a column and 4 rows, 18 loads up front.
__device__ __forceinline__ void ak_modg_strip(const float* __restrict__ src, int w, int h, int tile_y, float (&m)[4], bool (&ok)[4])
{
    const int x = 1 + (int)blockIdx.x * 64 + (int)(threadIdx.x & 63), y0 = 1 + tile_y * 16 + (int)(threadIdx.x >> 6) * 4;
    const bool vx = x < w - 1;
    const int xc = vx ? x : 1;
    float v[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        int yy = y0 - 1 + j;
        yy = yy < h - 1 ? yy : h - 1;
        const float* __restrict__ P = src + ((uint32_t)yy * (uint32_t)w + (uint32_t)xc);
        v[j][0] = P[-1]; v[j][1] = P[0]; v[j][2] = P[1];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float au = v[j][0], cu = v[j][1], bu = v[j][2], ac = v[j + 1][0], bc = v[j + 1][2], ad = v[j + 2][0], cd = v[j + 2][1], bd = v[j + 2][2];
        const float rdu = bu - au, rdc = bc - ac, rdd = bd - ad;
        const float rsu = cu * 10.0f + (au + bu) * 3.0f, rsd = cd * 10.0f + (ad + bd) * 3.0f;
        const float lx = (rdu + rdd) * 3.0f + rdc * 10.0f;
        const float ly = rsd - rsu;
        m[j] = sqrtf(lx * lx + ly * ly);
        ok[j] = vx && (y0 + j < h - 1);
    }
}
__global__ __launch_bounds__(256)
void ak_modg_max_kernel(const float* __restrict__ src, int w, int h, uint32_t* __restrict__ out_max)
{
    __shared__ float part[4];
    src = AK_PLANE(src, w, h); out_max += (size_t)blockIdx.z * kAkSmallWords;
    float m = 0.0f;
    const int tiles_y = (h - 2 + 15) / 16;
    for (int ty = (int)blockIdx.y; ty < tiles_y; ty += (int)gridDim.y) {          // one atomic per workgroup, not per tile
        float mg[4]; bool ok[4];
        ak_modg_strip(src, w, h, ty, mg, ok);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (ok[j]) m = mg[j] > m ? mg[j] : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    r3dm_syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) m = part[q] > m ? part[q] : m;
        if (m > 0.0f) atomicMax(out_max, __float_as_uint(m));          // m >= 0: bit order = value order
    }
}
// hmax_bits: the maximum found by ak_modg_max_kernel, read on the device (no host round trip); bin scale = (nbins - 1) / hmax
__global__ __launch_bounds__(256)
void ak_modg_hist_kernel(const float* __restrict__ src, int w, int h, const uint32_t* __restrict__ hmax_bits, int nbins, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[512];
    src = AK_PLANE(src, w, h); hmax_bits += (size_t)blockIdx.z * kAkSmallWords; hist += (size_t)blockIdx.z * kAkSmallWords;
    const float hmax = __uint_as_float(*hmax_bits);
    if (hmax == 0.0f) return;                              // compute_k_percentileV2 keeps its default then (workgroup-uniform)
    const float sc = (nbins - 1) / hmax;
    for (int k = threadIdx.x; k < nbins; k += 256) lh[k] = 0;
    r3dm_syncthreads();
    const int tiles_y = (h - 2 + 15) / 16;
    for (int ty = (int)blockIdx.y; ty < tiles_y; ty += (int)gridDim.y) {          // the 300 bins are flushed once per workgroup, not per tile
        float mg[4]; bool ok[4];
        ak_modg_strip(src, w, h, ty, mg, ok);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (ok[j]) atomicAdd(&lh[(int)(mg[j] * sc)], 1u);
    }
    r3dm_syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256) if (lh[k]) atomicAdd(hist + k, lh[k]);
}

// compute_k_percentileV2 (nldiffusion_functions.cpp:212-262), the scan over the 300 bins: one thread, the reference's own float
// operations.  out[o] = 1 / k_o^2 for octave o, k_0 = the percentile (0.03 when the image is flat), k_o = k_{o-1} * 0.75
// (Create_Nonlinear_Scale_Space: kcontrast *= 0.75 at every octave change).
__global__ void ak_kcontrast_kernel(const uint32_t* __restrict__ hmax_bits, const uint32_t* __restrict__ hist, int nbins, uint32_t total,
                                    int have_hist, float* __restrict__ inv_k2)
{
    hmax_bits += (size_t)blockIdx.x * kAkSmallWords; hist += (size_t)blockIdx.x * kAkSmallWords; inv_k2 += (size_t)blockIdx.x * kAkSmallWords;   // one thread per image
    float kcontrast = 0.03f;
    const float hmax = __uint_as_float(*hmax_bits);
    if (have_hist && hmax != 0.0f) {
        const int nthreshold = (int)((float)(total - hist[0]) * 0.7f);
        int nelements = 0;
        for (int k = 1; k < nbins; ++k) {
            if (nelements >= nthreshold) { kcontrast = hmax * (float)k / (float)nbins; break; }
            nelements = nelements + (int)hist[k];
        }
    }
    for (int o = 0; o < 8; ++o) { inv_k2[o] = 1.0f / (kcontrast * kcontrast); kcontrast = kcontrast * 0.75f; }
}

// ---- Scharr 3x3 (row pass + column pass, as above) and the PM-G2 conductivity in one kernel: flow = 1 / (1 + |grad|^2 / k^2).
// Same operations per pixel as the Scharr row + column pass followed by the conductivity, without the four intermediate images.
__device__ __forceinline__ float ak_g2_from_taps(float au, float cu, float bu, float ac, float bc, float ad, float cd, float bd, float inv_k2)
{
    const float rdu = bu - au, rdc = bc - ac, rdd = bd - ad;                       // row pass, derivative
    const float rsu = cu * 10.0f + (au + bu) * 3.0f, rsd = cd * 10.0f + (ad + bd) * 3.0f;     // row pass, smoothing
    const float lx = (rdu + rdd) * 3.0f + rdc * 10.0f;
    const float ly = rsd - rsu;
    return 1.0f / (1.0f + ((lx * lx + ly * ly) * inv_k2));
}
__device__ __forceinline__ float ak_scharr_g2_px(const float* __restrict__ src, int w, int h, int x, int y, float inv_k2)
{
    const int xl = ak_refl101(x - 1, w), xr = ak_refl101(x + 1, w);
    const float* __restrict__ Su = src + (uint32_t)ak_refl101(y - 1, h) * (uint32_t)w;
    const float* __restrict__ Sc = src + (uint32_t)y * (uint32_t)w;
    const float* __restrict__ Sd = src + (uint32_t)ak_refl101(y + 1, h) * (uint32_t)w;
    return ak_g2_from_taps(Su[xl], Su[x], Su[xr], Sc[xl], Sc[xr], Sd[xl], Sd[x], Sd[xr], inv_k2);
}
__global__ __launch_bounds__(256)
void ak_scharr_g2_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, const float* __restrict__ inv_k2_p)
{
    constexpr int R = 4;
    AK_STRIP(R);
    src = AK_PLANE(src, w, h); dst = AK_PLANE(dst, w, h); inv_k2_p += (size_t)blockIdx.z * kAkSmallWords;
    const float inv_k2 = *inv_k2_p;                        // 1 / k^2 of this octave, left on the device by ak_kcontrast_kernel
    if (ak_strip_interior<R>(w, h, 1, 1)) {
        const uint32_t uw = (uint32_t)w;
        const float* __restrict__ P = src + ((uint32_t)(y0 - 1) * uw + (uint32_t)x);
        float v[R + 2][3];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) { const float* __restrict__ Pj = P + (uint32_t)j * uw; v[j][0] = Pj[-1]; v[j][1] = Pj[0]; v[j][2] = Pj[1]; }
        float* __restrict__ D = dst + ((uint32_t)y0 * uw + (uint32_t)x);
#pragma unroll
        for (int j = 0; j < R; ++j)
            D[(uint32_t)j * uw] = ak_g2_from_taps(v[j][0], v[j][1], v[j][2], v[j + 1][0], v[j + 1][2], v[j + 2][0], v[j + 2][1], v[j + 2][2], inv_k2);
        return;
    }
    if (x >= w) return;
    for (int j = 0; j < R; ++j) if (y0 + j < h) dst[(uint32_t)(y0 + j) * (uint32_t)w + (uint32_t)x] = ak_scharr_g2_px(src, w, h, x, y0 + j, inv_k2);
}

// ---- one FED step: Lstep (nld_step_scalar_one_lane; corners 0) and out = Lt + Lstep * 0.5 * step
__device__ __forceinline__ void ak_fed_px_border(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ out, int w, int h, int x, int y, float step_size)
{
    const size_t p = (size_t)y * w + x;
    const bool has_l = x > 0, has_r = x < w - 1, has_a = y > 0, has_b = y < h - 1;
    const float tc = Lt[p], fc = Lf[p];
    float v;
    if (!has_a) {
        if (!has_l || !has_r) v = 0.0f;
        else v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc);
    } else if (!has_b) {
        if (!has_l || !has_r) v = 0.0f;
        else v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
    } else if (!has_l) {
        v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
    } else if (!has_r) {
        v = (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
    } else {
        v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) +
            (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
    }
    out[p] = tc + v * 0.5f * step_size;
}
__global__ __launch_bounds__(256)
void ak_fed_step_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ out, int w, int h, float step_size)
{
    constexpr int R = 4;
    AK_STRIP(R);
    Lt = AK_PLANE(Lt, w, h); Lf = AK_PLANE(Lf, w, h); out = AK_PLANE(out, w, h);
    if (ak_strip_interior<R>(w, h, 1, 1)) {
        const uint32_t uw = (uint32_t)w;
        const uint32_t q = (uint32_t)(y0 - 1) * uw + (uint32_t)x;
        const float* __restrict__ T = Lt + q;
        const float* __restrict__ F = Lf + q;
        float tc[R + 2], fc[R + 2], tl[R], tr[R], fl[R], fr[R];
#pragma unroll
        for (int j = 0; j < R + 2; ++j) { tc[j] = T[(uint32_t)j * uw]; fc[j] = F[(uint32_t)j * uw]; }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float* __restrict__ Tj = T + (uint32_t)(j + 1) * uw;
            const float* __restrict__ Fj = F + (uint32_t)(j + 1) * uw;
            tl[j] = Tj[-1]; tr[j] = Tj[1]; fl[j] = Fj[-1]; fr[j] = Fj[1];
        }
        float* __restrict__ O = out + ((uint32_t)y0 * uw + (uint32_t)x);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const float t = tc[j + 1], f = fc[j + 1];
            const float v = (f + fr[j]) * (tr[j] - t) + (f + fl[j]) * (tl[j] - t) + (f + fc[j + 2]) * (tc[j + 2] - t) + (f + fc[j]) * (tc[j] - t);
            O[(uint32_t)j * uw] = t + v * 0.5f * step_size;
        }
        return;
    }
    if (x >= w) return;
    for (int j = 0; j < R; ++j) if (y0 + j < h) ak_fed_px_border(Lt, Lf, out, w, h, x, y0 + j, step_size);
}

// ---- up to 4 FED steps in ONE launch, for the small octaves: there a step is a few microseconds of work behind a launch (8 us per
// step for eight 500 x 375 images: the launch floor), and a level of those octaves takes 8 .. 28 steps.  A workgroup owns a
// 32 x 32 tile: it loads the tile with a halo of 4 (Lt and the conductivity) into LDS, runs the steps there -- after step s the
// cells at least s from the halo's edge are exact -- and writes its tile.  Every cell update is the expression of
// ak_fed_step_kernel with the same border cases (decided by the cell's position in the IMAGE), so the result is bit-identical;
// the halo cells are simply updated twice (here and by the neighbouring tile).
struct AkFedSteps { float tau[6]; int n; };
__global__ __launch_bounds__(256)
void ak_fed_multi_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ out, int w, int h, AkFedSteps st)
{
    constexpr int K = 4, T = 32, HW = T + 2 * K;
    __shared__ float a[2][HW * HW];
    __shared__ float f[HW * HW];
    Lt = AK_PLANE(Lt, w, h); Lf = AK_PLANE(Lf, w, h); out = AK_PLANE(out, w, h);
    const int x0 = (int)blockIdx.x * T - K, y0 = (int)blockIdx.y * T - K;
    // a thread's cells c = tid + 256 i of the 40 x 40 tile: position, image coordinates and border case once, not per step
    constexpr int NC = (HW * HW + 255) / 256;
    int cq[NC]; unsigned char cl[NC], cy[NC], cm[NC];          // LDS index; column, row in the tile; bit 0..3 = has_l, has_r, has_a, has_b, bit 4 = inside the image
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (int)threadIdx.x + 256 * i;
        const int lx = c % HW, ly = c / HW, gx = x0 + lx, gy = y0 + ly;
        const bool in = c < HW * HW && gx >= 0 && gx < w && gy >= 0 && gy < h;
        cq[i] = c < HW * HW ? c : 0; cl[i] = (unsigned char)lx; cy[i] = (unsigned char)(c < HW * HW ? ly : 255);
        cm[i] = (unsigned char)((gx > 0 ? 1 : 0) | (gx < w - 1 ? 2 : 0) | (gy > 0 ? 4 : 0) | (gy < h - 1 ? 8 : 0) | (in ? 16 : 0));
        const uint32_t p = (uint32_t)(in ? gy : 0) * (uint32_t)w + (uint32_t)(in ? gx : 0);
        const float tv = Lt[p], fv = Lf[p];
        if (c < HW * HW) { a[0][c] = in ? tv : 0.0f; f[c] = in ? fv : 0.0f; }
    }
    r3dm_syncthreads();
    for (int s = 0; s < st.n; ++s) {
        const float* __restrict__ cur = a[s & 1];
        float* __restrict__ nxt = a[(s + 1) & 1];
        const int lo = s + 1, hi = HW - 1 - (s + 1);                    // cells [lo, hi] x [lo, hi] are exact after this step
        const float step_size = st.tau[s];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int lx = cl[i], ly = cy[i];
            if (!(cm[i] & 16) || lx < lo || lx > hi || ly < lo || ly > hi) continue;
            const int q = cq[i];
            const bool has_l = cm[i] & 1, has_r = cm[i] & 2, has_a = cm[i] & 4, has_b = cm[i] & 8;
            const float tc = cur[q], fc = f[q];
            float v;
            if (!has_a) {
                if (!has_l || !has_r) v = 0.0f;
                else v = (fc + f[q + 1]) * (cur[q + 1] - tc) + (fc + f[q - 1]) * (cur[q - 1] - tc) + (fc + f[q + HW]) * (cur[q + HW] - tc);
            } else if (!has_b) {
                if (!has_l || !has_r) v = 0.0f;
                else v = (fc + f[q + 1]) * (cur[q + 1] - tc) + (fc + f[q - 1]) * (cur[q - 1] - tc) + (fc + f[q - HW]) * (cur[q - HW] - tc);
            } else if (!has_l) {
                v = (fc + f[q + 1]) * (cur[q + 1] - tc) + (fc + f[q + HW]) * (cur[q + HW] - tc) + (fc + f[q - HW]) * (cur[q - HW] - tc);
            } else if (!has_r) {
                v = (fc + f[q - 1]) * (cur[q - 1] - tc) + (fc + f[q + HW]) * (cur[q + HW] - tc) + (fc + f[q - HW]) * (cur[q - HW] - tc);
            } else {
                v = (fc + f[q + 1]) * (cur[q + 1] - tc) + (fc + f[q - 1]) * (cur[q - 1] - tc) +
                    (fc + f[q + HW]) * (cur[q + HW] - tc) + (fc + f[q - HW]) * (cur[q - HW] - tc);
            }
            nxt[q] = tc + v * 0.5f * step_size;
        }
        r3dm_syncthreads();
    }
    const float* __restrict__ fin = a[st.n & 1];
    for (int c = threadIdx.x; c < T * T; c += 256) {
        const int lx = K + c % T, ly = K + c / T, gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) out[(uint32_t)gy * (uint32_t)w + (uint32_t)gx] = fin[ly * HW + lx];
    }
}

// ---- up to 4 FED steps in ONE pass over the image, in registers: fed_march.inc (the control flow is shared with a CPU emulation,
// tests/cpp/fed_march_emul.cpp).  A wavefront = a strip of 64 columns marching down a band of rows; four strips per workgroup.
}  // namespace r3dm
#define FED_HD __device__ __forceinline__
#include "fed_march.inc"
namespace r3dm {
struct AkWaveOps {
    using VF = float; using VM = bool; using VI = int;
    static __device__ __forceinline__ VF zero() { return 0.0f; }
    static __device__ __forceinline__ VI lane_plus(int b) { return b + (int)(threadIdx.x & 63u); }
    static __device__ __forceinline__ VM gt(VI a, int b) { return a > b; }
    static __device__ __forceinline__ VM lt(VI a, int b) { return a < b; }
    static __device__ __forceinline__ VM land(VM a, VM b) { return a && b; }
    static __device__ __forceinline__ VM core_lanes(int k) { const int l = (int)(threadIdx.x & 63u); return l >= k && l < 64 - k; }
    static __device__ __forceinline__ VI clampi(VI a, int lo, int hi) { return ak_clamp(a, lo, hi); }
    static __device__ __forceinline__ VF load(const float* __restrict__ p, int row, int w, VI xc) { return p[(uint32_t)row * (uint32_t)w + (uint32_t)xc]; }
    static __device__ __forceinline__ void store(float* __restrict__ p, int row, int w, VI x, VF v, VM m) { if (m) p[(uint32_t)row * (uint32_t)w + (uint32_t)x] = v; }
    static __device__ __forceinline__ VF add(VF a, VF b) { return a + b; }
    static __device__ __forceinline__ VF sub(VF a, VF b) { return a - b; }
    static __device__ __forceinline__ VF mul(VF a, VF b) { return a * b; }
    static __device__ __forceinline__ VF muls(VF a, float b) { return a * b; }
    static __device__ __forceinline__ VF sel(VM m, VF a, VF b) { return m ? a : b; }
    // DPP wavefront shifts: lane i reads lane i - 1 (wave_shr:1) / lane i + 1 (wave_shl:1); the end lanes read 0 -- garbage by design
    static __device__ __forceinline__ VF shr(VF a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x138, 0xF, 0xF, false)); }
    static __device__ __forceinline__ VF shl(VF a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xF, 0xF, false)); }
};
template <int K>
__global__ __launch_bounds__(256)
void ak_fed_march_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, float* __restrict__ out, int w, int h, AkFedSteps st, int rows_per_band)
{
    constexpr int VW = 64 - 2 * K;                         // columns a strip stores
    Lt = AK_PLANE(Lt, w, h); Lf = AK_PLANE(Lf, w, h); out = AK_PLANE(out, w, h);
    const int strip = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (strip * VW >= w) return;                           // (wave-uniform: no barrier anywhere in this kernel)
    const int x_first = strip * VW - K;
    const int y0 = (int)blockIdx.y * rows_per_band, y1 = y0 + rows_per_band < h ? y0 + rows_per_band : h;
    float tau[K];
#pragma unroll
    for (int k = 0; k < K; ++k) tau[k] = st.tau[k];
    const bool edge = x_first < 1 || x_first + 63 > w - 2;
    if (edge) fed_march_strip<K, true, AkWaveOps>(Lt, Lf, out, w, h, tau, x_first, y0, y1);
    else fed_march_strip<K, false, AkWaveOps>(Lt, Lf, out, w, h, tau, x_first, y0, y1);
}

// ---- Gaussian + both derivative images + determinant + conductivity of a level in ONE pass: level_head.inc (control flow shared with
// the CPU emulation tests/cpp/level_head_emul.cpp).  A wavefront = a strip of 64 columns marching down a band of rows with its
// intermediate rows in LDS rings; four strips per workgroup, no workgroup barrier.
}  // namespace r3dm
#include "level_head.inc"
namespace r3dm {
struct AkHeadOps : AkWaveOps {
    using Lds = float*;
    static __device__ __forceinline__ VI col_clamp(VI x, int off, int w, int x_first) { return ak_clamp(ak_clamp(x + off, 0, w - 1) - x_first, -16, 79); }
    static __device__ __forceinline__ VI col_refl(VI x, int off, int w, int x_first) { return ak_clamp(ak_refl101(x + off, w) - x_first, -16, 79); }
    static __device__ __forceinline__ void lds_store(Lds l, int at, VF v) { l[at + (int)(threadIdx.x & 63u)] = v; }
    static __device__ __forceinline__ VF lds_load(Lds l, int at, VI idx) { return l[at + idx]; }
    static __device__ __forceinline__ VF lds_load_off(Lds l, int at, int off) { return l[at + (int)(threadIdx.x & 63u) + off]; }
    static __device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    static __device__ __forceinline__ VF adds(VF a, float b) { return a + b; }
    static __device__ __forceinline__ VF neg(VF a) { return -a; }
    static __device__ __forceinline__ VF rcp_div(float a, VF b) { return a / b; }
};
template <int S>
__global__ __launch_bounds__(256)
void ak_level_head_kernel(const float* __restrict__ src, float* __restrict__ Lx, float* __restrict__ Ly, float* __restrict__ Ldet, float* __restrict__ flow,
                          int w, int h, AkTaps kf, const float* __restrict__ inv_k2_p, int rows_per_band)
{
    using G = LevelHeadGeom<S>;
    __shared__ float rings[4][G::FLOATS];
    src = AK_PLANE(src, w, h); Lx = AK_PLANE(Lx, w, h); Ly = AK_PLANE(Ly, w, h); Ldet = AK_PLANE(Ldet, w, h); flow = AK_PLANE(flow, w, h);
    const float inv_k2 = inv_k2_p[(size_t)blockIdx.z * kAkSmallWords];        // 1 / k^2 of this octave, left on the device by ak_kcontrast_kernel
    const int wave = (int)(threadIdx.x >> 6);
    const int strip = (int)blockIdx.x * 4 + wave;
    if (strip * G::VW >= w) return;                        // (wave-uniform: no workgroup barrier in this kernel)
    const int x_first = strip * G::VW - G::H;
    const int y0 = (int)blockIdx.y * rows_per_band, y1 = y0 + rows_per_band < h ? y0 + rows_per_band : h;
    float k5[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) k5[k] = kf.k[k];
    const bool edge = x_first < 0 || x_first + 63 > w - 1;
    if (edge) level_head_strip<S, true, AkHeadOps>(src, Lx, Ly, Ldet, flow, w, h, k5, inv_k2, x_first, y0, y1, rings[wave]);
    else level_head_strip<S, false, AkHeadOps>(src, Lx, Ly, Ldet, flow, w, h, k5, inv_k2, x_first, y0, y1, rings[wave]);
}

// ---- halfsample: INTER_AREA, exact 2x (resizeAreaFast_) or fractional cells (ResizeArea_, tables from the host)
__global__ __launch_bounds__(256)
void ak_half_fast_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int sh, int dw, int dh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    src = AK_PLANE(src, w, sh); dst = AK_PLANE(dst, dw, dh);
    const float* S = src + (size_t)(2 * y) * w + 2 * x;
    float sum = 0; sum += S[0]; sum += S[1]; sum += S[w]; sum += S[w + 1];
    dst[(size_t)y * dw + x] = sum * 0.25f;
}
__global__ __launch_bounds__(256)
void ak_half_area_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int sh, int dw, int dh,
                         const AkAreaTab* __restrict__ xt, const int* __restrict__ xb, const AkAreaTab* __restrict__ yt, const int* __restrict__ yb)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    src = AK_PLANE(src, w, sh); dst = AK_PLANE(dst, dw, dh);
    float sum = 0.0f;
    for (int j = yb[y]; j < yb[y + 1]; ++j) {
        const float* S = src + (size_t)yt[j].si * w;
        float buf = 0.0f;
        for (int k = xb[x]; k < xb[x + 1]; ++k) buf += S[xt[k].si] * xt[k].alpha;
        if (j == yb[y]) sum = yt[j].alpha * buf; else sum += yt[j].alpha * buf;
    }
    dst[(size_t)y * dw + x] = sum;
}

// ---- scale-space extrema of one level: per-row counts, then raster-ordered compaction
__device__ __forceinline__ bool ak_is_extremum(const float* __restrict__ ldet, int w, int x, int y, float thr)
{
    const float* curr = ldet + (size_t)y * w; const float* prev = curr - w; const float* next = curr + w;
    const float v = curr[x];
    if (v <= thr) return false;
    if (v <= curr[x - 1] || v <= curr[x + 1]) return false;
    if (v <= prev[x - 1] || v <= prev[x] || v <= prev[x + 1]) return false;
    if (v <= next[x - 1] || v <= next[x] || v <= next[x + 1]) return false;
    return true;
}
// The count pass: a workgroup owns a tile of 64 columns x 64 rows of ONE level's interior (every lane a column, every wave 16 rows;
// the 18 values a lane needs are loaded before anything is compared), blockIdx.x = tile over all levels (AkTileTable: where the
// tiles of each level begin), blockIdx.y = image.  The outcome of the 3 x 3 test is kept as one bit per pixel (a 64-bit ballot per
// 64 pixels, L.mask: 1/32 of the image) next to the per-row counts; after the scan of the counts and the device-side slot layout the
// emit pass reads only the bit masks and the determinant at the set bits -- no second sweep over the image, no barrier anywhere
// (the two-sweep, two-barriers-per-256-pixels form took 1.4 ms per call for eight 12 Mpx images, a wave-per-row sweep 1.0 ms).
// (Tile = 64 columns x 64 rows since round 5: a wave owns 16 rows and has the 54 loads of its 18 input rows in flight at once.  With
// 16-row tiles the launch was 471 k workgroups of two dependent memory round trips each -- bound by workgroup turnover at 1 TB/s.)
constexpr int kAkMaskRows = 16;                         // rows per wavefront of the extremum count pass (tile = 4 x that)
__global__ __launch_bounds__(256)
void ak_extrema_mask_kernel(const AkLevelDev* __restrict__ levels, int n_levels, AkTileTable tt, float thr)
{
    constexpr int RW = kAkMaskRows;
    const uint32_t t = blockIdx.x;
    int li = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) li += (t >= tt.begin[i]) ? 1 : 0;             // begin[i] = 0xFFFFFFFF beyond the last level
    const AkLevelDev L = levels[blockIdx.y * (unsigned)n_levels + (unsigned)li];
    const uint32_t local = t - tt.begin[li];
    const int tx = (int)(local % L.mask_words), ty = (int)(local / L.mask_words);
    const int rows = L.h - 2 * L.border, lane = threadIdx.x & 63;
    const int row0 = ty * (4 * RW) + (int)(threadIdx.x >> 6) * RW;
    if (row0 >= rows) return;                                              // wave-uniform
    const int x = L.border + tx * 64 + lane;
    const bool valid_x = x < L.w - L.border;
    const int xc = valid_x ? x : L.border;
    const float* __restrict__ ldet = L.Ldet;
    // (the neighbours' columns from the adjacent lanes by DPP wave shifts -- two loads per pixel row instead of three -- measured the
    // same 519 us, and a third form that loaded the outer column under `if (lane == 0 || lane == 63)` 926 us: the pass is bound by
    // the latency of its ~77 rounds of workgroups over 0.68 GB of cold determinant planes, not by its load instructions)
    float v[RW + 2][3];
#pragma unroll
    for (int j = 0; j < RW + 2; ++j) {
        int yy = L.border + row0 - 1 + j;
        yy = yy < L.h - 1 ? yy : L.h - 1;
        const float* __restrict__ P = ldet + ((uint32_t)yy * (uint32_t)L.w + (uint32_t)xc);
        v[j][0] = P[-1]; v[j][1] = P[0]; v[j][2] = P[1];
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int row = row0 + j;
        const float c = v[j + 1][1];
        // ak_is_extremum: `if (v <= n) return false` for the threshold and the eight neighbours
        const bool is = valid_x && row < rows && !(c <= thr) && !(c <= v[j + 1][0]) && !(c <= v[j + 1][2]) &&
                        !(c <= v[j][0]) && !(c <= v[j][1]) && !(c <= v[j][2]) && !(c <= v[j + 2][0]) && !(c <= v[j + 2][1]) && !(c <= v[j + 2][2]);
        const unsigned long long bal = __ballot(is);
        if (row < rows && lane == 0) {
            L.mask[(size_t)row * L.mask_words + (uint32_t)tx] = bal;
            if (bal) atomicAdd(L.row_cnt + row, (uint32_t)__builtin_popcountll(bal));      // (zeroed by the host before the pass)
        }
    }
}
__global__ __launch_bounds__(256)
void ak_extrema_emit_kernel(const AkLevelDev* __restrict__ levels)
{
    const AkLevelDev L = levels[blockIdx.z * gridDim.y + blockIdx.y];
    const int row = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (row >= L.h - 2 * L.border || L.counts[0] == 0) return;          // wave-uniform
    if (L.row_cnt[row] == 0) return;
    const int y = L.border + row, lane = threadIdx.x & 63;
    const unsigned long long* __restrict__ mrow = L.mask + (size_t)row * L.mask_words;
    uint32_t base = L.row_off[row];
    // (the parallel in-level pruning below: every candidate starts as its own component, opens no slot yet; its row rides in .w)
    uint32_t* __restrict__ par = reinterpret_cast<uint32_t*>(L.live);
    const uint32_t n_cand = L.counts[0];
    for (uint32_t c0 = 0; c0 < L.mask_words; c0 += 64) {
        const uint32_t c = c0 + (uint32_t)lane;
        unsigned long long m = c < L.mask_words ? mrow[c] : 0ull;
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        uint32_t o = base + incl - cnt;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1ull;
            const int x = L.border + (int)c * 64 + bit;
            L.cand[o] = make_float4((float)(x * L.ratio), (float)(y * L.ratio), L.Ldet[(size_t)y * L.w + x], __int_as_float(row));
            par[o] = o; par[n_cand + o] = 0u; par[2u * n_cand + o] = 0u; L.out_valid[o] = 0u;
            ++o;
        }
        base += (uint32_t)__shfl((int)incl, 63);
    }
}
// exclusive scan of the row counts of every level (one workgroup per level)
__global__ __launch_bounds__(1024)
void ak_scan_rows_kernel(const AkLevelDev* __restrict__ levels)
{
    __shared__ uint32_t part[1024];
    const AkLevelDev L = levels[blockIdx.x];
    const int n = L.h - 2 * L.border;
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = b + per < n ? b + per : n;
    uint32_t s = 0;
    for (int k = b; k < e; ++k) s += L.row_cnt[k];
    part[threadIdx.x] = s;
    r3dm_syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int t = 0; t < 1024; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; } L.counts[0] = run; }
    r3dm_syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int k = b; k < e; ++k) { L.row_off[k] = run; run += L.row_cnt[k]; }
}

// Candidate slots of every level of every image, laid out ON THE DEVICE from the counts the count pass left (no host round trip
// to size them): the slot arrays are field-major over the batch -- field f of image b starts at base_f + b * cap elements --
// and the levels of an image follow each other inside its slice.  An image with more candidates than `cap` gets empty levels
// (counts[0] = 0: every later kernel sees nothing to do for it) and reports its need; the host then grows the arrays and
// repeats the detection phase (api_features.cpp).  One thread per image.
__global__ void ak_layout_kernel(AkLevelDev* __restrict__ levels, int n_levels, unsigned char* __restrict__ slots, uint32_t cap,
                                 uint32_t n_images, AkBatchMeta* __restrict__ meta)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_images) return;
    AkLevelDev* L = levels + (size_t)b * n_levels;
    uint32_t total = 0;
    for (int i = 0; i < n_levels; ++i) total += L[i].counts[0];
    const bool over = total > cap;
    meta[b].need = total; meta[b].overflow = over ? 1u : 0u; meta[b].n_kp = 0u;
    const size_t field = (size_t)n_images * cap;            // elements per field region
    unsigned char* p = slots;
    float4* cand = (float4*)p + (size_t)b * cap;            p += field * 16;
    float4* list = (float4*)p + (size_t)b * cap;            p += field * 16;
    float*  live = (float*)p + 4 * (size_t)b * cap;         p += field * 16;
    float4* out0 = (float4*)p + (size_t)b * cap;            p += field * 16;
    float2* out1 = (float2*)p + (size_t)b * cap;            p += field * 8;
    uint32_t* valid = (uint32_t*)p + (size_t)b * cap;       p += field * 4;
    unsigned char* dl = p + (size_t)b * cap;                p += field;
    unsigned char* du = p + (size_t)b * cap;
    size_t off = 0;
    for (int i = 0; i < n_levels; ++i) {
        if (over) L[i].counts[0] = 0;
        L[i].cand = cand + off; L[i].list = list + off; L[i].live = live + 4 * off; L[i].out0 = out0 + off; L[i].out1 = out1 + off;
        L[i].out_valid = valid + off; L[i].dead_lower = dl + off; L[i].dead_upper = du + off;
        off += L[i].counts[0];
    }
}

// ---- in-level pruning (Find_Scale_Space_Extrema, first loop): candidates in raster order; a candidate within `size` of
// a kept point replaces it if stronger (the slot keeps its place), else is dropped; otherwise it is appended.  "The"
// kept point = the first one in list order.  A slot's row never decreases, so only slots within one radius of the scan
// line can still match: that live set is kept in LDS, in list order.  One wavefront per level.
constexpr int kAkLive = 3072;
// Returns false (LDS variant only) if the live set outgrows its kAkLive slots even right after a compaction; the caller then
// redoes the level with the live set in global scratch (the list is rebuilt from its start, nothing else was written).
template <bool GLOBAL_LIVE, class FP, class UP>
__device__ __forceinline__ bool ak_prune_body(const AkLevelDev& L, uint32_t n_cand, FP lx, FP ly, FP lr, UP lslot, uint32_t live_cap)
{
    const int lane = threadIdx.x;
    const float size = L.psize, size2 = size * size;
    uint32_t n_list = 0, n_live = 0;
    float cur_row = -1.0f;
    for (uint32_t c0 = 0; c0 < n_cand; c0 += 64) {
        const float4 mine = (c0 + lane < n_cand) ? L.cand[c0 + lane] : make_float4(0, 0, 0, 0);
        const uint32_t nb = (n_cand - c0 < 64u) ? n_cand - c0 : 64u;
        for (uint32_t k = 0; k < nb; ++k) {
            const float px = __shfl(mine.x, (int)k), py = __shfl(mine.y, (int)k), pr = __shfl(mine.z, (int)k);
            if ((py != cur_row && n_live >= 48u) || (!GLOBAL_LIVE && n_live >= live_cap)) {
                // new scan line and the live set is about to need a second 64-entry scan: drop the entries that no later
                // candidate can reach (they can never match -- their rows are more than one radius behind -- so leaving them
                // in while the set is small changes nothing but saves this pass on most scan lines)
                cur_row = py;
                uint32_t w = 0;
                for (uint32_t b = 0; b < n_live; b += 64) {
                    const uint32_t i = b + lane;
                    float ex = 0, ey = 0, er = 0; uint32_t es = 0; bool keep = false;
                    if (i < n_live) { ex = lx[i]; ey = ly[i]; er = lr[i]; es = lslot[i]; const float dy = py - ey; keep = dy * dy <= size2; }
                    const unsigned long long bal = __ballot(keep);
                    const uint32_t o = w + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (GLOBAL_LIVE) __threadfence();    // live set in global scratch: make the wave's own stores visible to its later loads
                    if (keep) { lx[o] = ex; ly[o] = ey; lr[o] = er; lslot[o] = es; }
                    w += (uint32_t)__builtin_popcountll(bal);
                }
                if (GLOBAL_LIVE) __threadfence();
                n_live = w;
            }
            // first live entry within the radius
            int found = -1;
            for (uint32_t b = 0; b < n_live && found < 0; b += 64) {
                const uint32_t i = b + lane;
                bool hit = false;
                if (i < n_live) { const float dx = px - lx[i], dy = py - ly[i]; hit = dx * dx + dy * dy <= size2; }
                const unsigned long long bal = __ballot(hit);
                if (bal) found = (int)b + __builtin_ctzll(bal);
            }
            if (found >= 0) {
                if (pr > lr[found]) {
                    if (lane == 0) { lx[found] = px; ly[found] = py; lr[found] = pr; L.list[lslot[found]] = make_float4(px, py, pr, 1.0f); }
                }
            } else {
                if (!GLOBAL_LIVE && n_live >= live_cap) return false;       // (wave-uniform) still full after the compaction above
                if (lane == 0) { lx[n_live] = px; ly[n_live] = py; lr[n_live] = pr; lslot[n_live] = n_list; L.list[n_list] = make_float4(px, py, pr, 1.0f); }
                ++n_live; ++n_list;
            }
            if (GLOBAL_LIVE) __threadfence();
        }
    }
    if (lane == 0) L.counts[1] = n_list;
    return true;
}

// The same rule, 64 candidates at a time.  A candidate can only interact with an EARLIER candidate of its batch if the two lie
// within 2 size of each other: whatever the earlier one does -- append itself, or take over a slot (which moves that slot from a
// place within `size` of it to its own place) -- changes the kept set only within `size` of itself or of the slot's old place.
// So every lane first looks its candidate up in the live set as it stood BEFORE the batch (one pass over the live set for all
// 64 candidates instead of one per candidate); the outcomes of the candidates without a close earlier batch-mate are final
// and are applied block-wise (appends keep raster order through a prefix count); the few candidates WITH a close earlier
// batch-mate are evaluated one by one, in order, exactly like ak_prune_body does.  Identical list, identical order.
// BYCAND (the parallel form below, one large component at a time): the candidates are cand[mem[0 .. n_cand)], a slot is known by the
// candidate that opened it and lives in out0[that candidate] (out_valid = 1) until ak_prune_emit_kernel numbers the slots.
__device__ __forceinline__ uint32_t ak_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool BYCAND>
__device__ __forceinline__ bool ak_prune_body_batched(const AkLevelDev& L, uint32_t n_cand, const uint32_t* mem, float* lx, float* ly, float* lr, uint32_t* lslot, uint32_t live_cap)
{
    const uint32_t lane = threadIdx.x & 63u;
    float4* __restrict__ out = BYCAND ? L.out0 : L.list;
    // the dependency box is a little wider than 2 size: the keep / replace rule compares dx*dx + dy*dy with size*size in floats, so
    // a candidate may be `within size` of a slot at an axis distance a few ulps beyond size; classifying a few more candidates as
    // dependent only sends them through the one-by-one path
    const float size = L.psize, size2 = size * size, box = 2.0f * size * 1.0001f + 1.0f;
    uint32_t n_list = 0, n_live = 0;
    for (uint32_t c0 = 0; c0 < n_cand; c0 += 64) {
        const uint32_t nb = (n_cand - c0 < 64u) ? n_cand - c0 : 64u;
        const bool have = lane < nb;
        const uint32_t ci = BYCAND ? (have ? ak_ld(mem + c0 + lane) : 0u) : c0 + lane;
        const float4 mine = have ? L.cand[ci] : make_float4(0, 0, 0, 0);
        const float px = mine.x, py = mine.y, pr = mine.z;
        // ---- drop the live entries that no candidate from this batch on can reach (rows more than one radius behind the
        // batch's first row; candidates come in raster order)
        {
            const float row0 = __shfl(py, 0);
            uint32_t w = 0;
            for (uint32_t b = 0; b < n_live; b += 64) {
                const uint32_t i = b + lane;
                float ex = 0, ey = 0, er = 0; uint32_t es = 0; bool keep = false;
                if (i < n_live) { ex = lx[i]; ey = ly[i]; er = lr[i]; es = lslot[i]; const float dy = row0 - ey; keep = dy * dy <= size2; }
                const unsigned long long bal = __ballot(keep);
                const uint32_t o = w + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (keep) { lx[o] = ex; ly[o] = ey; lr[o] = er; lslot[o] = es; }
                w += (uint32_t)__builtin_popcountll(bal);
            }
            n_live = w;
        }
        if (n_live + nb > live_cap) return false;           // (wave-uniform) the caller redoes the level with the live set in scratch
        // ---- which candidates have an earlier batch-mate within the 2 size box?
        bool dep = false;
        for (uint32_t j = 1; j < nb; ++j) {
            const float ox = __shfl_up(px, j), oy = __shfl_up(py, j);
            const bool valid = have && lane >= j;
            const bool near_row = valid && (py - oy) <= box;                 // raster order: py >= oy
            dep |= near_row && fabsf(px - ox) <= box;
            if (__ballot(near_row) == 0ull) break;                           // later j are even farther behind for every lane
        }
        // ---- look every candidate up in the pre-batch live set: first entry in list order within the radius
        uint32_t found = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < n_live; ++i) {
            const float dx = px - lx[i], dy = py - ly[i];
            if (found == 0xFFFFFFFFu && dx * dx + dy * dy <= size2) found = i;
        }
        const unsigned long long dmask = __ballot(dep && have);
        uint32_t p = 0;
        while (p < nb) {
            const unsigned long long rest = dmask >> p;
            const uint32_t q = rest ? p + (uint32_t)__builtin_ctzll(rest) : nb;      // next candidate that must go one by one
            // -- block [p, q): independent outcomes
            const bool in_blk = have && lane >= p && lane < q;
            const bool app = in_blk && found == 0xFFFFFFFFu;
            const bool rep = in_blk && found != 0xFFFFFFFFu && pr > lr[found];
            const unsigned long long amask = __ballot(app);
            if (rep) { lx[found] = px; ly[found] = py; lr[found] = pr; out[lslot[found]] = make_float4(px, py, pr, 1.0f); }
            if (app) {
                const uint32_t r = (uint32_t)__builtin_popcountll(amask & ((1ull << lane) - 1ull));
                const uint32_t slot = BYCAND ? ci : n_list + r;
                lx[n_live + r] = px; ly[n_live + r] = py; lr[n_live + r] = pr; lslot[n_live + r] = slot;
                out[slot] = make_float4(px, py, pr, 1.0f);
                if (BYCAND) L.out_valid[slot] = 1u;
            }
            { const uint32_t na = (uint32_t)__builtin_popcountll(amask); n_live += na; n_list += na; }
            p = q;
            if (q < nb) {
                // -- candidate q against the CURRENT live set, the whole wave scanning it (as ak_prune_body)
                const float qx = __shfl(px, (int)q), qy = __shfl(py, (int)q), qr = __shfl(pr, (int)q);
                int f = -1;
                for (uint32_t b = 0; b < n_live && f < 0; b += 64) {
                    const uint32_t i = b + lane;
                    bool hit = false;
                    if (i < n_live) { const float dx = qx - lx[i], dy = qy - ly[i]; hit = dx * dx + dy * dy <= size2; }
                    const unsigned long long bal = __ballot(hit);
                    if (bal) f = (int)b + __builtin_ctzll(bal);
                }
                if (f >= 0) {
                    if (qr > lr[f]) { if (lane == 0) { lx[f] = qx; ly[f] = qy; lr[f] = qr; out[lslot[f]] = make_float4(qx, qy, qr, 1.0f); } }
                } else {
                    const uint32_t slot = BYCAND ? (uint32_t)__shfl((int)ci, (int)q) : n_list;
                    if (lane == 0) { lx[n_live] = qx; ly[n_live] = qy; lr[n_live] = qr; lslot[n_live] = slot; out[slot] = make_float4(qx, qy, qr, 1.0f);
                                     if (BYCAND) L.out_valid[slot] = 1u; }
                    ++n_live; ++n_list;
                }
                p = q + 1;
            }
        }
    }
    if (!BYCAND && lane == 0) L.counts[1] = n_list;
    return true;
}

constexpr uint32_t kAkHandBack = 0x80000000u;          // counts[3]: this level goes back to the one-wavefront form
// The whole level by one wavefront (rounds 2-5; since round 6 the levels the parallel form below hands back -- counts[3] set: a large
// component whose live set outgrew LDS -- and, in the developer build, every level when R3DM_AK_PRUNE=0).
__global__ __launch_bounds__(64)
void ak_prune_level_kernel(const AkLevelDev* __restrict__ levels, uint32_t live_cap, int only_flagged)
{
    __shared__ float lx[kAkLive], ly[kAkLive], lr[kAkLive];
    __shared__ uint32_t lslot[kAkLive];
    const AkLevelDev L = levels[blockIdx.x];
    if (only_flagged && !(L.counts[3] & kAkHandBack)) return;
    const uint32_t n_cand = L.counts[0];
    // The live set holds the kept points within one radius of the scan line: a few dozen in practice, so it lives in LDS;
    // only if it ever outgrows kAkLive slots (bounded by the candidate count alone) is the level redone with it in scratch.
    if (!ak_prune_body_batched<false>(L, n_cand, nullptr, lx, ly, lr, lslot, live_cap))
        ak_prune_body<true>(L, n_cand, L.live, L.live + n_cand, L.live + 2 * (size_t)n_cand, (uint32_t*)(L.live + 3 * (size_t)n_cand), 0u);
}

// ---- The same rule, in parallel (round 6).  A candidate only ever meets slots within `size` of it, and a slot only ever stands where
// some candidate stood (its opener, or a stronger candidate within `size` of where it stood before).  So the candidates fall into the
// connected components of "within size of each other" (the very float predicate of the rule), and nothing one component does is seen
// by another: first slot in list order within the radius = first slot OF THE COMPONENT in the order its candidates opened them.
//   link     one lane per candidate, union-find over the earlier candidates of the rows within the radius (a wave reads them uniformly)
//   flatten  root, member count and last member of every component
//   small    a component of <= 64 candidates by one wavefront with its slots in registers (lane = slot, in opening order); a lone
//            candidate opens its slot on the spot; larger components are queued
//   large    a queued component by one wavefront: members gathered in raster order, then the batched rule above with the live set in LDS
//            (a live set that outgrows LDS hands the level back to ak_prune_level_kernel)
//   emit     slots numbered by a running count of the openers in raster order = the order the sequential rule appends them
// Scratch (n = candidates of the level; all of it idle until the refinement): live[0..n) parent, [n..2n) member count (at roots),
// [2n..3n) last member (at roots), [3n..4n) queue of (root, where its members are gathered) pairs; out0[i] / out_valid[i] = the slot
// candidate i opened; out1 = the gathered members of the queued components; counts[2] = queued components, counts[3] = gather cursor
// (< n) with the hand-back flag as its top bit.
__device__ __forceinline__ uint32_t ak_uf_find(uint32_t* par, uint32_t x)
{
    for (;;) {
        const uint32_t p = ak_ld(par + x);
        if (p == x) return x;
        const uint32_t g = ak_ld(par + p);
        if (g == p) return p;
        __hip_atomic_store(par + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // path halving: g is an ancestor of x whatever else happens
        x = g;
    }
}
__device__ __forceinline__ void ak_uf_union(uint32_t* par, uint32_t a, uint32_t b)
{
    for (;;) {
        a = ak_uf_find(par, a); b = ak_uf_find(par, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }                                // the later root goes under the earlier one: links point backwards, no cycle
        if (atomicCAS(par + a, a, b) == a) return;
    }
}
__global__ __launch_bounds__(256)
void ak_prune_link_kernel(const AkLevelDev* __restrict__ levels)
{
    const AkLevelDev L = levels[blockIdx.y];
    const uint32_t n = L.counts[0];
    uint32_t* par = reinterpret_cast<uint32_t*>(L.live);
    const float size = L.psize, size2 = size * size;
    const int reach = (int)(size / L.ratio) + 1;                     // rows: dy = rows x ratio exactly
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = gridDim.x * 4u;
    for (uint32_t c0 = wave * 64u; c0 < n; c0 += n_waves * 64u) {
        const uint32_t i = c0 + lane;
        const bool have = i < n;
        const float4 me = L.cand[have ? i : c0];
        int row_lo = __builtin_amdgcn_readfirstlane(__float_as_int(me.w)) - reach;     // lane 0 holds the wave's first (lowest) row
        row_lo = row_lo < 0 ? 0 : row_lo;
        const uint32_t lo = L.row_off[row_lo];
        const uint32_t hi = c0 + 64u < n ? c0 + 64u : n;
        for (uint32_t j = lo; j + 1u < hi; ++j) {
            const float4 o = L.cand[j];                                                 // (wave-uniform address)
            const float dx = me.x - o.x, dy = me.y - o.y;
            if (have && j < i && dx * dx + dy * dy <= size2) ak_uf_union(par, i, j);
        }
    }
}
__global__ __launch_bounds__(256)
void ak_prune_flatten_kernel(const AkLevelDev* __restrict__ levels)
{
    const AkLevelDev L = levels[blockIdx.y];
    const uint32_t n = L.counts[0];
    uint32_t* par = reinterpret_cast<uint32_t*>(L.live);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t r = ak_uf_find(par, i);
        if (r != i) __hip_atomic_store(par + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(par + n + r, 1u);
        atomicMax(par + 2u * n + r, i);
    }
}
constexpr uint32_t kAkSmallComp = 64;
__global__ __launch_bounds__(256)
void ak_prune_small_kernel(const AkLevelDev* __restrict__ levels)
{
    const AkLevelDev L = levels[blockIdx.y];
    const uint32_t n = L.counts[0];
    const uint32_t* __restrict__ par = reinterpret_cast<const uint32_t*>(L.live);
    uint32_t* __restrict__ queue = reinterpret_cast<uint32_t*>(L.live) + 3u * (size_t)n;
    const float size = L.psize, size2 = size * size;
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = gridDim.x * 4u;
    for (uint32_t c0 = wave * 64u; c0 < n; c0 += n_waves * 64u) {
        const uint32_t i = c0 + lane;
        const bool have = i < n;
        const bool root = have && par[i] == i;
        const uint32_t sz = root ? par[n + i] : 0u, last = root ? par[2u * n + i] : 0u;
        if (sz == 1u) { const float4 c = L.cand[i]; L.out0[i] = make_float4(c.x, c.y, c.z, 1.0f); L.out_valid[i] = 1u; }
        if (sz > kAkSmallComp) {
            const uint32_t at = atomicAdd(L.counts + 2, 1u), first = atomicAdd(L.counts + 3, sz);     // (counts[3]: gather cursor until the large pass; then the hand-back flag)
            queue[2u * at] = i; queue[2u * at + 1u] = first;
        }
        unsigned long long todo = __ballot(sz >= 2u && sz <= kAkSmallComp);
        while (todo) {
            const int bit = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t r = c0 + (uint32_t)bit, r_last = (uint32_t)__shfl((int)last, bit);
            // the slots of this component: lane = slot, in opening order
            float sx = 0.f, sy = 0.f, sr = 0.f; uint32_t sidx = 0u, n_slots = 0u;
            for (uint32_t b = r; b <= r_last; b += 64u) {
                const uint32_t j = b + lane;
                const bool in = j <= r_last;
                const uint32_t rt = in ? par[j] : 0xFFFFFFFFu;
                const float4 cj = in ? L.cand[j] : make_float4(0, 0, 0, 0);
                unsigned long long mm = __ballot(rt == r);
                while (mm) {
                    const int k = __builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const float px = __shfl(cj.x, k), py = __shfl(cj.y, k), pr = __shfl(cj.z, k);
                    const float dx = px - sx, dy = py - sy;
                    const unsigned long long bal = __ballot(lane < n_slots && dx * dx + dy * dy <= size2);
                    if (bal) {
                        const int f = __builtin_ctzll(bal);
                        if (pr > __shfl(sr, f) && (int)lane == f) { sx = px; sy = py; sr = pr; }
                    } else {
                        if (lane == n_slots) { sx = px; sy = py; sr = pr; sidx = b + (uint32_t)k; }
                        ++n_slots;
                    }
                }
            }
            if (lane < n_slots) { L.out0[sidx] = make_float4(sx, sy, sr, 1.0f); L.out_valid[sidx] = 1u; }
        }
    }
}
__global__ __launch_bounds__(64)
void ak_prune_large_kernel(const AkLevelDev* __restrict__ levels, uint32_t live_cap)
{
    __shared__ float lx[kAkLive], ly[kAkLive], lr[kAkLive];
    __shared__ uint32_t lslot[kAkLive];
    const AkLevelDev L = levels[blockIdx.y];
    const uint32_t n = L.counts[0], n_queued = L.counts[2];
    const uint32_t* __restrict__ par = reinterpret_cast<const uint32_t*>(L.live);
    const uint32_t* __restrict__ queue = par + 3u * (size_t)n;
    uint32_t* __restrict__ members = reinterpret_cast<uint32_t*>(L.out1);
    const uint32_t lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < n_queued; q += gridDim.x) {
        const uint32_t r = queue[2u * q], first = queue[2u * q + 1u], r_last = par[2u * n + r], sz = par[n + r];
        uint32_t w = 0;
        for (uint32_t b = r; b <= r_last; b += 64u) {
            const uint32_t j = b + lane;
            const bool is = j <= r_last && par[j] == r;
            const unsigned long long bal = __ballot(is);
            if (is) members[first + w + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = j;
            w += (uint32_t)__builtin_popcountll(bal);
        }
        __threadfence();                                     // the wave's own stores, before it reads them back (ak_ld: past the L1)
        if (!ak_prune_body_batched<true>(L, sz, members + first, lx, ly, lr, lslot, live_cap)) {
            if (lane == 0) atomicOr(L.counts + 3, kAkHandBack);
            return;
        }
    }
}
__global__ __launch_bounds__(1024)
void ak_prune_emit_kernel(const AkLevelDev* __restrict__ levels)
{
    __shared__ uint32_t wsum[16];
    const AkLevelDev L = levels[blockIdx.x];
    const uint32_t n = L.counts[0];
    if (L.counts[3] & kAkHandBack) return;                  // ak_prune_level_kernel rebuilds this level's list
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 1024u) {
        const uint32_t i = c0 + tid;
        const bool open = i < n && L.out_valid[i] != 0u;
        const unsigned long long bal = __ballot(open);
        if (lane == 0) wsum[wave] = (uint32_t)__builtin_popcountll(bal);
        r3dm_syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) { const uint32_t v = wsum[k]; before += k < wave ? v : 0u; all += v; }
        if (open) L.list[base + before + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = L.out0[i];
        base += all;
        r3dm_syncthreads();
    }
    if (tid == 0) L.counts[1] = base;
}

// ---- cross-level pruning.  mode 0 ("lower"): a point q of level i-1 dies when some point p of level i lies within
// p.size of it with a larger response.  mode 1 ("upper"): a point v of level i+1 dies when some point p of level i that
// survived mode 0 lies within v.size of it with a larger response.  One thread per victim.
//
// The test is "any killer within the radius", so the order in which killers are visited is free.  The lists are in insertion
// order of a raster scan, i.e. almost sorted by row: ak_list_ranges_kernel records the row span (min y, max y) of every chunk of
// 256 list entries, and a block of 256 victims only loads the killer chunks whose span comes within the radius of its own --
// a few chunks instead of all of them (the all-pairs form was quadratic: 6.9 ms per call at 72 k keypoints per image).  The
// spans are taken from the data, so nothing depends on the lists actually being sorted.
__global__ __launch_bounds__(256)
void ak_list_ranges_kernel(const AkLevelDev* __restrict__ levels)
{
    __shared__ float smin[4], smax[4];
    const AkLevelDev L = levels[blockIdx.y];
    const uint32_t n = L.counts[1];
    float2* __restrict__ ranges = reinterpret_cast<float2*>(L.live);       // the in-level pruning is done with its scratch
    for (uint32_t c = blockIdx.x; c * 256u < n; c += gridDim.x) {
        const uint32_t j = c * 256u + threadIdx.x;
        float lo = __builtin_inff(), hi = -__builtin_inff();
        if (j < n) { const float y = L.list[j].y; lo = y; hi = y; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
        if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
        r3dm_syncthreads();
        if (threadIdx.x == 0) ranges[c] = make_float2(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])), fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
        r3dm_syncthreads();
    }
}

__global__ __launch_bounds__(256)
void ak_cross_kernel(const AkLevelDev* __restrict__ levels, int n_levels, int mode)
{
    __shared__ float4 sk[256];
    __shared__ unsigned char sdead[256];
    // blockIdx.y = image * n_levels + victim level; the list lengths live on the device, so the victims are covered by a
    // grid-stride loop over chunks of 256 instead of a grid sized by the host
    const int vi = (int)(blockIdx.y % (unsigned)n_levels);       // victim level
    const int ki = mode == 0 ? vi + 1 : vi - 1;                  // killer level
    if (ki < 0 || ki >= n_levels) return;
    const AkLevelDev V = levels[blockIdx.y], K = levels[(int)blockIdx.y + (ki - vi)];
    const uint32_t nv = V.counts[1], nk = K.counts[1];
    const float r = mode == 0 ? K.psize : V.psize, r2 = r * r;
    const float2* __restrict__ vr = reinterpret_cast<const float2*>(V.live);
    const float2* __restrict__ kr = reinterpret_cast<const float2*>(K.live);
    const uint32_t n_kchunks = (nk + 255u) / 256u;
    __shared__ float2 sspan[256];
    for (uint32_t q0 = blockIdx.x * 256u; q0 < nv; q0 += 256u * gridDim.x) {      // workgroup-uniform bounds
        const uint32_t q = q0 + threadIdx.x;
        const float4 v = q < nv ? V.list[q] : make_float4(0, 0, 0, 0);
        const float2 span = vr[q0 >> 8];                                          // rows of this victim block
        const float lo = span.x - r - 1.0f, hi = span.y + r + 1.0f;               // (one row of slack: the comparison below is exact anyway)
        bool dead = false;
        for (uint32_t kb = 0; kb < n_kchunks; kb += 256u) {                       // the spans of 256 killer chunks at a time, through LDS
            r3dm_syncthreads();
            if (kb + threadIdx.x < n_kchunks) sspan[threadIdx.x] = kr[kb + threadIdx.x];
            r3dm_syncthreads();
            const uint32_t nb = n_kchunks - kb < 256u ? n_kchunks - kb : 256u;
            for (uint32_t kk = blockIdx.z; kk < nb; kk += gridDim.z) {           // the killer chunks are dealt to gridDim.z workgroups (they only ever store a 1)
                const float2 ks = sspan[kk];
                if (ks.y < lo || ks.x > hi) continue;                             // workgroup-uniform: no killer of this chunk can reach a victim of the block
                const uint32_t j0 = (kb + kk) * 256u, j = j0 + threadIdx.x;
                if (j < nk) { sk[threadIdx.x] = K.list[j]; sdead[threadIdx.x] = (mode == 1 && K.dead_lower[j]) ? 1 : 0; }
                r3dm_syncthreads();
                const uint32_t cnt = nk - j0 < 256u ? nk - j0 : 256u;
#pragma unroll 8
                for (uint32_t t = 0; t < cnt; ++t) {                              // (a dependent LDS read per killer made one chunk ~16 us)
                    const float4 p = sk[t];
                    const float dx = p.x - v.x, dy = p.y - v.y;
                    dead |= !sdead[t] && (dx * dx + dy * dy <= r2) && (p.z > v.z);
                }
                r3dm_syncthreads();
            }
        }
        if (q < nv && dead) (mode == 0 ? V.dead_lower : V.dead_upper)[q] = 1;
    }
}

// ---- sub-pixel refinement + dominant gradient direction (Do_Subpixel_Refinement, Compute_Main_Orientation up to the
// vector whose angle the host takes).  One thread per list entry; out: (x, y, size, response), (maxX, maxY), valid.
__device__ const float kGauss25[7][7] = {
    { 0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f },
    { 0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f },
    { 0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f },
    { 0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f },
    { 0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f },
    { 0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f },
    { 0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f } };

__device__ __forceinline__ float ak_fast_atan2(float y, float x)
{
    const float kR = 0x1.ca5dc2p+5f;                              // (float)(180 / CV_PI)
    const float p1 = 0.9997878412794807f * kR, p3 = -0.3258083974640975f * kR, p5 = 0.1555786518463281f * kR, p7 = -0.04432655554792128f * kR;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + 0x1p-52f); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + 0x1p-52f); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a * 0x1.1df46ap-6f;                                    // (float)(CV_PI / 180)
}

// sample offsets (dy, dx) of Sample_Derivative_Response_Radius6 in its loop order: i, j in [-6, 6], i*i + j*j < 36
__device__ const signed char kRad6[109][2] = { {-5, -3}, {-5, -2}, {-5, -1}, {-5, 0}, {-5, 1}, {-5, 2}, {-5, 3}, {-4, -4}, {-4, -3}, {-4, -2}, {-4, -1}, {-4, 0}, {-4, 1}, {-4, 2}, {-4, 3}, {-4, 4}, {-3, -5}, {-3, -4}, {-3, -3}, {-3, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-3, 2}, {-3, 3}, {-3, 4}, {-3, 5}, {-2, -5}, {-2, -4}, {-2, -3}, {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2}, {-2, 3}, {-2, 4}, {-2, 5}, {-1, -5}, {-1, -4}, {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3}, {-1, 4}, {-1, 5}, {0, -5}, {0, -4}, {0, -3}, {0, -2}, {0, -1}, {0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {1, -5}, {1, -4}, {1, -3}, {1, -2}, {1, -1}, {1, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {2, -5}, {2, -4}, {2, -3}, {2, -2}, {2, -1}, {2, 0}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {2, 5}, {3, -5}, {3, -4}, {3, -3}, {3, -2}, {3, -1}, {3, 0}, {3, 1}, {3, 2}, {3, 3}, {3, 4}, {3, 5}, {4, -4}, {4, -3}, {4, -2}, {4, -1}, {4, 0}, {4, 1}, {4, 2}, {4, 3}, {4, 4}, {5, -3}, {5, -2}, {5, -1}, {5, 0}, {5, 1}, {5, 2}, {5, 3} };

// One wavefront per list entry.  blockIdx.y = image; the entries of ALL levels of the image form one flat index range
// (prefix sums of the list lengths, which live on the device), covered by a grid-stride loop -- every wavefront gets the same
// share whatever the distribution over the levels.  Compute_Main_Orientation's order-sensitive parts run in parallel without
// changing a single float: the counting sort places sample i at start[key] + count[key] - 1 - #{j < i, same key} (what the
// reference's "--slice[key]" loop produces), every lane sums ONE of the 42 sliding windows in the sorted order, and the winner
// is the earliest window with the largest norm (the reference replaces only on a strictly larger norm).
__global__ __launch_bounds__(64)
void ak_refine_kernel(const AkLevelDev* __restrict__ levels, int n_levels)
{
    __shared__ float resX[112], resY[112];
    __shared__ uint32_t scnt[48], sstart[48];
    __shared__ unsigned char sinv[128];
    const AkLevelDev* __restrict__ Lb = levels + (size_t)blockIdx.y * n_levels;
    const int lane = threadIdx.x;
    uint32_t pre[17];
    pre[0] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) pre[i + 1] = pre[i] + (i < n_levels ? Lb[i].counts[1] : 0u);
    const uint32_t total = pre[16];
    constexpr int slices = 42, win = 7;
    const float ang_step = 0x1.32614ep-3f;                // (float)(2.0 * CV_PI / 42)
    for (uint32_t f = blockIdx.x; f < total; f += gridDim.x) {
        int li = 0;
#pragma unroll
        for (int i = 1; i < 16; ++i) li += (f >= pre[i]) ? 1 : 0;
        const AkLevelDev L = Lb[li];
        const uint32_t j = f - pre[li];
        const float* __restrict__ ldet = L.Ldet;
        const int cols = L.w;
        const float ratio = L.ratio;
        float4 kp = L.list[j];
        bool drop = L.dead_lower[j] || L.dead_upper[j];             // wave-uniform
        float dx = 0.0f, dy = 0.0f;
        if (!drop) {
            const int x = (int)(kp.x / ratio), y = (int)(kp.y / ratio);
            const float Dx = 0.5f * (ldet[y * cols + x + 1] - ldet[y * cols + x - 1]);
            const float Dy = 0.5f * (ldet[(y + 1) * cols + x] - ldet[(y - 1) * cols + x]);
            const float Dxx = ldet[y * cols + x + 1] + ldet[y * cols + x - 1] - 2.0f * ldet[y * cols + x];
            const float Dyy = ldet[(y + 1) * cols + x] + ldet[(y - 1) * cols + x] - 2.0f * ldet[y * cols + x];
            const float Dxy = 0.25f * (ldet[(y + 1) * cols + x + 1] + ldet[(y - 1) * cols + x - 1] -
                                       ldet[(y - 1) * cols + x + 1] - ldet[(y + 1) * cols + x - 1]);
            const float b0 = -Dx, b1 = -Dy;
            double d = (double)Dxx * Dyy - (double)Dxy * Dxy;
            if (d != 0.) {
                d = 1. / d;
                const double t = (float)(((double)b0 * Dyy - (double)b1 * Dxy) * d);
                dy = (float)(((double)b1 * Dxx - (double)b0 * Dxy) * d);
                dx = (float)t;
            }
            drop = fabsf(dx) > 1.0f || fabsf(dy) > 1.0f;            // wave-uniform: every lane computed the same values
        }
        if (drop) {
            if (lane == 0) { L.out0[j] = make_float4(0, 0, 0, 0); L.out1[j] = make_float2(0, 0); L.out_valid[j] = 0; }
            continue;
        }
        kp.x += dx * ratio; kp.y += dy * ratio;
        const float size = L.psize * 2.0f;
        const int scale = (int)(0.5f * size / ratio + 0.5f);
        const int x0 = (int)(kp.x / ratio + 0.5f), y0 = (int)(kp.y / ratio + 0.5f);
        // two samples per lane (k = lane and k = lane + 64 < 109): weighted responses and the angle slice of each
        float rxs[2], rys[2]; int key[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = lane + 64 * q;
            const int kk = k < 109 ? k : 0;
            const int i = kRad6[kk][0], jj = kRad6[kk][1];
            const float wgt = kGauss25[i < 0 ? -i : i][jj < 0 ? -jj : jj];
            const size_t p = (size_t)(y0 + i * scale) * cols + (x0 + jj * scale);
            rxs[q] = wgt * L.Lx[p]; rys[q] = wgt * L.Ly[p];
            key[q] = k < 109 ? (int)(ak_fast_atan2(rys[q], rxs[q]) / ang_step) : -1;
        }
        // counting sort by angle slice: sample i goes to start[key] + count[key] - 1 - #{j < i with the same key} (what the reference's
        // "--slice[key]" loop produces).  #{j < i, same key} = the sample's place inside its slice in the order (key, i): a bitonic
        // sort of the 128 composite keys (key << 8 | i), two per lane -- 28 compare-exchange steps, where one ballot pair per slice was 43
        // rounds of scalar bookkeeping (half of this kernel's instructions); slice populations by LDS atomics, their prefix by a lane scan.
        if (lane < 48) scnt[lane] = 0u;
        r3dm_syncthreads();
        uint32_t e0 = ((uint32_t)key[0] << 8) | (uint32_t)lane;                                   // sample lane (always one of the 109)
        uint32_t e1 = key[1] >= 0 ? (((uint32_t)key[1] << 8) | (uint32_t)(lane + 64)) : (0xFF00u | (uint32_t)(lane + 64));
        atomicAdd(&scnt[key[0]], 1u);
        if (key[1] >= 0) atomicAdd(&scnt[key[1]], 1u);
#pragma unroll
        for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                if (stride == 64) {                                  // (size 128: the lane's own pair, ascending)
                    const uint32_t lo = e0 < e1 ? e0 : e1, hi = e0 < e1 ? e1 : e0;
                    e0 = lo; e1 = hi;
                } else {
                    const bool lower = (lane & stride) == 0;
                    const uint32_t o0 = (uint32_t)__shfl_xor((int)e0, stride), o1 = (uint32_t)__shfl_xor((int)e1, stride);
                    // element index = lane + 64 q: bit `size` of it decides the direction (size 64: q itself; size 128: always up)
                    const bool up0 = size >= 64 ? true : (lane & size) == 0;
                    const bool up1 = size == 64 ? false : (size == 128 ? true : (lane & size) == 0);
                    const uint32_t mn0 = e0 < o0 ? e0 : o0, mx0 = e0 < o0 ? o0 : e0, mn1 = e1 < o1 ? e1 : o1, mx1 = e1 < o1 ? o1 : e1;
                    e0 = (lower == up0) ? mn0 : mx0;
                    e1 = (lower == up1) ? mn1 : mx1;
                }
            }
        }
        // sorted position p = lane + 64 q holds sample (e & 0xFF): tell that sample where it stands
        sinv[e0 & 0xFFu] = (unsigned char)lane;
        sinv[e1 & 0xFFu] = (unsigned char)(lane + 64);
        r3dm_syncthreads();
        uint32_t my_start = 109;
        {
            const uint32_t c = lane <= (uint32_t)slices ? scnt[lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= off) incl += o; }
            if (lane <= (uint32_t)slices) { my_start = incl - c; sstart[lane] = incl - c; }
        }
        r3dm_syncthreads();
        const uint32_t pos0 = 2u * sstart[key[0]] + scnt[key[0]] - 1u - (uint32_t)sinv[lane];
        const uint32_t k1 = key[1] >= 0 ? (uint32_t)key[1] : 0u;
        const uint32_t pos1 = 2u * sstart[k1] + scnt[k1] - 1u - (uint32_t)sinv[lane + 64];
        resX[pos0] = rxs[0]; resY[pos0] = rys[0];
        if (key[1] >= 0) { resX[pos1] = rxs[1]; resY[pos1] = rys[1]; }
        r3dm_syncthreads();
        // window sn = lane: slices [sn, sn + win) of the circle, summed in sorted order (resX / resY are in sorted order now)
        float sumX = 0.0f, sumY = 0.0f, nrm = -1.0f;
        const int sn = lane;
        const int e_lo = (int)__shfl((int)my_start, sn + win <= slices ? sn + win : slices);        // start[sn + win], or start[42] = 109 - #{key 42}
        const int e_wrap = (int)__shfl((int)my_start, sn + win > slices ? sn + win - slices : 0);  // start[remain]
        if (lane < slices) {
            for (int i = (int)my_start; i < e_lo; ++i) { sumX += resX[i]; sumY += resY[i]; }
            if (sn > slices - win) for (int i = 0; i < e_wrap; ++i) { sumX += resX[i]; sumY += resY[i]; }
            nrm = sumX * sumX + sumY * sumY;
        }
        int best = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float on = __shfl_xor(nrm, off); const int ob = __shfl_xor(best, off);
            const float ox = __shfl_xor(sumX, off), oy = __shfl_xor(sumY, off);
            if (on > nrm || (on == nrm && ob < best)) { nrm = on; best = ob; sumX = ox; sumY = oy; }
        }
        if (lane == 0) {
            L.out0[j] = make_float4(kp.x, kp.y, size, kp.z);
            L.out1[j] = make_float2(sumX, sumY);
            L.out_valid[j] = 1;
        }
        r3dm_syncthreads();                                       // the next entry overwrites the LDS arrays
    }
}

// ---- the surviving keypoints of every image, compacted in the reference's order (evolution level, then list order) into one
// record array per image: what crosses to the host for the angle (getAngleV2 = atan2 of the host libm, utils.h of fast-akaze)
// and comes back as the LIOP warp -- 32 bytes per keypoint, one copy each way.  One workgroup per (level, image): it counts the
// survivors of the levels before its own (4 bytes per list entry, a few hundred KB at most), then writes its level's records behind
// them.  (Rounds 2-5: one workgroup of 256 per image walking all levels, two barriers per 256 entries: 184 us per batch of eight.)
__global__ __launch_bounds__(1024)
void ak_compact_kernel(const AkLevelDev* __restrict__ levels, int n_levels, AkKpRec* __restrict__ recs, uint32_t cap,
                       AkBatchMeta* __restrict__ meta)
{
    __shared__ uint32_t wave_cnt[16];
    const uint32_t b = blockIdx.y;
    const int mine = (int)blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    AkKpRec* out = recs + (size_t)b * cap;
    uint32_t before = 0;
    for (int i = 0; i < mine; ++i) {
        const AkLevelDev L = levels[(size_t)b * n_levels + i];
        const uint32_t n = L.counts[1];
        for (uint32_t j = tid; j < n; j += 1024u) before += L.out_valid[j] != 0 ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off);
    if (lane == 0) wave_cnt[wave] = before;
    r3dm_syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) base += wave_cnt[q];
    r3dm_syncthreads();
    const AkLevelDev L = levels[(size_t)b * n_levels + mine];
    const uint32_t n = L.counts[1];
    for (uint32_t j0 = 0; j0 < n; j0 += 1024u) {
        const uint32_t j = j0 + tid;
        const bool v = j < n && L.out_valid[j] != 0;
        const unsigned long long bal = __ballot(v);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
        r3dm_syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (uint32_t q = 0; q < 16u; ++q) { const uint32_t cw = wave_cnt[q]; woff += q < wave ? cw : 0u; tot += cw; }
        if (v) {
            const float4 o0 = L.out0[j]; const float2 o1 = L.out1[j];
            AkKpRec r; r.x = o0.x; r.y = o0.y; r.size = o0.z; r.response = o0.w; r.max_x = o1.x; r.max_y = o1.y; r.level = (uint32_t)mine; r.pad = 0;
            out[base + woff + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull))] = r;
        }
        base += tot;
        r3dm_syncthreads();
    }
    if (mine == n_levels - 1 && tid == 0) meta[b].n_kp = base;
}

// ---- MLDB-486 descriptor (MLDB_Full_Descriptor_InvokerV2, AKAZEFeatures.cpp:1790-1909): 2 keypoints per wavefront, one lane per
// grid cell (4 + 9 + 16 cells of the 2x2 / 3x3 / 4x4 grids) accumulating its samples in the reference's order; the 486
// comparisons are then read off a (value a, value b) table, one output byte per lane.
__global__ __launch_bounds__(64)
void ak_mldb_kernel(const AkLevelDev* __restrict__ levels, const AkMldbItem* __restrict__ items, uint32_t n,
                    const unsigned char* __restrict__ pairs /* [486][2] */, unsigned char* __restrict__ out /* [n][61] */)
{
    __shared__ int32_t vals[2][96];
    const uint32_t half = threadIdx.x >> 5, lane = threadIdx.x & 31u;
    const uint32_t kp = blockIdx.x * 2 + half;
    const bool live = kp < n;
    if (live && lane < 29) {
        const AkMldbItem it = items[kp];
        const AkLevelDev L = levels[it.level];
        int step, ncell, base, c = (int)lane;
        if (c < 4) { step = 10; ncell = 2; base = 0; }
        else if (c < 13) { step = 7; ncell = 3; base = 12; c -= 4; }
        else { step = 5; ncell = 4; base = 39; c -= 13; }
        const int i0 = -10 + (c / ncell) * step, j0 = -10 + (c % ncell) * step;
        const float co = it.co, si = it.si, scale = it.scale, xf = it.xf, yf = it.yf;
        float di = 0.0f, dx = 0.0f, dy = 0.0f;
        int nsamples = 0;
        for (int k = i0; k < i0 + step; ++k)
            for (int l = j0; l < j0 + step; ++l) {
                const float sample_y = yf + (l * co * scale + k * si * scale);
                const float sample_x = xf + (-l * si * scale + k * co * scale);
                const int y1 = (int)(sample_y + 0.5f), x1 = (int)(sample_x + 0.5f);
                const size_t p = (size_t)y1 * L.w + x1;
                di += L.Lt[p];
                const float rx = L.Lx[p], ry = L.Ly[p];
                const float rry = rx * co + ry * si;
                const float rrx = -rx * si + ry * co;
                dx += rrx; dy += rry;
                nsamples++;
            }
        di /= nsamples; dx /= nsamples; dy /= nsamples;
        const int32_t a = (int32_t)__float_as_uint(di), b = (int32_t)__float_as_uint(dx), d = (int32_t)__float_as_uint(dy);
        vals[half][base + 3 * c] = a ^ (a < 0 ? 0x7fffffff : 0);            // CV_TOGGLE_FLT
        vals[half][base + 3 * c + 1] = b ^ (b < 0 ? 0x7fffffff : 0);
        vals[half][base + 3 * c + 2] = d ^ (d < 0 ? 0x7fffffff : 0);
    }
    r3dm_syncthreads();
    if (!live) return;
    for (uint32_t byte = lane; byte < 61; byte += 32) {
        uint32_t v = 0;
        for (uint32_t bit = 0; bit < 8; ++bit) {
            const uint32_t dpos = byte * 8 + bit;
            if (dpos < 486 && vals[half][pairs[2 * dpos]] > vals[half][pairs[2 * dpos + 1]]) v |= 1u << bit;
        }
        out[(size_t)kp * 61 + byte] = (unsigned char)v;
    }
}

// ---- image preparation of R3DFeaturesThread::processWorkItem (src/threads/R3DFeaturesThread.cpp:163-191): 8-bit BGR ->
// float (convertTo, scale 1/255) -> gray (cvtColor BGR2GRAY on floats: 0.114 B + 0.587 G + 0.299 R)
__global__ __launch_bounds__(256)
void ak_bgr_to_gray_kernel(const unsigned char* __restrict__ bgr, float* __restrict__ gray, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float sc = (float)(1.0 / 255.0);
    const float b = (float)bgr[3 * i] * sc, g = (float)bgr[3 * i + 1] * sc, r = (float)bgr[3 * i + 2] * sc;
    gray[i] = b * 0.114f + g * 0.587f + r * 0.299f;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
hipError_t ak_bgr_to_gray(hipStream_t st, const unsigned char* bgr, float* gray, size_t n)
{
    hipLaunchKernelGGL(ak_bgr_to_gray_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bgr, gray, n);
    return hipGetLastError();
}

hipError_t ak_mldb(hipStream_t st, const AkLevelDev* levels, const AkMldbItem* items, uint32_t n, const unsigned char* pairs, unsigned char* out)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ak_mldb_kernel, dim3((n + 1) / 2), dim3(64), 0, st, levels, items, n, pairs, out);
    return hipGetLastError();
}

static dim3 ak_grid(int w, int h, int B) { return dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)B); }
// strip kernels: a workgroup owns 64 x (4 R) pixels (R rows per wave)
static dim3 ak_strip_grid(int w, int h, int B, int R) { return dim3((unsigned)((w + 63) / 64), (unsigned)((h + 4 * R - 1) / (4 * R)), (unsigned)B); }

// every launcher: B = images of the batch (same size), buffers hold B planes of the launch's w x h back to back
hipError_t ak_gaussian(hipStream_t st, const float* src, float* tmp, float* dst, int w, int h, int B, const AkTaps& kf)
{
    if (kf.n == 5 || kf.n == 9) {
        hipLaunchKernelGGL(ak_gauss_kernel, ak_strip_grid(w, h, B, 4), dim3(256), 0, st, src, dst, w, h, kf);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ak_gauss_rows_kernel, ak_grid(w, h, B), dim3(256), 0, st, src, tmp, w, h, kf);
    hipLaunchKernelGGL(ak_gauss_cols_kernel, ak_grid(w, h, B), dim3(256), 0, st, tmp, dst, w, h, kf);
    return hipGetLastError();
}
hipError_t ak_scharr(hipStream_t st, const float* src, float* rd, float* rs, float* Lx, float* Ly, int w, int h, int B)
{
    hipLaunchKernelGGL(ak_scharr_rows_kernel, ak_grid(w, h, B), dim3(256), 0, st, src, rd, rs, w, h);
    hipLaunchKernelGGL(ak_scharr_cols_kernel, ak_grid(w, h, B), dim3(256), 0, st, rd, rs, Lx, Ly, w, h);
    return hipGetLastError();
}
hipError_t ak_scaled_deriv_xy(hipStream_t st, const float* src, float* dst_x, float* dst_y, int w, int h, int B, int s)
{
    hipLaunchKernelGGL(ak_sderiv_xy_kernel, ak_strip_grid(w, h, B, 4), dim3(256), 0, st, src, dst_x, dst_y, w, h, s);
    return hipGetLastError();
}
hipError_t ak_scaled_deriv_det(hipStream_t st, const float* lx, const float* ly, float* ldet, int w, int h, int B, int s)
{
    hipLaunchKernelGGL(ak_sderiv_det_kernel, ak_strip_grid(w, h, B, 2), dim3(256), 0, st, lx, ly, ldet, w, h, s);
    return hipGetLastError();
}
// src = the smoothed image (Gaussian 1.0 of the input); both kernels cover the interior [1, w-2] x [1, h-2] in 64 x 16 tiles
hipError_t ak_modg_max(hipStream_t st, const float* src, int w, int h, int B, uint32_t* out_max)
{
    if (w < 3 || h < 3) return hipSuccess;
    const int tiles_y = (h - 2 + 15) / 16;
    hipLaunchKernelGGL(ak_modg_max_kernel, dim3((unsigned)((w - 2 + 63) / 64), (unsigned)std::min(tiles_y, 24), (unsigned)B), dim3(256), 0, st, src, w, h, out_max);
    return hipGetLastError();
}
hipError_t ak_modg_hist(hipStream_t st, const float* src, int w, int h, int B, const uint32_t* hmax_bits, int nbins, uint32_t* hist)
{
    if (nbins > 512) return hipErrorInvalidValue;
    if (w < 3 || h < 3) return hipSuccess;
    const int tiles_y = (h - 2 + 15) / 16;
    hipLaunchKernelGGL(ak_modg_hist_kernel, dim3((unsigned)((w - 2 + 63) / 64), (unsigned)std::min(tiles_y, 24), (unsigned)B), dim3(256), 0, st, src, w, h, hmax_bits, nbins, hist);
    return hipGetLastError();
}
hipError_t ak_kcontrast(hipStream_t st, const uint32_t* hmax_bits, const uint32_t* hist, int nbins, uint32_t total, int have_hist, float* inv_k2, int B)
{
    hipLaunchKernelGGL(ak_kcontrast_kernel, dim3((unsigned)B), dim3(1), 0, st, hmax_bits, hist, nbins, total, have_hist, inv_k2);
    return hipGetLastError();
}
hipError_t ak_scharr_g2(hipStream_t st, const float* src, float* dst, int w, int h, int B, const float* inv_k2)
{
    hipLaunchKernelGGL(ak_scharr_g2_kernel, ak_strip_grid(w, h, B, 4), dim3(256), 0, st, src, dst, w, h, inv_k2);
    return hipGetLastError();
}
hipError_t ak_fed_step(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, float step_size)
{
    hipLaunchKernelGGL(ak_fed_step_kernel, ak_strip_grid(w, h, B, 4), dim3(256), 0, st, Lt, Lf, out, w, h, step_size);
    return hipGetLastError();
}
// n_steps <= 4 FED steps tau[0..n) in one launch (small levels: see ak_fed_multi_kernel)
hipError_t ak_fed_multi(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, const float* tau, int n_steps)
{
    if (n_steps < 1 || n_steps > 4) return hipErrorInvalidValue;
    AkFedSteps fs{};
    for (int k = 0; k < n_steps; ++k) fs.tau[k] = tau[k];
    fs.n = n_steps;
    hipLaunchKernelGGL(ak_fed_multi_kernel, dim3((unsigned)((w + 31) / 32), (unsigned)((h + 31) / 32), (unsigned)B), dim3(256), 0, st, Lt, Lf, out, w, h, fs);
    return hipGetLastError();
}
// n_steps <= 6 FED steps tau[0..n) in one register-marching pass (any level of at least 3 x 3 pixels); n_waves_hint = wavefronts the
// launch should at least have (rows per band are chosen for it: fewer rows per band = more wavefronts, more halo rows per output row)
hipError_t ak_fed_march(hipStream_t st, const float* Lt, const float* Lf, float* out, int w, int h, int B, const float* tau, int n_steps, int rows_per_band)
{
    if (n_steps < 1 || n_steps > 6 || w < 3 || h < 3 || rows_per_band < 1) return hipErrorInvalidValue;
    AkFedSteps fs{};
    for (int k = 0; k < n_steps; ++k) fs.tau[k] = tau[k];
    fs.n = n_steps;
    const int vw = 64 - 2 * n_steps, strips = (w + vw - 1) / vw;
    const dim3 grid((unsigned)((strips + 3) / 4), (unsigned)((h + rows_per_band - 1) / rows_per_band), (unsigned)B);
    switch (n_steps) {
        case 1: hipLaunchKernelGGL(ak_fed_march_kernel<1>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
        case 2: hipLaunchKernelGGL(ak_fed_march_kernel<2>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
        case 3: hipLaunchKernelGGL(ak_fed_march_kernel<3>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
        case 4: hipLaunchKernelGGL(ak_fed_march_kernel<4>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
        case 5: hipLaunchKernelGGL(ak_fed_march_kernel<5>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
        default: hipLaunchKernelGGL(ak_fed_march_kernel<6>, grid, dim3(256), 0, st, Lt, Lf, out, w, h, fs, rows_per_band); break;
    }
    return hipGetLastError();
}
// start image -> Lx, Ly, Ldet, conductivity of a level (5-tap Gaussian, derivative scale s in 2 .. 4, at least 16 x 16 pixels)
hipError_t ak_level_head(hipStream_t st, const float* src, float* Lx, float* Ly, float* Ldet, float* flow, int w, int h, int B, const AkTaps& kf, int s,
                         const float* inv_k2, int rows_per_band)
{
    if (kf.n != 5 || s < 2 || s > 4 || w < 16 || h < 16 || rows_per_band < 1) return hipErrorInvalidValue;
    const int vw = 64 - 2 * (2 + 2 * s), strips = (w + vw - 1) / vw;
    const dim3 grid((unsigned)((strips + 3) / 4), (unsigned)((h + rows_per_band - 1) / rows_per_band), (unsigned)B);
    if (s == 2) hipLaunchKernelGGL(ak_level_head_kernel<2>, grid, dim3(256), 0, st, src, Lx, Ly, Ldet, flow, w, h, kf, inv_k2, rows_per_band);
    else if (s == 3) hipLaunchKernelGGL(ak_level_head_kernel<3>, grid, dim3(256), 0, st, src, Lx, Ly, Ldet, flow, w, h, kf, inv_k2, rows_per_band);
    else hipLaunchKernelGGL(ak_level_head_kernel<4>, grid, dim3(256), 0, st, src, Lx, Ly, Ldet, flow, w, h, kf, inv_k2, rows_per_band);
    return hipGetLastError();
}
hipError_t ak_halfsample(hipStream_t st, const float* src, float* dst, int w, int h, int B, const AkAreaTab* xt, const int* xb,
                         const AkAreaTab* yt, const int* yb)
{
    const int dw = w / 2, dh = h / 2;
    if (dw * 2 == w && dh * 2 == h) hipLaunchKernelGGL(ak_half_fast_kernel, ak_grid(dw, dh, B), dim3(256), 0, st, src, dst, w, h, dw, dh);
    else hipLaunchKernelGGL(ak_half_area_kernel, ak_grid(dw, dh, B), dim3(256), 0, st, src, dst, w, h, dw, dh, xt, xb, yt, yb);
    return hipGetLastError();
}
// levels: [B][n_levels]
hipError_t ak_extrema(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, int max_rows, float thr, int pass)
{
    if (max_rows <= 0 || n_levels <= 0) return hipSuccess;
    const dim3 grid((unsigned)((max_rows + 3) / 4), (unsigned)n_levels, (unsigned)B);
    if (pass == 0) return hipErrorInvalidValue;          // the count pass is ak_extrema_mask (it needs the tile table)
    hipLaunchKernelGGL(ak_extrema_emit_kernel, grid, dim3(256), 0, st, levels);
    return hipGetLastError();
}
hipError_t ak_extrema_mask(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, const AkTileTable& tt, uint32_t n_tiles, float thr)
{
    if (n_tiles == 0 || n_levels <= 0) return hipSuccess;
    if (n_levels > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ak_extrema_mask_kernel, dim3(n_tiles, (unsigned)B), dim3(256), 0, st, levels, n_levels, tt, thr);
    return hipGetLastError();
}
hipError_t ak_scan_rows(hipStream_t st, const AkLevelDev* levels, int n_levels, int B)
{
    hipLaunchKernelGGL(ak_scan_rows_kernel, dim3((unsigned)(n_levels * B)), dim3(1024), 0, st, levels);
    return hipGetLastError();
}
hipError_t ak_layout(hipStream_t st, AkLevelDev* levels, int n_levels, int B, unsigned char* slots, uint32_t cap, AkBatchMeta* meta)
{
    hipLaunchKernelGGL(ak_layout_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, levels, n_levels, slots, cap, (uint32_t)B, meta);
    return hipGetLastError();
}
hipError_t ak_prune_levels(hipStream_t st, const AkLevelDev* levels, int n_levels, int B)
{
    // R3DM_AK_LIVE_CAP (test hook): a small LDS capacity forces the global-scratch fallback of the in-level pruning
    static const uint32_t live_cap = [] { const int c = r3dm_dev_knob("R3DM_AK_LIVE_CAP", kAkLive); return (uint32_t)(c < 1 ? 1 : c > kAkLive ? kAkLive : c); }();
    // R3DM_AK_PRUNE=0 (developer build): every level by one wavefront, the form of rounds 2-5
    static const bool parallel = r3dm_dev_knob("R3DM_AK_PRUNE", 1) != 0;
    const unsigned nl = (unsigned)(n_levels * B);
    if (parallel) {
        hipLaunchKernelGGL(ak_prune_link_kernel, dim3(32, nl), dim3(256), 0, st, levels);
        hipLaunchKernelGGL(ak_prune_flatten_kernel, dim3(16, nl), dim3(256), 0, st, levels);
        hipLaunchKernelGGL(ak_prune_small_kernel, dim3(32, nl), dim3(256), 0, st, levels);
        hipLaunchKernelGGL(ak_prune_large_kernel, dim3(8, nl), dim3(64), 0, st, levels, live_cap);
        hipLaunchKernelGGL(ak_prune_emit_kernel, dim3(nl), dim3(1024), 0, st, levels);
    }
    hipLaunchKernelGGL(ak_prune_level_kernel, dim3(nl), dim3(64), 0, st, levels, live_cap, parallel ? 1 : 0);
    return hipGetLastError();
}
hipError_t ak_list_ranges(hipStream_t st, const AkLevelDev* levels, int n_levels, int B)
{
    hipLaunchKernelGGL(ak_list_ranges_kernel, dim3(64, (unsigned)(n_levels * B)), dim3(256), 0, st, levels);
    return hipGetLastError();
}
hipError_t ak_cross(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, int mode)
{
    // 128 x 256 victims per pass of the grid-stride loop (ak_list_ranges must have run on the pruned lists)
    hipLaunchKernelGGL(ak_cross_kernel, dim3(128, (unsigned)(n_levels * B), 4), dim3(256), 0, st, levels, n_levels, mode);
    return hipGetLastError();
}
hipError_t ak_refine(hipStream_t st, const AkLevelDev* levels, int n_levels, int B)
{
    if (n_levels > 16) return hipErrorInvalidValue;
    // gridDim.x wavefronts share the flat entry range of an image
    const unsigned per_image = (unsigned)std::max(256, 8192 / B);
    hipLaunchKernelGGL(ak_refine_kernel, dim3(per_image, (unsigned)B), dim3(64), 0, st, levels, n_levels);
    return hipGetLastError();
}
hipError_t ak_compact(hipStream_t st, const AkLevelDev* levels, int n_levels, int B, AkKpRec* recs, uint32_t cap, AkBatchMeta* meta)
{
    if (n_levels <= 0) return hipSuccess;
    hipLaunchKernelGGL(ak_compact_kernel, dim3((unsigned)n_levels, (unsigned)B), dim3(1024), 0, st, levels, n_levels, recs, cap, meta);
    return hipGetLastError();
}

}  // namespace r3dm
