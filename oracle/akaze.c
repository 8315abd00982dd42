/*
 * akaze.c -- CPU restatement of the Fast-A-KAZE keypoint DETECTOR as Regard3D runs it.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).
 *
 * Follows (reference = /root/reference):
 *   - src/Regard3DFeatures.cpp:596-617           detectKeypoints "Fast-AKAZE": cv::AKAZE2::create() defaults, setThreshold,
 *                                                detect(); angle rad -> deg, + 90, wrapped into [0, 360]
 *   - src/thirdparty/fast-akaze/akaze.cpp:171-221          AKAZE2::detectAndCompute (options, scale space, detection)
 *   - src/thirdparty/fast-akaze/AKAZEConfig.h:18-43        defaults: omax 4, nsublevels 4, soffset 1.6, derivative_factor 1.5,
 *                                                          PM_G2, kcontrast percentile 0.7 / 300 bins
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:73-166   Allocate_Memory_Evolution (levels, sigma_size, border, FED steps)
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:199-369  Compute_Base_Evolution_Level, Create_Nonlinear_Scale_Space
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:389-410  Compute_Determinant_Hessian_Response
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:485-527,623-723  find_neighbor_point(_inv), Find_Scale_Space_Extrema
 *                                                          (the AKAZE_USE_CPP11_THREADING variant: AKAZEFeatures.h:16 defines it)
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:741-796  Do_Subpixel_Refinement
 *   - src/thirdparty/fast-akaze/AKAZEFeatures.cpp:1126-1297 Sample_Derivative_Response_Radius6, quantized_counting_sort,
 *                                                          Compute_Main_Orientation
 *   - src/thirdparty/fast-akaze/nldiffusion_functions.cpp  gaussian_2D_convolutionV2, image_derivatives_scharrV2, pm_g2V2,
 *                                                          compute_k_percentileV2, compute_scharr_derivative_kernelsV2,
 *                                                          nld_step_scalarV2, halfsample_imageV2
 *   - src/thirdparty/fast-akaze/fed.cpp                    fed_tau_by_process_timeV2 (reordered FED step sizes)
 *   - src/thirdparty/fast-akaze/utils.h                    fRoundV2, getAngleV2
 *
 * >>> PARITY UNPINNED.  The reference code above calls OpenCV (GaussianBlur, Scharr, sepFilter2D, resize INTER_AREA,
 * solve, hal::fastAtan2), an external dependency absent from /root/reference and from this image, so neither the reference
 * detector nor OpenCV can be built here and the reference ships no fixtures.  Those primitives are restated from
 * OpenCV 4's scalar code paths (imgproc/filter.cpp SymmRowSmallFilter / RowFilter / SymmColumnSmallFilter / SymmColumnFilter,
 * smooth.cpp getGaussianKernel, resize.cpp resizeAreaFast_ / ResizeArea_, core/lapack.cpp solve 2x2,
 * core/mathfuncs_core fastAtan32f); OpenCV's SIMD paths may fuse multiply-adds, so even a real build differs in the
 * last bit between CPUs.  Pinned by property tests only (tests/test_oracle_akaze.py). <<<
 *
 * Restatement decisions: the four corner samples of the FED step image are 0 (the reference never writes them and reads
 * whatever its workspace held); Lstep otherwise follows nld_step_scalar_one_lane exactly.
 */
#include "r3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AK_PI 3.1415926535897932384626433832795

typedef struct {
    int w, h, octave, sublevel, sigma_size, border;
    float esigma, etime, ratio;
    float *Lt, *Lsmooth, *Lx, *Ly, *Lxx, *Lxy, *Lyy, *Ldet;
} ak_level;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int refl101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
static int fround(float f) { return (int)(f + 0.5f); }

/* smooth.cpp getGaussianKernel(n, sigma, CV_32F), sigma > 0 */
static void gaussian_kernel(int n, double sigma, float* cf)
{
    const double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        const double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) cf[i] = (float)(cf[i] * sum);
}

/* gaussian_2D_convolutionV2(src, dst, 0, 0, sigma): ksize from sigma, GaussianBlur with BORDER_REPLICATE */
int orc_akaze_gauss_ksize(float sigma)
{
    int k = (int)ceil(2.0f * (1.0f + (sigma - 0.8f) / (0.3f)));
    if ((k % 2) == 0) k += 1;
    return k;
}
void orc_akaze_gaussian(const float* src, int w, int h, float sigma, float* dst)
{
    const int n = orc_akaze_gauss_ksize(sigma), r = n / 2;
    float kf[64];
    gaussian_kernel(n, sigma, kf);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const float* S = src + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float s;
            if (n == 5) {           /* SymmRowSmallFilter, ksize 5, symmetrical */
                s = S[x] * kf[2] + (S[clampi(x - 1, 0, w - 1)] + S[clampi(x + 1, 0, w - 1)]) * kf[3]
                    + (S[clampi(x - 2, 0, w - 1)] + S[clampi(x + 2, 0, w - 1)]) * kf[4];
            } else {                /* RowFilter: taps left to right */
                s = kf[0] * S[clampi(x - r, 0, w - 1)];
                for (int k = 1; k < n; ++k) s += kf[k] * S[clampi(x + k - r, 0, w - 1)];
            }
            tmp[(size_t)y * w + x] = s;
        }
    }
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {   /* SymmColumnFilter, symmetrical */
            float s = kf[r] * tmp[(size_t)y * w + x];
            for (int k = 1; k <= r; ++k)
                s += kf[r + k] * (tmp[(size_t)clampi(y + k, 0, h - 1) * w + x] + tmp[(size_t)clampi(y - k, 0, h - 1) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
    free(tmp);
}

/* image_derivatives_scharrV2: Scharr 3x3 (kernels [-1 0 1] x [3 10 3], not normalised), BORDER_REFLECT_101 */
void orc_akaze_scharr(const float* src, int w, int h, float* Lx, float* Ly)
{
    float* rd = (float*)malloc(sizeof(float) * (size_t)w * h);    /* row pass, derivative */
    float* rs = (float*)malloc(sizeof(float) * (size_t)w * h);    /* row pass, smoothing  */
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const float* S = src + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            const float a = S[refl101(x - 1, w)], b = S[refl101(x + 1, w)];
            rd[(size_t)y * w + x] = b - a;                        /* SymmRowSmallFilter asymmetrical, k1 == 1 */
            rs[(size_t)y * w + x] = S[x] * 10.0f + (a + b) * 3.0f;  /* symmetrical, general form */
        }
    }
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const int yu = refl101(y - 1, h), yd = refl101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            /* SymmColumnSmallFilter: symmetrical (S0 + S2)*f1 + S1*f0 ; asymmetrical with f1 == 1: S2 - S0 */
            Lx[(size_t)y * w + x] = (rd[(size_t)yu * w + x] + rd[(size_t)yd * w + x]) * 3.0f + rd[(size_t)y * w + x] * 10.0f;
            Ly[(size_t)y * w + x] = rs[(size_t)yd * w + x] - rs[(size_t)yu * w + x];
        }
    }
    free(rd); free(rs);
}

/* sepFilter2D with the kernels of compute_scharr_derivative_kernelsV2(scale s >= 2): taps at -s, 0, +s only.
 * dx != 0: x-derivative ([-1 .. 0 .. 1] along x, [norm .. w*norm .. norm] along y); else y-derivative. */
void orc_akaze_scaled_deriv(const float* src, int w, int h, int s, int dx, float* dst)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const int ksize = 3 + 2 * (s - 1);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const float* S = src + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            const float a = S[refl101(x - s, w)], b = S[refl101(x + s, w)];
            float v;
            if (dx) v = (-a) + b;                                  /* -1*S[-s] ... + 1*S[+s] (zeros in between add nothing) */
            else if (ksize == 5) v = S[x] * kc + (a + b) * norm;   /* SymmRowSmallFilter ksize 5: S0*k0 + (..)*0 + (a+b)*k2 */
            else v = (norm * a + kc * S[x]) + norm * b;            /* RowFilter, left to right */
            tmp[(size_t)y * w + x] = v;
        }
    }
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const int yu = refl101(y - s, h), yd = refl101(y + s, h);
        for (int x = 0; x < w; ++x) {
            const float u = tmp[(size_t)yu * w + x], d = tmp[(size_t)yd * w + x], c = tmp[(size_t)y * w + x];
            dst[(size_t)y * w + x] = dx ? (kc * c + norm * (d + u))   /* SymmColumnFilter symmetrical  */
                                        : (d - u);                    /* SymmColumnFilter asymmetrical */
        }
    }
    free(tmp);
}

/* pm_g2V2 */
static void pm_g2(const float* Lx, const float* Ly, size_t n, float k, float* dst)
{
    const float inv_k2 = 1.0f / (k * k);
    for (size_t i = 0; i < n; ++i) dst[i] = 1.0f / (1.0f + ((Lx[i] * Lx[i] + Ly[i] * Ly[i]) * inv_k2));
}

/* compute_k_percentileV2 */
float orc_akaze_kcontrast(const float* Lx, const float* Ly, int w, int h, float perc, int nbins)
{
    const size_t total = (size_t)(w - 2) * (h - 2);
    float* modg = (float*)malloc(sizeof(float) * total);
    size_t q = 0;
    for (int i = 1; i < h - 1; ++i)
        for (int j = 1; j < w - 1; ++j) {
            const float lx = Lx[(size_t)i * w + j], ly = Ly[(size_t)i * w + j];
            modg[q++] = sqrtf(lx * lx + ly * ly);
        }
    float hmax = 0.0f;
    for (size_t i = 0; i < total; ++i) if (hmax < modg[i]) hmax = modg[i];
    if (hmax == 0.0f) { free(modg); return 0.03f; }
    const float sc = (nbins - 1) / hmax;
    int* hist = (int*)calloc((size_t)nbins, sizeof(int));
    for (size_t i = 0; i < total; ++i) hist[(int)(modg[i] * sc)]++;
    const int nthreshold = (int)((total - hist[0]) * perc);
    int nelements = 0;
    float ret = 0.03f;
    for (int k = 1; k < nbins; ++k) {
        if (nelements >= nthreshold) { ret = (float)hmax * k / nbins; break; }
        nelements = nelements + hist[k];
    }
    free(hist); free(modg);
    return ret;
}

/* nld_step_scalarV2 + the update lt += lstep * 0.5 * step */
static void nld_step(const float* Lt, const float* Lf, int w, int h, float* Ls)
{
#pragma omp parallel for
    for (int y = 0; y < h; ++y) {
        const float* tc = Lt + (size_t)y * w; const float* fc = Lf + (size_t)y * w;
        const float* ta = y > 0 ? tc - w : NULL; const float* fa = y > 0 ? fc - w : NULL;
        const float* tb = y < h - 1 ? tc + w : NULL; const float* fb = y < h - 1 ? fc + w : NULL;
        float* d = Ls + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            const int has_l = x > 0, has_r = x < w - 1;
            float v;
            if (y == 0) {
                if (!has_l || !has_r) { d[x] = 0.0f; continue; }
                v = (fc[x] + fc[x + 1]) * (tc[x + 1] - tc[x]) + (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) + (fc[x] + fb[x]) * (tb[x] - tc[x]);
            } else if (y == h - 1) {
                if (!has_l || !has_r) { d[x] = 0.0f; continue; }
                v = (fc[x] + fc[x + 1]) * (tc[x + 1] - tc[x]) + (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) + (fc[x] + fa[x]) * (ta[x] - tc[x]);
            } else if (!has_l) {
                v = (fc[0] + fc[1]) * (tc[1] - tc[0]) + (fc[0] + fb[0]) * (tb[0] - tc[0]) + (fc[0] + fa[0]) * (ta[0] - tc[0]);
            } else if (!has_r) {
                v = (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) + (fc[x] + fb[x]) * (tb[x] - tc[x]) + (fc[x] + fa[x]) * (ta[x] - tc[x]);
            } else {
                v = (fc[x] + fc[x + 1]) * (tc[x + 1] - tc[x]) + (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) +
                    (fc[x] + fb[x]) * (tb[x] - tc[x]) + (fc[x] + fa[x]) * (ta[x] - tc[x]);
            }
            d[x] = v;
        }
    }
}

/* halfsample_imageV2: resize INTER_AREA to (w/2, h/2) */
void orc_akaze_halfsample(const float* src, int w, int h, float* dst)
{
    const int dw = w / 2, dh = h / 2;
    if (dw * 2 == w && dh * 2 == h) {                  /* resizeAreaFast_: ((a + b) + c) + d, times 1/4 */
#pragma omp parallel for
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x) {
                const float* S = src + (size_t)(2 * y) * w + 2 * x;
                float sum = 0; sum += S[0]; sum += S[1]; sum += S[w]; sum += S[w + 1];
                dst[(size_t)y * dw + x] = sum * 0.25f;
            }
        return;
    }
    /* ResizeArea_ with computeResizeAreaTab (fractional cells) */
    typedef struct { int si, di; float alpha; } tab_t;
    tab_t* xt = (tab_t*)malloc(sizeof(tab_t) * (size_t)(w * 2 + 4));
    tab_t* yt = (tab_t*)malloc(sizeof(tab_t) * (size_t)(h * 2 + 4));
    int nx = 0, ny = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int ssize = pass ? h : w, dsize = pass ? dh : dw;
        const double scale = (double)ssize / dsize;
        tab_t* tab = pass ? yt : xt;
        int k = 0;
        for (int dx = 0; dx < dsize; ++dx) {
            const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
            const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
            int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
            if (sx2 > ssize - 1) sx2 = ssize - 1;
            if (sx1 > sx2) sx1 = sx2;
            if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
            for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
            if (fsx2 - sx2 > 1e-3) {
                double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cell) a = cell;
                tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(a / cell);
            }
        }
        if (pass) ny = k; else nx = k;
    }
    float* buf = (float*)malloc(sizeof(float) * (size_t)dw);
    float* sum = (float*)calloc((size_t)dw, sizeof(float));
    int prev_dy = yt[0].di;
    for (int j = 0; j < ny; ++j) {
        const float beta = yt[j].alpha;
        const int dy = yt[j].di, sy = yt[j].si;
        const float* S = src + (size_t)sy * w;
        for (int x = 0; x < dw; ++x) buf[x] = 0;
        for (int k = 0; k < nx; ++k) buf[xt[k].di] += S[xt[k].si] * xt[k].alpha;
        if (dy != prev_dy) {
            for (int x = 0; x < dw; ++x) { dst[(size_t)prev_dy * dw + x] = sum[x]; sum[x] = beta * buf[x]; }
            prev_dy = dy;
        } else {
            for (int x = 0; x < dw; ++x) sum[x] += beta * buf[x];
        }
    }
    for (int x = 0; x < dw; ++x) dst[(size_t)prev_dy * dw + x] = sum[x];
    free(buf); free(sum); free(xt); free(yt);
}

/* fed_tau_by_process_timeV2(T, 1, 0.25, reordering = true) */
static int fed_is_prime(int number)
{
    if (number <= 1) return 0;
    if (number == 1 || number == 2 || number == 3 || number == 5 || number == 7) return 1;
    if ((number % 2) == 0 || (number % 3) == 0 || (number % 5) == 0 || (number % 7) == 0) return 0;
    int is_prime = 1;
    const int upper = (int)sqrt(1.0f + number);
    for (int divisor = 11; divisor <= upper; divisor += 2) if (number % divisor == 0) is_prime = 0;
    return is_prime;
}
int orc_akaze_fed_tau(float T, float* tau /* cap 256 */)
{
    const float tau_max = 0.25f;
    const int n = (int)(ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
    if (n <= 0) return 0;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    float tauh[256];
    const float c = 1.0f / (4.0f * n + 2.0f);
    const float d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) { const float hh = cosf((float)AK_PI * (2.0f * k + 1.0f) * c); tauh[k] = d / (hh * hh); }
    if (n == 1) { tau[0] = tauh[0]; return 1; }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}

/* ---------------------------------------------------------------- detector */

typedef struct { float x, y, size, response; int level, alive; } ak_kp;
typedef struct { ak_kp* v; int n, cap; } ak_list;
static void kl_push(ak_list* l, ak_kp k)
{
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1024; l->v = (ak_kp*)realloc(l->v, sizeof(ak_kp) * (size_t)l->cap); }
    l->v[l->n++] = k;
}

/* mathfuncs_core fastAtan32f, radians */
static float fast_atan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / AK_PI), p3 = -0.3258083974640975f * (float)(180 / AK_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / AK_PI), p7 = -0.04432655554792128f * (float)(180 / AK_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a * (float)(AK_PI / 180);
}

static const float gauss25[7][7] = {
    { 0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f },
    { 0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f },
    { 0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f },
    { 0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f },
    { 0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f },
    { 0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f },
    { 0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f } };

/* Compute_Main_Orientation up to the dominant vector (maxX, maxY); the angle is getAngleV2(maxX, maxY) */
void orc_akaze_orientation_vec(const float* Lx, const float* Ly, int cols, int x0, int y0, int scale, float* out_xy)
{
    static const int id[13] = { 6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6 };
    float resX[109], resY[109], Ang[109];
    int k = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                const float wgt = gauss25[id[i + 6]][id[j + 6]];
                const size_t p = (size_t)(y0 + i * scale) * cols + (x0 + j * scale);
                resX[k] = wgt * Lx[p]; resY[k] = wgt * Ly[p];
                ++k;
            }
    for (int i = 0; i < 109; ++i) Ang[i] = fast_atan2(resY[i], resX[i]);
    enum { slices = 42, win = 7 };
    const float ang_step = (float)(2.0 * AK_PI / slices);
    unsigned char slice[slices + 1], sorted_idx[109];
    {   /* quantized_counting_sort(Ang, 109, ang_step, 2 pi, sorted_idx, slice) */
        const int nkeys = (int)((float)(2.0 * AK_PI) / ang_step);
        memset(slice, 0, (size_t)nkeys + 1);
        for (int i = 0; i < 109; ++i) slice[(int)(Ang[i] / ang_step)]++;
        for (int i = 1; i <= nkeys; ++i) slice[i] += slice[i - 1];
        for (int i = 0; i < 109; ++i) sorted_idx[--slice[(int)(Ang[i] / ang_step)]] = (unsigned char)i;
    }
    float maxX = 0.0f, maxY = 0.0f;
    for (int i = slice[0]; i < slice[win]; ++i) { maxX += resX[sorted_idx[i]]; maxY += resY[sorted_idx[i]]; }
    float maxNorm = maxX * maxX + maxY * maxY;
    for (int sn = 1; sn <= slices - win; ++sn) {
        if (slice[sn] == slice[sn - 1] && slice[sn + win] == slice[sn + win - 1]) continue;
        float sumX = 0.0f, sumY = 0.0f;
        for (int i = slice[sn]; i < slice[sn + win]; ++i) { sumX += resX[sorted_idx[i]]; sumY += resY[sorted_idx[i]]; }
        const float nrm = sumX * sumX + sumY * sumY;
        if (nrm > maxNorm) { maxNorm = nrm; maxX = sumX; maxY = sumY; }
    }
    for (int sn = slices - win + 1; sn < slices; ++sn) {
        const int remain = sn + win - slices;
        if (slice[sn] == slice[sn - 1] && slice[remain] == slice[remain - 1]) continue;
        float sumX = 0.0f, sumY = 0.0f;
        for (int i = slice[sn]; i < slice[slices]; ++i) { sumX += resX[sorted_idx[i]]; sumY += resY[sorted_idx[i]]; }
        for (int i = slice[0]; i < slice[remain]; ++i) { sumX += resX[sorted_idx[i]]; sumY += resY[sorted_idx[i]]; }
        const float nrm = sumX * sumX + sumY * sumY;
        if (nrm > maxNorm) { maxNorm = nrm; maxX = sumX; maxY = sumY; }
    }
    out_xy[0] = maxX; out_xy[1] = maxY;
}

/* MLDB_Full_Descriptor_InvokerV2::Get_MLDB_Full_Descriptor (AKAZEFeatures.cpp:1877-1909) with MLDB_Fill_Values (:1790-1844) and
 * MLDB_Binary_Comparisons (:1846-1868): 3 grids (2x2, 3x3, 4x4 cells over a 20 sigma window rotated to the keypoint angle), 3
 * channels (mean intensity, mean rotated dx, dy), all pairwise comparisons per channel -> 486 bits, LSB first, 61 bytes. */
void orc_akaze_mldb(const float* Lt, const float* Lx, const float* Ly, int cols, float xf, float yf, float co, float si,
                    float scale, unsigned char* desc /* 61 */)
{
    const int pattern_size = 10, chan = 3;
    const double size_mult[3] = { 1, 2.0 / 3.0, 1.0 / 2.0 };
    float values[16 * 3];
    int dpos = 0;
    memset(desc, 0, 61);
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int val_count = (lvl + 2) * (lvl + 2);
        const int sample_step = (int)ceil(pattern_size * size_mult[lvl]);
        int valpos = 0;
        for (int i = -pattern_size; i < pattern_size; i += sample_step)
            for (int j = -pattern_size; j < pattern_size; j += sample_step) {
                float di = 0.0f, dx = 0.0f, dy = 0.0f;
                int nsamples = 0;
                for (int k = i; k < i + sample_step; ++k)
                    for (int l = j; l < j + sample_step; ++l) {
                        const float sample_y = yf + (l * co * scale + k * si * scale);
                        const float sample_x = xf + (-l * si * scale + k * co * scale);
                        const int y1 = fround(sample_y), x1 = fround(sample_x);
                        const float ri = Lt[(size_t)y1 * cols + x1];
                        di += ri;
                        const float rx = Lx[(size_t)y1 * cols + x1], ry = Ly[(size_t)y1 * cols + x1];
                        const float rry = rx * co + ry * si;
                        const float rrx = -rx * si + ry * co;
                        dx += rrx; dy += rry;
                        nsamples++;
                    }
                di /= nsamples; dx /= nsamples; dy /= nsamples;
                values[valpos] = di; values[valpos + 1] = dx; values[valpos + 2] = dy;
                valpos += chan;
            }
        int32_t iv[16 * 3];
        memcpy(iv, values, sizeof(float) * (size_t)(val_count * chan));
        for (int q = 0; q < val_count * chan; ++q) iv[q] = iv[q] ^ (iv[q] < 0 ? 0x7fffffff : 0);      /* CV_TOGGLE_FLT */
        for (int pos = 0; pos < chan; ++pos)
            for (int i = 0; i < val_count; ++i) {
                const int32_t ival = iv[chan * i + pos];
                for (int j = i + 1; j < val_count; ++j) {
                    if (ival > iv[chan * j + pos]) desc[dpos >> 3] |= (unsigned char)(1 << (dpos & 7));
                    else desc[dpos >> 3] &= (unsigned char)~(1 << (dpos & 7));
                    dpos++;
                }
            }
    }
    const int remain = dpos % 8;
    if (remain > 0) desc[dpos >> 3] &= (unsigned char)(0xff >> (8 - remain));
}

/* Regard3DFeatures::detectKeypoints, "Fast-AKAZE" arm: image = h x w floats in [0, 1].
 * keypoints out: (x, y, size, angle_degrees) x cap; responses/levels optional.  Returns the number detected (may exceed cap:
 * only the first cap are written).  dbg_level >= 0: copies that level's Ldet (and Lt) into dbg_ldet / dbg_lt if non-NULL. */
static int akaze_detect_impl(const float* image, int w, int h, float dthreshold, float* kps, int cap, float* responses, int* levels,
                             int dbg_level, float* dbg_ldet, float* dbg_lt, float* dbg_info, unsigned char* mldb /* cap x 61 or NULL */)
{
    const int omax = 4, nsub = 4;
    const float soffset = 1.6f, dfac = 1.5f;
    const float smax = 10.0f * sqrtf(2.0f);
    ak_level lv[16];
    int nl = 0;
    {   /* Allocate_Memory_Evolution */
        int lh = h, lw = w, power = 1, stop = 0;
        for (int i = 0; i < omax && !stop; ++i) {
            for (int j = 0; j < nsub; ++j) {
                ak_level e; memset(&e, 0, sizeof(e));
                e.w = lw; e.h = lh;
                e.esigma = soffset * powf(2.f, (float)j / nsub + i);
                e.sigma_size = fround(e.esigma * dfac / power);
                e.border = fround(smax * e.sigma_size) + 1;
                e.etime = 0.5f * (e.esigma * e.esigma);
                e.octave = i; e.sublevel = j; e.ratio = (float)power;
                if (e.border * 2 + 1 >= lw || e.border * 2 + 1 >= lh) { stop = 1; break; }
                lv[nl++] = e;
            }
            if (stop) break;
            power <<= 1; lh >>= 1; lw >>= 1;
            if (lw < 80 || lh < 40) break;
        }
    }
    if (nl == 0) return 0;
    for (int i = 0; i < nl; ++i) {
        const size_t n = (size_t)lv[i].w * lv[i].h;
        float* blk = (float*)malloc(sizeof(float) * n * 8);
        lv[i].Lt = blk; lv[i].Lsmooth = blk + n; lv[i].Lx = blk + 2 * n; lv[i].Ly = blk + 3 * n;
        lv[i].Lxx = blk + 4 * n; lv[i].Lxy = blk + 5 * n; lv[i].Lyy = blk + 6 * n; lv[i].Ldet = blk + 7 * n;
    }
    const size_t n0 = (size_t)w * h;
    float* wx = (float*)malloc(sizeof(float) * n0); float* wy = (float*)malloc(sizeof(float) * n0);
    float* wflow = (float*)malloc(sizeof(float) * n0); float* wstep = (float*)malloc(sizeof(float) * n0);

#define AK_HESSIAN(e)                                                                               \
    do {                                                                                            \
        orc_akaze_scaled_deriv((e).Lsmooth, (e).w, (e).h, (e).sigma_size, 1, (e).Lx);               \
        orc_akaze_scaled_deriv((e).Lx, (e).w, (e).h, (e).sigma_size, 1, (e).Lxx);                   \
        orc_akaze_scaled_deriv((e).Lx, (e).w, (e).h, (e).sigma_size, 0, (e).Lxy);                   \
        orc_akaze_scaled_deriv((e).Lsmooth, (e).w, (e).h, (e).sigma_size, 0, (e).Ly);               \
        orc_akaze_scaled_deriv((e).Ly, (e).w, (e).h, (e).sigma_size, 0, (e).Lyy);                   \
        const size_t nn = (size_t)(e).w * (e).h;                                                    \
        for (size_t q = 0; q < nn; ++q) (e).Ldet[q] = (e).Lxx[q] * (e).Lyy[q] - (e).Lxy[q] * (e).Lxy[q]; \
    } while (0)

    /* Create_Nonlinear_Scale_Space */
    float kcontrast = 0.03f;
    orc_akaze_gaussian(image, w, h, soffset, lv[0].Lsmooth);
    AK_HESSIAN(lv[0]);
    if (nl > 1) {
        orc_akaze_gaussian(image, w, h, 1.0f, wflow);
        orc_akaze_scharr(wflow, w, h, wx, wy);
        kcontrast = orc_akaze_kcontrast(wx, wy, w, h, 0.7f, 300);
    }
    memcpy(lv[0].Lt, lv[0].Lsmooth, sizeof(float) * n0);
    for (int i = 1; i < nl; ++i) {
        ak_level* e = &lv[i];
        const size_t n = (size_t)e->w * e->h;
        if (e->octave > lv[i - 1].octave) { orc_akaze_halfsample(lv[i - 1].Lt, lv[i - 1].w, lv[i - 1].h, e->Lt); kcontrast = kcontrast * 0.75f; }
        else memcpy(e->Lt, lv[i - 1].Lt, sizeof(float) * n);
        orc_akaze_gaussian(e->Lt, e->w, e->h, 1.0f, e->Lsmooth);
        orc_akaze_scharr(e->Lsmooth, e->w, e->h, wx, wy);
        AK_HESSIAN(*e);
        pm_g2(wx, wy, n, kcontrast, wflow);
        float tau[256];
        const int nt = orc_akaze_fed_tau(e->etime - lv[i - 1].etime, tau);
        for (int j = 0; j < nt; ++j) {
            nld_step(e->Lt, wflow, e->w, e->h, wstep);
            const float step_size = tau[j];
            for (size_t q = 0; q < n; ++q) e->Lt[q] += wstep[q] * 0.5f * step_size;
        }
    }
    if (dbg_info) { dbg_info[0] = (float)nl; dbg_info[1] = kcontrast; }
    if (dbg_level >= 0 && dbg_level < nl) {
        const size_t n = (size_t)lv[dbg_level].w * lv[dbg_level].h;
        if (dbg_ldet) memcpy(dbg_ldet, lv[dbg_level].Ldet, sizeof(float) * n);
        if (dbg_lt) memcpy(dbg_lt, lv[dbg_level].Lt, sizeof(float) * n);
        if (dbg_info) { dbg_info[2] = (float)lv[dbg_level].w; dbg_info[3] = (float)lv[dbg_level].h; dbg_info[4] = (float)lv[dbg_level].border;
                        dbg_info[5] = (float)lv[dbg_level].sigma_size; dbg_info[6] = lv[dbg_level].esigma; }
    }

    /* Find_Scale_Space_Extrema (threaded variant) */
    ak_list* kl = (ak_list*)calloc((size_t)nl, sizeof(ak_list));
    for (int i = 0; i < nl; ++i) {
        const ak_level* e = &lv[i];
        const float psize = e->esigma * dfac;
        for (int y = e->border; y < e->h - e->border; ++y) {
            const float* prev = e->Ldet + (size_t)(y - 1) * e->w; const float* curr = prev + e->w; const float* next = curr + e->w;
            for (int x = e->border; x < e->w - e->border; ++x) {
                const float value = curr[x];
                if (value <= dthreshold) continue;
                if (value <= curr[x - 1] || value <= curr[x + 1]) continue;
                if (value <= prev[x - 1] || value <= prev[x] || value <= prev[x + 1]) continue;
                if (value <= next[x - 1] || value <= next[x] || value <= next[x + 1]) continue;
                ak_kp p = { (float)(x * e->ratio), (float)(y * e->ratio), psize, value, i, 1 };
                int found = -1;
                for (int q = 0; q < kl[i].n; ++q) {
                    const float dx = p.x - kl[i].v[q].x, dy = p.y - kl[i].v[q].y;
                    if (dx * dx + dy * dy <= p.size * p.size) { found = q; break; }
                }
                if (found >= 0) { if (p.response > kl[i].v[found].response) kl[i].v[found] = p; continue; }
                kl_push(&kl[i], p);
            }
        }
    }
    for (int i = 1; i < nl; ++i)                       /* lower scale level */
        for (int j = 0; j < kl[i].n; ++j) {
            const ak_kp* pt = &kl[i].v[j];
            for (int q = 0; q < kl[i - 1].n; ++q) {
                ak_kp* v = &kl[i - 1].v[q];
                if (!v->alive) continue;
                const float dx = pt->x - v->x, dy = pt->y - v->y;
                if (dx * dx + dy * dy <= pt->size * pt->size && pt->response > v->response) v->alive = 0;
            }
        }
    for (int i = nl - 2; i >= 0; --i)                  /* upper scale level */
        for (int j = 0; j < kl[i].n; ++j) {
            const ak_kp* pt = &kl[i].v[j];
            if (!pt->alive) continue;
            for (int q = 0; q < kl[i + 1].n; ++q) {
                ak_kp* v = &kl[i + 1].v[q];
                if (!v->alive) continue;
                const float dx = pt->x - v->x, dy = pt->y - v->y;
                if (dx * dx + dy * dy <= v->size * v->size && pt->response > v->response) v->alive = 0;
            }
        }

    /* Do_Subpixel_Refinement + Compute_Main_Orientation + the Regard3D angle convention */
    int n_out = 0;
    for (int i = 0; i < nl; ++i) {
        const ak_level* e = &lv[i];
        const float* ldet = e->Ldet;
        const int cols = e->w;
        const float ratio = e->ratio;
        for (int j = 0; j < kl[i].n; ++j) {
            ak_kp kp = kl[i].v[j];
            if (!kp.alive) continue;
            const int x = (int)(kp.x / ratio), y = (int)(kp.y / ratio);
            const float Dx = 0.5f * (ldet[y * cols + x + 1] - ldet[y * cols + x - 1]);
            const float Dy = 0.5f * (ldet[(y + 1) * cols + x] - ldet[(y - 1) * cols + x]);
            const float Dxx = ldet[y * cols + x + 1] + ldet[y * cols + x - 1] - 2.0f * ldet[y * cols + x];
            const float Dyy = ldet[(y + 1) * cols + x] + ldet[(y - 1) * cols + x] - 2.0f * ldet[y * cols + x];
            const float Dxy = 0.25f * (ldet[(y + 1) * cols + x + 1] + ldet[(y - 1) * cols + x - 1] -
                                       ldet[(y - 1) * cols + x + 1] - ldet[(y + 1) * cols + x - 1]);
            /* cv::solve, 2x2, DECOMP_LU: Cramer's rule in double (core/lapack.cpp) */
            float dx = 0.0f, dy = 0.0f;
            {
                const float b0 = -Dx, b1 = -Dy;
                double d = (double)Dxx * Dyy - (double)Dxy * Dxy;
                if (d != 0.) {
                    d = 1. / d;
                    const double t = (float)(((double)b0 * Dyy - (double)b1 * Dxy) * d);
                    dy = (float)(((double)b1 * Dxx - (double)b0 * Dxy) * d);
                    dx = (float)t;
                }
            }
            if (fabsf(dx) > 1.0f || fabsf(dy) > 1.0f) continue;
            kp.x += dx * ratio; kp.y += dy * ratio;
            kp.size *= 2.0f;
            /* Compute_Main_Orientation */
            const int scale = fround(0.5f * kp.size / ratio);
            const int x0 = fround(kp.x / ratio), y0 = fround(kp.y / ratio);
            float mv[2];
            orc_akaze_orientation_vec(e->Lx, e->Ly, cols, x0, y0, scale, mv);
            float theta = atan2f(mv[1], mv[0]);
            if (!(theta >= 0)) theta = theta + (float)(2.0f * AK_PI);
            /* detectKeypoints: radians -> degrees, + 90, wrap */
            float ang = theta;
            ang *= 180.0 / AK_PI;
            ang += 90.0f;
            while (ang < 0) ang += 360.0f;
            while (ang > 360.0f) ang -= 360.0f;
            if (n_out < cap && mldb)      /* AKAZE2::detectAndCompute: descriptors from the raw (radian) orientation */
                orc_akaze_mldb(e->Lt, e->Lx, e->Ly, cols, kp.x / ratio, kp.y / ratio, cosf(theta), sinf(theta), (float)e->sigma_size, mldb + 61 * (size_t)n_out);
            if (n_out < cap) {
                kps[4 * n_out] = kp.x; kps[4 * n_out + 1] = kp.y; kps[4 * n_out + 2] = kp.size; kps[4 * n_out + 3] = ang;
                if (responses) responses[n_out] = kp.response;
                if (levels) levels[n_out] = i;
            }
            ++n_out;
        }
    }
    for (int i = 0; i < nl; ++i) { free(kl[i].v); free(lv[i].Lt); }
    free(kl); free(wx); free(wy); free(wflow); free(wstep);
    return n_out;
}

int orc_akaze_detect(const float* image, int w, int h, float dthreshold, float* kps, int cap, float* responses, int* levels,
                     int dbg_level, float* dbg_ldet, float* dbg_lt, float* dbg_info /* [8]: n_levels, kcontrast, ... */)
{
    return akaze_detect_impl(image, w, h, dthreshold, kps, cap, responses, levels, dbg_level, dbg_ldet, dbg_lt, dbg_info, NULL);
}

/* cv::AKAZE2::detectAndCompute with DESCRIPTOR_MLDB (akaze.cpp:171-221): keypoints as above + 61-byte MLDB-486 descriptors */
int orc_akaze_detect_mldb(const float* image, int w, int h, float dthreshold, float* kps, unsigned char* desc, int cap, float* responses)
{
    return akaze_detect_impl(image, w, h, dthreshold, kps, cap, responses, NULL, -1, NULL, NULL, NULL, desc);
}
