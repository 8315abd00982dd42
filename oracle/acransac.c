/*
 * oracle/acransac.c -- CPU restatement of the a-contrario RANSAC fundamental-matrix filter.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).  Build with -ffp-contract=off.
 *
 * Reference call site: /root/reference/src/R3DComputeMatches.cpp:2099-2115
 *   ImageCollectionGeometricFilter(&sfm_data, regions_provider)
 *     .Robust_model_estimation(GeometricFilter_FMatrix_AC(4.0, 2048), putative, false)
 * The algorithm is OpenMVG 1.4's (external, not vendored; restated from SURVEY.md A.5):
 *   robust_estimation/robust_estimator_ACRansac.hpp  ACRANSAC(), bestNFA(), makelogcombi()
 *   robust_estimation/robust_estimator_ACRansacKernelAdaptator.hpp  ACKernelAdaptor (point-to-line)
 *   multiview/solver_fundamental_kernel.hpp  SevenPointSolver, SymmetricEpipolarDistanceError
 *   multiview/conditioning.hpp  PreconditionerFromPoints(width, height)
 *   numeric/poly.h  SolveCubicPolynomial (trigonometric / Cardano form)
 *   matching_image_collection/F_ACRobust.hpp  accept iff #inliers > 2.5 * 7
 * Restatement decisions (no golden vectors exist in the reference -> parity unpinned):
 *   sample stream = counter-based generator (orc_rng_u64), rejection of repeated indices;
 *   null space of the 7x9 system = last two columns of Q in a Householder QR of A^T.
 */
#include "r3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------- sample stream */

static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

uint64_t orc_rng_u64(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter, uint32_t attempt)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    const uint64_t a = mix64(seed + G * (1ULL + (((uint64_t)I << 32) | (uint64_t)J)));
    return mix64(a + G * (1ULL + (((uint64_t)iter << 32) | (uint64_t)attempt)));
}

/* UniformSample(n, pool): n <= 7 distinct positions in the pool, drawn by rejection. */
void orc_sample_n(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter,
                  const uint32_t* pool, uint32_t pool_size, uint32_t n, uint32_t* sample)
{
    uint32_t pos[7];
    uint32_t cnt = 0, attempt = 0;
    while (cnt < n) {
        const uint64_t r = orc_rng_u64(seed, I, J, iter, attempt++);
        const uint32_t p = (uint32_t)(((r >> 32) * (uint64_t)pool_size) >> 32);
        int dup = 0;
        for (uint32_t k = 0; k < cnt; ++k) dup |= (pos[k] == p);
        if (!dup) pos[cnt++] = p;
    }
    for (uint32_t k = 0; k < n; ++k) sample[k] = pool ? pool[pos[k]] : pos[k];
}

void orc_sample7(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter,
                 const uint32_t* pool, uint32_t pool_size, uint32_t* sample7)
{
    orc_sample_n(seed, I, J, iter, pool, pool_size, 7, sample7);
}

/* ---------------------------------------------------------------- cubic */

/* roots of c0 + c1 x + c2 x^2 + c3 x^3; returns the number of real roots written (0, 1 or 3). */
int orc_solve_cubic(const double* c, double* roots)
{
    if (c[3] == 0.0) return 0;
    const double a = c[2] / c[3], b = c[1] / c[3], cc = c[0] / c[3];
    const double q = a * a - 3.0 * b;
    const double r = 2.0 * a * a * a - 9.0 * a * b + 27.0 * cc;
    const double Q = q / 9.0, R = r / 54.0;
    const double Q3 = Q * Q * Q, R2 = R * R;
    const double CR2 = 729.0 * r * r, CQ3 = 2916.0 * q * q * q;
    if (R == 0.0 && Q == 0.0) {
        roots[0] = roots[1] = roots[2] = -a / 3.0;
        return 3;
    } else if (CR2 == CQ3) {
        const double sqrtQ = sqrt(Q);
        if (R > 0.0) { roots[0] = -2.0 * sqrtQ - a / 3.0; roots[1] = sqrtQ - a / 3.0; roots[2] = sqrtQ - a / 3.0; }
        else         { roots[0] = -sqrtQ - a / 3.0; roots[1] = -sqrtQ - a / 3.0; roots[2] = 2.0 * sqrtQ - a / 3.0; }
        return 3;
    } else if (CR2 < CQ3) {
        const double sqrtQ = sqrt(Q);
        const double sqrtQ3 = sqrtQ * sqrtQ * sqrtQ;
        const double theta = acos(R / sqrtQ3);
        const double norm = -2.0 * sqrtQ;
        double x0 = norm * cos(theta / 3.0) - a / 3.0;
        double x1 = norm * cos((theta + 2.0 * M_PI) / 3.0) - a / 3.0;
        double x2 = norm * cos((theta - 2.0 * M_PI) / 3.0) - a / 3.0;
        double t;
        if (x0 > x1) { t = x0; x0 = x1; x1 = t; }
        if (x1 > x2) { t = x1; x1 = x2; x2 = t; if (x0 > x1) { t = x0; x0 = x1; x1 = t; } }
        roots[0] = x0; roots[1] = x1; roots[2] = x2;
        return 3;
    }
    const double sgnR = (R >= 0.0 ? 1.0 : -1.0);
    const double A = -sgnR * cbrt(fabs(R) + sqrt(R2 - Q3));
    const double B = Q / A;
    roots[0] = A + B - a / 3.0;
    return 1;
}

/* ---------------------------------------------------------------- seven-point solver */

static double det3(const double* r0, const double* r1, const double* r2)
{
    return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1])
         - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0])
         + r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

/* x1, x2: 7 x 2 (normalised coordinates).  Fs: up to 3 row-major 3x3 matrices. */
int orc_seven_point(const double* x1, const double* x2, double* Fs)
{
    /* M = A^T, 9 x 7, column p = epipolar constraint of correspondence p */
    double M[9][7];
    for (int p = 0; p < 7; ++p) {
        const double ax = x1[2 * p], ay = x1[2 * p + 1], bx = x2[2 * p], by = x2[2 * p + 1];
        M[0][p] = bx * ax; M[1][p] = bx * ay; M[2][p] = bx;
        M[3][p] = by * ax; M[4][p] = by * ay; M[5][p] = by;
        M[6][p] = ax;      M[7][p] = ay;      M[8][p] = 1.0;
    }
    /* Householder QR, reflectors kept as (v_j, beta_j) */
    double V[7][9];
    double beta[7];
    for (int j = 0; j < 7; ++j) {
        double nrm2 = 0.0;
        for (int r = j; r < 9; ++r) nrm2 += M[r][j] * M[r][j];
        const double nrm = sqrt(nrm2);
        for (int r = 0; r < 9; ++r) V[j][r] = 0.0;
        beta[j] = 0.0;
        if (nrm == 0.0) continue;
        const double alpha = (M[j][j] > 0.0) ? -nrm : nrm;
        double vn2 = 0.0;
        for (int r = j; r < 9; ++r) { V[j][r] = M[r][j]; }
        V[j][j] -= alpha;
        for (int r = j; r < 9; ++r) vn2 += V[j][r] * V[j][r];
        if (vn2 == 0.0) continue;
        beta[j] = 2.0 / vn2;
        for (int c = j + 1; c < 7; ++c) {
            double dot = 0.0;
            for (int r = j; r < 9; ++r) dot += V[j][r] * M[r][c];
            const double s = beta[j] * dot;
            for (int r = j; r < 9; ++r) M[r][c] -= s * V[j][r];
        }
    }
    /* f1 = Q e7, f2 = Q e8 with Q = H0 H1 ... H6 */
    double f[2][9];
    for (int e = 0; e < 2; ++e) {
        for (int r = 0; r < 9; ++r) f[e][r] = (r == 7 + e) ? 1.0 : 0.0;
        for (int j = 6; j >= 0; --j) {
            double dot = 0.0;
            for (int r = j; r < 9; ++r) dot += V[j][r] * f[e][r];
            const double s = beta[j] * dot;
            for (int r = j; r < 9; ++r) f[e][r] -= s * V[j][r];
        }
    }
    const double* A = f[0];
    const double* B = f[1];
    double P[4];
    P[0] = det3(A, A + 3, A + 6);
    P[1] = det3(B, A + 3, A + 6) + det3(A, B + 3, A + 6) + det3(A, A + 3, B + 6);
    P[2] = det3(A, B + 3, B + 6) + det3(B, A + 3, B + 6) + det3(B, B + 3, A + 6);
    P[3] = det3(B, B + 3, B + 6);
    double roots[3];
    const int n = orc_solve_cubic(P, roots);
    for (int k = 0; k < n; ++k)
        for (int e = 0; e < 9; ++e) Fs[9 * k + e] = A[e] + roots[k] * B[e];
    return n;
}

/* SymmetricEpipolarDistanceError::Error(F, x, y)  (squared, /4 to match Sampson's scale) */
double orc_sym_epipolar_err(const double* F, double x1, double y1, double x2, double y2)
{
    const double Fx0 = F[0] * x1 + F[1] * y1 + F[2];
    const double Fx1 = F[3] * x1 + F[4] * y1 + F[5];
    const double Fx2 = F[6] * x1 + F[7] * y1 + F[8];
    const double Fty0 = F[0] * x2 + F[3] * y2 + F[6];
    const double Fty1 = F[1] * x2 + F[4] * y2 + F[7];
    const double yFx = x2 * Fx0 + y2 * Fx1 + Fx2;
    return (yFx * yFx) * (1.0 / (Fx0 * Fx0 + Fx1 * Fx1) + 1.0 / (Fty0 * Fty0 + Fty1 * Fty1)) / 4.0;
}

/* ---------------------------------------------------------------- four-point homography */

/* OpenMVG homography::kernel::FourPointSolver (DLT; SURVEY.md A.5 skeleton, f-2): per correspondence
 *   [x y 1 0 0 0 -x'x -x'y -x'] and [0 0 0 x y 1 -y'x -y'y -y'];  h = null vector of the 8x9 system
 * (restatement: last column of Q in the Householder QR of L^T).  x1, x2: 4 x 2.  H: row-major 3x3. */
int orc_four_point_h(const double* x1, const double* x2, double* H)
{
    double M[9][8];
    for (int p = 0; p < 4; ++p) {
        const double x = x1[2 * p], y = x1[2 * p + 1], u = x2[2 * p], v = x2[2 * p + 1];
        const int c0 = 2 * p, c1 = 2 * p + 1;
        M[0][c0] = x; M[1][c0] = y; M[2][c0] = 1.0; M[3][c0] = 0.0; M[4][c0] = 0.0; M[5][c0] = 0.0;
        M[6][c0] = -u * x; M[7][c0] = -u * y; M[8][c0] = -u;
        M[0][c1] = 0.0; M[1][c1] = 0.0; M[2][c1] = 0.0; M[3][c1] = x; M[4][c1] = y; M[5][c1] = 1.0;
        M[6][c1] = -v * x; M[7][c1] = -v * y; M[8][c1] = -v;
    }
    double V[8][9], beta[8];
    for (int j = 0; j < 8; ++j) {
        double nrm2 = 0.0;
        for (int r = j; r < 9; ++r) nrm2 += M[r][j] * M[r][j];
        const double nrm = sqrt(nrm2);
        for (int r = 0; r < 9; ++r) V[j][r] = 0.0;
        beta[j] = 0.0;
        if (nrm == 0.0) continue;
        const double alpha = (M[j][j] > 0.0) ? -nrm : nrm;
        double vn2 = 0.0;
        for (int r = j; r < 9; ++r) V[j][r] = M[r][j];
        V[j][j] -= alpha;
        for (int r = j; r < 9; ++r) vn2 += V[j][r] * V[j][r];
        if (vn2 == 0.0) continue;
        beta[j] = 2.0 / vn2;
        for (int c = j + 1; c < 8; ++c) {
            double dot = 0.0;
            for (int r = j; r < 9; ++r) dot += V[j][r] * M[r][c];
            const double s = beta[j] * dot;
            for (int r = j; r < 9; ++r) M[r][c] -= s * V[j][r];
        }
    }
    double f[9];
    for (int r = 0; r < 9; ++r) f[r] = (r == 8) ? 1.0 : 0.0;
    for (int j = 7; j >= 0; --j) {
        double dot = 0.0;
        for (int r = j; r < 9; ++r) dot += V[j][r] * f[r];
        const double s = beta[j] * dot;
        for (int r = j; r < 9; ++r) f[r] -= s * V[j][r];
    }
    memcpy(H, f, sizeof(f));
    return 1;
}

/* homography::kernel::AsymmetricError: || x2 - hnormalized(H x1) ||^2 */
double orc_h_asym_err(const double* H, double x1, double y1, double x2, double y2)
{
    const double w = H[6] * x1 + H[7] * y1 + H[8];
    const double ex = x2 - (H[0] * x1 + H[1] * y1 + H[2]) / w;
    const double ey = y2 - (H[3] * x1 + H[4] * y1 + H[5]) / w;
    return ex * ex + ey * ey;
}

/* ---------------------------------------------------------------- log-combinatorial tables */

void orc_logcombi_tables(uint32_t n, uint32_t ks, float* logc_n, float* logc_k)
{
    float* l10 = (float*)malloc(sizeof(float) * ((size_t)n + 2));
    for (uint32_t k = 0; k <= n; ++k) l10[k] = log10f((float)k);
    /* logcombi(k, n) = sum_{i=1..min(k,n-k)} (l10[n-i+1] - l10[i]), float accumulation in i order;
     * 0 when k >= n or k == 0.  A running prefix reproduces the reference's loop bit for bit. */
    float* pre = (float*)malloc(sizeof(float) * ((size_t)n / 2 + 2));
    pre[0] = 0.0f;
    for (uint32_t i = 1; i <= n / 2; ++i) pre[i] = pre[i - 1] + (l10[n - i + 1] - l10[i]);
    for (uint32_t k = 0; k <= n; ++k) {
        if (k == 0 || k >= n) { logc_n[k] = 0.0f; continue; }
        const uint32_t kk = (n - k < k) ? n - k : k;
        logc_n[k] = pre[kk];
    }
    for (uint32_t nn = 0; nn <= n; ++nn) {
        if (ks >= nn || ks == 0) { logc_k[nn] = 0.0f; continue; }
        const uint32_t kk = (nn - ks < ks) ? nn - ks : ks;
        float r = 0.0f;
        for (uint32_t i = 1; i <= kk; ++i) r += l10[nn - i + 1] - l10[i];
        logc_k[nn] = r;
    }
    free(pre); free(l10);
}

/* ---------------------------------------------------------------- AC-RANSAC */

/* optional per-model trace (debugging aid for parity hunts): rows of 5 doubles
 * (iteration, model, #residuals <= bound, NFA, improved) */
static double* g_trace = NULL;
static int g_trace_cap = 0, g_trace_n = 0;
void orc_set_trace(double* buf, int cap_rows) { g_trace = buf; g_trace_cap = cap_rows; g_trace_n = 0; }
int  orc_trace_rows(void) { return g_trace_n; }
static int g_dbg_iter = -1;
static double g_dbg[20];
void orc_set_debug_iter(int it) { g_dbg_iter = it; }
const double* orc_debug_sample(void) { return g_dbg; }

static int cmp_u32(const void* pa, const void* pb)
{
    const uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
    return a < b ? -1 : (a > b ? 1 : 0);
}

typedef struct { double e; uint32_t idx; } err_idx;

static int cmp_err(const void* pa, const void* pb)
{
    const err_idx* a = (const err_idx*)pa;
    const err_idx* b = (const err_idx*)pb;
    if (a->e != b->e) return a->e < b->e ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}

/* kind 0: fundamental matrix (ACKernelAdaptor<SevenPointSolver, SymmetricEpipolarDistanceError, UnnormalizerT>,
 *         point-to-line); kind 1: homography (ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>,
 *         point-to-point: logalpha0 = log10(pi / A / N2(0,0)^2), multError = 1). */
/* kind 2: essential matrix (ACKernelAdaptorEssential<FivePointKernel, EpipolarDistanceError>): samples are fitted in
 *         camera coordinates (K^-1 x), residuals are taken in PIXELS through F = K2^-T E K1^-1, no normalisation:
 *         logalpha0 = log10(2 D / A * 0.5), multError = 0.5, threshold = precision^2, unormalizeError(e) = e. */
static int acransac_core(int kind, const double* xI, const double* xJ, int m,
                         int wI, int hI, int wJ, int hJ,
                         double precision_px, uint32_t max_iter,
                         uint64_t seed, uint32_t I, uint32_t J,
                         uint32_t* inliers_out, orc_fresult* res, const double* K1, const double* K2)
{
    const int SS = kind == 0 ? 7 : (kind == 1 ? 4 : 5);            /* MINIMUM_SAMPLES */
    const int MAX_MODELS = kind == 0 ? 3 : (kind == 1 ? 1 : 10);
    memset(res, 0, sizeof(*res));
    res->nfa = INFINITY;
    if (m <= SS) return 0;
    const uint32_t n = (uint32_t)m;

    /* ACKernelAdaptor: normalise with N = diag(s, s, 1) + translation, s = 1/sqrt(w*h) */
    double s1 = 1.0 / sqrt((double)(wI * hI));   /* static_cast<double>(width*height): int product */
    double s2 = 1.0 / sqrt((double)(wJ * hJ));
    double t1x = -0.5 * wI * s1, t1y = -0.5 * hI * s1;
    double t2x = -0.5 * wJ * s2, t2y = -0.5 * hJ * s2;
    double K1i[9] = {0}, K2i[9] = {0};
    if (kind == 2) { s1 = s2 = 1.0; t1x = t1y = t2x = t2y = 0.0; orc_inv3(K1, K1i); orc_inv3(K2, K2i); }
    double* x1 = (double*)malloc(sizeof(double) * 2 * n);
    double* x2 = (double*)malloc(sizeof(double) * 2 * n);
    for (uint32_t k = 0; k < n; ++k) {
        x1[2 * k] = s1 * xI[2 * k] + t1x; x1[2 * k + 1] = s1 * xI[2 * k + 1] + t1y;
        x2[2 * k] = s2 * xJ[2 * k] + t2x; x2[2 * k + 1] = s2 * xJ[2 * k + 1] + t2y;
    }
    /* point-to-line: logalpha0 = log10(2 D / A / N2(0,0)), multError = 0.5 */
    const double Dd = sqrt((double)wJ * (double)wJ + (double)hJ * (double)hJ);
    const double Aa = (double)wJ * (double)hJ;
    const double logalpha0 = kind == 0 ? log10(2.0 * Dd / Aa / s2) : (kind == 1 ? log10(M_PI / Aa / (s2 * s2)) : log10(2.0 * Dd / Aa * 0.5));
    const double multError = kind == 1 ? 1.0 : 0.5;
    /* camera coordinates of every point for the 5-point fits: hnormalized(K^-1 (x, y, 1)) */
    double* xk1 = NULL; double* xk2 = NULL;
    if (kind == 2) {
        xk1 = (double*)malloc(sizeof(double) * 2 * n); xk2 = (double*)malloc(sizeof(double) * 2 * n);
        for (uint32_t k = 0; k < n; ++k) {
            const double w1 = K1i[6] * xI[2 * k] + K1i[7] * xI[2 * k + 1] + K1i[8];
            xk1[2 * k] = (K1i[0] * xI[2 * k] + K1i[1] * xI[2 * k + 1] + K1i[2]) / w1;
            xk1[2 * k + 1] = (K1i[3] * xI[2 * k] + K1i[4] * xI[2 * k + 1] + K1i[5]) / w1;
            const double w2 = K2i[6] * xJ[2 * k] + K2i[7] * xJ[2 * k + 1] + K2i[8];
            xk2[2 * k] = (K2i[0] * xJ[2 * k] + K2i[1] * xJ[2 * k + 1] + K2i[2]) / w2;
            xk2[2 * k + 1] = (K2i[3] * xJ[2 * k] + K2i[4] * xJ[2 * k + 1] + K2i[5]) / w2;
        }
    }
    const double maxThreshold = precision_px * precision_px * s2 * s2;   /* precision = 4.0^2, in N2 units */

    const double loge0 = log10((double)MAX_MODELS * (double)(n - SS));
    float* logc_n = (float*)malloc(sizeof(float) * (n + 1));
    float* logc_k = (float*)malloc(sizeof(float) * (n + 1));
    orc_logcombi_tables(n, SS, logc_n, logc_k);

    uint32_t* pool = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t pool_size = n;
    for (uint32_t k = 0; k < n; ++k) pool[k] = k;
    uint32_t* inl = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t n_inl = 0;
    err_idx* er = (err_idx*)malloc(sizeof(err_idx) * n);
    double* rs = (double*)malloc(sizeof(double) * n);

    double minNFA = INFINITY, errorMax = INFINITY;
    double bestF[9] = {0};
    uint32_t nIter = max_iter;
    uint32_t nIterReserve = nIter / 10;
    nIter -= nIterReserve;
    int acMode = !(precision_px < INFINITY);
    uint32_t n_models_total = 0, iters_done = 0;

    for (uint32_t iter = 0; iter < nIter; ++iter) {
        uint32_t smp[7];
        orc_sample_n(seed, I, J, iter, pool, pool_size, (uint32_t)SS, smp);
        double sx1[14], sx2[14];
        const double* fx1 = kind == 2 ? xk1 : x1;
        const double* fx2 = kind == 2 ? xk2 : x2;
        for (int k = 0; k < SS; ++k) {
            sx1[2 * k] = fx1[2 * smp[k]]; sx1[2 * k + 1] = fx1[2 * smp[k] + 1];
            sx2[2 * k] = fx2[2 * smp[k]]; sx2[2 * k + 1] = fx2[2 * smp[k] + 1];
        }
        double Fs[90];
        const int nm = kind == 0 ? orc_seven_point(sx1, sx2, Fs) : (kind == 1 ? orc_four_point_h(sx1, sx2, Fs) : orc_five_point(sx1, sx2, Fs));
        if ((int)iter == g_dbg_iter) {
            for (int k = 0; k < 7; ++k) g_dbg[k] = smp[k];
            g_dbg[7] = nm; g_dbg[8] = pool_size; g_dbg[9] = iter;
            for (int k = 0; k < 7; ++k) { g_dbg[10 + k] = -1; for (uint32_t q = 0; q < pool_size; ++q) if (pool[q] == smp[k]) g_dbg[10 + k] = q; }
        }
        int better = 0;
        for (int k = 0; k < nm; ++k) {
            const double* F = Fs + 9 * k;
            ++n_models_total;
            double FE[9];
            if (kind == 2) orc_f_from_e(F, K1i, K2i, FE);
            for (uint32_t p = 0; p < n; ++p)
                rs[p] = kind == 0 ? orc_sym_epipolar_err(F, x1[2 * p], x1[2 * p + 1], x2[2 * p], x2[2 * p + 1])
                      : kind == 1 ? orc_h_asym_err(F, x1[2 * p], x1[2 * p + 1], x2[2 * p], x2[2 * p + 1])
                                  : orc_epipolar_dist_err(FE, x1[2 * p], x1[2 * p + 1], x2[2 * p], x2[2 * p + 1]);
            if (!acMode) {
                uint32_t nInlier = 0;
                for (uint32_t p = 0; p < n; ++p) if (rs[p] <= maxThreshold) ++nInlier;
                if ((double)nInlier > 2.5 * SS) acMode = 1;
            }
            if (!acMode && g_trace && g_trace_n < g_trace_cap) {
                uint32_t cntb = 0;
                for (uint32_t p = 0; p < n; ++p) if (rs[p] <= maxThreshold) ++cntb;
                double* t = g_trace + 5 * g_trace_n++;
                t[0] = iter; t[1] = k + 10.0 * smp[0]; t[2] = cntb + 10000.0 * pool_size; t[3] = INFINITY; t[4] = 2.0 * 7;
            }
            if (acMode) {
                for (uint32_t p = 0; p < n; ++p) { er[p].e = rs[p]; er[p].idx = p; }
                qsort(er, n, sizeof(err_idx), cmp_err);
                /* bestNFA */
                double bestv = INFINITY; uint32_t bestk = (uint32_t)SS;
                for (uint32_t kk = SS + 1; kk <= n && er[kk - 1].e <= maxThreshold; ++kk) {
                    const double logalpha = logalpha0 + multError * log10(er[kk - 1].e + (double)FLT_EPSILON);
                    const double v = loge0 + logalpha * (double)(kk - SS) + (double)logc_n[kk] + (double)logc_k[kk];
                    if (v < bestv) { bestv = v; bestk = kk; }
                }
                if (g_trace && g_trace_n < g_trace_cap) {
                    uint32_t cntb = 0;
                    for (uint32_t p = 0; p < n; ++p) if (rs[p] <= maxThreshold) ++cntb;
                    double* t = g_trace + 5 * g_trace_n++;
                    t[0] = iter; t[1] = k + 10.0 * smp[0]; t[2] = cntb + 10000.0 * pool_size; t[3] = bestv; t[4] = (bestv < minNFA) + 2.0 * bestk;
                }
                if (bestv < minNFA) {
                    better = 1;
                    minNFA = bestv;
                    n_inl = bestk;
                    for (uint32_t q = 0; q < bestk; ++q) inl[q] = er[q].idx;
                    errorMax = er[bestk - 1].e;
                    memcpy(bestF, F, sizeof(bestF));
                }
            }
        }
        iters_done = iter + 1;
        if ((better && minNFA < 0) || (iter + 1 == nIter && nIterReserve)) {
            if (n_inl == 0) {
                nIter++;
                nIterReserve--;
            } else {
                /* vec_index = vec_inliers.  Restatement decision: the pool holds the inlier SET in
                 * ascending index order.  OpenMVG keeps the residual order, but the residuals of the 7
                 * points a model was fitted to are rounding noise (~1e-30), so their order -- and with it
                 * which element a pool POSITION denotes -- would differ between compilers/devices. */
                memcpy(pool, inl, sizeof(uint32_t) * n_inl);
                qsort(pool, n_inl, sizeof(uint32_t), cmp_u32);
                pool_size = n_inl;
                if (nIterReserve) {
                    nIter = iter + 1 + nIterReserve;
                    nIterReserve = 0;
                }
            }
        }
    }

    if (minNFA >= 0) n_inl = 0;
    res->nfa = minNFA;
    res->n_iter = iters_done;
    res->n_models = n_models_total;
    res->n_inliers = n_inl;
    if (n_inl > 0) {
        /* Unnormalize: F = N2^T * F * N1 (UnnormalizerT) ; H = N2^-1 * H * N1 (UnnormalizerI);
         * threshold = sqrt(errorMax) / N2(0,0) */
        const double N1[9] = { s1, 0, t1x, 0, s1, t1y, 0, 0, 1 };
        const double N2[9] = { s2, 0, t2x, 0, s2, t2y, 0, 0, 1 };
        const double N2i[9] = { 1.0 / s2, 0, -t2x / s2, 0, 1.0 / s2, -t2y / s2, 0, 0, 1 };
        double T[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double v = 0.0;
                for (int k = 0; k < 3; ++k) v += (kind == 0 ? N2[3 * k + r] : N2i[3 * r + k]) * bestF[3 * k + c];
                T[3 * r + c] = v;
            }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                double v = 0.0;
                for (int k = 0; k < 3; ++k) v += T[3 * r + k] * N1[3 * k + c];
                res->F[3 * r + c] = v;
            }
        res->threshold = sqrt(errorMax) / s2;
        if (kind == 2) { memcpy(res->F, bestF, sizeof(bestF)); res->threshold = errorMax; }   /* E itself; unormalizeError(e) = e */
        memcpy(inliers_out, inl, sizeof(uint32_t) * n_inl);
    }
    res->accepted = ((double)n_inl > 2.5 * SS) ? 1 : 0;   /* F_ACRobust / H_ACRobust acceptance */

    free(rs); free(er); free(inl); free(pool); free(logc_k); free(logc_n); free(x2); free(x1); free(xk1); free(xk2);
    return (int)n_inl;
}

int orc_acransac_F(const double* xI, const double* xJ, int m, int wI, int hI, int wJ, int hJ,
                   double precision_px, uint32_t max_iter, uint64_t seed, uint32_t I, uint32_t J,
                   uint32_t* inliers_out, orc_fresult* res)
{
    return acransac_core(0, xI, xJ, m, wI, hI, wJ, hJ, precision_px, max_iter, seed, I, J, inliers_out, res, NULL, NULL);
}

int orc_acransac_H(const double* xI, const double* xJ, int m, int wI, int hI, int wJ, int hJ,
                   double precision_px, uint32_t max_iter, uint64_t seed, uint32_t I, uint32_t J,
                   uint32_t* inliers_out, orc_fresult* res)
{
    return acransac_core(1, xI, xJ, m, wI, hI, wJ, hJ, precision_px, max_iter, seed, I, J, inliers_out, res, NULL, NULL);
}

int orc_acransac_E(const double* xI, const double* xJ, int m, int wI, int hI, int wJ, int hJ,
                   const double* K1, const double* K2,
                   double precision_px, uint32_t max_iter, uint64_t seed, uint32_t I, uint32_t J,
                   uint32_t* inliers_out, orc_fresult* res)
{
    return acransac_core(2, xI, xJ, m, wI, hI, wJ, hJ, precision_px, max_iter, seed, I, J, inliers_out, res, K1, K2);
}

/* ---------------------------------------------------------------- collection filter */

static int64_t filter_collection(int kind, int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t* out_counts, orc_match* out, double* F_out,
                                const double* Ks /* 9 per image, K[8] == 0 marks "no intrinsics" */,
                                uint32_t prune_min_count, float prune_min_ratio)
{
    (void)n_images; (void)n_rows;
    int64_t* offs = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_pairs + 1));
    offs[0] = 0;
    for (int64_t p = 0; p < n_pairs; ++p) offs[p + 1] = offs[p] + counts[p];
    orc_match** res = (orc_match**)calloc((size_t)n_pairs, sizeof(orc_match*));
    memset(out_counts, 0, sizeof(uint32_t) * (size_t)n_pairs);

#pragma omp parallel for schedule(dynamic)
    for (int64_t p = 0; p < n_pairs; ++p) {
        const uint32_t m = counts[p];
        if (m == 0) continue;
        const uint32_t I = pairs[2 * p], J = pairs[2 * p + 1];
        const orc_match* pm = matches + offs[p];
        double* xI = (double*)malloc(sizeof(double) * 2 * m);
        double* xJ = (double*)malloc(sizeof(double) * 2 * m);
        for (uint32_t k = 0; k < m; ++k) {
            xI[2 * k] = (double)xy[I][2 * (size_t)pm[k].i]; xI[2 * k + 1] = (double)xy[I][2 * (size_t)pm[k].i + 1];
            xJ[2 * k] = (double)xy[J][2 * (size_t)pm[k].j]; xJ[2 * k + 1] = (double)xy[J][2 * (size_t)pm[k].j + 1];
        }
        uint32_t* inl = (uint32_t*)malloc(sizeof(uint32_t) * m);
        orc_fresult fr;
        int ni = 0;
        memset(&fr, 0, sizeof(fr));
        /* E_ACRobust: both views need valid pinhole intrinsics, otherwise the pair is not estimated */
        if (kind != 2 || (Ks[9 * I + 8] != 0.0 && Ks[9 * J + 8] != 0.0))
            ni = acransac_core(kind, xI, xJ, (int)m, (int)widths[I], (int)heights[I], (int)widths[J], (int)heights[J],
                               precision_px, max_iter, seed, I, J, inl, &fr, Ks ? Ks + 9 * I : NULL, Ks ? Ks + 9 * J : NULL);
        /* Regard3D's extra check after the E filter (src/R3DComputeMatches.cpp:2175-2192): drop pairs with poor overlap */
        if (fr.accepted && kind == 2 && ((uint32_t)ni < prune_min_count || (float)ni / (float)m < prune_min_ratio)) fr.accepted = 0;
        if (fr.accepted) {
            orc_match* r = (orc_match*)malloc(sizeof(orc_match) * (size_t)ni);
            for (int k = 0; k < ni; ++k) r[k] = pm[inl[k]];
            res[p] = r;
            out_counts[p] = (uint32_t)ni;
            if (F_out) memcpy(F_out + 9 * p, fr.F, sizeof(double) * 9);
        }
        free(inl); free(xJ); free(xI);
    }
    int64_t w = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        if (res[p]) { memcpy(out + w, res[p], sizeof(orc_match) * out_counts[p]); w += out_counts[p]; free(res[p]); }
    }
    free(res); free(offs);
    return w;
}

int64_t orc_filter_F_collection(int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t* out_counts, orc_match* out, double* F_out)
{
    return filter_collection(0, n_images, n_rows, xy, widths, heights, pairs, n_pairs, counts, matches,
                             precision_px, max_iter, seed, out_counts, out, F_out, NULL, 0, 0.f);
}

int64_t orc_filter_H_collection(int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t* out_counts, orc_match* out, double* H_out)
{
    return filter_collection(1, n_images, n_rows, xy, widths, heights, pairs, n_pairs, counts, matches,
                             precision_px, max_iter, seed, out_counts, out, H_out, NULL, 0, 0.f);
}

/* GeometricFilter_EMatrix_AC over the collection + Regard3D's overlap check (src/R3DComputeMatches.cpp:2169-2192):
 * a kept pair needs >= prune_min_count (50) geometric matches and a geometric/putative ratio >= prune_min_ratio (0.3). */
int64_t orc_filter_E_collection(int n_images, const int* n_rows, const float* const* xy,
                                const uint32_t* widths, const uint32_t* heights, const double* Ks,
                                const uint32_t* pairs, int64_t n_pairs,
                                const uint32_t* counts, const orc_match* matches,
                                double precision_px, uint32_t max_iter, uint64_t seed,
                                uint32_t prune_min_count, float prune_min_ratio,
                                uint32_t* out_counts, orc_match* out, double* E_out)
{
    return filter_collection(2, n_images, n_rows, xy, widths, heights, pairs, n_pairs, counts, matches,
                             precision_px, max_iter, seed, out_counts, out, E_out, Ks, prune_min_count, prune_min_ratio);
}
