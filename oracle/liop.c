/*
 * oracle/liop.c -- CPU restatement of the LIOP descriptor as Regard3D uses it.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).  Build without FMA contraction.
 *
 * Follows the reference's vendored VLFeat copy /root/reference/src/thirdparty/liop/vl_liop.c:
 *   r3d_vl_liopdesc_new   :344-421  circular pixel list (dx^2+dy^2 <= (int)((c - r + 0.6)^2)),
 *                                   4 neighbour sample positions per pixel on a circle of radius 6
 *                                   starting at atan2(y, x) (rotation invariance)
 *   r3d_vl_liopdesc_process :465-580 intensity sort (vl_qsort-def.h: middle pivot, Lomuto partition, "<= 0"),
 *                                   ordinal spatial bins, bilinear samples in double, permutation index,
 *                                   weight = #pairs differing by more than thr = 5/255 * (max - min),
 *                                   L2 normalisation with the float sum / float sqrt quirk (:567-575)
 * with the parameters Regard3D passes: new_basic(41) -> 4 neighbours, 6 bins, radius 6  (:231-234, and
 * /root/reference/src/Regard3DFeatures.cpp:727-752) -> 144 dimensions.
 *
 * THIS part of the oracle IS pinned by the reference itself: vl_liop.c compiles stand-alone, so
 * oracle/_ref/libref_liop.so (oracle/Makefile) is the real thing, tests/test_oracle_liop.py compares
 * the restatement with it bit for bit, and tests/golden/liop_patches.npz holds its outputs.
 */
#include "r3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LIOP_NEIGH 4
#define LIOP_BINS 6
#define LIOP_RADIUS 6.0
#define LIOP_THR (-(5.0 / 255))

typedef struct {
    int side, n_pix;
    int* pix;            /* pixel offsets x + y*side of the circular support, scan order */
    double *sx, *sy;     /* [n_pix][4] sample positions */
} liop_geom;

static liop_geom* geom_new(int side)
{
    liop_geom* g = (liop_geom*)calloc(1, sizeof(*g));
    g->side = side;
    g->pix = (int*)malloc(sizeof(int) * side * side);
    const int center = (side - 1) / 2;
    const double t = center - LIOP_RADIUS + 0.6;
    const long t2 = (long)(t * t);
    for (int y = 0; y < side; ++y)
        for (int x = 0; x < side; ++x) {
            const long dx = x - center, dy = y - center;
            if (x == 0 && y == 0) continue;
            if (dx * dx + dy * dy <= t2) g->pix[g->n_pix++] = x + y * side;
        }
    g->sx = (double*)malloc(sizeof(double) * LIOP_NEIGH * g->n_pix);
    g->sy = (double*)malloc(sizeof(double) * LIOP_NEIGH * g->n_pix);
    const double dangle = 2 * M_PI / (double)LIOP_NEIGH;
    for (int i = 0; i < g->n_pix; ++i) {
        const double x = (g->pix[i] % side) - center, y = (g->pix[i] / side) - center;
        const double angle0 = atan2(y, x);
        for (int k = 0; k < LIOP_NEIGH; ++k) {
            g->sx[k + LIOP_NEIGH * i] = x + LIOP_RADIUS * cos(angle0 + dangle * k) + center;
            g->sy[k + LIOP_NEIGH * i] = y + LIOP_RADIUS * sin(angle0 + dangle * k) + center;
        }
    }
    return g;
}

static void geom_free(liop_geom* g) { free(g->pix); free(g->sx); free(g->sy); free(g); }

/* geometry tables for the device path and the tests */
int orc_liop_geometry(int side, int* n_pix, int* pix, double* sx, double* sy)
{
    liop_geom* g = geom_new(side);
    *n_pix = g->n_pix;
    if (pix) memcpy(pix, g->pix, sizeof(int) * g->n_pix);
    if (sx) memcpy(sx, g->sx, sizeof(double) * LIOP_NEIGH * g->n_pix);
    if (sy) memcpy(sy, g->sy, sizeof(double) * LIOP_NEIGH * g->n_pix);
    geom_free(g);
    return 0;
}

/* the reference's quick sort (vl_qsort-def.h): permutation `perm` ordered by val[perm[.]], pivot =
 * middle element swapped to the end, one left-to-right pass moving "<= pivot" entries down, recursion
 * on both parts.  Restated because the order it leaves EQUAL values in decides which ordinal bin
 * (and which permutation index) tied pixels fall into. */
static void perm_qsort(const float* val, int* perm, long begin, long end)
{
    long pivot = (end + begin) / 2, low = begin;
    int t = perm[pivot]; perm[pivot] = perm[end]; perm[end] = t;
    for (long i = begin; i < end; ++i)
        if (val[perm[i]] - val[perm[end]] <= 0) { t = perm[low]; perm[low] = perm[i]; perm[i] = t; ++low; }
    t = perm[low]; perm[low] = perm[end]; perm[end] = t;
    if (low > begin) perm_qsort(val, perm, begin, low - 1);
    if (low < end) perm_qsort(val, perm, low + 1, end);
}

static long floor_d(double x)
{
    const long xi = (long)x;
    return (x >= 0 || (double)xi == x) ? xi : xi - 1;
}

static void liop_one(const liop_geom* g, const float* patch, float* desc, float* inten, int* perm)
{
    const int L = g->side, N = g->n_pix;
    memset(desc, 0, sizeof(float) * 24 * LIOP_BINS);
    for (int i = 0; i < N; ++i) { inten[i] = patch[g->pix[i]]; perm[i] = i; }
    perm_qsort(inten, perm, 0, N - 1);
    const float thr = (float)(-LIOP_THR) * (inten[perm[N - 1]] - inten[perm[0]]);   /* - threshold * (max - min), float */
    const int area = N / LIOP_BINS;
    int bin_end = area, bin = 0, offset = 0;
    for (int i = 0; i < N; ++i) {
        if (i >= bin_end && bin < LIOP_BINS - 1) { bin_end += area; ++bin; offset += 24; }
        const double* sx = g->sx + LIOP_NEIGH * perm[i];
        const double* sy = g->sy + LIOP_NEIGH * perm[i];
        float nv[LIOP_NEIGH]; int np[LIOP_NEIGH];
        for (int k = 0; k < LIOP_NEIGH; ++k) {
            const double x = sx[k], y = sy[k];
            const long ix = floor_d(x), iy = floor_d(y);
            const double wx = x - ix, wy = y - iy;
            double a = 0, b = 0, c = 0, d = 0;
            if (ix >= 0 && iy >= 0) a = patch[ix + iy * L];
            if (ix < L - 1 && iy >= 0) b = patch[ix + 1 + iy * L];
            if (ix >= 0 && iy < L - 1) c = patch[ix + (iy + 1) * L];
            if (ix < L - 1 && iy < L - 1) d = patch[ix + 1 + (iy + 1) * L];
            np[k] = k;
            nv[k] = (float)((1.0 - wy) * (a + (b - a) * wx) + wy * (c + (d - c) * wx));
        }
        perm_qsort(nv, np, 0, LIOP_NEIGH - 1);
        /* lexicographic index of the permutation (Lehmer code); np is consumed */
        int index = 0;
        for (int a = 0; a < LIOP_NEIGH; ++a) {
            index = index * (LIOP_NEIGH - a) + np[a];
            for (int b = a + 1; b < LIOP_NEIGH; ++b) if (np[b] > np[a]) --np[b];
        }
        float weight = 0;
        for (int a = 0; a < LIOP_NEIGH; ++a)
            for (int b = a + 1; b < LIOP_NEIGH; ++b)
                weight += (nv[a] > nv[b] + thr || nv[b] > nv[a] + thr);
        desc[index + offset] += weight;
    }
    float norm = 0;
    for (int i = 0; i < 24 * LIOP_BINS; ++i) norm += desc[i] * desc[i];
    norm = (float)(sqrt(norm) > 1e-12 ? sqrt(norm) : 1e-12);     /* VL_MAX(sqrt(norm), 1e-12) stored to a float */
    for (int i = 0; i < 24 * LIOP_BINS; ++i) desc[i] /= norm;
}

/* patches: n x side x side floats (row-major); desc: n x 144 */
int orc_liop_describe(const float* patches, int n, int side, float* desc)
{
    liop_geom* g = geom_new(side);
#pragma omp parallel
    {
        float* inten = (float*)malloc(sizeof(float) * g->n_pix);
        int* perm = (int*)malloc(sizeof(int) * g->n_pix);
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p)
            liop_one(g, patches + (size_t)p * side * side, desc + (size_t)p * 144, inten, perm);
        free(inten); free(perm);
    }
    geom_free(g);
    return 0;
}

/* ---------------------------------------------------------------- patch extraction
 * Regard3DFeatures::extractLIOPFeatures, /root/reference/src/Regard3DFeatures.cpp:768-808: per keypoint
 * a 41x41 patch = cv::warpAffine(img, M, Size(41,41), INTER_LINEAR | WARP_INVERSE_MAP) followed by
 * cv::GaussianBlur(patch, patch, Size(0,0), 1.2).  OpenCV 4.0 is an external dependency that is not in
 * /root/reference and not in this image, so this sub-stage restates OpenCV's documented scalar code paths and
 * its parity is UNPINNED:
 *   warpAffine: M (float) -> double; fixed point with AB_BITS = 10, INTER_BITS = 5: adelta[x] = rint(M0*x*1024),
 *     bdelta[x] = rint(M3*x*1024), X0 = rint((M1*y + M2)*1024) + 16, Y0 likewise; X = (X0 + adelta[x]) >> 5;
 *     integer part X >> 5, fraction (X & 31)/32; bilinear weights from the 32x32 float table
 *     {(1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx}; value = ((S00*w0 + S01*w1) + S10*w2) + S11*w3 in float;
 *     BORDER_CONSTANT 0 for taps outside the image;
 *   GaussianBlur: ksize = cvRound(1.2*4*2+1)|1 = 11, kernel k[i] = (float)exp(-0.5/sigma^2 * (i-5)^2) normalised
 *     with a double sum of the float taps, row pass s = sum_{k=0..10} k[k]*S[x+k-5] in k order, column pass in the
 *     symmetric form s = k[5]*S0 + sum_{j=1..5} k[5+j]*(S[+j] + S[-j]), BORDER_REFLECT_101, float throughout, no FMA.
 * kps: n x 4 floats (x, y, size = diameter, angle in degrees) -- cv::KeyPoint as detectKeypoints leaves it. */

static int reflect101(int p, int len)
{
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

void orc_liop_affine(const float* kp, float kp_size_factor, float* M6)
{
    const int patchResolution = 20, patchSize = 41;
    const float x = kp[0], y = kp[1];
    const float angle = -90.0f - kp[3];
    const float scale = kp[2] / (float)patchSize * kp_size_factor;
    const float alpha = (float)(scale * cos(angle * M_PI / 180.0f));
    const float beta = (float)(scale * sin(angle * M_PI / 180.0f));
    const float trans_x = x - (float)patchResolution, trans_y = y - (float)patchResolution;
    M6[0] = alpha; M6[1] = beta;
    M6[2] = beta * trans_y + alpha * trans_x - beta * y + (1.0f - alpha) * x;
    M6[3] = -beta; M6[4] = alpha;
    M6[5] = alpha * trans_y - beta * trans_x + beta * x + (1.0f - alpha) * y;
}

int orc_liop_extract_patches(const float* image, int w, int h, const float* kps, int n, float kp_size_factor, float* patches)
{
    const int S = 41;
    float kern[11];
    {
        const double scale2X = -0.5 / (1.2 * 1.2);
        double sum = 0;
        for (int i = 0; i < 11; ++i) { const double x = i - 5.0; kern[i] = (float)exp(scale2X * x * x); sum += kern[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 11; ++i) kern[i] = (float)(kern[i] * sum);
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < n; ++p) {
        float Mf[6]; double M[6];
        orc_liop_affine(kps + 4 * (size_t)p, kp_size_factor, Mf);
        for (int k = 0; k < 6; ++k) M[k] = Mf[k];
        float warped[41 * 41], rowp[41 * 41];
        for (int y = 0; y < S; ++y) {
            const int X0 = (int)lrint((M[1] * y + M[2]) * 1024) + 16;
            const int Y0 = (int)lrint((M[4] * y + M[5]) * 1024) + 16;
            for (int x = 0; x < S; ++x) {
                const int X = (X0 + (int)lrint(M[0] * x * 1024)) >> 5;
                const int Y = (Y0 + (int)lrint(M[3] * x * 1024)) >> 5;
                int sx = X >> 5, sy = Y >> 5;
                sx = sx > 32767 ? 32767 : (sx < -32768 ? -32768 : sx);            /* saturate_cast<short> */
                sy = sy > 32767 ? 32767 : (sy < -32768 ? -32768 : sy);
                const float fx = (float)(X & 31) * (1.f / 32), fy = (float)(Y & 31) * (1.f / 32);
                const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
                const float v0 = (sx >= 0 && sx < w && sy >= 0 && sy < h) ? image[(size_t)sy * w + sx] : 0.f;
                const float v1 = (sx + 1 >= 0 && sx + 1 < w && sy >= 0 && sy < h) ? image[(size_t)sy * w + sx + 1] : 0.f;
                const float v2 = (sx >= 0 && sx < w && sy + 1 >= 0 && sy + 1 < h) ? image[(size_t)(sy + 1) * w + sx] : 0.f;
                const float v3 = (sx + 1 >= 0 && sx + 1 < w && sy + 1 >= 0 && sy + 1 < h) ? image[(size_t)(sy + 1) * w + sx + 1] : 0.f;
                warped[y * S + x] = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
            }
        }
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                float s = kern[0] * warped[y * S + reflect101(x - 5, S)];
                for (int k = 1; k < 11; ++k) s += kern[k] * warped[y * S + reflect101(x + k - 5, S)];
                rowp[y * S + x] = s;
            }
        float* out = patches + (size_t)p * S * S;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                float s = kern[5] * rowp[y * S + x];
                for (int j = 1; j <= 5; ++j) s += kern[5 + j] * (rowp[reflect101(y + j, S) * S + x] + rowp[reflect101(y - j, S) * S + x]);
                out[y * S + x] = s;
            }
    }
    return 0;
}
