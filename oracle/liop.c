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
