/*
 * oracle/matching.c -- CPU restatement of the putative-matching half of the hot path.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).  Build with -ffp-contract=off.
 *
 * Follows:
 *   OpenMVG 1.4 matching/metric.hpp L2<T> (4-way unrolled squared L2, float accumulator) and
 *   metric_hamming.hpp -- external, restated from SURVEY.md A.2; the reference calls L2<float>
 *   as its distance oracle at /root/reference/src/R3DComputeMatches.cpp:285-291,313-323.
 *   OpenMVG ArrayMatcherBruteForce::SearchNeighbours + RegionsMatcherT::MatchDistanceRatio
 *   (SURVEY.md A.3); plugin contract /root/reference/src/utils/matcher_kgraph.h:205-251.
 *   Collection loop nest: /root/reference/src/R3DComputeMatches.cpp:428-489.
 */
#include "r3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- metrics */

/* L2<float>::operator(): result += d0*d0 + d1*d1 + d2*d2 + d3*d3, then a scalar tail.
 * C evaluates the sum left-to-right: ((d0^2 + d1^2) + d2^2) + d3^2, then adds to result. */
float orc_l2sq_f32(const float* a, const float* b, size_t n)
{
    float result = 0.0f;
    size_t k = 0;
    for (; k + 3 < n; k += 4) {
        const float d0 = a[k] - b[k];
        const float d1 = a[k + 1] - b[k + 1];
        const float d2 = a[k + 2] - b[k + 2];
        const float d3 = a[k + 3] - b[k + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; k < n; ++k) {
        const float d0 = a[k] - b[k];
        result += d0 * d0;
    }
    return result;
}

/* L2<unsigned char>: Accumulator<uchar>::Type is float; the element difference is formed in
 * int (integer promotion) and converted to float. */
float orc_l2sq_u8(const uint8_t* a, const uint8_t* b, size_t n)
{
    float result = 0.0f;
    size_t k = 0;
    for (; k + 3 < n; k += 4) {
        const float d0 = (float)((int)a[k] - (int)b[k]);
        const float d1 = (float)((int)a[k + 1] - (int)b[k + 1]);
        const float d2 = (float)((int)a[k + 2] - (int)b[k + 2]);
        const float d3 = (float)((int)a[k + 3] - (int)b[k + 3]);
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; k < n; ++k) {
        const float d0 = (float)((int)a[k] - (int)b[k]);
        result += d0 * d0;
    }
    return result;
}

uint32_t orc_hamming(const uint8_t* a, const uint8_t* b, size_t nbytes)
{
    uint32_t r = 0;
    for (size_t k = 0; k < nbytes; ++k)
        r += (uint32_t)__builtin_popcount((unsigned)(a[k] ^ b[k]));
    return r;
}

/* ---------------------------------------------------------------- brute-force 2-NN */

/* Running top-2 with "lowest dataset index wins on equal distance": rows are visited in
 * ascending order and only a strictly smaller distance displaces an entry. */
#define ORC_TOP2_UPDATE(d, r)                      \
    do {                                           \
        if ((d) < d0) { d1 = d0; i1 = i0; d0 = (d); i0 = (r); } \
        else if ((d) < d1) { d1 = (d); i1 = (r); } \
    } while (0)

int orc_knn2_l2_f32(const float* dataset, int nI, const float* query, int nJ, int dim,
                    int32_t* idx, float* dist)
{
    if (nJ < 1 || nI < 2) return -1;
    for (int q = 0; q < nJ; ++q) {
        const float* qv = query + (size_t)q * dim;
        float d0 = INFINITY, d1 = INFINITY;
        int32_t i0 = -1, i1 = -1;
        for (int r = 0; r < nI; ++r) {
            const float d = orc_l2sq_f32(dataset + (size_t)r * dim, qv, (size_t)dim);
            ORC_TOP2_UPDATE(d, r);
        }
        idx[2 * q] = i0; idx[2 * q + 1] = i1;
        dist[2 * q] = d0; dist[2 * q + 1] = d1;
    }
    return 0;
}

int orc_knn2_l2_u8(const uint8_t* dataset, int nI, const uint8_t* query, int nJ, int dim,
                   int32_t* idx, float* dist)
{
    if (nJ < 1 || nI < 2) return -1;
    for (int q = 0; q < nJ; ++q) {
        const uint8_t* qv = query + (size_t)q * dim;
        float d0 = INFINITY, d1 = INFINITY;
        int32_t i0 = -1, i1 = -1;
        for (int r = 0; r < nI; ++r) {
            const float d = orc_l2sq_u8(dataset + (size_t)r * dim, qv, (size_t)dim);
            ORC_TOP2_UPDATE(d, r);
        }
        idx[2 * q] = i0; idx[2 * q + 1] = i1;
        dist[2 * q] = d0; dist[2 * q + 1] = d1;
    }
    return 0;
}

int orc_knn2_hamming(const uint8_t* dataset, int nI, const uint8_t* query, int nJ, int nbytes,
                     int32_t* idx, uint32_t* dist)
{
    if (nJ < 1 || nI < 2) return -1;
    for (int q = 0; q < nJ; ++q) {
        const uint8_t* qv = query + (size_t)q * nbytes;
        uint32_t d0 = UINT32_MAX, d1 = UINT32_MAX;
        int32_t i0 = -1, i1 = -1;
        for (int r = 0; r < nI; ++r) {
            const uint32_t d = orc_hamming(dataset + (size_t)r * nbytes, qv, (size_t)nbytes);
            ORC_TOP2_UPDATE(d, r);
        }
        idx[2 * q] = i0; idx[2 * q + 1] = i1;
        dist[2 * q] = d0; dist[2 * q + 1] = d1;
    }
    return 0;
}

/* ---------------------------------------------------------------- MatchDistanceRatio */

static int cmp_match(const void* pa, const void* pb)
{
    const orc_match* a = (const orc_match*)pa;
    const orc_match* b = (const orc_match*)pb;
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    return 0;
}

typedef struct { float x1, y1, x2, y2; uint32_t pos; } coord_key;

static int cmp_coord(const void* pa, const void* pb)
{
    const coord_key* a = (const coord_key*)pa;
    const coord_key* b = (const coord_key*)pb;
    if (a->x1 != b->x1) return a->x1 < b->x1 ? -1 : 1;
    if (a->y1 != b->y1) return a->y1 < b->y1 ? -1 : 1;
    if (a->x2 != b->x2) return a->x2 < b->x2 ? -1 : 1;
    if (a->y2 != b->y2) return a->y2 < b->y2 ? -1 : 1;
    return a->pos < b->pos ? -1 : (a->pos > b->pos ? 1 : 0);
}

/* NNdistanceRatio + both de-duplications on a precomputed 2-NN table (shared by the brute-force and the ANN drivers).
 * dist_is_u32: Hamming distances (converted to float first). */
static int ratio_and_dedup(const int32_t* idx, const void* dist, int dist_is_u32, int nJ,
                           const float* xyI, const float* xyJ, float R, orc_match* out)
{
    /* keep q iff dist[2q] < R * dist[2q+1], R = ratio^2 for squared metrics.  All arithmetic in float. */
    int m = 0;
    for (int q = 0; q < nJ; ++q) {
        float a, b;
        if (dist_is_u32) { a = (float)((const uint32_t*)dist)[2 * q]; b = (float)((const uint32_t*)dist)[2 * q + 1]; }
        else             { a = ((const float*)dist)[2 * q];           b = ((const float*)dist)[2 * q + 1]; }
        if (a < R * b) {
            out[m].i = (uint32_t)idx[2 * q];   /* row of I (dataset) */
            out[m].j = (uint32_t)q;            /* row of J (query)   */
            ++m;
        }
    }

    /* IndMatch::getDeduplicated: sort by (i_, j_) + unique (q is unique, so nothing drops). */
    qsort(out, (size_t)m, sizeof(orc_match), cmp_match);

    /* IndMatchDecorator<float>::getDeduplicated: one match per distinct (xI,yI,xJ,yJ).
     * Restatement: keep the smallest (i_, j_) of every group, output stays (i_, j_)-sorted. */
    if (xyI && xyJ && m > 1) {
        coord_key* ck = (coord_key*)malloc(sizeof(coord_key) * (size_t)m);
        for (int k = 0; k < m; ++k) {
            ck[k].x1 = xyI[2 * (size_t)out[k].i]; ck[k].y1 = xyI[2 * (size_t)out[k].i + 1];
            ck[k].x2 = xyJ[2 * (size_t)out[k].j]; ck[k].y2 = xyJ[2 * (size_t)out[k].j + 1];
            ck[k].pos = (uint32_t)k;
        }
        qsort(ck, (size_t)m, sizeof(coord_key), cmp_coord);
        unsigned char* drop = (unsigned char*)calloc((size_t)m, 1);
        for (int k = 1; k < m; ++k)
            if (ck[k].x1 == ck[k - 1].x1 && ck[k].y1 == ck[k - 1].y1 &&
                ck[k].x2 == ck[k - 1].x2 && ck[k].y2 == ck[k - 1].y2)
                drop[ck[k].pos] = 1;   /* ck sorted by pos inside a group: first = smallest (i,j) */
        int w = 0;
        for (int k = 0; k < m; ++k)
            if (!drop[k]) out[w++] = out[k];
        m = w;
        free(drop); free(ck);
    }
    return m;
}

int orc_ratio_dedup_f32(const int32_t* idx, const float* dist, int nJ, const float* xyI, const float* xyJ,
                        float dist_ratio, int squared_metric, orc_match* out)
{
    return ratio_and_dedup(idx, dist, 0, nJ, xyI, xyJ, squared_metric ? dist_ratio * dist_ratio : dist_ratio, out);
}

int orc_match_distance_ratio(int dtype, const void* descI, int nI, const float* xyI,
                             const void* descJ, int nJ, const float* xyJ, int dim,
                             float dist_ratio, int squared_metric, orc_match* out)
{
    /* RegionsMatcherT::MatchDistanceRatio returns at once when either side is empty; the
     * brute-force SearchNeighbours fails (-> no matches) when NN=2 > nI. */
    if (nI < 2 || nJ < 1) return 0;

    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nJ);
    void* dist = malloc(sizeof(float) * 2 * (size_t)nJ);
    int rc;
    if (dtype == 0)      rc = orc_knn2_l2_f32((const float*)descI, nI, (const float*)descJ, nJ, dim, idx, (float*)dist);
    else if (dtype == 1) rc = orc_knn2_l2_u8((const uint8_t*)descI, nI, (const uint8_t*)descJ, nJ, dim, idx, (float*)dist);
    else                 rc = orc_knn2_hamming((const uint8_t*)descI, nI, (const uint8_t*)descJ, nJ, dim, idx, (uint32_t*)dist);
    if (rc != 0) { free(idx); free(dist); return 0; }
    const float R = squared_metric ? dist_ratio * dist_ratio : dist_ratio;
    const int m = ratio_and_dedup(idx, dist, dtype == 2, nJ, xyI, xyJ, R, out);
    free(idx); free(dist);
    return m;
}

/* ---------------------------------------------------------------- collection matcher */

typedef struct { uint32_t I; int64_t p; } ip_t;

static int cmp_ip(const void* a, const void* b)
{
    const ip_t* x = (const ip_t*)a;
    const ip_t* y = (const ip_t*)b;
    if (x->I != y->I) return x->I < y->I ? -1 : 1;
    return x->p < y->p ? -1 : (x->p > y->p ? 1 : 0);
}

int64_t orc_match_collection(int dtype, int n_images, const void* const* desc, const int* n_rows,
                             const float* const* xy, int dim, const uint32_t* pairs, int64_t n_pairs,
                             float dist_ratio, int squared_metric,
                             uint32_t* counts, orc_match* out, int64_t out_cap)
{
    (void)n_images;
    /* Per-pair scratch results, gathered afterwards in input order (the reference inserts into
     * a std::map under omp critical; a CSR over the input pair list carries the same content). */
    orc_match** res = (orc_match**)calloc((size_t)n_pairs, sizeof(orc_match*));
    memset(counts, 0, sizeof(uint32_t) * (size_t)n_pairs);

    /* group by I: pairs with equal first index are processed together, I serial, J parallel
     * (src/R3DComputeMatches.cpp:437-489).  The input list need not be sorted. */
    int64_t start = 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_pairs);
    {
        ip_t* ip = (ip_t*)malloc(sizeof(ip_t) * (size_t)n_pairs);
        for (int64_t p = 0; p < n_pairs; ++p) { ip[p].I = pairs[2 * p]; ip[p].p = p; }
        qsort(ip, (size_t)n_pairs, sizeof(ip_t), cmp_ip);
        for (int64_t p = 0; p < n_pairs; ++p) order[p] = ip[p].p;
        free(ip);
    }
    while (start < n_pairs) {
        const uint32_t I = pairs[2 * order[start]];
        int64_t end = start;
        while (end < n_pairs && pairs[2 * order[end]] == I) ++end;
        if (n_rows[I] > 0) {
#pragma omp parallel for schedule(dynamic)
            for (int64_t s = start; s < end; ++s) {
                const int64_t p = order[s];
                const uint32_t J = pairs[2 * p + 1];
                if (n_rows[J] == 0) continue;
                orc_match* tmp = (orc_match*)malloc(sizeof(orc_match) * (size_t)n_rows[J]);
                const int m = orc_match_distance_ratio(dtype, desc[I], n_rows[I], xy ? xy[I] : NULL,
                                                       desc[J], n_rows[J], xy ? xy[J] : NULL, dim,
                                                       dist_ratio, squared_metric, tmp);
                if (m > 0) { res[p] = tmp; counts[p] = (uint32_t)m; }
                else free(tmp);
            }
        }
        start = end;
    }
    free(order);

    int64_t total = 0;
    for (int64_t p = 0; p < n_pairs; ++p) total += counts[p];
    int64_t rc = total;
    if (total > out_cap) rc = -1;
    int64_t w = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        if (res[p]) {
            if (rc >= 0) { memcpy(out + w, res[p], sizeof(orc_match) * counts[p]); w += counts[p]; }
            free(res[p]);
        }
    }
    free(res);
    return rc;
}
