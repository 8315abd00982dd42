/* oracle/mrpt.c -- CPU model of the MRPT plugin path (matchingAlgorithm 5).  TEST INFRASTRUCTURE ONLY.
 *
 * Restates the index the reference's arm builds and queries:
 *   mrpt_match                 /root/reference/src/R3DComputeMatches.cpp:423-491   (n_trees 26, depth 6, votes 5, K = 2, b_squared_metric false)
 *   ArrayMatcher_mrpt          /root/reference/src/utils/matcher_mrpt.h:45-259      (depth clamp :93, retry with votes - 1 :224-232)
 *   Mrpt::grow / grow_subtree  /root/reference/src/thirdparty/mrpt/mrpt.h:84-137, 1051-1078
 *   Mrpt::query / exact_knn    mrpt.h:661-728, 1083-1128;  leaf sizes :1664-1690;  sparse random matrix :1248-1268
 *
 * PARITY UNPINNED, and deliberately not bit-compatible with a reference build in three places (mrpt.h is Eigen code and Eigen is
 * not in the image, so no reference-built index exists to compare with anyway):
 *   1. the sparse random matrix: the reference draws std::uniform_real_distribution / std::normal_distribution from std::mt19937
 *      (implementation-defined streams); here entry (row, col) is non-zero iff a counter-based uniform is <= density, its value a
 *      Box-Muller normal of two more counter-based uniforms (host double libm) -- the same density and distribution;
 *   2. ties: std::nth_element leaves rows whose projection EQUALS the median on either side; here the left child takes the
 *      ceil(n / 2) smallest rows in the order (projection, row).  exact_knn's partial_sort is given the order (distance, row);
 *   3. summation order: projections accumulate the non-zero terms of a row of the matrix in ascending column order in float (no
 *      FMA); candidate distances are the reference's brute-force metric orc_l2sq_f32 (Eigen's vectorised squaredNorm sums in
 *      another order).  Distances are returned as sqrtf of that, as exact_knn returns them.
 * What is held to this model: the GPU path, bit for bit (tests/test_gpu_mrpt.py); and the model to the arm's purpose: recall
 * against the exhaustive matcher on the fixtures (tests/test_oracle_mrpt.py).
 */
#include "r3d_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mr_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
/* uniform in [0, 1) with 53 bits, stream `k` of entry (row, col) */
static double mr_uniform(uint64_t seed, uint32_t row, uint32_t col, uint32_t k)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    const uint64_t a = mr_mix64(seed + G * (1ULL + (((uint64_t)row << 32) | (uint64_t)col)));
    const uint64_t b = mr_mix64(a + G * (uint64_t)(k + 1u));
    return (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

/* the random matrix, dense [n_pool][dim] (zeros where the sparse matrix has no entry) */
void orc_mrpt_random_matrix(uint32_t n_pool, uint32_t dim, float density, uint64_t seed, float* R)
{
    const double TWO_PI = 6.283185307179586476925286766559;
    for (uint32_t j = 0; j < n_pool; ++j)
        for (uint32_t c = 0; c < dim; ++c) {
            float v = 0.0f;
            if (!(mr_uniform(seed, j, c, 0) > (double)density)) {                 /* mrpt.h:1260: `if (uni_dist(gen) > density) continue;` */
                const double u1 = 1.0 - mr_uniform(seed, j, c, 1), u2 = mr_uniform(seed, j, c, 2);      /* u1 in (0, 1] */
                v = (float)(sqrt(-2.0 * log(u1)) * cos(TWO_PI * u2));
            }
            R[(size_t)j * dim + c] = v;
        }
}

/* ArrayMatcher_mrpt::Build (matcher_mrpt.h:93): max(2, min(depth, floor(log2 n) - 1)) */
uint32_t orc_mrpt_depth_for(uint32_t n, uint32_t depth)
{
    int lg = 0;
    while ((2u << lg) <= n && lg < 30) ++lg;                                     /* floor(log2 n) */
    int d = (int)depth < lg - 1 ? (int)depth : lg - 1;
    return (uint32_t)(d > 2 ? d : 2);
}

typedef struct {
    uint32_t n, dim, n_trees, depth, n_pool;
    float* R;            /* [n_pool][dim] */
    float* splits;       /* [n_trees][2^depth - 1]  heap order: node i, children 2i + 1 / 2i + 2 */
    int32_t* leaves;     /* [n_trees][n]           rows of every tree, leaf after leaf */
    int32_t* leaf_first; /* [2^depth + 1] */
    const float* X;
} orc_mrpt;

/* count_leaf_sizes (mrpt.h:1650-1662): a node of n rows gives n - n / 2 to the left, n / 2 to the right */
static void mr_leaf_sizes(uint32_t n, uint32_t level, uint32_t depth, int32_t* out, uint32_t* pos)
{
    if (level == depth) { out[(*pos)++] = (int32_t)n; return; }
    mr_leaf_sizes(n - n / 2, level + 1, depth, out, pos);
    mr_leaf_sizes(n / 2, level + 1, depth, out, pos);
}

typedef struct { float v; int32_t i; } mr_key;
static int mr_cmp(const void* a, const void* b)
{
    const mr_key* x = (const mr_key*)a; const mr_key* y = (const mr_key*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

static void mr_grow_subtree(orc_mrpt* ix, int32_t* idx, uint32_t n, uint32_t level, uint32_t node, uint32_t tree, const float* proj /* [depth][n_rows] */)
{
    if (level == ix->depth) return;
    mr_key* k = (mr_key*)malloc(sizeof(mr_key) * (n ? n : 1));
    for (uint32_t a = 0; a < n; ++a) { k[a].i = idx[a]; k[a].v = proj[(size_t)level * ix->n + (size_t)idx[a]]; }
    qsort(k, n, sizeof(mr_key), mr_cmp);
    for (uint32_t a = 0; a < n; ++a) idx[a] = k[a].i;
    const uint32_t n_left = n - n / 2;
    float split;
    if (n % 2) split = k[n_left - 1].v;                                          /* mrpt.h:1065-1066 */
    else { const float sum = k[n_left].v + k[n_left - 1].v; split = (float)((double)sum / 2.0); }   /* :1072-1073: a float sum, halved in double */
    ix->splits[(size_t)tree * ((1u << ix->depth) - 1u) + node] = split;
    free(k);
    mr_grow_subtree(ix, idx, n_left, level + 1, 2 * node + 1, tree, proj);
    mr_grow_subtree(ix, idx + n_left, n / 2, level + 1, 2 * node + 2, tree, proj);
}

/* projection of one row: the non-zero terms in ascending column order, float, no FMA (adding x * 0 changes nothing: dense loop) */
static float mr_project(const float* Rrow, const float* x, uint32_t dim)
{
    float acc = 0.0f;
    for (uint32_t c = 0; c < dim; ++c) { const float t = Rrow[c] * x[c]; acc = acc + t; }
    return acc;
}

void* orc_mrpt_build(const float* X, uint32_t n, uint32_t dim, uint32_t n_trees, uint32_t depth, float density, uint64_t seed)
{
    orc_mrpt* ix = (orc_mrpt*)calloc(1, sizeof(orc_mrpt));
    ix->n = n; ix->dim = dim; ix->n_trees = n_trees; ix->depth = depth; ix->n_pool = n_trees * depth; ix->X = X;
    ix->R = (float*)malloc(sizeof(float) * (size_t)ix->n_pool * dim);
    orc_mrpt_random_matrix(ix->n_pool, dim, density, seed, ix->R);
    const uint32_t n_leaf = 1u << depth;
    ix->splits = (float*)calloc((size_t)n_trees * (n_leaf - 1u), sizeof(float));
    ix->leaves = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_trees * n);
    ix->leaf_first = (int32_t*)calloc(n_leaf + 1u, sizeof(int32_t));
    {
        int32_t* sizes = (int32_t*)malloc(sizeof(int32_t) * n_leaf);
        uint32_t pos = 0;
        mr_leaf_sizes(n, 0, depth, sizes, &pos);
        for (uint32_t l = 0; l < n_leaf; ++l) ix->leaf_first[l + 1] = ix->leaf_first[l] + sizes[l];
        free(sizes);
    }
    float* proj = (float*)malloc(sizeof(float) * (size_t)depth * n);
    for (uint32_t t = 0; t < n_trees; ++t) {
        for (uint32_t l = 0; l < depth; ++l)
            for (uint32_t i = 0; i < n; ++i) proj[(size_t)l * n + i] = mr_project(ix->R + (size_t)(t * depth + l) * dim, X + (size_t)i * dim, dim);
        int32_t* idx = ix->leaves + (size_t)t * n;
        for (uint32_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
        mr_grow_subtree(ix, idx, n, 0, 0, t, proj);
    }
    free(proj);
    return ix;
}

void orc_mrpt_free(void* p)
{
    orc_mrpt* ix = (orc_mrpt*)p;
    if (!ix) return;
    free(ix->R); free(ix->splits); free(ix->leaves); free(ix->leaf_first); free(ix);
}

/* arrays of the index: R [n_pool][dim], splits [n_trees][2^depth - 1], leaves [n_trees][n], leaf_first [2^depth + 1] (any may be NULL) */
void orc_mrpt_export(const void* p, float* R, float* splits, int32_t* leaves, int32_t* leaf_first)
{
    const orc_mrpt* ix = (const orc_mrpt*)p;
    const uint32_t n_leaf = 1u << ix->depth;
    if (R) memcpy(R, ix->R, sizeof(float) * (size_t)ix->n_pool * ix->dim);
    if (splits) memcpy(splits, ix->splits, sizeof(float) * (size_t)ix->n_trees * (n_leaf - 1u));
    if (leaves) memcpy(leaves, ix->leaves, sizeof(int32_t) * (size_t)ix->n_trees * ix->n);
    if (leaf_first) memcpy(leaf_first, ix->leaf_first, sizeof(int32_t) * (n_leaf + 1u));
}

/* Mrpt::query(q, 2, votes) then, as ArrayMatcher_mrpt::SearchNeighbours does, once more with votes - 1 when fewer than two rows were
 * elected; idx -1 / dist -1 for a query that stays without two neighbours.  dist = sqrtf(squared L2).  n_elected (optional): size of
 * the candidate set of the attempt that answered. */
static void mr_query_one(const orc_mrpt* ix, const float* q, uint32_t votes_required, uint8_t* votes, int32_t* elected,
                         int32_t* out_idx, float* out_dist, uint32_t* n_elected_out)
{
    const uint32_t n_leaf = 1u << ix->depth;
    memset(votes, 0, ix->n);
    uint32_t ne = 0;
    for (uint32_t t = 0; t < ix->n_trees; ++t) {
        uint32_t node = 0;
        for (uint32_t d = 0; d < ix->depth; ++d) {
            const float pj = mr_project(ix->R + (size_t)(t * ix->depth + d) * ix->dim, q, ix->dim);
            const float sp = ix->splits[(size_t)t * (n_leaf - 1u) + node];
            node = (pj <= sp) ? 2 * node + 1 : 2 * node + 2;                      /* mrpt.h:698-702 */
        }
        const uint32_t leaf = node - (n_leaf - 1u);
        const int32_t* rows = ix->leaves + (size_t)t * ix->n;
        for (int32_t a = ix->leaf_first[leaf]; a < ix->leaf_first[leaf + 1]; ++a) {
            const int32_t r = rows[a];
            if (++votes[r] == votes_required) elected[ne++] = r;                  /* :718-719 */
        }
    }
    *n_elected_out = ne;
    int32_t i0 = -1, i1 = -1; float d0 = 0.f, d1 = 0.f;
    for (uint32_t e = 0; e < ne; ++e) {
        const int32_t r = elected[e];
        const float d = orc_l2sq_f32(ix->X + (size_t)r * ix->dim, q, ix->dim);
        if (i0 < 0 || d < d0 || (d == d0 && r < i0)) { i1 = i0; d1 = d0; i0 = r; d0 = d; }
        else if (i1 < 0 || d < d1 || (d == d1 && r < i1)) { i1 = r; d1 = d; }
    }
    out_idx[0] = i0; out_idx[1] = i1;
    out_dist[0] = i0 >= 0 ? sqrtf(d0) : -1.0f; out_dist[1] = i1 >= 0 ? sqrtf(d1) : -1.0f;
}

int orc_mrpt_knn2(const void* p, const float* query, uint32_t nq, uint32_t votes_required, int32_t* idx, float* dist, uint32_t* n_elected)
{
    const orc_mrpt* ix = (const orc_mrpt*)p;
    if (!ix || votes_required < 1 || votes_required > ix->n_trees) return -1;
    uint8_t* votes = (uint8_t*)malloc(ix->n);
    int32_t* elected = (int32_t*)malloc(sizeof(int32_t) * ix->n);
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t ne = 0;
        mr_query_one(ix, query + (size_t)q * ix->dim, votes_required, votes, elected, idx + 2 * q, dist + 2 * q, &ne);
        if ((idx[2 * q] < 0 || idx[2 * q + 1] < 0) && votes_required > 1)        /* matcher_mrpt.h:224-232 */
            mr_query_one(ix, query + (size_t)q * ix->dim, votes_required - 1, votes, elected, idx + 2 * q, dist + 2 * q, &ne);
        if (idx[2 * q] < 0 || idx[2 * q + 1] < 0) { idx[2 * q] = idx[2 * q + 1] = -1; dist[2 * q] = dist[2 * q + 1] = -1.0f; }   /* !isValid: the query is dropped */
        if (n_elected) n_elected[q] = ne;
    }
    free(votes); free(elected);
    return 0;
}
