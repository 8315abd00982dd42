/*
 * hnsw.c -- CPU restatement of the HNSW plugin path (matchingAlgorithm 6..8 of the reference's dispatch).
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).
 *
 * Follows (reference = /root/reference):
 *   - src/utils/matcher_hnsw.h:53-83          Build: HierarchicalNSW(space, n, M, efConstruction), addPoint per row, setEf(ef)
 *   - src/utils/matcher_hnsw.h:150-170        SearchNeighbours: searchKnn(query, NN), results reversed into ascending order
 *   - src/R3DComputeMatches.cpp:533-565       presets (efConstruction / ef / M): 112 / 5 / 5, 112 / 10 / 15, 100 / 15 / 19
 *   - src/thirdparty/hnswlib/hnswlib/hnswalg.h:146-151   getRandomLevel (std::default_random_engine seeded 100)
 *   - ... :153-213  searchBaseLayer (construction), :214-280 searchBaseLayerST (query), :282-322 getNeighborsByHeuristic2,
 *     :338-437 mutuallyConnectNewElement, :639-735 addPoint, :737-778 searchKnn
 *   - src/thirdparty/hnswlib/hnswlib/space_l2.h:40-75    L2SqrSIMD16Ext, AVX arm (oracle/_ref is built with -mavx): eight
 *     interleaved accumulators, horizontal sum left to right, no FMA
 *
 * What is implementation-defined in the reference and restated here literally, because results depend on it:
 *   - std::priority_queue = std::push_heap / std::pop_heap of libstdc++ (bits/stl_heap.h: __push_heap, __adjust_heap).  hnswlib
 *     compares heap entries by distance ONLY (CompareByFirst), so which of several equally distant entries leaves a heap first is
 *     decided by the sift order -- on integer-valued SIFT bins equal distances are common.
 *   - std::default_random_engine = minstd_rand0 (x <- 16807 x mod 2^31 - 1), std::uniform_real_distribution<double>(0, 1) =
 *     generate_canonical<double, 53>: two engine draws per level, (d1 + d2 * R) / R^2 with R = 2^31 - 2.
 * PINNED by the reference-built library: tests/test_oracle_hnsw.py holds the levels, every link list, the entry point and the
 * searchKnn results of this file against oracle/_ref/libref_hnsw.so built single-threaded (ref_hnsw_export), live in the authoring
 * container and through tests/golden/hnsw_ref_index.npz everywhere.  (The reference adds rows 1 .. n-1 from an OpenMP loop, so ITS
 * index depends on thread timing; the single-thread insertion order is the one reproducible instance.)
 */
#include "r3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; uint32_t id; } hn_pair;
typedef struct { hn_pair* v; size_t n, cap; int lex; } hn_heap;      /* lex = 0: CompareByFirst; 1: std::less<pair> */

static int hn_less(const hn_heap* h, hn_pair a, hn_pair b) { return h->lex ? (a.d < b.d || (!(b.d < a.d) && a.id < b.id)) : (a.d < b.d); }
static void hn_reserve(hn_heap* h, size_t n) { if (n > h->cap) { h->cap = n * 2 + 16; h->v = (hn_pair*)realloc(h->v, h->cap * sizeof(hn_pair)); } }
/* bits/stl_heap.h __push_heap */
static void hn_sift_up(hn_heap* h, size_t hole, size_t top, hn_pair value)
{
    size_t parent = (hole - 1) / 2;
    while (hole > top && hn_less(h, h->v[parent], value)) { h->v[hole] = h->v[parent]; hole = parent; parent = (hole - 1) / 2; }
    h->v[hole] = value;
}
static void hn_push(hn_heap* h, float d, uint32_t id)
{
    hn_reserve(h, h->n + 1);
    hn_pair value = {d, id};
    h->v[h->n] = value; h->n += 1;
    hn_sift_up(h, h->n - 1, 0, value);
}
/* pop_heap + pop_back: __pop_heap -> __adjust_heap */
static void hn_pop(hn_heap* h)
{
    if (h->n > 1) {
        const size_t len = h->n - 1;
        hn_pair value = h->v[len];
        h->v[len] = h->v[0];
        size_t hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (hn_less(h, h->v[second], h->v[second - 1])) second--;
            h->v[hole] = h->v[second]; hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            h->v[hole] = h->v[second - 1]; hole = second - 1;
        }
        hn_sift_up(h, hole, 0, value);
    }
    h->n -= 1;
}

/* L2SqrSIMD16Ext, AVX arm: dim % 16 == 0 */
float orc_hnsw_l2(const float* a, const float* b, uint32_t dim)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < dim; i += 8)
        for (int l = 0; l < 8; ++l) { const float t = a[i + l] - b[i + l]; acc[l] = acc[l] + t * t; }
    return acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
}

/* getRandomLevel for rows 0 .. n-1 in insertion order */
void orc_hnsw_levels(uint32_t n, uint32_t M, uint32_t seed, int32_t* out)
{
    const double mult = 1.0 / log(1.0 * (double)M);
    uint64_t x = seed % 2147483647u;
    if (x == 0) x = 1;                                          /* linear_congruential_engine::seed: 0 -> 1 when c == 0 */
    const long double R = 2147483646.0L;                        /* max() - min() + 1 = (2^31 - 2) - 1 + 1 */
    for (uint32_t i = 0; i < n; ++i) {
        double sum = 0.0, tmp = 1.0;
        for (int k = 0; k < 2; ++k) {                           /* generate_canonical<double, 53>: m = max(1, (53 + 30 - 1) / 30) = 2 draws */
            x = (x * 16807u) % 2147483647u;
            sum += (double)(x - 1u) * tmp;
            tmp = (double)((long double)tmp * R);
        }
        double u = sum / tmp;
        if (u >= 1.0) u = nextafter(1.0, 0.0);
        u = u * (1.0 - 0.0) + 0.0;                              /* uniform_real_distribution(0, 1) */
        const double r = -log(u) * mult;
        out[i] = (int32_t)r;
    }
}

struct orc_hnsw {
    uint32_t n, dim, M, maxM0, efc;
    const float* data;
    int32_t* level;
    int32_t* l0;            /* [n][1 + maxM0] */
    int32_t** up;           /* [n] -> [level][1 + M] */
    int32_t enter, maxlevel;
};

static int32_t* hn_list(const orc_hnsw* g, uint32_t id, int level)
{
    return level == 0 ? g->l0 + (size_t)id * (1 + g->maxM0) : g->up[id] + (size_t)(level - 1) * (1 + g->M);
}

/* searchBaseLayer (construction, any layer, ef = efConstruction) and searchBaseLayerST (layer 0, ef given): the same loop */
static void hn_search_layer(const orc_hnsw* g, uint32_t ep, const float* q, int layer, size_t ef, hn_heap* top, uint32_t* visited, uint32_t tag)
{
    hn_heap cand = {0, 0, 0, 0};
    top->n = 0; top->lex = 0;
    const float d0 = orc_hnsw_l2(q, g->data + (size_t)ep * g->dim, g->dim);
    hn_push(top, d0, ep);
    hn_push(&cand, -d0, ep);
    visited[ep] = tag;
    float lower = d0;
    while (cand.n) {
        const hn_pair cur = cand.v[0];
        if ((-cur.d) > lower) break;
        hn_pop(&cand);
        const int32_t* L = hn_list(g, cur.id, layer);
        const int size = L[0];
        for (int j = 1; j <= size; ++j) {
            const uint32_t c = (uint32_t)L[j];
            if (visited[c] == tag) continue;
            visited[c] = tag;
            const float d = orc_hnsw_l2(q, g->data + (size_t)c * g->dim, g->dim);
            if (top->v[0].d > d || top->n < ef) {
                hn_push(&cand, -d, c);
                hn_push(top, d, c);
                if (top->n > ef) hn_pop(top);
                lower = top->v[0].d;
            }
        }
    }
    free(cand.v);
}

/* getNeighborsByHeuristic2: `top` (CompareByFirst max-heap) -> at most M diverse entries, again as a CompareByFirst heap */
static void hn_heuristic(const orc_hnsw* g, hn_heap* top, size_t M)
{
    if (top->n < M) return;
    hn_heap closest = {0, 0, 0, 1};
    while (top->n) { hn_push(&closest, -top->v[0].d, top->v[0].id); hn_pop(top); }
    hn_pair* ret = (hn_pair*)malloc((M + 1) * sizeof(hn_pair));
    size_t nret = 0;
    while (closest.n) {
        if (nret >= M) break;
        const hn_pair cur = closest.v[0];
        const float dq = -cur.d;
        hn_pop(&closest);
        int good = 1;
        for (size_t s = 0; s < nret; ++s) {
            const float cd = orc_hnsw_l2(g->data + (size_t)ret[s].id * g->dim, g->data + (size_t)cur.id * g->dim, g->dim);
            if (cd < dq) { good = 0; break; }
        }
        if (good) ret[nret++] = cur;
    }
    for (size_t s = 0; s < nret; ++s) hn_push(top, -ret[s].d, ret[s].id);
    free(ret); free(closest.v);
}

static void hn_connect(orc_hnsw* g, const float* q, uint32_t cur_c, hn_heap* top, int level)
{
    const size_t Mcurmax = level ? g->M : g->maxM0;
    hn_heuristic(g, top, g->M);
    uint32_t sel[64]; size_t nsel = 0;
    while (top->n) { sel[nsel++] = top->v[0].id; hn_pop(top); }
    int32_t* lc = hn_list(g, cur_c, level);
    lc[0] = (int32_t)nsel;
    for (size_t k = 0; k < nsel; ++k) lc[1 + k] = (int32_t)sel[k];
    for (size_t k = 0; k < nsel; ++k) {
        int32_t* lo = hn_list(g, sel[k], level);
        const size_t sz = (size_t)lo[0];
        if (sz < Mcurmax) { lo[1 + sz] = (int32_t)cur_c; lo[0] = (int32_t)(sz + 1); }
        else {
            hn_heap cand = {0, 0, 0, 0};
            const float* pk = g->data + (size_t)sel[k] * g->dim;
            hn_push(&cand, orc_hnsw_l2(g->data + (size_t)cur_c * g->dim, pk, g->dim), cur_c);
            for (size_t j = 0; j < sz; ++j) hn_push(&cand, orc_hnsw_l2(g->data + (size_t)(uint32_t)lo[1 + j] * g->dim, pk, g->dim), (uint32_t)lo[1 + j]);
            hn_heuristic(g, &cand, Mcurmax);
            int indx = 0;
            while (cand.n) { lo[1 + indx] = (int32_t)cand.v[0].id; hn_pop(&cand); indx++; }
            lo[0] = indx;
            free(cand.v);
        }
    }
    (void)q;
}

/* HierarchicalNSW(space, n, M, efConstruction) + addPoint(row r, r) for r = 0 .. n-1 on one thread.  dim % 16 == 0, M <= 32. */
orc_hnsw* orc_hnsw_build(const float* data, uint32_t n, uint32_t dim, uint32_t M, uint32_t ef_construction, uint32_t seed)
{
    if (!data || n == 0 || (dim & 15u) || M < 2 || M > 32) return NULL;
    orc_hnsw* g = (orc_hnsw*)calloc(1, sizeof(orc_hnsw));
    g->n = n; g->dim = dim; g->M = M; g->maxM0 = 2 * M; g->efc = ef_construction; g->data = data;
    g->level = (int32_t*)malloc((size_t)n * 4);
    g->l0 = (int32_t*)calloc((size_t)n * (1 + g->maxM0), 4);
    g->up = (int32_t**)calloc(n, sizeof(int32_t*));
    g->enter = -1; g->maxlevel = -1;
    orc_hnsw_levels(n, M, seed, g->level);
    uint32_t* visited = (uint32_t*)calloc(n, 4);
    uint32_t tag = 0;
    hn_heap top = {0, 0, 0, 0};
    for (uint32_t c = 0; c < n; ++c) {
        const int curlevel = g->level[c];
        const int maxlevelcopy = g->maxlevel;
        uint32_t cur = (uint32_t)g->enter;
        if (curlevel) g->up[c] = (int32_t*)calloc((size_t)curlevel * (1 + M), 4);
        const float* q = data + (size_t)c * dim;
        if (g->enter != -1) {
            if (curlevel < maxlevelcopy) {
                float curdist = orc_hnsw_l2(q, data + (size_t)cur * dim, dim);
                for (int level = maxlevelcopy; level > curlevel; level--) {
                    int changed = 1;
                    while (changed) {
                        changed = 0;
                        const int32_t* L = hn_list(g, cur, level);
                        const int size = L[0];
                        for (int i = 0; i < size; ++i) {
                            const uint32_t cand = (uint32_t)L[1 + i];
                            const float d = orc_hnsw_l2(q, data + (size_t)cand * dim, dim);
                            if (d < curdist) { curdist = d; cur = cand; changed = 1; }
                        }
                    }
                }
            }
            for (int level = curlevel < maxlevelcopy ? curlevel : maxlevelcopy; level >= 0; level--) {
                hn_search_layer(g, cur, q, level, ef_construction, &top, visited, ++tag);
                hn_connect(g, q, c, &top, level);
            }
        } else { g->enter = 0; g->maxlevel = curlevel; }
        if (curlevel > maxlevelcopy) { g->enter = (int32_t)c; g->maxlevel = curlevel; }
    }
    free(visited); free(top.v);
    return g;
}

/* ---------------------------------------------------------------- the index the HIP path builds (DESIGN.md "HNSW")
 * hnswlib inserts rows one after another, every insertion searching the graph built so far: sequential by construction (and the
 * reference's own index varies from run to run, rows 1 .. n-1 being added from an OpenMP loop).  The HIP path builds the same KIND of
 * index in one batch, deterministically:
 *   levels      the reference's own draw (orc_hnsw_levels); entry point = the first row of the highest level, as in hnswlib
 *   layer 0     candidates of a row = its list in the exact 32-nearest-neighbour graph completed with reverse edges, 64 closest
 *               (orc_kgraph_build_exact: the index of the KGraph path), re-measured with hnswlib's distance
 *   layer L>0   candidates of a member = its min(32, members - 1) nearest members of the layer (exact scan)
 *   selection   hnswlib's getNeighborsByHeuristic2 (hnswalg.h:282-322) over the candidates in ascending (distance, row) order, at most
 *               2M links on layer 0 and M above; free places are then given to the closest rejected candidates (the paper's
 *               keepPrunedConnections); a candidate list shorter than the limit is kept whole (hnswlib: size < M returns)
 *   list order  farthest first, the order hnswlib's own lists come out of its max-heap
 * Chosen among four variants by recall at the reference's ef on the fixture scenes (forward-M + back links as in
 * mutuallyConnectNewElement, with / without refill: lower degree and LOWER recall than the reference-built index; this one: higher
 * in all six preset x scene cases -- tests/test_oracle_hnsw.py holds that).  Searching is hnswlib's searchKnn, unchanged
 * (orc_hnsw_knn2).  GPU parity against this model is bit-exact. */
typedef struct { float d; uint32_t id; } hb_c;
static int hb_cmp(const void* a, const void* b)
{
    const hb_c* x = (const hb_c*)a; const hb_c* y = (const hb_c*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
/* cand (ascending, nc <= 64) -> list[0] = count, list[1..] = the kept rows farthest first */
static void hb_select(const float* data, uint32_t dim, const hb_c* cand, uint32_t nc, uint32_t limit, int32_t* list)
{
    hb_c ret[64], pruned[64]; uint32_t nret = 0, np = 0;
    if (nc < limit) { for (uint32_t k = 0; k < nc; ++k) ret[nret++] = cand[k]; }
    else {
        for (uint32_t k = 0; k < nc && nret < limit; ++k) {
            int good = 1;
            for (uint32_t s = 0; s < nret; ++s)
                if (orc_hnsw_l2(data + (size_t)ret[s].id * dim, data + (size_t)cand[k].id * dim, dim) < cand[k].d) { good = 0; break; }
            if (good) ret[nret++] = cand[k]; else pruned[np++] = cand[k];
        }
        for (uint32_t k = 0; k < np && nret < limit; ++k) ret[nret++] = pruned[k];
        qsort(ret, nret, sizeof(hb_c), hb_cmp);
    }
    list[0] = (int32_t)nret;
    for (uint32_t k = 0; k < nret; ++k) list[1 + k] = (int32_t)ret[nret - 1 - k].id;
}

orc_hnsw* orc_hnsw_build_batch(const float* data, uint32_t n, uint32_t dim, uint32_t M, uint32_t seed)
{
    if (!data || n < 2 || (dim & 15u) || M < 2 || M > 32) return NULL;
    orc_hnsw* g = (orc_hnsw*)calloc(1, sizeof(orc_hnsw));
    g->n = n; g->dim = dim; g->M = M; g->maxM0 = 2 * M; g->efc = 0; g->data = data;
    g->level = (int32_t*)malloc((size_t)n * 4);
    g->l0 = (int32_t*)calloc((size_t)n * (1 + g->maxM0), 4);
    g->up = (int32_t**)calloc(n, sizeof(int32_t*));
    orc_hnsw_levels(n, M, seed, g->level);
    g->enter = 0; g->maxlevel = g->level[0];
    for (uint32_t i = 1; i < n; ++i) if (g->level[i] > g->maxlevel) { g->maxlevel = g->level[i]; g->enter = (int32_t)i; }
    for (uint32_t i = 0; i < n; ++i) if (g->level[i]) g->up[i] = (int32_t*)calloc((size_t)g->level[i] * (1 + M), 4);

    orc_kgraph* kg = orc_kgraph_build_exact(data, n, dim, 32, 64);
    if (!kg) { orc_hnsw_free(g); return NULL; }
    uint64_t* off = (uint64_t*)malloc(((size_t)n + 1) * 8);
    uint32_t* ids = (uint32_t*)malloc((size_t)orc_kgraph_edges(kg) * 4 + 4);
    orc_kgraph_export(kg, off, ids, NULL);
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; ++i) {
        hb_c c[64]; uint32_t k = 0;
        for (uint64_t e = off[i]; e < off[i + 1] && k < 64; ++e) { c[k].id = ids[e]; c[k].d = orc_hnsw_l2(data + (size_t)i * dim, data + (size_t)ids[e] * dim, dim); ++k; }
        qsort(c, k, sizeof(hb_c), hb_cmp);
        hb_select(data, dim, c, k, g->maxM0, g->l0 + (size_t)i * (1 + g->maxM0));
    }
    free(off); free(ids); orc_kgraph_free(kg);

    uint32_t* mem = (uint32_t*)malloc((size_t)n * 4);
    for (int L = 1; L <= g->maxlevel; ++L) {
        uint32_t nm = 0;
        for (uint32_t i = 0; i < n; ++i) if (g->level[i] >= L) mem[nm++] = i;
        const uint32_t K = nm - 1 < 32 ? nm - 1 : 32;
#pragma omp parallel for schedule(dynamic, 16)
        for (long a = 0; a < (long)nm; ++a) {
            hb_c c[33]; uint32_t k = 0;
            const uint32_t i = mem[a];
            for (uint32_t b = 0; b < nm && K; ++b) {
                if (b == (uint32_t)a) continue;
                hb_c e = { orc_hnsw_l2(data + (size_t)i * dim, data + (size_t)mem[b] * dim, dim), mem[b] };
                uint32_t p = k;                                    /* insertion keeps ascending (distance, row): members come in row order */
                while (p > 0 && e.d < c[p - 1].d) --p;
                if (p >= K) continue;
                for (uint32_t t = (k < K ? k : K - 1); t > p; --t) c[t] = c[t - 1];
                c[p] = e;
                if (k < K) ++k;
            }
            hb_select(data, dim, c, k, M, g->up[i] + (size_t)(L - 1) * (1 + M));
        }
    }
    free(mem);
    return g;
}

/* an index handed over as data (e.g. exported from the reference-built library): the arrays are borrowed, not copied */
orc_hnsw* orc_hnsw_from_arrays(const float* data, uint32_t n, uint32_t dim, uint32_t M, const int32_t* levels, const int32_t* links0,
                               const int32_t* up_off, const int32_t* up_links, int32_t enterpoint, int32_t maxlevel)
{
    orc_hnsw* g = (orc_hnsw*)calloc(1, sizeof(orc_hnsw));
    g->n = n; g->dim = dim; g->M = M; g->maxM0 = 2 * M; g->data = data;
    g->level = (int32_t*)malloc((size_t)n * 4);
    memcpy(g->level, levels, (size_t)n * 4);
    g->l0 = (int32_t*)malloc((size_t)n * (1 + g->maxM0) * 4);
    memcpy(g->l0, links0, (size_t)n * (1 + g->maxM0) * 4);
    g->up = (int32_t**)calloc(n, sizeof(int32_t*));
    for (uint32_t i = 0; i < n; ++i)
        if (levels[i] > 0) {
            g->up[i] = (int32_t*)malloc((size_t)levels[i] * (1 + M) * 4);
            memcpy(g->up[i], up_links + (size_t)up_off[i] * (1 + M), (size_t)levels[i] * (1 + M) * 4);
        }
    g->enter = enterpoint; g->maxlevel = maxlevel;
    return g;
}

void orc_hnsw_free(orc_hnsw* g)
{
    if (!g) return;
    for (uint32_t i = 0; i < g->n; ++i) free(g->up[i]);
    free(g->up); free(g->l0); free(g->level); free(g);
}

uint32_t orc_hnsw_up_rows(const orc_hnsw* g) { uint32_t r = 0; for (uint32_t i = 0; i < g->n; ++i) r += (uint32_t)g->level[i]; return r; }

/* the layout of ref_hnsw_export: links padded with -1 */
void orc_hnsw_export(const orc_hnsw* g, int32_t* levels, int32_t* links0, int32_t* up_off, int32_t* up_links, int32_t* enterpoint, int32_t* maxlevel)
{
    uint32_t rows = 0;
    for (uint32_t i = 0; i < g->n; ++i) {
        levels[i] = g->level[i];
        const int32_t* l = g->l0 + (size_t)i * (1 + g->maxM0);
        links0[(size_t)i * (1 + g->maxM0)] = l[0];
        for (uint32_t k = 0; k < g->maxM0; ++k) links0[(size_t)i * (1 + g->maxM0) + 1 + k] = (int32_t)k < l[0] ? l[1 + k] : -1;
        up_off[i] = (int32_t)rows;
        for (int L = 1; L <= g->level[i]; ++L) {
            const int32_t* u = g->up[i] + (size_t)(L - 1) * (1 + g->M);
            up_links[(size_t)rows * (1 + g->M)] = u[0];
            for (uint32_t k = 0; k < g->M; ++k) up_links[(size_t)rows * (1 + g->M) + 1 + k] = (int32_t)k < u[0] ? u[1 + k] : -1;
            ++rows;
        }
    }
    up_off[g->n] = (int32_t)rows;
    *enterpoint = g->enter; *maxlevel = g->maxlevel;
}

/* searchKnn(query, k = 2) with setEf(ef), results as ArrayMatcher_hnsw::SearchNeighbours orders them: ascending (distance, id).
 * idx -1 / dist 0 where fewer than two rows were found.  n_dist (optional) receives the distance evaluations. */
int orc_hnsw_knn2(const orc_hnsw* g, const float* query, uint32_t nq, uint32_t ef, int32_t* idx, float* dist, uint64_t* n_dist)
{
    if (!g || g->enter < 0 || !query) return -1;
    uint64_t evals = 0;
#pragma omp parallel reduction(+ : evals)
    {
        uint32_t* visited = (uint32_t*)calloc(g->n, 4);
        uint32_t tag = 0;
        hn_heap top = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 16)
        for (long qi = 0; qi < (long)nq; ++qi) {
            const float* q = query + (size_t)qi * g->dim;
            uint32_t cur = (uint32_t)g->enter;
            float curdist = orc_hnsw_l2(q, g->data + (size_t)cur * g->dim, g->dim);
            evals += 1;
            for (int level = g->maxlevel; level > 0; level--) {
                int changed = 1;
                while (changed) {
                    changed = 0;
                    const int32_t* L = hn_list(g, cur, level);
                    const int size = L[0];
                    for (int i = 0; i < size; ++i) {
                        const uint32_t cand = (uint32_t)L[1 + i];
                        const float d = orc_hnsw_l2(q, g->data + (size_t)cand * g->dim, g->dim);
                        evals += 1;
                        if (d < curdist) { curdist = d; cur = cand; changed = 1; }
                    }
                }
            }
            if (++tag == 0) { memset(visited, 0, (size_t)g->n * 4); tag = 1; }
            hn_search_layer(g, cur, q, 0, ef > 2 ? ef : 2, &top, visited, tag);
            while (top.n > 2) hn_pop(&top);
            /* results: std::priority_queue<pair<dist, label>> (lexicographic), popped largest first, then reversed */
            hn_pair r[2]; size_t nr = 0;
            while (top.n) { r[nr++] = top.v[0]; hn_pop(&top); }
            if (nr == 2) { const int swap = r[1].d < r[0].d || (!(r[0].d < r[1].d) && r[1].id < r[0].id); if (swap) { hn_pair t = r[0]; r[0] = r[1]; r[1] = t; } }
            for (size_t p = 0; p < 2; ++p) { idx[2 * qi + p] = p < nr ? (int32_t)r[p].id : -1; dist[2 * qi + p] = p < nr ? r[p].d : 0.0f; }
        }
        free(visited); free(top.v);
    }
    if (n_dist) *n_dist = evals;       /* (upper levels only; the layer-0 evaluations are counted by the GPU side's own statistics) */
    return 0;
}

/* ---------------------------------------------------------------- hnsw_match (src/R3DComputeMatches.cpp:497-593)
 * One index per first image (builder 0: the batch construction of the HIP path, 1: hnswlib's single-thread insertion), searchKnn(ef, 2)
 * of every row of J, then RegionsMatcherT::MatchDistanceRatio's ratio test and de-duplication (orc_ratio_dedup_f32, squared metric).
 * Views with fewer than min_rows rows are scanned exactly (what the HIP path does below 128 rows).  CSR convention of
 * orc_match_collection. */
int64_t orc_match_collection_hnsw(int n_images, const float* const* desc, const int* n_rows, const float* const* xy, int dim,
                                  const uint32_t* pairs, int64_t n_pairs, float dist_ratio, int builder, uint32_t M,
                                  uint32_t ef_construction, uint32_t ef, uint32_t seed, uint32_t min_rows,
                                  uint32_t* counts, orc_match* out, int64_t out_cap)
{
    orc_match** res = (orc_match**)calloc((size_t)n_pairs, sizeof(orc_match*));
    memset(counts, 0, sizeof(uint32_t) * (size_t)n_pairs);
    for (int I = 0; I < n_images; ++I) {
        orc_hnsw* g = NULL;
        for (int64_t p = 0; p < n_pairs; ++p) {
            if ((int)pairs[2 * p] != I) continue;
            const uint32_t J = pairs[2 * p + 1];
            if (n_rows[I] < 2 || n_rows[J] < 1) continue;
            const int scan = (uint32_t)n_rows[I] < min_rows;
            if (!g && !scan) {
                g = builder == 0 ? orc_hnsw_build_batch(desc[I], (uint32_t)n_rows[I], (uint32_t)dim, M, seed)
                                 : orc_hnsw_build(desc[I], (uint32_t)n_rows[I], (uint32_t)dim, M, ef_construction, seed);
                if (!g) break;
            }
            int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_rows[J]);
            float* dist = (float*)malloc(sizeof(float) * 2 * (size_t)n_rows[J]);
            const int rc = scan ? orc_knn2_l2_f32(desc[I], n_rows[I], desc[J], n_rows[J], dim, idx, dist)
                                : orc_hnsw_knn2(g, desc[J], (uint32_t)n_rows[J], ef, idx, dist, NULL);
            if (rc == 0) {
                orc_match* tmp = (orc_match*)malloc(sizeof(orc_match) * (size_t)n_rows[J]);
                const int m = orc_ratio_dedup_f32(idx, dist, n_rows[J], xy ? xy[I] : NULL, xy ? xy[J] : NULL, dist_ratio, 1, tmp);
                if (m > 0) { res[p] = tmp; counts[p] = (uint32_t)m; } else free(tmp);
            }
            free(idx); free(dist);
        }
        orc_hnsw_free(g);
    }
    int64_t total = 0;
    for (int64_t p = 0; p < n_pairs; ++p) total += counts[p];
    int64_t rc = total > out_cap ? -1 : total, w = 0;
    for (int64_t p = 0; p < n_pairs; ++p)
        if (res[p]) {
            if (rc >= 0) { memcpy(out + w, res[p], sizeof(orc_match) * counts[p]); w += counts[p]; }
            free(res[p]);
        }
    free(res);
    return rc;
}
