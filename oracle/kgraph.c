/*
 * kgraph.c -- CPU restatement of the KGraph plugin path (BASELINE config C5).
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).
 *
 * Follows (reference = /root/reference):
 *   - src/thirdparty/kgraph/kgraph.cpp:46-68     GenRandom (distinct random ids)
 *   - src/thirdparty/kgraph/kgraph.cpp:180-205   UpdateKnnList (sorted insert, spare slot addr[K])
 *   - src/thirdparty/kgraph/kgraph.cpp:411-552   KGraphImpl::search (greedy pool expansion)
 *   - src/thirdparty/kgraph/kgraph.cpp:660-700   reverse(-1) + re-rank + unique;  :572-579 prune1
 *   - src/thirdparty/kgraph/kgraph.cpp:703-997   KGraphConstructor: init / join / update (NN-descent)
 *   - src/utils/matcher_kgraph.h:42-104,138-153,204-251  oracles (metric = OpenMVG L2<float>), Build, SearchNeighbours
 *   - src/R3DComputeMatches.cpp:808-902          kgraph_match loop + presets
 *
 * Two index builders feed the SAME search restatement:
 *   orc_kgraph_build_nndescent  the reference's NN-descent, run on one thread.  The reference build is
 *                               thread-count and lock-order dependent and seeds std::mt19937 per
 *                               OpenMP thread, so no run of it is reproducible; this restatement draws from the
 *                               counter-based generator below instead.  It provides the RECALL BASELINE.
 *   orc_kgraph_build_exact      the deterministic index the HIP path builds (DESIGN.md "ANN"): exact K nearest
 *                               neighbours of every row (ties -> lowest row), completed with all reverse edges,
 *                               every list ordered by (distance, id), unique, cut to the `cap` closest.
 *                               GPU parity against this model is bit-exact.
 * Deliberate deviations from kgraph.cpp: (1) the local join skips i == j (the reference can insert a node into
 * its own pool when it is both a new forward and an old reverse neighbour; a self edge never helps a search and
 * trips its own "distance is unstable" check); (2) recall on the control points is evaluated over the valid
 * part of the pool only (the reference walks the uninitialised tail as well).
 */
#include "r3d_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t id; float dist; uint8_t flag; } kg_nb;
typedef struct { uint32_t id; float dist; uint8_t flag; uint32_t m, M; } kg_nbx;

struct orc_kgraph {
    uint32_t n;
    uint64_t* off;     /* n + 1 */
    uint32_t* ids;
    float*    dist;
};

/* ---------------------------------------------------------------- small helpers */

static uint64_t kg_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
static uint32_t kg_rand(uint64_t* state)
{
    *state += 0x9E3779B97F4A7C15ULL;
    return (uint32_t)(kg_mix64(*state) >> 32);
}

typedef struct { uint32_t* v; uint32_t n, cap; } uvec;
static void uv_push(uvec* u, uint32_t x)
{
    if (u->n == u->cap) { u->cap = u->cap ? u->cap * 2 : 16; u->v = (uint32_t*)realloc(u->v, sizeof(uint32_t) * u->cap); }
    u->v[u->n++] = x;
}

static int cmp_u32(const void* a, const void* b)
{
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* kgraph.cpp:46-68 */
static void gen_random(uint64_t* rng, uint32_t* addr, uint32_t size, uint32_t N)
{
    if (N == size) { for (uint32_t i = 0; i < size; ++i) addr[i] = i; return; }
    for (uint32_t i = 0; i < size; ++i) addr[i] = kg_rand(rng) % (N - size);
    qsort(addr, size, sizeof(uint32_t), cmp_u32);
    for (uint32_t i = 1; i < size; ++i) if (addr[i] <= addr[i - 1]) addr[i] = addr[i - 1] + 1;
    const uint32_t off = kg_rand(rng) % N;
    for (uint32_t i = 0; i < size; ++i) addr[i] = (addr[i] + off) % N;
}

/* kgraph.cpp:180-205: addr[0..K) sorted, addr[K] spare.  Returns the insert position (may be K), or K + 1 when
 * an entry of the same distance already carries this id. */
#define KG_UPDATE(T)                                                                     \
    static unsigned kg_update_##T(T* addr, unsigned K, T nn)                             \
    {                                                                                    \
        unsigned i = K, j;                                                               \
        while (i > 0) { j = i - 1; if (addr[j].dist <= nn.dist) break; i = j; }          \
        unsigned l = i;                                                                  \
        while (l > 0) { j = l - 1; if (addr[j].dist < nn.dist) break; if (addr[j].id == nn.id) return K + 1; l = j; } \
        j = K;                                                                           \
        while (j > i) { addr[j] = addr[j - 1]; --j; }                                    \
        addr[i] = nn;                                                                    \
        return i;                                                                        \
    }
KG_UPDATE(kg_nb)
KG_UPDATE(kg_nbx)

static int cmp_nb_dist_id(const void* a, const void* b)
{
    const kg_nb* x = (const kg_nb*)a; const kg_nb* y = (const kg_nb*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
static int cmp_nb_dist(const void* a, const void* b)      /* Neighbor::operator<, made total with the id */
{
    return cmp_nb_dist_id(a, b);
}

static orc_kgraph* graph_from_lists(uint32_t n, kg_nb** lists, const uint32_t* len)
{
    orc_kgraph* g = (orc_kgraph*)calloc(1, sizeof(orc_kgraph));
    g->n = n;
    g->off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) { g->off[i] = tot; tot += len[i]; }
    g->off[n] = tot;
    g->ids = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(tot ? tot : 1));
    g->dist = (float*)malloc(sizeof(float) * (size_t)(tot ? tot : 1));
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t k = 0; k < len[i]; ++k) { g->ids[g->off[i] + k] = lists[i][k].id; g->dist[g->off[i] + k] = lists[i][k].dist; }
    return g;
}

/* reverse(-1) (kgraph.cpp:660-700): every node keeps its first M forward edges and receives the reverse of every
 * kept edge; lists are re-ranked by distance and made unique.  `cap` = 0 keeps everything (the reference); the
 * HIP design keeps the `cap` closest. */
static orc_kgraph* complete_with_reverse(uint32_t n, kg_nb** fwd, const uint32_t* M, uint32_t cap)
{
    uint32_t* len = (uint32_t*)calloc(n, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) { len[i] += M[i]; for (uint32_t k = 0; k < M[i]; ++k) len[fwd[i][k].id]++; }
    kg_nb** ng = (kg_nb**)malloc(sizeof(kg_nb*) * n);
    for (uint32_t i = 0; i < n; ++i) { ng[i] = (kg_nb*)malloc(sizeof(kg_nb) * (len[i] ? len[i] : 1)); len[i] = 0; }
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t k = 0; k < M[i]; ++k) {
            kg_nb e = fwd[i][k], re = e;
            re.id = i;
            ng[i][len[i]++] = e;
            ng[e.id][len[e.id]++] = re;
        }
    for (uint32_t i = 0; i < n; ++i) {
        qsort(ng[i], len[i], sizeof(kg_nb), cmp_nb_dist_id);
        uint32_t w = 0;
        for (uint32_t k = 0; k < len[i]; ++k) if (w == 0 || ng[i][w - 1].id != ng[i][k].id) ng[i][w++] = ng[i][k];
        len[i] = (cap && w > cap) ? cap : w;
    }
    orc_kgraph* g = graph_from_lists(n, ng, len);
    for (uint32_t i = 0; i < n; ++i) free(ng[i]);
    free(ng); free(len);
    return g;
}

/* ---------------------------------------------------------------- the index the HIP path builds */

orc_kgraph* orc_kgraph_build_exact(const float* data, uint32_t n, uint32_t dim, uint32_t K, uint32_t cap)
{
    if (n < 2) return NULL;
    if (K > n - 1) K = n - 1;
    kg_nb** fwd = (kg_nb**)malloc(sizeof(kg_nb*) * n);
    uint32_t* M = (uint32_t*)malloc(sizeof(uint32_t) * n);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        kg_nb* l = (kg_nb*)malloc(sizeof(kg_nb) * (K + 1));
        uint32_t L = 0;
        for (uint32_t j = 0; j < n; ++j) {
            if (j == (uint32_t)i) continue;
            kg_nb nn = { j, orc_l2sq_f32(data + (size_t)j * dim, data + (size_t)i * dim, dim), 1 };
            kg_update_kg_nb(l, L, nn);           /* later equal distances go behind: lowest row first */
            if (L < K) ++L;
        }
        fwd[i] = l; M[i] = L;
    }
    orc_kgraph* g = complete_with_reverse(n, fwd, M, cap);
    for (uint32_t i = 0; i < n; ++i) free(fwd[i]);
    free(fwd); free(M);
    return g;
}

/* ---------------------------------------------------------------- NN-descent (kgraph.cpp:703-997) */

typedef struct {
    kg_nb* pool; uint32_t pool_size;      /* L + 1 entries */
    uint32_t L, M;
    float radius, radiusM;
    int found;
    uvec nn_new, nn_old, rnn_new, rnn_old;
} nhood_t;

/* Nhood::parallel_try_insert (kgraph.cpp:720-734) */
static unsigned nh_try_insert(nhood_t* h, uint32_t id, float dist)
{
    if (dist > h->radius) return h->pool_size;
    kg_nb nn = { id, dist, 1 };
    const unsigned l = kg_update_kg_nb(h->pool, h->L, nn);
    if (l <= h->L) {
        if (h->L + 1 < h->pool_size) ++h->L;
        else h->radius = h->pool[h->L - 1].dist;
    }
    return l;
}

static void shuffle_cut(uvec* u, uint32_t R, uint64_t* rng)
{
    if (!R || u->n <= R) return;
    for (uint32_t i = u->n - 1; i > 0; --i) { const uint32_t j = kg_rand(rng) % (i + 1); const uint32_t t = u->v[i]; u->v[i] = u->v[j]; u->v[j] = t; }
    u->n = R;
}

orc_kgraph* orc_kgraph_build_nndescent(const float* data, uint32_t n, uint32_t dim,
                                       uint32_t K, uint32_t L, uint32_t S, uint32_t R, uint32_t iterations,
                                       float recall_target, float delta_target, uint32_t n_controls, uint32_t seed,
                                       float* info /* [4]: iterations, recall, delta, cost; may be NULL */)
{
    if (n <= K) return NULL;                                     /* "K larger than dataset size" */
    if (n_controls > n) n_controls = n;
    if (n <= L) L = n - 1;
    if (n <= S) S = n - 1;
    uint64_t rng = kg_mix64(seed);

    /* GenerateControl: C random rows with their exact K-NN (LinearSearch, self excluded) */
    uint32_t* ctl_id = (uint32_t*)malloc(sizeof(uint32_t) * (n_controls ? n_controls : 1));
    float* ctl_d = (float*)malloc(sizeof(float) * (size_t)(n_controls ? n_controls : 1) * K);
    {
        uint32_t* index = (uint32_t*)malloc(sizeof(uint32_t) * n);
        for (uint32_t i = 0; i < n; ++i) index[i] = i;
        for (uint32_t i = n - 1; i > 0; --i) { const uint32_t j = kg_rand(&rng) % (i + 1); const uint32_t t = index[i]; index[i] = index[j]; index[j] = t; }
        kg_nb* l = (kg_nb*)malloc(sizeof(kg_nb) * (K + 1));
        for (uint32_t c = 0; c < n_controls; ++c) {
            ctl_id[c] = index[c];
            uint32_t k = 0;
            for (uint32_t j = 0; j < n; ++j) {
                if (j == index[c]) continue;
                kg_nb nn = { j, orc_l2sq_f32(data + (size_t)index[c] * dim, data + (size_t)j * dim, dim), 1 };
                kg_update_kg_nb(l, k, nn);
                if (k < K) ++k;
            }
            for (uint32_t q = 0; q < K; ++q) ctl_d[(size_t)c * K + q] = l[q].dist;
        }
        free(l); free(index);
    }

    /* init (kgraph.cpp:757-790) */
    nhood_t* nh = (nhood_t*)calloc(n, sizeof(nhood_t));
    uint32_t* rnd = (uint32_t*)malloc(sizeof(uint32_t) * (S + 1));
    for (uint32_t i = 0; i < n; ++i) {
        nhood_t* h = &nh[i];
        h->pool_size = L + 1;
        h->pool = (kg_nb*)calloc(h->pool_size, sizeof(kg_nb));
        h->radius = FLT_MAX;
        h->nn_new.v = (uint32_t*)malloc(sizeof(uint32_t) * S * 2); h->nn_new.cap = h->nn_new.n = S * 2;
        gen_random(&rng, h->nn_new.v, h->nn_new.n, n);
        gen_random(&rng, rnd, S + 1, n);
        h->L = S; h->M = S;
        uint32_t r = 0;
        for (uint32_t l = 0; l < h->L; ++l) {
            if (rnd[r] == i) ++r;
            h->pool[l].id = rnd[r++];
            h->pool[l].dist = orc_l2sq_f32(data + (size_t)h->pool[l].id * dim, data + (size_t)i * dim, dim);
            h->pool[l].flag = 1;
        }
        qsort(h->pool, h->L, sizeof(kg_nb), cmp_nb_dist);
    }
    free(rnd);

    const float total = (float)n * (float)(n - 1) / 2;
    double n_comps = 0;
    float it_done = 0, recall = 0, delta = 1.0f;
    for (uint32_t it = 0; iterations == 0 || it < iterations; ++it) {
        it_done += 1;
        /* join (kgraph.cpp:791-809 with Nhood::join :737-748) */
        for (uint32_t v = 0; v < n; ++v) {
            nhood_t* h = &nh[v];
            unsigned uu = 0;
            for (uint32_t a = 0; a < h->nn_new.n; ++a) {
                const uint32_t i = h->nn_new.v[a];
                for (int pass = 0; pass < 2; ++pass) {
                    const uvec* other = pass == 0 ? &h->nn_new : &h->nn_old;
                    for (uint32_t b = 0; b < other->n; ++b) {
                        const uint32_t j = other->v[b];
                        if (pass == 0 ? !(i < j) : (i == j)) continue;
                        const float d = orc_l2sq_f32(data + (size_t)i * dim, data + (size_t)j * dim, dim);
                        n_comps += 1;
                        const unsigned r = nh_try_insert(&nh[i], j, d);
                        if (r < K) ++uu;
                        nh_try_insert(&nh[j], i, d);
                        if (r < K) ++uu;
                    }
                }
            }
            h->found = uu > 0;
        }
        /* statistics (kgraph.cpp:921-963) */
        {
            double sd = 0;
            for (uint32_t v = 0; v < n; ++v) {
                unsigned c = 0;
                const uint32_t N = K < nh[v].pool_size ? K : nh[v].pool_size;
                for (uint32_t k = 0; k < N; ++k) c += (k < nh[v].L && nh[v].pool[k].flag);
                sd += (float)c / K;
            }
            delta = (float)(sd / n);
            double sr = 0;
            for (uint32_t c = 0; c < n_controls; ++c) {
                const nhood_t* h = &nh[ctl_id[c]];
                unsigned found = 0, np = 0, nk = 0;
                while (np < h->L && nk < K) {
                    const float kd = ctl_d[(size_t)c * K + nk];
                    if (kd < h->pool[np].dist) ++nk;
                    else if (kd == h->pool[np].dist) { ++found; ++nk; ++np; }
                    else ++np;                                    /* (the reference throws "distance is unstable") */
                }
                sr += (float)found / K;
            }
            recall = n_controls ? (float)(sr / n_controls) : 0.f;
        }
        if (delta <= delta_target) break;
        if (recall >= recall_target) break;
        /* update (kgraph.cpp:810-882) */
        for (uint32_t v = 0; v < n; ++v) {
            nhood_t* h = &nh[v];
            h->nn_new.n = h->nn_old.n = h->rnn_new.n = h->rnn_old.n = 0;
            h->radius = (h->L + 1 == h->pool_size) ? h->pool[h->pool_size - 1].dist : FLT_MAX;   /* pool.back() */
            if (h->L + 1 == h->pool_size && h->radius < h->pool[h->L - 1].dist) h->radius = h->pool[h->L - 1].dist;
        }
        for (uint32_t v = 0; v < n; ++v) {
            nhood_t* h = &nh[v];
            if (h->found) {
                const uint32_t maxl = (h->M + S < h->L) ? h->M + S : h->L;
                uint32_t c = 0, l = 0;
                while (l < maxl && c < S) { if (h->pool[l].flag) ++c; ++l; }
                h->M = l;
            }
            h->radiusM = h->pool[h->M - 1].dist;
        }
        for (uint32_t v = 0; v < n; ++v) {
            nhood_t* h = &nh[v];
            for (uint32_t l = 0; l < h->M; ++l) {
                kg_nb* nn = &h->pool[l];
                nhood_t* o = &nh[nn->id];
                if (nn->flag) {
                    uv_push(&h->nn_new, nn->id);
                    if (nn->dist > o->radiusM) uv_push(&o->rnn_new, v);
                    nn->flag = 0;
                } else {
                    uv_push(&h->nn_old, nn->id);
                    if (nn->dist > o->radiusM) uv_push(&o->rnn_old, v);
                }
            }
        }
        for (uint32_t v = 0; v < n; ++v) {
            nhood_t* h = &nh[v];
            shuffle_cut(&h->rnn_new, R, &rng);
            for (uint32_t k = 0; k < h->rnn_new.n; ++k) uv_push(&h->nn_new, h->rnn_new.v[k]);
            shuffle_cut(&h->rnn_old, R, &rng);
            for (uint32_t k = 0; k < h->rnn_old.n; ++k) uv_push(&h->nn_old, h->rnn_old.v[k]);
        }
    }
    if (info) { info[0] = it_done; info[1] = recall; info[2] = delta; info[3] = (float)(n_comps / total); }

    /* graph[n] = pool[0..L), M[n] = nhood.M  (kgraph.cpp:973-984), then reverse(-1) and prune(1) */
    kg_nb** fwd = (kg_nb**)malloc(sizeof(kg_nb*) * n);
    uint32_t* M = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (uint32_t v = 0; v < n; ++v) { fwd[v] = nh[v].pool; M[v] = nh[v].M < nh[v].L ? nh[v].M : nh[v].L; }
    orc_kgraph* g = complete_with_reverse(n, fwd, M, 0);
    for (uint32_t v = 0; v < n; ++v) { free(nh[v].pool); free(nh[v].nn_new.v); free(nh[v].nn_old.v); free(nh[v].rnn_new.v); free(nh[v].rnn_old.v); }
    free(nh); free(fwd); free(M); free(ctl_id); free(ctl_d);
    return g;
}

/* ---------------------------------------------------------------- accessors */

void orc_kgraph_free(orc_kgraph* g) { if (g) { free(g->off); free(g->ids); free(g->dist); free(g); } }
uint32_t orc_kgraph_size(const orc_kgraph* g) { return g ? g->n : 0; }
uint64_t orc_kgraph_edges(const orc_kgraph* g) { return g ? g->off[g->n] : 0; }
void orc_kgraph_export(const orc_kgraph* g, uint64_t* off, uint32_t* ids, float* dist)
{
    memcpy(off, g->off, sizeof(uint64_t) * ((size_t)g->n + 1));
    memcpy(ids, g->ids, sizeof(uint32_t) * (size_t)g->off[g->n]);
    if (dist) memcpy(dist, g->dist, sizeof(float) * (size_t)g->off[g->n]);
}

/* ---------------------------------------------------------------- search (kgraph.cpp:411-552) */

/* P distinct start rows of query q of pair (I, J): one per stratum of [0, n), drawn from the counter-based
 * generator shared with AC-RANSAC (the reference: GenRandom on a std::mt19937 seeded 1998 for every query). */
void orc_kgraph_seeds(uint64_t seed, uint32_t I, uint32_t J, uint32_t q, uint32_t n, uint32_t P, uint32_t* out)
{
    for (uint32_t s = 0; s < P; ++s) {
        const uint32_t lo = (uint32_t)(((uint64_t)s * n) / P), hi = (uint32_t)(((uint64_t)(s + 1) * n) / P);
        const uint64_t r = orc_rng_u64(seed ^ 0x6b67726170680000ULL /* "kgraph" */, I, J, q, s);
        out[s] = lo + (uint32_t)(((r >> 32) * (uint64_t)(hi - lo)) >> 32);
    }
}

/* T = 1, init = 0, M = 0 (all neighbours), epsilon = inf: the configuration of ArrayMatcher_kgraph::SearchNeighbours.
 * `min_rows`: indices smaller than this are scanned exhaustively (the reference: only when P >= n). */
uint32_t orc_kgraph_search(const orc_kgraph* g, const float* data, uint32_t dim, const float* query,
                           uint32_t K, uint32_t P, uint32_t S, const uint32_t* seeds, uint32_t min_rows,
                           uint32_t* ids, float* dists, uint32_t* n_comps_out)
{
    const uint32_t n = g->n;
    uint32_t n_comps = 0;
    if (P >= n || n < min_rows) {                      /* SearchOracle::search: linear scan */
        kg_nb* l = (kg_nb*)malloc(sizeof(kg_nb) * (K + 1));
        uint32_t L = 0;
        for (uint32_t k = 0; k < n; ++k) {
            kg_nb nn = { k, orc_l2sq_f32(data + (size_t)k * dim, query, dim), 1 };
            kg_update_kg_nb(l, L, nn);
            if (L < K) ++L;
        }
        for (uint32_t k = 0; k < L; ++k) { ids[k] = l[k].id; dists[k] = l[k].dist; }
        free(l);
        if (n_comps_out) *n_comps_out = n;
        return L;
    }
    const uint32_t size = K + P + 1;
    kg_nbx* knn = (kg_nbx*)calloc(size, sizeof(kg_nbx));
    uint8_t* flags = (uint8_t*)calloc(n, 1);
    uint32_t L = 0;
    for (uint32_t s = 0; s < P; ++s) if (!flags[seeds[s]]) knn[L++].id = seeds[s];
    for (uint32_t k = 0; k < L; ++k) {
        kg_nbx* e = &knn[k];
        flags[e->id] = 1;
        e->flag = 1;
        e->dist = orc_l2sq_f32(data + (size_t)e->id * dim, query, dim);
        e->m = 0;
        e->M = (uint32_t)(g->off[e->id + 1] - g->off[e->id]);
    }
    /* sort(knn, knn + L) by distance; seeds are distinct, ties keep seed order (insertion sort = stable) */
    for (uint32_t a = 1; a < L; ++a) { kg_nbx t = knn[a]; uint32_t b = a; while (b > 0 && knn[b - 1].dist > t.dist) { knn[b] = knn[b - 1]; --b; } knn[b] = t; }
    uint32_t k = 0;
    while (k < L) {
        kg_nbx* e = &knn[k];
        if (!e->flag) { ++k; continue; }
        const uint32_t beginM = e->m;
        uint32_t endM = beginM + S;
        if (endM > e->M) { e->flag = 0; endM = e->M; }
        e->m = endM;
        const uint32_t* nb = g->ids + g->off[e->id];
        for (uint32_t m = beginM; m < endM; ++m) {
            const uint32_t id = nb[m];
            if (flags[id]) continue;
            flags[id] = 1;
            ++n_comps;
            kg_nbx nn = { id, orc_l2sq_f32(data + (size_t)id * dim, query, dim), 1, 0, 0 };
            const unsigned r = kg_update_kg_nbx(knn, L, nn);
            if (L + 1 < size) ++L;
            if (r < L) {
                knn[r].M = (uint32_t)(g->off[id + 1] - g->off[id]);
                if (r < k) k = r;
            }
        }
    }
    if (L > K) L = K;
    for (uint32_t q = 0; q < L; ++q) { ids[q] = knn[q].id; dists[q] = knn[q].dist; }
    free(knn); free(flags);
    if (n_comps_out) *n_comps_out = n_comps + P;
    return L;
}

/* 2-NN of every query row (SearchNeighbours with NN = 2, matcher_kgraph.h:204-251).  Returns -1 like the brute
 * force matcher when the call cannot produce two neighbours. */
int orc_kgraph_knn2(const orc_kgraph* g, const float* data, uint32_t dim, const float* query, uint32_t nq,
                    uint32_t P, uint32_t S, uint64_t seed, uint32_t I, uint32_t J, uint32_t min_rows,
                    int32_t* idx, float* dist, uint64_t* n_comps)
{
    if (!g || g->n < 2 || nq < 1 || P < 2 || P > 61) return -1;
    uint64_t comps = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+:comps)
    for (int64_t q = 0; q < (int64_t)nq; ++q) {
        uint32_t seeds[64], ids[2] = {0, 0}, nc = 0;
        float d[2] = {0, 0};
        orc_kgraph_seeds(seed, I, J, (uint32_t)q, g->n, P, seeds);
        orc_kgraph_search(g, data, dim, query + (size_t)q * dim, 2, P, S, seeds, min_rows, ids, d, &nc);
        idx[2 * q] = (int32_t)ids[0]; idx[2 * q + 1] = (int32_t)ids[1];
        dist[2 * q] = d[0]; dist[2 * q + 1] = d[1];
        comps += nc;
    }
    if (n_comps) *n_comps = comps;
    return 0;
}

/* ---------------------------------------------------------------- kgraph_match (src/R3DComputeMatches.cpp:808-902) */

int64_t orc_match_collection_kgraph(int n_images, const float* const* desc, const int* n_rows,
                                    const float* const* xy, int dim, const uint32_t* pairs, int64_t n_pairs,
                                    float dist_ratio, int builder, uint32_t K, uint32_t L, float recall, uint32_t cap,
                                    uint32_t P, uint32_t S, uint64_t seed, uint32_t min_rows,
                                    uint32_t* counts, orc_match* out, int64_t out_cap, uint64_t* n_comps)
{
    orc_match** res = (orc_match**)calloc((size_t)n_pairs, sizeof(orc_match*));
    memset(counts, 0, sizeof(uint32_t) * (size_t)n_pairs);
    uint64_t comps = 0;
    /* one index per first image, built when its first pair comes up (the reference builds inside RegionsMatcherT's
     * constructor, once per I); pairs of one I are then searched J by J */
    for (int I = 0; I < n_images; ++I) {
        orc_kgraph* g = NULL;
        for (int64_t p = 0; p < n_pairs; ++p) {
            if ((int)pairs[2 * p] != I) continue;
            const uint32_t J = pairs[2 * p + 1];
            if (n_rows[I] < 2 || n_rows[J] < 1) continue;
            if (!g) {
                g = builder == 0 ? orc_kgraph_build_exact(desc[I], (uint32_t)n_rows[I], (uint32_t)dim, K, cap)
                                 : orc_kgraph_build_nndescent(desc[I], (uint32_t)n_rows[I], (uint32_t)dim, K, L, 10, 100, 30,
                                                              recall, 0.002f, 100, 1998, NULL);
                if (!g) break;
            }
            int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n_rows[J]);
            float* dist = (float*)malloc(sizeof(float) * 2 * (size_t)n_rows[J]);
            uint64_t c = 0;
            if (orc_kgraph_knn2(g, desc[I], (uint32_t)dim, desc[J], (uint32_t)n_rows[J], P, S, seed, (uint32_t)I, J, min_rows,
                                idx, dist, &c) == 0) {
                orc_match* tmp = (orc_match*)malloc(sizeof(orc_match) * (size_t)n_rows[J]);
                const int m = orc_ratio_dedup_f32(idx, dist, n_rows[J], xy ? xy[I] : NULL, xy ? xy[J] : NULL, dist_ratio, 1, tmp);
                if (m > 0) { res[p] = tmp; counts[p] = (uint32_t)m; } else free(tmp);
                comps += c;
            }
            free(idx); free(dist);
        }
        orc_kgraph_free(g);
    }
    if (n_comps) *n_comps = comps;
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
