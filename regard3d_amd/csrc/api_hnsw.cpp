// api_hnsw.cpp -- part of the host side of libr3dm.so: the HNSW plugin path of the C ABI (include/r3dm.h).
//
// Replaces hnsw_match (/root/reference/src/R3DComputeMatches.cpp:497-593): per first view I an HNSW index over its descriptors
// (ArrayMatcher_hnsw::Build, src/utils/matcher_hnsw.h:53-83), per query row of J hnswlib's searchKnn(row, 2) with setEf(ef)
// (SearchNeighbours :150-190), then the ratio test / de-duplication / pair rules every arm shares.  The search is hnswlib's, step
// for step (kernels_hnsw.hip); the index is built in one batch on the device instead of row by row (DESIGN.md "HNSW").
// There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"

#include <cmath>
#include <random>

extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);

// src/R3DComputeMatches.cpp:533-565: (efConstruction, ef, M) = 112 / 5 / 5, 112 / 10 / 15, 100 / 15 / 19
extern "C" int r3dm_hnsw_preset(int preset, r3dm_hnsw_params* out)
{
    if (!out) return R3DM_ERR_INVALID;
    r3dm_hnsw_params k{};
    k.seed = 100;                                            // hnswalg.h:55 random_seed default
    switch (preset) {
        case 0:  k.ef_construction = 112; k.ef = 5;  k.M = 5;  break;
        case 1:  k.ef_construction = 112; k.ef = 10; k.M = 15; break;
        default: k.ef_construction = 100; k.ef = 15; k.M = 19; break;
    }
    *out = k;
    return R3DM_OK;
}

static int check_hnsw_params(r3dm_ctx* c, const r3dm_hnsw_params* hp)
{
    if (!hp) return R3DM_ERR_INVALID;
    if (hp->M < 2 || hp->M > 32 || hp->ef < 1 || hp->ef > 512) { c->err = "hnsw parameters out of range (M 2..32, ef 1..512)"; return R3DM_ERR_INVALID; }
    return R3DM_OK;
}

static bool hnsw_dim_ok(uint32_t dim) { return dim == 64 || dim == 128 || dim == 144 || dim == 256; }

// HierarchicalNSW::getRandomLevel (hnswalg.h:146-151) for rows 0 .. n-1 in insertion order: the same engine, distribution and
// expression, from the same standard library the reference is built with
static void hnsw_levels(uint32_t n, uint32_t M, uint32_t seed, std::vector<int32_t>& out)
{
    std::default_random_engine gen(seed);
    std::uniform_real_distribution<double> distribution(0.0, 1.0);
    const double mult = 1.0 / std::log(1.0 * (double)M);
    out.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const double r = -std::log(distribution(gen)) * mult;
        out[i] = (int32_t)r;
    }
}

// builds the HNSW index of every listed slot that does not hold one for this (M, seed)
static int ensure_hnsw_indices(r3dm_ctx* c, std::vector<uint32_t> slots, const r3dm_hnsw_params& hp)
{
    { const int rcl = ensure_layouts(c, slots, kLayRows); if (rcl != R3DM_OK) return rcl; }      // the index is built from (and searched on) row-major rows

    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    std::vector<uint32_t> todo;
    for (uint32_t s : slots) if (c->imgs[s]->hnsw_M != hp.M || c->imgs[s]->hnsw_seed != hp.seed) todo.push_back(s);
    if (todo.empty()) return R3DM_OK;
    // layer-0 candidates: the exact 32-NN graph completed with reverse edges (the KGraph path's index; its own timing is added below)
    const r3dm_stats before = c->stats;
    int rc = ensure_ann_indices(c, todo, kAnnMaxK);
    if (rc != R3DM_OK) return rc;
    const double ms_knn = c->stats.ms_ann_build - before.ms_ann_build;
    c->stats.n_ann_built = before.n_ann_built;

    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    size_t start = 0;
    std::vector<int32_t> levels;
    while (start < todo.size()) {
        const uint32_t dim = c->imgs[todo[start]]->dim;
        size_t end = start;
        while (end < todo.size() && end - start < 1024 && c->imgs[todo[end]]->dim == dim) ++end;
        std::vector<uint32_t> aux;                            // per job: up_off[n + 1] | up_node | up_level | members | mem_off
        std::vector<HnswBuildJob> jobs;
        struct Off { size_t up_off, up_node, up_level, members, mem_off; };
        std::vector<Off> offs;
        uint32_t max_items = 0;
        for (size_t k = start; k < end; ++k) {
            HostImage& h = *c->imgs[todo[k]];
            hnsw_levels(h.n, hp.M, hp.seed, levels);
            int32_t maxlevel = levels[0], enter = 0;
            for (uint32_t i = 1; i < h.n; ++i) if (levels[i] > maxlevel) { maxlevel = levels[i]; enter = (int32_t)i; }
            Off o{};
            o.up_off = aux.size();
            uint32_t rows = 0;
            for (uint32_t i = 0; i < h.n; ++i) { aux.push_back(rows); rows += (uint32_t)levels[i]; }
            aux.push_back(rows);
            o.up_node = aux.size();
            for (uint32_t i = 0; i < h.n; ++i) for (int32_t L = 1; L <= levels[i]; ++L) aux.push_back(i);
            o.up_level = aux.size();
            for (uint32_t i = 0; i < h.n; ++i) for (int32_t L = 1; L <= levels[i]; ++L) aux.push_back((uint32_t)L);
            o.members = aux.size();
            std::vector<uint32_t> mem_off{0};
            for (int32_t L = 1; L <= maxlevel; ++L) {
                for (uint32_t i = 0; i < h.n; ++i) if (levels[i] >= L) aux.push_back(i);
                mem_off.push_back((uint32_t)(aux.size() - o.members));
            }
            o.mem_off = aux.size();
            aux.insert(aux.end(), mem_off.begin(), mem_off.end());
            offs.push_back(o);
            R3DM_HIP(c, h.hnsw_l0.ensure((size_t)h.n * (1 + 2 * hp.M) * 4));
            R3DM_HIP(c, h.hnsw_up_off.ensure(((size_t)h.n + 1) * 4));
            R3DM_HIP(c, h.hnsw_up.ensure((size_t)std::max(rows, 1u) * (1 + hp.M) * 4));
            h.hnsw_up_rows = rows; h.hnsw_enter = enter; h.hnsw_maxlevel = maxlevel;
            max_items = std::max(max_items, h.n + rows);
        }
        R3DM_HIP(c, c->h_aux.ensure(aux.size() * 4 + 64));
        R3DM_HIP(c, hipMemcpyAsync(c->h_aux.p, aux.data(), aux.size() * 4, hipMemcpyHostToDevice, c->stream));
        for (size_t k = start; k < end; ++k) {
            HostImage& h = *c->imgs[todo[k]];
            const Off& o = offs[k - start];
            const uint32_t* A = c->h_aux.as<uint32_t>();
            HnswBuildJob j{};
            j.rows = h.rows.as<float>(); j.adj = h.ann_adj.as<uint32_t>(); j.adj_deg = h.ann_deg.as<uint32_t>();
            j.up_node = A + o.up_node; j.up_level = A + o.up_level; j.members = A + o.members; j.mem_off = A + o.mem_off;
            j.l0 = h.hnsw_l0.as<int32_t>(); j.up = h.hnsw_up.as<int32_t>();
            j.n = h.n; j.dim = h.dim; j.M = hp.M; j.up_rows = h.hnsw_up_rows;
            jobs.push_back(j);
            R3DM_HIP(c, hipMemcpyAsync(h.hnsw_up_off.p, A + o.up_off, ((size_t)h.n + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
        }
        R3DM_HIP(c, c->h_jobs.ensure(jobs.size() * sizeof(HnswBuildJob)));
        R3DM_HIP(c, hipMemcpyAsync(c->h_jobs.p, jobs.data(), jobs.size() * sizeof(HnswBuildJob), hipMemcpyHostToDevice, c->stream));
        HnswBuildParams bp{};
        bp.jobs = c->h_jobs.as<HnswBuildJob>();
        hipError_t e = launch_hnsw_link(c->stream, bp, (uint32_t)jobs.size(), max_items, dim);
        if (e == hipErrorInvalidValue) { c->err = "no HNSW kernel for this descriptor length (64 / 128 / 144 / 256)"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, e);
        R3DM_HIP(c, hipStreamSynchronize(c->stream));          // aux / jobs are host temporaries and h_aux is reused
        for (size_t k = start; k < end; ++k) { c->imgs[todo[k]]->hnsw_M = hp.M; c->imgs[todo[k]]->hnsw_seed = hp.seed; }
        start = end;
    }
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_build = before.ms_ann_build + ms_knn + ms;
    c->stats.n_ann_built += todo.size();
    return R3DM_OK;
}

static HnswView view_of(const HostImage& h)
{
    HnswView v{};
    v.rows = h.rows.as<float>(); v.rows8 = h.compact_ready ? h.ann_rows8.as<uint8_t>() : nullptr; v.l0 = h.hnsw_l0.as<int32_t>(); v.up_off = h.hnsw_up_off.as<int32_t>(); v.up = h.hnsw_up.as<int32_t>();
    v.n = h.n; v.dim = h.dim; v.M = h.hnsw_M; v.enter = h.hnsw_enter; v.maxlevel = h.hnsw_maxlevel;
    return v;
}

// searchKnn + ratio test over `jobs` (all of one dim; every sI holds an index), results appended to g in job order
static int run_hnsw_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio_R, uint32_t ef, r3dm_graph* g,
                          int32_t* knn_idx_host, float* knn_dist_host)
{
    const uint32_t P = (uint32_t)jobs.size();
    if (P == 0) return R3DM_OK;
    {   // queries and index rows are gathered from the row-major rows
        std::vector<uint32_t> slots;
        for (const PairJob& j : jobs) { slots.push_back(j.sI); slots.push_back(j.sJ); }
        const int rcl = ensure_layouts(c, slots, kLayRows);
        if (rcl != R3DM_OK) return rcl;
    }
    uint32_t max_nJ = 0, max_nI = 0;
    uint64_t n_queries = 0;
    for (const PairJob& j : jobs) {
        max_nI = std::max(max_nI, c->imgs[j.sI]->n);
        max_nJ = std::max(max_nJ, c->imgs[j.sJ]->n);
        n_queries += c->imgs[j.sJ]->n;
    }
    const uint32_t dim = c->imgs[jobs[0].sI]->dim;
    const uint32_t q_stride = std::max<uint32_t>(32, (max_nJ + 31) / 32 * 32);
    const uint32_t sort_cap = std::min<uint32_t>(16384, std::max<uint32_t>(8, next_pow2(q_stride)));
    std::vector<uint2> hp(P);
    std::vector<HnswSearchJob> sj(P);
    bool rows8 = true;                                        // every index view of the batch holds its byte rows (integers 0 .. 255: SIFT bins)
    for (uint32_t p = 0; p < P; ++p) {
        hp[p] = make_uint2(jobs[p].sI, jobs[p].sJ);
        sj[p].ix = view_of(*c->imgs[jobs[p].sI]);
        sj[p].query = c->imgs[jobs[p].sJ]->rows.as<float>();
        sj[p].nq = c->imgs[jobs[p].sJ]->n;
        sj[p].out_base = p * q_stride;
        rows8 = rows8 && sj[p].ix.rows8 != nullptr;
    }
    R3DM_HIP(c, c->d_pairs.ensure(sizeof(uint2) * P));
    R3DM_HIP(c, c->h_jobs.ensure(sizeof(HnswSearchJob) * P));
    R3DM_HIP(c, hipMemcpyAsync(c->d_pairs.p, hp.data(), sizeof(uint2) * P, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->h_jobs.p, sj.data(), sizeof(HnswSearchJob) * P, hipMemcpyHostToDevice, c->stream));
    const uint64_t total_slots = (uint64_t)P * q_stride;
    R3DM_HIP(c, c->d_nn.ensure((size_t)total_slots * 4));
    R3DM_HIP(c, c->d_cnt.ensure(64));
    if (knn_idx_host) {
        R3DM_HIP(c, c->d_knn_idx.ensure((size_t)total_slots * 8));
        R3DM_HIP(c, c->d_knn_dist.ensure((size_t)total_slots * 8));
    }
    HnswSearchParams sp{};
    sp.jobs = c->h_jobs.as<HnswSearchJob>(); sp.n_jobs = P;
    sp.ef = std::max(ef, 2u);                                  // searchKnn: max(ef_, k)
    sp.ratio_R = ratio_R;
    sp.rows8 = rows8 ? 1u : 0u;
    sp.dense_steps = r3dm_dev_knob("R3DM_HNSW_DENSE_STEPS", 0) ? 1u : 0u;                // developer build: the round-3 stepping, for A/B runs
    sp.queries_per_wave = (uint32_t)r3dm_dev_knob("R3DM_HNSW_QW", 0);                    // developer build: 1 / 2 / 4 / 8 queries per wavefront
    sp.nn_idx = c->d_nn.as<uint32_t>();
    sp.knn_idx = knn_idx_host ? c->d_knn_idx.as<int32_t>() : nullptr;
    sp.knn_dist = knn_idx_host ? c->d_knn_dist.as<float>() : nullptr;
    sp.n_comps = reinterpret_cast<unsigned long long*>(c->d_cnt.as<uint32_t>() + 4);
    sp.n_overflow = c->d_cnt.as<uint32_t>() + 2;
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    unsigned long long comps = 0;
    // the candidate heap of a query lives in LDS; a query that outgrows it (rare) makes the whole launch run again with twice the room
    for (uint32_t cand_cap = 256;; cand_cap *= 2) {
        sp.cand_cap = cand_cap;
        R3DM_HIP(c, hipMemsetAsync(c->d_cnt.p, 0, 64, c->stream));
        hipError_t e = launch_hnsw_search(c->stream, sp, max_nJ, max_nI, dim);
        if (e == hipErrorInvalidValue) { c->err = "HNSW search: unsupported descriptor length (64 / 128 / 144 / 256) or the view / candidate heap exceeds the LDS"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, e);
        uint32_t over = 0;
        R3DM_HIP(c, hipMemcpyAsync(&over, sp.n_overflow, 4, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(&comps, sp.n_comps, 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        c->stats.n_hnsw_launches += 1;
        if (over == 0) break;
        c->stats.n_hnsw_retries += 1;
    }
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    const double t_post = now_ms();
    int rc = finalize_batch(c, jobs, q_stride, sort_cap, n_queries, max_nJ, g, knn_idx_host, knn_dist_host);
    if (rc != R3DM_OK) return rc;
    c->stats.ms_wall_match_post += now_ms() - t_post;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_search += ms;
    c->stats.n_ann_dist += comps;
    c->stats.n_ann_rows8 += rows8 ? 1 : 0;
    c->stats.n_match_launches += 1;
    c->stats.n_pairs += P;
    c->stats.n_queries += n_queries;
    return R3DM_OK;
}

static int r3dm_match_pairs_hnsw_impl(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                     const r3dm_hnsw_params* hp, r3dm_graph** out)
{
    if (!c || !out || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    int rc = check_hnsw_params(c, hp);
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipSetDevice(c->device));
    c->stats = r3dm_stats{};
    const double t_call = now_ms();
    std::vector<PairJob> ann_jobs, small_jobs;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        const uint32_t I = pairs_ij[2 * p], J = pairs_ij[2 * p + 1];
        auto a = c->slot_of.find(I), b = c->slot_of.find(J);
        if (a == c->slot_of.end() || b == c->slot_of.end()) { c->err = "pair references an unregistered view"; return R3DM_ERR_INVALID; }
        const HostImage& A = *c->imgs[a->second];
        const HostImage& B = *c->imgs[b->second];
        if (A.n == 0 || B.n == 0 || A.dtype != B.dtype || A.dim != B.dim) continue;
        if (A.dtype == R3DM_BIN || !hnsw_dim_ok(A.dim)) { c->err = "HNSW matching needs F32/U8 descriptors of length 64 / 128 / 144 / 256"; return R3DM_ERR_UNSUPPORTED; }
        if (A.n > (1u << 18)) { c->err = "HNSW matching: more than 262,144 rows in one view"; return R3DM_ERR_UNSUPPORTED; }
        // an index over a handful of rows finds all of them anyway: small views are scanned
        if (A.n < kAnnMinRows) small_jobs.push_back({I, J, a->second, b->second});
        else ann_jobs.push_back({I, J, a->second, b->second});
    }
    auto by_pair = [](const PairJob& x, const PairJob& y) { return x.I != y.I ? x.I < y.I : x.J < y.J; };
    auto same = [](const PairJob& x, const PairJob& y) { return x.I == y.I && x.J == y.J; };
    for (auto* v : {&ann_jobs, &small_jobs}) { std::sort(v->begin(), v->end(), by_pair); v->erase(std::unique(v->begin(), v->end(), same), v->end()); }
    const float R = dist_ratio * dist_ratio;

    r3dm_graph ga, gs;
    ga.offsets.push_back(0); gs.offsets.push_back(0);
    // (r3dm_set_device_graphs) the two part graphs are merged on the host: a device mirror survives that only when one part is the whole
    // result -- then it is built and handed over; with both kinds of pairs present no mirror is built at all (it would be dropped)
    PartMirrorGuard mirror_guard(c, !ann_jobs.empty() && !small_jobs.empty());
    if (!ann_jobs.empty()) {
        std::vector<uint32_t> slots;
        for (const PairJob& j : ann_jobs) slots.push_back(j.sI);
        rc = ensure_hnsw_indices(c, slots, *hp);
        if (rc != R3DM_OK) return rc;
    }
    size_t start = 0;
    while (start < ann_jobs.size()) {
        const uint32_t dim = c->imgs[ann_jobs[start].sI]->dim;
        size_t end = start;
        uint32_t max_n = 0;
        while (end < ann_jobs.size() && end - start < 65535) {
            if (c->imgs[ann_jobs[end].sI]->dim != dim) break;
            const uint32_t mn = std::max(max_n, c->imgs[ann_jobs[end].sJ]->n);
            const uint64_t s = (uint64_t)(end - start + 1) * ((mn + 31) / 32 * 32);
            if (end > start && (s * 4 > (3ull << 30) || s / 4 > kMaxBlocksOf256 - 4096)) break;
            max_n = mn; ++end;
        }
        std::vector<PairJob> batch(ann_jobs.begin() + start, ann_jobs.begin() + end);
        rc = run_hnsw_batch(c, batch, R, hp->ef, &ga, nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    start = 0;
    while (start < small_jobs.size()) {
        size_t end = start;
        const HostImage& F = *c->imgs[small_jobs[start].sI];
        while (end < small_jobs.size() && c->imgs[small_jobs[end].sI]->dtype == F.dtype && c->imgs[small_jobs[end].sI]->dim == F.dim) ++end;
        std::vector<PairJob> batch(small_jobs.begin() + start, small_jobs.begin() + end);
        rc = run_match_batch(c, batch, R, &gs, nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    rc = merge_parts_keep_mirror(ga, gs, out);
    c->stats.ms_wall_match = now_ms() - t_call;
    return rc;
}

extern "C" int r3dm_match_pairs_hnsw(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                     const r3dm_hnsw_params* hp, r3dm_graph** out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_match_pairs_hnsw_impl(c, pairs_ij, n_pairs, dist_ratio, hp, out); });
}

// two private slots: dataset (+ index) and query
static int hnsw_knn2_common(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query, uint32_t dim,
                            const r3dm_hnsw_params* hp, const r3dm_hnsw_arrays* ix, int32_t* out_idx, float* out_dist)
{
    if (!c || !dataset || !query || !out_idx || !out_dist) return R3DM_ERR_INVALID;
    if (n_query < 1 || n_dataset < 2) return R3DM_ERR_INVALID;
    if (!hnsw_dim_ok(dim)) { c->err = "HNSW matching needs descriptors of length 64 / 128 / 144 / 256"; return R3DM_ERR_UNSUPPORTED; }
    if (n_dataset > (1u << 18)) { c->err = "HNSW matching: more than 262,144 rows in one view"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, hipSetDevice(c->device));
    const uint32_t s0 = (uint32_t)c->imgs.size();
    c->imgs.emplace_back(new HostImage());
    c->imgs.emplace_back(new HostImage());
    int rc = stage_into_slot(c, s0, 0, 0, 0, dataset, n_dataset, dim, R3DM_F32, nullptr);
    if (rc == R3DM_OK) rc = stage_into_slot(c, s0 + 1, 1, 0, 0, query, n_query, dim, R3DM_F32, nullptr);
    const r3dm_stats keep = c->stats;
    if (rc == R3DM_OK) {
        HostImage& h = *c->imgs[s0];
        if (ix) {                                              // an index handed over as arrays (e.g. written by hnswlib itself)
            const uint32_t M = ix->M;
            bool ok = M >= 2 && M <= 32 && ix->links0 && ix->up_off && ix->enterpoint >= 0 && (uint32_t)ix->enterpoint < n_dataset &&
                      ix->maxlevel >= 0 && (ix->up_rows == 0 || ix->up_links);
            for (uint32_t i = 0; ok && i < n_dataset; ++i) {
                const int32_t* l = ix->links0 + (size_t)i * (1 + 2 * M);
                ok = l[0] >= 0 && (uint32_t)l[0] <= 2 * M && ix->up_off[i] >= 0 && ix->up_off[i] <= ix->up_off[i + 1] && (uint32_t)ix->up_off[i + 1] <= ix->up_rows;
                for (int32_t k = 0; ok && k < l[0]; ++k) ok = l[1 + k] >= 0 && (uint32_t)l[1 + k] < n_dataset;
            }
            for (uint32_t r = 0; ok && r < ix->up_rows; ++r) {
                const int32_t* l = ix->up_links + (size_t)r * (1 + M);
                ok = l[0] >= 0 && (uint32_t)l[0] <= M;
                for (int32_t k = 0; ok && k < l[0]; ++k) ok = l[1 + k] >= 0 && (uint32_t)l[1 + k] < n_dataset;
            }
            // the descent reads layer L of every row it reaches there: the rows linked on a layer must own that layer
            if (ok && (uint32_t)(ix->up_off[ix->enterpoint + 1] - ix->up_off[ix->enterpoint]) < (uint32_t)ix->maxlevel) ok = false;
            for (uint32_t i = 0; ok && i < n_dataset; ++i)
                for (int32_t L = 1; ok && L <= ix->up_off[i + 1] - ix->up_off[i]; ++L) {
                    const int32_t* l = ix->up_links + ((size_t)ix->up_off[i] + (uint32_t)(L - 1)) * (1 + M);
                    for (int32_t k = 0; ok && k < l[0]; ++k) ok = ix->up_off[l[1 + k] + 1] - ix->up_off[l[1 + k]] >= L;
                }
            if (!ok) { c->err = "r3dm_hnsw_knn2_on_index: malformed index arrays"; rc = R3DM_ERR_INVALID; }
            if (rc == R3DM_OK) {
                hipError_t e = h.hnsw_l0.ensure((size_t)n_dataset * (1 + 2 * M) * 4);
                if (e == hipSuccess) e = h.hnsw_up_off.ensure(((size_t)n_dataset + 1) * 4);
                if (e == hipSuccess) e = h.hnsw_up.ensure((size_t)std::max(ix->up_rows, 1u) * (1 + M) * 4);
                if (e == hipSuccess) e = hipMemcpyAsync(h.hnsw_l0.p, ix->links0, (size_t)n_dataset * (1 + 2 * M) * 4, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(h.hnsw_up_off.p, ix->up_off, ((size_t)n_dataset + 1) * 4, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess && ix->up_rows) e = hipMemcpyAsync(h.hnsw_up.p, ix->up_links, (size_t)ix->up_rows * (1 + M) * 4, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) { c->err = std::string("r3dm_hnsw_knn2_on_index: ") + hipGetErrorString(e); rc = R3DM_ERR_HIP; }
                h.hnsw_M = M; h.hnsw_up_rows = ix->up_rows; h.hnsw_enter = ix->enterpoint; h.hnsw_maxlevel = ix->maxlevel;
            }
        } else rc = ensure_hnsw_indices(c, {s0}, *hp);
        if (rc == R3DM_OK) {
            std::vector<PairJob> jobs{{0, 1, s0, s0 + 1}};
            rc = run_hnsw_batch(c, jobs, 1.0f, hp->ef, nullptr, out_idx, out_dist);
        }
    }
    const uint64_t evals = c->stats.n_ann_dist - keep.n_ann_dist, launches = c->stats.n_hnsw_launches - keep.n_hnsw_launches,
                   retries = c->stats.n_hnsw_retries - keep.n_hnsw_retries;
    const double ms_b = c->stats.ms_ann_build - keep.ms_ann_build, ms_s = c->stats.ms_ann_search - keep.ms_ann_search;
    c->stats = keep;
    c->stats.n_ann_dist = evals; c->stats.n_hnsw_launches = launches; c->stats.n_hnsw_retries = retries;
    c->stats.ms_ann_build = ms_b; c->stats.ms_ann_search = ms_s;
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
}

extern "C" int r3dm_hnsw_knn2(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                              uint32_t dim, const r3dm_hnsw_params* hp, int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c) return R3DM_ERR_INVALID;
        int rc = check_hnsw_params(c, hp);
        if (rc != R3DM_OK) return rc;
        if (n_dataset < kAnnMinRows) { c->err = "r3dm_hnsw_knn2: fewer than 128 rows (such views are scanned: r3dm_knn2)"; return R3DM_ERR_UNSUPPORTED; }
        return hnsw_knn2_common(c, dataset, n_dataset, query, n_query, dim, hp, nullptr, out_idx, out_dist);
    });
}

extern "C" int r3dm_hnsw_knn2_on_index(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, uint32_t dim, const r3dm_hnsw_arrays* ix,
                                       const float* query, uint32_t n_query, uint32_t ef, int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c || !ix) return R3DM_ERR_INVALID;
        r3dm_hnsw_params hp{};
        hp.M = ix->M; hp.ef = ef; hp.seed = 0;
        int rc = check_hnsw_params(c, &hp);
        if (rc != R3DM_OK) return rc;
        return hnsw_knn2_common(c, dataset, n_dataset, query, n_query, dim, &hp, ix, out_idx, out_dist);
    });
}

extern "C" int r3dm_hnsw_index(r3dm_ctx* c, uint32_t view_id, const r3dm_hnsw_params* hp, int32_t* links0, int32_t* up_off,
                               int32_t* up_links, uint32_t up_cap, uint32_t* up_rows, int32_t* enterpoint, int32_t* maxlevel)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c || !links0 || !up_off || !up_rows || !enterpoint || !maxlevel) return R3DM_ERR_INVALID;
        int rc = check_hnsw_params(c, hp);
        if (rc != R3DM_OK) return rc;
        auto it = c->slot_of.find(view_id);
        if (it == c->slot_of.end()) { c->err = "unregistered view"; return R3DM_ERR_INVALID; }
        HostImage& h = *c->imgs[it->second];
        if (h.dtype == R3DM_BIN || !hnsw_dim_ok(h.dim) || h.n < kAnnMinRows) { c->err = "HNSW index needs >= 128 F32/U8 rows of length 64 / 128 / 144 / 256"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, hipSetDevice(c->device));
        rc = ensure_hnsw_indices(c, {it->second}, *hp);
        if (rc != R3DM_OK) return rc;
        *up_rows = h.hnsw_up_rows; *enterpoint = h.hnsw_enter; *maxlevel = h.hnsw_maxlevel;
        if (h.hnsw_up_rows > up_cap || (h.hnsw_up_rows && !up_links)) { c->err = "r3dm_hnsw_index: up_links too small (see *up_rows)"; return R3DM_ERR_INVALID; }
        R3DM_HIP(c, hipMemcpy(links0, h.hnsw_l0.p, (size_t)h.n * (1 + 2 * hp->M) * 4, hipMemcpyDeviceToHost));
        R3DM_HIP(c, hipMemcpy(up_off, h.hnsw_up_off.p, ((size_t)h.n + 1) * 4, hipMemcpyDeviceToHost));
        if (h.hnsw_up_rows) R3DM_HIP(c, hipMemcpy(up_links, h.hnsw_up.p, (size_t)h.hnsw_up_rows * (1 + hp->M) * 4, hipMemcpyDeviceToHost));
        return R3DM_OK;
    });
}
