// api_mrpt.cpp -- part of the host side of libr3dm.so: the MRPT plugin path of the C ABI (include/r3dm.h).
//
// Replaces mrpt_match (/root/reference/src/R3DComputeMatches.cpp:423-491): per first view I an index of random projection trees over
// its descriptors (ArrayMatcher_mrpt::Build, src/utils/matcher_mrpt.h:76-128), per query row of J Mrpt::query(row, 2, votes) with the
// adapter's retry (SearchNeighbours :188-245), then the ratio test on the returned square roots and the rules every arm shares.
// Index and queries run on the device (kernels_mrpt.hip); the only host arithmetic is the random matrix -- a few thousand normal
// deviates drawn once per (dim, n_trees x depth, density, seed) with the host's libm, as the reference draws its own on the host.
// There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"

#include <cmath>

extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);

extern "C" int r3dm_mrpt_preset(r3dm_mrpt_params* out)
{
    if (!out) return R3DM_ERR_INVALID;
    r3dm_mrpt_params k{};
    k.n_trees = 26; k.depth = 6; k.votes = 5;                 // src/R3DComputeMatches.cpp:453-455
    k.density = -1.0f;                                        // Mrpt::grow(n_trees, depth): density_ defaults to -1 = 1 / sqrt(dim) (mrpt.h:84,107-111)
    k.seed = 0;
    *out = k;
    return R3DM_OK;
}

static int check_mrpt_params(r3dm_ctx* c, const r3dm_mrpt_params* mp)
{
    if (!mp) return R3DM_ERR_INVALID;
    if (mp->n_trees < 1 || mp->n_trees > 255 || mp->depth < 1 || mp->depth > 6 || mp->votes < 1 || mp->votes > mp->n_trees || mp->density > 1.0f) {
        c->err = "mrpt parameters out of range (n_trees 1..255, depth 1..6, votes 1..n_trees, density <= 1)";
        return R3DM_ERR_INVALID;
    }
    return R3DM_OK;
}

static bool mrpt_dim_ok(uint32_t dim) { return dim >= 4 && dim <= 512 && (dim & 3u) == 0; }

// ArrayMatcher_mrpt::Build (matcher_mrpt.h:93)
static uint32_t mrpt_depth_for(uint32_t n, uint32_t depth)
{
    int lg = 0;
    while ((2u << lg) <= n && lg < 30) ++lg;
    const int d = std::min<int>((int)depth, lg - 1);
    return (uint32_t)std::max(d, 2);
}
static float mrpt_density_for(const r3dm_mrpt_params& mp, uint32_t dim) { return mp.density > 0.0f ? mp.density : (float)(1.0 / std::sqrt((double)dim)); }

// the counter-based stream of the random vectors (oracle/mrpt.c holds the same formulas)
static uint64_t mr_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
static double mr_uniform(uint64_t seed, uint32_t row, uint32_t col, uint32_t k)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    const uint64_t a = mr_mix64(seed + G * (1ULL + (((uint64_t)row << 32) | (uint64_t)col)));
    const uint64_t b = mr_mix64(a + G * (uint64_t)(k + 1u));
    return (double)(b >> 11) * (1.0 / 9007199254740992.0);
}
static void mrpt_random_matrix(uint32_t n_pool, uint32_t dim, float density, uint64_t seed, std::vector<float>& R)
{
    const double TWO_PI = 6.283185307179586476925286766559;
    R.assign((size_t)n_pool * dim, 0.0f);
    for (uint32_t j = 0; j < n_pool; ++j)
        for (uint32_t c = 0; c < dim; ++c) {
            if (mr_uniform(seed, j, c, 0) > (double)density) continue;          // mrpt.h:1260
            const double u1 = 1.0 - mr_uniform(seed, j, c, 1), u2 = mr_uniform(seed, j, c, 2);
            R[(size_t)j * dim + c] = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(TWO_PI * u2));
        }
}
// count_first_leaf_indices (mrpt.h:1664-1672)
// rows the query kernel may elect for an index view of n rows (run_mrpt_batch sizes the elected list with it)
static uint32_t mrpt_elected_cap(uint32_t n, uint32_t n_trees, uint32_t depth, uint32_t votes)
{
    // rows that can reach the lowest threshold used (votes, then votes - 1): every elected row holds that many of the n_trees x max_leaf votes
    const uint64_t max_leaf = n / (1u << depth) + 1u;
    const uint64_t bound = std::min<uint64_t>(n, (uint64_t)n_trees * max_leaf / std::max<uint32_t>(votes > 1 ? votes - 1 : 1, 1u));
    return (uint32_t)bound + 64u;
}
// LDS one wavefront of mrpt_query_kernel needs for such a view (kernels_mrpt.hip: launch_mrpt_query, the same formula): the vote bytes
// (n) and the elected list do not fit the CU's 160 KB for every view the index itself admits -- with the reference's preset from
// ~115 k rows on, with votes <= 2 much sooner.  Views beyond it are matched exhaustively (recall 1), like views too small for a forest.
static bool mrpt_query_fits_lds(uint32_t n, const r3dm_mrpt_params& mp)
{
    const uint32_t depth = mrpt_depth_for(n, mp.depth);
    const size_t pool_pad = ((size_t)mp.n_trees * depth + 63u) / 64u * 64u;
    const size_t per_wave = (pool_pad * 4 + 256 * 4 + 8 + (size_t)mrpt_elected_cap(n, mp.n_trees, depth, mp.votes) * 4 + (size_t)((n + 3u) / 4u) * 4 + 15) / 16 * 16;
    return per_wave <= 160u * 1024u;
}

static void mrpt_leaf_sizes(uint32_t n, uint32_t level, uint32_t depth, std::vector<int32_t>& out)
{
    if (level == depth) { out.push_back((int32_t)n); return; }
    mrpt_leaf_sizes(n - n / 2, level + 1, depth, out);
    mrpt_leaf_sizes(n / 2, level + 1, depth, out);
}

// builds the MRPT index of every listed slot that does not hold one for these parameters
static int ensure_mrpt_indices(r3dm_ctx* c, std::vector<uint32_t> slots, const r3dm_mrpt_params& mp)
{
    { const int rcl = ensure_layouts(c, slots, kLayRows); if (rcl != R3DM_OK) return rcl; }      // the index is built from (and searched on) row-major rows

    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    std::vector<uint32_t> todo;
    for (uint32_t s : slots) {
        const HostImage& h = *c->imgs[s];
        if (h.mrpt_trees != mp.n_trees || h.mrpt_depth != mrpt_depth_for(h.n, mp.depth) || h.mrpt_density != mrpt_density_for(mp, h.dim) || h.mrpt_seed != mp.seed)
            todo.push_back(s);
    }
    if (todo.empty()) return R3DM_OK;
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    DevBuf proj, keys;
    std::vector<float> R, RT;
    int rc = R3DM_OK;
    for (uint32_t s : todo) {
        HostImage& h = *c->imgs[s];
        const uint32_t depth = mrpt_depth_for(h.n, mp.depth), n_pool = mp.n_trees * depth, n_leaf = 1u << depth;
        const float density = mrpt_density_for(mp, h.dim);
        mrpt_random_matrix(n_pool, h.dim, density, mp.seed, R);
        RT.resize(R.size());
        for (uint32_t j = 0; j < n_pool; ++j) for (uint32_t cc = 0; cc < h.dim; ++cc) RT[(size_t)cc * n_pool + j] = R[(size_t)j * h.dim + cc];
        std::vector<int32_t> sizes, lf(n_leaf + 1, 0);
        mrpt_leaf_sizes(h.n, 0, depth, sizes);
        for (uint32_t l = 0; l < n_leaf; ++l) lf[l + 1] = lf[l] + sizes[l];
        const uint32_t cap = next_pow2(h.n);
        hipError_t e = h.mrpt_R.ensure(R.size() * 4);
        if (e == hipSuccess) e = h.mrpt_RT.ensure(RT.size() * 4);
        if (e == hipSuccess) e = h.mrpt_splits.ensure((size_t)mp.n_trees * (n_leaf - 1) * 4);
        if (e == hipSuccess) e = h.mrpt_leaves.ensure((size_t)mp.n_trees * h.n * 4);
        if (e == hipSuccess) e = h.mrpt_lf.ensure(((size_t)n_leaf + 1) * 4);
        if (e == hipSuccess) e = proj.ensure((size_t)n_pool * h.n * 4);
        if (e == hipSuccess) e = keys.ensure((size_t)mp.n_trees * cap * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(h.mrpt_R.p, R.data(), R.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h.mrpt_RT.p, RT.data(), RT.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h.mrpt_lf.p, lf.data(), lf.size() * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_mrpt_project(c->stream, h.rows.as<float>(), h.n, h.dim, h.mrpt_R.as<float>(), mp.n_trees, depth, proj.as<float>());
        if (e == hipSuccess) e = launch_mrpt_trees(c->stream, proj.as<float>(), h.n, mp.n_trees, depth, cap, keys.as<unsigned long long>(),
                                                   h.mrpt_leaves.as<int32_t>(), h.mrpt_splits.as<float>());
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);             // R / RT / lf are host temporaries
        if (e != hipSuccess) { c->err = std::string("MRPT index build: ") + hipGetErrorString(e); rc = R3DM_ERR_HIP; break; }
        h.mrpt_trees = mp.n_trees; h.mrpt_depth = depth; h.mrpt_density = density; h.mrpt_seed = mp.seed;
    }
    proj.release(); keys.release();
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_build += ms;
    c->stats.n_ann_built += todo.size();
    return R3DM_OK;
}

static MrptView mrpt_view_of(const HostImage& h)
{
    MrptView v{};
    v.rows = h.rows.as<float>(); v.RT = h.mrpt_RT.as<float>(); v.splits = h.mrpt_splits.as<float>();
    v.leaves = h.mrpt_leaves.as<int32_t>(); v.leaf_first = h.mrpt_lf.as<int32_t>();
    v.n = h.n; v.dim = h.dim; v.n_trees = h.mrpt_trees; v.depth = h.mrpt_depth;
    return v;
}

// Mrpt::query + ratio test over `jobs` (every sI holds an index), results appended to g in job order
static int run_mrpt_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio, uint32_t votes, r3dm_graph* g,
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
    uint32_t max_nJ = 0, max_nI = 0, max_pool = 0, elected_cap = 0;
    uint64_t n_queries = 0;
    for (const PairJob& j : jobs) {
        const HostImage& I = *c->imgs[j.sI];
        max_nI = std::max(max_nI, I.n);
        max_nJ = std::max(max_nJ, c->imgs[j.sJ]->n);
        n_queries += c->imgs[j.sJ]->n;
        max_pool = std::max(max_pool, I.mrpt_trees * I.mrpt_depth);
        elected_cap = std::max<uint32_t>(elected_cap, mrpt_elected_cap(I.n, I.mrpt_trees, I.mrpt_depth, votes));
    }
    const uint32_t q_stride = std::max<uint32_t>(32, (max_nJ + 31) / 32 * 32);
    const uint32_t sort_cap = std::min<uint32_t>(16384, std::max<uint32_t>(8, next_pow2(q_stride)));
    std::vector<uint2> hp(P);
    std::vector<MrptQueryJob> sj(P);
    for (uint32_t p = 0; p < P; ++p) {
        hp[p] = make_uint2(jobs[p].sI, jobs[p].sJ);
        sj[p].ix = mrpt_view_of(*c->imgs[jobs[p].sI]);
        sj[p].query = c->imgs[jobs[p].sJ]->rows.as<float>();
        sj[p].nq = c->imgs[jobs[p].sJ]->n;
        sj[p].out_base = p * q_stride;
    }
    R3DM_HIP(c, c->d_pairs.ensure(sizeof(uint2) * P));
    R3DM_HIP(c, c->h_jobs.ensure(sizeof(MrptQueryJob) * P));
    R3DM_HIP(c, hipMemcpyAsync(c->d_pairs.p, hp.data(), sizeof(uint2) * P, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->h_jobs.p, sj.data(), sizeof(MrptQueryJob) * P, hipMemcpyHostToDevice, c->stream));
    const uint64_t total_slots = (uint64_t)P * q_stride;
    R3DM_HIP(c, c->d_nn.ensure((size_t)total_slots * 4));
    R3DM_HIP(c, c->d_cnt.ensure(64));
    if (knn_idx_host) {
        R3DM_HIP(c, c->d_knn_idx.ensure((size_t)total_slots * 8));
        R3DM_HIP(c, c->d_knn_dist.ensure((size_t)total_slots * 8));
    }
    MrptQueryParams qp{};
    qp.jobs = c->h_jobs.as<MrptQueryJob>(); qp.n_jobs = P;
    qp.votes = votes; qp.elected_cap = elected_cap; qp.ratio = ratio;
    qp.nn_idx = c->d_nn.as<uint32_t>();
    qp.knn_idx = knn_idx_host ? c->d_knn_idx.as<int32_t>() : nullptr;
    qp.knn_dist = knn_idx_host ? c->d_knn_dist.as<float>() : nullptr;
    qp.n_comps = reinterpret_cast<unsigned long long*>(c->d_cnt.as<uint32_t>() + 4);
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->d_cnt.p, 0, 64, c->stream));
    hipError_t e = launch_mrpt_query(c->stream, qp, max_nJ, max_nI, max_pool);
    if (e == hipErrorInvalidValue) { c->err = "MRPT query: vote bytes + elected list of the largest index view of the batch exceed the 160 KB of LDS a wavefront can have"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, e);
    unsigned long long comps = 0;
    R3DM_HIP(c, hipMemcpyAsync(&comps, qp.n_comps, 8, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));              // hp / sj are host temporaries
    const double t_post = now_ms();
    int rc = finalize_batch(c, jobs, q_stride, sort_cap, n_queries, max_nJ, g, knn_idx_host, knn_dist_host);
    if (rc != R3DM_OK) return rc;
    c->stats.ms_wall_match_post += now_ms() - t_post;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_search += ms;
    c->stats.n_ann_dist += comps;
    c->stats.n_match_launches += 1;
    c->stats.n_pairs += P;
    c->stats.n_queries += n_queries;
    return R3DM_OK;
}

static int r3dm_match_pairs_mrpt_impl(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                     const r3dm_mrpt_params* mp, r3dm_graph** out)
{
    if (!c || !out || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    int rc = check_mrpt_params(c, mp);
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
        if (A.dtype == R3DM_BIN || !mrpt_dim_ok(A.dim)) { c->err = "MRPT matching needs F32/U8 descriptors of a length that is a multiple of 4, at most 512"; return R3DM_ERR_UNSUPPORTED; }
        if (A.n > (1u << 17)) { c->err = "MRPT matching: more than 131,072 rows in one view"; return R3DM_ERR_UNSUPPORTED; }
        if (A.n < kAnnMinRows || !mrpt_query_fits_lds(A.n, *mp)) small_jobs.push_back({I, J, a->second, b->second});    // a forest over a handful of rows, or a view whose vote table no wavefront's LDS holds: such views are scanned
        else ann_jobs.push_back({I, J, a->second, b->second});
    }
    auto by_pair = [](const PairJob& x, const PairJob& y) { return x.I != y.I ? x.I < y.I : x.J < y.J; };
    auto same = [](const PairJob& x, const PairJob& y) { return x.I == y.I && x.J == y.J; };
    for (auto* v : {&ann_jobs, &small_jobs}) { std::sort(v->begin(), v->end(), by_pair); v->erase(std::unique(v->begin(), v->end(), same), v->end()); }

    r3dm_graph ga, gs;
    ga.offsets.push_back(0); gs.offsets.push_back(0);
    // (r3dm_set_device_graphs) the two part graphs are merged on the host: a device mirror survives that only when one part is the whole
    // result -- then it is built and handed over; with both kinds of pairs present no mirror is built at all (it would be dropped)
    PartMirrorGuard mirror_guard(c, !ann_jobs.empty() && !small_jobs.empty());
    if (!ann_jobs.empty()) {
        std::vector<uint32_t> slots;
        for (const PairJob& j : ann_jobs) slots.push_back(j.sI);
        rc = ensure_mrpt_indices(c, slots, *mp);
        if (rc != R3DM_OK) return rc;
    }
    size_t start = 0;
    while (start < ann_jobs.size()) {
        size_t end = start;
        uint32_t max_n = 0;
        while (end < ann_jobs.size() && end - start < 65535) {
            const uint32_t mn = std::max(max_n, c->imgs[ann_jobs[end].sJ]->n);
            const uint64_t s = (uint64_t)(end - start + 1) * ((mn + 31) / 32 * 32);
            if (end > start && (s * 4 > (3ull << 30) || s / 4 > kMaxBlocksOf256 - 4096)) break;
            max_n = mn; ++end;
        }
        std::vector<PairJob> batch(ann_jobs.begin() + start, ann_jobs.begin() + end);
        rc = run_mrpt_batch(c, batch, dist_ratio, mp->votes, &ga, nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    start = 0;
    while (start < small_jobs.size()) {
        size_t end = start;
        const HostImage& F = *c->imgs[small_jobs[start].sI];
        while (end < small_jobs.size() && c->imgs[small_jobs[end].sI]->dtype == F.dtype && c->imgs[small_jobs[end].sI]->dim == F.dim) ++end;
        std::vector<PairJob> batch(small_jobs.begin() + start, small_jobs.begin() + end);
        rc = run_match_batch(c, batch, dist_ratio * dist_ratio, &gs, nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    rc = merge_parts_keep_mirror(ga, gs, out);
    c->stats.ms_wall_match = now_ms() - t_call;
    return rc;
}

extern "C" int r3dm_match_pairs_mrpt(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                     const r3dm_mrpt_params* mp, r3dm_graph** out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_match_pairs_mrpt_impl(c, pairs_ij, n_pairs, dist_ratio, mp, out); });
}

extern "C" int r3dm_mrpt_knn2(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                              uint32_t dim, const r3dm_mrpt_params* mp, int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c || !dataset || !query || !out_idx || !out_dist) return R3DM_ERR_INVALID;
        int rc = check_mrpt_params(c, mp);
        if (rc != R3DM_OK) return rc;
        if (n_query < 1 || n_dataset < 2) return R3DM_ERR_INVALID;
        if (!mrpt_dim_ok(dim)) { c->err = "MRPT matching needs descriptors of a length that is a multiple of 4, at most 512"; return R3DM_ERR_UNSUPPORTED; }
        if (n_dataset < kAnnMinRows) { c->err = "r3dm_mrpt_knn2: fewer than 128 rows (such views are scanned: r3dm_knn2)"; return R3DM_ERR_UNSUPPORTED; }
        if (n_dataset > (1u << 17)) { c->err = "MRPT matching: more than 131,072 rows in one view"; return R3DM_ERR_UNSUPPORTED; }
        if (!mrpt_query_fits_lds(n_dataset, *mp)) { c->err = "r3dm_mrpt_knn2: the vote table of this many rows exceeds a wavefront's LDS with these parameters (such views are scanned: r3dm_knn2)"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, hipSetDevice(c->device));
        const uint32_t s0 = (uint32_t)c->imgs.size();
        c->imgs.emplace_back(new HostImage());
        c->imgs.emplace_back(new HostImage());
        rc = stage_into_slot(c, s0, 0, 0, 0, dataset, n_dataset, dim, R3DM_F32, nullptr);
        if (rc == R3DM_OK) rc = stage_into_slot(c, s0 + 1, 1, 0, 0, query, n_query, dim, R3DM_F32, nullptr);
        const r3dm_stats keep = c->stats;
        if (rc == R3DM_OK) rc = ensure_mrpt_indices(c, {s0}, *mp);
        if (rc == R3DM_OK) {
            std::vector<PairJob> jobs{{0, 1, s0, s0 + 1}};
            rc = run_mrpt_batch(c, jobs, 1.0f, mp->votes, nullptr, out_idx, out_dist);
        }
        const uint64_t evals = c->stats.n_ann_dist - keep.n_ann_dist;
        const double ms_b = c->stats.ms_ann_build - keep.ms_ann_build, ms_s = c->stats.ms_ann_search - keep.ms_ann_search;
        c->stats = keep;
        c->stats.n_ann_dist = evals; c->stats.ms_ann_build = ms_b; c->stats.ms_ann_search = ms_s;
        (void)hipStreamSynchronize(c->stream);
        c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
        c->imgs.pop_back(); c->imgs.pop_back();
        return rc;
    });
}

extern "C" int r3dm_mrpt_index(r3dm_ctx* c, uint32_t view_id, const r3dm_mrpt_params* mp, float* R, float* splits, int32_t* leaves,
                               int32_t* leaf_first, uint32_t* depth_out)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c) return R3DM_ERR_INVALID;
        int rc = check_mrpt_params(c, mp);
        if (rc != R3DM_OK) return rc;
        auto it = c->slot_of.find(view_id);
        if (it == c->slot_of.end()) { c->err = "unregistered view"; return R3DM_ERR_INVALID; }
        HostImage& h = *c->imgs[it->second];
        if (h.dtype == R3DM_BIN || !mrpt_dim_ok(h.dim) || h.n < kAnnMinRows || h.n > (1u << 17)) { c->err = "MRPT index needs 128 .. 131,072 F32/U8 rows of a length that is a multiple of 4, at most 512"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, hipSetDevice(c->device));
        rc = ensure_mrpt_indices(c, {it->second}, *mp);
        if (rc != R3DM_OK) return rc;
        const uint32_t n_leaf = 1u << h.mrpt_depth, n_pool = h.mrpt_trees * h.mrpt_depth;
        if (depth_out) *depth_out = h.mrpt_depth;
        if (R) R3DM_HIP(c, hipMemcpy(R, h.mrpt_R.p, (size_t)n_pool * h.dim * 4, hipMemcpyDeviceToHost));
        if (splits) R3DM_HIP(c, hipMemcpy(splits, h.mrpt_splits.p, (size_t)h.mrpt_trees * (n_leaf - 1) * 4, hipMemcpyDeviceToHost));
        if (leaves) R3DM_HIP(c, hipMemcpy(leaves, h.mrpt_leaves.p, (size_t)h.mrpt_trees * h.n * 4, hipMemcpyDeviceToHost));
        if (leaf_first) R3DM_HIP(c, hipMemcpy(leaf_first, h.mrpt_lf.p, ((size_t)n_leaf + 1) * 4, hipMemcpyDeviceToHost));
        return R3DM_OK;
    });
}
