// api_multi.cpp -- part of the host side of libr3dm.so: the single-process multi-GPU entry of the C ABI (include/r3dm.h,
// "multi-GPU" section).
//
// The reference treats image pairs as independent OpenMP iterations (/root/reference/src/R3DComputeMatches.cpp:437-489: I
// serial, `#pragma omp parallel for schedule(dynamic)` over J, critical insert into the std::map) and so does OpenMVG's
// geometric filter loop (:2099).  A C++ host therefore needs no collective to use the 8 GPUs of a node: one context per
// device, every view replicated, the pair list dealt to the devices by rows of I in snake order (cost-balanced, keeps the
// pairs of one I on one device for L2 reuse -- the same rule as regard3d_amd/dist.py uses between processes), one host
// thread per device, results merged into one PairWiseMatches ordered by (I, J).  Results are identical to a single-device
// run by construction: every per-pair computation (2-NN, ratio, AC-RANSAC sample stream keyed by (seed, I, J)) is
// independent of which device runs it.
#include "r3dm_ctx.hpp"

#include <thread>

struct r3dm_multi {
    std::vector<r3dm_ctx*> ctx;
    std::string err;
    uint64_t n_host_uploads = 0, n_peer_copies = 0;       // r3dm_multi_set_image: views that crossed PCIe / copies device to device
};

extern "C" {
int  r3dm_create(int device_id, r3dm_ctx** out);
void r3dm_destroy(r3dm_ctx* ctx);
int  r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);
}

// owner of every pair under the snake deal of rows I: rows sorted by decreasing pair count (stable: ascending I among
// equals), dealt r = 0..W-1, W-1..0, ...
extern "C" int r3dm_shard_pairs(const uint32_t* pairs_ij, uint64_t n_pairs, uint32_t world, uint32_t* owner_out)
{
    if ((n_pairs && (!pairs_ij || !owner_out)) || world == 0) return R3DM_ERR_INVALID;
    if (world == 1) { for (uint64_t p = 0; p < n_pairs; ++p) owner_out[p] = 0; return R3DM_OK; }
    try {
        std::vector<std::pair<uint32_t, uint64_t>> rows;             // (I, count)
        {
            std::vector<uint32_t> is(n_pairs);
            for (uint64_t p = 0; p < n_pairs; ++p) is[p] = pairs_ij[2 * p];
            std::sort(is.begin(), is.end());
            for (uint64_t p = 0; p < n_pairs;) {
                uint64_t e = p;
                while (e < n_pairs && is[e] == is[p]) ++e;
                rows.push_back({is[p], e - p});
                p = e;
            }
        }
        std::vector<uint32_t> order(rows.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rows[a].second > rows[b].second; });
        std::vector<uint32_t> owner_of_row(rows.size());
        for (size_t pos = 0; pos < order.size(); ++pos) {
            const size_t rnd = pos / world, off = pos % world;
            owner_of_row[order[pos]] = (uint32_t)((rnd & 1) ? world - 1 - off : off);
        }
        for (uint64_t p = 0; p < n_pairs; ++p) {
            const uint32_t I = pairs_ij[2 * p];
            const size_t k = std::lower_bound(rows.begin(), rows.end(), std::make_pair(I, (uint64_t)0),
                                              [](const std::pair<uint32_t, uint64_t>& a, const std::pair<uint32_t, uint64_t>& b) { return a.first < b.first; }) - rows.begin();
            owner_out[p] = owner_of_row[k];
        }
    } catch (...) { return R3DM_ERR_NOMEM; }
    return R3DM_OK;
}

extern "C" int r3dm_multi_create(const int* device_ids, int n_dev, r3dm_multi** out)
{
    if (!out || !device_ids || n_dev < 1) return R3DM_ERR_INVALID;
    *out = nullptr;
    auto* m = new (std::nothrow) r3dm_multi();
    if (!m) return R3DM_ERR_NOMEM;
    for (int k = 0; k < n_dev; ++k) {
        r3dm_ctx* c = nullptr;
        const int rc = r3dm_create(device_ids[k], &c);
        if (rc != R3DM_OK) {
            for (r3dm_ctx* x : m->ctx) r3dm_destroy(x);
            delete m;
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m;
    return R3DM_OK;
}

extern "C" void r3dm_multi_destroy(r3dm_multi* m)
{
    if (!m) return;
    for (r3dm_ctx* c : m->ctx) r3dm_destroy(c);
    delete m;
}

extern "C" int r3dm_multi_num_devices(const r3dm_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" r3dm_ctx* r3dm_multi_ctx(r3dm_multi* m, int k) { return (m && k >= 0 && k < (int)m->ctx.size()) ? m->ctx[k] : nullptr; }
extern "C" const char* r3dm_multi_last_error(const r3dm_multi* m) { return m ? m->err.c_str() : "null context"; }

// run fn(k, ctx) on one host thread per device; first failure wins
template <class Fn>
static int for_each_device(r3dm_multi* m, Fn fn)
{
    const size_t W = m->ctx.size();
    std::vector<int> rc(W, R3DM_OK);
    if (W == 1) rc[0] = fn(0, m->ctx[0]);
    else {
        std::vector<std::thread> th;
        try {
            for (size_t k = 0; k < W; ++k) th.emplace_back([&, k] { rc[k] = fn((uint32_t)k, m->ctx[k]); });
        } catch (...) { for (auto& t : th) t.join(); m->err = "could not start a host thread"; return R3DM_ERR_NOMEM; }
        for (auto& t : th) t.join();
    }
    for (size_t k = 0; k < W; ++k)
        if (rc[k] != R3DM_OK) { m->err = "device " + std::to_string(m->ctx[k]->device) + ": " + m->ctx[k]->err; return rc[k]; }
    return R3DM_OK;
}

// A view is replicated on every device.  It crosses PCIe ONCE: the raw descriptors and positions go host -> the first context's
// device (or are already device memory: then nothing crosses), and from there to every other context's device with
// hipMemcpyPeerAsync -- over xGMI where peer access exists, through a staged copy where it does not; every device then runs its own
// re-layout kernels on its local copy (r3dm_set_image takes device pointers).  C4 (1000 views, 4.2 GB) used to cross PCIe 8 times.
extern "C" int r3dm_multi_set_image(r3dm_multi* m, uint32_t view_id, uint32_t width, uint32_t height,
                                    const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    if (!m) return R3DM_ERR_INVALID;
    const size_t W = m->ctx.size();
    if (W == 1 || n == 0 || !desc)
        return for_each_device(m, [&](uint32_t, r3dm_ctx* c) { return r3dm_set_image(c, view_id, width, height, desc, n, dim, dtype, xy); });
    const size_t desc_bytes = (size_t)n * dim * (dtype == R3DM_F32 ? 4 : 1), xy_bytes = xy ? (size_t)n * 8 : 0;
    const size_t xy_off = (desc_bytes + 255) / 256 * 256;
    // where the caller's buffers live: device memory is the source of the fan-out as it is
    auto device_of = [](const void* p) -> int {
        hipPointerAttribute_t a{};
        if (p && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeDevice) return a.device;
        (void)hipGetLastError();
        return -1;
    };
    const int desc_dev = device_of(desc), xy_dev = xy ? device_of(xy) : -1;
    r3dm_ctx* c0 = m->ctx[0];
    const void* src_desc = desc; const void* src_xy = xy;
    int src_desc_dev = desc_dev, src_xy_dev = xy_dev;
    if (desc_dev < 0 || (xy && xy_dev < 0)) {
        R3DM_HIP(c0, hipSetDevice(c0->device));
        R3DM_HIP(c0, c0->m_raw.ensure(xy_off + xy_bytes + 256));
        if (desc_dev < 0) {
            R3DM_HIP(c0, hipMemcpyAsync(c0->m_raw.p, desc, desc_bytes, hipMemcpyHostToDevice, c0->stream));
            src_desc = c0->m_raw.p; src_desc_dev = c0->device;
        }
        if (xy && xy_dev < 0) {
            R3DM_HIP(c0, hipMemcpyAsync(c0->m_raw.as<unsigned char>() + xy_off, xy, xy_bytes, hipMemcpyHostToDevice, c0->stream));
            src_xy = c0->m_raw.as<unsigned char>() + xy_off; src_xy_dev = c0->device;
        }
        R3DM_HIP(c0, hipStreamSynchronize(c0->stream));
        m->n_host_uploads += 1;
    }
    std::vector<uint64_t> peer(W, 0);
    const int rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) -> int {
        const void* d = src_desc; const void* x = src_xy;
        R3DM_HIP(c, hipSetDevice(c->device));
        if (k != 0 || src_desc_dev != c->device || (xy && src_xy_dev != c->device)) {
            // this context's local copy (context 0 reads its own upload in place when it already sits on its device)
            const bool own = (c == c0) && src_desc == c0->m_raw.p;
            if (!own) {
                R3DM_HIP(c, c->m_peer.ensure(xy_off + xy_bytes + 256));
                R3DM_HIP(c, hipMemcpyPeerAsync(c->m_peer.p, c->device, src_desc, src_desc_dev, desc_bytes, c->stream));
                if (xy) R3DM_HIP(c, hipMemcpyPeerAsync(c->m_peer.as<unsigned char>() + xy_off, c->device, src_xy, src_xy_dev, xy_bytes, c->stream));
                R3DM_HIP(c, hipStreamSynchronize(c->stream));
                d = c->m_peer.p; x = xy ? (const void*)(c->m_peer.as<unsigned char>() + xy_off) : nullptr;
                peer[k] = 1;
            }
        }
        return r3dm_set_image(c, view_id, width, height, d, n, dim, dtype, (const float*)x);
    });
    for (uint64_t v : peer) m->n_peer_copies += v;
    return rc;
}

// how the views registered so far travelled: uploads from host memory (one per view, whatever the number of devices) and
// device-to-device copies (one per view and further context)
extern "C" int r3dm_multi_transfer_counts(const r3dm_multi* m, uint64_t* host_uploads, uint64_t* peer_copies)
{
    if (!m) return R3DM_ERR_INVALID;
    if (host_uploads) *host_uploads = m->n_host_uploads;
    if (peer_copies) *peer_copies = m->n_peer_copies;
    return R3DM_OK;
}

extern "C" int r3dm_multi_set_intrinsics(r3dm_multi* m, uint32_t view_id, const double* K)
{
    if (!m) return R3DM_ERR_INVALID;
    return for_each_device(m, [&](uint32_t, r3dm_ctx* c) { return r3dm_set_intrinsics(c, view_id, K); });
}

extern "C" int r3dm_multi_clear_images(r3dm_multi* m)
{
    if (!m) return R3DM_ERR_INVALID;
    return for_each_device(m, [&](uint32_t, r3dm_ctx* c) { return r3dm_clear_images(c); });
}

extern "C" int r3dm_multi_set_integer_mfma(r3dm_multi* m, int enable)
{
    if (!m) return R3DM_ERR_INVALID;
    for (r3dm_ctx* c : m->ctx) r3dm_set_integer_mfma(c, enable);
    return R3DM_OK;
}

extern "C" int r3dm_multi_match_pairs(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs,
                                      float dist_ratio, int squared_metric, r3dm_graph** out)
{
    if (!m || !out || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    const uint32_t W = (uint32_t)m->ctx.size();
    try {
        std::vector<uint32_t> owner(n_pairs);
        int rc = r3dm_shard_pairs(pairs_ij, n_pairs, W, owner.data());
        if (rc != R3DM_OK) return rc;
        std::vector<std::vector<uint32_t>> mine(W);
        for (uint64_t p = 0; p < n_pairs; ++p) { mine[owner[p]].push_back(pairs_ij[2 * p]); mine[owner[p]].push_back(pairs_ij[2 * p + 1]); }
        std::vector<r3dm_graph*> parts(W, nullptr);
        rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) {
            return r3dm_match_pairs(c, mine[k].data(), mine[k].size() / 2, dist_ratio, squared_metric, &parts[k]);
        });
        if (rc == R3DM_OK) rc = r3dm_graph_merge(parts.data(), W, out);
        for (r3dm_graph* g : parts) r3dm_graph_free(g);
        return rc;
    } catch (...) { m->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}

extern "C" int r3dm_multi_match_pairs_kgraph(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                             const r3dm_kgraph_params* params, r3dm_graph** out)
{
    if (!m || !out || !params || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    const uint32_t W = (uint32_t)m->ctx.size();
    try {
        std::vector<uint32_t> owner(n_pairs);
        int rc = r3dm_shard_pairs(pairs_ij, n_pairs, W, owner.data());
        if (rc != R3DM_OK) return rc;
        std::vector<std::vector<uint32_t>> mine(W);
        for (uint64_t p = 0; p < n_pairs; ++p) { mine[owner[p]].push_back(pairs_ij[2 * p]); mine[owner[p]].push_back(pairs_ij[2 * p + 1]); }
        std::vector<r3dm_graph*> parts(W, nullptr);
        rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) {
            return r3dm_match_pairs_kgraph(c, mine[k].data(), mine[k].size() / 2, dist_ratio, params, &parts[k]);
        });
        if (rc == R3DM_OK) rc = r3dm_graph_merge(parts.data(), W, out);
        for (r3dm_graph* g : parts) r3dm_graph_free(g);
        return rc;
    } catch (...) { m->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}

// the same deal for the HNSW matcher (hnsw_match, matchingAlgorithm 6..8): a device builds the index of every image I whose row it owns
extern "C" int r3dm_multi_match_pairs_hnsw(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                           const r3dm_hnsw_params* params, r3dm_graph** out)
{
    if (!m || !out || !params || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    const uint32_t W = (uint32_t)m->ctx.size();
    try {
        std::vector<uint32_t> owner(n_pairs);
        int rc = r3dm_shard_pairs(pairs_ij, n_pairs, W, owner.data());
        if (rc != R3DM_OK) return rc;
        std::vector<std::vector<uint32_t>> mine(W);
        for (uint64_t p = 0; p < n_pairs; ++p) { mine[owner[p]].push_back(pairs_ij[2 * p]); mine[owner[p]].push_back(pairs_ij[2 * p + 1]); }
        std::vector<r3dm_graph*> parts(W, nullptr);
        rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) {
            return r3dm_match_pairs_hnsw(c, mine[k].data(), mine[k].size() / 2, dist_ratio, params, &parts[k]);
        });
        if (rc == R3DM_OK) rc = r3dm_graph_merge(parts.data(), W, out);
        for (r3dm_graph* g : parts) r3dm_graph_free(g);
        return rc;
    } catch (...) { m->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}

// ... and for the MRPT matcher (mrpt_match, matchingAlgorithm 5)
extern "C" int r3dm_multi_match_pairs_mrpt(r3dm_multi* m, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                           const r3dm_mrpt_params* params, r3dm_graph** out)
{
    if (!m || !out || !params || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    const uint32_t W = (uint32_t)m->ctx.size();
    try {
        std::vector<uint32_t> owner(n_pairs);
        int rc = r3dm_shard_pairs(pairs_ij, n_pairs, W, owner.data());
        if (rc != R3DM_OK) return rc;
        std::vector<std::vector<uint32_t>> mine(W);
        for (uint64_t p = 0; p < n_pairs; ++p) { mine[owner[p]].push_back(pairs_ij[2 * p]); mine[owner[p]].push_back(pairs_ij[2 * p + 1]); }
        std::vector<r3dm_graph*> parts(W, nullptr);
        rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) {
            return r3dm_match_pairs_mrpt(c, mine[k].data(), mine[k].size() / 2, dist_ratio, params, &parts[k]);
        });
        if (rc == R3DM_OK) rc = r3dm_graph_merge(parts.data(), W, out);
        for (r3dm_graph* g : parts) r3dm_graph_free(g);
        return rc;
    } catch (...) { m->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}

// model_kind as in api_filter.cpp: 0 F, 1 H, 2 E.  Putative pairs are dealt longest list first (the AC-RANSAC of a pair
// costs about as much as it has putatives) round-robin in snake order; the models of the kept pairs are re-ordered with
// the merged graph.
static int multi_filter(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter, uint64_t seed,
                        int model_kind, uint32_t min_count, float min_ratio, r3dm_graph** out, double* M_out)
{
    if (!m || !putative || !out) return R3DM_ERR_INVALID;
    *out = nullptr;
    const uint32_t W = (uint32_t)m->ctx.size();
    try {
        const uint64_t NP = putative->pairs.size() / 2;
        std::vector<uint64_t> order(NP);
        std::iota(order.begin(), order.end(), 0ull);
        std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            return putative->offsets[a + 1] - putative->offsets[a] > putative->offsets[b + 1] - putative->offsets[b]; });
        std::vector<r3dm_graph> sub(W);
        std::vector<std::vector<uint64_t>> idx(W);
        for (uint64_t pos = 0; pos < NP; ++pos) {
            const uint64_t rnd = pos / W, off = pos % W;
            idx[(rnd & 1) ? W - 1 - off : off].push_back(order[pos]);
        }
        for (uint32_t k = 0; k < W; ++k) {
            std::sort(idx[k].begin(), idx[k].end());                 // PairWiseMatches order inside every shard
            sub[k].offsets.push_back(0);
            for (uint64_t p : idx[k]) {
                sub[k].pairs.push_back(putative->pairs[2 * p]); sub[k].pairs.push_back(putative->pairs[2 * p + 1]);
                sub[k].matches.insert(sub[k].matches.end(), putative->matches.begin() + putative->offsets[p],
                                      putative->matches.begin() + putative->offsets[p + 1]);
                sub[k].offsets.push_back(sub[k].matches.size());
            }
        }
        std::vector<r3dm_graph*> parts(W, nullptr);
        std::vector<std::vector<double>> models(W);
        int rc = for_each_device(m, [&](uint32_t k, r3dm_ctx* c) {
            double* mo = nullptr;
            if (M_out) { models[k].assign(9 * std::max<size_t>(idx[k].size(), 1), 0.0); mo = models[k].data(); }
            switch (model_kind) {
                case 0:  return r3dm_filter_F(c, &sub[k], max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, &parts[k], mo);
                case 1:  return r3dm_filter_H(c, &sub[k], max_residual_px, max_iter, seed, &parts[k], mo);
                default: return r3dm_filter_E(c, &sub[k], max_residual_px, max_iter, seed, min_count, min_ratio, &parts[k], mo);
            }
        });
        if (rc == R3DM_OK) rc = r3dm_graph_merge(parts.data(), W, out);
        if (rc == R3DM_OK && M_out) {
            std::map<std::pair<uint32_t, uint32_t>, const double*> where;
            for (uint32_t k = 0; k < W; ++k)
                for (size_t q = 0; q < parts[k]->pairs.size() / 2; ++q)
                    where[{parts[k]->pairs[2 * q], parts[k]->pairs[2 * q + 1]}] = models[k].data() + 9 * q;
            const r3dm_graph* g = *out;
            for (size_t q = 0; q < g->pairs.size() / 2; ++q)
                memcpy(M_out + 9 * q, where[{g->pairs[2 * q], g->pairs[2 * q + 1]}], 72);
        }
        for (r3dm_graph* g : parts) r3dm_graph_free(g);
        return rc;
    } catch (...) { m->err = "out of host memory"; return R3DM_ERR_NOMEM; }
}

extern "C" int r3dm_multi_filter_F(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                                   uint64_t seed, r3dm_graph** out, double* F_out)
{
    return multi_filter(m, putative, max_residual_px, max_iter, seed, 0, 0, 0.f, out, F_out);
}

extern "C" int r3dm_multi_filter_H(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                                   uint64_t seed, r3dm_graph** out, double* H_out)
{
    return multi_filter(m, putative, max_residual_px, max_iter, seed, 1, 0, 0.f, out, H_out);
}

extern "C" int r3dm_multi_filter_E(r3dm_multi* m, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                                   uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out)
{
    return multi_filter(m, putative, max_residual_px, max_iter, seed, 2, min_count, min_ratio, out, E_out);
}
