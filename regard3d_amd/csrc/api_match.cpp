// api_match.cpp -- part of the host side of libr3dm.so: the C ABI declared in include/r3dm.h (see r3dm_ctx.hpp for the file map).
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 and src/Regard3DFeatures.cpp -- with every arithmetic stage running as
// HIP kernels on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"
#include <cstddef>

// ------------------------------------------------------------------------------------------------
// putative matching
// ------------------------------------------------------------------------------------------------
extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);

// compaction + ordering + de-duplication of nn_idx[pair][*] (finalize_pairs_kernel), copy back, append the non-empty
// pairs to `g` in job order.  Shared by the exhaustive and the graph-search drivers.
int finalize_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, uint32_t q_stride, uint32_t sort_cap,
                          uint64_t n_queries, uint32_t max_nJ, r3dm_graph* g, int32_t* knn_idx_host, float* knn_dist_host)
{
    const uint32_t P = (uint32_t)jobs.size();
    // ---- finalisation: compact + order + de-duplicate, per pair
    R3DM_HIP(c, c->d_pair_off.ensure((size_t)P * 8));
    R3DM_HIP(c, c->d_pair_cnt.ensure((size_t)P * 4));
    uint64_t out_cap = std::max<uint64_t>(1u << 20, n_queries / 4);
    std::vector<uint64_t> h_off(P);
    std::vector<uint32_t> h_cnt(P);
    unsigned long long total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        R3DM_HIP(c, c->d_out.ensure(out_cap * sizeof(r3dm_match)));
        R3DM_HIP(c, hipMemsetAsync(c->d_cnt.as<uint32_t>() + 8, 0, 8, c->stream));
        FinalizeParams fp{};
        fp.imgs = c->d_imgs.as<ImgDev>(); fp.pairs = c->d_pairs.as<uint2>(); fp.n_pairs = P; fp.q_stride = q_stride;
        fp.nn_idx = c->d_nn.as<uint32_t>(); fp.sort_cap = sort_cap;
        fp.spill_keys = nullptr; fp.spill_drop = nullptr; fp.spill_stride = 0;
        if (q_stride > sort_cap) {                         // views with more rows than the LDS sort holds
            fp.spill_stride = next_pow2(q_stride);
            R3DM_HIP(c, c->d_spill.ensure((size_t)P * fp.spill_stride * 9));
            fp.spill_keys = c->d_spill.as<unsigned long long>();
            fp.spill_drop = c->d_spill.as<unsigned char>() + (size_t)P * fp.spill_stride * 8;
        }
        fp.out = c->d_out.as<r3dm_match>(); fp.out_cap = out_cap;
        fp.total = reinterpret_cast<unsigned long long*>(c->d_cnt.as<uint32_t>() + 8);
        fp.pair_off = c->d_pair_off.as<uint64_t>(); fp.pair_cnt = c->d_pair_cnt.as<uint32_t>();
        R3DM_HIP(c, launch_finalize(c->stream, fp));
        // (every copy back lands in page-locked memory: pageable destinations go through the runtime's staging path)
        R3DM_HIP(c, c->pin_small.ensure(64 + (size_t)P * 12));
        unsigned char* ps = static_cast<unsigned char*>(c->pin_small.p);
        R3DM_HIP(c, hipMemcpyAsync(ps, fp.total, 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(ps + 64, fp.pair_off, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(ps + 64 + (size_t)P * 8, fp.pair_cnt, (size_t)P * 4, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(&total, ps, 8); memcpy(h_off.data(), ps + 64, (size_t)P * 8); memcpy(h_cnt.data(), ps + 64 + (size_t)P * 8, (size_t)P * 4);
        if (total <= out_cap) break;
        out_cap = total;                                   // overflow: nothing was lost, run it again with room
    }
    // the matches come back through the context's page-locked landing buffer on the context's stream (a synchronous hipMemcpy into a
    // fresh pageable vector runs on the null stream, pins the pages on the way and was seen to take 80 ms for 7.7 MB every few calls)
    const r3dm_match* h_m = nullptr;
    if (total) {
        R3DM_HIP(c, c->pin_out.ensure((size_t)total * sizeof(r3dm_match)));
        R3DM_HIP(c, hipMemcpyAsync(c->pin_out.p, c->d_out.p, (size_t)total * sizeof(r3dm_match), hipMemcpyDeviceToHost, c->stream));
        h_m = static_cast<const r3dm_match*>(c->pin_out.p);
    }
    if (knn_idx_host) {
        // single-pair use (r3dm_knn2): copy the raw 2-NN of pair 0
        R3DM_HIP(c, hipMemcpyAsync(knn_idx_host, c->d_knn_idx.p, (size_t)max_nJ * 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(knn_dist_host, c->d_knn_dist.p, (size_t)max_nJ * 8, hipMemcpyDeviceToHost, c->stream));
    }
    R3DM_HIP(c, hipStreamSynchronize(c->stream));

    if (g) {
        // r3dm_set_device_graphs: the same pairs, in the same order, into the graph's device mirror straight from d_out (no payload byte
        // of the mirror crosses PCIe: ids and counts are 12 bytes per pair of host-side bookkeeping)
        const bool mirror = c->device_graphs && (g->dev.valid || (g->pairs.empty() && g->dev.P == 0));
        if (mirror && !g->dev.valid) { g->dev.valid = true; g->dev.device = c->device; }
        std::vector<uint32_t> ids, cnts;
        std::vector<GraphSeg> segs;
        uint64_t dst = 0;
        g->matches.reserve(g->matches.size() + (size_t)total);     // (one allocation: a graph of millions of matches otherwise re-grows a dozen times)
        g->pairs.reserve(g->pairs.size() + 2 * (size_t)P); g->offsets.reserve(g->offsets.size() + P);
        for (uint32_t p = 0; p < P; ++p) {
            if (h_cnt[p] == 0) continue;                   // empty vectors never enter the map
            g->pairs.push_back(jobs[p].I); g->pairs.push_back(jobs[p].J);
            g->matches.insert(g->matches.end(), h_m + h_off[p], h_m + h_off[p] + h_cnt[p]);
            g->offsets.push_back(g->matches.size());
            if (mirror) { ids.push_back(jobs[p].I); ids.push_back(jobs[p].J); cnts.push_back(h_cnt[p]); segs.push_back(GraphSeg{h_off[p], 0, dst, h_cnt[p], 0}); dst += h_cnt[p]; }
        }
        if (mirror) (void)graph_dev_append(c, g, ids, cnts, segs, c->d_out.as<r3dm_match>(), nullptr);
    }
    return R3DM_OK;
}

PartMirrorGuard::PartMirrorGuard(r3dm_ctx* c_, bool suppress) : c(c_), keep(c_->device_graphs) { if (suppress) c->device_graphs = false; }
PartMirrorGuard::~PartMirrorGuard() { c->device_graphs = keep; }

// merged = ga + gs ordered by (I, J).  When one part is empty the other one IS the result (its batches appended pairs in (I, J) order):
// its device mirror moves to the merged graph instead of being dropped with the part.
int merge_parts_keep_mirror(r3dm_graph& ga, r3dm_graph& gs, r3dm_graph** out)
{
    const r3dm_graph* parts[2] = {&ga, &gs};
    const int rc = r3dm_graph_merge(parts, 2, out);
    if (rc != R3DM_OK || !*out) return rc;
    r3dm_graph* whole = ga.pairs.empty() ? &gs : (gs.pairs.empty() ? &ga : nullptr);
    if (whole && whole->dev.valid && whole->dev.P == (*out)->pairs.size() / 2 && whole->dev.M == (*out)->matches.size() && (*out)->pairs == whole->pairs) {
        (*out)->dev = whole->dev;                 // (plain handles: the part forgets them, the merged graph frees them)
        whole->dev = GraphDev();
    }
    return rc;
}

// runs the 2-NN + ratio kernels over `jobs` (slot pairs, all of one dtype/dim) and appends the
// non-empty results to `g` in job order.  knn_idx/knn_dist (host, optional) receive the raw 2-NN.
int run_match_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio_R, r3dm_graph* g,
                    int32_t* knn_idx_host, float* knn_dist_host)
{
    const uint32_t P = (uint32_t)jobs.size();
    if (P == 0) return R3DM_OK;
    const double t_dbg0 = now_ms();
    { const int rcs = sync_view_stats(c); if (rcs != R3DM_OK) return rcs; }      // which path a batch takes depends on its views' statistics
    const HostImage& first = *c->imgs[jobs[0].sI];
    const r3dm_dtype dtype = first.dtype;
    uint32_t max_nJ = 0, max_tiles = 0;
    uint64_t n_queries = 0;
    double flops = 0, bytes = 0;
    for (const PairJob& j : jobs) {
        const HostImage& A = *c->imgs[j.sI];
        const HostImage& B = *c->imgs[j.sJ];
        max_nJ = std::max(max_nJ, B.n);
        max_tiles = std::max(max_tiles, B.n_tiles);
        n_queries += B.n;
        if (dtype == R3DM_BIN) {
            flops += 2.0 * A.n * (double)B.n * A.words;
            bytes += ((double)A.n + B.n) * A.words * 4 + (double)B.n * 16;
        } else {
            flops += 2.0 * A.n * (double)B.n * A.dim;
            bytes += ((double)A.n + B.n) * A.dim * 4 + (double)B.n * 16;
        }
    }
    // the views of the batch (layouts are staged per view, on first use)
    std::vector<uint32_t> batch_slots;
    batch_slots.reserve(2 * (size_t)P);
    for (const PairJob& j : jobs) { batch_slots.push_back(j.sI); batch_slots.push_back(j.sJ); }
    std::sort(batch_slots.begin(), batch_slots.end());
    batch_slots.erase(std::unique(batch_slots.begin(), batch_slots.end()), batch_slots.end());
    // same test as the kernels' exact_pair (kernels_match_common.hpp): key + ||q||^2 IS the reference distance, bit for bit
    auto exact_pair = [&](const HostImage& A, const HostImage& B, bool bf16) {
        const float dpad = (float)(first.G * 8), mI = A.max_abs, mJ = B.max_abs;
        return !A.not_integer && !B.not_integer &&
               ((A.has_negative || B.has_negative)
                    ? dpad * (mI + mJ) * (mI + mJ) < 16777216.0f
                    : (2.0f * dpad * mI * mJ < 16777216.0f && dpad * mI * mI < 16777216.0f && dpad * mJ * mJ < 16777216.0f)) &&
               (!bf16 || (mI <= 256.0f && mJ <= 256.0f));
    };
    // integer fast path (r3dm_set_integer_mfma): every view of the batch must hold bf16-exact integers
    bool int_mfma = c->integer_mfma && dtype != R3DM_BIN && first.G != 18;
    if (int_mfma)
        for (const PairJob& j : jobs)
            if (!exact_pair(*c->imgs[j.sI], *c->imgs[j.sJ], true)) { int_mfma = false; break; }
    // split-f16 nominator (r3dm_set_split_mfma): batches with at least one real-valued view (integer-valued batches are exact
    // on the f32 tiles already and have their own fast path); every view finite, scales within reach of one another
    bool split = c->split_mfma && !int_mfma && dtype != R3DM_BIN && has_tensor_kernel(first.G);
    if (split) {
        bool any_real = false;
        for (const PairJob& j : jobs) {
            const HostImage& A = *c->imgs[j.sI];
            const HostImage& B = *c->imgs[j.sJ];
            any_real |= A.not_integer || B.not_integer;
            if (!std::isfinite(A.max_abs) || !std::isfinite(B.max_abs) || !(A.max_abs > 0.0f) || !(B.max_abs > 0.0f) ||
                std::abs(A.split_k - B.split_k) > 40 || std::abs(A.split_k + B.split_k) > 100) { split = false; break; }
        }
        split = split && any_real;
    }
    // ... and its cheaper form for views whose rows are small integers x a row scale (LIOP: one f16 MFMA per 16 dimensions on the
    // count tiles instead of three on the split planes): every view of the batch must pass the staging check
    bool counts = split && !r3dm_dev_knob("R3DM_NO_COUNT_TILES", 0);          // (developer build: A/B against the split planes)
    if (counts)
        for (uint32_t s : batch_slots) { const HostImage& h = *c->imgs[s]; if (!(h.dtype == R3DM_F32 && h.n && h.dim <= 256)) { counts = false; break; } }
    // ---- the layouts this batch reads that its views do not hold yet (staged once per view; kernels_match.hip, kernels_match_16bit.hip)
    {
        int rcl = R3DM_OK;
        if (dtype == R3DM_BIN) { if (c->hamming_mfma) rcl = ensure_layouts(c, batch_slots, kLayBin8); }
        else if (int_mfma) rcl = ensure_layouts(c, batch_slots, kLayBf16);
        else if (split) {
            if (counts) { bool all_ok = false; rcl = ensure_layouts(c, batch_slots, kLayRows | kLayCounts, &all_ok); counts = all_ok; }
            if (rcl == R3DM_OK && !counts) rcl = ensure_layouts(c, batch_slots, kLayRows | kLaySplit);
        } else if (has_tensor_kernel(first.G)) {
            // the f32 tiles: a pair of integer-valued views never re-reads a row (exact_pair); any other pair re-scores its nominees from
            // the row-major rows
            bool all_exact = true;
            for (const PairJob& j : jobs) if (!exact_pair(*c->imgs[j.sI], *c->imgs[j.sJ], false)) { all_exact = false; break; }
            if (!all_exact) rcl = ensure_layouts(c, batch_slots, kLayRows);
        } else rcl = ensure_layouts(c, batch_slots, kLayRows);             // no tensor kernel: the exact scan of every query reads the rows
        if (rcl != R3DM_OK) return rcl;
    }
    const uint32_t q_stride = std::max<uint32_t>(32, (max_nJ + 31) / 32 * 32);
    const uint32_t sort_cap = std::min<uint32_t>(16384, std::max<uint32_t>(8, next_pow2(q_stride)));   // LDS budget; larger views may spill

    std::vector<uint2> hp(P);
    for (uint32_t p = 0; p < P; ++p) hp[p] = make_uint2(jobs[p].sI, jobs[p].sJ);
    R3DM_HIP(c, c->d_pairs.ensure(sizeof(uint2) * P));
    R3DM_HIP(c, hipMemcpyAsync(c->d_pairs.p, hp.data(), sizeof(uint2) * P, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, c->d_nn.ensure((size_t)P * q_stride * 4));
    const uint64_t total_slots = (uint64_t)P * q_stride;
    // per-pair lists of uncertified queries: [fb_total(2 words) | pad][fb_cnt: P][fb_q: P x kFbPerPair]
    const size_t fb_words = 16 + (size_t)P + (size_t)P * kFbPerPair;
    R3DM_HIP(c, c->d_fb.ensure(fb_words * 4));
    R3DM_HIP(c, hipMemsetAsync(c->d_fb.p, 0, (16 + (size_t)P) * 4, c->stream));
    R3DM_HIP(c, c->d_cnt.ensure(64));
    R3DM_HIP(c, hipMemsetAsync(c->d_cnt.p, 0, 64, c->stream));
    if (knn_idx_host) {
        R3DM_HIP(c, c->d_knn_idx.ensure((size_t)total_slots * 8));
        R3DM_HIP(c, c->d_knn_dist.ensure((size_t)total_slots * 8));
    }

    MatchParams mp{};
    mp.imgs = c->d_imgs.as<ImgDev>();
    mp.pairs = c->d_pairs.as<uint2>();
    mp.n_pairs = P; mp.qb_per_pair = 0; mp.q_stride = q_stride;
    mp.ratio_R = ratio_R;
    // certification slack factor: |MFMA-path distance - reference distance| <= (3.5 D + 14) u (max||a||^2 + ||q||^2),
    // u = 2^-24 (DESIGN.md "Certification"); 4.25 D u covers it for every padded D >= 64
    mp.err_scale = 4.25f * (float)(first.G * 8) * 5.9604645e-08f;
    // split-f16 keys: residue of the two-piece split 3 x 2^-22 ||a|| ||b|| <= 1.5 x 2^-22 (||a||^2 + ||b||^2), f32 accumulation of
    // 3 Dpad products + one C operand per MFMA with a one-sided 2^-23 per addition on partial sums <= 2 (||a||^2 + ||b||^2), plus the
    // reference sum's own (D/2 + 12) 2^-24 -- together below (3 Dpad + 34) 2^-22, + 2 for the count kernel's bias (kernels_match.hip, l2_knn2_split_kernel)
    if (split) mp.err_scale = (3.0f * (float)(first.G * 8) + 36.0f) * 2.3841858e-07f;      // (+ 2: the count kernel's keys carry ||b||^2 / (2 s_b) and drop it again)
    mp.nn_idx = c->d_nn.as<uint32_t>();
    mp.knn_idx = knn_idx_host ? c->d_knn_idx.as<int32_t>() : nullptr;
    mp.knn_dist = knn_idx_host ? c->d_knn_dist.as<float>() : nullptr;
    mp.fb_total = c->d_fb.as<uint32_t>();
    mp.fb_cnt = c->d_fb.as<uint32_t>() + 16;
    mp.fb_q = c->d_fb.as<uint32_t>() + 16 + P;

    const double t_dbg1 = now_ms();
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    uint64_t n_fallback = 0;
    if (dtype == R3DM_BIN) {
        if (c->hamming_mfma) {
            R3DM_HIP(c, launch_hamming_mfma(c->stream, mp, first.words, max_tiles));
            c->stats.n_hamming_mfma += 1;
        } else R3DM_HIP(c, launch_hamming_knn2(c->stream, mp, first.words, max_nJ));
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));      // (the finaliser would wait here anyway; keeps the wall breakdown honest)
    } else if (has_tensor_kernel(first.G)) {
        bool counts_ran = false;
        if (counts) {
            // hipErrorInvalidValue = no count kernel for this launch (descriptor length, or a grid beyond the launcher's bound): the split
            // tiles, which every such view also holds, serve the batch
            // (the one-list kernel packs a key and its row into 32 bits: views beyond 65,536 rows would leave its keys fewer than seven
            //  mantissa bits and send a growing share of their queries to the exact scan -- they take the two-list kernel, float keys)
            uint32_t max_nI_rows = 0;
            for (const PairJob& j : jobs) max_nI_rows = std::max(max_nI_rows, c->imgs[j.sI]->n);
            const int two_lists = (max_nI_rows > 65536u || r3dm_dev_knob("R3DM_COUNTS_TWO_LISTS", 0)) ? 1 : 0;
            const hipError_t ec = launch_l2_knn2_counts(c->stream, mp, first.G, max_tiles, two_lists);
            if (ec == hipErrorInvalidValue) (void)hipGetLastError();
            else { R3DM_HIP(c, ec); counts_ran = true; c->stats.n_split_mfma += 1; c->stats.n_counts_mfma += 1; }
        }
        if (counts_ran) {
            // (launched above)
        } else if (split) {
            if (counts) { const int rcl = ensure_layouts(c, batch_slots, kLayRows | kLaySplit); if (rcl != R3DM_OK) return rcl; }      // (no count kernel for this launch: the split planes after all)
            R3DM_HIP(c, launch_l2_knn2_split(c->stream, mp, first.G, max_tiles));
            c->stats.n_split_mfma += 1;
        } else {
            R3DM_HIP(c, launch_l2_knn2(c->stream, mp, first.G, max_tiles, int_mfma));
            if (int_mfma) c->stats.n_integer_mfma += 1;
        }
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
        uint32_t fbt[2] = {0, 0};
        R3DM_HIP(c, c->pin_small.ensure(64));
        R3DM_HIP(c, hipMemcpyAsync(c->pin_small.p, mp.fb_total, 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        memcpy(fbt, c->pin_small.p, 8);
        n_fallback = fbt[0];
        if (fbt[0] > 0) {
            bool rescan = fbt[1] > 0;                      // some pair overflowed its list
            if ((first.dim & 3u) == 0) {
                // a pair's uncertified queries are scanned by one workgroup per ~1024 rows of image I (at most 64): few pairs with
                // long views (24 views of 28 k rows: 276 workgroups of 7.7 ms each) otherwise leave the chip idle behind one round,
                // and a workgroup walks its rows tile by tile behind two barriers each (4096 rows per workgroup: 3.9 ms on those views)
                uint32_t max_nI = 0;
                for (const PairJob& j : jobs) max_nI = std::max(max_nI, c->imgs[j.sI]->n);
                uint32_t S = std::min<uint32_t>(64u, std::max<uint32_t>(1u, (max_nI + 1023u) / 1024u));
                while (S > 1 && (uint64_t)P * S > 65535ull * 4) --S;
                mp.fb_slices = S; mp.fb_part = nullptr; mp.fb_done = nullptr;
                if (S > 1) {
                    R3DM_HIP(c, c->d_fb2.ensure((size_t)P * kFbPerPair * S * 16 + (size_t)P * 4 + 64));
                    mp.fb_part = c->d_fb2.as<float4>();
                    mp.fb_done = reinterpret_cast<uint32_t*>(c->d_fb2.as<unsigned char>() + (size_t)P * kFbPerPair * S * 16);
                    R3DM_HIP(c, hipMemsetAsync(mp.fb_done, 0, (size_t)P * 4, c->stream));
                }
                R3DM_HIP(c, launch_l2_exact_batch(c->stream, mp, first.G));
            }
            else rescan = true;                            // scalar-tail dims: generic exact kernel
            if (rescan) {
                if (total_slots > 0xFFFFFFFFull) { c->err = "batch too large for the exact rescan"; return R3DM_ERR_UNSUPPORTED; }
                { const int rcl = ensure_layouts(c, batch_slots, kLayRows); if (rcl != R3DM_OK) return rcl; }      // the per-query scan reads row-major rows
                R3DM_HIP(c, launch_l2_exact_items(c->stream, mp, (uint32_t)total_slots, 2));
            }
        }
    } else {
        // descriptor length without a tensor kernel: exact scan of every query (slow, still on the GPU)
        if (total_slots > 0xFFFFFFFFull) { c->err = "batch too large for the exact scan"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, launch_l2_exact_items(c->stream, mp, (uint32_t)total_slots, 1));
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
        n_fallback = n_queries;
    }

    const double t_post = now_ms();            // (the stream is idle here only on the tensor path; good enough for a breakdown)
    int rcf = finalize_batch(c, jobs, q_stride, sort_cap, n_queries, max_nJ, g, knn_idx_host, knn_dist_host);
    if (rcf != R3DM_OK) return rcf;
    c->stats.ms_wall_match_post += now_ms() - t_post;
    if (r3dm_dev_knob("R3DM_MATCH_TIMING", 0))
        fprintf(stderr, "run_match_batch: prepare %.2f ms, launch .. exact scan issued %.2f ms, finalize %.2f ms\n", t_dbg1 - t_dbg0, t_post - t_dbg1, now_ms() - t_post);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_match_kernels += ms;
    c->stats.n_match_launches += 1;
    c->stats.n_pairs += P;
    c->stats.n_queries += n_queries;
    c->stats.n_exact_fallback += n_fallback;
    c->stats.algorithmic_flops += flops;
    c->stats.algorithmic_bytes += bytes;
    return R3DM_OK;
}

static int r3dm_match_pairs_impl(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs,
                                float dist_ratio, int squared_metric, r3dm_graph** out)
{
    if (!c || !out || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    R3DM_HIP(c, hipSetDevice(c->device));
    c->stats = r3dm_stats{};
    const double t_call = now_ms();
    // Matcher_Regions::Match: pairs whose views are missing, empty or of different region types are skipped
    std::vector<PairJob> jobs;
    jobs.reserve(n_pairs);
    for (uint64_t p = 0; p < n_pairs; ++p) {
        const uint32_t I = pairs_ij[2 * p], J = pairs_ij[2 * p + 1];
        auto a = c->slot_of.find(I), b = c->slot_of.find(J);
        if (a == c->slot_of.end() || b == c->slot_of.end()) { c->err = "pair references an unregistered view"; return R3DM_ERR_INVALID; }
        const HostImage& A = *c->imgs[a->second];
        const HostImage& B = *c->imgs[b->second];
        if (A.n == 0 || B.n == 0 || A.dtype != B.dtype || A.dim != B.dim) continue;
        jobs.push_back({I, J, a->second, b->second});
    }
    std::sort(jobs.begin(), jobs.end(), [](const PairJob& x, const PairJob& y) { return x.I != y.I ? x.I < y.I : x.J < y.J; });
    jobs.erase(std::unique(jobs.begin(), jobs.end(), [](const PairJob& x, const PairJob& y) { return x.I == y.I && x.J == y.J; }), jobs.end());

    auto g = std::unique_ptr<r3dm_graph>(new r3dm_graph());
    g->offsets.push_back(0);
    const float R = squared_metric ? dist_ratio * dist_ratio : dist_ratio;

    // batches: same (dtype, dim) and a bounded nn_idx footprint
    size_t start = 0;
    while (start < jobs.size()) {
        const HostImage& F = *c->imgs[jobs[start].sI];
        size_t end = start;
        uint64_t slots = 0;
        uint32_t max_n = 0;
        while (end < jobs.size()) {
            const HostImage& A = *c->imgs[jobs[end].sI];
            if (A.dtype != F.dtype || A.dim != F.dim) break;
            const uint32_t mn = std::max(max_n, c->imgs[jobs[end].sJ]->n);
            const uint64_t s = (uint64_t)(end - start + 1) * ((mn + 31) / 32 * 32);
            // <= 3 GiB of nn_idx per batch; one 256-thread workgroup per >= 128 queries keeps the dispatch below 2^32 work-items,
            // one workgroup per pair in the finaliser / exact scan below kMaxBlocksOf256
            if (end > start && (s * 4 > (3ull << 30) || end - start >= kMaxBlocksOf256 - 8)) break;
            max_n = mn; slots = s; ++end;
        }
        (void)slots;
        std::vector<PairJob> batch(jobs.begin() + start, jobs.begin() + end);
        int rc = run_match_batch(c, batch, R, g.get(), nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    c->stats.ms_wall_match = now_ms() - t_call;
    *out = g.release();
    return R3DM_OK;
}

extern "C" int r3dm_match_pairs(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs,
                                float dist_ratio, int squared_metric, r3dm_graph** out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_match_pairs_impl(c, pairs_ij, n_pairs, dist_ratio, squared_metric, out); });
}

static int r3dm_knn2_impl(r3dm_ctx* c, const void* dataset, uint32_t n_dataset, const void* query, uint32_t n_query,
                         uint32_t dim, r3dm_dtype dtype, int32_t* out_idx, float* out_dist)
{
    if (!c || !dataset || !query || !out_idx || !out_dist || dim == 0) return R3DM_ERR_INVALID;
    if (n_query < 1 || n_dataset < 2) return R3DM_ERR_INVALID;      // ArrayMatcherBruteForce: NN > nbRows / nbQuery < 1
    if (dtype == R3DM_BIN && !(((dim + 3) / 4) == 8 || ((dim + 3) / 4) == 16)) return R3DM_ERR_UNSUPPORTED;
    R3DM_HIP(c, hipSetDevice(c->device));
    // two private slots at the end of the table (never visible through view ids)
    const uint32_t s0 = (uint32_t)c->imgs.size();
    c->imgs.emplace_back(new HostImage());
    c->imgs.emplace_back(new HostImage());
    int rc = stage_into_slot(c, s0, 0, 0, 0, dataset, n_dataset, dim, dtype, nullptr);
    if (rc == R3DM_OK) rc = stage_into_slot(c, s0 + 1, 0, 0, 0, query, n_query, dim, dtype, nullptr);
    if (rc == R3DM_OK) {
        std::vector<PairJob> jobs{{0, 1, s0, s0 + 1}};
        const r3dm_stats keep = c->stats;
        rc = run_match_batch(c, jobs, 1.0f, nullptr, out_idx, out_dist);
        const uint64_t int_launches = c->stats.n_integer_mfma - keep.n_integer_mfma;
        const uint64_t split_launches = c->stats.n_split_mfma - keep.n_split_mfma;
        const uint64_t counts_launches = c->stats.n_counts_mfma - keep.n_counts_mfma;
        const uint64_t ham_launches = c->stats.n_hamming_mfma - keep.n_hamming_mfma;
        const uint64_t fb = c->stats.n_exact_fallback - keep.n_exact_fallback;
        c->stats = keep;
        c->stats.n_integer_mfma = int_launches;           // which tiles this call ran on (r3dm_set_integer_mfma / r3dm_set_split_mfma)
        c->stats.n_split_mfma = split_launches; c->stats.n_hamming_mfma = ham_launches; c->stats.n_counts_mfma = counts_launches;
        c->stats.n_exact_fallback = fb;                   // ... and how many of its queries went through the exact scan
    }
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
}

extern "C" int r3dm_knn2(r3dm_ctx* c, const void* dataset, uint32_t n_dataset, const void* query, uint32_t n_query,
                         uint32_t dim, r3dm_dtype dtype, int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_knn2_impl(c, dataset, n_dataset, query, n_query, dim, dtype, out_idx, out_dist); });
}

// ------------------------------------------------------------------------------------------------
// ArrayMatcher::Build / SearchNeighbours with the dataset staged ONCE (the reference builds per I and searches per J:
// /root/reference/src/R3DComputeMatches.cpp:462-479, plugin contract src/utils/matcher_kgraph.h:120-166,205-251)
// ------------------------------------------------------------------------------------------------
static int r3dm_index_create_impl(r3dm_ctx* c, const void* dataset, uint32_t n_dataset, uint32_t dim, r3dm_dtype dtype, r3dm_index** out)
{
    if (!c || !out || !dataset || dim == 0 || n_dataset < 1) return R3DM_ERR_INVALID;
    *out = nullptr;
    if (dtype != R3DM_F32 && dtype != R3DM_U8 && dtype != R3DM_BIN) return R3DM_ERR_INVALID;
    if (n_dataset >= (1u << 22)) { c->err = "more than 4M rows in one index"; return R3DM_ERR_UNSUPPORTED; }
    if (dtype == R3DM_BIN && !(((dim + 3) / 4) == 8 || ((dim + 3) / 4) == 16)) return R3DM_ERR_UNSUPPORTED;
    R3DM_HIP(c, hipSetDevice(c->device));
    auto ix = std::unique_ptr<r3dm_index>(new (std::nothrow) r3dm_index());
    if (!ix) return R3DM_ERR_NOMEM;
    const uint32_t s0 = (uint32_t)c->imgs.size();
    c->imgs.emplace_back(new HostImage());
    int rc = stage_into_slot(c, s0, 0, 0, 0, dataset, n_dataset, dim, dtype, nullptr);
    // Build stages what searches will read: the row-major rows beside the tiles (an index is one dataset: memory is not the concern,
    // a search from any context must not have to add to it) and the layouts of the paths that are switched on; a path switched on
    // later adds its layout on first use, under the index's lock
    if (rc == R3DM_OK) rc = ensure_layouts(c, {s0}, dtype == R3DM_BIN ? (c->hamming_mfma ? kLayBin8 : 0u)
                                                                     : (kLayRows | (c->integer_mfma ? kLayBf16 : 0u) | (c->split_mfma ? (kLayCounts | kLaySplit) : 0u)));
    if (rc == R3DM_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { c->err = std::string("r3dm_index_create: ") + hipGetErrorString(e); rc = R3DM_ERR_HIP; }
    }
    if (rc == R3DM_OK) {
        ix->device = c->device;
        ix->img = *c->imgs[s0];                   // the index takes the buffers over ...
        *c->imgs[s0] = HostImage();               // ... and the private slot forgets them
        *out = ix.release();
    } else c->imgs[s0]->release();
    c->imgs.pop_back();
    return rc;
}

extern "C" int r3dm_index_create(r3dm_ctx* c, const void* dataset, uint32_t n_dataset, uint32_t dim, r3dm_dtype dtype, r3dm_index** out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_index_create_impl(c, dataset, n_dataset, dim, dtype, out); });
}

extern "C" void r3dm_index_destroy(r3dm_index* ix)
{
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    ix->img.release();
    delete ix;
}

static int r3dm_index_knn2_impl(r3dm_ctx* c, const r3dm_index* ix, const void* query, uint32_t n_query, int32_t* out_idx, float* out_dist)
{
    if (!c || !ix || !query || !out_idx || !out_dist) return R3DM_ERR_INVALID;
    if (n_query < 1 || ix->img.n < 2) return R3DM_ERR_INVALID;          // ArrayMatcherBruteForce: NN > nbRows / nbQuery < 1
    if (ix->device != c->device) { c->err = "r3dm_index_knn2: the index lives on another device"; return R3DM_ERR_INVALID; }
    R3DM_HIP(c, hipSetDevice(c->device));
    const uint32_t s0 = (uint32_t)c->imgs.size();
    c->imgs.emplace_back(new HostImage());
    c->imgs.emplace_back(new HostImage());
    {
        std::lock_guard<std::mutex> lk(const_cast<r3dm_index*>(ix)->mu);
        *c->imgs[s0] = ix->img;                   // aliases of the index's buffers: mounted for this call only
    }
    c->imgs[s0]->borrowed = true;
    c->imgs[s0]->owner = const_cast<r3dm_index*>(ix);
    int rc = publish_entry(c, s0);
    if (rc == R3DM_OK) rc = stage_into_slot(c, s0 + 1, 0, 0, 0, query, n_query, ix->img.dim, ix->img.dtype, nullptr);
    if (rc == R3DM_OK) {
        std::vector<PairJob> jobs{{0, 1, s0, s0 + 1}};
        const r3dm_stats keep = c->stats;
        rc = run_match_batch(c, jobs, 1.0f, nullptr, out_idx, out_dist);
        const uint64_t int_launches = c->stats.n_integer_mfma - keep.n_integer_mfma;
        const uint64_t split_launches = c->stats.n_split_mfma - keep.n_split_mfma;
        const uint64_t counts_launches = c->stats.n_counts_mfma - keep.n_counts_mfma;
        const uint64_t ham_launches = c->stats.n_hamming_mfma - keep.n_hamming_mfma;
        c->stats = keep;
        c->stats.n_integer_mfma = int_launches; c->stats.n_split_mfma = split_launches; c->stats.n_hamming_mfma = ham_launches; c->stats.n_counts_mfma = counts_launches;
    }
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
}

extern "C" int r3dm_index_knn2(r3dm_ctx* c, const r3dm_index* ix, const void* query, uint32_t n_query, int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_index_knn2_impl(c, ix, query, n_query, out_idx, out_dist); });
}

// ------------------------------------------------------------------------------------------------
// approximate matching: graph index + graph search (kernels_ann.hip)
// ------------------------------------------------------------------------------------------------
extern "C" int r3dm_kgraph_preset(int preset, r3dm_kgraph_params* out)
{
    if (!out) return R3DM_ERR_INVALID;
    // src/R3DComputeMatches.cpp:844-873: (K, L, recall, P) = default (16, 24, .99, 10); 0: (2, 20, .6, 2); 1: (16, 24, .2, 6);
    // 2: (16, 24, .8, 12).  NN-descent keeps between S and L neighbours per row depending on how far it converged (its
    // recall target); the exact index has no such knob, so index_K takes the pool length L of the preset.
    r3dm_kgraph_params k{};
    k.search_S = 10; k.seed = 1998;
    switch (preset) {
        case 0:  k.index_K = 20; k.search_P = 2;  break;
        case 1:  k.index_K = 24; k.search_P = 6;  break;
        case 2:  k.index_K = 24; k.search_P = 12; break;
        default: k.index_K = 24; k.search_P = 10; break;
    }
    *out = k;
    return R3DM_OK;
}

// The approximate arms of the reference's dispatch (src/R3DComputeMatches.cpp:2035-2062) all trade recall for speed with a
// different index each (FLANN kd-trees, KGraph, MRPT random-projection trees, HNSW); none of them is reproducible bit for bit
// (random trees / seeds / thread schedules), so parity with any of them is recall.  The HNSW arms have their own matcher
// (api_hnsw.cpp: hnswlib's search, bit-exact on a reference-built index); this table is how the one deterministic graph matcher
// serves an arm when the host asks for the fastest matcher of at least the arm's recall (the facade's default policy), and the
// arms whose index is not built here at all (FLANN, MRPT):
//   1..3  kgraph_match presets                        -> fast / medium / precise
//   6..8  hnsw_match presets (:533-565)               -> fast / medium / precise  (reference-built HNSW on tests/golden/
//                                                        ann_hnsw_ref.npz: 0.573 / 0.933 / 0.975 recall@1; here 0.851 / 0.947 / 0.980)
//   5     mrpt_match (:453-460, targetRecall_ 0.8)     -> medium (0.947)
//   0     Matcher_Regions(ANN_L2): FLANN kd-trees     -> precise
// 4 and 9 are the exhaustive arms (r3dm_match_pairs) and are not ANN.
extern "C" int r3dm_ann_params_for_algorithm(int matching_algorithm, r3dm_kgraph_params* out)
{
    switch (matching_algorithm) {
        case 1: case 6: return r3dm_kgraph_preset(0, out);
        case 2: case 7: case 5: return r3dm_kgraph_preset(1, out);
        case 3: case 8: case 0: return r3dm_kgraph_preset(2, out);
        default: return R3DM_ERR_INVALID;
    }
}

static int check_kgraph_params(r3dm_ctx* c, const r3dm_kgraph_params* kp)
{
    if (!kp) return R3DM_ERR_INVALID;
    if (kp->index_K < 1 || kp->index_K > kAnnMaxK || kp->search_P < 2 || kp->search_P > 61 || kp->search_S < 1 || kp->search_S > 16) {
        c->err = "kgraph parameters out of range (index_K 1..32, search_P 2..61, search_S 1..16)";
        return R3DM_ERR_INVALID;
    }
    return R3DM_OK;
}

extern "C" int r3dm_exhaustive_is_faster(const r3dm_ctx* c)
{
    if (!c) return 0;
    if (sync_view_stats(const_cast<r3dm_ctx*>(c)) != R3DM_OK) return 0;        // (integer-valued? is a statistic of the staging kernel)
    bool any = false;
    for (const auto& up : c->imgs) {
        if (!up || !up->live) continue;
        const HostImage& h = *up;
        any = true;
        if (h.dtype == R3DM_BIN) return 0;                                   // no graph matcher for Hamming anyway
        if (!h.not_integer) return 0;                                        // integer-valued rows: the byte-row graph search is the faster one
        if (!has_tensor_kernel(kernel_G_for(h.dim))) return 0;               // other lengths run the one-workgroup-per-query exact scan
        if (h.n > 32768u) return 0;
    }
    return any ? 1 : 0;
}

// compact copy of a view's rows for the search's gathers: bytes for integers 0 .. 255 (ImgDev::ann_rows8), bf16 for other integers
// of magnitude <= 256 (ImgDev::ann_rows16), nothing otherwise.  R3DM_ANN_ROWS16 (developer build): 0 = never, 1 = bf16 only,
// 2 = bytes too, 3 (the product) = and integer dot products when both views of every pair are bytes.
static int stage_compact_rows(r3dm_ctx* c, HostImage& h)
{
    const int compact = r3dm_dev_knob("R3DM_ANN_ROWS16", 3);
    const bool ints = h.dtype != R3DM_BIN && !h.not_integer;
    h.ann_rows16.release(); h.ann_rows8.release();
    if (ints && !h.has_negative && h.max_abs <= 255.0f && (h.dim & 15u) == 0 && compact >= 2) {
        R3DM_HIP(c, h.ann_rows8.ensure((size_t)h.n * h.dim + kSlackBytes));
        R3DM_HIP(c, launch_ann_rows8(c->stream, h.rows.as<float>(), h.ann_rows8.as<uint8_t>(), (size_t)h.n * h.dim));
    } else if (ints && h.max_abs <= 256.0f && (h.dim & 7u) == 0 && compact >= 1) {
        R3DM_HIP(c, h.ann_rows16.ensure((size_t)h.n * h.dim * 2 + kSlackBytes));
        R3DM_HIP(c, launch_ann_rows16(c->stream, h.rows.as<float>(), h.ann_rows16.as<uint16_t>(), (size_t)h.n * h.dim));
    }
    h.compact_ready = true;
    return R3DM_OK;
}

// the query side of the integer-dot-product search: views that are only ever J hold no index, but may hold the byte copy
static int ensure_compact_rows(r3dm_ctx* c, std::vector<uint32_t> slots)
{
    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    std::vector<const void*> ptrs;
    ptrs.reserve(2 * slots.size());                          // must outlive the asynchronous copies: no reallocation below
    bool any = false;
    for (uint32_t s : slots) {
        HostImage& h = *c->imgs[s];
        if (h.compact_ready) continue;
        int rc = stage_compact_rows(c, h);
        if (rc != R3DM_OK) return rc;
        ptrs.push_back(h.ann_rows16.p); ptrs.push_back(h.ann_rows8.p);
        R3DM_HIP(c, hipMemcpyAsync((void*)&(c->d_imgs.as<ImgDev>() + s)->ann_rows16, &ptrs[ptrs.size() - 2], 2 * sizeof(void*),
                                   hipMemcpyHostToDevice, c->stream));
        any = true;
    }
    if (any) R3DM_HIP(c, hipStreamSynchronize(c->stream));
    return R3DM_OK;
}

// builds the graph index of every listed slot that does not hold one for this K
int ensure_ann_indices(r3dm_ctx* c, std::vector<uint32_t> slots, uint32_t K)
{
    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    std::vector<uint32_t> todo;
    for (uint32_t s : slots) if (c->imgs[s]->ann_K != K) todo.push_back(s);
    if (todo.empty()) return R3DM_OK;
    // The index is the EXACT K-NN graph (+ reverse edges): an all-pairs scan of the view against itself, O(n^2 dim) -- 1.1-4.4 ms per
    // 16 k-row view, the right builder at every size BASELINE names, and the wrong one far beyond: at R3DM_KGRAPH_MAX_ROWS rows it is
    // ~0.3 s per view and grows fourfold per doubling.  The reference's NN-descent (kgraph.cpp:703-999) is not built here; a larger view
    // is refused by name rather than indexed silently in quadratic time (the exhaustive matcher serves it: r3dm_match_pairs).
    for (uint32_t s : todo)
        if (c->imgs[s]->n > R3DM_KGRAPH_MAX_ROWS) {
            c->err = "kgraph index: view of " + std::to_string(c->imgs[s]->n) + " rows exceeds R3DM_KGRAPH_MAX_ROWS (" + std::to_string(R3DM_KGRAPH_MAX_ROWS) +
                     "): the exact K-NN graph build is quadratic in the rows";
            return R3DM_ERR_UNSUPPORTED;
        }
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    size_t start = 0;
    while (start < todo.size()) {
        // chunk: bounded scratch (fwd + rev keys: 16 B per edge slot)
        size_t end = start, bytes = 0;
        uint32_t max_n = 0;
        const uint32_t dim = c->imgs[todo[start]]->dim;
        while (end < todo.size() && end - start < 256) {
            const HostImage& h = *c->imgs[todo[end]];
            if (h.dim != dim) break;
            const size_t need = (size_t)h.n * K * 16 + (size_t)h.n * 12 + 64;
            if (end > start && bytes + need > (4ull << 30)) break;
            bytes += need; max_n = std::max(max_n, h.n); ++end;
        }
        R3DM_HIP(c, c->a_scratch.ensure(bytes));
        R3DM_HIP(c, hipMemsetAsync(c->a_scratch.p, 0, bytes, c->stream));
        std::vector<AnnBuildJob> jobs;
        bool all_rows8 = r3dm_dev_knob("R3DM_ANN_ROWS16", 3) >= 3;
        unsigned char* cur = c->a_scratch.as<unsigned char>();
        for (size_t k = start; k < end; ++k) {
            HostImage& h = *c->imgs[todo[k]];
            R3DM_HIP(c, h.ann_adj.ensure((size_t)h.n * kAnnDeg * 4));
            R3DM_HIP(c, h.ann_deg.ensure((size_t)h.n * 4));
            AnnBuildJob j{};
            j.slot = todo[k];
            j.fwd = (unsigned long long*)cur; cur += (size_t)h.n * K * 8;
            j.rev = (unsigned long long*)cur; cur += (size_t)h.n * K * 8;
            j.rev_cnt = (uint32_t*)cur; cur += (size_t)h.n * 4;
            j.rev_cur = (uint32_t*)cur; cur += (size_t)h.n * 4;
            j.rev_off = (uint32_t*)cur; cur += (size_t)h.n * 4 + 64;
            j.adj = h.ann_adj.as<uint32_t>(); j.deg = h.ann_deg.as<uint32_t>();
            if (!h.compact_ready) { const int rcc = stage_compact_rows(c, h); if (rcc != R3DM_OK) return rcc; }
            j.rows8 = h.ann_rows8.as<uint8_t>();
            all_rows8 = all_rows8 && j.rows8 != nullptr;
            jobs.push_back(j);
        }
        R3DM_HIP(c, c->a_jobs.ensure(jobs.size() * sizeof(AnnBuildJob)));
        R3DM_HIP(c, hipMemcpyAsync(c->a_jobs.p, jobs.data(), jobs.size() * sizeof(AnnBuildJob), hipMemcpyHostToDevice, c->stream));
        AnnBuildParams bp{};
        bp.imgs = c->d_imgs.as<ImgDev>(); bp.jobs = c->a_jobs.as<AnnBuildJob>(); bp.K = K;
        hipError_t e = launch_ann_build(c->stream, bp, (uint32_t)jobs.size(), max_n, dim, all_rows8 && dim <= 256 && (dim & 15u) == 0);
        if (e == hipErrorInvalidValue) { c->err = "no graph-index kernel for this descriptor length (dim % 4 != 0 or too long)"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, e);
        static_assert(offsetof(ImgDev, ann_deg) == offsetof(ImgDev, ann_adj) + sizeof(void*) &&
                      offsetof(ImgDev, ann_rows16) == offsetof(ImgDev, ann_adj) + 2 * sizeof(void*) &&
                      offsetof(ImgDev, ann_rows8) == offsetof(ImgDev, ann_adj) + 3 * sizeof(void*), "index pointers are set with one copy");
        std::vector<const void*> ptrs(4 * (end - start));      // must outlive the asynchronous copies
        for (size_t k = start; k < end; ++k) {
            HostImage& h = *c->imgs[todo[k]];
            const void** q = &ptrs[4 * (k - start)];
            q[0] = h.ann_adj.p; q[1] = h.ann_deg.p; q[2] = h.ann_rows16.p; q[3] = h.ann_rows8.p;
            R3DM_HIP(c, hipMemcpyAsync((void*)&(c->d_imgs.as<ImgDev>() + todo[k])->ann_adj, q, 4 * sizeof(void*),
                                       hipMemcpyHostToDevice, c->stream));
            h.ann_K = K;
        }
        R3DM_HIP(c, hipStreamSynchronize(c->stream));          // jobs / ptrs are host temporaries; scratch is reused
        start = end;
    }
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_build += ms;
    c->stats.n_ann_built += todo.size();
    return R3DM_OK;
}

// graph search + ratio test over `jobs` (all of one dim; every sI holds an index), results appended to g in job order
static int run_ann_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio_R, const r3dm_kgraph_params& kp,
                         r3dm_graph* g, int32_t* knn_idx_host, float* knn_dist_host)
{
    const uint32_t P = (uint32_t)jobs.size();
    if (P == 0) return R3DM_OK;
    uint32_t max_nJ = 0, max_nI = 0;
    uint64_t n_queries = 0;
    for (const PairJob& j : jobs) {
        max_nI = std::max(max_nI, c->imgs[j.sI]->n);
        max_nJ = std::max(max_nJ, c->imgs[j.sJ]->n);
        n_queries += c->imgs[j.sJ]->n;
    }
    const uint32_t dim = c->imgs[jobs[0].sI]->dim;
    const uint32_t q_stride = std::max<uint32_t>(32, (max_nJ + 31) / 32 * 32);
    const uint32_t sort_cap = std::min<uint32_t>(16384, std::max<uint32_t>(8, next_pow2(q_stride)));   // LDS budget; larger views may spill
    std::vector<uint2> hp(P), hid(P);
    for (uint32_t p = 0; p < P; ++p) { hp[p] = make_uint2(jobs[p].sI, jobs[p].sJ); hid[p] = make_uint2(jobs[p].I, jobs[p].J); }
    R3DM_HIP(c, c->d_pairs.ensure(sizeof(uint2) * P));
    R3DM_HIP(c, c->a_ids.ensure(sizeof(uint2) * P));
    R3DM_HIP(c, hipMemcpyAsync(c->d_pairs.p, hp.data(), sizeof(uint2) * P, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->a_ids.p, hid.data(), sizeof(uint2) * P, hipMemcpyHostToDevice, c->stream));
    const uint64_t total_slots = (uint64_t)P * q_stride;
    R3DM_HIP(c, c->d_nn.ensure((size_t)total_slots * 4));
    R3DM_HIP(c, c->d_cnt.ensure(64));
    R3DM_HIP(c, hipMemsetAsync(c->d_cnt.p, 0, 64, c->stream));
    if (knn_idx_host) {
        R3DM_HIP(c, c->d_knn_idx.ensure((size_t)total_slots * 8));
        R3DM_HIP(c, c->d_knn_dist.ensure((size_t)total_slots * 8));
    }
    AnnSearchParams sp{};
    sp.imgs = c->d_imgs.as<ImgDev>(); sp.pairs = c->d_pairs.as<uint2>(); sp.pair_ids = c->a_ids.as<uint2>();
    sp.n_pairs = P; sp.q_stride = q_stride;
    sp.P = kp.search_P; sp.S = kp.search_S; sp.pool_cap = 2 + kp.search_P; sp.seed = kp.seed; sp.ratio_R = ratio_R;
    sp.nn_idx = c->d_nn.as<uint32_t>();
    sp.knn_idx = knn_idx_host ? c->d_knn_idx.as<int32_t>() : nullptr;
    sp.knn_dist = knn_idx_host ? c->d_knn_dist.as<float>() : nullptr;
    sp.n_comps = reinterpret_cast<unsigned long long*>(c->d_cnt.as<uint32_t>() + 4);
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    bool rows16 = true, rows8 = true, dot8 = r3dm_dev_knob("R3DM_ANN_ROWS16", 3) >= 3;   // every indexed view of the batch holds that compact row copy
    for (const PairJob& j : jobs) {
        const HostImage& A = *c->imgs[j.sI];
        const HostImage& B = *c->imgs[j.sJ];
        rows16 = rows16 && A.compact_ready && A.ann_rows16.p != nullptr;
        rows8 = rows8 && A.compact_ready && A.ann_rows8.p != nullptr;
        dot8 = dot8 && B.compact_ready && B.ann_rows8.p != nullptr;        // ... and every query view its byte copy
    }
    dot8 = dot8 && rows8 && dim <= 256 && (dim & 15u) == 0;
    // descriptor lengths 132 .. 144 (nine float4 per lane: LIOP-144) have no compact-row instantiation of the search kernel
    // (launch_ann_search): integer-valued views of that length gather the f32 rows
    if ((dim / 4 + 3) / 4 == 9) { dot8 = false; rows8 = false; rows16 = false; }
    hipError_t e = launch_ann_search(c->stream, sp, max_nJ, max_nI, dim, dot8 ? 3 : rows8 ? 2 : (rows16 ? 1 : 0));
    if (e == hipErrorInvalidValue) { c->err = "graph search: unsupported descriptor length / view size / parameters"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, e);
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    unsigned long long comps = 0;
    R3DM_HIP(c, hipMemcpyAsync(&comps, sp.n_comps, 8, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    const double t_post = now_ms();
    int rc = finalize_batch(c, jobs, q_stride, sort_cap, n_queries, max_nJ, g, knn_idx_host, knn_dist_host);
    if (rc != R3DM_OK) return rc;
    c->stats.ms_wall_match_post += now_ms() - t_post;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_ann_search += ms;
    c->stats.n_ann_dist += comps;
    c->stats.n_ann_rows16 += (rows16 && !rows8) ? 1 : 0;
    c->stats.n_ann_rows8 += rows8 ? 1 : 0;
    c->stats.n_ann_dot8 += dot8 ? 1 : 0;
    c->stats.n_match_launches += 1;
    c->stats.n_pairs += P;
    c->stats.n_queries += n_queries;
    return R3DM_OK;
}

static int r3dm_match_pairs_kgraph_impl(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                       const r3dm_kgraph_params* kp, r3dm_graph** out)
{
    if (!c || !out || (n_pairs && !pairs_ij)) return R3DM_ERR_INVALID;
    *out = nullptr;
    int rc = check_kgraph_params(c, kp);
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
        if (A.dtype == R3DM_BIN || (A.dim & 3u)) { c->err = "kgraph matching needs F32/U8 descriptors with dim % 4 == 0"; return R3DM_ERR_UNSUPPORTED; }
        // KGraphImpl::search scans linearly when P >= n (kgraph.cpp:415-421); here every small index is scanned
        if (A.n < kAnnMinRows || kp->search_P >= A.n) small_jobs.push_back({I, J, a->second, b->second});
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
        // the graph build and the graph search gather row-major rows (f32, or the compact copies made from them)
        for (const PairJob& j : ann_jobs) { slots.push_back(j.sI); slots.push_back(j.sJ); }
        rc = ensure_layouts(c, slots, kLayRows);
        if (rc != R3DM_OK) return rc;
        slots.clear();
        for (const PairJob& j : ann_jobs) slots.push_back(j.sI);
        rc = ensure_ann_indices(c, slots, kp->index_K);
        if (rc != R3DM_OK) return rc;
        slots.clear();
        for (const PairJob& j : ann_jobs) slots.push_back(j.sJ);
        rc = ensure_compact_rows(c, slots);
        if (rc != R3DM_OK) return rc;
    }
    size_t start = 0;
    while (start < ann_jobs.size()) {
        const uint32_t dim = c->imgs[ann_jobs[start].sI]->dim;
        size_t end = start;
        uint32_t max_n = 0;
        while (end < ann_jobs.size()) {
            if (c->imgs[ann_jobs[end].sI]->dim != dim) break;
            const uint32_t mn = std::max(max_n, c->imgs[ann_jobs[end].sJ]->n);
            const uint64_t s = (uint64_t)(end - start + 1) * ((mn + 31) / 32 * 32);
            // <= 3 GiB of nn_idx, and one workgroup per 4 queries: the dispatch must stay below 2^32 work-items (kMaxBlocksOf256)
            if (end > start && (s * 4 > (3ull << 30) || s / 4 > kMaxBlocksOf256 - 4096)) break;
            max_n = mn; ++end;
        }
        std::vector<PairJob> batch(ann_jobs.begin() + start, ann_jobs.begin() + end);
        rc = run_ann_batch(c, batch, R, *kp, &ga, nullptr, nullptr);
        if (rc != R3DM_OK) return rc;
        start = end;
    }
    start = 0;
    while (start < small_jobs.size()) {                      // few and tiny: one batch per (dtype, dim) run
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

extern "C" int r3dm_match_pairs_kgraph(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
                                       const r3dm_kgraph_params* kp, r3dm_graph** out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_match_pairs_kgraph_impl(c, pairs_ij, n_pairs, dist_ratio, kp, out); });
}

static int r3dm_kgraph_knn2_impl(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                                uint32_t dim, const r3dm_kgraph_params* kp, uint32_t pair_i, uint32_t pair_j,
                                int32_t* out_idx, float* out_dist)
{
    if (!c || !dataset || !query || !out_idx || !out_dist || dim == 0) return R3DM_ERR_INVALID;
    if (n_query < 1 || n_dataset < 2) return R3DM_ERR_INVALID;
    int rc = check_kgraph_params(c, kp);
    if (rc != R3DM_OK) return rc;
    if (dim & 3u) { c->err = "kgraph matching needs dim % 4 == 0"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, hipSetDevice(c->device));
    const uint32_t s0 = (uint32_t)c->imgs.size();
    c->imgs.emplace_back(new HostImage());
    c->imgs.emplace_back(new HostImage());
    rc = stage_into_slot(c, s0, pair_i, 0, 0, dataset, n_dataset, dim, R3DM_F32, nullptr);
    if (rc == R3DM_OK) rc = stage_into_slot(c, s0 + 1, pair_j, 0, 0, query, n_query, dim, R3DM_F32, nullptr);
    const r3dm_stats keep = c->stats;
    if (rc == R3DM_OK) {
        std::vector<PairJob> jobs{{pair_i, pair_j, s0, s0 + 1}};
        if (n_dataset < kAnnMinRows || kp->search_P >= n_dataset) rc = run_match_batch(c, jobs, 1.0f, nullptr, out_idx, out_dist);
        else {
            rc = ensure_layouts(c, {s0, s0 + 1}, kLayRows);
            if (rc == R3DM_OK) rc = ensure_ann_indices(c, {s0}, kp->index_K);
            if (rc == R3DM_OK) rc = ensure_compact_rows(c, {s0 + 1});
            if (rc == R3DM_OK) rc = run_ann_batch(c, jobs, 1.0f, *kp, nullptr, out_idx, out_dist);
        }
    }
    const uint64_t r16 = c->stats.n_ann_rows16 - keep.n_ann_rows16, r8 = c->stats.n_ann_rows8 - keep.n_ann_rows8, d8 = c->stats.n_ann_dot8 - keep.n_ann_dot8,
                   evals = c->stats.n_ann_dist - keep.n_ann_dist;
    c->stats = keep;
    c->stats.n_ann_rows16 = r16; c->stats.n_ann_rows8 = r8; c->stats.n_ann_dot8 = d8; c->stats.n_ann_dist = evals;   // like r3dm_knn2: which rows this call gathered, how many evaluations
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
}

extern "C" int r3dm_kgraph_knn2(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
                                uint32_t dim, const r3dm_kgraph_params* kp, uint32_t pair_i, uint32_t pair_j,
                                int32_t* out_idx, float* out_dist)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_kgraph_knn2_impl(c, dataset, n_dataset, query, n_query, dim, kp, pair_i, pair_j, out_idx, out_dist); });
}

extern "C" int r3dm_drop_indices(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    for (auto& h : c->imgs) if (h) { h->ann_K = 0; h->hnsw_M = 0; h->mrpt_trees = 0; }            // the device pointers stay valid until the rebuild replaces them
    return R3DM_OK;
}

extern "C" int r3dm_kgraph_index(r3dm_ctx* c, uint32_t view_id, uint32_t index_K, uint32_t* adj_out, uint32_t* deg_out)
{
    if (!c || !adj_out || !deg_out) return R3DM_ERR_INVALID;
    auto it = c->slot_of.find(view_id);
    if (it == c->slot_of.end()) { c->err = "unregistered view"; return R3DM_ERR_INVALID; }
    if (index_K < 1 || index_K > kAnnMaxK) return R3DM_ERR_INVALID;
    HostImage& h = *c->imgs[it->second];
    if (h.dtype == R3DM_BIN || (h.dim & 3u) || h.n < 2) { c->err = "kgraph index needs >= 2 F32/U8 rows with dim % 4 == 0"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, hipSetDevice(c->device));
    int rc = ensure_layouts(c, {it->second}, kLayRows);
    if (rc == R3DM_OK) rc = ensure_ann_indices(c, {it->second}, index_K);
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipMemcpy(adj_out, h.ann_adj.p, (size_t)h.n * kAnnDeg * 4, hipMemcpyDeviceToHost));
    R3DM_HIP(c, hipMemcpy(deg_out, h.ann_deg.p, (size_t)h.n * 4, hipMemcpyDeviceToHost));
    return R3DM_OK;
}

