// api_filter.cpp -- part of the host side of libr3dm.so: the C ABI declared in include/r3dm.h (see r3dm_ctx.hpp for the file map).
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 and src/Regard3DFeatures.cpp -- with every arithmetic stage running as
// HIP kernels on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"

#include <functional>
#include <memory>

// What filter_prepare leaves for the shared launch of the cooperative kernel (kernels_filter_coop.hip): the long pairs of this kind
struct CoopPlan {
    std::vector<uint32_t> G, len;      // slices and putative count of every cooperative pair of the kind, by pair index
    uint32_t slots = 0;                // sum of G
};

// a device buffer that lives as long as the closure that captured it (the trace / check buffers of the developer build)
struct SharedDevBuf { DevBuf b; ~SharedDevBuf() { b.release(); } };

// ------------------------------------------------------------------------------------------------
// geometric filter
// ------------------------------------------------------------------------------------------------
// model_kind 0 = fundamental matrix (GeometricFilter_FMatrix_AC), 1 = homography (GeometricFilter_HMatrix_AC)
//            2 = essential matrix (GeometricFilter_EMatrix_AC) + Regard3D's overlap rule (min_count / min_ratio)
// What a filter call leaves behind besides its graph: written by the call's own thread, folded into the context afterwards (three
// calls of r3dm_filter_FEH run side by side)
struct FilterCallOut {
    std::string err;
    double ms_kernels = 0.0, ms_wall = 0.0;
    std::vector<r3dm_pair_report> report;
    r3dm_graph* pending = nullptr;        // the graph under construction between filter_prepare and its collect
    ~FilterCallOut() { delete pending; }
};
#define FHIP(call)                                                                     \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            o.err = std::string(#call) + ": " + hipGetErrorString(e__);                \
            return R3DM_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)

// Everything in front of the launch (work items, tables, uploads on the context's stream, kernel parameters) and, as `collect`, everything
// behind it (copy back, per-pair report, acceptance rules, the filtered graph); collect runs once the launch has been waited for and is
// given its HIP-event time.  launch = false in fp_out.n_items == 0 (nothing to run: collect still delivers the empty graph).
static int filter_prepare(r3dm_ctx* c, FilterCallOut& o, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                          uint64_t seed, r3dm_ferror err_kind, int model_kind, r3dm_graph** out, double* F_out,
                          uint32_t min_count, float min_ratio, FilterParams& fp_out, CoopPlan& plan, std::function<int(float)>& collect,
                          const r3dm_match* dev_matches = nullptr /* the putative matches already uploaded by another kind of the call */)
{
    fp_out = FilterParams{};
    fp_out.model_kind = model_kind;                       // (also on the early returns: coop_launch_shared files the parameter blocks by kind)
    plan = CoopPlan{};
    if (!c || !putative || !out || max_iter == 0) return R3DM_ERR_INVALID;
    FilterBufs& B = c->fb[model_kind];
    const uint32_t SS = model_kind == 0 ? 7u : (model_kind == 1 ? 4u : 5u);          // Kernel::MINIMUM_SAMPLES
    *out = nullptr;
    FHIP(hipSetDevice(c->device));

    const double t_call = now_ms();
    const uint64_t NP = putative->pairs.size() / 2;
    // work items: pairs with more than SS putatives (ACRANSAC returns nothing for n <= MINIMUM_SAMPLES)
    std::vector<uint32_t> item_pair;
    std::vector<uint2> slots, ids;
    uint32_t max_m = 0;
    uint64_t sum_m = 0;
    for (uint64_t p = 0; p < NP; ++p) {
        const uint64_t m = putative->offsets[p + 1] - putative->offsets[p];
        if (m <= SS) continue;
        const uint32_t I = putative->pairs[2 * p], J = putative->pairs[2 * p + 1];
        auto a = c->slot_of.find(I), b = c->slot_of.find(J);
        if (a == c->slot_of.end() || b == c->slot_of.end()) { o.err = "filter: pair references an unregistered view"; return R3DM_ERR_INVALID; }
        const HostImage& A = *c->imgs[a->second];
        const HostImage& B = *c->imgs[b->second];
        if (!A.has_xy || !B.has_xy) { o.err = "filter: view registered without feature positions"; return R3DM_ERR_INVALID; }
        // ACKernelAdaptor normalises with 1 / sqrt(w h) and the NFA scale is D / A of image J: a view registered with a zero
        // width or height would turn every residual into NaN and the filter into a silent "no inliers"
        if (A.width == 0 || A.height == 0 || B.width == 0 || B.height == 0) {
            o.err = "filter: view " + std::to_string(A.width == 0 || A.height == 0 ? I : J) + " was registered without its image size (width / height = 0)";
            return R3DM_ERR_INVALID;
        }
        if (m > (1u << 22)) { o.err = "filter: more than 4M putative matches in one pair"; return R3DM_ERR_UNSUPPORTED; }
        // E_ACRobust: a pair whose views lack valid pinhole intrinsics is not estimated (and so not kept)
        if (model_kind == 2 && (!A.has_K || !B.has_K)) continue;
        item_pair.push_back((uint32_t)p);
        slots.push_back(make_uint2(a->second, b->second));
        ids.push_back(make_uint2(I, J));
        max_m = std::max<uint32_t>(max_m, (uint32_t)m);
        sum_m += m;
    }
    auto g = std::unique_ptr<r3dm_graph>(new r3dm_graph());
    g->offsets.push_back(0);
    const uint32_t NI = (uint32_t)item_pair.size();
    if (NI == 0) {
        r3dm_graph* empty = g.release();
        o.pending = empty;
        collect = [&o, out, empty, t_call](float) -> int { o.pending = nullptr; *out = empty; o.ms_wall = now_ms() - t_call; return R3DM_OK; };
        return R3DM_OK;
    }
    (void)sum_m;
    // [begin, end) of every item's putative list inside the full match array
    std::vector<uint64_t> begin_end(2 * (size_t)NI);
    for (uint32_t k = 0; k < NI; ++k) {
        const uint32_t p = item_pair[k];
        begin_end[2 * k] = putative->offsets[p];
        begin_end[2 * k + 1] = putative->offsets[p + 1];
    }
    // ---- long pairs run on the cooperative kernel (kernels_filter_coop.hip): G workgroups per batch of models, slices of the match list.
    // G follows the pair-length distribution: slices of 2048 .. 8192 matches, short enough that the long pairs of the call fill the
    // device about one and a half times over; collections of short pairs (C2: every pair below the threshold) keep the one-workgroup
    // kernel and its launch shape untouched.
    const uint32_t coop_min = (uint32_t)r3dm_dev_knob("R3DM_FILTER_COOP_MIN", 4096);      // pairs with more putatives; 0 = never
    const uint32_t coop_g_knob = (uint32_t)r3dm_dev_knob("R3DM_FILTER_COOP_G", 0);        // > 0: this many slices for every such pair
    std::vector<unsigned char> is_coop(NI, 0);
    std::vector<uint32_t> coop_items, coop_G, coop_slice, coop_hoff;
    uint32_t max_m_short = 0, coop_slots = 0;
    {
        uint64_t sum_long = 0;
        for (uint32_t k = 0; k < NI; ++k) {
            const uint64_t mk = begin_end[2 * k + 1] - begin_end[2 * k];
            if (coop_min && mk > coop_min && mk <= (uint64_t)kCoopMaxG * 65472u) { is_coop[k] = 1; sum_long += mk; }
            else max_m_short = std::max<uint32_t>(max_m_short, (uint32_t)mk);
        }
        const uint64_t fill = std::max<uint64_t>(1, (uint64_t)std::max(c->n_cu, 1) * 3 / 2);
        const uint32_t slice_target = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(2048, ((sum_long / fill + 511) / 512) * 512));
        for (uint32_t k = 0; k < NI; ++k) {
            if (!is_coop[k]) continue;
            const uint32_t mk = (uint32_t)(begin_end[2 * k + 1] - begin_end[2 * k]);
            uint32_t G = coop_g_knob ? coop_g_knob : (mk + slice_target - 1) / slice_target;
            G = std::max<uint32_t>(G, (mk + 65471u) / 65472u);                              // a slice counts in 16 bits
            G = std::min<uint32_t>(std::max<uint32_t>(G, 1u), kCoopMaxG);
            const uint32_t len = (((mk + G - 1) / G + 63) / 64) * 64;
            coop_items.push_back(k); coop_G.push_back(G); coop_slice.push_back(len); coop_hoff.push_back(coop_slots);
            coop_slots += G;
        }
    }
    uint32_t n_coop = (uint32_t)coop_items.size();
    // slice offsets of the per-item work arrays: multiples of 32 elements, room for m + 1 (kernels_filter.hip explains why)
    std::vector<uint64_t> soff(NI + 1, 0);
    for (uint32_t k = 0; k < NI; ++k) soff[k + 1] = soff[k] + ((begin_end[2 * k + 1] - begin_end[2 * k] + 1 + 31) / 32) * 32;
    const uint64_t n_slice = soff[NI];
    // the cooperative kernel addresses the points and the slice histograms through 32-bit buffer offsets: a call beyond them (about 67 M
    // putatives) runs every pair on the one-workgroup kernel, as every call did before the cooperative kernel existed
    if (n_coop && (32 * (uint64_t)n_slice >= 0x7FFFFFFFull || 4 * (uint64_t)coop_slots * kCoopB * 512 + 256 >= 0x7FFFFFFFull)) {
        for (uint32_t k = 0; k < NI; ++k) {
            is_coop[k] = 0;
            max_m_short = std::max<uint32_t>(max_m_short, (uint32_t)(begin_end[2 * k + 1] - begin_end[2 * k]));
        }
        coop_items.clear(); coop_G.clear(); coop_slice.clear(); coop_hoff.clear();
        coop_slots = 0; n_coop = 0;
    }
    if (model_kind == 0 && err_kind != R3DM_ERR_SYMMETRIC_EPIPOLAR) { o.err = "filter: only the symmetric epipolar error is implemented"; return R3DM_ERR_UNSUPPORTED; }
    // host tables in the reference's own float arithmetic (glibc log10f), see kernels_filter.hip
    std::vector<float> l10(max_m + 2), lck(max_m + 2);
    for (uint32_t k = 0; k <= max_m + 1; ++k) l10[k] = std::log10((float)k);
    for (uint32_t n = 0; n <= max_m + 1; ++n) {
        const uint32_t ks = SS;
        if (ks >= n) { lck[n] = 0.f; continue; }
        const uint32_t kk = (n - ks < ks) ? n - ks : ks;
        float r = 0.f;
        for (uint32_t i = 1; i <= kk; ++i) r += l10[n - i + 1] - l10[i];
        lck[n] = r;
    }
    const uint64_t n_match_total = putative->matches.size();
    FHIP(B.f_pairs.ensure(sizeof(uint2) * NI));
    FHIP(B.f_ids.ensure(sizeof(uint2) * NI));
    FHIP(B.f_offs.ensure(sizeof(uint64_t) * 2 * NI));
    if (!dev_matches) FHIP(B.f_matches.ensure(sizeof(r3dm_match) * std::max<uint64_t>(n_match_total, 1)));
    FHIP(B.f_inl_cnt.ensure(4 * (size_t)NI));
    FHIP(B.f_inl_idx.ensure(4 * (size_t)n_slice + 64));
    FHIP(B.f_soff.ensure(8 * (size_t)(NI + 1)));
    FHIP(hipMemcpyAsync(B.f_soff.p, soff.data(), 8 * (size_t)(NI + 1), hipMemcpyHostToDevice, c->stream));
    FHIP(B.f_F.ensure(72 * (size_t)NI));
    FHIP(B.f_thr.ensure(16 * (size_t)NI));
    FHIP(B.f_iters.ensure(8 * (size_t)NI));
    FHIP(B.f_log10.ensure(4 * l10.size()));
    FHIP(B.f_logck.ensure(4 * lck.size()));
    FHIP(hipMemcpyAsync(B.f_pairs.p, slots.data(), sizeof(uint2) * NI, hipMemcpyHostToDevice, c->stream));
    FHIP(hipMemcpyAsync(B.f_ids.p, ids.data(), sizeof(uint2) * NI, hipMemcpyHostToDevice, c->stream));
    FHIP(hipMemcpyAsync(B.f_offs.p, begin_end.data(), sizeof(uint64_t) * 2 * NI, hipMemcpyHostToDevice, c->stream));
    if (!dev_matches) FHIP(hipMemcpyAsync(B.f_matches.p, putative->matches.data(), sizeof(r3dm_match) * n_match_total, hipMemcpyHostToDevice, c->stream));
    FHIP(hipMemcpyAsync(B.f_log10.p, l10.data(), 4 * l10.size(), hipMemcpyHostToDevice, c->stream));
    FHIP(hipMemcpyAsync(B.f_logck.p, lck.data(), 4 * lck.size(), hipMemcpyHostToDevice, c->stream));
    FHIP(hipMemsetAsync(B.f_inl_cnt.p, 0, 4 * (size_t)NI, c->stream));

    FilterParams fp{};
    fp.imgs = c->d_imgs.as<ImgDev>();
    fp.pairs = B.f_pairs.as<uint2>(); fp.pair_ids = B.f_ids.as<uint2>();
    fp.offsets = B.f_offs.as<uint64_t>(); fp.matches = dev_matches ? dev_matches : B.f_matches.as<r3dm_match>();
    // LDS sort capacity: 8192 (x 12 B) fits beside the hypothesis buffer; pairs with more putatives sort in global scratch
    // (essential matrix: 4096, so that header + 16 hypotheses + sort buffers stay below 80 KB and two workgroups share a CU)
    // collections with long match lists (some pair above 4096 putatives: LDS admits one workgroup per CU anyway) run the 512-thread
    // variant of the kernel -- the same results, every pass over a pair's matches in half the trips
    const int wide_knob = r3dm_dev_knob("R3DM_FILTER_WIDE", -1);
    fp.wide = wide_knob >= 0 ? (uint32_t)(wide_knob != 0) : (max_m_short > 4096 ? 1u : 0u);
    fp.n_items = NI; fp.m_cap = std::min<uint32_t>((model_kind == 2 && !fp.wide) ? 4096 : 8192, std::max<uint32_t>(64, next_pow2(std::max(max_m_short, 1u))));
    fp.spill_keys = nullptr; fp.spill_idx = nullptr; fp.spill_off = nullptr;
    if (max_m_short > fp.m_cap || n_coop) {                  // (the cooperative kernel keeps every pair's sort lists in global memory)
        std::vector<uint64_t> soff(NI, 0);
        uint64_t tot = 0;
        for (uint32_t k = 0; k < NI; ++k) {
            const uint64_t mk = begin_end[2 * k + 1] - begin_end[2 * k];
            soff[k] = tot;
            if (is_coop[k]) tot += 2 * (uint64_t)next_pow2((uint32_t)mk);      // [sort | spare]: the bucket pass of the cooperative kernel's full evaluation
            else if (mk > fp.m_cap) tot += next_pow2((uint32_t)mk);
        }
        FHIP(B.f_spill.ensure(tot * 12 + NI * 8 + 64));
        unsigned char* base = B.f_spill.as<unsigned char>();
        FHIP(hipMemcpyAsync(base + tot * 12, soff.data(), NI * 8, hipMemcpyHostToDevice, c->stream));
        FHIP(hipStreamSynchronize(c->stream));       // `soff` leaves scope
        fp.spill_keys = reinterpret_cast<unsigned long long*>(base);
        fp.spill_idx = reinterpret_cast<uint32_t*>(base + tot * 8);
        fp.spill_off = reinterpret_cast<const uint64_t*>(base + tot * 12);
    }
    fp.precision_px = max_residual_px; fp.max_iter = max_iter; fp.seed = seed; fp.err_kind = (int)err_kind;
    fp.model_kind = model_kind;
    fp.kinv = nullptr;
    if (model_kind == 2) {
        std::vector<double> kinv(9 * c->imgs.size(), 0.0);
        for (size_t s = 0; s < c->imgs.size(); ++s)
            if (c->imgs[s] && c->imgs[s]->has_K) memcpy(&kinv[9 * s], c->imgs[s]->Kinv, 72);
        FHIP(B.f_kinv.ensure(kinv.size() * 8));
        FHIP(hipMemcpyAsync(B.f_kinv.p, kinv.data(), kinv.size() * 8, hipMemcpyHostToDevice, c->stream));
        FHIP(hipStreamSynchronize(c->stream));       // `kinv` leaves scope
        fp.kinv = B.f_kinv.as<double>();
    }
    fp.log10_tab = B.f_log10.as<float>(); fp.logc_k = B.f_logck.as<float>();
    fp.inl_count = B.f_inl_cnt.as<uint32_t>(); fp.inl_idx = B.f_inl_idx.as<uint32_t>();
    fp.F_out = B.f_F.as<double>(); fp.thr_nfa = B.f_thr.as<double>(); fp.iters = B.f_iters.as<uint32_t>();
    FHIP(B.f_scratch.ensure(40 * (size_t)n_slice + 256));
    fp.pts_scratch = B.f_scratch.as<double>();
    fp.pool_scratch = reinterpret_cast<uint32_t*>(B.f_scratch.as<unsigned char>() + 32 * (size_t)n_slice);
    fp.scratch_logc = reinterpret_cast<float*>(B.f_scratch.as<unsigned char>() + 36 * (size_t)n_slice);
    fp.soff = B.f_soff.as<uint64_t>();
    FHIP(B.f_la.ensure((size_t)NI * 1024 * 8 + 64));
    fp.la_tab = B.f_la.as<double>();
    { const int sc = r3dm_dev_knob("R3DM_FILTER_SCOUT", 1); fp.scout = (uint32_t)(sc < 0 ? 0 : sc > 5 ? 1 : sc) | ((uint32_t)r3dm_dev_knob("R3DM_FILTER_SCOUT_SUB", 0) << 8); }   // (developer build: 0 = every model through the full evaluation, 2 = the scout divides exactly, 3 = check mode with R3DM_FILTER_CHECK=1; A/B and parity)
    // launch order: the workgroup of a pair runs for a time roughly proportional to its putative count, and a C2 call has
    // ~1.5 x as many pairs as resident workgroups -- start the long ones first so the tail of the launch is short ones
    {
        static const int lpt = r3dm_dev_knob("R3DM_FILTER_LPT", 1);
        fp.order = nullptr;
        fp.n_short = NI - n_coop;
        if ((lpt && NI > 1) || n_coop) {
            std::vector<uint32_t> order;
            order.reserve(NI);
            for (uint32_t k = 0; k < NI; ++k) if (!is_coop[k]) order.push_back(k);
            if (lpt) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                return begin_end[2 * a + 1] - begin_end[2 * a] > begin_end[2 * b + 1] - begin_end[2 * b]; });
            FHIP(B.f_order.ensure(4 * (size_t)NI));
            if (!order.empty()) FHIP(hipMemcpyAsync(B.f_order.p, order.data(), 4 * order.size(), hipMemcpyHostToDevice, c->stream));
            FHIP(hipStreamSynchronize(c->stream));       // `order` leaves scope
            fp.order = B.f_order.as<uint32_t>();
        }
    }
    fp.n_coop = n_coop;
    fp.coop_workers = 0;
    if (n_coop) {
        // one allocation: [items | G | slice | hoff] [published records] [models] [batch matrices] [slice counts]
        // [slope tables] [T*] [slice histograms]
        auto up = [](size_t v) { return (v + 255) / 256 * 256; };
        const size_t small_bytes = up(16 * (size_t)n_coop);
        const size_t o_q = small_bytes, q_bytes = 0;
        const size_t o_pub = o_q + q_bytes, pub_bytes = up(64 * (size_t)n_coop);
        const size_t o_models = o_pub + pub_bytes, models_bytes = up(8 * (size_t)n_coop * filter_coop_model_doubles(model_kind));
        const size_t o_bm = o_models + models_bytes, bm_bytes = up(8 * (size_t)n_coop * kCoopB * 9);
        const size_t o_cnt = o_bm + bm_bytes, cnt_bytes = up(4 * (size_t)coop_slots * kCoopB);
        const size_t o_la = o_cnt + cnt_bytes, la_bytes = up(8 * (size_t)n_coop * 1024);
        const size_t o_ts = o_la + la_bytes, ts_bytes = up(8 * (size_t)n_slice + 64);
        const size_t o_hist = o_ts + ts_bytes, hist_bytes = up(4 * (size_t)coop_slots * kCoopB * 512);
        // (the kernel addresses the points and the slice histograms through 32-bit buffer offsets: checked above, kept as a guard)
        if (32 * (uint64_t)n_slice >= 0x7FFFFFFFull || hist_bytes >= 0x7FFFFFFFull) { o.err = "filter: putative graph too large for the cooperative kernel's buffer offsets"; return R3DM_ERR_UNSUPPORTED; }
        FHIP(B.f_coop.ensure(o_hist + hist_bytes));
        unsigned char* base = B.f_coop.as<unsigned char>();
        std::vector<uint32_t> stage(small_bytes / 4, 0u);
        memcpy(&stage[0], coop_items.data(), 4 * (size_t)n_coop);
        memcpy(&stage[n_coop], coop_G.data(), 4 * (size_t)n_coop);
        memcpy(&stage[2 * (size_t)n_coop], coop_slice.data(), 4 * (size_t)n_coop);
        memcpy(&stage[3 * (size_t)n_coop], coop_hoff.data(), 4 * (size_t)n_coop);
        FHIP(hipMemcpyAsync(base, stage.data(), 4 * stage.size(), hipMemcpyHostToDevice, c->stream));
        FHIP(hipMemsetAsync(base + o_pub, 0, pub_bytes, c->stream));
        plan.G = coop_G; plan.slots = coop_slots;
        for (uint32_t qi = 0; qi < n_coop; ++qi) plan.len.push_back((uint32_t)(begin_end[2 * coop_items[qi] + 1] - begin_end[2 * coop_items[qi]]));
        // logcombi(k, m) of every cooperative pair as a running prefix in the reference's float accumulation order (makelogcombi_n,
        // SURVEY.md A.5): pre[i] = pre[i-1] + (l10[m-i+1] - l10[i]), mirrored for k > m/2 -- the same operations, in the same order, as
        // thread 0 of the one-workgroup kernel performs (this translation unit is compiled with -ffp-contract=off like the kernels)
        if (B.pin_idx.ensure(4 * (size_t)n_slice + 64) != hipSuccess) { o.err = "filter: out of page-locked host memory"; return R3DM_ERR_NOMEM; }
        {
            float* lc = static_cast<float*>(B.pin_idx.p);
            for (uint32_t qi = 0; qi < n_coop; ++qi) {
                const uint32_t k = coop_items[qi];
                const uint32_t m = (uint32_t)(begin_end[2 * k + 1] - begin_end[2 * k]);
                float* t = lc + soff[k];
                float pre = 0.0f;
                t[0] = 0.0f; t[m] = 0.0f;
                for (uint32_t i = 1; i <= m / 2; ++i) {
                    pre = pre + (l10[m - i + 1] - l10[i]);
                    t[i] = pre;
                    if (m - i > i) t[m - i] = pre;
                }
            }
            // (one copy of the whole table array: the slices of the short pairs carry whatever the buffer held -- the one-workgroup
            // kernel fills its own tables before it reads them)
            FHIP(hipMemcpyAsync(fp.scratch_logc, lc, 4 * (size_t)n_slice, hipMemcpyHostToDevice, c->stream));
        }
        FHIP(hipStreamSynchronize(c->stream));           // `stage` leaves scope; the landing buffer is reused for the results
        fp.coop_items = reinterpret_cast<const uint32_t*>(base);
        fp.coop_G = fp.coop_items + n_coop; fp.coop_slice = fp.coop_items + 2 * (size_t)n_coop; fp.coop_hoff = fp.coop_items + 3 * (size_t)n_coop;
        fp.coop_q = nullptr;                              // (the scheduling words are shared by the kinds of a call: filter_launch)
        fp.coop_pub = base + o_pub;
        fp.coop_models = reinterpret_cast<double*>(base + o_models);
        fp.coop_bm = reinterpret_cast<double*>(base + o_bm);
        fp.coop_cnt = reinterpret_cast<uint32_t*>(base + o_cnt);
        fp.coop_la = reinterpret_cast<double*>(base + o_la);
        fp.coop_tstar = reinterpret_cast<double*>(base + o_ts);
        fp.coop_hist = reinterpret_cast<uint32_t*>(base + o_hist);
        fp.coop_prof = nullptr;
        if (r3dm_dev_knob("R3DM_COOP_PROF", 0)) {               // developer build: per-pair phase times of the cooperative kernel
            FHIP(B.f_coop_prof.ensure(128 * (size_t)n_coop));
            FHIP(hipMemsetAsync(B.f_coop_prof.p, 0, 128 * (size_t)n_coop, c->stream));
            fp.coop_prof = B.f_coop_prof.as<unsigned long long>();
        }
        if (filter_coop_lds_bytes() > 160 * 1024) { o.err = "filter: LDS budget exceeded (cooperative kernel)"; return R3DM_ERR_UNSUPPORTED; }
    }
    if (filter_F_lds_bytes(fp.m_cap, model_kind) > 160 * 1024) { o.err = "filter: LDS budget exceeded"; return R3DM_ERR_UNSUPPORTED; }
    // debug aid: R3DM_TRACE_PAIR="I,J" + R3DM_TRACE_FILE=path dump the per-model trace of one pair
    auto trace_own = std::make_shared<SharedDevBuf>();
    DevBuf& trace_buf = trace_own->b;
    const uint32_t trace_cap = 16384;
    const char* tp = r3dm_dev_str("R3DM_TRACE_PAIR");
    const char* tf = r3dm_dev_str("R3DM_TRACE_FILE");
    fp.trace = nullptr; fp.trace_item = 0xFFFFFFFFu; fp.trace_cap = trace_cap; fp.trace_rows = nullptr;
    fp.trace_iter = (uint32_t)r3dm_dev_knob("R3DM_TRACE_ITER", -1);
    if (tp && tf) {
        unsigned tI = 0, tJ = 0;
        if (sscanf(tp, "%u,%u", &tI, &tJ) == 2)
            for (uint32_t k = 0; k < NI; ++k)
                if (ids[k].x == tI && ids[k].y == tJ) fp.trace_item = k;
        if (fp.trace_item != 0xFFFFFFFFu) {
            FHIP(trace_buf.ensure(40 * (size_t)trace_cap + 64));
            FHIP(hipMemsetAsync(trace_buf.p, 0, 40 * (size_t)trace_cap + 64, c->stream));
            fp.trace = trace_buf.as<double>() + 8;
            fp.trace_rows = trace_buf.as<uint32_t>();
        }
    }
    auto dbg_own = std::make_shared<SharedDevBuf>();
    DevBuf& dbg_buf = dbg_own->b;
    fp.dbg = nullptr;
    if (r3dm_dev_knob("R3DM_FILTER_CHECK", 0)) {
        FHIP(dbg_buf.ensure(64));
        FHIP(hipMemsetAsync(dbg_buf.p, 0, 64, c->stream));
        fp.dbg = dbg_buf.as<uint32_t>();
    }
    fp_out = fp;
    // the host staging vectors above (slots, ids, begin_end, l10, lck, soff) were handed to asynchronous copies on the context's stream:
    // they must not leave scope before the copies have read them
    FHIP(hipStreamSynchronize(c->stream));
    r3dm_graph* graw = g.release();
    o.pending = graw;
    FilterBufs* const Bp = &B;
    collect = [=, &o](float ms) mutable -> int {
    std::unique_ptr<r3dm_graph> g(graw);
    o.pending = nullptr;
    FilterBufs& B = *Bp;                                   // (the context's buffer set itself, not a copy captured with the closure)
    DevBuf& trace_buf = trace_own->b;                      // (released with the closure, whether or not it ever runs)
    DevBuf& dbg_buf = dbg_own->b;
    if (fp.dbg) {                                          // developer build only
        uint32_t d[4] = {0, 0, 0, 0};
        hipError_t e = hipMemcpyAsync(d, fp.dbg, 16, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);       // the kernel runs on c->stream (non-blocking): wait for it
        dbg_buf.release();
        if (e != hipSuccess || d[0]) {
            trace_buf.release();
            o.err = e != hipSuccess ? std::string("filter check: ") + hipGetErrorString(e)
                   : "filter invariant " + std::to_string(d[0]) + " violated at item " + std::to_string(d[1]) + " (" + std::to_string(d[2]) + ", " + std::to_string(d[3]) + ")";
            if (e == hipSuccess && (d[0] == 10u || d[0] == 11u)) {     // (the scout's bound, and the NFA / the full evaluation's bound it exceeds, as float bits)
                float fa, fb; memcpy(&fa, &d[2], 4); memcpy(&fb, &d[3], 4);
                char b[96]; snprintf(b, sizeof b, " = (%.7g, %.7g)", (double)fa, (double)fb); o.err += b;
            }
            return R3DM_ERR_HIP;
        }
    }

    std::vector<uint32_t> h_cnt(NI);
    // (inlier indices through the kind's page-locked landing buffer: pageable destinations are pinned on the way by the runtime)
    if (B.pin_idx.ensure(4 * (size_t)n_slice + 64) != hipSuccess) { o.err = "filter: out of page-locked host memory"; return R3DM_ERR_NOMEM; }
    uint32_t* h_idx = static_cast<uint32_t*>(B.pin_idx.p);
    std::vector<double> h_F(9 * (size_t)NI);
    FHIP(hipMemcpyAsync(h_cnt.data(), B.f_inl_cnt.p, 4 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
    FHIP(hipMemcpyAsync(h_idx, B.f_inl_idx.p, 4 * (size_t)n_slice, hipMemcpyDeviceToHost, c->stream));
    FHIP(hipMemcpyAsync(h_F.data(), B.f_F.p, 72 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
    FHIP(hipStreamSynchronize(c->stream));
    o.ms_kernels = ms;
    if (fp.n_coop) {
        uint32_t qh[96];
        FHIP(hipMemcpy(qh, c->coop_sched.p, sizeof(qh), hipMemcpyDeviceToHost));
        qh[2] = qh[64];                                       // pairs finished (of all kinds of the call)
        if (qh[8] != 0u || qh[2] != qh[5]) {
            o.err = "filter: the cooperative kernel (kind " + std::to_string(model_kind) + ") stalled (code " + std::to_string(qh[8]) + ", info " + std::to_string(qh[9]) + ", " +
                    std::to_string(qh[2]) + " of " + std::to_string(qh[5]) + " pairs finished; at the stall: idle bits " + std::to_string(qh[10]) + " / " +
                    std::to_string(qh[11]) + " finished " + std::to_string(qh[12]) + " potential " + std::to_string(qh[13]) + " workers " + std::to_string(qh[14]) +
                    " started " + std::to_string(qh[15]) + ")";
            return R3DM_ERR_HIP;
        }
    }
    if (fp.n_coop && fp.coop_prof) {
        std::vector<unsigned long long> pr(16 * (size_t)fp.n_coop);
        FHIP(hipMemcpy(pr.data(), fp.coop_prof, 8 * pr.size(), hipMemcpyDeviceToHost));
        static const char* kind_name[3] = {"F", "H", "E"};
        for (uint32_t q = 0; q < fp.n_coop; ++q) {
            const unsigned long long* r = &pr[16 * (size_t)q];
            const uint32_t k = coop_items[q];
            fprintf(stderr, "coop %s pair %u m %llu G %u: wall %.0f us | init %.0f solve %.0f form+publish %.0f slices %.0f (sum over workgroups) arrive %.0f bounds %.0f "
                    "full %.0f (%llu) walk %.0f | batches %llu models %llu | task delay sum %.0f max %.0f, longest slice %.0f %u\n", kind_name[model_kind], q,
                    (unsigned long long)(begin_end[2 * k + 1] - begin_end[2 * k]), coop_G[q], r[10] / 100.0, r[0] / 100.0, r[1] / 100.0, r[2] / 100.0, r[3] / 100.0,
                    r[4] / 100.0, r[5] / 100.0, r[6] / 100.0, r[7], r[8] / 100.0, r[9], r[11], r[12] / 100.0, r[14] / 100.0, r[13] / 100.0, 0u);
        }
    }
    if (fp.trace) {
        std::vector<double> tr(5 * (size_t)trace_cap + 8);
        FHIP(hipMemcpy(tr.data(), trace_buf.p, tr.size() * 8, hipMemcpyDeviceToHost));
        const uint32_t rows = std::min<uint32_t>(*reinterpret_cast<uint32_t*>(tr.data()), trace_cap);
        if (FILE* f = fopen(tf, "w")) {
            for (uint32_t r = 0; r < rows; ++r)
                fprintf(f, "%.0f %.0f %.0f %.17g %.0f\n", tr[8 + 5 * r], tr[9 + 5 * r], tr[10 + 5 * r], tr[11 + 5 * r], tr[12 + 5 * r]);
            if (fp.trace_iter != 0xFFFFFFFFu) {
                fprintf(f, "# sample");
                for (int k = 0; k < 20; ++k) fprintf(f, " %.0f", tr[8 + 5 * (size_t)(trace_cap - 4) + k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
        trace_buf.release();
    }
    // per-item diagnostics of this call (threshold px, NFA, iterations, models), in putative-pair order
    {
        std::vector<double> h_thr(2 * (size_t)NI);
        std::vector<uint32_t> h_it(2 * (size_t)NI);
        FHIP(hipMemcpyAsync(h_thr.data(), B.f_thr.p, 16 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
        FHIP(hipMemcpyAsync(h_it.data(), B.f_iters.p, 8 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
        FHIP(hipStreamSynchronize(c->stream));
        o.report.assign(NP, r3dm_pair_report{});
        for (uint32_t k = 0; k < NI; ++k) {
            r3dm_pair_report& r = o.report[item_pair[k]];
            r.threshold_px = h_thr[2 * k]; r.nfa = h_thr[2 * k + 1];
            r.iterations = h_it[2 * k]; r.models = h_it[2 * k + 1]; r.inliers = h_cnt[k];
        }
    }

    uint64_t kept = 0;
    {
        uint64_t total_kept = 0;
        for (uint32_t k = 0; k < NI; ++k) total_kept += h_cnt[k];
        g->matches.reserve(total_kept);
    }
    const bool mirror = c->device_graphs;
    if (mirror) { g->dev.valid = true; g->dev.device = c->device; }
    std::vector<uint32_t> m_ids, m_cnts;
    std::vector<GraphSeg> m_segs;
    for (uint32_t k = 0; k < NI; ++k) {
        // GeometricFilter_{F,H}Matrix_AC: accept iff #inliers > 2.5 * MINIMUM_SAMPLES
        if ((double)h_cnt[k] <= 2.5 * SS) continue;
        const uint32_t p = item_pair[k];
        const uint64_t base = putative->offsets[p];
        // the reference's extra check after the E filter (src/R3DComputeMatches.cpp:2175-2192): pairs with poor overlap go
        if (model_kind == 2 && (h_cnt[k] < min_count ||
                                (float)h_cnt[k] / (float)(putative->offsets[p + 1] - base) < min_ratio)) continue;
        g->pairs.push_back(putative->pairs[2 * p]); g->pairs.push_back(putative->pairs[2 * p + 1]);
        {
            const size_t at = g->matches.size();
            g->matches.resize(at + h_cnt[k]);
            r3dm_match* dst = g->matches.data() + at;
            const r3dm_match* src = putative->matches.data() + base;
            const uint32_t* ix = h_idx + soff[k];
            for (uint32_t q = 0; q < h_cnt[k]; ++q) dst[q] = src[ix[q]];
        }
        g->offsets.push_back(g->matches.size());
        if (mirror) {      // r3dm_set_device_graphs: the same inliers gathered on the device (putative matches through the inlier indices)
            m_ids.push_back(putative->pairs[2 * p]); m_ids.push_back(putative->pairs[2 * p + 1]); m_cnts.push_back(h_cnt[k]);
            m_segs.push_back(GraphSeg{base, soff[k], g->offsets[g->offsets.size() - 2], h_cnt[k], 0});
        }
        if (F_out) memcpy(F_out + 9 * kept, h_F.data() + 9 * (size_t)k, 72);
        ++kept;
    }
    if (mirror) (void)graph_dev_append(c, g.get(), m_ids, m_cnts, m_segs, fp.matches, B.f_inl_idx.as<uint32_t>());
    o.ms_wall = now_ms() - t_call;
    o.pending = nullptr;
    *out = g.release();
    return R3DM_OK;
    };
    return R3DM_OK;
}

// The cooperative kernel of a call: ONE pool of workers for the long pairs of all its filters (kernels_filter_coop.hip).  Builds the
// scheduling words (counters, idle bitmap, a mailbox line per worker), the start order (kind << 30 | pair, most work
// first) and the device copy of the FilterParams, and launches on the context's cooperative stream; c->coop_ev is recorded behind it.
static int coop_launch_shared(r3dm_ctx* c, FilterCallOut& o, FilterParams* fps, const CoopPlan* plans, int n, bool& launched)
{
    launched = false;
    uint32_t n_pairs = 0, slots = 0;
    for (int k = 0; k < n; ++k) { n_pairs += fps[k].n_coop; slots += plans[k].slots; }
    if (!n_pairs) return R3DM_OK;
    const int workers_knob = r3dm_dev_knob("R3DM_FILTER_COOP_WORKERS", 0);
    const uint32_t workers = std::min<uint32_t>(256u, workers_knob > 0 ? (uint32_t)workers_knob : std::min<uint32_t>(slots, (uint32_t)std::max(c->n_cu, 1)));   // (the idle bitmap has 256 bits)
    const size_t q_words = 96 + 32 * (size_t)workers;          // three lines of counters + one mailbox line per worker (kernels_filter_coop.hip)
    const size_t start_off = (q_words + 63) / 64 * 64, params_off = ((start_off + n_pairs) * 4 + 255) / 256 * 256;
    std::vector<unsigned char> stage(params_off + 3 * sizeof(FilterParams), 0);
    uint32_t* q = reinterpret_cast<uint32_t*>(stage.data());
    const int leaders_knob = r3dm_dev_knob("R3DM_FILTER_COOP_LEADERS", 0);
    q[5] = n_pairs; q[20] = workers; q[65] = slots; q[66] = workers;
    q[21] = leaders_knob > 0 ? (uint32_t)leaders_knob : std::max<uint32_t>(1u, (workers * 55u + 99u) / 100u);     // pairs led at a time
    for (uint32_t w = 0; w < workers; ++w) q[96 + 32 * (size_t)w] = 0xFFFFFFFEu;   // mailboxes: nobody waits yet
    // start order: the pair with the most work first.  Work ~ putatives x models per iteration (E ~4.5, F ~2.6, H ~1) + E's solves.
    struct Ent { double w; uint32_t v; };
    std::vector<Ent> ents;
    for (int k = 0; k < n; ++k) {
        const int kind = fps[k].model_kind;
        const double per = kind == 2 ? 6.0 : (kind == 0 ? 2.6 : 1.0);
        for (uint32_t p = 0; p < fps[k].n_coop; ++p) ents.push_back({per * plans[k].len[p], ((uint32_t)kind << 30) | p});
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.w > b.w; });
    for (uint32_t i = 0; i < n_pairs; ++i) q[start_off + i] = ents[i].v;
    FHIP(c->coop_sched.ensure(stage.size()));
    unsigned char* base = c->coop_sched.as<unsigned char>();
    FilterParams* hp = reinterpret_cast<FilterParams*>(stage.data() + params_off);
    for (int k = 0; k < n; ++k) {
        // a kind without long pairs (none of its pairs is long, or it has no work items at all: E on views without intrinsics) leaves
        // its block zeroed -- the start order never names it -- and must not overwrite another kind's block
        if (!fps[k].n_coop) continue;
        fps[k].coop_q = reinterpret_cast<uint32_t*>(base);
        fps[k].coop_workers = workers;
        hp[fps[k].model_kind] = fps[k];
    }
    if (!c->coop_stream || !c->coop_ev) {
        hipStream_t s2 = nullptr; hipEvent_t e2 = nullptr;
        FHIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const hipError_t e = hipEventCreate(&e2);
        if (e != hipSuccess) { (void)hipStreamDestroy(s2); FHIP(e); }
        c->coop_stream = s2; c->coop_ev = e2;
    }
    FHIP(hipMemcpyAsync(base, stage.data(), stage.size(), hipMemcpyHostToDevice, c->coop_stream));
    FHIP(launch_filter_coop(c->coop_stream, reinterpret_cast<const FilterParams*>(base + params_off), reinterpret_cast<uint32_t*>(base),
                            reinterpret_cast<const uint32_t*>(base) + start_off, workers));
    FHIP(hipEventRecord(c->coop_ev, c->coop_stream));
    FHIP(hipStreamSynchronize(c->coop_stream));          // `stage` leaves scope (and the kernels below are waited for anyway)
    launched = true;
    return R3DM_OK;
}

// launch + wait + HIP-event time of the kernels of `n` prepared filters: the one-workgroup-per-pair kernel of every kind's short pairs
// on a stream of its own (one filter: the context's stream; several: the kinds' priority streams), the cooperative kernel of all long
// pairs once for the call.  ms[k] = HIP-event time from the first launch to the end of kind k's short-pair kernel, or to the end of the
// cooperative kernel if that came later.
static int filter_launch(r3dm_ctx* c, FilterCallOut& o, FilterParams* fps, const CoopPlan* plans, int n, float* ms)
{
    for (int k = 0; k < n; ++k) ms[k] = 0.f;
    int live = 0;
    for (int k = 0; k < n; ++k) live += fps[k].n_items ? 1 : 0;
    if (!live) return R3DM_OK;
    FHIP(hipStreamSynchronize(c->stream));                // every upload of the prepare steps has landed
    int prio_low = 0, prio_high = 0;
    FHIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));          // numerically: low >= high
    std::vector<hipStream_t> st(n, nullptr);
    std::vector<hipEvent_t> e0(n, nullptr), e1(n, nullptr);
    for (int k = 0; k < n; ++k) {
        if (!fps[k].n_items) continue;
        if (n == 1) { st[k] = c->stream; e0[k] = c->ev0; e1[k] = c->ev1; continue; }
        FilterBufs& B = c->fb[fps[k].model_kind];
        if (!B.stream || !B.ev0 || !B.ev1) {
            // (all three or none: a half-made set would break every later call of the context)
            if (B.ev0) (void)hipEventDestroy(B.ev0);
            if (B.ev1) (void)hipEventDestroy(B.ev1);
            if (B.stream) (void)hipStreamDestroy(B.stream);
            B.stream = nullptr; B.ev0 = B.ev1 = nullptr;
            const int prio = fps[k].model_kind == 2 ? prio_high : (fps[k].model_kind == 0 ? (prio_low + prio_high) / 2 : prio_low);
            hipStream_t s = nullptr; hipEvent_t a = nullptr, b = nullptr;
            hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
            if (e == hipSuccess) e = hipEventCreate(&a);
            if (e == hipSuccess) e = hipEventCreate(&b);
            if (e != hipSuccess) {
                if (a) (void)hipEventDestroy(a);
                if (b) (void)hipEventDestroy(b);
                if (s) (void)hipStreamDestroy(s);
                FHIP(e);
            }
            B.stream = s; B.ev0 = a; B.ev1 = b;
        }
        st[k] = B.stream; e0[k] = B.ev0; e1[k] = B.ev1;
    }
    // the short pairs of every kind first (cheap to launch), then the shared cooperative kernel; every bracket closes behind both
    for (int k = 0; k < n; ++k) {
        if (!fps[k].n_items) continue;
        FHIP(hipEventRecord(e0[k], st[k]));
        if (fps[k].n_short) FHIP(launch_filter_F(st[k], fps[k]));
    }
    bool coop = false;
    const int rc = coop_launch_shared(c, o, fps, plans, n, coop);
    {   // occupancy bookkeeping of the call (r3dm_stats): workgroups launched, items on the cooperative kernel
        uint64_t wgs = 0, items = 0;
        for (int k = 0; k < n; ++k) { wgs += fps[k].n_short; items += fps[k].n_coop; }
        for (int k = 0; k < n; ++k) if (coop && fps[k].n_coop) { wgs += fps[k].coop_workers; break; }
        c->stats.n_filter_workgroups = wgs; c->stats.n_filter_coop_pairs = coop ? items : 0;
    }
    hipError_t first = hipSuccess;
    for (int k = 0; k < n; ++k) {
        if (!fps[k].n_items) continue;
        if (rc == R3DM_OK) {
            if (coop && fps[k].n_coop) (void)hipStreamWaitEvent(st[k], c->coop_ev, 0);
            (void)hipEventRecord(e1[k], st[k]);
        }
        const hipError_t e = hipStreamSynchronize(st[k]);                 // wait for ALL of them, whatever one of them says
        if (e != hipSuccess && first == hipSuccess) first = e;
        if (e == hipSuccess && rc == R3DM_OK) (void)hipEventElapsedTime(&ms[k], e0[k], e1[k]);
    }
    if (rc != R3DM_OK) return rc;
    FHIP(first);
    return R3DM_OK;
}

// one filter call on the context: its outcome folded into the context's error / statistics / report
static int filter_one(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter, uint64_t seed, r3dm_ferror err_kind,
                      int model_kind, r3dm_graph** out, double* M_out, uint32_t min_count = 0, float min_ratio = 0.f)
{
    if (!c) return R3DM_ERR_INVALID;
    FilterCallOut o;
    FilterParams fp{};
    CoopPlan plan;
    std::function<int(float)> collect;
    float ms = 0.f;
    int rc = filter_prepare(c, o, putative, max_residual_px, max_iter, seed, err_kind, model_kind, out, M_out, min_count, min_ratio, fp, plan, collect);
    if (rc == R3DM_OK) rc = filter_launch(c, o, &fp, &plan, 1, &ms);
    if (rc == R3DM_OK) rc = collect(ms);
    if (rc != R3DM_OK && !o.err.empty()) c->err = o.err;
    c->stats.ms_filter_kernels = o.ms_kernels;
    c->stats.ms_wall_filter = o.ms_wall;
    if (rc == R3DM_OK) c->report = std::move(o.report);
    return rc;
}

static int r3dm_filter_F_impl(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_ferror err_kind, r3dm_graph** out, double* F_out)
{
    return filter_one(c, putative, max_residual_px, max_iter, seed, err_kind, 0, out, F_out);
}

extern "C" int r3dm_filter_F(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_ferror err_kind, r3dm_graph** out, double* F_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_filter_F_impl(c, putative, max_residual_px, max_iter, seed, err_kind, out, F_out); });
}

static int r3dm_filter_H_impl(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_graph** out, double* H_out)
{
    return filter_one(c, putative, max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, 1, out, H_out);
}

extern "C" int r3dm_filter_H(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_graph** out, double* H_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_filter_H_impl(c, putative, max_residual_px, max_iter, seed, out, H_out); });
}

extern "C" int r3dm_set_intrinsics(r3dm_ctx* c, uint32_t view_id, const double* K)
{
    if (!c) return R3DM_ERR_INVALID;
    auto it = c->slot_of.find(view_id);
    if (it == c->slot_of.end()) { c->err = "r3dm_set_intrinsics: unregistered view"; return R3DM_ERR_INVALID; }
    HostImage& h = *c->imgs[it->second];
    if (!K) { h.has_K = false; return R3DM_OK; }
    // inverse by the adjugate, operation for operation what oracle/essential.c orc_inv3 does
    const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[5] * K[6] - K[3] * K[8], c02 = K[3] * K[7] - K[4] * K[6];
    const double det = K[0] * c00 + K[1] * c01 + K[2] * c02;
    if (!(det != 0.0) || !std::isfinite(det)) { c->err = "r3dm_set_intrinsics: singular K"; return R3DM_ERR_INVALID; }
    const double id = 1.0 / det;
    h.Kinv[0] = c00 * id; h.Kinv[1] = (K[2] * K[7] - K[1] * K[8]) * id; h.Kinv[2] = (K[1] * K[5] - K[2] * K[4]) * id;
    h.Kinv[3] = c01 * id; h.Kinv[4] = (K[0] * K[8] - K[2] * K[6]) * id; h.Kinv[5] = (K[2] * K[3] - K[0] * K[5]) * id;
    h.Kinv[6] = c02 * id; h.Kinv[7] = (K[1] * K[6] - K[0] * K[7]) * id; h.Kinv[8] = (K[0] * K[4] - K[1] * K[3]) * id;
    h.has_K = true;
    return R3DM_OK;
}

static int r3dm_filter_E_impl(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out)
{
    return filter_one(c, putative, max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, 2, out, E_out, min_count, min_ratio);
}

extern "C" int r3dm_filter_E(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_filter_E_impl(c, putative, max_residual_px, max_iter, seed, min_count, min_ratio, out, E_out); });
}

// F, E and H of one putative graph side by side.  Long pairs (>= R3DM_FILTER_COOP_MIN putatives, 4096) of ALL requested filters run on
// ONE cooperative kernel -- a pool of <= 256 persistent workgroups in which a pair's residual passes are row slices handed to idle
// workers (kernels_filter_coop.hip) -- so a collection of few, long pairs (24 photographs: 94 putative pairs of
// 10-20 k matches) fills the chip instead of a third of it, and a pair is no longer bound by one CU's f64 rate.  Short pairs keep the
// one-workgroup-per-pair kernels, one launch per kind on streams of three priority classes, beside the cooperative kernel.  What lets
// kernels overlap at all was taking the agent-scope fences out of their barriers (kernels_filter.hip: wg_fence).  Measured and
// dropped in round 3: one merged one-workgroup-per-pair kernel with the three model kinds as branches (hipcc's code for the
// fundamental-matrix branch faulted, product build only, spilled lists only); the cooperative kernel keeps its phases out of line
// (noinline) for the same family of reasons.
extern "C" int r3dm_filter_FEH(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter, uint64_t seed, int which,
                               uint32_t e_min_count, float e_min_ratio, r3dm_graph** out_F, r3dm_graph** out_E, r3dm_graph** out_H,
                               double* ms_kernels3, double* ms_wall3)
{
    if (!c || !putative || (which & 7) == 0) return R3DM_ERR_INVALID;
    if (((which & 1) && !out_F) || ((which & 2) && !out_E) || ((which & 4) && !out_H)) return R3DM_ERR_INVALID;
    return r3dm_guarded(c, [&]() -> int {
        if (out_F) *out_F = nullptr;
        if (out_E) *out_E = nullptr;
        if (out_H) *out_H = nullptr;
        if (ms_kernels3) ms_kernels3[0] = ms_kernels3[1] = ms_kernels3[2] = 0.0;
        if (ms_wall3) ms_wall3[0] = ms_wall3[1] = ms_wall3[2] = 0.0;
        const double t_feh0 = now_ms();
        struct Call { int kind, slot; r3dm_graph** out; FilterCallOut o; std::function<int(float)> collect; };
        std::vector<std::unique_ptr<Call>> calls;          // in the order F, E, H; kinds of the buffer sets: 0 F, 1 H, 2 E
        if (which & 1) { calls.emplace_back(new Call()); calls.back()->kind = 0; calls.back()->slot = 0; calls.back()->out = out_F; }
        if (which & 2) { calls.emplace_back(new Call()); calls.back()->kind = 2; calls.back()->slot = 1; calls.back()->out = out_E; }
        if (which & 4) { calls.emplace_back(new Call()); calls.back()->kind = 1; calls.back()->slot = 2; calls.back()->out = out_H; }
        std::vector<FilterParams> fps(calls.size());
        std::vector<CoopPlan> plans(calls.size());
        int rc = R3DM_OK;
        for (size_t i = 0; i < calls.size() && rc == R3DM_OK; ++i) {
            Call& k = *calls[i];
            rc = filter_prepare(c, k.o, putative, max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, k.kind, k.out, nullptr,
                                k.kind == 2 ? e_min_count : 0u, k.kind == 2 ? e_min_ratio : 0.f, fps[i], plans[i], k.collect,
                                i > 0 && fps[0].n_items ? fps[0].matches : nullptr);
            if (rc != R3DM_OK && !k.o.err.empty()) c->err = k.o.err;
        }
        const double t_prep = now_ms();
        float ms[3] = {0.f, 0.f, 0.f};
        if (rc == R3DM_OK) {
            FilterCallOut lo;
            rc = filter_launch(c, lo, fps.data(), plans.data(), (int)fps.size(), ms);
            if (rc != R3DM_OK && !lo.err.empty()) c->err = lo.err;
        }
        const double t_launch = now_ms();
        float ms_max = 0.f;
        // the kinds' results come back side by side as well: each collect is a few copies into its own page-locked buffer and the host-side
        // assembly of its graph (inlier indices -> matches), 1.5-2 ms per kind on a stage's graph; a thread per kind.  (Not with device
        // mirrors -- they share the context's gather scratch -- and not in the developer build, whose traces and checks are written
        // for one collect at a time.)
        std::vector<int> rcs(calls.size(), R3DM_OK);
        bool collected = false;
#ifndef R3DM_DEVTOOLS
        if (rc == R3DM_OK && calls.size() > 1 && !c->device_graphs) {
            std::vector<std::thread> th;
            try {
                for (size_t i = 1; i < calls.size(); ++i)
                    th.emplace_back([&, i]() {
                        (void)hipSetDevice(c->device);
                        try { rcs[i] = calls[i]->collect(ms[i]); } catch (...) { rcs[i] = R3DM_ERR_NOMEM; }
                    });
            } catch (...) {}                                   // fewer threads than kinds: the rest is collected below
            try { rcs[0] = calls[0]->collect(ms[0]); } catch (...) { rcs[0] = R3DM_ERR_NOMEM; }
            const size_t started = th.size();
            for (std::thread& t : th) t.join();
            for (size_t i = 1 + started; i < calls.size(); ++i) { try { rcs[i] = calls[i]->collect(ms[i]); } catch (...) { rcs[i] = R3DM_ERR_NOMEM; } }
            collected = true;
        }
#endif
        for (size_t i = 0; i < calls.size() && rc == R3DM_OK; ++i) {
            Call& k = *calls[i];
            rc = collected ? rcs[i] : k.collect(ms[i]);
            if (rc != R3DM_OK && !k.o.err.empty()) c->err = k.o.err;
            ms_max = std::max(ms_max, ms[i]);
            if (ms_kernels3) ms_kernels3[k.slot] = ms[i];
            if (ms_wall3) ms_wall3[k.slot] = k.o.ms_wall;
            if (rc == R3DM_OK && (k.kind == 2 || !(which & 2))) c->report = k.o.report;       // the E call's diagnostics, else the last one's
        }
        c->stats.ms_filter_kernels = ms_max;
        if (r3dm_dev_knob("R3DM_FILTER_TIMING", 0))
            fprintf(stderr, "r3dm_filter_FEH: prepare %.2f ms, launch + wait %.2f ms (kernels %.2f), collect %.2f ms\n", t_prep - t_feh0, t_launch - t_prep, ms_max,
                    now_ms() - t_launch);
        if (rc != R3DM_OK) {
            if (out_F && *out_F) { r3dm_graph_free(*out_F); *out_F = nullptr; }
            if (out_E && *out_E) { r3dm_graph_free(*out_E); *out_E = nullptr; }
            if (out_H && *out_H) { r3dm_graph_free(*out_H); *out_H = nullptr; }
        }
        return rc;
    });
}

extern "C" int r3dm_filter_report(const r3dm_ctx* c, r3dm_pair_report* out, uint64_t cap)
{
    if (!c || (cap && !out)) return R3DM_ERR_INVALID;
    const uint64_t n = std::min<uint64_t>(cap, c->report.size());
    if (n) memcpy(out, c->report.data(), n * sizeof(r3dm_pair_report));
    return (int)std::min<uint64_t>(c->report.size(), 0x7FFFFFFF);
}

