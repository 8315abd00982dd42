// api_core.cpp -- part of the host side of libr3dm.so: the C ABI declared in include/r3dm.h (see r3dm_ctx.hpp for the file map).
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 and src/Regard3DFeatures.cpp -- with every arithmetic stage running as
// HIP kernels on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int r3dm_create(int device_id, r3dm_ctx** out)
{
    if (!out) return R3DM_ERR_INVALID;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return R3DM_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n_dev) return R3DM_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return R3DM_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return R3DM_ERR_NO_DEVICE;
    std::string arch = prop.gcnArchName;
    if (arch.find("gfx950") == std::string::npos) return R3DM_ERR_NO_DEVICE;   // the kernels are gfx950-only
    auto* c = new (std::nothrow) r3dm_ctx();
    if (!c) return R3DM_ERR_NOMEM;
    c->device = device_id;
    c->arch = arch;
    c->n_cu = prop.multiProcessorCount;
    c->hbm = prop.totalGlobalMem;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return R3DM_ERR_HIP;
    }
    *out = c;
    return R3DM_OK;
}

extern "C" void r3dm_destroy(r3dm_ctx* c)
{
    if (!c) return;
    if (c->file_writer.joinable()) c->file_writer.join();       // deferred feature files: the writer reads pin_desc
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& im : c->imgs) if (im) im->release();
    DevBuf* bufs[] = {&c->d_imgs, &c->d_pairs, &c->d_nn, &c->d_knn_idx, &c->d_knn_dist, &c->d_fb, &c->d_cnt, &c->d_out,
                      &c->d_pair_off, &c->d_pair_cnt, &c->d_raw, &c->m_raw, &c->m_peer,
                      &c->liop_pix, &c->liop_sx, &c->liop_sy, &c->liop_in, &c->liop_out, &c->liop_cnt, &c->liop_img, &c->liop_M, &c->liop_kern,
                      &c->a_jobs, &c->h_aux, &c->h_jobs, &c->a_scratch, &c->a_ids, &c->d_spill, &c->d_fb2, &c->g_segs};
    for (FilterBufs& fb : c->fb) fb.release();
    c->coop_sched.release();
    if (c->coop_ev) (void)hipEventDestroy(c->coop_ev);
    if (c->coop_stream) (void)hipStreamDestroy(c->coop_stream);
    c->coop_ev = nullptr; c->coop_stream = nullptr;
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : c->ak_bufs) b.release();
    for (auto& im : c->spare) if (im) im->release();
    c->pin_desc.release(); c->pin_out.release(); c->pin_small.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_desc) (void)hipEventDestroy(c->ev_desc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* r3dm_last_error(const r3dm_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int r3dm_device_info(const r3dm_ctx* c, char* arch, size_t arch_cap, int* n_cu, uint64_t* hbm_bytes)
{
    if (!c) return R3DM_ERR_INVALID;
    if (arch && arch_cap) { strncpy(arch, c->arch.c_str(), arch_cap - 1); arch[arch_cap - 1] = 0; }
    if (n_cu) *n_cu = c->n_cu;
    if (hbm_bytes) *hbm_bytes = c->hbm;
    return R3DM_OK;
}

extern "C" int r3dm_get_stats(const r3dm_ctx* c, r3dm_stats* out)
{
    if (!c || !out) return R3DM_ERR_INVALID;
    *out = c->stats;
    out->n_views_staged = c->n_views_staged;
    return R3DM_OK;
}

extern "C" int r3dm_get_features_totals(const r3dm_ctx* c, r3dm_features_totals* out)
{
    if (!c || !out) return R3DM_ERR_INVALID;
    *out = c->feat_totals;
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// views
// ------------------------------------------------------------------------------------------------
// writes the table entry of `slot`; stat_bits3 / split_k are given when the slot mounts an already staged r3dm_index (the
// staging kernels fill them otherwise)
int upload_imgdev(r3dm_ctx* c, uint32_t slot, const uint32_t* stat_bits3, int32_t split_k, bool counts_ok)
{
    const size_t need = sizeof(ImgDev) * c->imgs.size();
    if (need > c->d_imgs.cap) {
        // grow and re-upload every live slot; max_norm_bits must survive -> read the old table back first
        std::vector<ImgDev> old(c->d_imgs.cap / sizeof(ImgDev));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        if (!old.empty()) {
            R3DM_HIP(c, hipMemcpyAsync(old.data(), c->d_imgs.p, old.size() * sizeof(ImgDev), hipMemcpyDeviceToHost, c->stream));
            R3DM_HIP(c, hipStreamSynchronize(c->stream));
        }
        DevBuf nb;
        R3DM_HIP(c, nb.ensure(sizeof(ImgDev) * std::max<size_t>(64, c->imgs.size() * 2)));
        R3DM_HIP(c, hipMemsetAsync(nb.p, 0, nb.cap, c->stream));
        const size_t keep = std::min(old.size(), c->imgs.size());
        if (keep) R3DM_HIP(c, hipMemcpyAsync(nb.p, old.data(), keep * sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        c->d_imgs.release();
        c->d_imgs = nb;
    }
    const HostImage& h = *c->imgs[slot];
    ImgDev d{};
    d.rows = h.rows.as<float>(); d.tiled = h.tiled.as<float>(); d.norms = h.norms.as<float>();
    d.bin = h.bin.as<uint32_t>(); d.xy = h.has_xy ? h.xy.as<float>() : nullptr;
    d.canon = h.has_dup ? h.canon.as<uint32_t>() : nullptr;
    d.n = h.n; d.n_tiles = h.n_tiles; d.dim = h.dim; d.G = h.G; d.words = h.words;
    d.width = h.width; d.height = h.height;
    d.max_norm_bits = stat_bits3 ? stat_bits3[0] : 0; d.max_abs_bits = stat_bits3 ? stat_bits3[1] : 0; d.not_integer = stat_bits3 ? stat_bits3[2] : 0;
    d.ann_adj = nullptr; d.ann_deg = nullptr; d.ann_rows16 = nullptr; d.ann_rows8 = nullptr;          // staging invalidates the graph index
    d.tiled16 = h.tiled16.as<uint16_t>();
    d.tiledh = h.tiledh.as<uint16_t>(); d.split_k = split_k;
    d.tiledc = h.tiledc.as<uint16_t>(); d.cscale = h.cscale.as<float>(); d.cquad = h.cquad.as<float>(); d.tiledp = h.tiledp.as<uint16_t>(); d.cperm = h.cperm.as<uint32_t>(); d.counts_fail = counts_ok ? 0u : 1u;
    d.tiled8 = h.tiled8.as<uint8_t>();
    R3DM_HIP(c, hipMemcpyAsync(c->d_imgs.as<ImgDev>() + slot, &d, sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    return R3DM_OK;
}

// copy + re-layout one view into slot `slot`
int stage_into_slot(r3dm_ctx* c, uint32_t slot, uint32_t view_id, uint32_t width, uint32_t height,
                    const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    HostImage& h = *c->imgs[slot];
    h.view_id = view_id; h.n = n; h.dim = dim; h.dtype = dtype; h.width = width; h.height = height;
    h.has_xy = (xy != nullptr); h.has_dup = false; h.live = true;
    h.G = 0; h.n_tiles = 0; h.words = 0; h.ann_K = 0; h.hnsw_M = 0; h.mrpt_trees = 0; h.compact_ready = false;
    if (dtype == R3DM_BIN) {
        h.words = (dim + 3) / 4;
        const uint32_t n_pad = n + 8;
        R3DM_HIP(c, h.bin.ensure((size_t)n_pad * h.words * 4 + kSlackBytes));
        if (n) {
            R3DM_HIP(c, c->d_raw.ensure((size_t)n * dim));
            R3DM_HIP(c, hipMemcpyAsync(c->d_raw.p, desc, (size_t)n * dim, hipMemcpyDefault, c->stream));
        }
        R3DM_HIP(c, launch_stage_bin(c->stream, c->d_raw.as<uint8_t>(), n, dim, h.bin.as<uint32_t>(), h.words, n_pad));
        // one byte per bit in i8 MFMA fragment order + biased popcounts: the tiles of the opt-in MFMA Hamming (r3dm_set_hamming_mfma)
        h.n_tiles = (n + kTileRows - 1) / kTileRows;
        const size_t t8_bytes = (size_t)h.n_tiles * h.words * 1024 + 2 * kSlackBytes;
        const size_t nrm_bytes = (size_t)h.n_tiles * 32 * 4 + kSlackBytes;
        R3DM_HIP(c, h.tiled8.ensure(t8_bytes));
        R3DM_HIP(c, h.norms.ensure(nrm_bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiled8.p, 0, t8_bytes, c->stream));
        R3DM_HIP(c, hipMemsetAsync(h.norms.p, 0x7F, nrm_bytes, c->stream));
        R3DM_HIP(c, launch_stage_bin8(c->stream, h.bin.as<uint32_t>(), n, h.words, h.n_tiles, h.tiled8.as<uint8_t>(), h.norms.as<float>()));
    } else {
        h.G = kernel_G_for(dim);
        h.n_tiles = (n + kTileRows - 1) / kTileRows;
        const size_t tiled_bytes = (size_t)h.n_tiles * h.G * 1024 + kSlackBytes;
        const size_t norm_bytes = (size_t)h.n_tiles * 32 * 4 + kSlackBytes;
        R3DM_HIP(c, h.rows.ensure((size_t)std::max<uint32_t>(n, 1) * dim * 4 + 256));
        R3DM_HIP(c, h.tiled.ensure(tiled_bytes));
        const size_t tiled16_bytes = (size_t)h.n_tiles * ((h.G + 1) / 2) * 1024 + kSlackBytes;
        R3DM_HIP(c, h.tiled16.ensure(tiled16_bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiled16.p, 0, tiled16_bytes, c->stream));
        const size_t tiledh_bytes = (size_t)h.n_tiles * ((h.G + 1) / 2) * 2048 + kSlackBytes;      // f16 hi | lo planes (split nominator)
        R3DM_HIP(c, h.tiledh.ensure(tiledh_bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiledh.p, 0, tiledh_bytes, c->stream));
        // count tiles (rows = small integers x a row scale: LIOP): f16 integers, half the bytes of the split planes, + a scale per row
        const size_t tiledc_bytes = (size_t)h.n_tiles * ((h.G + 1) / 2) * 1024 + kSlackBytes;
        R3DM_HIP(c, h.tiledc.ensure(tiledc_bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiledc.p, 0, tiledc_bytes, c->stream));
        R3DM_HIP(c, h.cscale.ensure((size_t)h.n_tiles * 32 * 4 + kSlackBytes));
        R3DM_HIP(c, hipMemsetAsync(h.cscale.p, 0, (size_t)h.n_tiles * 32 * 4 + kSlackBytes, c->stream));
        R3DM_HIP(c, h.cquad.ensure((counts_summary_offset(h.n_tiles) + ((size_t)h.n_tiles + 1) * 16) * 4 + kSlackBytes));
        R3DM_HIP(c, h.tiledp.ensure(tiledc_bytes));
        R3DM_HIP(c, h.cperm.ensure((size_t)h.n_tiles * 32 * 4 + 256));
        R3DM_HIP(c, h.norms.ensure(norm_bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiled.p, 0, tiled_bytes, c->stream));
        R3DM_HIP(c, hipMemsetAsync(h.norms.p, 0, norm_bytes, c->stream));
        const void* raw = nullptr;
        if (n) {
            if (dtype == R3DM_F32) {
                R3DM_HIP(c, c->d_raw.ensure((size_t)n * dim * 4));
                R3DM_HIP(c, hipMemcpyAsync(c->d_raw.p, desc, (size_t)n * dim * 4, hipMemcpyDefault, c->stream));
            } else {
                R3DM_HIP(c, c->d_raw.ensure((size_t)n * dim));
                R3DM_HIP(c, hipMemcpyAsync(c->d_raw.p, desc, (size_t)n * dim, hipMemcpyDefault, c->stream));
            }
            raw = c->d_raw.p;
        }
        (void)raw;
    }
    if (xy && n) {
        R3DM_HIP(c, h.xy.ensure((size_t)n * 8));
        R3DM_HIP(c, hipMemcpyAsync(h.xy.p, xy, (size_t)n * 8, hipMemcpyDefault, c->stream));
    }
    // position classes (IndMatchDecorator de-duplication needs to know which features share a position): canon[k] = the smallest
    // index among the features at k's position.  One hash pass (equal floats <-> equal bit patterns once -0 is folded into +0; a NaN
    // equals nothing, itself included); host positions are read where they are.
    if (xy && n > 1) {
        std::vector<float> hxy_copy;
        const float* hxy = xy;
        {
            hipPointerAttribute_t at{};
            const bool on_device = hipPointerGetAttributes(&at, xy) == hipSuccess && (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged);
            (void)hipGetLastError();
            if (on_device) {
                hxy_copy.resize((size_t)n * 2);
                R3DM_HIP(c, hipMemcpyAsync(hxy_copy.data(), h.xy.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
                R3DM_HIP(c, hipStreamSynchronize(c->stream));
                hxy = hxy_copy.data();
            }
        }
        std::vector<uint32_t> canon(n);
        bool dup = false;
        {
            // open addressing, linear probing, table of >= 2 n slots: a slot holds the first feature index seen at a position
            uint32_t bits = 4;
            while ((1u << bits) < 2u * n) ++bits;
            const uint32_t mask = (1u << bits) - 1u;
            std::vector<uint64_t> keys((size_t)1 << bits);
            std::vector<uint32_t> vals((size_t)1 << bits, 0xFFFFFFFFu);
            for (uint32_t k = 0; k < n; ++k) {
                const float fx = hxy[2 * (size_t)k], fy = hxy[2 * (size_t)k + 1];
                canon[k] = k;
                if (fx != fx || fy != fy) continue;                           // NaN: a class of its own
                uint32_t bx, by;
                const float zx = fx == 0.0f ? 0.0f : fx, zy = fy == 0.0f ? 0.0f : fy;
                std::memcpy(&bx, &zx, 4); std::memcpy(&by, &zy, 4);
                const uint64_t key = ((uint64_t)bx << 32) | by;
                uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits)) & mask;
                while (vals[slot] != 0xFFFFFFFFu && keys[slot] != key) slot = (slot + 1u) & mask;
                if (vals[slot] == 0xFFFFFFFFu) { keys[slot] = key; vals[slot] = k; }
                else { canon[k] = vals[slot]; dup = true; }                   // ascending k: the stored index is the smallest of the class
            }
        }
        if (dup) {
            h.has_dup = true;
            R3DM_HIP(c, h.canon.ensure((size_t)n * 4));
            R3DM_HIP(c, hipMemcpyAsync(h.canon.p, canon.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            R3DM_HIP(c, hipStreamSynchronize(c->stream));
        }
    }
    int rc = upload_imgdev(c, slot);
    if (rc != R3DM_OK) return rc;
    if (dtype != R3DM_BIN) {
        uint32_t* mx = &(c->d_imgs.as<ImgDev>() + slot)->max_norm_bits;
        R3DM_HIP(c, launch_stage_f32(c->stream, n ? c->d_raw.p : nullptr, dtype == R3DM_U8, n, dim, h.rows.as<float>(),
                                     h.tiled.as<float>(), h.tiled16.as<uint16_t>(), h.norms.as<float>(), h.G, h.n_tiles, mx));
        int32_t* sk = &(c->d_imgs.as<ImgDev>() + slot)->split_k;
        R3DM_HIP(c, launch_stage_split(c->stream, h.rows.as<float>(), n, dim, (h.G + 1) / 2, h.n_tiles, h.tiledh.as<uint16_t>(), mx, sk));
        // count tiles: the staging kernel sets counts_fail when some row is not integers x a scale (the table entry starts at "fails";
        // clear it first: upload_imgdev wrote 1)
        uint32_t* cf = &(c->d_imgs.as<ImgDev>() + slot)->counts_fail;
        uint32_t cfail = 1;
        if (dtype == R3DM_F32 && n && dim <= 256) {
            R3DM_HIP(c, hipMemsetAsync(cf, 0, 4, c->stream));
            R3DM_HIP(c, launch_stage_counts(c->stream, h.rows.as<float>(), n, dim, (h.G + 1) / 2, h.n_tiles, h.tiledc.as<uint16_t>(), h.cscale.as<float>(), h.norms.as<float>(),
                                            h.tiledp.as<uint16_t>(), h.cquad.as<float>(), h.cperm.as<uint32_t>(), cf));
            R3DM_HIP(c, hipMemcpyAsync(&cfail, cf, 4, hipMemcpyDeviceToHost, c->stream));
        }
        uint32_t st3[3] = {0, 0, 1};
        R3DM_HIP(c, hipMemcpyAsync(st3, mx, 12, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(&h.split_k, sk, 4, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        h.counts_ok = cfail == 0;
        std::memcpy(&h.max_abs, &st3[1], 4);
        h.not_integer = (st3[2] & 1u) != 0;
        h.has_negative = (st3[2] & 2u) != 0;
    }
    R3DM_HIP(c, hipStreamSynchronize(c->stream));     // d_raw is reused by the next call
    c->n_views_staged += 1;
    return R3DM_OK;
}

static int r3dm_set_image_impl(r3dm_ctx* c, uint32_t view_id, uint32_t width, uint32_t height,
                              const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    if (!c || dim == 0 || (n && !desc)) return R3DM_ERR_INVALID;
    if (dtype != R3DM_F32 && dtype != R3DM_U8 && dtype != R3DM_BIN) return R3DM_ERR_INVALID;
    if (n >= (1u << 22)) { c->err = "more than 4M features in one view"; return R3DM_ERR_UNSUPPORTED; }
    if (dtype == R3DM_BIN && !(((dim + 3) / 4) == 8 || ((dim + 3) / 4) == 16)) {
        c->err = "binary descriptors must be 29..32 or 61..64 bytes"; return R3DM_ERR_UNSUPPORTED;
    }
    R3DM_HIP(c, hipSetDevice(c->device));
    uint32_t slot;
    auto it = c->slot_of.find(view_id);
    if (it == c->slot_of.end()) {
        slot = (uint32_t)c->imgs.size();
        if (!c->spare.empty()) { c->imgs.emplace_back(std::move(c->spare.back())); c->spare.pop_back(); }      // buffers of a cleared view
        else c->imgs.emplace_back(new HostImage());
        c->slot_of[view_id] = slot;
    } else slot = it->second;
    return stage_into_slot(c, slot, view_id, width, height, desc, n, dim, dtype, xy);
}

extern "C" int r3dm_set_image(r3dm_ctx* c, uint32_t view_id, uint32_t width, uint32_t height,
                              const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_set_image_impl(c, view_id, width, height, desc, n, dim, dtype, xy); });
}

// the size of a helper team of host threads that fits the cores this process may use (affinity mask and cgroup CPU quota), at most `want`
extern "C" int r3dm_host_threads(int want) { return r3dm_host_team(want > 0 ? want : 1, 1); }

extern "C" int r3dm_clear_images(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    // the views are forgotten, their device buffers are kept for the next collection (a stage object that lives across runs, or a
    // bench loop, registers views of the same sizes again and again: six hipMalloc per view were most of the registration time)
    // Kept: the staging buffers of at most kSpareViews views (a collection larger than that gives the rest back at once).  Never kept:
    // the per-view INDEX buffers (graph adjacency, compact row copies, HNSW arrays) -- they are rebuilt per collection and nothing
    // reuses them as they stand, so a context that once held a 1000-view collection does not sit on their gigabytes.
    constexpr size_t kSpareViews = 256;
    for (auto& im : c->imgs) {
        if (!im) continue;
        if (im->borrowed) { im->release(); continue; }
        if (c->spare.size() >= kSpareViews) { im->release(); continue; }
        im->ann_adj.release(); im->ann_deg.release(); im->ann_rows16.release(); im->ann_rows8.release();
        im->hnsw_l0.release(); im->hnsw_up_off.release(); im->hnsw_up.release();
        im->mrpt_R.release(); im->mrpt_RT.release(); im->mrpt_splits.release(); im->mrpt_leaves.release(); im->mrpt_lf.release();
        im->live = false; im->has_K = false; im->ann_K = 0; im->hnsw_M = 0; im->mrpt_trees = 0; im->compact_ready = false; im->n = 0; im->counts_ok = false;
        c->spare.push_back(std::move(im));
    }
    c->imgs.clear();
    c->slot_of.clear();
    return R3DM_OK;
}

// give the spare staging buffers back to the device (a long-lived context between two collections of very different sizes)
extern "C" int r3dm_trim(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& im : c->spare) if (im) im->release();
    c->spare.clear();
    return R3DM_OK;
}

extern "C" int r3dm_set_integer_mfma(r3dm_ctx* c, int enable)
{
    if (!c) return R3DM_ERR_INVALID;
    c->integer_mfma = (enable != 0);
    return R3DM_OK;
}

extern "C" int r3dm_set_split_mfma(r3dm_ctx* c, int enable)
{
    if (!c) return R3DM_ERR_INVALID;
    c->split_mfma = (enable != 0);
    return R3DM_OK;
}

extern "C" int r3dm_set_hamming_mfma(r3dm_ctx* c, int enable)
{
    if (!c) return R3DM_ERR_INVALID;
    c->hamming_mfma = (enable != 0);
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// graph accessors, merge, files
// ------------------------------------------------------------------------------------------------
extern "C" uint64_t r3dm_graph_num_pairs(const r3dm_graph* g) { return g ? g->pairs.size() / 2 : 0; }
extern "C" uint64_t r3dm_graph_num_matches(const r3dm_graph* g) { return g ? g->matches.size() : 0; }
extern "C" const uint32_t* r3dm_graph_pairs(const r3dm_graph* g) { return g ? g->pairs.data() : nullptr; }
extern "C" const uint64_t* r3dm_graph_offsets(const r3dm_graph* g) { return g ? g->offsets.data() : nullptr; }
extern "C" const r3dm_match* r3dm_graph_matches(const r3dm_graph* g) { return g ? g->matches.data() : nullptr; }
extern "C" void r3dm_graph_free(r3dm_graph* g) { delete g; }

// ---- the device mirror of a graph (GraphDev): appended to where the matches already are in device memory
extern "C" int r3dm_set_device_graphs(r3dm_ctx* c, int enable)
{
    if (!c) return R3DM_ERR_INVALID;
    c->device_graphs = enable != 0;
    return R3DM_OK;
}
extern "C" int r3dm_graph_on_device(const r3dm_graph* g) { return g && g->dev.valid ? g->dev.device : -1; }

// grows `b` to hold `need` bytes keeping its first `used` bytes
static hipError_t dev_grow(DevBuf& b, size_t used, size_t need, hipStream_t st)
{
    if (need <= b.cap) return hipSuccess;
    DevBuf n;
    hipError_t e = n.ensure(need + need / 2);
    if (e != hipSuccess) return e;
    if (used) e = hipMemcpyAsync(n.p, b.p, used, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { n.release(); return e; }
    b.release();
    b = n;
    return hipSuccess;
}

// Appends segs.size() kept pairs to the mirror of g: their view ids (2 per pair), their counts, and their matches gathered on the device
// from `src` (through the index list `idx` when given).  segs[k].dst counts from the start of the appended block.  A failure only
// invalidates the mirror (the host vectors of the graph are the product either way).
int graph_dev_append(r3dm_ctx* c, r3dm_graph* g, const std::vector<uint32_t>& pair_ids, const std::vector<uint32_t>& counts, std::vector<GraphSeg>& segs,
                     const r3dm_match* src, const uint32_t* idx)
{
    GraphDev& d = g->dev;
    if (!d.valid) return R3DM_OK;
    const size_t n = segs.size();
    if (n == 0) return R3DM_OK;
    uint64_t add = 0;
    for (uint32_t v : counts) add += v;
    for (GraphSeg& sgm : segs) sgm.dst += d.M;
    hipError_t e = dev_grow(d.pairs, d.P * 8, (d.P + n) * 8, c->stream);
    if (e == hipSuccess) e = dev_grow(d.counts, d.P * 4, (d.P + n) * 4, c->stream);
    if (e == hipSuccess) e = dev_grow(d.matches, d.M * sizeof(r3dm_match), (d.M + add) * sizeof(r3dm_match) + 16, c->stream);
    if (e == hipSuccess) e = c->g_segs.ensure(n * sizeof(GraphSeg));
    if (e == hipSuccess) e = hipMemcpyAsync(d.pairs.as<uint32_t>() + 2 * d.P, pair_ids.data(), n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d.counts.as<uint32_t>() + d.P, counts.data(), n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->g_segs.p, segs.data(), n * sizeof(GraphSeg), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_graph_gather(c->stream, src, idx, c->g_segs.as<GraphSeg>(), (uint32_t)n, d.matches.as<r3dm_match>());
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);       // the host arrays above leave scope with the caller
    if (e != hipSuccess) { d.release(); (void)hipGetLastError(); return R3DM_OK; }
    d.P += n; d.M += add;
    return R3DM_OK;
}

static int r3dm_graph_from_csr_impl(const uint32_t* pairs_ij, uint64_t n_pairs, const uint64_t* offsets,
                                   const r3dm_match* matches, r3dm_graph** out)
{
    if (!out || (n_pairs && (!pairs_ij || !offsets))) return R3DM_ERR_INVALID;
    auto g = std::unique_ptr<r3dm_graph>(new (std::nothrow) r3dm_graph());
    if (!g) return R3DM_ERR_NOMEM;
    g->offsets.push_back(0);
    // keep the PairWiseMatches invariants: ordered by (I, J), no empty entries
    std::vector<uint64_t> ord(n_pairs);
    std::iota(ord.begin(), ord.end(), 0ull);
    std::sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) {
        if (pairs_ij[2 * a] != pairs_ij[2 * b]) return pairs_ij[2 * a] < pairs_ij[2 * b];
        return pairs_ij[2 * a + 1] < pairs_ij[2 * b + 1];
    });
    for (uint64_t k = 0; k < n_pairs; ++k) {
        const uint64_t p = ord[k];
        const uint64_t b = offsets[p], e = offsets[p + 1];
        if (e <= b) continue;
        if (!matches) return R3DM_ERR_INVALID;
        g->pairs.push_back(pairs_ij[2 * p]); g->pairs.push_back(pairs_ij[2 * p + 1]);
        g->matches.insert(g->matches.end(), matches + b, matches + e);
        g->offsets.push_back(g->matches.size());
    }
    *out = g.release();
    return R3DM_OK;
}

extern "C" int r3dm_graph_from_csr(const uint32_t* pairs_ij, uint64_t n_pairs, const uint64_t* offsets,
                                   const r3dm_match* matches, r3dm_graph** out)
{
    return r3dm_guarded(nullptr, [&]() -> int { return r3dm_graph_from_csr_impl(pairs_ij, n_pairs, offsets, matches, out); });
}

static int r3dm_graph_merge_impl(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out)
{
    if (!out || (n_parts && !parts)) return R3DM_ERR_INVALID;
    std::vector<uint32_t> pairs;
    std::vector<uint64_t> offs{0};
    std::vector<r3dm_match> m;
    for (uint32_t k = 0; k < n_parts; ++k) {
        const r3dm_graph* g = parts[k];
        if (!g) continue;
        const uint64_t np = g->pairs.size() / 2;
        for (uint64_t p = 0; p < np; ++p) {
            pairs.push_back(g->pairs[2 * p]); pairs.push_back(g->pairs[2 * p + 1]);
            m.insert(m.end(), g->matches.begin() + g->offsets[p], g->matches.begin() + g->offsets[p + 1]);
            offs.push_back(m.size());
        }
    }
    return r3dm_graph_from_csr(pairs.data(), pairs.size() / 2, offs.data(), m.data(), out);
}

extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out)
{
    return r3dm_guarded(nullptr, [&]() -> int { return r3dm_graph_merge_impl(parts, n_parts, out); });
}

// matches.*.txt / matches.*.bin -- OpenMVG Save/Load(PairWiseMatches)
extern "C" int r3dm_save_matches(const r3dm_graph* g, const char* path)
{
    if (!g || !path) return R3DM_ERR_INVALID;
    const bool bin = has_ext(path, ".bin");
    if (!bin && !has_ext(path, ".txt")) return R3DM_ERR_INVALID;
    FILE* f = fopen(path, bin ? "wb" : "w");
    if (!f) return R3DM_ERR_IO;
    const uint64_t np = g->pairs.size() / 2;
    bool ok = true;
    if (bin) {
        // cereal PortableBinaryOutputArchive: endianness flag, then the std::map as size + (key, value) items
        const uint8_t le = 1;
        ok &= fwrite(&le, 1, 1, f) == 1;
        ok &= fwrite(&np, 8, 1, f) == 1;
        for (uint64_t p = 0; p < np && ok; ++p) {
            const uint64_t cnt = g->offsets[p + 1] - g->offsets[p];
            ok &= fwrite(&g->pairs[2 * p], 4, 2, f) == 2;
            ok &= fwrite(&cnt, 8, 1, f) == 1;
            ok &= fwrite(g->matches.data() + g->offsets[p], sizeof(r3dm_match), cnt, f) == cnt;
        }
    } else {
        // "I J\ncount\n" then one "i j\n" line per match: decimal digits written by hand, two at a time from a table (the same bytes
        // as the "%u %u\n" this replaces, an order of magnitude faster), by a few host threads on runs of pairs of about equal
        // match counts -- each into its own buffer, written out in order (the stage writes four such files of ~10 MB, three of them
        // at the same moment behind the filters: one thread each was 15-20 ms at the end of every step)
        static const char* const kDigits2 =
            "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
        auto put_u64 = [](char* o, uint64_t v, char sep) -> char* {
            char d[20]; int k = 0;
            while (v >= 100) { const uint64_t q = v / 100; const uint32_t r = (uint32_t)(v - q * 100); d[k++] = kDigits2[2 * r + 1]; d[k++] = kDigits2[2 * r]; v = q; }
            if (v >= 10) { d[k++] = kDigits2[2 * v + 1]; d[k++] = kDigits2[2 * v]; } else d[k++] = (char)('0' + v);
            while (k) *o++ = d[--k];
            *o++ = sep;
            return o;
        };
        const uint64_t total = g->matches.size();
        int T = total > 200000 ? r3dm_host_team(4, 4) : 1;
        if (T < 1) T = 1;
        // chunk c = pairs [cut[c], cut[c + 1]): boundaries where the running match count passes c / T of the total
        std::vector<uint64_t> cut((size_t)T + 1, np);
        cut[0] = 0;
        for (int c = 1; c < T; ++c) {
            const uint64_t want = total / (uint64_t)T * (uint64_t)c;
            cut[c] = (uint64_t)(std::upper_bound(g->offsets.begin(), g->offsets.begin() + (ptrdiff_t)np, want) - g->offsets.begin());
            if (cut[c] > np) cut[c] = np;
            if (cut[c] < cut[c - 1]) cut[c] = cut[c - 1];
        }
        std::vector<std::vector<char>> bufs((size_t)T);
        std::vector<size_t> lens((size_t)T, 0);
        std::vector<int> failed((size_t)T, 0);
        r3dm_parallel_for((long)T, T, [&](long c) {
            try {
                const uint64_t p0 = cut[c], p1 = cut[c + 1];
                if (p1 <= p0) return;
                const uint64_t mcount = g->offsets[p1] - g->offsets[p0];
                bufs[(size_t)c].resize((size_t)(mcount * 22 + (p1 - p0) * 44 + 64));        // a line <= 22 bytes, a pair header <= 11 + 11 + 21
                char* o = bufs[(size_t)c].data();
                for (uint64_t p = p0; p < p1; ++p) {
                    o = put_u64(o, g->pairs[2 * p], ' '); o = put_u64(o, g->pairs[2 * p + 1], '\n'); o = put_u64(o, g->offsets[p + 1] - g->offsets[p], '\n');
                    for (uint64_t k = g->offsets[p]; k < g->offsets[p + 1]; ++k) { o = put_u64(o, g->matches[k].i, ' '); o = put_u64(o, g->matches[k].j, '\n'); }
                }
                lens[(size_t)c] = (size_t)(o - bufs[(size_t)c].data());
            } catch (...) { failed[(size_t)c] = 1; }
        });
        for (int c = 0; c < T && ok; ++c) {
            if (failed[(size_t)c]) { ok = false; break; }
            if (lens[(size_t)c]) ok &= fwrite(bufs[(size_t)c].data(), 1, lens[(size_t)c], f) == lens[(size_t)c];
        }
    }
    ok &= (fclose(f) == 0);
    return ok ? R3DM_OK : R3DM_ERR_IO;
}

static int r3dm_load_matches_impl(const char* path, r3dm_graph** out)
{
    if (!path || !out) return R3DM_ERR_INVALID;
    *out = nullptr;
    const bool bin = has_ext(path, ".bin");
    if (!bin && !has_ext(path, ".txt")) return R3DM_ERR_INVALID;
    FILE* f = fopen(path, bin ? "rb" : "r");
    if (!f) return R3DM_ERR_IO;
    std::vector<uint32_t> pairs;
    std::vector<uint64_t> offs{0};
    std::vector<r3dm_match> m;
    bool ok = true;
    if (bin) {
        uint8_t le = 0; uint64_t np = 0;
        ok = fread(&le, 1, 1, f) == 1 && le == 1 && fread(&np, 8, 1, f) == 1;
        // counts come from the file: never size anything by them beyond what the rest of the file can hold
        long fsz = -1;
        if (ok) { const long at = ftell(f); ok = at >= 0 && fseek(f, 0, SEEK_END) == 0; fsz = ok ? ftell(f) : -1; ok = ok && fseek(f, at, SEEK_SET) == 0; }
        ok = ok && np <= (uint64_t)fsz / 16;                     // every entry takes at least 16 bytes
        for (uint64_t p = 0; p < np && ok; ++p) {
            uint32_t ij[2]; uint64_t cnt = 0;
            ok = fread(ij, 4, 2, f) == 2 && fread(&cnt, 8, 1, f) == 1 && cnt < (1ull << 32);
            if (ok) { const long at = ftell(f); ok = at >= 0 && cnt <= (uint64_t)(fsz - at) / sizeof(r3dm_match); }
            if (!ok) break;
            const size_t at = m.size();
            m.resize(at + cnt);
            ok = fread(m.data() + at, sizeof(r3dm_match), cnt, f) == cnt;
            pairs.push_back(ij[0]); pairs.push_back(ij[1]); offs.push_back(m.size());
        }
    } else {
        // whitespace-separated unsigned decimals, read from one buffer holding the whole file (what fscanf("%u") accepts)
        ok = fseek(f, 0, SEEK_END) == 0;
        const long fsz = ok ? ftell(f) : -1;
        ok = ok && fsz >= 0 && fseek(f, 0, SEEK_SET) == 0;
        std::vector<char> txt(ok ? (size_t)fsz + 1 : 1);
        ok = ok && fread(txt.data(), 1, (size_t)fsz, f) == (size_t)fsz;
        const char* s = txt.data();
        const char* e = s + (ok ? (size_t)fsz : 0);
        // -> 1 number, 0 clean end of input, -1 something that is not a number (or one that does not fit `limit`)
        auto next = [&](uint64_t limit, uint64_t& v) -> int {
            while (s < e && (*s == ' ' || *s == '\n' || *s == '\r' || *s == '\t' || *s == '\v' || *s == '\f')) ++s;
            if (s == e) return 0;
            if (*s < '0' || *s > '9') return -1;
            v = 0;
            while (s < e && *s >= '0' && *s <= '9') { v = v * 10 + (uint64_t)(*s++ - '0'); if (v > limit) return -1; }
            return 1;
        };
        while (ok) {
            uint64_t I, J, cnt, a, b;
            const int r = next(0xFFFFFFFFull, I);
            if (r == 0) break;                                    // end of file between entries
            if (r < 0 || next(0xFFFFFFFFull, J) != 1 || next(0xFFFFFFFFull, cnt) != 1 || cnt > (uint64_t)(e - s) / 4 + 1) { ok = false; break; }
            const size_t at = m.size();
            m.resize(at + cnt);
            for (uint64_t k = 0; k < cnt; ++k) {
                if (next(0xFFFFFFFFull, a) != 1 || next(0xFFFFFFFFull, b) != 1) { ok = false; break; }
                m[at + k] = {(uint32_t)a, (uint32_t)b};
            }
            if (!ok) break;
            pairs.push_back((uint32_t)I); pairs.push_back((uint32_t)J); offs.push_back(m.size());
        }
    }
    fclose(f);
    if (!ok) return R3DM_ERR_IO;
    return r3dm_graph_from_csr(pairs.data(), pairs.size() / 2, offs.data(), m.data(), out);
}

extern "C" int r3dm_load_matches(const char* path, r3dm_graph** out)
{
    return r3dm_guarded(nullptr, [&]() -> int { return r3dm_load_matches_impl(path, out); });
}

