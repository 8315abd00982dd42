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
                      &c->a_jobs, &c->h_aux, &c->h_jobs, &c->a_scratch, &c->a_ids, &c->d_spill, &c->d_fb2, &c->g_segs, &c->d_verdict};
    for (FilterBufs& fb : c->fb) fb.release();
    c->coop_sched.release();
    if (c->coop_ev) (void)hipEventDestroy(c->coop_ev);
    if (c->coop_stream) (void)hipStreamDestroy(c->coop_stream);
    c->coop_ev = nullptr; c->coop_stream = nullptr;
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : c->ak_bufs) b.release();
    for (auto& im : c->spare) if (im) im->release();
    c->pin_desc.release(); c->pin_out.release(); c->pin_small.release(); c->tab_host.release(); c->tab_back.release();
    c->ring.release();
    c->arena.release_all();
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
// views: registration (the reference loads every view's regions before it matches: Regions_Provider::load,
// /root/reference/src/R3DComputeMatches.cpp:2040,2094-2095)
// ------------------------------------------------------------------------------------------------
// A view costs ONE pass over its descriptors: host rows -> page-locked ring slot (the caller's pageable memory is consumed when the call
// returns) -> one DMA -> one kernel that writes the fragment-order tiles, the norms, the statistics and the table entry.  No
// synchronisation per view: the statistics the host needs to choose a matching path (integer-valued? largest element? votes x scale?)
// stay in the table until the first call that asks (sync_view_stats), and the layouts only some paths read -- row-major rows, bf16
// tiles, split planes, count tiles, byte tiles -- are staged by that path's first call (ensure_layouts).
static void fill_entry(const HostImage& h, ImgDev& d)
{
    d = ImgDev{};
    d.rows = (h.have & kLayRows) ? h.rows.as<float>() : nullptr;
    d.tiled = h.tiled.as<float>(); d.norms = h.norms.as<float>();
    d.bin = h.bin.as<uint32_t>(); d.xy = h.has_xy ? h.xy.as<float>() : nullptr;
    d.canon = h.has_xy ? h.canon.as<uint32_t>() : nullptr;
    d.has_dup = h.has_dup ? 1u : 0u;
    d.n = h.n; d.n_tiles = h.n_tiles; d.dim = h.dim; d.G = h.G; d.words = h.words;
    d.width = h.width; d.height = h.height;
    d.max_norm_bits = h.stats_valid ? h.stat_bits[0] : 0u; d.max_abs_bits = h.stats_valid ? h.stat_bits[1] : 0u; d.not_integer = h.stats_valid ? h.stat_bits[2] : 0u;
    d.ann_adj = h.ann_K ? h.ann_adj.as<uint32_t>() : nullptr; d.ann_deg = h.ann_K ? h.ann_deg.as<uint32_t>() : nullptr;
    d.ann_rows16 = h.compact_ready ? h.ann_rows16.as<uint16_t>() : nullptr; d.ann_rows8 = h.compact_ready ? h.ann_rows8.as<uint8_t>() : nullptr;
    d.tiled16 = (h.have & kLayBf16) ? h.tiled16.as<uint16_t>() : nullptr;
    d.tiledh = (h.have & kLaySplit) ? h.tiledh.as<uint16_t>() : nullptr; d.split_k = h.split_k;
    const bool cn = (h.have & kLayCounts) != 0;
    d.tiledc = cn ? h.tiledc.as<uint16_t>() : nullptr; d.cscale = cn ? h.cscale.as<float>() : nullptr; d.cquad = cn ? h.cquad.as<float>() : nullptr;
    d.tiledp = cn ? h.tiledp.as<uint16_t>() : nullptr; d.cperm = cn ? h.cperm.as<uint32_t>() : nullptr; d.counts_fail = (cn && h.counts_ok) ? 0u : 1u;
    d.tiled8 = (h.have & kLayBin8) ? h.tiled8.as<uint8_t>() : nullptr;
}

// room for `slots` entries in the device table and its page-locked mirror.  Growing re-uploads every live entry: the statistics of
// views staged since the last read live only in the old table, so they are read back first.
static int table_reserve(r3dm_ctx* c, size_t slots)
{
    const size_t need = sizeof(ImgDev) * slots;
    if (need <= c->d_imgs.cap && need <= c->tab_host.cap) return R3DM_OK;
    int rc = sync_view_stats(c);
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    const size_t cap_slots = std::max<size_t>(256, slots * 2);
    DevBuf nb;
    R3DM_HIP(c, nb.ensure(sizeof(ImgDev) * cap_slots));
    R3DM_HIP(c, hipMemsetAsync(nb.p, 0, nb.cap, c->stream));
    PinBuf np;
    R3DM_HIP(c, np.ensure(sizeof(ImgDev) * cap_slots));
    std::memset(np.p, 0, np.cap);
    const size_t live = std::min(c->imgs.size(), cap_slots);
    for (size_t k = 0; k < live; ++k) if (c->imgs[k]) fill_entry(*c->imgs[k], np.as<ImgDev>()[k]);
    if (live) R3DM_HIP(c, hipMemcpyAsync(nb.p, np.p, live * sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    c->d_imgs.release(); c->d_imgs = nb;
    c->tab_host.release(); c->tab_host = np;
    return R3DM_OK;
}

int publish_entry(r3dm_ctx* c, uint32_t slot)
{
    int rc = table_reserve(c, c->imgs.size());
    if (rc != R3DM_OK) return rc;
    ImgDev* m = c->tab_host.as<ImgDev>() + slot;
    fill_entry(*c->imgs[slot], *m);
    R3DM_HIP(c, hipMemcpyAsync(c->d_imgs.as<ImgDev>() + slot, m, sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
    return R3DM_OK;
}

static inline int32_t split_k_of(float mx)
{
    // the same function of max|x| as stage_split_kernel's: max|x| 2^k in [2^13, 2^14)
    int k = 0;
    if (mx > 0.0f && std::isfinite(mx)) {
        uint32_t b; std::memcpy(&b, &mx, 4);
        k = 13 - ((int)((b >> 23) & 0xFFu) - 127);
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
    }
    return k;
}

int sync_view_stats(r3dm_ctx* c)
{
    if (c->pending_stats.empty()) return R3DM_OK;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint32_t s : c->pending_stats) if (s < c->imgs.size()) { lo = std::min(lo, s); hi = std::max(hi, s); }
    if (lo <= hi) {
        const size_t cnt = (size_t)hi - lo + 1;
        R3DM_HIP(c, c->tab_back.ensure(cnt * sizeof(ImgDev)));
        R3DM_HIP(c, hipMemcpyAsync(c->tab_back.p, c->d_imgs.as<ImgDev>() + lo, cnt * sizeof(ImgDev), hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        const ImgDev* t = c->tab_back.as<ImgDev>();
        for (uint32_t s : c->pending_stats) {
            if (s >= c->imgs.size() || !c->imgs[s]) continue;
            HostImage& h = *c->imgs[s];
            if (h.stats_valid) continue;
            const ImgDev& d = t[s - lo];
            h.stat_bits[0] = d.max_norm_bits; h.stat_bits[1] = d.max_abs_bits; h.stat_bits[2] = d.not_integer;
            std::memcpy(&h.max_abs, &h.stat_bits[1], 4);
            h.not_integer = (d.not_integer & 1u) != 0;
            h.has_negative = (d.not_integer & 2u) != 0;
            h.has_dup = d.has_dup != 0;
            h.split_k = split_k_of(h.max_abs);
            if (h.have & kLayCounts) h.counts_ok = d.counts_fail == 0;
            h.stats_valid = true;
            if (h.dtype == R3DM_BIN) { h.not_integer = true; h.has_negative = true; }
            // (the table entry itself already holds all of this: the kernels wrote it there)
            ImgDev* m = c->tab_host.as<ImgDev>() + s;
            m->max_norm_bits = d.max_norm_bits; m->max_abs_bits = d.max_abs_bits; m->not_integer = d.not_integer; m->has_dup = d.has_dup; m->counts_fail = d.counts_fail;
        }
    }
    c->pending_stats.clear();
    return R3DM_OK;
}

// ---- on-demand layouts
// launches the staging kernels of the layouts in `want` the image does not hold (allocating their buffers); the caller publishes the entry
int ensure_layouts_image(r3dm_ctx* c, HostImage& h, uint32_t want)
{
    if (h.dtype == R3DM_BIN) {
        want &= kLayBin8;
        if ((want & ~h.have) & kLayBin8) {
            const size_t t8_bytes = (size_t)h.n_tiles * h.words * 1024 + 2 * kSlackBytes;
            const size_t nrm_bytes = (size_t)h.n_tiles * 32 * 4 + kSlackBytes;
            R3DM_HIP(c, h.tiled8.ensure(t8_bytes));
            R3DM_HIP(c, h.norms.ensure(nrm_bytes));
            R3DM_HIP(c, hipMemsetAsync(h.tiled8.p, 0, t8_bytes, c->stream));
            R3DM_HIP(c, hipMemsetAsync(h.norms.p, 0x7F, nrm_bytes, c->stream));
            R3DM_HIP(c, launch_stage_bin8(c->stream, h.bin.as<uint32_t>(), h.n, h.words, h.n_tiles, h.tiled8.as<uint8_t>(), h.norms.as<float>()));
            h.have |= kLayBin8;
        }
        return R3DM_OK;
    }
    want &= (kLayRows | kLayBf16 | kLaySplit | kLayCounts);
    if (want & (kLayCounts | kLaySplit)) want |= kLayRows;
    if ((want & kLayCounts) && !(h.dtype == R3DM_F32 && h.n && h.dim <= 256)) want &= ~kLayCounts;      // never votes x scale: counts_ok stays false
    const uint32_t todo = want & ~h.have;
    if (!todo) return R3DM_OK;
    const uint32_t GB = (h.G + 1) / 2;
    if (todo & kLayRows) {
        R3DM_HIP(c, h.rows.ensure((size_t)std::max<uint32_t>(h.n, 1) * h.dim * 4 + 256));
        R3DM_HIP(c, launch_untile_rows(c->stream, h.tiled.as<float>(), h.n, h.dim, h.G, h.n_tiles, h.rows.as<float>()));
        h.have |= kLayRows;
    }
    if (todo & kLayBf16) {
        const size_t bytes = (size_t)h.n_tiles * GB * 1024 + kSlackBytes;
        R3DM_HIP(c, h.tiled16.ensure(bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiled16.p, 0, bytes, c->stream));
        R3DM_HIP(c, launch_stage_bf16(c->stream, h.tiled.as<float>(), h.G, h.n_tiles, h.tiled16.as<uint16_t>()));
        h.have |= kLayBf16;
    }
    if (todo & kLaySplit) {
        // f16 hi | lo planes of the values scaled by 2^split_k; the kernel derives split_k from max|x| in `stats` exactly as split_k_of does
        const size_t bytes = (size_t)h.n_tiles * GB * 2048 + kSlackBytes;
        R3DM_HIP(c, h.tiledh.ensure(bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiledh.p, 0, bytes, c->stream));
        R3DM_HIP(c, c->d_cnt.ensure(64));
        uint32_t* st3 = c->d_cnt.as<uint32_t>() + 12;             // (words 12 .. 15 of the counter block: statistics in, split_k out)
        R3DM_HIP(c, hipMemcpyAsync(st3, h.stat_bits, 12, hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, launch_stage_split(c->stream, h.rows.as<float>(), h.n, h.dim, GB, h.n_tiles, h.tiledh.as<uint16_t>(), st3, reinterpret_cast<int32_t*>(st3 + 3)));
        h.have |= kLaySplit;
    }
    if (todo & kLayCounts) {
        // count tiles (rows = small integers x a row scale: LIOP): f16 integers, half the bytes of the split planes, + a scale per row
        const size_t bytes = (size_t)h.n_tiles * GB * 1024 + kSlackBytes;
        R3DM_HIP(c, h.tiledc.ensure(bytes));
        R3DM_HIP(c, hipMemsetAsync(h.tiledc.p, 0, bytes, c->stream));
        R3DM_HIP(c, h.cscale.ensure((size_t)h.n_tiles * 32 * 4 + kSlackBytes));
        R3DM_HIP(c, hipMemsetAsync(h.cscale.p, 0, (size_t)h.n_tiles * 32 * 4 + kSlackBytes, c->stream));
        R3DM_HIP(c, h.cquad.ensure((counts_summary_offset(h.n_tiles) + ((size_t)h.n_tiles + 1) * 16) * 4 + kSlackBytes));
        R3DM_HIP(c, h.tiledp.ensure(bytes));
        R3DM_HIP(c, h.cperm.ensure((size_t)h.n_tiles * 32 * 4 + 256));
        h.have |= kLayCounts;
        h.counts_ok = false;                                      // until the check's verdict is read (ensure_layouts)
    }
    return R3DM_OK;
}

// the count tiles' kernels write their verdict into a word the caller names (the table entry's counts_fail, or scratch)
static int launch_counts_of(r3dm_ctx* c, HostImage& h, uint32_t* fail_dev)
{
    R3DM_HIP(c, hipMemsetAsync(fail_dev, 0, 4, c->stream));
    R3DM_HIP(c, launch_stage_counts(c->stream, h.rows.as<float>(), h.n, h.dim, (h.G + 1) / 2, h.n_tiles, h.tiledc.as<uint16_t>(), h.cscale.as<float>(), h.norms.as<float>(),
                                    h.tiledp.as<uint16_t>(), h.cquad.as<float>(), h.cperm.as<uint32_t>(), fail_dev));
    return R3DM_OK;
}

int ensure_layouts(r3dm_ctx* c, std::vector<uint32_t> slots, uint32_t want, bool* counts_all_ok)
{
    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    int rc = sync_view_stats(c);                               // entries are republished below: the host must hold their statistics
    if (rc != R3DM_OK) return rc;
    // verdict words of count-tile checks launched here: one per slot, in scratch (an entry republished behind the kernel would overwrite its own)
    std::vector<uint32_t> checked;
    std::vector<std::unique_lock<std::mutex>> locks;
    for (uint32_t s : slots) {
        HostImage& m = *c->imgs[s];
        HostImage* h = &m;
        if (m.borrowed && m.owner) {
            // a mounted index: its layouts belong to the index and are added there, under its lock (searches from other contexts run
            // on the buffers it already holds)
            uint32_t w = want;
            if (m.dtype != R3DM_BIN && (w & (kLayCounts | kLaySplit))) w |= kLayRows;
            if ((w & ~m.owner->img.have) == 0 && (w & ~m.have) == 0) continue;
            locks.emplace_back(m.owner->mu);
            h = &m.owner->img;
        }
        const uint32_t before = h->have;
        rc = ensure_layouts_image(c, *h, want);
        if (rc != R3DM_OK) return rc;
        if ((h->have & ~before) & kLayCounts) checked.push_back(s);
        if (h != &m) {
            m.rows = h->rows; m.tiled16 = h->tiled16; m.tiledh = h->tiledh; m.tiledc = h->tiledc; m.tiledp = h->tiledp; m.cscale = h->cscale; m.cquad = h->cquad;
            m.cperm = h->cperm; m.tiled8 = h->tiled8; m.norms = h->norms; m.have = h->have; m.counts_ok = h->counts_ok;
        }
        if (h->have != before || h != &m) { rc = publish_entry(c, s); if (rc != R3DM_OK) return rc; }
    }
    if (!checked.empty()) {
        R3DM_HIP(c, c->d_verdict.ensure(checked.size() * 4 + 64));
        R3DM_HIP(c, c->pin_small.ensure(checked.size() * 4 + 64));
        for (size_t k = 0; k < checked.size(); ++k) {
            HostImage& m = *c->imgs[checked[k]];
            rc = launch_counts_of(c, (m.borrowed && m.owner) ? m.owner->img : m, c->d_verdict.as<uint32_t>() + k);
            if (rc != R3DM_OK) return rc;
        }
        R3DM_HIP(c, hipMemcpyAsync(c->pin_small.p, c->d_verdict.p, checked.size() * 4, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        const uint32_t* v = c->pin_small.as<uint32_t>();
        for (size_t k = 0; k < checked.size(); ++k) {
            HostImage& m = *c->imgs[checked[k]];
            m.counts_ok = v[k] == 0;
            if (m.borrowed && m.owner) m.owner->img.counts_ok = m.counts_ok;
        }
    }
    if (!locks.empty()) R3DM_HIP(c, hipStreamSynchronize(c->stream));     // other contexts may read the index's new layouts once the locks are gone
    if (counts_all_ok) {
        *counts_all_ok = true;
        for (uint32_t s : slots) if (!((c->imgs[s]->have & kLayCounts) && c->imgs[s]->counts_ok)) { *counts_all_ok = false; break; }
    }
    return R3DM_OK;
}

// ---- the upload ring
enum { kSrcPageable = 0, kSrcPinned = 1, kSrcDevice = 2, kSrcPeer = 3 };
// where a caller's buffer lives: pageable host memory (copied into the ring's page-locked slot by the host first); page-locked host
// memory, memory of THIS device, memory of another device (all three copied by the runtime straight into the ring's device slot)
static int source_kind(const void* p, const void** dev_ptr, int this_device)
{
    *dev_ptr = p;
    if (!p) return kSrcPageable;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return kSrcPageable; }
    if (at.type == hipMemoryTypeDevice) return at.device == this_device ? kSrcDevice : kSrcPeer;
    if (at.type == hipMemoryTypeManaged) return kSrcDevice;
    if (at.type == hipMemoryTypeHost && at.devicePointer) { *dev_ptr = at.devicePointer; return kSrcPinned; }
    return kSrcPageable;
}

// next ring slot, free to be overwritten (its last view's kernels have run)
static int ring_acquire(r3dm_ctx* c, int* slot_out)
{
    UploadRing& r = c->ring;
    const int s = (int)(r.next++ % UploadRing::kSlots);
    if (!r.ev[s]) R3DM_HIP(c, hipEventCreateWithFlags(&r.ev[s], hipEventDisableTiming));
    if (r.busy[s]) { R3DM_HIP(c, hipEventSynchronize(r.ev[s])); r.busy[s] = false; }
    *slot_out = s;
    return R3DM_OK;
}

struct ViewSrc {
    uint32_t view_id, width, height;
    const void* desc; uint32_t n, dim; r3dm_dtype dtype; const float* xy;
};

static inline size_t desc_bytes_of(const ViewSrc& v) { return (size_t)v.n * v.dim * (v.dtype == R3DM_F32 ? 4 : 1); }
static inline size_t ring_xy_offset(const ViewSrc& v) { return (desc_bytes_of(v) + 255) / 256 * 256; }

// host rows -> the ring slot's page-locked buffer (callable from helper threads: touches nothing but the slot)
static void ring_fill(const UploadRing& r, int s, const ViewSrc& v, bool desc_from_host, bool xy_from_host)
{
    unsigned char* dst = r.pin[s].as<unsigned char>();
    if (desc_from_host && v.n) std::memcpy(dst, v.desc, desc_bytes_of(v));
    if (xy_from_host && v.xy && v.n) std::memcpy(dst + ring_xy_offset(v), v.xy, (size_t)v.n * 8);
}

// Stage one view into `slot`.  `s` = its ring slot (acquired; the page-locked buffer already holds the host rows when filled = true).
// Leaves the stream with: [DMA of the slot] -> staging kernel(s) -> the slot's event.  Nothing is waited for.
static int stage_enqueue(r3dm_ctx* c, uint32_t slot, const ViewSrc& v, int s, int kind_desc, const void* dev_desc, int kind_xy, const void* dev_xy, bool filled, bool* copy_wait_out)
{
    HostImage& h = *c->imgs[slot];
    UploadRing& r = c->ring;
    h.view_id = v.view_id; h.n = v.n; h.dim = v.dim; h.dtype = v.dtype; h.width = v.width; h.height = v.height;
    h.has_xy = (v.xy != nullptr); h.has_dup = false; h.live = true;
    h.G = 0; h.n_tiles = 0; h.words = 0; h.ann_K = 0; h.hnsw_M = 0; h.mrpt_trees = 0; h.compact_ready = false;
    h.have = 0; h.stats_valid = false; h.counts_ok = false; h.split_k = 0; h.max_abs = 0.0f; h.not_integer = true; h.has_negative = true;
    const uint32_t n = v.n, dim = v.dim;
    const size_t dbytes = desc_bytes_of(v), xy_off = ring_xy_offset(v), xy_bytes = v.xy ? (size_t)n * 8 : 0;
    const bool desc_ring = kind_desc == kSrcPageable && n, xy_ring = v.xy && kind_xy == kSrcPageable && n;
    // ---- every source ends in the slot's device buffer, by a copy on the ring's copy stream: pageable rows from the page-locked slot
    // the host filled, anything else (device memory of this or another GPU, page-locked host memory) straight from where the caller
    // keeps it -- the caller's buffers are consumed when THAT copy is done (copy_event_out), not when the staging kernels are; the
    // kernels of this view (context's stream) wait for the slot's copy event.  The slot's previous copies and the kernel that read
    // them are done: the host waited for the slot's event before the slot was handed out.  (Round 6 first read device sources in
    // place and waited for the kernels: 0.3-0.5 ms per view of a features worker's time in the stage.)
    const void* raw = nullptr; const float* xy_src = nullptr;
    (void)dev_desc; (void)dev_xy;
    *copy_wait_out = false;
    if (n) {
        R3DM_HIP(c, r.raw[s].ensure(xy_off + xy_bytes + 256));
        if (!r.copy_stream) R3DM_HIP(c, hipStreamCreateWithFlags(&r.copy_stream, hipStreamNonBlocking));
        if (!r.ev_copy[s]) R3DM_HIP(c, hipEventCreateWithFlags(&r.ev_copy[s], hipEventDisableTiming));
        unsigned char* base = r.raw[s].as<unsigned char>();
        if (desc_ring || xy_ring) {
            if (!filled) {
                R3DM_HIP(c, r.pin[s].ensure(xy_off + xy_bytes + 256));
                ring_fill(r, s, v, desc_ring, xy_ring);
            }
            // one DMA for rows + positions when both are in the slot, else the part that is
            const size_t from = desc_ring ? 0 : xy_off, to = xy_ring ? xy_off + xy_bytes : dbytes;
            R3DM_HIP(c, hipMemcpyAsync(base + from, r.pin[s].as<unsigned char>() + from, to - from, hipMemcpyHostToDevice, r.copy_stream));
            c->n_ring_uploads += 1;
        } else c->n_direct_uploads += 1;
        if (!desc_ring) { R3DM_HIP(c, hipMemcpyAsync(base, v.desc, dbytes, hipMemcpyDefault, r.copy_stream)); *copy_wait_out = true; }
        if (v.xy && !xy_ring) { R3DM_HIP(c, hipMemcpyAsync(base + xy_off, v.xy, xy_bytes, hipMemcpyDefault, r.copy_stream)); *copy_wait_out = true; }
        R3DM_HIP(c, hipEventRecord(r.ev_copy[s], r.copy_stream));
        R3DM_HIP(c, hipStreamWaitEvent(c->stream, r.ev_copy[s], 0));
        raw = base;
        if (v.xy) xy_src = (const float*)(base + xy_off);
    } else c->n_direct_uploads += 1;

    int rc = table_reserve(c, c->imgs.size());
    if (rc != R3DM_OK) return rc;
    ImgDev* entry_dev = c->d_imgs.as<ImgDev>() + slot;
    StageViewArgs A{};
    A.n = n; A.dim = dim;
    if (v.xy) {
        R3DM_HIP(c, h.xy.ensure((size_t)std::max<uint32_t>(n, 1) * 8));
        R3DM_HIP(c, h.canon.ensure((size_t)std::max<uint32_t>(n, 1) * 4));
        A.xy_src = n ? xy_src : nullptr; A.xy_dst = h.xy.as<float>(); A.canon_dst = h.canon.as<uint32_t>(); A.has_dup = &entry_dev->has_dup;
        if (n > 1) {
            // position classes: a hash table of >= 2 n slots per view in flight
            uint32_t bits = 4;
            while ((1u << bits) < 2u * n) ++bits;
            const size_t tab_bytes = ((size_t)12) << bits;
            R3DM_HIP(c, r.ctab[s].ensure(tab_bytes));
            R3DM_HIP(c, hipMemsetAsync(r.ctab[s].p, 0xFF, tab_bytes, c->stream));
            A.canon_keys = r.ctab[s].as<unsigned long long>(); A.canon_vals = reinterpret_cast<uint32_t*>(r.ctab[s].as<unsigned char>() + ((size_t)8 << bits)); A.canon_bits = bits;
        }
    }
    // layouts of the paths that are switched on already are staged behind the view (else: their first call stages them)
    uint32_t eager = 0;
    if (v.dtype == R3DM_BIN) { if (c->hamming_mfma) eager |= kLayBin8; }
    else {
        // (the bf16 tiles of r3dm_set_integer_mfma are never staged here: whether a view is integer-valued is a statistic of the very
        //  kernel this call queues -- the facade switches every exact path on and registers LIOP rows -- so they wait for a first match call
        //  that can use them, 12 us per view)
        if (c->split_mfma) eager |= kLayRows | ((v.dtype == R3DM_F32 && n && dim <= 256) ? kLayCounts : 0u);
    }
    if (v.dtype == R3DM_BIN) {
        h.words = (dim + 3) / 4;
        R3DM_HIP(c, h.bin.ensure((size_t)(n + 8) * h.words * 4 + kSlackBytes));
        h.n_tiles = (n + kTileRows - 1) / kTileRows;
    } else {
        h.G = kernel_G_for(dim);
        h.n_tiles = (n + kTileRows - 1) / kTileRows;
        R3DM_HIP(c, h.tiled.ensure((size_t)h.n_tiles * h.G * 1024 + kSlackBytes));
        R3DM_HIP(c, h.norms.ensure((size_t)h.n_tiles * 32 * 4 + kSlackBytes));
        if (eager & kLayRows) {                                   // written by the staging kernel itself, from the tile it holds in LDS
            R3DM_HIP(c, h.rows.ensure((size_t)std::max<uint32_t>(n, 1) * dim * 4 + 256));
            h.have |= kLayRows;
        }
        // the count tiles' buffers exist before the entry goes up (one upload carries every pointer); their kernels follow the staging kernel
        if (eager & kLayCounts) { rc = ensure_layouts_image(c, h, kLayCounts); if (rc != R3DM_OK) return rc; }
    }
    // the table entry: written once from the mirror with the statistics zeroed; the kernels accumulate into it
    ImgDev* m = c->tab_host.as<ImgDev>() + slot;
    fill_entry(h, *m);
    R3DM_HIP(c, hipMemcpyAsync(entry_dev, m, sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
    if (v.xy && n == 1) R3DM_HIP(c, hipMemsetAsync(h.canon.p, 0, 4, c->stream));          // one feature: its own class
    if (v.dtype == R3DM_BIN) {
        R3DM_HIP(c, launch_stage_bin(c->stream, (const uint8_t*)raw, n, dim, h.bin.as<uint32_t>(), h.words, n + 8));
        R3DM_HIP(c, launch_stage_positions(c->stream, A));
    } else {
        A.raw = raw; A.raw_is_u8 = v.dtype == R3DM_U8; A.G = h.G; A.n_tiles = h.n_tiles;
        A.tiled = h.tiled.as<float>(); A.norms = h.norms.as<float>(); A.rows = (h.have & kLayRows) ? h.rows.as<float>() : nullptr;
        A.img_stats = &entry_dev->max_norm_bits;
        R3DM_HIP(c, launch_stage_view(c->stream, A));
    }
    if (v.dtype == R3DM_BIN && (eager & ~h.have)) {
        // byte tiles are made from the word rows the kernel above wrote: staged behind it, their two pointers patched into the entry
        // (pointer fields only: the statistics words of the entry belong to the kernels until sync_view_stats)
        rc = ensure_layouts_image(c, h, eager);
        if (rc != R3DM_OK) return rc;
        ImgDev e; fill_entry(h, e);
        m->tiled8 = e.tiled8; m->norms = e.norms;
        R3DM_HIP(c, hipMemcpyAsync((void*)&entry_dev->norms, &m->norms, sizeof(void*), hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, hipMemcpyAsync((void*)&entry_dev->tiled8, &m->tiled8, sizeof(void*), hipMemcpyHostToDevice, c->stream));
    }
    if (v.dtype != R3DM_BIN && (h.have & kLayCounts)) { rc = launch_counts_of(c, h, &entry_dev->counts_fail); if (rc != R3DM_OK) return rc; }
    R3DM_HIP(c, hipEventRecord(r.ev[s], c->stream));
    r.busy[s] = true;
    c->pending_stats.push_back(slot);
    c->n_views_staged += 1;
    return R3DM_OK;
}

static int check_view(r3dm_ctx* c, const ViewSrc& v)
{
    if (v.dim == 0 || (v.n && !v.desc)) return R3DM_ERR_INVALID;
    if (v.dtype != R3DM_F32 && v.dtype != R3DM_U8 && v.dtype != R3DM_BIN) return R3DM_ERR_INVALID;
    if (v.n >= (1u << 22)) { c->err = "more than 4M features in one view"; return R3DM_ERR_UNSUPPORTED; }
    if (v.dtype == R3DM_BIN && !(((v.dim + 3) / 4) == 8 || ((v.dim + 3) / 4) == 16)) {
        c->err = "binary descriptors must be 29..32 or 61..64 bytes"; return R3DM_ERR_UNSUPPORTED;
    }
    return R3DM_OK;
}

// copy + re-layout one view into slot `slot`.  Returns with the caller's buffers consumed: host rows sit in the ring; device (or
// page-locked) buffers the kernel reads in place are waited for.
int stage_into_slot(r3dm_ctx* c, uint32_t slot, uint32_t view_id, uint32_t width, uint32_t height,
                    const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    const ViewSrc v{view_id, width, height, desc, n, dim, dtype, xy};
    const void *dd = nullptr, *dx = nullptr;
    const int kd = source_kind(desc, &dd, c->device), kx = source_kind(xy, &dx, c->device);
    int s = 0;
    int rc = ring_acquire(c, &s);
    if (rc != R3DM_OK) return rc;
    bool wait_copy = false;
    rc = stage_enqueue(c, slot, v, s, kd, dd, kx, dx, false, &wait_copy);
    if (rc != R3DM_OK) return rc;
    if (wait_copy) R3DM_HIP(c, hipEventSynchronize(c->ring.ev_copy[s]));      // the caller's device / page-locked buffers have been read
    return R3DM_OK;
}

static uint32_t slot_for_view(r3dm_ctx* c, uint32_t view_id)
{
    auto it = c->slot_of.find(view_id);
    if (it != c->slot_of.end()) return it->second;
    const uint32_t slot = (uint32_t)c->imgs.size();
    if (!c->spare.empty()) { c->imgs.emplace_back(std::move(c->spare.back())); c->spare.pop_back(); }      // buffers of a cleared view
    else { c->imgs.emplace_back(new HostImage()); c->imgs.back()->use_arena(&c->arena); }
    c->slot_of[view_id] = slot;
    return slot;
}

static int r3dm_set_image_impl(r3dm_ctx* c, uint32_t view_id, uint32_t width, uint32_t height,
                              const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    if (!c) return R3DM_ERR_INVALID;
    const ViewSrc v{view_id, width, height, desc, n, dim, dtype, xy};
    int rc = check_view(c, v);
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipSetDevice(c->device));
    return stage_into_slot(c, slot_for_view(c, view_id), view_id, width, height, desc, n, dim, dtype, xy);
}

extern "C" int r3dm_set_image(r3dm_ctx* c, uint32_t view_id, uint32_t width, uint32_t height,
                              const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_set_image_impl(c, view_id, width, height, desc, n, dim, dtype, xy); });
}

// A whole collection in one call: helper threads copy the pageable rows of the next views into ring slots while the DMA and the kernels
// of the previous ones run; one synchronisation at the end, and only when some view was read where the caller keeps it.
static int r3dm_set_images_impl(r3dm_ctx* c, const r3dm_view_desc* views, uint32_t n_views)
{
    if (!c || (n_views && !views)) return R3DM_ERR_INVALID;
    if (n_views == 0) return R3DM_OK;
    R3DM_HIP(c, hipSetDevice(c->device));
    std::vector<ViewSrc> vs(n_views);
    std::vector<int> kd(n_views), kx(n_views);
    std::vector<const void*> dd(n_views), dx(n_views);
    std::vector<uint32_t> slots(n_views);
    for (uint32_t k = 0; k < n_views; ++k) {
        const r3dm_view_desc& w = views[k];
        vs[k] = ViewSrc{w.view_id, w.width, w.height, w.desc, w.n, w.dim, (r3dm_dtype)w.dtype, w.xy};
        int rc = check_view(c, vs[k]);
        if (rc != R3DM_OK) return rc;
        kd[k] = source_kind(w.desc, &dd[k], c->device); kx[k] = source_kind(w.xy, &dx[k], c->device);
    }
    for (uint32_t k = 0; k < n_views; ++k) slots[k] = slot_for_view(c, vs[k].view_id);
    int rc = table_reserve(c, c->imgs.size());
    if (rc != R3DM_OK) return rc;
    UploadRing& r = c->ring;
    constexpr int S = UploadRing::kSlots;
    // every slot's page-locked buffer sized for the largest view before the helpers start (they only memcpy)
    size_t max_slot = 0;
    bool any_host = false, any_inplace = false;
    for (uint32_t k = 0; k < n_views; ++k) {
        const bool dh = kd[k] == kSrcPageable && vs[k].n, xh = vs[k].xy && kx[k] == kSrcPageable && vs[k].n;
        if (dh || xh) { any_host = true; max_slot = std::max(max_slot, ring_xy_offset(vs[k]) + (size_t)vs[k].n * 8 + 256); }
        if ((vs[k].n && kd[k] != kSrcPageable) || (vs[k].xy && vs[k].n && kx[k] != kSrcPageable)) any_inplace = true;
    }
    if (any_host) {
        for (int s = 0; s < S; ++s) {
            if (r.busy[s]) { R3DM_HIP(c, hipEventSynchronize(r.ev[s])); r.busy[s] = false; }
            R3DM_HIP(c, r.pin[s].ensure(max_slot));
        }
    }
    // view k uses ring slot (base + k) % S.  state[k]: 0 waiting, 1 filled by a helper, 2 enqueued (its event is recorded)
    const uint64_t base = r.next;
    std::vector<std::atomic<int>> state(n_views);
    for (auto& a : state) a.store(0, std::memory_order_relaxed);
    std::atomic<uint32_t> next_fill{0};
    std::atomic<bool> abort{false};
    const int T = any_host ? r3dm_host_team(4, 1) : 0;
    auto helper = [&]() {
        (void)hipSetDevice(c->device);
        for (;;) {
            const uint32_t k = next_fill.fetch_add(1, std::memory_order_relaxed);
            if (k >= n_views || abort.load(std::memory_order_relaxed)) break;
            const int s = (int)((base + k) % S);
            // the slot's previous user in this batch must have been enqueued, and its kernels done
            if (k >= (uint32_t)S) {
                while (state[k - S].load(std::memory_order_acquire) != 2) { if (abort.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
                (void)hipEventSynchronize(r.ev[s]);
            }
            ring_fill(r, s, vs[k], kd[k] == kSrcPageable, kx[k] == kSrcPageable);
            state[k].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    try { for (int t = 0; t < T; ++t) th.emplace_back(helper); } catch (...) {}
    const bool helpers = !th.empty();
    int last_copy_slot = -1;
    rc = R3DM_OK;
    for (uint32_t k = 0; k < n_views && rc == R3DM_OK; ++k) {
        const int s = (int)((base + k) % S);
        if (!r.ev[s]) { hipError_t e = hipEventCreateWithFlags(&r.ev[s], hipEventDisableTiming); if (e != hipSuccess) { c->err = hipGetErrorString(e); rc = R3DM_ERR_HIP; break; } }
        bool filled = false;
        if (helpers) { while (state[k].load(std::memory_order_acquire) != 1) std::this_thread::yield(); filled = true; }
        else if (r.busy[s]) { if (hipEventSynchronize(r.ev[s]) != hipSuccess) { c->err = "hipEventSynchronize"; rc = R3DM_ERR_HIP; break; } r.busy[s] = false; }
        bool wc = false;
        rc = stage_enqueue(c, slots[k], vs[k], s, kd[k], dd[k], kx[k], dx[k], filled, &wc);
        if (wc) last_copy_slot = s;
        state[k].store(2, std::memory_order_release);
    }
    if (rc != R3DM_OK) { abort.store(true); for (auto& a : state) a.store(2, std::memory_order_release); }
    for (std::thread& t : th) t.join();
    r.next = base + n_views;
    if (rc != R3DM_OK) return rc;
    // buffers the copy stream read where the caller keeps them: consumed when the last such copy is done (the stream runs in order)
    if (any_inplace && last_copy_slot >= 0) R3DM_HIP(c, hipEventSynchronize(r.ev_copy[last_copy_slot]));
    return R3DM_OK;
}

extern "C" int r3dm_set_images(r3dm_ctx* c, const r3dm_view_desc* views, uint32_t n_views)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_set_images_impl(c, views, n_views); });
}

extern "C" int r3dm_view_info(r3dm_ctx* c, uint32_t view_id, uint32_t* layouts, uint64_t* bytes, uint64_t* ring_uploads, uint64_t* direct_uploads)
{
    if (!c) return R3DM_ERR_INVALID;
    auto it = c->slot_of.find(view_id);
    if (it == c->slot_of.end()) { c->err = "unregistered view"; return R3DM_ERR_INVALID; }
    const HostImage& h = *c->imgs[it->second];
    static_assert(kLayRows == R3DM_LAYOUT_ROWS && kLayBf16 == R3DM_LAYOUT_BF16 && kLaySplit == R3DM_LAYOUT_SPLIT && kLayCounts == R3DM_LAYOUT_COUNTS &&
                  kLayBin8 == R3DM_LAYOUT_BIN8, "public layout bits");
    if (layouts) *layouts = h.have;
    if (bytes) {
        // what the view's live layouts occupy (a recycled buffer may be larger than its present tenant; buffers of layouts the view
        // does not hold -- kept from an earlier tenant -- are not counted)
        uint64_t b = 0;
        if (h.dtype == R3DM_BIN) b += h.bin.cap; else b += h.tiled.cap + h.norms.cap;
        if (h.has_xy) b += h.xy.cap + h.canon.cap;
        if (h.have & kLayRows) b += h.rows.cap;
        if (h.have & kLayBf16) b += h.tiled16.cap;
        if (h.have & kLaySplit) b += h.tiledh.cap;
        if (h.have & kLayCounts) b += h.tiledc.cap + h.tiledp.cap + h.cscale.cap + h.cquad.cap + h.cperm.cap;
        if (h.have & kLayBin8) b += h.tiled8.cap + h.norms.cap;
        if (h.ann_K) b += h.ann_adj.cap + h.ann_deg.cap;
        if (h.compact_ready) b += h.ann_rows16.cap + h.ann_rows8.cap;
        if (h.hnsw_M) b += h.hnsw_l0.cap + h.hnsw_up_off.cap + h.hnsw_up.cap;
        if (h.mrpt_trees) b += h.mrpt_R.cap + h.mrpt_RT.cap + h.mrpt_splits.cap + h.mrpt_leaves.cap + h.mrpt_lf.cap;
        *bytes = b;
    }
    if (ring_uploads) *ring_uploads = c->n_ring_uploads;
    if (direct_uploads) *direct_uploads = c->n_direct_uploads;
    return R3DM_OK;
}

extern "C" int r3dm_memory_info(const r3dm_ctx* c, uint64_t* views_device_bytes, uint64_t* ring_device_bytes, uint64_t* ring_host_bytes)
{
    if (!c) return R3DM_ERR_INVALID;
    if (views_device_bytes) *views_device_bytes = c->arena.bytes_held();
    uint64_t rd = 0, rh = 0;
    for (int k = 0; k < UploadRing::kSlots; ++k) { rd += c->ring.raw[k].cap + c->ring.ctab[k].cap; rh += c->ring.pin[k].cap; }
    if (ring_device_bytes) *ring_device_bytes = rd;
    if (ring_host_bytes) *ring_host_bytes = rh;
    return R3DM_OK;
}

// everything registered so far is resident and laid out (the registration calls return with work still queued on the context's stream)
extern "C" int r3dm_images_wait(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    R3DM_HIP(c, hipSetDevice(c->device));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    return R3DM_OK;
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
        im->live = false; im->has_K = false; im->ann_K = 0; im->hnsw_M = 0; im->mrpt_trees = 0; im->compact_ready = false; im->n = 0; im->counts_ok = false; im->have = 0; im->stats_valid = false;
        c->spare.push_back(std::move(im));
    }
    c->imgs.clear();
    c->slot_of.clear();
    c->pending_stats.clear();
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
        // rounds of at most ~64 MiB of text (a graph of 1e8 matches is 2 GB of text: it is never held whole): a round = a run of pairs,
        // split among the T threads where the running match count passes c / T of the round's, written in order, buffers reused
        constexpr uint64_t kRoundMatches = (64ull << 20) / 22;
        std::vector<std::vector<char>> bufs((size_t)T);
        std::vector<size_t> lens((size_t)T, 0);
        std::vector<int> failed((size_t)T, 0);
        std::vector<uint64_t> cut((size_t)T + 1, 0);
        uint64_t r0 = 0;
        while (r0 < np && ok) {
            uint64_t r1 = (uint64_t)(std::upper_bound(g->offsets.begin() + (ptrdiff_t)r0, g->offsets.begin() + (ptrdiff_t)np, g->offsets[r0] + kRoundMatches) - g->offsets.begin());
            if (r1 <= r0) r1 = r0 + 1;                     // (one pair longer than a round: a round of its own)
            if (r1 > np) r1 = np;
            const uint64_t rtotal = g->offsets[r1] - g->offsets[r0];
            cut[0] = r0; cut[(size_t)T] = r1;
            for (int c = 1; c < T; ++c) {
                const uint64_t want = g->offsets[r0] + rtotal / (uint64_t)T * (uint64_t)c;
                cut[c] = (uint64_t)(std::upper_bound(g->offsets.begin() + (ptrdiff_t)r0, g->offsets.begin() + (ptrdiff_t)r1, want) - g->offsets.begin());
                if (cut[c] > r1) cut[c] = r1;
                if (cut[c] < cut[c - 1]) cut[c] = cut[c - 1];
            }
            r3dm_parallel_for((long)T, T, [&](long c) {
                lens[(size_t)c] = 0;
                try {
                    const uint64_t p0 = cut[c], p1 = cut[c + 1];
                    if (p1 <= p0) return;
                    const uint64_t mcount = g->offsets[p1] - g->offsets[p0];
                    const size_t need = (size_t)(mcount * 22 + (p1 - p0) * 44 + 64);        // a line <= 22 bytes, a pair header <= 11 + 11 + 21
                    if (bufs[(size_t)c].size() < need) bufs[(size_t)c].resize(need);
                    char* o = bufs[(size_t)c].data();
                    for (uint64_t p = p0; p < p1; ++p) {
                        o = put_u64(o, g->pairs[2 * p], ' '); o = put_u64(o, g->pairs[2 * p + 1], '\n'); o = put_u64(o, g->offsets[p + 1] - g->offsets[p], '\n');
                        for (uint64_t k = g->offsets[p]; k < g->offsets[p + 1]; ++k) { o = put_u64(o, g->matches[k].i, ' '); o = put_u64(o, g->matches[k].j, '\n'); }
                    }
                    lens[(size_t)c] = (size_t)(o - bufs[(size_t)c].data());
                } catch (...) { failed[(size_t)c] = 1; }
            });
            for (int c = 0; c < T && ok; ++c) {
                if (failed[(size_t)c]) { fclose(f); return R3DM_ERR_NOMEM; }        // (a formatting buffer could not be allocated: not an I/O failure)
                if (lens[(size_t)c]) ok &= fwrite(bufs[(size_t)c].data(), 1, lens[(size_t)c], f) == lens[(size_t)c];
            }
            r0 = r1;
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

