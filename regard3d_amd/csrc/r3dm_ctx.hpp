// r3dm_ctx.hpp -- host-side state shared by the translation units of libr3dm.so (not part of the public ABI).
//   api_core.cpp      context, views (staging), match-graph objects, matches.* files
//   api_match.cpp     exhaustive and graph-based putative matching
//   api_filter.cpp    AC-RANSAC geometric filters (F, E, H)
//   api_features.cpp  Fast-A-KAZE detection, MLDB / LIOP description, the features work item
#pragma once

#include "r3dm_internal.hpp"

#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

using namespace r3dm;

struct r3dm_index;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------

// How many host threads a helper team of this library may start: the cores this process may actually use -- the affinity mask AND the
// cgroup CPU quota (a container with `cpu.max = 1600000 100000` sees 256 processors and owns 16: a burst of more runnable threads than
// that is throttled until the end of the 100 ms period, which showed as random 60-100 ms stalls of the stage's main thread) -- divided
// among the workers that may run such a team at the same time, at most `want`.
// called first thing by the library's background writer threads (feature files, match files): their work has a whole phase of the
// caller's to hide behind, so under contention for the host's cores they stand back (Linux: a per-thread nice value, inherited
// by the OpenMP helpers they start)
// (only when the host asked for it -- r3dm_set_background_nice, R3DComputeMatches::setBackgroundThreadsNice: a library does not
// change thread priorities on its own)
inline void r3dm_background_thread(int nice_value)
{
    if (nice_value > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nice_value);
}

// fn(i) for i in [0, n) on up to `threads` host threads that EXIT when the work is done.  (An OpenMP team keeps spinning for
// work for milliseconds after every region -- libgomp sizes that spin by the CPUs it can see, not by the cgroup quota the process
// lives under -- and several such teams at once ran a 16-core quota dry: CFS then stalls the whole process for the rest of its
// period.  These regions are a few milliseconds long and few; six thread starts per region cost less than that.)
// fn must not throw.
template <class F>
inline void r3dm_parallel_for(long n, int threads, F&& fn)
{
    if (n <= 0) return;
    if (threads > n) threads = (int)n;
    if (threads <= 1) { for (long i = 0; i < n; ++i) fn(i); return; }
    std::atomic<long> next{0};
    auto worker = [&]() { for (;;) { const long i = next.fetch_add(1, std::memory_order_relaxed); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    th.reserve((size_t)threads - 1);
    try { for (int t = 1; t < threads; ++t) th.emplace_back(worker); } catch (...) {}      // fewer helpers: the work is still done
    worker();
    for (std::thread& t : th) t.join();
}

inline int r3dm_host_team(int want, int concurrent_teams = 2)
{
    static const int cores = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < n) n = a; }
        for (const char* path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
            if (FILE* f = fopen(path, "r")) {
                char a[64] = {0}, b[64] = {0};
                const int got = fscanf(f, "%63s %63s", a, b);
                fclose(f);
                if (got >= 1 && a[0] >= '0' && a[0] <= '9') {
                    double quota = atof(a), period = got >= 2 ? atof(b) : 100000.0;
                    if (got < 2) if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%63s", b) == 1) period = atof(b); fclose(g); }
                    if (quota > 0 && period > 0) { const int q = (int)(quota / period); if (q >= 1 && q < n) n = q; }
                }
                break;
            }
        }
        return n;
    }();
    int t = (cores - 4) / (concurrent_teams > 0 ? concurrent_teams : 1);       // four cores stay free for the main thread, writers and the runtime's own threads
    if (t > want) t = want;
    return t < 1 ? 1 : t;
}

// Device memory of the views of a context: blocks cut from large slabs instead of one hipMalloc per buffer.  A view of 8,192 x 128
// floats is a 4 MiB tile image + 32 KiB of slack + three small arrays: hipMalloc rounds each of them up to its page size (the tile
// image alone occupied 6 MiB), which made a registered collection 1.5 x its own bytes; cut from 64 MiB slabs at 256-byte granularity
// it is 1.05 x.  Blocks freed by a view (replaced, trimmed) go to a free list and serve later requests of at most 1.25 x their size; a
// slab whose blocks are all free goes back to the device.  Not thread-safe: a context is used by one thread at a time.
struct DevArena {
    struct Slab { unsigned char* p = nullptr; size_t cap = 0, used = 0; uint32_t live = 0; };
    std::vector<Slab> slabs;
    std::multimap<size_t, std::pair<uint32_t, size_t>> free_blocks;       // size -> (slab, offset)
    int bump[2] = {-1, -1};                                               // the slabs new blocks are cut from: the newest and the one before
    static constexpr size_t kSlab = (size_t)64 << 20, kAlign = 256;
    hipError_t alloc(size_t bytes, void** out, size_t* got)
    {
        bytes = (bytes + kAlign - 1) / kAlign * kAlign;
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + bytes / 4) {
            const auto blk = it->second;
            *out = slabs[blk.first].p + blk.second; *got = it->first;
            slabs[blk.first].live += 1;
            free_blocks.erase(it);
            return hipSuccess;
        }
        for (int b : bump) {                                                     // the two newest slabs are bumped
            if (b < 0 || (size_t)b >= slabs.size()) continue;
            Slab& sl = slabs[(size_t)b];
            if (sl.p && sl.cap - sl.used >= bytes) { *out = sl.p + sl.used; *got = bytes; sl.used += bytes; sl.live += 1; return hipSuccess; }
        }
        Slab sl;
        sl.cap = std::max(kSlab, bytes);
        hipError_t e = hipMalloc((void**)&sl.p, sl.cap);
        if (e != hipSuccess) return e;
        sl.used = bytes; sl.live = 1;
        *out = sl.p; *got = bytes;
        uint32_t at = (uint32_t)slabs.size();
        for (uint32_t k = 0; k < slabs.size(); ++k) if (!slabs[k].p) { at = k; break; }      // reuse a dead entry: indices in free_blocks stay valid
        bump[1] = bump[0]; bump[0] = (int)at;
        if (at == slabs.size()) slabs.push_back(sl); else slabs[at] = sl;
        return hipSuccess;
    }
    void free(void* p, size_t bytes)
    {
        unsigned char* q = static_cast<unsigned char*>(p);
        for (uint32_t k = 0; k < slabs.size(); ++k) {
            Slab& sl = slabs[k];
            if (!sl.p || q < sl.p || q >= sl.p + sl.cap) continue;
            sl.live -= 1;
            if (sl.live == 0) {
                for (auto it = free_blocks.begin(); it != free_blocks.end();) it = it->second.first == k ? free_blocks.erase(it) : std::next(it);
                (void)hipFree(sl.p);
                sl = Slab();
            } else free_blocks.insert({bytes, {k, (size_t)(q - sl.p)}});
            return;
        }
    }
    size_t bytes_held() const { size_t b = 0; for (const Slab& sl : slabs) b += sl.p ? sl.cap : 0; return b; }
    void release_all() { for (Slab& sl : slabs) if (sl.p) (void)hipFree(sl.p); slabs.clear(); free_blocks.clear(); bump[0] = bump[1] = -1; }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevArena* arena = nullptr;        // set: the buffer is a block of this arena (per-view layouts of a context), else its own hipMalloc
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        if (arena) {
            hipError_t e = arena->alloc(bytes, &p, &cap);
            if (e != hipSuccess) { p = nullptr; cap = 0; }
            return e;
        }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) { if (arena) arena->free(p, cap); else (void)hipFree(p); }
        p = nullptr; cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// page-locked host memory (device-to-host copies into it run at link speed and are truly asynchronous)
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Layouts of a view beyond the ones every view holds from registration on (float views: fragment-order f32 tiles + norms + statistics;
// binary views: padded word rows): staged on FIRST USE by the path that reads them (api_core.cpp: ensure_layouts), or at registration
// when the path's flag (r3dm_set_integer_mfma / _split_mfma / _hamming_mfma) is already on.
enum : uint32_t {
    kLayRows   = 1u,     // row-major f32 rows: exact re-scoring of real-valued pairs, the generic exact scan, the approximate matchers
    kLayBf16   = 2u,     // bf16 tiles (r3dm_set_integer_mfma)
    kLaySplit  = 4u,     // split-f16 planes (r3dm_set_split_mfma)
    kLayCounts = 8u,     // count tiles + scale order (r3dm_set_split_mfma, rows = integer votes x a row scale); needs kLayRows
    kLayBin8   = 16u,    // one byte per bit in i8 fragment order (r3dm_set_hamming_mfma)
};

struct HostImage {
    uint32_t view_id = 0, n = 0, dim = 0, width = 0, height = 0;
    r3dm_dtype dtype = R3DM_F32;
    uint32_t G = 0, n_tiles = 0, words = 0;
    bool has_xy = false, has_dup = false, live = false;
    bool borrowed = false;            // the buffers belong to an r3dm_index mounted into this slot for one call: never freed here
    struct r3dm_index* owner = nullptr;   // ... that index (layouts staged on first use are added to IT, under its lock)
    DevBuf rows, tiled, tiled16, tiledh, tiledc, tiledp, cscale, cquad, cperm, tiled8, norms, bin, xy, canon;
    uint32_t have = 0;                // kLay* bits: which on-demand layouts reflect the staged view
    // staging statistics: accumulated by the staging kernel in the view's table entry, read back -- for all views staged since the last
    // read -- by sync_view_stats() at the first call that needs them (no per-view synchronisation at registration)
    bool stats_valid = false;
    uint32_t stat_bits[3] = {0, 0, 1};                                      // ImgDev::max_norm_bits, max_abs_bits, not_integer
    float max_abs = 0.0f; bool not_integer = true, has_negative = true;
    int32_t split_k = 0;                                                    // scale exponent of the f16 split tiles (a function of max_abs)
    bool counts_ok = false;                                                 // every row is small integers x a row scale: the count tiles are valid
    bool compact_ready = false;     // ann_rows16 / ann_rows8 reflect the staged rows (reset by staging)
    DevBuf ann_adj, ann_deg, ann_rows16, ann_rows8;   // graph index (r3dm_match_pairs_kgraph), valid when ann_K != 0; compact row copies only for bf16- / u8-exact views
    uint32_t ann_K = 0;
    // HNSW index (r3dm_match_pairs_hnsw), valid when hnsw_M != 0: hnswlib's arrays (kernels_hnsw.hip: HnswView) + the host-side scalars
    DevBuf hnsw_l0, hnsw_up_off, hnsw_up;
    uint32_t hnsw_M = 0, hnsw_seed = 0, hnsw_up_rows = 0; int32_t hnsw_enter = -1, hnsw_maxlevel = -1;
    // MRPT index (r3dm_match_pairs_mrpt), valid when mrpt_trees != 0: kernels_mrpt.hip: MrptView
    DevBuf mrpt_R, mrpt_RT, mrpt_splits, mrpt_leaves, mrpt_lf;
    uint32_t mrpt_trees = 0, mrpt_depth = 0; float mrpt_density = 0.0f; uint64_t mrpt_seed = 0;
    bool has_K = false;               // pinhole intrinsics (r3dm_set_intrinsics), needed by the essential-matrix filter
    double Kinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the layout buffers of a view registered with a context are blocks of the context's arena (an r3dm_index owns plain allocations:
    // it outlives contexts)
    void use_arena(DevArena* a)
    {
        DevBuf* b[] = {&rows, &tiled, &tiled16, &tiledh, &tiledc, &tiledp, &cscale, &cquad, &cperm, &tiled8, &norms, &bin, &xy, &canon};
        for (DevBuf* x : b) x->arena = a;
    }
    void release()
    {
        if (borrowed) { *this = HostImage(); return; }     // drop the aliases, keep the index's memory
        rows.release(); tiled.release(); tiled16.release(); tiledh.release(); tiledc.release(); tiledp.release(); cscale.release(); cquad.release(); cperm.release(); tiled8.release(); norms.release(); bin.release(); xy.release(); canon.release();
        ann_adj.release(); ann_deg.release(); ann_rows16.release(); ann_rows8.release(); ann_K = 0; compact_ready = false; live = false; have = 0; stats_valid = false;
        hnsw_l0.release(); hnsw_up_off.release(); hnsw_up.release(); hnsw_M = 0;
        mrpt_R.release(); mrpt_RT.release(); mrpt_splits.release(); mrpt_leaves.release(); mrpt_lf.release(); mrpt_trees = 0;
    }
};

int graph_dev_append(r3dm_ctx* c, r3dm_graph* g, const std::vector<uint32_t>& pair_ids, const std::vector<uint32_t>& counts, std::vector<GraphSeg>& segs,
                     const r3dm_match* src, const uint32_t* idx);      // api_core.cpp

inline uint32_t kernel_G_for(uint32_t dim)
{
    const uint32_t g = (dim + 7) / 8;
    if (g <= 8) return 8;
    if (g <= 16) return 16;
    if (g <= 18) return 18;
    if (g <= 32) return 32;
    return g;               // no tensor kernel: exact scan only
}
inline bool has_tensor_kernel(uint32_t G) { return G == 8 || G == 16 || G == 18 || G == 32; }

inline uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline bool has_ext(const char* path, const char* ext)
{
    const size_t lp = strlen(path), le = strlen(ext);
    return lp >= le && strcmp(path + lp - le, ext) == 0;
}


// The same CSR in device memory, kept beside the host vectors when the context was asked to (r3dm_set_device_graphs): what
// r3dm_allgather_graphs puts on the wire without a host round trip of the payload (api_comm.cpp).  Filled where the data already is
// on the device: by the gather kernels behind the match finalisation and the filters (kernels_graph.hip).
struct GraphDev {
    DevBuf pairs, counts, matches;    // u32 [2 P], u32 [P], r3dm_match [M]
    uint64_t P = 0, M = 0;
    int device = -1;
    bool valid = false;
    void release() { pairs.release(); counts.release(); matches.release(); P = M = 0; valid = false; device = -1; }
};

struct r3dm_graph {
    std::vector<uint32_t> pairs;      // 2 per pair
    std::vector<uint64_t> offsets;    // n_pairs + 1
    std::vector<r3dm_match> matches;
    GraphDev dev;                     // optional device mirror (never copied with the graph)
    r3dm_graph() = default;
    r3dm_graph(const r3dm_graph&) = delete;
    r3dm_graph& operator=(const r3dm_graph&) = delete;
    ~r3dm_graph() { dev.release(); }
};

struct FilterBufs {
    DevBuf f_pairs, f_ids, f_offs, f_matches, f_inl_cnt, f_inl_idx, f_F, f_thr, f_iters, f_log10, f_logck, f_scratch, f_kinv, f_spill, f_soff, f_order, f_coop, f_coop_prof, f_la;
    // r3dm_filter_FEH: the kernel of this kind on a stream of its own PRIORITY class (E high, F normal, H low).  Streams of one
    // priority share a handful of hardware queues -- three plain streams ran the three kernels mostly one after the other --, streams
    // of different priorities never do.
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the cooperative kernel of the kind's long pairs (kernels_filter_coop.hip) runs beside the one-workgroup kernel of its short ones
    hipStream_t stream2 = nullptr;
    hipEvent_t ev2 = nullptr;
    PinBuf pin_idx;           // page-locked landing zone of the inlier indices
    void release()
    {
        pin_idx.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
        if (ev2) (void)hipEventDestroy(ev2);
        if (stream2) (void)hipStreamDestroy(stream2);
        ev0 = ev1 = ev2 = nullptr; stream = stream2 = nullptr;
        DevBuf* b[] = {&f_pairs, &f_ids, &f_offs, &f_matches, &f_inl_cnt, &f_inl_idx, &f_F, &f_thr, &f_iters, &f_log10, &f_logck, &f_scratch, &f_kinv, &f_spill, &f_soff, &f_order, &f_coop, &f_coop_prof, &f_la};
        for (DevBuf* x : b) x->release();
    }
};

// The way of a view from host memory to HBM (r3dm_set_image / r3dm_set_images): a ring of page-locked slots the caller's pageable rows
// are copied into by the host (several threads for a batch of views), one asynchronous DMA per view from there into the slot's
// device buffer, the staging kernel behind it, an event behind that -- a slot is reused when its event has passed; nothing waits
// per view.  Every view in flight also owns the slot's position-class hash table.
struct UploadRing {
    static constexpr int kSlots = 8;
    PinBuf pin[kSlots];
    DevBuf raw[kSlots];
    DevBuf ctab[kSlots];                 // position-class hash table: u64 keys [2^bits] then u32 values [2^bits]
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool busy[kSlots] = {false, false, false, false, false, false, false, false};
    uint64_t next = 0;
    // the DMAs run on a stream of their own, each followed by an event the view's staging kernel waits for on the context's stream: the
    // DMA of view k + 1 then runs beside the kernel of view k instead of behind it (one stream: 85 us + 29 us per 8,192 x 128 view)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy[kSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void release()
    {
        for (int k = 0; k < kSlots; ++k) {
            pin[k].release(); raw[k].release(); ctab[k].release();
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (ev_copy[k]) (void)hipEventDestroy(ev_copy[k]);
            ev[k] = nullptr; ev_copy[k] = nullptr; busy[k] = false;
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
    }
};

struct r3dm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::string arch;
    int n_cu = 0;
    uint64_t hbm = 0;
    std::vector<std::unique_ptr<HostImage>> imgs;           // slot -> image
    std::vector<std::unique_ptr<HostImage>> spare;          // views dropped by r3dm_clear_images: their device buffers serve the next ones
    std::unordered_map<uint32_t, uint32_t> slot_of;         // view id -> slot
    DevBuf d_imgs;                                           // ImgDev[slots]
    PinBuf tab_host;                                         // page-locked mirror of the table's entries as the host last published them
    PinBuf tab_back;                                         // landing zone of the table read back by sync_view_stats()
    std::vector<uint32_t> pending_stats;                     // slots staged since the last sync_view_stats()
    UploadRing ring;
    DevArena arena;                                          // device memory of the registered views' layouts
    DevBuf d_verdict;                                        // verdict words of count-tile checks launched by ensure_layouts
    uint64_t n_ring_uploads = 0, n_direct_uploads = 0;       // views that went through the ring / were read where the caller had them
    // scratch (grown on demand, reused across calls)
    DevBuf d_pairs, d_nn, d_knn_idx, d_knn_dist, d_fb, d_cnt, d_out, d_pair_off, d_pair_cnt, d_raw;
    // geometric filters: one set of work buffers per model kind (0 F, 1 H, 2 E), so that r3dm_filter_FEH can run the three
    // AC-RANSAC kernels of a putative graph side by side (a collection with few, long pairs leaves most CUs idle under one)
    FilterBufs fb[3];
    // the cooperative AC-RANSAC kernel of a call (long pairs of all its filters, one pool of workers): scheduling words + start order +
    // device copy of the kinds' parameters; its stream and the event recorded behind it
    DevBuf coop_sched;
    hipStream_t coop_stream = nullptr;
    hipEvent_t coop_ev = nullptr;
    DevBuf liop_pix, liop_sx, liop_sy, liop_in, liop_out, liop_cnt, liop_img, liop_M, liop_kern;
    DevBuf h_aux, h_jobs;            // HNSW: per-batch layer tables / job records
    DevBuf a_jobs, a_scratch, a_ids, d_spill, d_fb2;
    DevBuf m_raw, m_peer;                                   // r3dm_multi_set_image: the one upload of a view / this device's copy of it
    std::vector<DevBuf> ak_bufs;                            // Fast-A-KAZE work buffers of the last image size, ak_B planes each
    int ak_w = 0, ak_h = 0, ak_B = 0;
    uint32_t ak_cap = 0;                                    // candidate slots per image the detector last needed (grows, never shrinks)
    int ak_n_levels = 0;
    AkLevelDev* ak_levels_dev = nullptr;                    // level table of the last detector pass (inside ak_bufs; read by the MLDB kernel)
    PinBuf pin_small;                                       // ... of the counters and per-pair tables that come back with them
    PinBuf pin_out;                                         // page-locked landing zone of the match lists of a batch (finalize_batch)
    PinBuf pin_desc;                                        // page-locked landing zone of the LIOP descriptors of a batch
    bool integer_mfma = false;                              // r3dm_set_integer_mfma
    bool split_mfma = false;                                // r3dm_set_split_mfma
    bool hamming_mfma = false;                              // r3dm_set_hamming_mfma
    bool device_graphs = false;                             // r3dm_set_device_graphs: match / filter results keep a device mirror (GraphDev)
    DevBuf g_segs;                                          // segment table of the graph gather kernel (kernels_graph.hip)
    uint32_t liop_npix = 0;
    uint64_t n_views_staged = 0;                            // copies + re-layouts since r3dm_create (never reset)
    r3dm_stats stats{};
    r3dm_features_sink feat_sink = nullptr; void* feat_sink_user = nullptr;   // r3dm_set_features_sink
    const uint32_t* feat_sink_ids = nullptr;                 // indices of the running batch in its caller's arrays (r3dm_multi_extract_features*), else 0 .. B-1
    r3dm_features_totals feat_totals{};                      // since r3dm_create (r3dm_get_features_totals)
    // r3dm_set_deferred_feature_files: the fwrite of a batch's .feat / .desc runs on `file_writer` behind the sink calls; the thread
    // reads pin_desc, so it is joined before that buffer is filled again, by r3dm_features_files_wait and by r3dm_destroy
    bool defer_files = false;
    int background_nice = 0;                                  // r3dm_set_background_nice: nice value of this context's background writer threads (0 = unchanged)
    hipEvent_t ev_desc = nullptr;                            // the descriptors of the batch have landed in pin_desc (created on first use)
    std::thread file_writer;
    int file_writer_rc = 0; std::string file_writer_err;
    double file_writer_ms = 0.0;                             // wall time the writer threads spent (sum since the last wait)
    std::vector<r3dm_pair_report> report;                    // last r3dm_filter_F call, one per putative pair
};

#define R3DM_HIP(ctx, call)                                                            \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);          \
            return R3DM_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)



// The C ABI never throws: entry points whose bodies size host containers from caller- or file-provided counts run behind this
// guard (std::bad_alloc / std::length_error would otherwise cross the extern "C" boundary and terminate the host application).
template <class F>
static inline int r3dm_guarded(r3dm_ctx* c, F&& body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc&) { if (c) { try { c->err = "out of host memory"; } catch (...) {} } return R3DM_ERR_NOMEM; }
    catch (const std::exception& e) { if (c) { try { c->err = e.what(); } catch (...) {} } return R3DM_ERR_INVALID; }
    catch (...) { return R3DM_ERR_INVALID; }
}

struct PairJob { uint32_t I, J, sI, sJ; };

// A dataset staged once for many queries (ArrayMatcher::Build): owns its device buffers, belongs to a device, not to a context
struct r3dm_index {
    int device = 0;
    HostImage img;                              // (statistics included: HostImage::stat_bits)
    std::mutex mu;                              // layouts staged after Build (a flag switched on later) are added under this lock
};

// shared between the translation units
// part graphs of the approximate matchers (graph-searched pairs + exhaustively scanned small pairs): see their use
struct PartMirrorGuard {
    r3dm_ctx* c; bool keep;
    PartMirrorGuard(r3dm_ctx* c_, bool suppress);
    ~PartMirrorGuard();
};
int merge_parts_keep_mirror(r3dm_graph& ga, r3dm_graph& gs, r3dm_graph** out);
// (re)writes the table entry of `slot` from its HostImage, statistics included: only for views whose statistics the host holds
int publish_entry(r3dm_ctx* c, uint32_t slot);
// the statistics of every view staged since the last call -> HostImage (one read of the table, one synchronisation)
int sync_view_stats(r3dm_ctx* c);
// stages the layouts `want` (kLay* bits) of every listed slot that does not hold them; kLayCounts: *counts_all_ok = every listed view
// passed the votes-x-scale check.  Slots that mount an r3dm_index are staged under the index's lock.
int ensure_layouts(r3dm_ctx* c, std::vector<uint32_t> slots, uint32_t want, bool* counts_all_ok = nullptr);
int ensure_layouts_image(r3dm_ctx* c, HostImage& h, uint32_t want);        // (no table entry involved; the caller publishes)
int run_match_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio_R, r3dm_graph* g,
                    int32_t* knn_idx_host, float* knn_dist_host);
int finalize_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, uint32_t q_stride, uint32_t sort_cap,
                   uint64_t n_queries, uint32_t max_nJ, r3dm_graph* g, int32_t* knn_idx_host, float* knn_dist_host);
int ensure_ann_indices(r3dm_ctx* c, std::vector<uint32_t> slots, uint32_t K);
int stage_into_slot(r3dm_ctx* c, uint32_t slot, uint32_t view_id, uint32_t width, uint32_t height,
                    const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy);
