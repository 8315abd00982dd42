// r3dm_api.cpp -- host side of libr3dm.so: the C ABI declared in include/r3dm.h.
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 (load regions, exhaustive pairs, match,
// save putative, AC-RANSAC F filter, save) -- with every arithmetic stage running as HIP kernels
// on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_internal.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

using namespace r3dm;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostImage {
    uint32_t view_id = 0, n = 0, dim = 0, width = 0, height = 0;
    r3dm_dtype dtype = R3DM_F32;
    uint32_t G = 0, n_tiles = 0, words = 0;
    bool has_xy = false, has_dup = false, live = false;
    DevBuf rows, tiled, norms, bin, xy, canon;
    DevBuf ann_adj, ann_deg;          // graph index (r3dm_match_pairs_kgraph), valid when ann_K != 0
    uint32_t ann_K = 0;
    bool has_K = false;               // pinhole intrinsics (r3dm_set_intrinsics), needed by the essential-matrix filter
    double Kinv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    void release()
    {
        rows.release(); tiled.release(); norms.release(); bin.release(); xy.release(); canon.release();
        ann_adj.release(); ann_deg.release(); ann_K = 0; live = false;
    }
};

uint32_t kernel_G_for(uint32_t dim)
{
    const uint32_t g = (dim + 7) / 8;
    if (g <= 8) return 8;
    if (g <= 16) return 16;
    if (g <= 18) return 18;
    if (g <= 32) return 32;
    return g;               // no tensor kernel: exact scan only
}
bool has_tensor_kernel(uint32_t G) { return G == 8 || G == 16 || G == 18 || G == 32; }

uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool has_ext(const char* path, const char* ext)
{
    const size_t lp = strlen(path), le = strlen(ext);
    return lp >= le && strcmp(path + lp - le, ext) == 0;
}

}  // namespace

struct r3dm_graph {
    std::vector<uint32_t> pairs;      // 2 per pair
    std::vector<uint64_t> offsets;    // n_pairs + 1
    std::vector<r3dm_match> matches;
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
    std::unordered_map<uint32_t, uint32_t> slot_of;         // view id -> slot
    DevBuf d_imgs;                                           // ImgDev[slots]
    // scratch (grown on demand, reused across calls)
    DevBuf d_pairs, d_nn, d_knn_idx, d_knn_dist, d_fb, d_cnt, d_out, d_pair_off, d_pair_cnt, d_raw;
    DevBuf f_pairs, f_ids, f_offs, f_matches, f_inl_cnt, f_inl_idx, f_F, f_thr, f_iters, f_log10, f_logck, f_scratch;
    DevBuf liop_pix, liop_sx, liop_sy, liop_in, liop_out, liop_cnt, liop_img, liop_M, liop_kern;
    DevBuf a_jobs, a_scratch, a_ids, f_kinv, d_spill, f_spill;
    std::vector<DevBuf> ak_bufs;                            // Fast-A-KAZE work buffers of the last image size
    int ak_w = 0, ak_h = 0;
    uint32_t liop_npix = 0;
    r3dm_stats stats{};
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
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& im : c->imgs) if (im) im->release();
    DevBuf* bufs[] = {&c->d_imgs, &c->d_pairs, &c->d_nn, &c->d_knn_idx, &c->d_knn_dist, &c->d_fb, &c->d_cnt, &c->d_out,
                      &c->d_pair_off, &c->d_pair_cnt, &c->d_raw, &c->f_pairs, &c->f_ids, &c->f_offs, &c->f_matches,
                      &c->f_inl_cnt, &c->f_inl_idx, &c->f_F, &c->f_thr, &c->f_iters, &c->f_log10, &c->f_logck, &c->f_scratch,
                      &c->liop_pix, &c->liop_sx, &c->liop_sy, &c->liop_in, &c->liop_out, &c->liop_cnt, &c->liop_img, &c->liop_M, &c->liop_kern,
                      &c->a_jobs, &c->a_scratch, &c->a_ids, &c->f_kinv, &c->d_spill, &c->f_spill};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : c->ak_bufs) b.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
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
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// views
// ------------------------------------------------------------------------------------------------
static int upload_imgdev(r3dm_ctx* c, uint32_t slot)
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
    d.width = h.width; d.height = h.height; d.max_norm_bits = 0; d.max_abs_bits = 0; d.not_integer = 0;
    d.ann_adj = nullptr; d.ann_deg = nullptr;          // staging invalidates the graph index
    R3DM_HIP(c, hipMemcpyAsync(c->d_imgs.as<ImgDev>() + slot, &d, sizeof(ImgDev), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    return R3DM_OK;
}

// copy + re-layout one view into slot `slot`
static int stage_into_slot(r3dm_ctx* c, uint32_t slot, uint32_t view_id, uint32_t width, uint32_t height,
                           const void* desc, uint32_t n, uint32_t dim, r3dm_dtype dtype, const float* xy)
{
    HostImage& h = *c->imgs[slot];
    h.view_id = view_id; h.n = n; h.dim = dim; h.dtype = dtype; h.width = width; h.height = height;
    h.has_xy = (xy != nullptr); h.has_dup = false; h.live = true;
    h.G = 0; h.n_tiles = 0; h.words = 0; h.ann_K = 0;
    if (dtype == R3DM_BIN) {
        h.words = (dim + 3) / 4;
        const uint32_t n_pad = n + 8;
        R3DM_HIP(c, h.bin.ensure((size_t)n_pad * h.words * 4 + kSlackBytes));
        if (n) {
            R3DM_HIP(c, c->d_raw.ensure((size_t)n * dim));
            R3DM_HIP(c, hipMemcpyAsync(c->d_raw.p, desc, (size_t)n * dim, hipMemcpyDefault, c->stream));
        }
        R3DM_HIP(c, launch_stage_bin(c->stream, c->d_raw.as<uint8_t>(), n, dim, h.bin.as<uint32_t>(), h.words, n_pad));
    } else {
        h.G = kernel_G_for(dim);
        h.n_tiles = (n + kTileRows - 1) / kTileRows;
        const size_t tiled_bytes = (size_t)h.n_tiles * h.G * 1024 + kSlackBytes;
        const size_t norm_bytes = (size_t)h.n_tiles * 32 * 4 + kSlackBytes;
        R3DM_HIP(c, h.rows.ensure((size_t)std::max<uint32_t>(n, 1) * dim * 4 + 256));
        R3DM_HIP(c, h.tiled.ensure(tiled_bytes));
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
    // position classes (IndMatchDecorator de-duplication needs to know which features share a position)
    if (xy && n > 1) {
        std::vector<float> hxy((size_t)n * 2);
        R3DM_HIP(c, hipMemcpyAsync(hxy.data(), h.xy.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        std::vector<uint32_t> ord(n);
        std::iota(ord.begin(), ord.end(), 0u);
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
            if (hxy[2 * a] != hxy[2 * b]) return hxy[2 * a] < hxy[2 * b];
            if (hxy[2 * a + 1] != hxy[2 * b + 1]) return hxy[2 * a + 1] < hxy[2 * b + 1];
            return a < b;
        });
        std::vector<uint32_t> canon(n);
        bool dup = false;
        for (uint32_t k = 0; k < n;) {
            uint32_t e = k + 1;
            while (e < n && hxy[2 * ord[e]] == hxy[2 * ord[k]] && hxy[2 * ord[e] + 1] == hxy[2 * ord[k] + 1]) ++e;
            for (uint32_t q = k; q < e; ++q) canon[ord[q]] = ord[k];     // ord[k] is the smallest index of the group
            if (e - k > 1) dup = true;
            k = e;
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
                                     h.tiled.as<float>(), h.norms.as<float>(), h.G, h.n_tiles, mx));
    }
    R3DM_HIP(c, hipStreamSynchronize(c->stream));     // d_raw is reused by the next call
    return R3DM_OK;
}

extern "C" int r3dm_set_image(r3dm_ctx* c, uint32_t view_id, uint32_t width, uint32_t height,
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
        c->imgs.emplace_back(new HostImage());
        c->slot_of[view_id] = slot;
    } else slot = it->second;
    return stage_into_slot(c, slot, view_id, width, height, desc, n, dim, dtype, xy);
}

extern "C" int r3dm_clear_images(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& im : c->imgs) if (im) im->release();
    c->imgs.clear();
    c->slot_of.clear();
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// putative matching
// ------------------------------------------------------------------------------------------------
struct PairJob { uint32_t I, J, sI, sJ; };
extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out);

// compaction + ordering + de-duplication of nn_idx[pair][*] (finalize_pairs_kernel), copy back, append the non-empty
// pairs to `g` in job order.  Shared by the exhaustive and the graph-search drivers.
static int finalize_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, uint32_t q_stride, uint32_t sort_cap,
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
        R3DM_HIP(c, hipMemcpyAsync(&total, fp.total, 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(h_off.data(), fp.pair_off, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(h_cnt.data(), fp.pair_cnt, (size_t)P * 4, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        if (total <= out_cap) break;
        out_cap = total;                                   // overflow: nothing was lost, run it again with room
    }
    std::vector<r3dm_match> h_m((size_t)total);
    if (total) R3DM_HIP(c, hipMemcpy(h_m.data(), c->d_out.p, (size_t)total * sizeof(r3dm_match), hipMemcpyDeviceToHost));
    if (knn_idx_host) {
        // single-pair use (r3dm_knn2): copy the raw 2-NN of pair 0
        R3DM_HIP(c, hipMemcpy(knn_idx_host, c->d_knn_idx.p, (size_t)max_nJ * 8, hipMemcpyDeviceToHost));
        R3DM_HIP(c, hipMemcpy(knn_dist_host, c->d_knn_dist.p, (size_t)max_nJ * 8, hipMemcpyDeviceToHost));
    }

    if (g) {
        for (uint32_t p = 0; p < P; ++p) {
            if (h_cnt[p] == 0) continue;                   // empty vectors never enter the map
            g->pairs.push_back(jobs[p].I); g->pairs.push_back(jobs[p].J);
            g->matches.insert(g->matches.end(), h_m.begin() + h_off[p], h_m.begin() + h_off[p] + h_cnt[p]);
            g->offsets.push_back(g->matches.size());
        }
    }
    return R3DM_OK;
}

// runs the 2-NN + ratio kernels over `jobs` (slot pairs, all of one dtype/dim) and appends the
// non-empty results to `g` in job order.  knn_idx/knn_dist (host, optional) receive the raw 2-NN.
static int run_match_batch(r3dm_ctx* c, const std::vector<PairJob>& jobs, float ratio_R, r3dm_graph* g,
                           int32_t* knn_idx_host, float* knn_dist_host)
{
    const uint32_t P = (uint32_t)jobs.size();
    if (P == 0) return R3DM_OK;
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
    mp.nn_idx = c->d_nn.as<uint32_t>();
    mp.knn_idx = knn_idx_host ? c->d_knn_idx.as<int32_t>() : nullptr;
    mp.knn_dist = knn_idx_host ? c->d_knn_dist.as<float>() : nullptr;
    mp.fb_total = c->d_fb.as<uint32_t>();
    mp.fb_cnt = c->d_fb.as<uint32_t>() + 16;
    mp.fb_q = c->d_fb.as<uint32_t>() + 16 + P;

    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    uint64_t n_fallback = 0;
    if (dtype == R3DM_BIN) {
        R3DM_HIP(c, launch_hamming_knn2(c->stream, mp, first.words, max_nJ));
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    } else if (has_tensor_kernel(first.G)) {
        R3DM_HIP(c, launch_l2_knn2(c->stream, mp, first.G, max_tiles));
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
        uint32_t fbt[2] = {0, 0};
        R3DM_HIP(c, hipMemcpyAsync(fbt, mp.fb_total, 8, hipMemcpyDeviceToHost, c->stream));
        R3DM_HIP(c, hipStreamSynchronize(c->stream));
        n_fallback = fbt[0];
        if (fbt[0] > 0) {
            bool rescan = fbt[1] > 0;                      // some pair overflowed its list
            if ((first.dim & 3u) == 0) R3DM_HIP(c, launch_l2_exact_batch(c->stream, mp, first.G));
            else rescan = true;                            // scalar-tail dims: generic exact kernel
            if (rescan) {
                if (total_slots > 0xFFFFFFFFull) { c->err = "batch too large for the exact rescan"; return R3DM_ERR_UNSUPPORTED; }
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

extern "C" int r3dm_match_pairs(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs,
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
            if (end > start && s * 4 > (3ull << 30)) break;       // <= 3 GiB of nn_idx per batch
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

extern "C" int r3dm_knn2(r3dm_ctx* c, const void* dataset, uint32_t n_dataset, const void* query, uint32_t n_query,
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
        c->stats = keep;
    }
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
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

static int check_kgraph_params(r3dm_ctx* c, const r3dm_kgraph_params* kp)
{
    if (!kp) return R3DM_ERR_INVALID;
    if (kp->index_K < 1 || kp->index_K > kAnnMaxK || kp->search_P < 2 || kp->search_P > 61 || kp->search_S < 1 || kp->search_S > 16) {
        c->err = "kgraph parameters out of range (index_K 1..32, search_P 2..61, search_S 1..16)";
        return R3DM_ERR_INVALID;
    }
    return R3DM_OK;
}

// builds the graph index of every listed slot that does not hold one for this K
static int ensure_ann_indices(r3dm_ctx* c, std::vector<uint32_t> slots, uint32_t K)
{
    std::sort(slots.begin(), slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    std::vector<uint32_t> todo;
    for (uint32_t s : slots) if (c->imgs[s]->ann_K != K) todo.push_back(s);
    if (todo.empty()) return R3DM_OK;
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
            jobs.push_back(j);
        }
        R3DM_HIP(c, c->a_jobs.ensure(jobs.size() * sizeof(AnnBuildJob)));
        R3DM_HIP(c, hipMemcpyAsync(c->a_jobs.p, jobs.data(), jobs.size() * sizeof(AnnBuildJob), hipMemcpyHostToDevice, c->stream));
        AnnBuildParams bp{};
        bp.imgs = c->d_imgs.as<ImgDev>(); bp.jobs = c->a_jobs.as<AnnBuildJob>(); bp.K = K;
        hipError_t e = launch_ann_build(c->stream, bp, (uint32_t)jobs.size(), max_n, dim);
        if (e == hipErrorInvalidValue) { c->err = "no graph-index kernel for this descriptor length (dim % 4 != 0 or too long)"; return R3DM_ERR_UNSUPPORTED; }
        R3DM_HIP(c, e);
        std::vector<const void*> ptrs(2 * (end - start));      // must outlive the asynchronous copies
        for (size_t k = start; k < end; ++k) {
            HostImage& h = *c->imgs[todo[k]];
            ptrs[2 * (k - start)] = h.ann_adj.p; ptrs[2 * (k - start) + 1] = h.ann_deg.p;
            R3DM_HIP(c, hipMemcpyAsync((void*)&(c->d_imgs.as<ImgDev>() + todo[k])->ann_adj, &ptrs[2 * (k - start)], 2 * sizeof(void*),
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
    hipError_t e = launch_ann_search(c->stream, sp, max_nJ, max_nI, dim);
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
    c->stats.n_pairs += P;
    c->stats.n_queries += n_queries;
    return R3DM_OK;
}

extern "C" int r3dm_match_pairs_kgraph(r3dm_ctx* c, const uint32_t* pairs_ij, uint64_t n_pairs, float dist_ratio,
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
    if (!ann_jobs.empty()) {
        std::vector<uint32_t> slots;
        for (const PairJob& j : ann_jobs) slots.push_back(j.sI);
        rc = ensure_ann_indices(c, slots, kp->index_K);
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
            if (end > start && (s * 4 > (3ull << 30) || s / 4 > 0x40000000ull)) break;
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
    const r3dm_graph* parts[2] = {&ga, &gs};
    rc = r3dm_graph_merge(parts, 2, out);
    c->stats.ms_wall_match = now_ms() - t_call;
    return rc;
}

extern "C" int r3dm_kgraph_knn2(r3dm_ctx* c, const float* dataset, uint32_t n_dataset, const float* query, uint32_t n_query,
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
            rc = ensure_ann_indices(c, {s0}, kp->index_K);
            if (rc == R3DM_OK) rc = run_ann_batch(c, jobs, 1.0f, *kp, nullptr, out_idx, out_dist);
        }
    }
    c->stats = keep;
    (void)hipStreamSynchronize(c->stream);
    c->imgs[s0]->release(); c->imgs[s0 + 1]->release();
    c->imgs.pop_back(); c->imgs.pop_back();
    return rc;
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
    int rc = ensure_ann_indices(c, {it->second}, index_K);
    if (rc != R3DM_OK) return rc;
    R3DM_HIP(c, hipMemcpy(adj_out, h.ann_adj.p, (size_t)h.n * kAnnDeg * 4, hipMemcpyDeviceToHost));
    R3DM_HIP(c, hipMemcpy(deg_out, h.ann_deg.p, (size_t)h.n * 4, hipMemcpyDeviceToHost));
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// geometric filter
// ------------------------------------------------------------------------------------------------
// model_kind 0 = fundamental matrix (GeometricFilter_FMatrix_AC), 1 = homography (GeometricFilter_HMatrix_AC)
//            2 = essential matrix (GeometricFilter_EMatrix_AC) + Regard3D's overlap rule (min_count / min_ratio)
static int filter_common(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                         uint64_t seed, r3dm_ferror err_kind, int model_kind, r3dm_graph** out, double* F_out,
                         uint32_t min_count = 0, float min_ratio = 0.f)
{
    if (!c || !putative || !out || max_iter == 0) return R3DM_ERR_INVALID;
    const uint32_t SS = model_kind == 0 ? 7u : (model_kind == 1 ? 4u : 5u);          // Kernel::MINIMUM_SAMPLES
    *out = nullptr;
    R3DM_HIP(c, hipSetDevice(c->device));
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
        if (a == c->slot_of.end() || b == c->slot_of.end()) { c->err = "filter: pair references an unregistered view"; return R3DM_ERR_INVALID; }
        const HostImage& A = *c->imgs[a->second];
        const HostImage& B = *c->imgs[b->second];
        if (!A.has_xy || !B.has_xy) { c->err = "filter: view registered without feature positions"; return R3DM_ERR_INVALID; }
        if (m > (1u << 22)) { c->err = "filter: more than 4M putative matches in one pair"; return R3DM_ERR_UNSUPPORTED; }
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
    c->stats.ms_filter_kernels = 0;
    if (NI == 0) { *out = g.release(); return R3DM_OK; }
    (void)sum_m;
    // [begin, end) of every item's putative list inside the full match array
    std::vector<uint64_t> begin_end(2 * (size_t)NI);
    for (uint32_t k = 0; k < NI; ++k) {
        const uint32_t p = item_pair[k];
        begin_end[2 * k] = putative->offsets[p];
        begin_end[2 * k + 1] = putative->offsets[p + 1];
    }
    if (model_kind == 0 && err_kind != R3DM_ERR_SYMMETRIC_EPIPOLAR) { c->err = "filter: only the symmetric epipolar error is implemented"; return R3DM_ERR_UNSUPPORTED; }
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
    R3DM_HIP(c, c->f_pairs.ensure(sizeof(uint2) * NI));
    R3DM_HIP(c, c->f_ids.ensure(sizeof(uint2) * NI));
    R3DM_HIP(c, c->f_offs.ensure(sizeof(uint64_t) * 2 * NI));
    R3DM_HIP(c, c->f_matches.ensure(sizeof(r3dm_match) * std::max<uint64_t>(n_match_total, 1)));
    R3DM_HIP(c, c->f_inl_cnt.ensure(4 * (size_t)NI));
    R3DM_HIP(c, c->f_inl_idx.ensure(4 * (size_t)n_match_total + 64));
    R3DM_HIP(c, c->f_F.ensure(72 * (size_t)NI));
    R3DM_HIP(c, c->f_thr.ensure(16 * (size_t)NI));
    R3DM_HIP(c, c->f_iters.ensure(8 * (size_t)NI));
    R3DM_HIP(c, c->f_log10.ensure(4 * l10.size()));
    R3DM_HIP(c, c->f_logck.ensure(4 * lck.size()));
    R3DM_HIP(c, hipMemcpyAsync(c->f_pairs.p, slots.data(), sizeof(uint2) * NI, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->f_ids.p, ids.data(), sizeof(uint2) * NI, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->f_offs.p, begin_end.data(), sizeof(uint64_t) * 2 * NI, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->f_matches.p, putative->matches.data(), sizeof(r3dm_match) * n_match_total, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->f_log10.p, l10.data(), 4 * l10.size(), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->f_logck.p, lck.data(), 4 * lck.size(), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->f_inl_cnt.p, 0, 4 * (size_t)NI, c->stream));

    FilterParams fp{};
    fp.imgs = c->d_imgs.as<ImgDev>();
    fp.pairs = c->f_pairs.as<uint2>(); fp.pair_ids = c->f_ids.as<uint2>();
    fp.offsets = c->f_offs.as<uint64_t>(); fp.matches = c->f_matches.as<r3dm_match>();
    // LDS sort capacity: 8192 (x 12 B) fits beside the hypothesis buffer; pairs with more putatives sort in global scratch
    fp.n_items = NI; fp.m_cap = std::min<uint32_t>(8192, std::max<uint32_t>(64, next_pow2(max_m)));
    fp.spill_keys = nullptr; fp.spill_idx = nullptr; fp.spill_off = nullptr;
    if (max_m > fp.m_cap) {
        std::vector<uint64_t> soff(NI, 0);
        uint64_t tot = 0;
        for (uint32_t k = 0; k < NI; ++k) {
            const uint64_t mk = begin_end[2 * k + 1] - begin_end[2 * k];
            soff[k] = tot;
            if (mk > fp.m_cap) tot += next_pow2((uint32_t)mk);
        }
        R3DM_HIP(c, c->f_spill.ensure(tot * 12 + NI * 8 + 64));
        unsigned char* base = c->f_spill.as<unsigned char>();
        R3DM_HIP(c, hipMemcpy(base + tot * 12, soff.data(), NI * 8, hipMemcpyHostToDevice));
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
        R3DM_HIP(c, c->f_kinv.ensure(kinv.size() * 8));
        R3DM_HIP(c, hipMemcpy(c->f_kinv.p, kinv.data(), kinv.size() * 8, hipMemcpyHostToDevice));
        fp.kinv = c->f_kinv.as<double>();
    }
    fp.log10_tab = c->f_log10.as<float>(); fp.logc_k = c->f_logck.as<float>();
    fp.inl_count = c->f_inl_cnt.as<uint32_t>(); fp.inl_idx = c->f_inl_idx.as<uint32_t>();
    fp.F_out = c->f_F.as<double>(); fp.thr_nfa = c->f_thr.as<double>(); fp.iters = c->f_iters.as<uint32_t>();
    R3DM_HIP(c, c->f_scratch.ensure(32 * (size_t)n_match_total + 4 * (size_t)n_match_total + 4 * ((size_t)n_match_total + NI + 1) + 256));
    fp.pts_scratch = c->f_scratch.as<double>();
    fp.pool_scratch = reinterpret_cast<uint32_t*>(c->f_scratch.as<unsigned char>() + 32 * (size_t)n_match_total);
    fp.scratch_logc = reinterpret_cast<float*>(c->f_scratch.as<unsigned char>() + 36 * (size_t)n_match_total);
    if (filter_F_lds_bytes(fp.m_cap, model_kind) > 160 * 1024) { c->err = "filter: LDS budget exceeded"; return R3DM_ERR_UNSUPPORTED; }
    // debug aid: R3DM_TRACE_PAIR="I,J" + R3DM_TRACE_FILE=path dump the per-model trace of one pair
    DevBuf trace_buf;
    const uint32_t trace_cap = 16384;
    const char* tp = getenv("R3DM_TRACE_PAIR");
    const char* tf = getenv("R3DM_TRACE_FILE");
    fp.trace = nullptr; fp.trace_item = 0xFFFFFFFFu; fp.trace_cap = trace_cap; fp.trace_rows = nullptr;
    fp.trace_iter = getenv("R3DM_TRACE_ITER") ? (uint32_t)atoi(getenv("R3DM_TRACE_ITER")) : 0xFFFFFFFFu;
    if (tp && tf) {
        unsigned tI = 0, tJ = 0;
        if (sscanf(tp, "%u,%u", &tI, &tJ) == 2)
            for (uint32_t k = 0; k < NI; ++k)
                if (ids[k].x == tI && ids[k].y == tJ) fp.trace_item = k;
        if (fp.trace_item != 0xFFFFFFFFu) {
            R3DM_HIP(c, trace_buf.ensure(40 * (size_t)trace_cap + 64));
            R3DM_HIP(c, hipMemsetAsync(trace_buf.p, 0, 40 * (size_t)trace_cap + 64, c->stream));
            fp.trace = trace_buf.as<double>() + 8;
            fp.trace_rows = trace_buf.as<uint32_t>();
        }
    }
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, launch_filter_F(c->stream, fp));
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));

    std::vector<uint32_t> h_cnt(NI);
    std::vector<uint32_t> h_idx(n_match_total);
    std::vector<double> h_F(9 * (size_t)NI);
    R3DM_HIP(c, hipMemcpyAsync(h_cnt.data(), c->f_inl_cnt.p, 4 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(h_idx.data(), c->f_inl_idx.p, 4 * (size_t)n_match_total, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(h_F.data(), c->f_F.p, 72 * (size_t)NI, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_filter_kernels = ms;
    if (fp.trace) {
        std::vector<double> tr(5 * (size_t)trace_cap + 8);
        R3DM_HIP(c, hipMemcpy(tr.data(), trace_buf.p, tr.size() * 8, hipMemcpyDeviceToHost));
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
        R3DM_HIP(c, hipMemcpy(h_thr.data(), c->f_thr.p, 16 * (size_t)NI, hipMemcpyDeviceToHost));
        R3DM_HIP(c, hipMemcpy(h_it.data(), c->f_iters.p, 8 * (size_t)NI, hipMemcpyDeviceToHost));
        c->report.assign(NP, r3dm_pair_report{});
        for (uint32_t k = 0; k < NI; ++k) {
            r3dm_pair_report& r = c->report[item_pair[k]];
            r.threshold_px = h_thr[2 * k]; r.nfa = h_thr[2 * k + 1];
            r.iterations = h_it[2 * k]; r.models = h_it[2 * k + 1]; r.inliers = h_cnt[k];
        }
    }

    uint64_t kept = 0;
    for (uint32_t k = 0; k < NI; ++k) {
        // GeometricFilter_{F,H}Matrix_AC: accept iff #inliers > 2.5 * MINIMUM_SAMPLES
        if ((double)h_cnt[k] <= 2.5 * SS) continue;
        const uint32_t p = item_pair[k];
        const uint64_t base = putative->offsets[p];
        // the reference's extra check after the E filter (src/R3DComputeMatches.cpp:2175-2192): pairs with poor overlap go
        if (model_kind == 2 && (h_cnt[k] < min_count ||
                                (float)h_cnt[k] / (float)(putative->offsets[p + 1] - base) < min_ratio)) continue;
        g->pairs.push_back(putative->pairs[2 * p]); g->pairs.push_back(putative->pairs[2 * p + 1]);
        for (uint32_t q = 0; q < h_cnt[k]; ++q) g->matches.push_back(putative->matches[base + h_idx[base + q]]);
        g->offsets.push_back(g->matches.size());
        if (F_out) memcpy(F_out + 9 * kept, h_F.data() + 9 * (size_t)k, 72);
        ++kept;
    }
    c->stats.ms_wall_filter = now_ms() - t_call;
    *out = g.release();
    return R3DM_OK;
}

extern "C" int r3dm_filter_F(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_ferror err_kind, r3dm_graph** out, double* F_out)
{
    return filter_common(c, putative, max_residual_px, max_iter, seed, err_kind, 0, out, F_out);
}

extern "C" int r3dm_filter_H(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, r3dm_graph** out, double* H_out)
{
    return filter_common(c, putative, max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, 1, out, H_out);
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

extern "C" int r3dm_filter_E(r3dm_ctx* c, const r3dm_graph* putative, double max_residual_px, uint32_t max_iter,
                             uint64_t seed, uint32_t min_count, float min_ratio, r3dm_graph** out, double* E_out)
{
    return filter_common(c, putative, max_residual_px, max_iter, seed, R3DM_ERR_SYMMETRIC_EPIPOLAR, 2, out, E_out, min_count, min_ratio);
}

extern "C" int r3dm_filter_report(const r3dm_ctx* c, r3dm_pair_report* out, uint64_t cap)
{
    if (!c || (cap && !out)) return R3DM_ERR_INVALID;
    const uint64_t n = std::min<uint64_t>(cap, c->report.size());
    if (n) memcpy(out, c->report.data(), n * sizeof(r3dm_pair_report));
    return (int)std::min<uint64_t>(c->report.size(), 0x7FFFFFFF);
}

// ------------------------------------------------------------------------------------------------
// keypoint detection: Fast-A-KAZE (kernels_akaze.hip)
// ------------------------------------------------------------------------------------------------
namespace {

struct AkLevelHost {
    int w, h, octave, sublevel, sigma_size, border;
    float esigma, etime, ratio;
};

// AKAZEFeaturesV2::Allocate_Memory_Evolution (src/thirdparty/fast-akaze/AKAZEFeatures.cpp:73-131) with the AKAZE2::create()
// defaults (AKAZEConfig.h:18-43): 4 octaves x 4 sublevels, soffset 1.6, derivative_factor 1.5, MLDB border 10 sqrt(2) sigma
std::vector<AkLevelHost> ak_levels(int w, int h)
{
    std::vector<AkLevelHost> lv;
    const int omax = 4, nsub = 4;
    const float soffset = 1.6f, dfac = 1.5f;
    const float smax = 10.0f * sqrtf(2.0f);
    int lh = h, lw = w, power = 1;
    for (int i = 0; i < omax; ++i) {
        for (int j = 0; j < nsub; ++j) {
            AkLevelHost e{};
            e.w = lw; e.h = lh;
            e.esigma = soffset * powf(2.f, (float)j / nsub + i);
            e.sigma_size = (int)(e.esigma * dfac / power + 0.5f);
            e.border = (int)(smax * e.sigma_size + 0.5f) + 1;
            e.etime = 0.5f * (e.esigma * e.esigma);
            e.octave = i; e.sublevel = j; e.ratio = (float)power;
            if (e.border * 2 + 1 >= lw || e.border * 2 + 1 >= lh) return lv;
            lv.push_back(e);
        }
        power <<= 1; lh >>= 1; lw >>= 1;
        if (lw < 80 || lh < 40) break;
    }
    return lv;
}

// getGaussianKernel(n, sigma, CV_32F) for gaussian_2D_convolutionV2's kernel size rule (nldiffusion_functions.cpp:39-58)
AkTaps ak_taps(float sigma)
{
    AkTaps t{};
    int k = (int)ceil(2.0f * (1.0f + (sigma - 0.8f) / (0.3f)));
    if ((k % 2) == 0) k += 1;
    t.n = k;
    const double s = sigma;
    const double scale2X = -0.5 / (s * s);
    double sum = 0;
    for (int i = 0; i < k; ++i) {
        const double x = i - (k - 1) * 0.5;
        const double v = std::exp(scale2X * x * x);
        t.k[i] = (float)v;
        sum += t.k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < k; ++i) t.k[i] = (float)(t.k[i] * sum);
    return t;
}

// fed_tau_by_process_timeV2(T, 1, 0.25, reordering) (fed.cpp)
bool ak_is_prime(int number)
{
    if (number <= 1) return false;
    if (number == 1 || number == 2 || number == 3 || number == 5 || number == 7) return true;
    if ((number % 2) == 0 || (number % 3) == 0 || (number % 5) == 0 || (number % 7) == 0) return false;
    bool is_prime = true;
    const int upper = (int)sqrt(1.0f + number);
    for (int divisor = 11; divisor <= upper; divisor += 2) if (number % divisor == 0) is_prime = false;
    return is_prime;
}
std::vector<float> ak_fed_tau(float T)
{
    const float tau_max = 0.25f;
    const int n = (int)(ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
    std::vector<float> tau;
    if (n <= 0) return tau;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    std::vector<float> tauh(n);
    const float cc = 1.0f / (4.0f * n + 2.0f);
    const float d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) { const float hh = cosf((float)3.1415926535897932384626433832795 * (2.0f * k + 1.0f) * cc); tauh[k] = d / (hh * hh); }
    if (n == 1) return tauh;
    const int kappa = n / 2;
    int prime = n + 1;
    while (!ak_is_prime(prime)) prime++;
    tau.resize(n);
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index = 0;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return tau;
}

// computeResizeAreaTab (imgproc/resize.cpp) as a CSR over destination cells
void ak_area_tab(int ssize, int dsize, std::vector<AkAreaTab>& tab, std::vector<int>& begin)
{
    const double scale = (double)ssize / dsize;
    tab.clear(); begin.assign(dsize + 1, 0);
    for (int dx = 0; dx < dsize; ++dx) {
        begin[dx] = (int)tab.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) tab.push_back({sx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
    begin[dsize] = (int)tab.size();
}

}  // namespace

static int detect_akaze_impl(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                             float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out, unsigned char* mldb_out)
{
    if (!c || !image || !n_out || (cap && !keypoints_out)) return R3DM_ERR_INVALID;
    *n_out = 0;
    if (width < 3 || height < 3 || (uint64_t)width * height > (1ull << 30)) return R3DM_ERR_INVALID;
    R3DM_HIP(c, hipSetDevice(c->device));
    const double t_call = now_ms();
    const int w = (int)width, h = (int)height;
    const std::vector<AkLevelHost> lv = ak_levels(w, h);
    const int nl = (int)lv.size();
    if (nl == 0) return R3DM_OK;                                  // image too small for a single evolution level
    hipStream_t st = c->stream;
    const size_t n0 = (size_t)w * h;

    // ---- buffers: [0] image, [1..10] level-0 sized work images, then 4 per level (Lt, Lx, Ly, Ldet)
    enum { B_IMG = 0, B_SMOOTH, B_LXX, B_LXY, B_LYY, B_TMP, B_TMP2, B_WX, B_WY, B_FLOW, B_LT2, B_SMALL, B_LEVEL0 };
    if (c->ak_w != w || c->ak_h != h || c->ak_bufs.size() != (size_t)B_LEVEL0 + 4 * nl + 8) {
        for (DevBuf& b : c->ak_bufs) b.release();
        c->ak_bufs.assign((size_t)B_LEVEL0 + 4 * nl + 8, DevBuf());
        c->ak_w = w; c->ak_h = h;
    }
    auto buf = [&](int k) -> DevBuf& { return c->ak_bufs[k]; };
    for (int k = B_IMG; k <= B_LT2; ++k) R3DM_HIP(c, buf(k).ensure(n0 * 4));
    R3DM_HIP(c, buf(B_SMALL).ensure(4096 * 4));
    for (int i = 0; i < nl; ++i)
        for (int q = 0; q < 4; ++q) R3DM_HIP(c, buf(B_LEVEL0 + 4 * i + q).ensure((size_t)lv[i].w * lv[i].h * 4));
    auto Lt = [&](int i) { return buf(B_LEVEL0 + 4 * i).as<float>(); };
    auto Lx = [&](int i) { return buf(B_LEVEL0 + 4 * i + 1).as<float>(); };
    auto Ly = [&](int i) { return buf(B_LEVEL0 + 4 * i + 2).as<float>(); };
    auto Ldet = [&](int i) { return buf(B_LEVEL0 + 4 * i + 3).as<float>(); };
    float* img = buf(B_IMG).as<float>();
    float* smooth = buf(B_SMOOTH).as<float>();
    float* lxx = buf(B_LXX).as<float>(); float* lxy = buf(B_LXY).as<float>(); float* lyy = buf(B_LYY).as<float>();
    float* tmp = buf(B_TMP).as<float>(); float* tmp2 = buf(B_TMP2).as<float>();
    float* wx = buf(B_WX).as<float>(); float* wy = buf(B_WY).as<float>();
    float* flow = buf(B_FLOW).as<float>(); float* lt2 = buf(B_LT2).as<float>();
    uint32_t* small = buf(B_SMALL).as<uint32_t>();

    R3DM_HIP(c, hipMemcpyAsync(img, image, n0 * 4, hipMemcpyDefault, st));
    const AkTaps taps_off = ak_taps(1.6f), taps_one = ak_taps(1.0f);

    // Compute_Determinant_Hessian_Response_Single (AKAZEFeatures.cpp:389-410)
    auto hessian = [&](int i) -> hipError_t {
        const int lw = lv[i].w, lh = lv[i].h, s = lv[i].sigma_size;
        hipError_t e;
        if ((e = ak_scaled_deriv(st, smooth, tmp, Lx(i), lw, lh, s, 1)) != hipSuccess) return e;
        if ((e = ak_scaled_deriv(st, Lx(i), tmp, lxx, lw, lh, s, 1)) != hipSuccess) return e;
        if ((e = ak_scaled_deriv(st, Lx(i), tmp, lxy, lw, lh, s, 0)) != hipSuccess) return e;
        if ((e = ak_scaled_deriv(st, smooth, tmp, Ly(i), lw, lh, s, 0)) != hipSuccess) return e;
        if ((e = ak_scaled_deriv(st, Ly(i), tmp, lyy, lw, lh, s, 0)) != hipSuccess) return e;
        return ak_det(st, lxx, lyy, lxy, Ldet(i), (size_t)lw * lh);
    };

    // ---- Create_Nonlinear_Scale_Space (:245-369), Compute_Base_Evolution_Level (:199-237)
    R3DM_HIP(c, ak_gaussian(st, img, tmp, smooth, w, h, taps_off));
    R3DM_HIP(c, hessian(0));
    float kcontrast = 0.03f;
    if (nl > 1) {
        // compute_k_percentileV2 (nldiffusion_functions.cpp:212-262): maximum and 300-bin histogram on the device, the scan here
        const int nbins = 300;
        R3DM_HIP(c, ak_gaussian(st, img, tmp, flow, w, h, taps_one));
        R3DM_HIP(c, ak_scharr(st, flow, tmp, tmp2, wx, wy, w, h));
        R3DM_HIP(c, hipMemsetAsync(small, 0, 4096 * 4, st));
        R3DM_HIP(c, ak_modg_max(st, wx, wy, w, h, small));
        uint32_t hbits = 0;
        R3DM_HIP(c, hipMemcpyAsync(&hbits, small, 4, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipStreamSynchronize(st));
        float hmax; memcpy(&hmax, &hbits, 4);
        if (hmax != 0.0f) {
            const float sc = (nbins - 1) / hmax;
            R3DM_HIP(c, ak_modg_hist(st, wx, wy, w, h, sc, nbins, small + 16));
            std::vector<uint32_t> hist(nbins);
            R3DM_HIP(c, hipMemcpyAsync(hist.data(), small + 16, nbins * 4, hipMemcpyDeviceToHost, st));
            R3DM_HIP(c, hipStreamSynchronize(st));
            const size_t total = (size_t)(w - 2) * (h - 2);
            const int nthreshold = (int)((total - hist[0]) * 0.7f);
            int nelements = 0;
            for (int k = 1; k < nbins; ++k) {
                if (nelements >= nthreshold) { kcontrast = (float)hmax * k / nbins; break; }
                nelements = nelements + (int)hist[k];
            }
        }
    }
    R3DM_HIP(c, hipMemcpyAsync(Lt(0), smooth, n0 * 4, hipMemcpyDeviceToDevice, st));
    DevBuf tab_buf;
    for (int i = 1; i < nl; ++i) {
        const int lw = lv[i].w, lh = lv[i].h;
        const size_t n = (size_t)lw * lh;
        if (lv[i].octave > lv[i - 1].octave) {
            const int sw = lv[i - 1].w, sh = lv[i - 1].h;
            const AkAreaTab* xt = nullptr; const AkAreaTab* yt = nullptr; const int* xb = nullptr; const int* yb = nullptr;
            if (lw * 2 != sw || lh * 2 != sh) {
                std::vector<AkAreaTab> tx, ty; std::vector<int> bx, by;
                ak_area_tab(sw, lw, tx, bx); ak_area_tab(sh, lh, ty, by);
                const size_t bytes = (tx.size() + ty.size()) * sizeof(AkAreaTab) + (bx.size() + by.size()) * 4 + 64;
                R3DM_HIP(c, hipStreamSynchronize(st));             // the previous table may still be in use
                R3DM_HIP(c, tab_buf.ensure(bytes));
                unsigned char* base = tab_buf.as<unsigned char>();
                size_t o = 0;
                R3DM_HIP(c, hipMemcpy(base + o, tx.data(), tx.size() * sizeof(AkAreaTab), hipMemcpyHostToDevice)); xt = (const AkAreaTab*)(base + o); o += tx.size() * sizeof(AkAreaTab);
                R3DM_HIP(c, hipMemcpy(base + o, ty.data(), ty.size() * sizeof(AkAreaTab), hipMemcpyHostToDevice)); yt = (const AkAreaTab*)(base + o); o += ty.size() * sizeof(AkAreaTab);
                R3DM_HIP(c, hipMemcpy(base + o, bx.data(), bx.size() * 4, hipMemcpyHostToDevice)); xb = (const int*)(base + o); o += bx.size() * 4;
                R3DM_HIP(c, hipMemcpy(base + o, by.data(), by.size() * 4, hipMemcpyHostToDevice)); yb = (const int*)(base + o);
            }
            R3DM_HIP(c, ak_halfsample(st, Lt(i - 1), Lt(i), sw, sh, xt, xb, yt, yb));
            kcontrast = kcontrast * 0.75f;
        } else {
            R3DM_HIP(c, hipMemcpyAsync(Lt(i), Lt(i - 1), n * 4, hipMemcpyDeviceToDevice, st));
        }
        R3DM_HIP(c, ak_gaussian(st, Lt(i), tmp, smooth, lw, lh, taps_one));
        R3DM_HIP(c, ak_scharr(st, smooth, tmp, tmp2, wx, wy, lw, lh));
        R3DM_HIP(c, hessian(i));
        R3DM_HIP(c, ak_pm_g2(st, wx, wy, flow, n, 1.0f / (kcontrast * kcontrast)));
        // Fast Explicit Diffusion: lt += lstep * 0.5 * tau_j (ping-pong between the level's Lt and a work image)
        const std::vector<float> tau = ak_fed_tau(lv[i].etime - lv[i - 1].etime);
        float* cur = Lt(i); float* oth = lt2;
        for (float step : tau) { R3DM_HIP(c, ak_fed_step(st, cur, flow, oth, lw, lh, step)); std::swap(cur, oth); }
        if (cur != Lt(i)) R3DM_HIP(c, hipMemcpyAsync(Lt(i), cur, n * 4, hipMemcpyDeviceToDevice, st));
    }

    // ---- Feature_Detection (:371-382): extrema -> in-level pruning -> cross-level pruning -> refinement + orientation
    std::vector<AkLevelDev> ld(nl);
    size_t rows_total = 0;
    for (int i = 0; i < nl; ++i) rows_total += (size_t)std::max(0, lv[i].h - 2 * lv[i].border);
    DevBuf& meta = buf(B_LEVEL0 + 4 * nl);                        // row counts/offsets + per-level counters + level table
    R3DM_HIP(c, meta.ensure(rows_total * 8 + (size_t)nl * 16 + (size_t)nl * sizeof(AkLevelDev) + 256));
    R3DM_HIP(c, hipMemsetAsync(meta.p, 0, rows_total * 8 + (size_t)nl * 16, st));
    {
        uint32_t* rc = meta.as<uint32_t>();
        uint32_t* cnt = rc + 2 * rows_total;
        size_t ro = 0;
        for (int i = 0; i < nl; ++i) {
            AkLevelDev& L = ld[i];
            L = AkLevelDev{};
            L.w = lv[i].w; L.h = lv[i].h; L.border = lv[i].border; L.ratio = lv[i].ratio; L.psize = lv[i].esigma * 1.5f;
            L.Ldet = Ldet(i); L.Lx = Lx(i); L.Ly = Ly(i); L.Lt = Lt(i);
            L.row_cnt = rc + ro; L.row_off = rc + rows_total + ro; L.counts = cnt + 4 * i;
            ro += (size_t)std::max(0, lv[i].h - 2 * lv[i].border);
        }
    }
    AkLevelDev* d_levels = reinterpret_cast<AkLevelDev*>(meta.as<unsigned char>() + ((rows_total * 8 + (size_t)nl * 16 + 15) / 16) * 16);
    for (int i = 0; i < nl; ++i) R3DM_HIP(c, ak_extrema(st, ld[i], threshold, 0));
    R3DM_HIP(c, hipMemcpyAsync(d_levels, ld.data(), nl * sizeof(AkLevelDev), hipMemcpyHostToDevice, st));
    R3DM_HIP(c, ak_scan_rows(st, d_levels, nl));
    std::vector<uint32_t> counts(4 * (size_t)nl);
    R3DM_HIP(c, hipMemcpyAsync(counts.data(), ld[0].counts, counts.size() * 4, hipMemcpyDeviceToHost, st));
    R3DM_HIP(c, hipStreamSynchronize(st));
    size_t cand_total = 0;
    for (int i = 0; i < nl; ++i) cand_total += counts[4 * i];
    DevBuf& pts = buf(B_LEVEL0 + 4 * nl + 1);
    // per candidate slot: cand 16 + list 16 + live 16 + out0 16 + out1 8 + valid 4 + dead 2
    R3DM_HIP(c, pts.ensure(cand_total * 80 + 256));
    {
        unsigned char* base = pts.as<unsigned char>();
        size_t off = 0;
        for (int i = 0; i < nl; ++i) {
            const size_t n = counts[4 * i];
            ld[i].cand = (float4*)(base + 0 * cand_total * 16) + off;
            ld[i].list = (float4*)(base + 1 * cand_total * 16) + off;
            ld[i].live = (float*)(base + 2 * cand_total * 16) + 4 * off;
            ld[i].out0 = (float4*)(base + 3 * cand_total * 16) + off;
            ld[i].out1 = (float2*)(base + 4 * cand_total * 16) + off;
            ld[i].out_valid = (uint32_t*)(base + 4 * cand_total * 16 + cand_total * 8) + off;
            ld[i].dead_lower = base + 4 * cand_total * 16 + cand_total * 12 + off;
            ld[i].dead_upper = base + 4 * cand_total * 16 + cand_total * 13 + off;
            off += n;
        }
    }
    R3DM_HIP(c, hipMemsetAsync(pts.as<unsigned char>() + 4 * cand_total * 16 + cand_total * 12, 0, cand_total * 2 + 64, st));
    R3DM_HIP(c, hipMemcpyAsync(d_levels, ld.data(), nl * sizeof(AkLevelDev), hipMemcpyHostToDevice, st));
    for (int i = 0; i < nl; ++i) if (counts[4 * i]) R3DM_HIP(c, ak_extrema(st, ld[i], threshold, 1));
    R3DM_HIP(c, ak_prune_levels(st, d_levels, nl));
    R3DM_HIP(c, hipMemcpyAsync(counts.data(), ld[0].counts, counts.size() * 4, hipMemcpyDeviceToHost, st));
    R3DM_HIP(c, hipStreamSynchronize(st));
    uint32_t max_list = 0;
    for (int i = 0; i < nl; ++i) max_list = std::max(max_list, counts[4 * i + 1]);
    R3DM_HIP(c, ak_cross(st, d_levels, nl, max_list, 0));
    R3DM_HIP(c, ak_cross(st, d_levels, nl, max_list, 1));
    R3DM_HIP(c, ak_refine(st, d_levels, nl, max_list));

    // ---- gather in (level, list) order; angle = getAngleV2(maxX, maxY) then the detectKeypoints conversion (:604-613)
    uint32_t n_kp = 0;
    std::vector<AkMldbItem> items;
    for (int i = 0; i < nl; ++i) {
        const uint32_t n = counts[4 * i + 1];
        if (!n) continue;
        std::vector<float4> o0(n); std::vector<float2> o1(n); std::vector<uint32_t> ov(n);
        R3DM_HIP(c, hipMemcpyAsync(o0.data(), ld[i].out0, n * 16, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipMemcpyAsync(o1.data(), ld[i].out1, n * 8, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipMemcpyAsync(ov.data(), ld[i].out_valid, n * 4, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipStreamSynchronize(st));
        for (uint32_t j = 0; j < n; ++j) {
            if (!ov[j]) continue;
            if (n_kp < cap) {
                float theta = atan2f(o1[j].y, o1[j].x);
                if (!(theta >= 0)) theta = theta + (float)(2.0f * 3.1415926535897932384626433832795);
                if (mldb_out)          // Get_MLDB_Full_Descriptor: level coordinates, cos / sin of the raw (radian) angle
                    items.push_back({(uint32_t)i, o0[j].x / lv[i].ratio, o0[j].y / lv[i].ratio, cosf(theta), sinf(theta), (float)lv[i].sigma_size});
                float ang = theta;
                ang *= 180.0 / 3.1415926535897932384626433832795;
                ang += 90.0f;
                while (ang < 0) ang += 360.0f;
                while (ang > 360.0f) ang -= 360.0f;
                keypoints_out[4 * (size_t)n_kp] = o0[j].x; keypoints_out[4 * (size_t)n_kp + 1] = o0[j].y;
                keypoints_out[4 * (size_t)n_kp + 2] = o0[j].z; keypoints_out[4 * (size_t)n_kp + 3] = ang;
                if (responses_out) responses_out[n_kp] = o0[j].w;
            }
            ++n_kp;
        }
    }
    if (mldb_out && !items.empty()) {
        // comparison table of MLDB_Binary_Comparisons: per grid, per channel, all value pairs i < j
        std::vector<unsigned char> pairs;
        const int bases[3] = {0, 12, 39}, cnts[3] = {4, 9, 16};
        for (int g = 0; g < 3; ++g)
            for (int pos = 0; pos < 3; ++pos)
                for (int i = 0; i < cnts[g]; ++i)
                    for (int j = i + 1; j < cnts[g]; ++j) { pairs.push_back((unsigned char)(bases[g] + 3 * i + pos)); pairs.push_back((unsigned char)(bases[g] + 3 * j + pos)); }
        DevBuf& mb = buf(B_LEVEL0 + 4 * nl + 2);
        const size_t ni = items.size();
        R3DM_HIP(c, mb.ensure(ni * sizeof(AkMldbItem) + 1024 + ni * 61 + 64));
        unsigned char* base = mb.as<unsigned char>();
        R3DM_HIP(c, hipMemcpyAsync(base, items.data(), ni * sizeof(AkMldbItem), hipMemcpyHostToDevice, st));
        R3DM_HIP(c, hipMemcpyAsync(base + ni * sizeof(AkMldbItem), pairs.data(), pairs.size(), hipMemcpyHostToDevice, st));
        unsigned char* d_out = base + ni * sizeof(AkMldbItem) + 1024;
        R3DM_HIP(c, ak_mldb(st, d_levels, (const AkMldbItem*)base, (uint32_t)ni, base + ni * sizeof(AkMldbItem), d_out));
        R3DM_HIP(c, hipMemcpyAsync(mldb_out, d_out, ni * 61, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipStreamSynchronize(st));
    }
    tab_buf.release();
    *n_out = n_kp;
    c->stats.ms_detect = now_ms() - t_call;
    return R3DM_OK;
}

extern "C" int r3dm_detect_akaze(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                 float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out)
{
    return detect_akaze_impl(c, image, width, height, threshold, keypoints_out, responses_out, cap, n_out, nullptr);
}

extern "C" int r3dm_detect_akaze_mldb(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                      float* keypoints_out, unsigned char* descriptors_out, uint32_t cap, uint32_t* n_out)
{
    if (!descriptors_out && cap) return R3DM_ERR_INVALID;
    return detect_akaze_impl(c, image, width, height, threshold, keypoints_out, nullptr, cap, n_out, descriptors_out);
}

// ------------------------------------------------------------------------------------------------
// LIOP descriptor on patches
// ------------------------------------------------------------------------------------------------
// geometry of the 41x41 patch exactly as vl_liopdesc_new builds it (vl_liop.c:371-421): circular support
// dx^2+dy^2 <= (long)((center - radius + 0.6)^2), 4 samples per pixel on a circle of radius 6 starting at
// atan2(y, x); computed once on the host with the host libm (like the reference) and kept in HBM
static int liop_prepare(r3dm_ctx* c)
{
    if (c->liop_npix) return R3DM_OK;
    const int side = 41, center = (side - 1) / 2;
    const double radius = 6.0, t = center - radius + 0.6;
    const long t2 = (long)(t * t);
    std::vector<int> pix;
    for (int y = 0; y < side; ++y)
        for (int x = 0; x < side; ++x) {
            const long dx = x - center, dy = y - center;
            if (x == 0 && y == 0) continue;
            if (dx * dx + dy * dy <= t2) pix.push_back(x + y * side);
        }
    std::vector<double> sx(4 * pix.size()), sy(4 * pix.size());
    const double dangle = 2 * M_PI / 4.0;
    for (size_t i = 0; i < pix.size(); ++i) {
        const double x = (pix[i] % side) - center, y = (pix[i] / side) - center;
        const double angle0 = std::atan2(y, x);
        for (int k = 0; k < 4; ++k) {
            sx[4 * i + k] = x + radius * std::cos(angle0 + dangle * k) + center;
            sy[4 * i + k] = y + radius * std::sin(angle0 + dangle * k) + center;
        }
    }
    if (pix.size() > 1024) { c->err = "liop: support larger than the sort capacity"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, c->liop_pix.ensure(pix.size() * 4));
    R3DM_HIP(c, c->liop_sx.ensure(sx.size() * 8));
    R3DM_HIP(c, c->liop_sy.ensure(sy.size() * 8));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_pix.p, pix.data(), pix.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_sx.p, sx.data(), sx.size() * 8, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_sy.p, sy.data(), sy.size() * 8, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    c->liop_npix = (uint32_t)pix.size();
    return R3DM_OK;
}

extern "C" int r3dm_liop_describe_patches(r3dm_ctx* c, const float* patches, uint32_t n, uint32_t side, float* desc_out,
                                          uint32_t* n_resorted)
{
    if (!c || (n && (!patches || !desc_out))) return R3DM_ERR_INVALID;
    if (side != 41) { c->err = "liop: only the 41x41 patch of Regard3D (patchResolution 20) is supported"; return R3DM_ERR_UNSUPPORTED; }
    R3DM_HIP(c, hipSetDevice(c->device));
    int rc = liop_prepare(c);
    if (rc != R3DM_OK) return rc;
    if (n_resorted) *n_resorted = 0;
    if (n == 0) return R3DM_OK;
    const size_t in_bytes = (size_t)n * 41 * 41 * 4, out_bytes = (size_t)n * 144 * 4;
    R3DM_HIP(c, c->liop_in.ensure(in_bytes));
    R3DM_HIP(c, c->liop_out.ensure(out_bytes));
    R3DM_HIP(c, c->liop_cnt.ensure(64));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_in.p, patches, in_bytes, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, launch_liop(c->stream, c->liop_in.as<float>(), c->liop_pix.as<int>(), c->liop_sx.as<double>(),
                            c->liop_sy.as<double>(), n, c->liop_npix, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>()));
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(desc_out, c->liop_out.p, out_bytes, hipMemcpyDefault, c->stream));
    uint32_t nt = 0;
    R3DM_HIP(c, hipMemcpyAsync(&nt, c->liop_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    if (n_resorted) *n_resorted = nt;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_liop_kernel = ms;
    return R3DM_OK;
}

extern "C" int r3dm_extract_liop(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height,
                                 const float* keypoints, uint32_t n, float kp_size_factor, float* desc_out, float* patches_out)
{
    if (!c || !image || width == 0 || height == 0 || (n && (!keypoints || !desc_out))) return R3DM_ERR_INVALID;
    R3DM_HIP(c, hipSetDevice(c->device));
    int rc = liop_prepare(c);
    if (rc != R3DM_OK) return rc;
    if (n == 0) return R3DM_OK;
    // keypoints to the host (they may live in device memory), 2x3 inverse maps exactly as :786-799 computes them
    std::vector<float> kp(4 * (size_t)n), M6(6 * (size_t)n);
    R3DM_HIP(c, hipMemcpyAsync(kp.data(), keypoints, kp.size() * 4, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    const int patchResolution = 20, patchSize = 41;
    for (uint32_t k = 0; k < n; ++k) {
        const float x = kp[4 * k], y = kp[4 * k + 1];
        const float angle = -90.0f - kp[4 * k + 3];
        const float scale = kp[4 * k + 2] / static_cast<float>(patchSize) * kp_size_factor;
        const float alpha = scale * std::cos(angle * M_PI / 180.0f);
        const float beta = scale * std::sin(angle * M_PI / 180.0f);
        const float trans_x = x - static_cast<float>(patchResolution), trans_y = y - static_cast<float>(patchResolution);
        float* m = &M6[6 * (size_t)k];
        m[0] = alpha; m[1] = beta;  m[2] = beta * trans_y + alpha * trans_x - beta * y + (1.0f - alpha) * x;
        m[3] = -beta; m[4] = alpha; m[5] = alpha * trans_y - beta * trans_x + beta * x + (1.0f - alpha) * y;
    }
    // cv::getGaussianKernel(11, 1.2, CV_32F)
    float kern[11];
    {
        const double scale2X = -0.5 / (1.2 * 1.2);
        double sum = 0;
        for (int i = 0; i < 11; ++i) { const double xx = i - 5.0; kern[i] = (float)std::exp(scale2X * xx * xx); sum += kern[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 11; ++i) kern[i] = (float)(kern[i] * sum);
    }
    const size_t img_bytes = (size_t)width * height * 4, patch_bytes = (size_t)n * 41 * 41 * 4, out_bytes = (size_t)n * 144 * 4;
    R3DM_HIP(c, c->liop_img.ensure(img_bytes));
    R3DM_HIP(c, c->liop_M.ensure(M6.size() * 4));
    R3DM_HIP(c, c->liop_kern.ensure(64));
    R3DM_HIP(c, c->liop_in.ensure(patch_bytes));
    R3DM_HIP(c, c->liop_out.ensure(out_bytes));
    R3DM_HIP(c, c->liop_cnt.ensure(64));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_img.p, image, img_bytes, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_M.p, M6.data(), M6.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_kern.p, kern, sizeof(kern), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, launch_liop_extract(c->stream, c->liop_img.as<float>(), (int)width, (int)height, c->liop_M.as<float>(),
                                    c->liop_kern.as<float>(), n, c->liop_in.as<float>()));
    R3DM_HIP(c, launch_liop(c->stream, c->liop_in.as<float>(), c->liop_pix.as<int>(), c->liop_sx.as<double>(),
                            c->liop_sy.as<double>(), n, c->liop_npix, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>()));
    R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(desc_out, c->liop_out.p, out_bytes, hipMemcpyDefault, c->stream));
    if (patches_out) R3DM_HIP(c, hipMemcpyAsync(patches_out, c->liop_in.p, patch_bytes, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_liop_kernel = ms;
    return R3DM_OK;
}

// ------------------------------------------------------------------------------------------------
// the per-image work item of the features stage
// ------------------------------------------------------------------------------------------------
extern "C" int r3dm_gray_from_bgr8(r3dm_ctx* c, const unsigned char* bgr, uint32_t width, uint32_t height, float* gray_out)
{
    if (!c || !bgr || !gray_out || !width || !height) return R3DM_ERR_INVALID;
    R3DM_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)width * height;
    DevBuf in, out;
    R3DM_HIP(c, in.ensure(n * 3));
    R3DM_HIP(c, out.ensure(n * 4));
    R3DM_HIP(c, hipMemcpyAsync(in.p, bgr, n * 3, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, ak_bgr_to_gray(c->stream, in.as<unsigned char>(), out.as<float>(), n));
    R3DM_HIP(c, hipMemcpyAsync(gray_out, out.p, n * 4, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    in.release(); out.release();
    return R3DM_OK;
}

extern "C" int r3dm_extract_features_to_files(r3dm_ctx* c, const float* gray, uint32_t width, uint32_t height, float threshold,
                                              const char* feat_path, const char* desc_path, uint32_t* n_features)
{
    if (!c || !gray || !feat_path || !desc_path) return R3DM_ERR_INVALID;
    if (n_features) *n_features = 0;
    // detectAndExtract (src/Regard3DFeatures.cpp:206-222) for keypointDetectorList_ = {"Fast-AKAZE"}
    uint32_t n = 0;
    std::vector<float> kps(4 * 65536);
    int rc = r3dm_detect_akaze(c, gray, width, height, threshold, kps.data(), nullptr, 65536, &n);
    if (rc != R3DM_OK) return rc;
    if (n > 65536) {
        kps.resize(4 * (size_t)n);
        const uint32_t cap = n;
        rc = r3dm_detect_akaze(c, gray, width, height, threshold, kps.data(), nullptr, cap, &n);
        if (rc != R3DM_OK) return rc;
    }
    std::vector<float> desc(144 * (size_t)std::max<uint32_t>(n, 1));
    if (n) {
        rc = r3dm_extract_liop(c, gray, width, height, kps.data(), n, 8.0f /* getKpSizeFactor("Fast-AKAZE"), :703-704 */, desc.data(), nullptr);
        if (rc != R3DM_OK) return rc;
    }
    // KeypointSet::saveToBinFile (src/keypointSet.hpp:61-67): .feat = one "x y scale orientation" line per feature
    // (SIOPointFeature::operator<<, default float formatting; scale = size / 2, :835-836), .desc = count + raw rows
    FILE* f = fopen(feat_path, "w");
    if (!f) { c->err = std::string("cannot write ") + feat_path; return R3DM_ERR_IO; }
    for (uint32_t k = 0; k < n; ++k)
        fprintf(f, "%g %g %g %g\n", kps[4 * (size_t)k], kps[4 * (size_t)k + 1], kps[4 * (size_t)k + 2] / 2.0f, kps[4 * (size_t)k + 3]);
    if (fclose(f) != 0) return R3DM_ERR_IO;
    f = fopen(desc_path, "wb");
    if (!f) { c->err = std::string("cannot write ") + desc_path; return R3DM_ERR_IO; }
    const uint64_t cnt = n;
    bool ok = fwrite(&cnt, 8, 1, f) == 1 && (n == 0 || fwrite(desc.data(), 144 * 4, n, f) == n);
    ok = (fclose(f) == 0) && ok;
    if (!ok) return R3DM_ERR_IO;
    if (n_features) *n_features = n;
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

extern "C" int r3dm_graph_from_csr(const uint32_t* pairs_ij, uint64_t n_pairs, const uint64_t* offsets,
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

extern "C" int r3dm_graph_merge(const r3dm_graph* const* parts, uint32_t n_parts, r3dm_graph** out)
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
        std::string buf;
        buf.reserve(1 << 20);
        char tmp[64];
        for (uint64_t p = 0; p < np && ok; ++p) {
            const uint64_t cnt = g->offsets[p + 1] - g->offsets[p];
            int len = snprintf(tmp, sizeof(tmp), "%u %u\n%llu\n", g->pairs[2 * p], g->pairs[2 * p + 1], (unsigned long long)cnt);
            buf.append(tmp, len);
            for (uint64_t k = g->offsets[p]; k < g->offsets[p + 1]; ++k) {
                len = snprintf(tmp, sizeof(tmp), "%u %u\n", g->matches[k].i, g->matches[k].j);
                buf.append(tmp, len);
            }
            if (buf.size() > (1 << 20) - 4096) { ok &= fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
        }
        if (ok && !buf.empty()) ok &= fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    }
    ok &= (fclose(f) == 0);
    return ok ? R3DM_OK : R3DM_ERR_IO;
}

extern "C" int r3dm_load_matches(const char* path, r3dm_graph** out)
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
        for (uint64_t p = 0; p < np && ok; ++p) {
            uint32_t ij[2]; uint64_t cnt = 0;
            ok = fread(ij, 4, 2, f) == 2 && fread(&cnt, 8, 1, f) == 1 && cnt < (1ull << 32);
            if (!ok) break;
            const size_t at = m.size();
            m.resize(at + cnt);
            ok = fread(m.data() + at, sizeof(r3dm_match), cnt, f) == cnt;
            pairs.push_back(ij[0]); pairs.push_back(ij[1]); offs.push_back(m.size());
        }
    } else {
        unsigned I, J; unsigned long long cnt;
        while (fscanf(f, "%u %u %llu", &I, &J, &cnt) == 3) {
            for (unsigned long long k = 0; k < cnt; ++k) {
                unsigned a, b;
                if (fscanf(f, "%u %u", &a, &b) != 2) { ok = false; break; }
                m.push_back({a, b});
            }
            if (!ok) break;
            pairs.push_back(I); pairs.push_back(J); offs.push_back(m.size());
        }
    }
    fclose(f);
    if (!ok) return R3DM_ERR_IO;
    return r3dm_graph_from_csr(pairs.data(), pairs.size() / 2, offs.data(), m.data(), out);
}
