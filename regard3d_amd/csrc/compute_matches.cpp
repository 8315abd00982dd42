// compute_matches.cpp -- implementation of the R3DComputeMatches facade (include/r3d_compute_matches.hpp)
// on top of the C ABI only.  Stage order follows /root/reference/src/R3DComputeMatches.cpp:1996-2129:
// load regions -> exhaustive pairs -> match -> save matches.putative.txt -> F filter -> save matches.f.txt.
#include "../../include/r3d_compute_matches.hpp"

#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace r3d_amd {

namespace {

// .feat: one "x y scale orientation" text line per feature; .desc: 8-byte count + raw rows
// (/root/reference/src/keypointSet.hpp:49-67 -> OpenMVG loadFeatsFromFile / loadDescsFromBinFile)
bool load_feat(const std::string& path, std::vector<float>& xy)
{
    // the whole file in one buffer, one from_chars per field (what `stream >> float` does, without the stream): groups of four
    // numbers until the first group that is not complete, like `while (f >> x >> y >> s >> o)`
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<char> txt;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long sz = ok ? ftell(f) : -1;
    ok = ok && sz >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok) { txt.resize((size_t)sz + 1); ok = fread(txt.data(), 1, (size_t)sz, f) == (size_t)sz; txt[(size_t)sz] = 0; }
    fclose(f);
    if (!ok) return false;
    xy.clear();
    // locale-independent, like the classic-locale `stream >> float` of OpenMVG's reader (the host application calls
    // setlocale(LC_ALL, ""), so strtof would read "12.5" as 12 under a comma-decimal locale): std::from_chars, after the white
    // space and the optional sign that operator>> accepts; hex / inf / nan tokens are not numbers for operator>> and end the file here too
    const char* s = txt.data();
    const char* const end_txt = s + (size_t)sz;
    for (;;) {
        float v[4];
        int k = 0;
        for (; k < 4; ++k) {
            while (s < end_txt && (*s == ' ' || *s == '\n' || *s == '\t' || *s == '\r' || *s == '\f' || *s == '\v')) ++s;
            const char* t = s;
            bool neg = false;
            if (t < end_txt && (*t == '+' || *t == '-')) { neg = *t == '-'; ++t; }
            if (t >= end_txt || !((*t >= '0' && *t <= '9') || *t == '.')) break;
            const std::from_chars_result r = std::from_chars(t, end_txt, v[k], std::chars_format::general);
            if (r.ec == std::errc::invalid_argument) break;
            if (r.ec == std::errc::result_out_of_range) v[k] = 0.0f;     // (never for pixel coordinates) operator>> sets failbit; keep going with 0
            if (neg) v[k] = -v[k];
            s = r.ptr;
        }
        if (k < 4) break;
        xy.push_back(v[0]); xy.push_back(v[1]);
    }
    return true;
}

// first thing in a background writer thread: its work has a whole phase to hide behind, so under contention for the host's cores it
// stands back (Linux: a per-thread nice value, inherited by the helpers it starts)
// (only when the host asked for it: R3DComputeMatches::setBackgroundThreadsNice)
static inline void stand_back(int nice_value) { if (nice_value > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), nice_value); }

// Save(PairWiseMatches, file) of the .txt and the .bin of a graph on two host threads while the next filter runs on the GPU; the
// destructor waits for every write and frees the graphs it was given
struct MatchFileWriter {
    struct Job { std::thread th; int rc = R3DM_OK; std::string path; };
    std::vector<std::unique_ptr<Job>> jobs;
    std::vector<r3dm_graph*> owned;
    int nice_value = 0;
    void save(const r3dm_graph* g, const std::string& txt_path, const std::string& bin_path)
    {
        const int nv = nice_value;
        for (const std::string& p : {txt_path, bin_path}) {
            std::unique_ptr<Job> j(new Job());
            j->path = p;
            Job* raw = j.get();
            try { raw->th = std::thread([g, raw, nv]() noexcept { stand_back(nv); raw->rc = r3dm_save_matches(g, raw->path.c_str()); }); }
            catch (...) { raw->rc = r3dm_save_matches(g, raw->path.c_str()); }             // no thread to be had: write here
            jobs.push_back(std::move(j));
        }
    }
    void own(r3dm_graph* g) { owned.push_back(g); }
    // waits for all writes; the path of the first one that failed, or ""
    std::string finish()
    {
        std::string bad;
        for (auto& j : jobs) { if (j->th.joinable()) j->th.join(); if (j->rc != R3DM_OK && bad.empty()) bad = j->path; }
        jobs.clear();
        for (r3dm_graph* g : owned) r3dm_graph_free(g);
        owned.clear();
        return bad;
    }
    ~MatchFileWriter() { (void)finish(); }
};

bool load_desc(const std::string& path, size_t row_bytes, std::vector<unsigned char>& data, uint64_t& n)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = fread(&n, 8, 1, f) == 1;
    if (ok) {
        // the count comes from the file: check it against what the file can hold before sizing anything by it
        ok = fseek(f, 0, SEEK_END) == 0;
        const long sz = ok ? ftell(f) : -1;
        ok = ok && sz >= 8 && row_bytes > 0 && n <= (uint64_t)(sz - 8) / row_bytes && fseek(f, 8, SEEK_SET) == 0;
    }
    if (ok) {
        data.resize((size_t)n * row_bytes);
        ok = n == 0 || fread(data.data(), row_bytes, n, f) == n;
    }
    fclose(f);
    return ok;
}

void graph_to_map(const r3dm_graph* g, PairWiseMatches& out)
{
    out.clear();
    const uint64_t np = r3dm_graph_num_pairs(g);
    const uint32_t* p = r3dm_graph_pairs(g);
    const uint64_t* o = r3dm_graph_offsets(g);
    const r3dm_match* m = r3dm_graph_matches(g);
    for (uint64_t k = 0; k < np; ++k)
        out.emplace(std::make_pair(p[2 * k], p[2 * k + 1]), MatchList(m + o[k], m + o[k + 1]));
}

// PairWiseMatchingToAdjacencyMatrixSVG (/root/reference/src/R3DComputeMatches.cpp:2074,2238; OpenMVG, external):
// an N x N grid, 5 px per view, a blue square at (J, I) for every pair that kept matches, axis labels 0 / N.
// Same geometry as upstream; the markup is not byte-identical to OpenMVG's svgDrawer output.
bool write_adjacency_svg(const std::string& path, size_t n_views, const PairWiseMatches& m)
{
    if (m.empty()) return true;                             // upstream writes nothing for an empty map
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    const double s = 5.0;
    const double wh = (n_views + 3) * s;
    fprintf(f, "<?xml version=\"1.0\" standalone=\"yes\"?>\n<svg width=\"%g\" height=\"%g\" version=\"1.1\" "
               "xmlns=\"http://www.w3.org/2000/svg\">\n", wh, wh);
    for (const auto& kv : m) {
        if (kv.second.empty()) continue;
        fprintf(f, "<rect x=\"%g\" y=\"%g\" width=\"%g\" height=\"%g\" fill=\"blue\" stroke=\"none\">"
                   "<title>(%u,%u %zu)</title></rect>\n",
                kv.first.second * s, kv.first.first * s, s / 2.0, s / 2.0, kv.first.second, kv.first.first, kv.second.size());
    }
    fprintf(f, "<text x=\"%g\" y=\"%g\" font-size=\"%g\" fill=\"black\">0</text>\n", (n_views + 1) * s, s, s);
    fprintf(f, "<text x=\"%g\" y=\"%g\" font-size=\"%g\" fill=\"black\">%zu</text>\n", (n_views + 1) * s, n_views * s - s, s, n_views);
    fprintf(f, "<polyline points=\"%g,0 %g,%g 0,%g\" fill=\"none\" stroke=\"black\" stroke-width=\"1\"/>\n",
            n_views * s, n_views * s, n_views * s, n_views * s);
    fprintf(f, "</svg>\n");
    return fclose(f) == 0;
}

double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (!f) return false; fclose(f); return true; }

std::string with_ext(const std::string& path, const char* ext)
{
    const size_t dot = path.find_last_of('.');
    return (dot == std::string::npos ? path : path.substr(0, dot)) + ext;
}

}  // namespace

R3DComputeMatches::R3DComputeMatches(int device_id)
{
    devices_.assign(1, device_id);
    const int rc = r3dm_create(device_id, &ctx_);
    if (rc != R3DM_OK) { ctx_ = nullptr; errorMessage_ = "r3dm_create failed (" + std::to_string(rc) + "): no gfx950 GPU"; }
    setExactFastPaths(true);
}

R3DComputeMatches::R3DComputeMatches(const std::vector<int>& device_ids)
{
    devices_ = device_ids;
    if (device_ids.size() == 1) {
        const int rc = r3dm_create(device_ids[0], &ctx_);
        if (rc != R3DM_OK) { ctx_ = nullptr; errorMessage_ = "r3dm_create failed (" + std::to_string(rc) + "): no gfx950 GPU"; }
        setExactFastPaths(true);
        return;
    }
    const int rc = r3dm_multi_create(device_ids.data(), (int)device_ids.size(), &multi_);
    if (rc != R3DM_OK) { multi_ = nullptr; errorMessage_ = "r3dm_multi_create failed (" + std::to_string(rc) + ")"; }
    setExactFastPaths(true);
}

R3DComputeMatches::~R3DComputeMatches()
{
    if (feat_multi_) r3dm_multi_destroy(feat_multi_);
    if (ctx_) r3dm_destroy(ctx_);
    if (multi_) r3dm_multi_destroy(multi_);
}

// r3dm_features_sink of the features stage: the view is registered with the matcher from the worker thread that computed it
// (descriptors device to device, positions as written to the .feat file).  A registration that fails is not an error: the view
// is then read back from its files like any other.
int R3DComputeMatches::features_sink(void* self_, uint32_t image_index, uint32_t n_features, const float* desc_device, const float* xy_as_written)
{
    R3DComputeMatches* self = static_cast<R3DComputeMatches*>(self_);
    try {                                                  // nothing leaves a C callback by exception (the caller is a worker thread of the library)
        const size_t vi = self->sink_need_[image_index];
        const View& v = self->views_[vi];
        std::lock_guard<std::mutex> lk(self->sink_mu_);
        const int rc = self->ctx_ ? r3dm_set_image(self->ctx_, v.id_view, v.ui_width, v.ui_height, desc_device, n_features, self->dim_, self->dtype_, xy_as_written)
                                  : r3dm_multi_set_image(self->multi_, v.id_view, v.ui_width, v.ui_height, desc_device, n_features, self->dim_, self->dtype_, xy_as_written);
        if (rc == R3DM_OK) { self->registered_[vi] = 1; self->registered_n_[vi] = n_features; }
    } catch (...) {}
    return 0;
}

// R3DFeaturesThread::extractFeaturesAndDescriptors(vec_fileNames, sOutDir, params) (/root/reference/src/R3DComputeMatches.cpp:1994-1995,
// src/threads/R3DFeaturesThread.cpp:38-210) for the views whose two files are not both there.  The reference's worker threads
// admit one image at a time into the detector; here feat_conc_ batches of feat_batch_ same-size images are in flight per device.
bool R3DComputeMatches::runFeaturesStage(const R3DFParams& params, const std::string& dir)
{
    std::vector<size_t> need;
    for (size_t vi = 0; vi < views_.size(); ++vi)
        if (!(file_exists(dir + "/" + views_[vi].basename + ".feat") && file_exists(dir + "/" + views_[vi].basename + ".desc"))) need.push_back(vi);
    if (need.empty()) return true;
    // the detector list of the GUI: only the default arm runs on the GPU (include/regard3d_features.hpp says so too)
    for (const std::string& d : params.keypointDetectorList_)
        if (d != "Fast-AKAZE") { errorMessage_ = "keypoint detector \"" + d + "\" is not served by the GPU path (Fast-AKAZE is)"; return false; }
    if (dtype_ != R3DM_F32 || dim_ != 144) { errorMessage_ = "the features stage writes LIOP regions (float x 144); setRegionsType disagrees"; return false; }
    if (!feat_multi_) {
        std::vector<int> ids;
        for (int d : devices_) for (int k = 0; k < std::max(1, feat_conc_); ++k) ids.push_back(d);
        const int rc = r3dm_multi_create(ids.data(), (int)ids.size(), &feat_multi_);
        if (rc != R3DM_OK) { feat_multi_ = nullptr; errorMessage_ = "r3dm_multi_create (features stage) failed (" + std::to_string(rc) + ")"; return false; }
    }
    const int n_ctx = r3dm_multi_num_devices(feat_multi_);
    (void)r3dm_multi_set_features_sink(feat_multi_, direct_registration_ ? &R3DComputeMatches::features_sink : nullptr, this);
    // the .feat / .desc of a batch are written behind the sink calls, beside the match phase (computeMatches waits for them before it
    // returns): only when the views are registered straight from the device, else Regions_Provider::load reads those files next
    (void)r3dm_multi_set_deferred_feature_files(feat_multi_, direct_registration_ ? 1 : 0);
    (void)r3dm_multi_set_background_nice(feat_multi_, background_nice_);
    r3dm_features_totals before{};
    for (int k = 0; k < n_ctx; ++k) {
        r3dm_features_totals t{};
        (void)r3dm_get_features_totals(r3dm_multi_ctx(feat_multi_, k), &t);
        before.n_images += t.n_images; before.n_passes += t.n_passes; before.n_keypoints += t.n_keypoints; before.n_regrows += t.n_regrows;
        before.ms_detect_kernels += t.ms_detect_kernels; before.detect_algorithmic_bytes += t.detect_algorithmic_bytes;
        before.ms_liop_kernels += t.ms_liop_kernels; before.ms_wall += t.ms_wall; before.ms_files += t.ms_files;
    }
    // with a provider, pixels are requested one chunk at a time (every context gets two batches per chunk), else all at once
    const size_t chunk = provider_ ? (size_t)n_ctx * (size_t)std::max(1, feat_batch_) * 2 : need.size();
    for (size_t c0 = 0; c0 < need.size(); c0 += chunk) {
        const size_t cn = std::min(chunk, need.size() - c0);
        std::vector<const float*> grays(cn, nullptr); std::vector<const unsigned char*> bgrs(cn, nullptr);
        std::vector<uint32_t> ws(cn), hs(cn), nf(cn, 0), sk(cn, 0);
        std::vector<std::string> fps(cn), dps(cn);
        std::vector<const char*> fp(cn), dp(cn);
        std::vector<char> provided(cn, 0);
        bool ok = true;
        for (size_t k = 0; k < cn && ok; ++k) {
            const View& v = views_[need[c0 + k]];
            grays[k] = v.gray; bgrs[k] = v.bgr8;
            if (!grays[k] && !bgrs[k] && provider_) {
                Pixels px;
                if (provider_(v, &px, provider_user_)) { grays[k] = px.gray; bgrs[k] = px.bgr8; provided[k] = 1; }
            }
            if (!grays[k] && !bgrs[k]) {
                // the reference would cv::imread the file here; decoding is the caller's, so a view without files AND pixels is an error
                errorMessage_ = "Invalid features: " + v.basename + " (no .feat/.desc in the matches directory and no pixels for the features stage)";
                ok = false;
            }
            ws[k] = v.ui_width; hs[k] = v.ui_height;
            fps[k] = dir + "/" + v.basename + ".feat"; dps[k] = dir + "/" + v.basename + ".desc";
            fp[k] = fps[k].c_str(); dp[k] = dps[k].c_str();
        }
        int rc = R3DM_OK;
        char err[512] = {0};
        sink_need_ = need.data() + c0;
        if (ok) rc = r3dm_multi_extract_features_ex(feat_multi_, (uint32_t)cn, grays.data(), bgrs.data(), ws.data(), hs.data(), params.threshold_,
                                                    fp.data(), dp.data(), nf.data(), sk.data(), (uint32_t)std::max(1, feat_batch_), err, sizeof(err));
        if (provider_release_) for (size_t k = 0; k < cn; ++k) if (provided[k]) provider_release_(views_[need[c0 + k]], provider_user_);
        if (!ok) return false;
        if (rc != R3DM_OK) { errorMessage_ = std::string("features stage failed: ") + err; return false; }
        phases_.images_extracted += cn;
        if (progress_) progress_(0.7f * (float)(c0 + cn) / (float)need.size(), "Extracting features", progress_user_);   // sendMsgToMainFrame, :212-240
    }
    r3dm_features_totals& T = phases_.features_totals;
    T = r3dm_features_totals{};
    for (int k = 0; k < n_ctx; ++k) {
        r3dm_features_totals t{};
        (void)r3dm_get_features_totals(r3dm_multi_ctx(feat_multi_, k), &t);
        T.n_images += t.n_images; T.n_passes += t.n_passes; T.n_keypoints += t.n_keypoints; T.n_regrows += t.n_regrows;
        T.ms_detect_kernels += t.ms_detect_kernels; T.detect_algorithmic_bytes += t.detect_algorithmic_bytes;
        T.ms_liop_kernels += t.ms_liop_kernels; T.ms_wall += t.ms_wall; T.ms_files += t.ms_files;
    }
    T.n_images -= before.n_images; T.n_passes -= before.n_passes; T.n_keypoints -= before.n_keypoints; T.n_regrows -= before.n_regrows;
    T.ms_detect_kernels -= before.ms_detect_kernels; T.detect_algorithmic_bytes -= before.detect_algorithmic_bytes;
    T.ms_liop_kernels -= before.ms_liop_kernels; T.ms_wall -= before.ms_wall; T.ms_files -= before.ms_files;
    return true;
}

void R3DComputeMatches::addViews(const std::vector<View>& views) { views_.insert(views_.end(), views.begin(), views.end()); }

void R3DComputeMatches::setIntegerFastPath(bool on)
{
    if (ctx_) (void)r3dm_set_integer_mfma(ctx_, on ? 1 : 0);
    if (multi_) (void)r3dm_multi_set_integer_mfma(multi_, on ? 1 : 0);
}

void R3DComputeMatches::setSplitFastPath(bool on)
{
    if (ctx_) (void)r3dm_set_split_mfma(ctx_, on ? 1 : 0);
    if (multi_) for (int k = 0; k < r3dm_multi_num_devices(multi_); ++k) (void)r3dm_set_split_mfma(r3dm_multi_ctx(multi_, k), on ? 1 : 0);
}

void R3DComputeMatches::setHammingFastPath(bool on)
{
    if (ctx_) (void)r3dm_set_hamming_mfma(ctx_, on ? 1 : 0);
    if (multi_) for (int k = 0; k < r3dm_multi_num_devices(multi_); ++k) (void)r3dm_set_hamming_mfma(r3dm_multi_ctx(multi_, k), on ? 1 : 0);
}

void R3DComputeMatches::setRegionsType(r3dm_dtype dtype, uint32_t dim) { dtype_ = dtype; dim_ = dim; }

bool R3DComputeMatches::computeMatches(R3DFParams& params, bool svgOutput, const R3DProjectPaths& paths,
                                       int /*cameraModel*/, int matchingAlgorithm)
{
    statistics_ = R3DComputeMatchesStatistics();
    phases_ = PhaseTimes();
    if (!ctx_ && !multi_) return false;
    const double t_begin = wall_ms();
    // the feature files of this call are complete when it returns, however it returns (r3dm_set_deferred_feature_files)
    struct FeatureFilesGuard {
        R3DComputeMatches* self;
        ~FeatureFilesGuard() { if (self->feat_multi_) (void)r3dm_multi_features_files_wait(self->feat_multi_, nullptr, 0); }
    } feature_files_guard{this};
    // HIP-event time of the dominant kernel of the last match / filter call (the slowest device of a multi-device deal)
    auto kernel_ms = [&](bool filter) -> double {
        double ms = 0;
        const int n = ctx_ ? 1 : r3dm_multi_num_devices(multi_);
        for (int k = 0; k < n; ++k) {
            r3dm_stats st{};
            if (r3dm_get_stats(ctx_ ? ctx_ : r3dm_multi_ctx(multi_, k), &st) == R3DM_OK) ms = std::max(ms, filter ? st.ms_filter_kernels : st.ms_match_kernels + st.ms_ann_search + st.ms_ann_build);
        }
        return ms;
    };
    // one device or the multi-device deal: the same calls either way
    auto last_error = [&]() -> std::string { return ctx_ ? r3dm_last_error(ctx_) : r3dm_multi_last_error(multi_); };
    auto clear_images = [&]() { return ctx_ ? r3dm_clear_images(ctx_) : r3dm_multi_clear_images(multi_); };
    auto set_image = [&](uint32_t id, uint32_t w, uint32_t h, const void* d, uint32_t n, const float* xy) {
        return ctx_ ? r3dm_set_image(ctx_, id, w, h, d, n, dim_, dtype_, xy) : r3dm_multi_set_image(multi_, id, w, h, d, n, dim_, dtype_, xy); };
    auto set_intrinsics = [&](uint32_t id, const double* K) { return ctx_ ? r3dm_set_intrinsics(ctx_, id, K) : r3dm_multi_set_intrinsics(multi_, id, K); };
    auto match = [&](const std::vector<uint32_t>& p, float ratio, int squared, r3dm_graph** out) {
        return ctx_ ? r3dm_match_pairs(ctx_, p.data(), p.size() / 2, ratio, squared, out) : r3dm_multi_match_pairs(multi_, p.data(), p.size() / 2, ratio, squared, out); };
    auto match_kgraph = [&](const std::vector<uint32_t>& p, float ratio, const r3dm_kgraph_params* kpp, r3dm_graph** out) {
        return ctx_ ? r3dm_match_pairs_kgraph(ctx_, p.data(), p.size() / 2, ratio, kpp, out) : r3dm_multi_match_pairs_kgraph(multi_, p.data(), p.size() / 2, ratio, kpp, out); };
    auto match_hnsw = [&](const std::vector<uint32_t>& p, float ratio, const r3dm_hnsw_params* hpp, r3dm_graph** out) {
        return ctx_ ? r3dm_match_pairs_hnsw(ctx_, p.data(), p.size() / 2, ratio, hpp, out) : r3dm_multi_match_pairs_hnsw(multi_, p.data(), p.size() / 2, ratio, hpp, out); };
    auto filter_F = [&](const r3dm_graph* g, r3dm_graph** out) {
        return ctx_ ? r3dm_filter_F(ctx_, g, 4.0, 2048, seed_, R3DM_ERR_SYMMETRIC_EPIPOLAR, out, nullptr) : r3dm_multi_filter_F(multi_, g, 4.0, 2048, seed_, out, nullptr); };
    auto filter_E = [&](const r3dm_graph* g, r3dm_graph** out) {
        return ctx_ ? r3dm_filter_E(ctx_, g, 4.0, 2048, seed_, 50, 0.3f, out, nullptr) : r3dm_multi_filter_E(multi_, g, 4.0, 2048, seed_, 50, 0.3f, out, nullptr); };
    auto filter_H = [&](const r3dm_graph* g, r3dm_graph** out) {
        return ctx_ ? r3dm_filter_H(ctx_, g, 4.0, 2048, seed_, out, nullptr) : r3dm_multi_filter_H(multi_, g, 4.0, 2048, seed_, out, nullptr); };
    // dispatch of src/R3DComputeMatches.cpp:2035-2062: 4 = brute force and 9 = the new arm run the exhaustive matcher; every
    // approximate arm (0 FLANN kd-trees, 1..3 KGraph, 5 MRPT, 6..8 HNSW) runs the graph matcher with a preset of at least the
    // arm's recall (r3dm_ann_params_for_algorithm); anything else is refused.
    r3dm_kgraph_params kp;
    bool use_kgraph = r3dm_ann_params_for_algorithm(matchingAlgorithm, &kp) == R3DM_OK;
    if (matchingAlgorithm != kMatchingAlgorithmGPU && matchingAlgorithm != 4 && !use_kgraph) {
        errorMessage_ = "matchingAlgorithm " + std::to_string(matchingAlgorithm) + " is not served by the GPU path (0..9 are)";
        return false;
    }
    const std::string dir = paths.relativeMatchesPath_;
    const size_t row_bytes = dtype_ == R3DM_F32 ? (size_t)dim_ * 4 : (size_t)dim_;

    // ---- R3DFeaturesThread::extractFeaturesAndDescriptors(vec_fileNames, sOutDir, params) (:1994-1995)
    if (clear_images() != R3DM_OK) { errorMessage_ = last_error(); return false; }
    registered_.assign(views_.size(), 0); registered_n_.assign(views_.size(), 0);
    {
        const double t0 = wall_ms();
        if (!runFeaturesStage(params, dir)) return false;
        phases_.features = wall_ms() - t0;
    }
    const double t_load = wall_ms();

    // ---- Regions_Provider::load + Features_Provider::load (src/R3DComputeMatches.cpp:2040,2094-2095): the views the features stage
    // computed in this call are registered already (features_sink); the files of the others are read here
    // files are read and parsed by all host threads, 64 views at a time; registration (device copies) stays in view order
    struct Loaded { std::vector<float> xy; std::vector<unsigned char> desc; uint64_t n = 0; bool ok = false; };
    std::vector<Loaded> chunk;
    std::vector<char> batch_done;
    for (size_t vi = 0; vi < views_.size(); ++vi) {
        if (vi % 64 == 0) {
            const size_t cn = std::min<size_t>(64, views_.size() - vi);
            chunk.assign(cn, Loaded());
#pragma omp parallel for schedule(dynamic) num_threads(r3dm_host_threads(64))
            for (long k = 0; k < (long)cn; ++k) {
                const View& u = views_[vi + (size_t)k];
                Loaded& L = chunk[(size_t)k];
                if (registered_[vi + (size_t)k]) { L.ok = true; continue; }
                // nothing may leave an OpenMP region by exception (std::terminate): a failed allocation is a failed load
                try { L.ok = load_feat(dir + "/" + u.basename + ".feat", L.xy) && load_desc(dir + "/" + u.basename + ".desc", row_bytes, L.desc, L.n); }
                catch (...) { L.ok = false; }
            }
            // one device: the chunk's views go to the matcher in ONE call (r3dm_set_images: helper threads fill the page-locked ring beside
            // the DMAs) -- up to the first view the per-view pass below will refuse, so that its error is the one reported
            batch_done.assign(cn, 0);
            if (ctx_) {
                std::vector<r3dm_view_desc> vd;
                std::vector<size_t> which;
                for (size_t k = 0; k < cn; ++k) {
                    const Loaded& L = chunk[k];
                    if (!L.ok) break;
                    if (registered_[vi + k]) continue;
                    if (L.xy.size() != 2 * L.n) break;
                    const View& u = views_[vi + k];
                    vd.push_back(r3dm_view_desc{u.id_view, u.ui_width, u.ui_height, (uint32_t)L.n, dim_, (int32_t)dtype_, L.desc.data(), L.xy.data()});
                    which.push_back(k);
                }
                if (!vd.empty()) {
                    const int rcb = r3dm_set_images(ctx_, vd.data(), (uint32_t)vd.size());
                    if (rcb != R3DM_OK) { errorMessage_ = last_error(); return false; }
                    for (size_t k : which) batch_done[k] = 1;
                }
            }
        }
        const View& v = views_[vi];
        Loaded& L = chunk[vi % 64];
        std::vector<float>& xy = L.xy;
        std::vector<unsigned char>& desc = L.desc;
        const uint64_t n = L.n;
        if (!L.ok) {
            errorMessage_ = "Invalid features: " + v.basename;       // reference: MLOG "Invalid features." + return false (:2096-2097)
            return false;
        }
        if (registered_[vi]) statistics_.numberOfKeypoints_.push_back((int)registered_n_[vi]);
        else {
            if (xy.size() != 2 * n) { errorMessage_ = "feature/descriptor count mismatch: " + v.basename; return false; }
            statistics_.numberOfKeypoints_.push_back((int)n);
            if (!batch_done[vi % 64]) {
                const int rc = set_image(v.id_view, v.ui_width, v.ui_height, desc.data(), (uint32_t)n, xy.data());
                if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
            }
        }
        if (v.focal_px > 0.0) {
            const double K[9] = {v.focal_px, 0.0, v.ppx, 0.0, v.focal_px, v.ppy, 0.0, 0.0, 1.0};      // Pinhole_Intrinsic::K()
            if (set_intrinsics(v.id_view, K) != R3DM_OK) { errorMessage_ = last_error(); return false; }
        }
    }

    // ---- exhaustivePairs(#views) (:2042): all (I, J) with I < J, in view-id order
    std::vector<uint32_t> ids;
    for (const View& v : views_) ids.push_back(v.id_view);
    std::sort(ids.begin(), ids.end());
    std::vector<uint32_t> pairs;
    for (size_t a = 0; a < ids.size(); ++a)
        for (size_t b = a + 1; b < ids.size(); ++b) { pairs.push_back(ids[a]); pairs.push_back(ids[b]); }

    phases_.load = wall_ms() - t_load;
    if (progress_) progress_(0.7f, "Find putative matches", progress_user_);
    double t_phase = wall_ms();
    // ---- photometric matching (:2048) + Save(matches.putative.txt) (:2064)
    r3dm_graph* putative = nullptr;
    const int squared = dtype_ == R3DM_BIN ? 0 : 1;        // RegionsMatcherT squared flag: true for L2 metrics
    int rc;
    // an approximate arm whose job the exhaustive matcher does at least as fast -- and exactly -- is served by it (setApproximateArmsPolicy)
    if (use_kgraph && arms_policy_ == kArmsFastest && r3dm_exhaustive_is_faster(ctx_ ? ctx_ : r3dm_multi_ctx(multi_, 0)) == 1) use_kgraph = false;
    last_exhaustive_ = !use_kgraph;
    // arms 6 / 7 / 8 taken literally (kArmsAsRequested) run hnsw_match itself: hnswlib's search on a batch-built HNSW index
    // (r3dm_match_pairs_hnsw), for the descriptor lengths hnswlib's SIMD16 distance serves; other lengths keep the graph matcher
    const bool use_hnsw = use_kgraph && arms_policy_ == kArmsAsRequested && matchingAlgorithm >= 6 && matchingAlgorithm <= 8 &&
                          dtype_ != R3DM_BIN && (dim_ == 64 || dim_ == 128 || dim_ == 144 || dim_ == 256);
    last_hnsw_ = use_hnsw;
    // arm 5 taken literally runs mrpt_match itself: random projection trees on the device (r3dm_match_pairs_mrpt), the reference's
    // parameters (src/R3DComputeMatches.cpp:453-456); the index takes float rows of any length that is a multiple of 4
    const bool use_mrpt = use_kgraph && arms_policy_ == kArmsAsRequested && matchingAlgorithm == 5 && dtype_ != R3DM_BIN && (dim_ & 3u) == 0 && dim_ <= 512;
    last_mrpt_ = use_mrpt;
    if (use_mrpt) {
        r3dm_mrpt_params mp;
        (void)r3dm_mrpt_preset(&mp);
        rc = ctx_ ? r3dm_match_pairs_mrpt(ctx_, pairs.data(), pairs.size() / 2, params.distRatio_, &mp, &putative)
                  : r3dm_multi_match_pairs_mrpt(multi_, pairs.data(), pairs.size() / 2, params.distRatio_, &mp, &putative);
    } else if (use_hnsw) {
        r3dm_hnsw_params hp;
        (void)r3dm_hnsw_preset(matchingAlgorithm - 6, &hp);
        rc = match_hnsw(pairs, params.distRatio_, &hp, &putative);
    } else if (use_kgraph) {
        rc = match_kgraph(pairs, params.distRatio_, &kp, &putative);
    } else {
        rc = match(pairs, params.distRatio_, squared, &putative);
    }
    if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
    phases_.match = wall_ms() - t_phase; phases_.match_kernels = kernel_ms(false);
#ifdef R3DM_DEVTOOLS
    { r3dm_stats st{}; if (ctx_ && r3dm_get_stats(ctx_, &st) == R3DM_OK) fprintf(stderr, "facade match phase %.2f ms, r3dm_match_pairs inside %.2f ms\n", phases_.match, st.ms_wall_match); }
#endif
    {
        r3dm_stats st{};
        if (ctx_ && r3dm_get_stats(ctx_, &st) == R3DM_OK) phases_.match_post = st.ms_wall_match_post;
    }
    t_phase = wall_ms();
    const std::string put_path = paths.matchesPutitativeFilename_.empty() ? dir + "/matches.putative.txt" : paths.matchesPutitativeFilename_;
    // the match files are written behind the filters (two host threads per graph); failures are reported at the end -- the reference
    // returns EXIT_FAILURE (== true) from a bool function there (:2069), a real failure is reported instead
    MatchFileWriter writer;
    writer.nice_value = background_nice_;
    writer.own(putative);
    writer.save(putative, put_path, with_ext(put_path, ".bin"));
    // statistics_.putativeMatches_ (the PairWiseMatches map the reference keeps, :2040-2069) is filled beside the filters as well: a
    // million matches into per-pair vectors is host work nothing on the device waits for
    struct MapJob {
        std::thread th;
        const r3dm_graph* g = nullptr; PairWiseMatches* out = nullptr;
        std::atomic<bool> failed{false};
        void start(const r3dm_graph* g_, PairWiseMatches* out_, int nv)
        {
            g = g_; out = out_;
            th = std::thread([this, nv]() { stand_back(nv); try { graph_to_map(g, *out); } catch (...) { failed.store(true); } });
        }
        // a map the background thread could not build (out of memory) is built again HERE, where a second failure reaches the caller
        // as the exception it is -- never an empty map behind a `true` from computeMatches
        void join() { if (th.joinable()) th.join(); if (failed.exchange(false)) { out->clear(); graph_to_map(g, *out); } }
        ~MapJob() { if (th.joinable()) th.join(); }
    };
    MapJob put_map;
    try { put_map.start(putative, &statistics_.putativeMatches_, background_nice_); } catch (...) { graph_to_map(putative, statistics_.putativeMatches_); }
    if (svgOutput) { put_map.join(); write_adjacency_svg(dir + "/PutativeAdjacencyMatrix.svg", views_.size(), statistics_.putativeMatches_); }   // :2074
    phases_.files += wall_ms() - t_phase;

    // ---- the three geometric filters side by side on one device (r3dm_filter_FEH): they only read the putative graph; the files
    //      and maps follow in the reference's order.  (A device list deals each filter's pairs to the devices instead.)
    const int which = (params.computeFundalmentalMatrix_ ? 1 : 0) | (params.computeEssentialMatrix_ ? 2 : 0) | (params.computeHomographyMatrix_ ? 4 : 0);
    const bool side_by_side = ctx_ && (which & (which - 1)) != 0;
    const double t_filters = wall_ms();
    if (side_by_side) {
        if (progress_) progress_(0.8f, "Calculate fundamental matrix", progress_user_);
        r3dm_graph *gF = nullptr, *gE = nullptr, *gH = nullptr;
        double msk[3] = {0, 0, 0}, msw[3] = {0, 0, 0};
        rc = r3dm_filter_FEH(ctx_, putative, 4.0, 2048, seed_, which, 50, 0.3f, &gF, &gE, &gH, msk, msw);
        if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
        phases_.filter_F = msw[0]; phases_.filter_E = msw[1]; phases_.filter_H = msw[2];
        phases_.F_kernels = msk[0]; phases_.E_kernels = msk[1]; phases_.H_kernels = msk[2];
        phases_.filters_wall = wall_ms() - t_filters;
        t_phase = wall_ms();
        struct Out { r3dm_graph* g; PairWiseMatches* map; const std::string* named; const char* def; float frac; const char* msg; };
        const Out outs[3] = {{gF, &statistics_.fundamentalMatches_, &paths.matchesFFilename_, "/matches.f.txt", 0.9f, "Calculate essential matrix"},
                             {gE, &statistics_.essentialMatches_, &paths.matchesEFilename_, "/matches.e.txt", 0.95f, "Calculate homography matrix"},
                             {gH, &statistics_.homographyMatches_, &paths.matchesHFilename_, "/matches.h.txt", 1.0f, nullptr}};
        MapJob maps[3];
        for (int k = 0; k < 3; ++k) {
            const Out& o = outs[k];
            if (!o.g) continue;
            try { maps[k].start(o.g, o.map, background_nice_); } catch (...) { graph_to_map(o.g, *o.map); }      // the three maps side by side
            const std::string path = o.named->empty() ? dir + o.def : *o.named;
            writer.own(o.g);
            writer.save(o.g, path, with_ext(path, ".bin"));
            if (progress_ && o.msg) progress_(o.frac, o.msg, progress_user_);
        }
        for (MapJob& mj : maps) mj.join();
        put_map.join();
        phases_.files += wall_ms() - t_phase;
    } else {
    // ---- geometric filtering, fundamental matrix (:2113-2120): AC-RANSAC, 4.0 px upper bound, 2048 iterations
    if (params.computeFundalmentalMatrix_) {
        if (progress_) progress_(0.8f, "Calculate fundamental matrix", progress_user_);
        r3dm_graph* geo = nullptr;
        t_phase = wall_ms();
        rc = filter_F(putative, &geo);
        if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
        phases_.filter_F = wall_ms() - t_phase; phases_.F_kernels = kernel_ms(true);
        t_phase = wall_ms();
        graph_to_map(geo, statistics_.fundamentalMatches_);
        const std::string f_path = paths.matchesFFilename_.empty() ? dir + "/matches.f.txt" : paths.matchesFFilename_;
        writer.own(geo);
        writer.save(geo, f_path, with_ext(f_path, ".bin"));
        phases_.files += wall_ms() - t_phase;
    }
    // ---- essential-matrix filter (:2130-2204): 5-point solver on K^-1 x, then the overlap rule (>= 50 matches and
    //      >= 30 % of the putative matches, :2175-2192); matches.e.txt feeds the global SfM engine
    if (params.computeEssentialMatrix_) {
        if (progress_) progress_(0.9f, "Calculate essential matrix", progress_user_);
        r3dm_graph* geo = nullptr;
        t_phase = wall_ms();
        rc = filter_E(putative, &geo);
        if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
        phases_.filter_E = wall_ms() - t_phase; phases_.E_kernels = kernel_ms(true);
        t_phase = wall_ms();
        graph_to_map(geo, statistics_.essentialMatches_);
        const std::string e_path = paths.matchesEFilename_.empty() ? dir + "/matches.e.txt" : paths.matchesEFilename_;
        writer.own(geo);
        writer.save(geo, e_path, with_ext(e_path, ".bin"));
        phases_.files += wall_ms() - t_phase;
    }
    // ---- homography filter (:2216-2233): same skeleton, 4-point solver; matches.h.txt
    if (params.computeHomographyMatrix_) {
        if (progress_) progress_(0.95f, "Calculate homography matrix", progress_user_);
        r3dm_graph* geo = nullptr;
        t_phase = wall_ms();
        rc = filter_H(putative, &geo);
        if (rc != R3DM_OK) { errorMessage_ = last_error(); return false; }
        phases_.filter_H = wall_ms() - t_phase; phases_.H_kernels = kernel_ms(true);
        t_phase = wall_ms();
        graph_to_map(geo, statistics_.homographyMatches_);
        const std::string h_path = paths.matchesHFilename_.empty() ? dir + "/matches.h.txt" : paths.matchesHFilename_;
        writer.own(geo);
        writer.save(geo, h_path, with_ext(h_path, ".bin"));
        phases_.files += wall_ms() - t_phase;
    }
    phases_.filters_wall = wall_ms() - t_filters;
    }
    // GeometricAdjacencyMatrix.svg (:2238): the reference draws whichever filter ran last (H, else E, else F)
    if (svgOutput) {
        const PairWiseMatches& last = params.computeHomographyMatrix_ ? statistics_.homographyMatches_
                                    : (params.computeEssentialMatrix_ ? statistics_.essentialMatches_ : statistics_.fundamentalMatches_);
        write_adjacency_svg(dir + "/GeometricAdjacencyMatrix.svg", views_.size(), last);
    }
    {
        const double t_w = wall_ms();
        const std::string bad = writer.finish();
        char ferr[512] = {0};
        const int frc = feat_multi_ ? r3dm_multi_features_files_wait(feat_multi_, ferr, sizeof(ferr)) : R3DM_OK;
        phases_.files += wall_ms() - t_w;
        if (!bad.empty()) { errorMessage_ = "Cannot save computed matches in: " + bad; return false; }
        if (frc != R3DM_OK) { errorMessage_ = std::string("features stage failed: ") + ferr; return false; }
    }
    phases_.total = wall_ms() - t_begin;
    return true;
}

}  // namespace r3d_amd

struct r3dm_stage { r3d_amd::R3DComputeMatches* stage = nullptr; };

extern "C" int r3dm_stage_create(const int* device_ids, int n_devices, r3dm_stage** out)
{
    if (!out || !device_ids || n_devices < 1) return R3DM_ERR_INVALID;
    *out = nullptr;
    try {
        auto* s = new r3dm_stage();
        s->stage = new r3d_amd::R3DComputeMatches(std::vector<int>(device_ids, device_ids + n_devices));
        if (!s->stage->errorMessage().empty()) { delete s->stage; delete s; return R3DM_ERR_NO_DEVICE; }
        *out = s;
        return R3DM_OK;
    } catch (const std::bad_alloc&) { return R3DM_ERR_NOMEM; }
    catch (...) { return R3DM_ERR_INVALID; }
}

extern "C" void r3dm_stage_destroy(r3dm_stage* s)
{
    if (!s) return;
    delete s->stage;
    delete s;
}

extern "C" int r3dm_stage_run(r3dm_stage* sp, const char* matches_dir, const r3dm_view_image* views, uint32_t n_views,
                              float threshold, float dist_ratio, int matching_algorithm, int compute_F, int compute_E, int compute_H,
                              uint64_t seed, int features_batches_in_flight, int features_images_per_batch, uint32_t flags,
                              r3dm_stage_report* report, char* err, size_t err_cap)
{
    if (!sp || !sp->stage || !matches_dir || (n_views && !views)) return R3DM_ERR_INVALID;
    if (err && err_cap) err[0] = 0;
    try {
        r3d_amd::R3DComputeMatches& stage = *sp->stage;
        stage.clearViews();
        std::vector<r3d_amd::View> vs(n_views);
        for (uint32_t k = 0; k < n_views; ++k) {
            r3d_amd::View& v = vs[k];
            v.id_view = views[k].id; v.ui_width = views[k].width; v.ui_height = views[k].height; v.basename = views[k].basename ? views[k].basename : "";
            v.focal_px = views[k].focal_px; v.ppx = views[k].ppx; v.ppy = views[k].ppy;
            v.bgr8 = views[k].bgr8; v.gray = views[k].gray;
        }
        stage.addViews(vs);
        stage.setSeed(seed);
        if (features_batches_in_flight > 0 && features_images_per_batch > 0) stage.setFeaturesConcurrency(features_batches_in_flight, features_images_per_batch);
        stage.setApproximateArmsPolicy((flags & R3DM_STAGE_ARMS_AS_REQUESTED) ? r3d_amd::R3DComputeMatches::kArmsAsRequested : r3d_amd::R3DComputeMatches::kArmsFastest);
        stage.setBackgroundThreadsNice((flags & R3DM_STAGE_BACKGROUND_NICE) ? 10 : 0);
        stage.setExactFastPaths((flags & R3DM_STAGE_F32_TILES) == 0);      // (R3DM_STAGE_SPLIT_MFMA / _INTEGER_MFMA: implied since round 3)
        r3d_amd::R3DFParams params;
        params.keypointDetectorList_ = {"Fast-AKAZE"};
        params.threshold_ = threshold;
        params.distRatio_ = dist_ratio;
        params.computeFundalmentalMatrix_ = compute_F != 0;
        params.computeEssentialMatrix_ = compute_E != 0;
        params.computeHomographyMatrix_ = compute_H != 0;
        r3d_amd::R3DProjectPaths paths;
        paths.relativeMatchesPath_ = matches_dir;
        const bool ok = stage.computeMatches(params, false, paths, 1, matching_algorithm);
        if (err && err_cap) { strncpy(err, stage.errorMessage().c_str(), err_cap - 1); err[err_cap - 1] = 0; }
        if (report) {
            const auto& P = stage.getPhaseTimes();
            const auto& S = stage.getStatistics();
            *report = r3dm_stage_report{};
            report->ms_features = P.features; report->ms_load = P.load; report->ms_match = P.match; report->ms_filter_F = P.filter_F;
            report->ms_filter_E = P.filter_E; report->ms_filter_H = P.filter_H; report->ms_files = P.files; report->ms_total = P.total;
            report->ms_filters_wall = P.filters_wall; report->ms_match_post = P.match_post;
            report->ms_match_kernels = P.match_kernels; report->ms_F_kernels = P.F_kernels; report->ms_E_kernels = P.E_kernels; report->ms_H_kernels = P.H_kernels;
            report->images_extracted = P.images_extracted; report->features = P.features_totals;
            report->match_was_exhaustive = stage.lastMatchWasExhaustive() ? 1 : 0;
            for (int n : S.numberOfKeypoints_) report->n_keypoints += (uint64_t)n;
            auto count = [](const r3d_amd::PairWiseMatches& m, uint64_t& pairs, uint64_t& matches) { pairs = m.size(); matches = 0; for (const auto& kv : m) matches += kv.second.size(); };
            count(S.putativeMatches_, report->n_putative_pairs, report->n_putative_matches);
            count(S.fundamentalMatches_, report->n_F_pairs, report->n_F_matches);
            count(S.essentialMatches_, report->n_E_pairs, report->n_E_matches);
            count(S.homographyMatches_, report->n_H_pairs, report->n_H_matches);
        }
        return ok ? R3DM_OK : R3DM_ERR_IO;
    } catch (const std::bad_alloc&) { return R3DM_ERR_NOMEM; }          // nothing crosses the C boundary
    catch (...) { return R3DM_ERR_INVALID; }
}

extern "C" int r3dm_compute_matches_stage(const int* device_ids, int n_devices, const char* matches_dir, const r3dm_view_image* views, uint32_t n_views,
                                          float threshold, float dist_ratio, int matching_algorithm, int compute_F, int compute_E, int compute_H,
                                          uint64_t seed, int features_batches_in_flight, int features_images_per_batch, uint32_t flags,
                                          r3dm_stage_report* report, char* err, size_t err_cap)
{
    if (err && err_cap) err[0] = 0;
    r3dm_stage* s = nullptr;
    int rc = r3dm_stage_create(device_ids, n_devices, &s);
    if (rc != R3DM_OK) { if (err && err_cap) snprintf(err, err_cap, "r3dm_create failed (%d): no gfx950 GPU", rc); return rc; }
    rc = r3dm_stage_run(s, matches_dir, views, n_views, threshold, dist_ratio, matching_algorithm, compute_F, compute_E, compute_H, seed,
                        features_batches_in_flight, features_images_per_batch, flags, report, err, err_cap);
    r3dm_stage_destroy(s);
    return rc;
}

extern "C" int r3dm_compute_matches_dir(int device_id, const char* matches_dir, const r3dm_view* views, uint32_t n_views,
                                        r3dm_dtype dtype, uint32_t dim, float dist_ratio, int compute_F, uint64_t seed,
                                        uint64_t* n_putative_pairs, uint64_t* n_geometric_pairs, char* err, size_t err_cap)
{
    if (!matches_dir || (n_views && !views)) return R3DM_ERR_INVALID;
    try {
    r3d_amd::R3DComputeMatches stage(device_id);
    std::vector<r3d_amd::View> vs;
    for (uint32_t k = 0; k < n_views; ++k) vs.push_back({views[k].id, views[k].width, views[k].height, views[k].basename});
    stage.addViews(vs);
    stage.setRegionsType(dtype, dim);
    stage.setSeed(seed);
    r3d_amd::R3DFParams params;
    params.distRatio_ = dist_ratio;
    params.computeFundalmentalMatrix_ = compute_F != 0;
    params.computeEssentialMatrix_ = false;
    params.computeHomographyMatrix_ = false;
    r3d_amd::R3DProjectPaths paths;
    paths.relativeMatchesPath_ = matches_dir;
    const bool ok = stage.computeMatches(params, false, paths, 1, r3d_amd::R3DComputeMatches::kMatchingAlgorithmGPU);
    if (n_putative_pairs) *n_putative_pairs = stage.getStatistics().putativeMatches_.size();
    if (n_geometric_pairs) *n_geometric_pairs = stage.getStatistics().fundamentalMatches_.size();
    if (err && err_cap) { strncpy(err, stage.errorMessage().c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return ok ? R3DM_OK : R3DM_ERR_IO;
    } catch (const std::bad_alloc&) { return R3DM_ERR_NOMEM; }          // nothing crosses the C boundary
    catch (...) { return R3DM_ERR_INVALID; }
}
