// api_features.cpp -- part of the host side of libr3dm.so: the C ABI declared in include/r3dm.h (see r3dm_ctx.hpp for the file map).
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 and src/Regard3DFeatures.cpp -- with every arithmetic stage running as
// HIP kernels on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "r3dm_ctx.hpp"

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
    for (int k = B_IMG; k <= B_LT2; ++k) if (k != B_LYY) R3DM_HIP(c, buf(k).ensure(n0 * 4));      // (Lyy is folded into the determinant kernel)
    R3DM_HIP(c, buf(B_SMALL).ensure(4096 * 4));
    for (int i = 0; i < nl; ++i)
        for (int q = 0; q < 4; ++q) R3DM_HIP(c, buf(B_LEVEL0 + 4 * i + q).ensure((size_t)lv[i].w * lv[i].h * 4));
    auto Lt = [&](int i) { return buf(B_LEVEL0 + 4 * i).as<float>(); };
    auto Lx = [&](int i) { return buf(B_LEVEL0 + 4 * i + 1).as<float>(); };
    auto Ly = [&](int i) { return buf(B_LEVEL0 + 4 * i + 2).as<float>(); };
    auto Ldet = [&](int i) { return buf(B_LEVEL0 + 4 * i + 3).as<float>(); };
    float* img = buf(B_IMG).as<float>();
    float* smooth = buf(B_SMOOTH).as<float>();
    float* lxx = buf(B_LXX).as<float>(); float* lxy = buf(B_LXY).as<float>();
    float* tmp = buf(B_TMP).as<float>(); float* tmp2 = buf(B_TMP2).as<float>();
    float* wx = buf(B_WX).as<float>(); float* wy = buf(B_WY).as<float>();
    float* flow = buf(B_FLOW).as<float>(); float* lt2 = buf(B_LT2).as<float>();
    uint32_t* small = buf(B_SMALL).as<uint32_t>();

    R3DM_HIP(c, hipMemcpyAsync(img, image, n0 * 4, hipMemcpyDefault, st));
    const AkTaps taps_off = ak_taps(1.6f), taps_one = ak_taps(1.0f);

    // INTER_AREA tables of the octave transitions whose size is not an exact halving (they depend on the image size only):
    // built and uploaded before the launch sequence so that the sequence itself never touches the host
    struct HalfTabs { const AkAreaTab* xt = nullptr; const AkAreaTab* yt = nullptr; const int* xb = nullptr; const int* yb = nullptr; };
    std::vector<HalfTabs> half_tabs(nl);
    DevBuf& tab_buf = buf(B_LEVEL0 + 4 * nl + 3);
    {
        std::vector<unsigned char> blob;
        std::vector<size_t> offs(4 * (size_t)nl, (size_t)-1);
        auto put = [&](const void* p, size_t bytes) { const size_t at = (blob.size() + 15) / 16 * 16; blob.resize(at + bytes); memcpy(blob.data() + at, p, bytes); return at; };
        for (int i = 1; i < nl; ++i) {
            if (lv[i].octave == lv[i - 1].octave) continue;
            const int sw = lv[i - 1].w, sh = lv[i - 1].h, lw = lv[i].w, lh = lv[i].h;
            if (lw * 2 == sw && lh * 2 == sh) continue;
            std::vector<AkAreaTab> tx, ty; std::vector<int> bx, by;
            ak_area_tab(sw, lw, tx, bx); ak_area_tab(sh, lh, ty, by);
            offs[4 * i] = put(tx.data(), tx.size() * sizeof(AkAreaTab)); offs[4 * i + 1] = put(ty.data(), ty.size() * sizeof(AkAreaTab));
            offs[4 * i + 2] = put(bx.data(), bx.size() * 4); offs[4 * i + 3] = put(by.data(), by.size() * 4);
        }
        if (!blob.empty()) {
            R3DM_HIP(c, tab_buf.ensure(blob.size() + 64));
            R3DM_HIP(c, hipMemcpyAsync(tab_buf.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
            R3DM_HIP(c, hipStreamSynchronize(st));                 // `blob` leaves scope
            const unsigned char* base = tab_buf.as<unsigned char>();
            for (int i = 1; i < nl; ++i)
                if (offs[4 * i] != (size_t)-1)
                    half_tabs[i] = {(const AkAreaTab*)(base + offs[4 * i]), (const AkAreaTab*)(base + offs[4 * i + 1]),
                                    (const int*)(base + offs[4 * i + 2]), (const int*)(base + offs[4 * i + 3])};
        }
    }
    uint32_t* hmax_bits = small;                  // [0]      maximum of the gradient modulus (float bits)
    uint32_t* hist = small + 16;                  // [16..)   300-bin histogram
    float* inv_k2 = reinterpret_cast<float*>(small + 1024);     // [1024 + o] 1 / k^2 of octave o (ak_kcontrast_kernel)

    // Compute_Determinant_Hessian_Response_Single (AKAZEFeatures.cpp:389-410)
    auto hessian = [&](int i) -> hipError_t {
        const int lw = lv[i].w, lh = lv[i].h, s = lv[i].sigma_size;
        hipError_t e;
        // three launches: smooth -> (Lx, Ly);  Lx -> (Lxx, Lxy);  Ly -> Lyy, folded into the determinant
        if ((e = ak_scaled_deriv_xy(st, smooth, Lx(i), Ly(i), lw, lh, s)) != hipSuccess) return e;
        if ((e = ak_scaled_deriv_xy(st, Lx(i), lxx, lxy, lw, lh, s)) != hipSuccess) return e;
        return ak_scaled_deriv_det(st, Ly(i), lxx, lxy, Ldet(i), lw, lh, s);
    };

    // ---- Create_Nonlinear_Scale_Space (:245-369), Compute_Base_Evolution_Level (:199-237): ~550 launches for a 12 Mpx image,
    // none of which needs the host -- the k-contrast (compute_k_percentileV2: maximum, 300-bin histogram, percentile scan) stays
    // on the device.  The sequence depends on the image SIZE only, so it CAN be captured into a hipGraph and replayed (below).
    auto scale_space = [&]() -> hipError_t {
        hipError_t e;
#define AK_TRY(call) do { if ((e = (call)) != hipSuccess) return e; } while (0)
        AK_TRY(ak_gaussian(st, img, tmp, smooth, w, h, taps_off));
        AK_TRY(hessian(0));
        AK_TRY(hipMemsetAsync(small, 0, 4096 * 4, st));
        const int nbins = 300;
        if (nl > 1) {
            AK_TRY(ak_gaussian(st, img, tmp, flow, w, h, taps_one));
            AK_TRY(ak_scharr(st, flow, tmp, tmp2, wx, wy, w, h));
            AK_TRY(ak_modg_max(st, wx, wy, w, h, hmax_bits));
            AK_TRY(ak_modg_hist(st, wx, wy, w, h, hmax_bits, nbins, hist));
        }
        AK_TRY(ak_kcontrast(st, hmax_bits, hist, nbins, (uint32_t)((size_t)(w - 2) * (h - 2)), nl > 1 ? 1 : 0, inv_k2));
        AK_TRY(hipMemcpyAsync(Lt(0), smooth, n0 * 4, hipMemcpyDeviceToDevice, st));
        for (int i = 1; i < nl; ++i) {
            const int lw = lv[i].w, lh = lv[i].h;
            const size_t n = (size_t)lw * lh;
            const std::vector<float> tau = ak_fed_tau(lv[i].etime - lv[i - 1].etime);
            const float* start = nullptr;
            if (lv[i].octave > lv[i - 1].octave) {
                // the FED steps ping-pong between Lt(i) and the work image and must END in Lt(i): the half-sampled start image goes
                // to whichever of the two the first step does not write
                float* half = (tau.size() % 2 == 1) ? lt2 : Lt(i);
                const HalfTabs& ht = half_tabs[i];
                AK_TRY(ak_halfsample(st, Lt(i - 1), half, lv[i - 1].w, lv[i - 1].h, ht.xt, ht.xb, ht.yt, ht.yb));
                start = half;
            } else {
                start = Lt(i - 1);                                  // same octave: the previous level IS the start image, no copy
            }
            if (tau.empty()) {                                      // (never for the reference's time steps) plain copy
                if (start != Lt(i)) AK_TRY(hipMemcpyAsync(Lt(i), start, n * 4, hipMemcpyDeviceToDevice, st));
                start = Lt(i);
            }
            AK_TRY(ak_gaussian(st, start, tmp, smooth, lw, lh, taps_one));
            AK_TRY(hessian(i));
            AK_TRY(ak_scharr_g2(st, smooth, flow, lw, lh, inv_k2 + lv[i].octave));      // kcontrast * 0.75^octave
            // Fast Explicit Diffusion: lt += lstep * 0.5 * tau_j; step k of n writes Lt(i) when n - k is even, else the work image
            const float* cur = start;
            for (size_t k = 1; k <= tau.size(); ++k) {
                float* out = ((tau.size() - k) % 2 == 0) ? Lt(i) : lt2;
                AK_TRY(ak_fed_step(st, cur, flow, out, lw, lh, tau[k - 1]));
                cur = out;
            }
        }
#undef AK_TRY
        return hipSuccess;
    };
    // Measured (profiles/r02_e_akaze_perf.txt): replaying the ~600-node graph costs MORE than issuing the launches -- 15.8 ms per
    // 12 Mpx image against 7.5 ms with plain stream launches (the runtime walks the nodes one by one at ~20 us each) -- so the
    // capture stays a developer option (R3DM_AK_GRAPH=1 in the developer build); the product issues the launches directly.
    static const bool use_graph = r3dm_dev_knob("R3DM_AK_GRAPH", 0) != 0;
    bool replayed = false;
    if (use_graph && !c->ak_graph_off) {
        if (c->ak_graph && (c->ak_graph_w != w || c->ak_graph_h != h)) { (void)hipGraphExecDestroy(c->ak_graph); c->ak_graph = nullptr; }
        if (!c->ak_graph) {
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (e == hipSuccess) {
                const hipError_t el = scale_space();
                e = hipStreamEndCapture(st, &g);
                if (e == hipSuccess) e = el;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&c->ak_graph, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess) { c->ak_graph = nullptr; c->ak_graph_off = true; (void)hipGetLastError(); }
            else { c->ak_graph_w = w; c->ak_graph_h = h; }
        }
        if (c->ak_graph) { R3DM_HIP(c, hipGraphLaunch(c->ak_graph, st)); replayed = true; }
    }
    if (!replayed) R3DM_HIP(c, scale_space());
    c->n_ak_graph_replays += replayed ? 1 : 0;

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
    int max_rows = 0;
    for (int i = 0; i < nl; ++i) max_rows = std::max(max_rows, lv[i].h - 2 * lv[i].border);
    R3DM_HIP(c, hipMemcpyAsync(d_levels, ld.data(), nl * sizeof(AkLevelDev), hipMemcpyHostToDevice, st));
    R3DM_HIP(c, ak_extrema(st, d_levels, nl, max_rows, threshold, 0));          // all levels in one launch
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
    R3DM_HIP(c, ak_extrema(st, d_levels, nl, max_rows, threshold, 1));
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
    // all levels' results are copied back behind the refinement kernel with ONE stream synchronisation
    std::vector<std::vector<float4>> all0(nl); std::vector<std::vector<float2>> all1(nl); std::vector<std::vector<uint32_t>> allv(nl);
    for (int i = 0; i < nl; ++i) {
        const uint32_t n = counts[4 * i + 1];
        if (!n) continue;
        all0[i].resize(n); all1[i].resize(n); allv[i].resize(n);
        R3DM_HIP(c, hipMemcpyAsync(all0[i].data(), ld[i].out0, n * 16, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipMemcpyAsync(all1[i].data(), ld[i].out1, n * 8, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipMemcpyAsync(allv[i].data(), ld[i].out_valid, n * 4, hipMemcpyDeviceToHost, st));
    }
    R3DM_HIP(c, hipStreamSynchronize(st));
    for (int i = 0; i < nl; ++i) {
        const uint32_t n = counts[4 * i + 1];
        if (!n) continue;
        const std::vector<float4>& o0 = all0[i]; const std::vector<float2>& o1 = all1[i]; const std::vector<uint32_t>& ov = allv[i];
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
    *n_out = n_kp;
    c->stats.ms_detect = now_ms() - t_call;
    return R3DM_OK;
}

static int r3dm_detect_akaze_impl(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                 float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out)
{
    return detect_akaze_impl(c, image, width, height, threshold, keypoints_out, responses_out, cap, n_out, nullptr);
}

extern "C" int r3dm_detect_akaze(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                 float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_detect_akaze_impl(c, image, width, height, threshold, keypoints_out, responses_out, cap, n_out); });
}

static int r3dm_detect_akaze_mldb_impl(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                      float* keypoints_out, unsigned char* descriptors_out, uint32_t cap, uint32_t* n_out)
{
    if (!descriptors_out && cap) return R3DM_ERR_INVALID;
    return detect_akaze_impl(c, image, width, height, threshold, keypoints_out, nullptr, cap, n_out, descriptors_out);
}

extern "C" int r3dm_detect_akaze_mldb(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                                      float* keypoints_out, unsigned char* descriptors_out, uint32_t cap, uint32_t* n_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_detect_akaze_mldb_impl(c, image, width, height, threshold, keypoints_out, descriptors_out, cap, n_out); });
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

static int r3dm_liop_describe_patches_impl(r3dm_ctx* c, const float* patches, uint32_t n, uint32_t side, float* desc_out,
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

extern "C" int r3dm_liop_describe_patches(r3dm_ctx* c, const float* patches, uint32_t n, uint32_t side, float* desc_out,
                                          uint32_t* n_resorted)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_liop_describe_patches_impl(c, patches, n, side, desc_out, n_resorted); });
}

// resident_image: the image is already on the device (the features stage: r3dm_detect_akaze has just uploaded it into its own
// buffer, which it only reads) -- then `image` is not copied a second time
static int r3dm_extract_liop_impl(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height,
                                 const float* keypoints, uint32_t n, float kp_size_factor, float* desc_out, float* patches_out,
                                 const float* resident_image = nullptr)
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
    if (!resident_image) R3DM_HIP(c, c->liop_img.ensure(img_bytes));
    R3DM_HIP(c, c->liop_M.ensure(M6.size() * 4));
    R3DM_HIP(c, c->liop_kern.ensure(64));
    R3DM_HIP(c, c->liop_in.ensure(patch_bytes));
    R3DM_HIP(c, c->liop_out.ensure(out_bytes));
    R3DM_HIP(c, c->liop_cnt.ensure(64));
    if (!resident_image) R3DM_HIP(c, hipMemcpyAsync(c->liop_img.p, image, img_bytes, hipMemcpyDefault, c->stream));
    const float* dev_image = resident_image ? resident_image : c->liop_img.as<float>();
    R3DM_HIP(c, hipMemcpyAsync(c->liop_M.p, M6.data(), M6.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_kern.p, kern, sizeof(kern), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, launch_liop_extract(c->stream, dev_image, (int)width, (int)height, c->liop_M.as<float>(),
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

extern "C" int r3dm_extract_liop(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height,
                                 const float* keypoints, uint32_t n, float kp_size_factor, float* desc_out, float* patches_out)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_extract_liop_impl(c, image, width, height, keypoints, n, kp_size_factor, desc_out, patches_out); });
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

static int r3dm_extract_features_to_files_impl(r3dm_ctx* c, const float* gray, uint32_t width, uint32_t height, float threshold,
                                              const char* feat_path, const char* desc_path, uint32_t* n_features)
{
    if (!c || !gray || !feat_path || !desc_path) return R3DM_ERR_INVALID;
    if (n_features) *n_features = 0;
    // "Test if descriptor and feature was already computed" (src/threads/R3DFeaturesThread.cpp:139-142): when BOTH files exist the
    // work item does nothing -- files left by a run with other parameters are reused, the reference wipes the matches directory
    // instead (src/threads/R3DComputeMatchesThread.cpp:84-86).  n_features then reports the row count of the existing .desc.
    {
        FILE* ff = fopen(feat_path, "rb");
        FILE* fd = ff ? fopen(desc_path, "rb") : nullptr;
        if (ff) fclose(ff);
        if (fd) {
            uint64_t cnt = 0;
            if (fread(&cnt, 8, 1, fd) != 1) cnt = 0;
            fclose(fd);
            if (n_features) *n_features = (uint32_t)cnt;
            return R3DM_OK;
        }
    }
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
        // the detector has just uploaded `gray` into its image buffer (ak_bufs[0], read-only for it): no second 4 w h-byte copy
        const float* resident = (c->ak_w == (int)width && c->ak_h == (int)height && !c->ak_bufs.empty()) ? c->ak_bufs[0].as<float>() : nullptr;
        rc = r3dm_extract_liop_impl(c, gray, width, height, kps.data(), n, 8.0f /* getKpSizeFactor("Fast-AKAZE"), :703-704 */, desc.data(), nullptr, resident);
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

extern "C" int r3dm_extract_features_to_files(r3dm_ctx* c, const float* gray, uint32_t width, uint32_t height, float threshold,
                                              const char* feat_path, const char* desc_path, uint32_t* n_features)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_extract_features_to_files_impl(c, gray, width, height, threshold, feat_path, desc_path, n_features); });
}


// ------------------------------------------------------------------------------------------------
// the features stage over a whole image list: R3DFeaturesThread::extractFeaturesAndDescriptors
// (src/threads/R3DFeaturesThread.cpp:38-89) -- a pool of CPUs + 1 worker threads pulling images off a work list (:93-121), each
// running processWorkItem (:123-210).  The reference admits ONE image at a time into the A-KAZE scale space
// (initAKAZESemaphore(1), src/R3DComputeMatches.cpp:1847; src/Regard3DFeatures.cpp:71-125) to bound host memory; HBM does not
// need that: every context of an r3dm_multi (r3dm_multi_create with a device id repeated K times = K streams + work buffers on
// that device; or one per GPU) pulls images off the list from its own host thread, so K images are in flight at once.  An image whose .feat AND .desc both exist is skipped, exactly
// as processWorkItem does (:139-142: stale files of other parameters are reused; the reference wipes the matches directory
// instead, src/threads/R3DComputeMatchesThread.cpp:84-86); n_features then reports the row count of the existing .desc.
// ------------------------------------------------------------------------------------------------
#include <atomic>
#include <thread>

static bool file_exists(const char* p) { FILE* f = fopen(p, "rb"); if (!f) return false; fclose(f); return true; }

extern "C" {
int r3dm_multi_num_devices(const r3dm_multi* m);
r3dm_ctx* r3dm_multi_ctx(r3dm_multi* m, int k);
}

extern "C" int r3dm_multi_extract_features(r3dm_multi* m, uint32_t n_images, const float* const* grays, const uint32_t* widths,
                                           const uint32_t* heights, float threshold, const char* const* feat_paths,
                                           const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped,
                                           char* err, size_t err_cap)
{
    if (!m || (n_images && (!grays || !widths || !heights || !feat_paths || !desc_paths))) return R3DM_ERR_INVALID;
    if (err && err_cap) err[0] = 0;
    const uint32_t concurrency = (uint32_t)r3dm_multi_num_devices(m);
    if (concurrency == 0) return R3DM_ERR_INVALID;
    int rc_all = R3DM_OK;
    try {
        std::atomic<uint32_t> next{0};
        std::vector<int> rcs(concurrency, R3DM_OK);
        std::vector<std::string> errs(concurrency);
        auto worker = [&](uint32_t k) {
            r3dm_ctx* c = r3dm_multi_ctx(m, (int)k);
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= n_images || rcs[k] != R3DM_OK) return;
                if (skipped) skipped[i] = 0;
                if (file_exists(feat_paths[i]) && file_exists(desc_paths[i])) {           // processWorkItem: already computed
                    uint64_t cnt = 0;
                    if (FILE* f = fopen(desc_paths[i], "rb")) { if (fread(&cnt, 8, 1, f) != 1) cnt = 0; fclose(f); }
                    if (n_features) n_features[i] = (uint32_t)cnt;
                    if (skipped) skipped[i] = 1;
                    continue;
                }
                uint32_t n = 0;
                const int rc = r3dm_extract_features_to_files(c, grays[i], widths[i], heights[i], threshold, feat_paths[i], desc_paths[i], &n);
                if (n_features) n_features[i] = n;
                if (rc != R3DM_OK) { rcs[k] = rc; errs[k] = std::string("image ") + std::to_string(i) + ": " + r3dm_last_error(c); }
            }
        };
        std::vector<std::thread> th;
        for (uint32_t k = 1; k < concurrency; ++k) th.emplace_back(worker, k);
        worker(0);
        for (auto& t : th) t.join();
        for (uint32_t k = 0; k < concurrency; ++k)
            if (rcs[k] != R3DM_OK && rc_all == R3DM_OK) {
                rc_all = rcs[k];
                if (err && err_cap) { strncpy(err, errs[k].c_str(), err_cap - 1); err[err_cap - 1] = 0; }
            }
    } catch (...) { rc_all = R3DM_ERR_NOMEM; }
    return rc_all;
}
