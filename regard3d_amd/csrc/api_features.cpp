// api_features.cpp -- part of the host side of libr3dm.so: the C ABI declared in include/r3dm.h (see r3dm_ctx.hpp for the file map).
//
// Mirrors, for the compute-matches hot path only, what the reference does in
// /root/reference/src/R3DComputeMatches.cpp:2035-2129 and src/Regard3DFeatures.cpp -- with every arithmetic stage running as
// HIP kernels on one MI355X.  There is no CPU fallback in this file: when HIP fails, the call fails.
#include "fmt_g6.hpp"
#include "r3dm_ctx.hpp"

#include <charconv>

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

// ------------------------------------------------------------------------------------------------
// The detector over a BATCH of B same-size images.  One image alone cannot fill the chip below the first octave: its ~550
// dependent launches are a few microseconds each (profiles/r02_e_akaze_kernel_stats.txt), so a single image waits for launch
// latency, not for HBM.  The launch sequence depends on the image SIZE only, so the same ~550 launches serve B images
// (blockIdx.z = image), and nothing in the chain goes to the host: the k-contrast stays on the device, the candidate slots are
// laid out on the device from a capacity (ak_layout_kernel) instead of counts read back, list lengths are read by grid-stride
// kernels, and the survivors of all levels are compacted into one 32-byte record per keypoint (ak_compact_kernel).  The host sees
// the batch twice: the per-image counts, then the records.
// ------------------------------------------------------------------------------------------------
struct AkBatchOut {
    std::vector<AkLevelHost> lv;                       // evolution levels of this image size
    std::vector<std::vector<AkKpRec>> recs;            // per image: surviving keypoints in the reference's order (level, list)
};

// angle of a keypoint: getAngleV2(maxX, maxY) (fast-akaze utils.h:11-19: atan2f of the host libm, + 2 pi when negative)
static inline float ak_theta(const AkKpRec& r)
{
    float theta = atan2f(r.max_y, r.max_x);
    if (!(theta >= 0)) theta = theta + (float)(2.0f * 3.1415926535897932384626433832795);
    return theta;
}
// ... and the conversion of detectKeypoints (src/Regard3DFeatures.cpp:604-613): degrees, + 90, wrapped into [0, 360]
static inline float ak_angle_deg(float theta)
{
    float ang = theta;
    ang *= 180.0 / 3.1415926535897932384626433832795;
    ang += 90.0f;
    while (ang < 0) ang += 360.0f;
    while (ang > 360.0f) ang -= 360.0f;
    return ang;
}

// images: B pointers to height x width floats (host or device), or bgrs: B pointers to height x width x 3 bytes (cv::imread's
// BGR order; converted on the device exactly as processWorkItem does, src/threads/R3DFeaturesThread.cpp:163-191).
// Leaves the B gray images in ak_bufs[0] (B planes) for the LIOP patch extraction and the level images for MLDB.
static int ak_detect_batch(r3dm_ctx* c, uint32_t B, const float* const* images, const unsigned char* const* bgrs,
                           uint32_t width, uint32_t height, float threshold, AkBatchOut& out)
{
    if (!c || B == 0 || (!images && !bgrs)) return R3DM_ERR_INVALID;
    for (uint32_t b = 0; b < B; ++b) if (!(images ? (const void*)images[b] : (const void*)bgrs[b])) return R3DM_ERR_INVALID;
    out.recs.assign(B, std::vector<AkKpRec>());
    if (width < 3 || height < 3 || (uint64_t)width * height > (1ull << 30) || B > 4096) return R3DM_ERR_INVALID;
    R3DM_HIP(c, hipSetDevice(c->device));
    const double t_call = now_ms();
    const int w = (int)width, h = (int)height, iB = (int)B;
    out.lv = ak_levels(w, h);
    const std::vector<AkLevelHost>& lv = out.lv;
    const int nl = (int)lv.size();
    c->stats.n_detect_images = B; c->stats.ms_detect_kernels = 0.0; c->stats.detect_algorithmic_bytes = 0.0; c->stats.detect_compulsory_bytes = 0.0;
    if (nl == 0) { c->stats.ms_detect = now_ms() - t_call; return R3DM_OK; }     // image too small for a single evolution level
    hipStream_t st = c->stream;
    const size_t n0 = (size_t)w * h;

    // ---- buffers, each B planes: [0] image, [1..10] level-0 sized work images, then 4 per level (Lt, Lx, Ly, Ldet)
    enum { B_IMG = 0, B_SMOOTH, B_LXX, B_LXY, B_LYY, B_TMP, B_TMP2, B_WX, B_WY, B_FLOW, B_LT2, B_SMALL, B_LEVEL0 };
    if (c->ak_w != w || c->ak_h != h || c->ak_B < iB || c->ak_bufs.size() != (size_t)B_LEVEL0 + 4 * nl + 8) {
        for (DevBuf& b : c->ak_bufs) b.release();
        c->ak_bufs.assign((size_t)B_LEVEL0 + 4 * nl + 8, DevBuf());
        c->ak_w = w; c->ak_h = h; c->ak_B = iB;
    }
    const size_t PB = (size_t)c->ak_B;                            // planes per buffer (the largest batch of this size so far)
    auto buf = [&](int k) -> DevBuf& { return c->ak_bufs[k]; };
    for (int k = B_IMG; k <= B_LT2; ++k) if (k != B_LYY && k != B_LXX && k != B_LXY && k != B_WX && k != B_WY) R3DM_HIP(c, buf(k).ensure(PB * n0 * 4));      // (the second derivatives are folded into the determinant kernel)
    R3DM_HIP(c, buf(B_SMALL).ensure(PB * 4096 * 4));
    for (int i = 0; i < nl; ++i)
        for (int q = 0; q < 4; ++q) R3DM_HIP(c, buf(B_LEVEL0 + 4 * i + q).ensure(PB * (size_t)lv[i].w * lv[i].h * 4));
    auto Lt = [&](int i) { return buf(B_LEVEL0 + 4 * i).as<float>(); };
    auto Lx = [&](int i) { return buf(B_LEVEL0 + 4 * i + 1).as<float>(); };
    auto Ly = [&](int i) { return buf(B_LEVEL0 + 4 * i + 2).as<float>(); };
    auto Ldet = [&](int i) { return buf(B_LEVEL0 + 4 * i + 3).as<float>(); };
    float* img = buf(B_IMG).as<float>();
    float* smooth = buf(B_SMOOTH).as<float>();
    float* tmp = buf(B_TMP).as<float>(); float* tmp2 = buf(B_TMP2).as<float>();
    float* flow = buf(B_FLOW).as<float>(); float* lt2 = buf(B_LT2).as<float>();
    uint32_t* small = buf(B_SMALL).as<uint32_t>();

    if (images) {
        for (uint32_t b = 0; b < B; ++b) R3DM_HIP(c, hipMemcpyAsync(img + b * n0, images[b], n0 * 4, hipMemcpyDefault, st));
    } else {
        // 8-bit BGR -> float / 255 -> gray: the bytes are staged in the (not yet used) work image tmp2
        unsigned char* stage = reinterpret_cast<unsigned char*>(tmp2);
        for (uint32_t b = 0; b < B; ++b) R3DM_HIP(c, hipMemcpyAsync(stage + b * n0 * 4, bgrs[b], n0 * 3, hipMemcpyDefault, st));
        for (uint32_t b = 0; b < B; ++b) R3DM_HIP(c, ak_bgr_to_gray(st, stage + b * n0 * 4, img + b * n0, n0));
    }
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
    // per image (4096 words apart, kernels_akaze.hip kAkSmallWords): [0] maximum of the gradient modulus (float bits),
    // [16..) 300-bin histogram, [1024 + o] 1 / k^2 of octave o (ak_kcontrast_kernel)
    uint32_t* hmax_bits = small;
    uint32_t* hist = small + 16;
    float* inv_k2 = reinterpret_cast<float*>(small + 1024);

    // algorithmic HBM bytes of the launch sequence: every pass reads / writes whole image planes once
    // Beside it the COMPULSORY count: the planes a perfectly fused level would still move -- a smoothed plane in and out, the determinant
    // out, the conductivity out, and per FED step the evolving plane in and out (a step needs its neighbours' previous step, so steps do
    // not fuse across a plane without halo recomputation); the k-contrast statistics ride on the Gaussian.  This is round 2's 8 bytes per
    // pixel and pass; a roofline fraction on it falls when launches are fused, the as-structured one does not.
    double planes_px = 0.0, compulsory_px = 0.0;
    auto tally = [&](int lw, int lh, int n_planes, int n_compulsory) { planes_px += (double)lw * lh * n_planes; compulsory_px += (double)lw * lh * n_compulsory; };

    // Compute_Determinant_Hessian_Response_Single (AKAZEFeatures.cpp:389-410)
    auto hessian = [&](int i, const float* src) -> hipError_t {
        const int lw = lv[i].w, lh = lv[i].h, s = lv[i].sigma_size;
        hipError_t e;
        // two launches: smooth -> (Lx, Ly);  (Lx, Ly) -> Lxx, Lxy, Lyy on the spot -> the determinant
        if ((e = ak_scaled_deriv_xy(st, src, Lx(i), Ly(i), lw, lh, iB, s)) != hipSuccess) return e;
        tally(lw, lh, 3 + 3, 1);
        return ak_scaled_deriv_det(st, Lx(i), Ly(i), Ldet(i), lw, lh, iB, s);
    };

    // ---- Create_Nonlinear_Scale_Space (:245-369), Compute_Base_Evolution_Level (:199-237): ~550 launches for a 12 Mpx image,
    // none of which needs the host -- the k-contrast (compute_k_percentileV2: maximum, 300-bin histogram, percentile scan) stays
    // on the device.  (Replaying the sequence as a hipGraph was measured SLOWER than issuing it, 15.8 vs 7.5 ms per image:
    // DESIGN.md section 4.8 (c); the capture code is in the history.)
    auto scale_space = [&]() -> hipError_t {
        hipError_t e;
#define AK_TRY(call) do { if ((e = (call)) != hipSuccess) return e; } while (0)
        // (the base level's smoothed image IS evolution level 0: written straight into Lt(0), no copy)
        AK_TRY(ak_gaussian(st, img, tmp, Lt(0), w, h, iB, taps_off)); tally(w, h, 2, 2);
        AK_TRY(hessian(0, Lt(0)));
        AK_TRY(hipMemsetAsync(small, 0, (size_t)B * 4096 * 4, st));
        const int nbins = 300;
        if (nl > 1) {
            AK_TRY(ak_gaussian(st, img, tmp, flow, w, h, iB, taps_one)); tally(w, h, 2, 1);
            // (the Scharr derivative images of the reference exist only inside these two kernels: DESIGN.md section 4.8)
            AK_TRY(ak_modg_max(st, flow, w, h, iB, hmax_bits)); tally(w, h, 1, 0);
            AK_TRY(ak_modg_hist(st, flow, w, h, iB, hmax_bits, nbins, hist)); tally(w, h, 1, 0);
        }
        AK_TRY(ak_kcontrast(st, hmax_bits, hist, nbins, (uint32_t)((size_t)(w - 2) * (h - 2)), nl > 1 ? 1 : 0, inv_k2, iB));
        for (int i = 1; i < nl; ++i) {
            const int lw = lv[i].w, lh = lv[i].h;
            const size_t n = (size_t)lw * lh;
            const std::vector<float> tau = ak_fed_tau(lv[i].etime - lv[i - 1].etime);
            // (Splitting the launches of the 3 Mpx octave into sub-batches whose planes fit the Infinity Cache was measured: 2.31-2.34 ms
            // per image against 2.36 -- not worth a second launch order; profiles/r03_f_*.)
            const uint32_t SUB = B;
            bool head_used = false;
            size_t fed_launches = tau.size();                           // plane passes of the FED part (3 planes each): launches, not steps
            for (uint32_t b0 = 0; b0 < B; b0 += SUB) {
                const int nb = (int)std::min<uint32_t>(SUB, B - b0);
                const size_t po = (size_t)b0 * n;                       // every plane of this level's launches is n floats
                float* Lti = Lt(i) + po; float* lt2i = lt2 + po; float* tmpi = tmp + po; float* smoothi = smooth + po; float* flowi = flow + po;
                // FED launches of this level.  Product: up to four steps per launch in registers (ak_fed_march_kernel: a wavefront marches a
                // strip of columns down the rows, every step level three rows deep in registers -- 12 bytes of HBM traffic per pixel and
                // LAUNCH instead of per step).  The older forms stay for the developer build's A/B runs: one step per launch
                // (R3DM_AK_FED_MARCH=0 R3DM_AK_FED_MULTI=0), four steps through LDS on the levels of <= 3.2 Mpx (R3DM_AK_FED_MARCH=0).
                static const int march_knob = r3dm_dev_knob("R3DM_AK_FED_MARCH", 1);      // 0 = never, 1 = every level, > 1 = levels of at least that many pixels
                // steps per launch: up to 4 on the large levels (a step there is bound by the arithmetic of its cells, more per pass only
                // widens the halo), up to 6 below 1 Mpx per image, where a launch is mostly its own latency and fewer launches is the gain
                static const int kmax_knob = r3dm_dev_knob("R3DM_AK_FED_KMAX", 0);
                const int march_kmax = kmax_knob > 0 ? std::min(6, kmax_knob) : (n < (size_t)1000000 ? 6 : 4);
                static const int march_waves = std::max(256, r3dm_dev_knob("R3DM_AK_FED_WAVES", 12000));
                const bool march = march_knob && lw >= 3 && lh >= 3 && (march_knob == 1 || n >= (size_t)march_knob);
                static const int multi_knob = r3dm_dev_knob("R3DM_AK_FED_MULTI", 1);      // developer build: 0 = never, 1 = the product, > 1 = that many pixels
                static const int multi_px = multi_knob > 1 ? multi_knob : (multi_knob ? 3200000 : 0);
                // chunk[m] = steps of launch m
                std::vector<int> chunk;
                if (march) {
                    const int q = ((int)tau.size() + march_kmax - 1) / march_kmax;         // launches, steps spread evenly over them
                    for (int m = 0; m < q; ++m) chunk.push_back(((int)tau.size() * (m + 1)) / q - ((int)tau.size() * m) / q);
                } else {
                    const int per = n <= (size_t)multi_px ? 4 : 1;
                    for (size_t k0 = 0; k0 < tau.size(); k0 += (size_t)per) chunk.push_back((int)std::min<size_t>((size_t)per, tau.size() - k0));
                }
                const size_t n_launch = chunk.size();
                fed_launches = n_launch;
                const float* start = nullptr;
                if (lv[i].octave > lv[i - 1].octave) {
                    // the FED launches ping-pong between Lt(i) and the work image and must END in Lt(i): the half-sampled start image
                    // goes to whichever of the two the first launch does not write
                    float* half = (n_launch % 2 == 1) ? lt2i : Lti;
                    const HalfTabs& ht = half_tabs[i];
                    AK_TRY(ak_halfsample(st, Lt(i - 1) + (size_t)b0 * lv[i - 1].w * lv[i - 1].h, half, lv[i - 1].w, lv[i - 1].h, nb, ht.xt, ht.xb, ht.yt, ht.yb));
                    start = half;
                } else {
                    start = Lt(i - 1) + po;                             // same octave: the previous level IS the start image, no copy
                }
                if (tau.empty()) {                                      // (never for the reference's time steps) plain copy
                    if (start != Lti) AK_TRY(hipMemcpyAsync(Lti, start, (size_t)nb * n * 4, hipMemcpyDeviceToDevice, st));
                    start = Lti;
                }
                // Gaussian -> derivatives -> determinant -> conductivity.  Product: four HBM-bound launches (10 plane moves).  The one-pass form
                // (ak_level_head_kernel: a marching wavefront with the intermediates in LDS rings, 5 plane moves, bit-identical -- level_head.inc,
                // tests/cpp/level_head_emul.cpp) is built and MEASURED SLOWER, twice: 1,369 us (stages chained inside an iteration) and 1,573 us
                // (stages one iteration apart) against 1,083 us for the four launches at 8 x 12 Mpx.  PMC (profiles/r05_pmc_level_head.txt):
                // 271 scalar + 138 vector + 28 LDS instructions per 48 stored pixels -- the per-row bookkeeping of a marching wavefront
                // (ring slots, border rows, stage predicates) is paid once per 64 lanes and row, where a thread-per-pixel kernel pays it
                // once per four rows of loads; at ~620 G wave instructions/s the fused form is instruction-bound above the four launches'
                // HBM time.  It stays behind the developer knob R3DM_AK_HEAD=1 (tests/test_gpu_akaze.py runs it for bit-identity).
                static const int head_knob = r3dm_dev_knob("R3DM_AK_HEAD", 0);
                static const int head_waves = std::max(256, r3dm_dev_knob("R3DM_AK_HEAD_WAVES", 8000));
                const int s_i = lv[i].sigma_size;
                const bool head = head_knob && taps_one.n == 5 && s_i >= 2 && s_i <= 4 && lw >= 16 && lh >= 16;
                head_used = head;
                if (head) {
                    const int vw = 64 - 2 * (2 + 2 * s_i), strips = (lw + vw - 1) / vw;
                    int rows = (int)(((int64_t)lh * strips * nb + head_waves - 1) / head_waves);
                    rows = std::max(32, std::min(256, (rows + 15) / 16 * 16));
                    AK_TRY(ak_level_head(st, start, Lx(i) + po, Ly(i) + po, Ldet(i) + po, flowi, lw, lh, nb, taps_one, s_i,
                                         inv_k2 + (size_t)b0 * 4096 + lv[i].octave, rows));
                } else {
                    AK_TRY(ak_gaussian(st, start, tmpi, smoothi, lw, lh, nb, taps_one));
                    AK_TRY(ak_scaled_deriv_xy(st, smoothi, Lx(i) + po, Ly(i) + po, lw, lh, nb, s_i));
                    AK_TRY(ak_scaled_deriv_det(st, Lx(i) + po, Ly(i) + po, Ldet(i) + po, lw, lh, nb, s_i));
                    AK_TRY(ak_scharr_g2(st, smoothi, flowi, lw, lh, nb, inv_k2 + (size_t)b0 * 4096 + lv[i].octave));     // kcontrast * 0.75^octave
                }
                // Fast Explicit Diffusion: lt += lstep * 0.5 * tau_j; launch m of M writes Lt(i) when M - m is even, else the work image
                const float* cur = start;
                size_t k0 = 0;
                for (size_t m = 1; m <= n_launch; ++m) {
                    float* o = ((n_launch - m) % 2 == 0) ? Lti : lt2i;
                    const int kn = chunk[m - 1];
                    if (march) {
                        // rows per band: enough wavefronts to fill the chip (strips x bands x images >= march_waves), at most 128 rows
                        const int vw = 64 - 2 * kn, strips = (lw + vw - 1) / vw;
                        int rows = (int)(((int64_t)lh * strips * nb + march_waves - 1) / march_waves);
                        rows = std::max(16, std::min(128, (rows + 7) / 8 * 8));
                        AK_TRY(ak_fed_march(st, cur, flowi, o, lw, lh, nb, tau.data() + k0, kn, rows));
                    } else if (kn == 1 && n > (size_t)multi_px) AK_TRY(ak_fed_step(st, cur, flowi, o, lw, lh, nb, tau[k0]));
                    else AK_TRY(ak_fed_multi(st, cur, flowi, o, lw, lh, nb, tau.data() + k0, kn));
                    cur = o; k0 += (size_t)kn;
                }
            }
            if (lv[i].octave > lv[i - 1].octave) { tally(lv[i - 1].w, lv[i - 1].h, 1, 1); tally(lw, lh, 1, 1); }
            tally(lw, lh, (head_used ? 5 : 2 + 6 + 2) + 3 * (int)fed_launches, 2 + 1 + 1 + 2 * (int)tau.size());       // Gaussian (fused row + column pass) 2, derivatives + determinant 6, conductivity 2, 3 per FED launch (as structured); compulsory: 2 per FED step, the round-2 count
        }
#undef AK_TRY
        return hipSuccess;
    };
    R3DM_HIP(c, hipEventRecord(c->ev0, st));
    R3DM_HIP(c, scale_space());

    // ---- Feature_Detection (:371-382): extrema -> in-level pruning -> cross-level pruning -> refinement + orientation
    std::vector<AkLevelDev> ld((size_t)nl * B);
    size_t rows_img = 0;
    for (int i = 0; i < nl; ++i) rows_img += (size_t)std::max(0, lv[i].h - 2 * lv[i].border);
    const size_t rows_total = rows_img * B, n_lv = (size_t)nl * B;
    // row counts + row offsets, per-level counters (4 words), level table, per-image meta
    DevBuf& meta = buf(B_LEVEL0 + 4 * nl);
    const size_t off_levels = ((rows_total * 8 + n_lv * 16 + 15) / 16) * 16;
    const size_t off_bmeta = off_levels + ((n_lv * sizeof(AkLevelDev) + 15) / 16) * 16;
    R3DM_HIP(c, meta.ensure(off_bmeta + (size_t)B * sizeof(AkBatchMeta) + 256));
    uint32_t* rc = meta.as<uint32_t>();
    uint32_t* cnt = rc + 2 * rows_total;
    AkLevelDev* d_levels = reinterpret_cast<AkLevelDev*>(meta.as<unsigned char>() + off_levels);
    AkBatchMeta* d_bmeta = reinterpret_cast<AkBatchMeta*>(meta.as<unsigned char>() + off_bmeta);
    {
        std::vector<size_t> mask_off(nl + 1, 0);                          // in 64-bit words, per level, inside an image's mask area
        for (int i = 0; i < nl; ++i)
            mask_off[i + 1] = mask_off[i] + (size_t)std::max(0, lv[i].h - 2 * lv[i].border) * (size_t)std::max(0, (lv[i].w - 2 * lv[i].border + 63) / 64);
        if (mask_off[nl] * 8 + 8 > n0 * 4) { c->err = "detector: extremum masks do not fit the work image"; return R3DM_ERR_HIP; }
        size_t ro = 0;
        for (uint32_t b = 0; b < B; ++b)
            for (int i = 0; i < nl; ++i) {
                AkLevelDev& L = ld[(size_t)b * nl + i];
                L = AkLevelDev{};
                const size_t plane = (size_t)lv[i].w * lv[i].h;
                L.w = lv[i].w; L.h = lv[i].h; L.border = lv[i].border; L.ratio = lv[i].ratio; L.psize = lv[i].esigma * 1.5f;
                L.Ldet = Ldet(i) + b * plane; L.Lx = Lx(i) + b * plane; L.Ly = Ly(i) + b * plane; L.Lt = Lt(i) + b * plane;
                L.row_cnt = rc + ro; L.row_off = rc + rows_total + ro; L.counts = cnt + 4 * ((size_t)b * nl + i);
                // extremum bit masks: in the row-pass work image of the Gaussian (idle once the scale space is built), image b's
                // plane, the levels one after another (1 bit per pixel + at most 8 bytes per row: far below the plane's 4 bytes per pixel)
                const int rows_i = std::max(0, lv[i].h - 2 * lv[i].border);
                L.mask_words = (uint32_t)std::max(0, (lv[i].w - 2 * lv[i].border + 63) / 64);
                L.mask = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(tmp) + (((size_t)b * n0 * 4 + 7) / 8) * 8) + mask_off[i];   // (8-byte aligned also when w h is odd)
                ro += (size_t)rows_i;
            }
    }
    int max_rows = 0;
    for (int i = 0; i < nl; ++i) max_rows = std::max(max_rows, lv[i].h - 2 * lv[i].border);
    // tiles of the extremum count pass: 64 columns x 64 rows of a level's interior (kernels_akaze.hip kAkMaskRows = 16 rows per wave), the levels one after another
    AkTileTable tiles;
    uint32_t n_tiles = 0;
    for (int i = 0; i < 17; ++i) tiles.begin[i] = 0xFFFFFFFFu;
    if (nl > 16) { c->err = "detector: more than 16 evolution levels"; return R3DM_ERR_UNSUPPORTED; }
    for (int i = 0; i < nl; ++i) {
        tiles.begin[i] = n_tiles;
        const int rows_i = std::max(0, lv[i].h - 2 * lv[i].border), words_i = std::max(0, (lv[i].w - 2 * lv[i].border + 63) / 64);
        n_tiles += (uint32_t)words_i * (uint32_t)((rows_i + 63) / 64);
    }
    // slot capacity per image: a strict 3x3 maximum excludes its eight neighbours, so a level holds at most ceil(w/2) ceil(h/2)
    // candidates; start from min(that bound, 256 k) and grow only if an image reports more (the bound itself never overflows)
    uint64_t bound = 0;
    for (int i = 0; i < nl; ++i) bound += (uint64_t)((lv[i].w + 1) / 2) * (uint64_t)((lv[i].h + 1) / 2);
    // (R3DM_AK_CAP, developer build only: a tiny first capacity so that the tests reach the grow-and-repeat path)
    static const uint32_t cap0 = (uint32_t)std::max(1, r3dm_dev_knob("R3DM_AK_CAP", 1 << 18));
    uint32_t cap = (uint32_t)std::min<uint64_t>(bound, std::max<uint64_t>(c->ak_cap, cap0));
    std::vector<AkBatchMeta> bm(B);
    DevBuf& slots = buf(B_LEVEL0 + 4 * nl + 1);
    DevBuf& recs = buf(B_LEVEL0 + 4 * nl + 4);
    for (int attempt = 0;; ++attempt) {
        const size_t field = (size_t)B * cap;
        R3DM_HIP(c, slots.ensure(field * kAkSlotBytes + 256));
        R3DM_HIP(c, recs.ensure(field * sizeof(AkKpRec) + 256));
        R3DM_HIP(c, hipMemsetAsync(meta.p, 0, rows_total * 8 + n_lv * 16, st));
        R3DM_HIP(c, hipMemcpyAsync(d_levels, ld.data(), n_lv * sizeof(AkLevelDev), hipMemcpyHostToDevice, st));
        R3DM_HIP(c, ak_extrema_mask(st, d_levels, nl, iB, tiles, n_tiles, threshold));   // all levels of all images in one launch
        R3DM_HIP(c, ak_scan_rows(st, d_levels, nl, iB));
        R3DM_HIP(c, ak_layout(st, d_levels, nl, iB, slots.as<unsigned char>(), cap, d_bmeta));
        R3DM_HIP(c, hipMemsetAsync(slots.as<unsigned char>() + field * 76, 0, field * 2, st));   // dead_lower / dead_upper flags
        R3DM_HIP(c, ak_extrema(st, d_levels, nl, iB, max_rows, threshold, 1));
        R3DM_HIP(c, ak_prune_levels(st, d_levels, nl, iB));
        R3DM_HIP(c, ak_list_ranges(st, d_levels, nl, iB));
        R3DM_HIP(c, ak_cross(st, d_levels, nl, iB, 0));
        R3DM_HIP(c, ak_cross(st, d_levels, nl, iB, 1));
        R3DM_HIP(c, ak_refine(st, d_levels, nl, iB));
        R3DM_HIP(c, ak_compact(st, d_levels, nl, iB, recs.as<AkKpRec>(), cap, d_bmeta));
        R3DM_HIP(c, hipEventRecord(c->ev1, st));
        R3DM_HIP(c, hipMemcpyAsync(bm.data(), d_bmeta, (size_t)B * sizeof(AkBatchMeta), hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipStreamSynchronize(st));                                            // host visit 1 of 2: the counts
        uint32_t need = 0;
        for (uint32_t b = 0; b < B; ++b) if (bm[b].overflow) need = std::max(need, bm[b].need);
        if (!need) break;
        if (attempt >= 2 || need > bound) { c->err = "detector: candidate count exceeds its own bound"; return R3DM_ERR_HIP; }
        cap = (uint32_t)std::min<uint64_t>(bound, (uint64_t)need + need / 4 + 1024);       // grow and redo the detection phase (the scale space stays)
        c->feat_totals.n_regrows += 1;
    }
    c->ak_cap = cap;
    c->ak_n_levels = nl;
    c->ak_levels_dev = d_levels;
    for (uint32_t b = 0; b < B; ++b) {
        out.recs[b].resize(bm[b].n_kp);
        if (bm[b].n_kp) R3DM_HIP(c, hipMemcpyAsync(out.recs[b].data(), recs.as<AkKpRec>() + (size_t)b * cap, (size_t)bm[b].n_kp * sizeof(AkKpRec), hipMemcpyDeviceToHost, st));
    }
    R3DM_HIP(c, hipStreamSynchronize(st));                                                // host visit 2 of 2: the records
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.ms_detect_kernels = ms;
    c->stats.detect_algorithmic_bytes = planes_px * 4.0 * B; c->stats.detect_compulsory_bytes = compulsory_px * 4.0 * B;
    c->stats.ms_detect = now_ms() - t_call;
    r3dm_features_totals& T = c->feat_totals;
    T.n_images += B; T.n_passes += 1; T.ms_detect_kernels += ms; T.detect_algorithmic_bytes += planes_px * 4.0 * B; T.ms_wall += c->stats.ms_detect;
    for (uint32_t b = 0; b < B; ++b) T.n_keypoints += bm[b].n_kp;
    return R3DM_OK;
}

static int detect_akaze_impl(r3dm_ctx* c, const float* image, uint32_t width, uint32_t height, float threshold,
                             float* keypoints_out, float* responses_out, uint32_t cap, uint32_t* n_out, unsigned char* mldb_out)
{
    if (!c || !image || !n_out || (cap && !keypoints_out)) return R3DM_ERR_INVALID;
    *n_out = 0;
    AkBatchOut bo;
    const int rc = ak_detect_batch(c, 1, &image, nullptr, width, height, threshold, bo);
    if (rc != R3DM_OK) return rc;
    const std::vector<AkKpRec>& recs = bo.recs[0];
    const std::vector<AkLevelHost>& lv = bo.lv;
    hipStream_t st = c->stream;
    const uint32_t n_kp = (uint32_t)recs.size();
    std::vector<AkMldbItem> items;
    for (uint32_t k = 0; k < n_kp && k < cap; ++k) {
        const AkKpRec& r = recs[k];
        const float theta = ak_theta(r);
        if (mldb_out)          // Get_MLDB_Full_Descriptor: level coordinates, cos / sin of the raw (radian) angle
            items.push_back({r.level, r.x / lv[r.level].ratio, r.y / lv[r.level].ratio, cosf(theta), sinf(theta), (float)lv[r.level].sigma_size});
        keypoints_out[4 * (size_t)k] = r.x; keypoints_out[4 * (size_t)k + 1] = r.y;
        keypoints_out[4 * (size_t)k + 2] = r.size; keypoints_out[4 * (size_t)k + 3] = ak_angle_deg(theta);
        if (responses_out) responses_out[k] = r.response;
    }
    if (mldb_out && !items.empty()) {
        // comparison table of MLDB_Binary_Comparisons: per grid, per channel, all value pairs i < j
        std::vector<unsigned char> pairs;
        const int bases[3] = {0, 12, 39}, cnts[3] = {4, 9, 16};
        for (int g = 0; g < 3; ++g)
            for (int pos = 0; pos < 3; ++pos)
                for (int i = 0; i < cnts[g]; ++i)
                    for (int j = i + 1; j < cnts[g]; ++j) { pairs.push_back((unsigned char)(bases[g] + 3 * i + pos)); pairs.push_back((unsigned char)(bases[g] + 3 * j + pos)); }
        DevBuf& mb = c->ak_bufs[c->ak_bufs.size() - 6];          // B_LEVEL0 + 4 nl + 2
        const size_t ni = items.size();
        R3DM_HIP(c, mb.ensure(ni * sizeof(AkMldbItem) + 1024 + ni * 61 + 64));
        unsigned char* base = mb.as<unsigned char>();
        R3DM_HIP(c, hipMemcpyAsync(base, items.data(), ni * sizeof(AkMldbItem), hipMemcpyHostToDevice, st));
        R3DM_HIP(c, hipMemcpyAsync(base + ni * sizeof(AkMldbItem), pairs.data(), pairs.size(), hipMemcpyHostToDevice, st));
        unsigned char* d_out = base + ni * sizeof(AkMldbItem) + 1024;
        R3DM_HIP(c, ak_mldb(st, c->ak_levels_dev, (const AkMldbItem*)base, (uint32_t)ni, base + ni * sizeof(AkMldbItem), d_out));
        R3DM_HIP(c, hipMemcpyAsync(mldb_out, d_out, ni * 61, hipMemcpyDeviceToHost, st));
        R3DM_HIP(c, hipStreamSynchronize(st));
    }
    *n_out = n_kp;
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

// B same-size images in one pass of the detector.  keypoints_out[b]: cap x 4 floats (x, y, size, angle in degrees),
// responses_out (optional, entries optional): cap floats, n_out[b] = number detected (may exceed cap).
extern "C" int r3dm_detect_akaze_batch(r3dm_ctx* c, uint32_t n_images, const float* const* images, uint32_t width, uint32_t height,
                                       float threshold, float* const* keypoints_out, float* const* responses_out, uint32_t cap,
                                       uint32_t* n_out)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!c || !images || !n_out || (cap && !keypoints_out)) return R3DM_ERR_INVALID;
        AkBatchOut bo;
        const int rc = ak_detect_batch(c, n_images, images, nullptr, width, height, threshold, bo);
        if (rc != R3DM_OK) return rc;
        for (uint32_t b = 0; b < n_images; ++b) {
            const std::vector<AkKpRec>& recs = bo.recs[b];
            n_out[b] = (uint32_t)recs.size();
            for (uint32_t k = 0; k < recs.size() && k < cap; ++k) {
                float* o = keypoints_out[b] + 4 * (size_t)k;
                o[0] = recs[k].x; o[1] = recs[k].y; o[2] = recs[k].size; o[3] = ak_angle_deg(ak_theta(recs[k]));
                if (responses_out && responses_out[b]) responses_out[b][k] = recs[k].response;
            }
        }
        return R3DM_OK;
    });
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
    if (pix.size() > 1024) { c->err = "liop: support larger than the sort capacity"; return R3DM_ERR_UNSUPPORTED; }
    // per support pixel: the four sample positions as (offset of the top-left tap, fractional parts).  The kernel keeps the patch with
    // a ring of zeros (43 x 43), so vl_liop's guarded taps (:516-535: `if (ix >= 0 && iy >= 0) a = ...`) are plain reads; floor and
    // fraction are the reference's own double operations, done once here instead of once per sample and keypoint
    std::vector<double> sw(8 * pix.size());
    std::vector<int> so(4 * pix.size()), pixr(pix.size());
    const double dangle = 2 * M_PI / 4.0;
    for (size_t i = 0; i < pix.size(); ++i) {
        const double x = (pix[i] % side) - center, y = (pix[i] / side) - center;
        const double angle0 = std::atan2(y, x);
        pixr[i] = (pix[i] % side + 1) + (pix[i] / side + 1) * 43;
        for (int k = 0; k < 4; ++k) {
            const double sx = x + radius * std::cos(angle0 + dangle * k) + center;
            const double sy = y + radius * std::sin(angle0 + dangle * k) + center;
            const long xi = (long)sx, yi = (long)sy;
            const long ix = (sx >= 0 || (double)xi == sx) ? xi : xi - 1;          // vl_floor_d
            const long iy = (sy >= 0 || (double)yi == sy) ? yi : yi - 1;
            if (ix < -1 || ix > side - 1 || iy < -1 || iy > side - 1) { c->err = "liop: a sample leaves the ringed patch"; return R3DM_ERR_UNSUPPORTED; }
            sw[8 * i + 2 * k] = sx - ix; sw[8 * i + 2 * k + 1] = sy - iy;
            so[4 * i + k] = (int)((ix + 1) + (iy + 1) * 43);
        }
    }
    R3DM_HIP(c, c->liop_pix.ensure(pixr.size() * 4));
    R3DM_HIP(c, c->liop_sx.ensure(sw.size() * 8));
    R3DM_HIP(c, c->liop_sy.ensure(so.size() * 4));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_pix.p, pixr.data(), pixr.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_sx.p, sw.data(), sw.size() * 8, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_sy.p, so.data(), so.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipStreamSynchronize(c->stream));
    c->liop_npix = (uint32_t)pix.size();
    return R3DM_OK;
}
static inline LiopTables liop_tables(const r3dm_ctx* c)
{
    return LiopTables{c->liop_pix.as<int>(), c->liop_sx.as<double>(), c->liop_sy.as<int>(), c->liop_npix};
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
    R3DM_HIP(c, c->liop_cnt.ensure(64 + (size_t)n * 4));          // [tie count | ...][tie list: n]
    R3DM_HIP(c, hipMemcpyAsync(c->liop_in.p, patches, in_bytes, hipMemcpyDefault, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    R3DM_HIP(c, launch_liop(c->stream, liop_tables(c), c->liop_in.as<float>(), n, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>(), c->liop_cnt.as<uint32_t>() + 16));
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

// cv::getGaussianKernel(11, 1.2, CV_32F): the blur of every LIOP patch (src/Regard3DFeatures.cpp:807)
static void liop_blur_taps(float (&kern)[11])
{
    const double scale2X = -0.5 / (1.2 * 1.2);
    double sum = 0;
    for (int i = 0; i < 11; ++i) { const double xx = i - 5.0; kern[i] = (float)std::exp(scale2X * xx * xx); sum += kern[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 11; ++i) kern[i] = (float)(kern[i] * sum);
}

// 2x3 inverse map of one keypoint's patch exactly as src/Regard3DFeatures.cpp:786-799 computes it
static inline void liop_patch_map(float x, float y, float size, float angle_deg, float kp_size_factor, float* m)
{
    const int patchResolution = 20, patchSize = 41;
    const float angle = -90.0f - angle_deg;
    const float scale = size / static_cast<float>(patchSize) * kp_size_factor;
    const float alpha = scale * std::cos(angle * M_PI / 180.0f);
    const float beta = scale * std::sin(angle * M_PI / 180.0f);
    const float trans_x = x - static_cast<float>(patchResolution), trans_y = y - static_cast<float>(patchResolution);
    m[0] = alpha; m[1] = beta;  m[2] = beta * trans_y + alpha * trans_x - beta * y + (1.0f - alpha) * x;
    m[3] = -beta; m[4] = alpha; m[5] = alpha * trans_y - beta * trans_x + beta * x + (1.0f - alpha) * y;
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
    for (uint32_t k = 0; k < n; ++k) liop_patch_map(kp[4 * k], kp[4 * k + 1], kp[4 * k + 2], kp[4 * k + 3], kp_size_factor, &M6[6 * (size_t)k]);
    float kern[11];
    liop_blur_taps(kern);
    const size_t img_bytes = (size_t)width * height * 4, patch_bytes = (size_t)n * 41 * 41 * 4, out_bytes = (size_t)n * 144 * 4;
    if (!resident_image) R3DM_HIP(c, c->liop_img.ensure(img_bytes));
    R3DM_HIP(c, c->liop_M.ensure(M6.size() * 4));
    R3DM_HIP(c, c->liop_kern.ensure(64));
    static const int fused_knob = r3dm_dev_knob("R3DM_LIOP_FUSED", 1);
    const bool via_patches = patches_out || !fused_knob;
    if (via_patches) R3DM_HIP(c, c->liop_in.ensure(patch_bytes));
    R3DM_HIP(c, c->liop_out.ensure(out_bytes));
    R3DM_HIP(c, c->liop_cnt.ensure(64 + (size_t)n * 4));
    if (!resident_image) R3DM_HIP(c, hipMemcpyAsync(c->liop_img.p, image, img_bytes, hipMemcpyDefault, c->stream));
    const float* dev_image = resident_image ? resident_image : c->liop_img.as<float>();
    R3DM_HIP(c, hipMemcpyAsync(c->liop_M.p, M6.data(), M6.size() * 4, hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemcpyAsync(c->liop_kern.p, kern, sizeof(kern), hipMemcpyHostToDevice, c->stream));
    R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
    R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
    // the patches only exist in HBM when the caller asks for them (or the developer build's R3DM_LIOP_FUSED=0): otherwise the warp + blur
    // runs inside the descriptor kernel's wavefront
    if (via_patches) {
        R3DM_HIP(c, launch_liop_extract(c->stream, dev_image, (int)width, (int)height, c->liop_M.as<float>(),
                                        c->liop_kern.as<float>(), n, c->liop_in.as<float>()));
        R3DM_HIP(c, launch_liop(c->stream, liop_tables(c), c->liop_in.as<float>(), n, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>(), c->liop_cnt.as<uint32_t>() + 16));
    } else {
        R3DM_HIP(c, launch_liop_fused(c->stream, liop_tables(c), dev_image, (int)width, (int)height, c->liop_M.as<float>(), c->liop_kern.as<float>(), nullptr,
                                      n, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>(), c->liop_cnt.as<uint32_t>() + 16));
    }
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

// "%g" of one float, locale-independent (the host application runs under setlocale(LC_ALL, "")): printf("%.6g") in the "C" locale,
// i.e. std::to_chars(general, 6) -- through the exact fast path of fmt_g6.hpp for the magnitudes a .feat file holds
static inline char* put_g(char* p, char* end, float v, float* as_parsed = nullptr) { return r3dm_fmt::put_g6(p, end, v, as_parsed); }

// KeypointSet::saveToBinFile (src/keypointSet.hpp:61-67): .feat = one "x y scale orientation" line per feature
// (SIOPointFeature::operator<<, default float formatting; scale = size / 2, :835-836), .desc = count + raw rows
// xy_as_written (optional, n x 2): the positions as a reader of the file parses them (std::from_chars on the text just written)
// (two halves: the text of the .feat file -- and with it the positions as a reader parses them -- and the two fwrites, which a context
//  with r3dm_set_deferred_feature_files leaves to its writer thread)
static size_t format_feat(std::vector<char>& txt, const float* kps, uint32_t n, float* xy_as_written)
{
    txt.resize((size_t)n * 64 + 64);
    char* p = txt.data(); char* const end = p + txt.size();
    for (uint32_t k = 0; k < n; ++k) {
        float* const back = xy_as_written ? xy_as_written + 2 * (size_t)k : nullptr;
        p = put_g(p, end, kps[4 * (size_t)k], back); *p++ = ' ';
        p = put_g(p, end, kps[4 * (size_t)k + 1], back ? back + 1 : nullptr); *p++ = ' ';
        p = put_g(p, end, kps[4 * (size_t)k + 2] / 2.0f); *p++ = ' ';
        p = put_g(p, end, kps[4 * (size_t)k + 3]); *p++ = '\n';
    }
    return (size_t)(p - txt.data());
}

static int write_feat_desc_files(std::string& err, const char* feat_path, const char* desc_path, const char* txt, size_t txt_len, const float* desc, uint32_t n)
{
    FILE* f = fopen(feat_path, "wb");
    if (!f) { err = std::string("cannot write ") + feat_path; return R3DM_ERR_IO; }
    bool ok = txt_len == 0 || fwrite(txt, 1, txt_len, f) == txt_len;
    ok = (fclose(f) == 0) && ok;
    if (!ok) { err = std::string("cannot write ") + feat_path; return R3DM_ERR_IO; }
    f = fopen(desc_path, "wb");
    if (!f) { err = std::string("cannot write ") + desc_path; return R3DM_ERR_IO; }
    const uint64_t cnt = n;
    ok = fwrite(&cnt, 8, 1, f) == 1 && (n == 0 || fwrite(desc, 144 * 4, n, f) == n);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { err = std::string("cannot write ") + desc_path; return R3DM_ERR_IO; }
    return R3DM_OK;
}

static int write_feat_desc(std::string& err, const char* feat_path, const char* desc_path, const float* kps, const float* desc, uint32_t n,
                           float* xy_as_written = nullptr)
{
    std::vector<char> txt;
    const size_t len = format_feat(txt, kps, n, xy_as_written);
    return write_feat_desc_files(err, feat_path, desc_path, txt.data(), len, desc, n);
}

// the writer thread of a context with deferred feature files: joined before pin_desc is filled again and by the wait entry
static int features_files_join(r3dm_ctx* c)
{
    if (c->file_writer.joinable()) c->file_writer.join();
    const int rc = c->file_writer_rc;
    if (rc != R3DM_OK) c->err = c->file_writer_err.empty() ? "writing the feature files failed" : c->file_writer_err;
    c->file_writer_rc = R3DM_OK; c->file_writer_err.clear();
    return rc;
}

static bool both_files_exist(const char* feat_path, const char* desc_path, uint32_t* n_rows)
{
    FILE* ff = fopen(feat_path, "rb");
    FILE* fd = ff ? fopen(desc_path, "rb") : nullptr;
    if (ff) fclose(ff);
    if (!fd) return false;
    uint64_t cnt = 0;
    if (fread(&cnt, 8, 1, fd) != 1) cnt = 0;
    fclose(fd);
    if (n_rows) *n_rows = (uint32_t)cnt;
    return true;
}

// detectAndExtract (src/Regard3DFeatures.cpp:206-222) for keypointDetectorList_ = {"Fast-AKAZE"} over a batch of B same-size
// images + KeypointSet::saveToBinFile of each: detector batch -> (host: angle and patch map of every keypoint, 24 bytes each back
// to the device) -> one LIOP patch-extraction launch and one LIOP launch over the keypoints of ALL images -> descriptors to page-locked
// host memory -> files.  Every image of the batch is computed (the skip rule is the caller's: it only batches images it wants).
// kps_out / desc_out (optional): the keypoints (x, y, size, angle) and descriptors of every image, for callers that register the
// views without reading the files back.
static int extract_features_batch_impl(r3dm_ctx* c, uint32_t B, const float* const* grays, const unsigned char* const* bgrs,
                                       uint32_t width, uint32_t height, float threshold, const char* const* feat_paths,
                                       const char* const* desc_paths, uint32_t* n_features,
                                       std::vector<std::vector<float>>* kps_out = nullptr, std::vector<std::vector<float>>* desc_out = nullptr)
{
    if (!c || B == 0) return R3DM_ERR_INVALID;
    AkBatchOut bo;
    int rc = ak_detect_batch(c, B, grays, bgrs, width, height, threshold, bo);
    if (rc != R3DM_OK) return rc;
    const double t_liop = now_ms();
    size_t n_total = 0;
    std::vector<size_t> first(B + 1, 0);
    for (uint32_t b = 0; b < B; ++b) { first[b] = n_total; n_total += bo.recs[b].size(); }
    first[B] = n_total;
    std::vector<float> kps(4 * n_total), M6(6 * n_total);
    std::vector<uint32_t> img_of(n_total);
    // helper threads of this batch: the cores the process really owns (its cgroup quota), shared with the other batches in flight
    static std::atomic<int> batches_in_flight{0};
    struct InFlight { std::atomic<int>& n; int mine; InFlight(std::atomic<int>& a) : n(a), mine(a.fetch_add(1) + 1) {} ~InFlight() { n.fetch_sub(1); } } in_flight(batches_in_flight);
    const int host_team = r3dm_host_team(8, std::max(2, in_flight.mine));
    // angle (atan2f of the host libm, as the reference) and LIOP patch map of every keypoint: a few host threads share the loop
    {
        // chunks of 4,096 keypoints over all images of the batch
        struct Chunk { uint32_t b; long k0, k1; };
        std::vector<Chunk> chunks;
        for (uint32_t b = 0; b < B; ++b) {
            const long nk = (long)bo.recs[b].size();
            for (long k0 = 0; k0 < nk; k0 += 4096) chunks.push_back({b, k0, std::min(nk, k0 + 4096)});
        }
        r3dm_parallel_for((long)chunks.size(), host_team, [&](long ci) {
            const Chunk& ch = chunks[(size_t)ci];
            const AkKpRec* recs = bo.recs[ch.b].data();
            const size_t f0 = first[ch.b];
            for (long k = ch.k0; k < ch.k1; ++k) {
                const AkKpRec& r = recs[k];
                const size_t g = f0 + (size_t)k;
                float* o = &kps[4 * g];
                o[0] = r.x; o[1] = r.y; o[2] = r.size; o[3] = ak_angle_deg(ak_theta(r));
                liop_patch_map(o[0], o[1], o[2], o[3], 8.0f /* getKpSizeFactor("Fast-AKAZE"), :703-704 */, &M6[6 * g]);
                img_of[g] = ch.b;
            }
        });
    }
    const float* desc_host = nullptr;
    const bool deferred = c->defer_files && feat_paths && desc_paths;
    if (n_total) {
        rc = liop_prepare(c);
        if (rc != R3DM_OK) return rc;
        float kern[11];
        liop_blur_taps(kern);
        const size_t patch_bytes = n_total * 41 * 41 * 4, out_bytes = n_total * 144 * 4;
        R3DM_HIP(c, c->liop_M.ensure(M6.size() * 4 + img_of.size() * 4));
        R3DM_HIP(c, c->liop_kern.ensure(64));
        R3DM_HIP(c, c->liop_out.ensure(out_bytes));
        R3DM_HIP(c, c->liop_cnt.ensure(64 + n_total * 4));
        { const int wrc_prev = features_files_join(c); if (wrc_prev != R3DM_OK) return wrc_prev; }      // the previous batch's writer still reads pin_desc
        R3DM_HIP(c, c->pin_desc.ensure(out_bytes));
        uint32_t* d_img_of = reinterpret_cast<uint32_t*>(c->liop_M.as<float>() + M6.size());
        R3DM_HIP(c, hipMemcpyAsync(c->liop_M.p, M6.data(), M6.size() * 4, hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(d_img_of, img_of.data(), img_of.size() * 4, hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(c->liop_kern.p, kern, sizeof(kern), hipMemcpyHostToDevice, c->stream));
        R3DM_HIP(c, hipMemsetAsync(c->liop_cnt.p, 0, 64, c->stream));
        R3DM_HIP(c, hipEventRecord(c->ev0, c->stream));
        // the detector has left the B gray images in its image buffer (ak_bufs[0], B planes, read-only for it)
        static const int fused_knob = r3dm_dev_knob("R3DM_LIOP_FUSED", 1);     // developer build: 0 = patches through HBM (two kernels)
        if (!fused_knob) {
            R3DM_HIP(c, c->liop_in.ensure(patch_bytes));
            R3DM_HIP(c, launch_liop_extract(c->stream, c->ak_bufs[0].as<float>(), (int)width, (int)height, c->liop_M.as<float>(),
                                            c->liop_kern.as<float>(), (uint32_t)n_total, c->liop_in.as<float>(), d_img_of));
            R3DM_HIP(c, launch_liop(c->stream, liop_tables(c), c->liop_in.as<float>(), (uint32_t)n_total, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>(), c->liop_cnt.as<uint32_t>() + 16));
        } else {
            R3DM_HIP(c, launch_liop_fused(c->stream, liop_tables(c), c->ak_bufs[0].as<float>(), (int)width, (int)height, c->liop_M.as<float>(), c->liop_kern.as<float>(),
                                          d_img_of, (uint32_t)n_total, c->liop_out.as<float>(), c->liop_cnt.as<uint32_t>(), c->liop_cnt.as<uint32_t>() + 16));
        }
        R3DM_HIP(c, hipEventRecord(c->ev1, c->stream));
        R3DM_HIP(c, hipMemcpyAsync(c->pin_desc.p, c->liop_out.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
        desc_host = c->pin_desc.as<float>();
        if (deferred) {
            // the descriptors travel to the host behind the kernel; only the writer thread waits for them (ev_desc)
            if (!c->ev_desc) R3DM_HIP(c, hipEventCreateWithFlags(&c->ev_desc, hipEventDisableTiming));
            R3DM_HIP(c, hipEventRecord(c->ev_desc, c->stream));
        } else {
            R3DM_HIP(c, hipStreamSynchronize(c->stream));
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            c->stats.ms_liop_kernel = ms;
        }
    }
    if (!deferred) c->stats.ms_liop_wall = now_ms() - t_liop;
    const double t_io = now_ms();
    // the files of the B images are formatted and written by up to 8 host threads (28 k keypoints = 113 k decimal conversions and
    // 16 MB per image); the sink then sees the images in batch order from this thread
    const bool to_files = feat_paths && desc_paths;
    std::vector<int> wrc(B, R3DM_OK);
    std::vector<std::string> werr(B);
    if (to_files && deferred) {
        // deferred files (r3dm_set_deferred_feature_files): the text of the .feat files is formatted while the LIOP kernel runs (it needs
        // the keypoints only); when the kernel is done the images go to the sink (which reads the descriptors on the device); the
        // fwrites -- 16 MB of descriptors per 28 k keypoints, still on their way to pin_desc -- are left to the context's writer thread,
        // which runs beside whatever the caller does next
        struct Job { std::string feat, desc; std::vector<char> txt; size_t len; const float* rows; uint32_t n; std::vector<float> xy; };
        auto jobs = std::make_shared<std::vector<Job>>(B);
        r3dm_parallel_for((long)B, host_team, [&](long b) {
            if (!feat_paths[b] || !desc_paths[b]) return;
            const uint32_t n = (uint32_t)bo.recs[b].size();
            try {
                Job& j = (*jobs)[(size_t)b];
                if (c->feat_sink) j.xy.resize((size_t)n * 2 + 2);
                j.feat = feat_paths[b]; j.desc = desc_paths[b]; j.n = n; j.rows = desc_host ? desc_host + 144 * first[b] : nullptr;
                j.len = format_feat(j.txt, kps.data() + 4 * first[b], n, c->feat_sink ? j.xy.data() : nullptr);
            } catch (...) { wrc[b] = R3DM_ERR_NOMEM; }
        });
        if (n_total) {
            R3DM_HIP(c, hipEventSynchronize(c->ev1));            // the LIOP kernel (the copy to the host is still running)
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            c->stats.ms_liop_kernel = ms;
        }
        c->stats.ms_liop_wall = now_ms() - t_liop;
        if (c->feat_sink) {
            r3dm_parallel_for((long)B, host_team, [&](long b) {
                if (!feat_paths[b] || !desc_paths[b] || wrc[b] != R3DM_OK) return;
                const uint32_t n = (uint32_t)bo.recs[b].size();
                const uint32_t id = c->feat_sink_ids ? c->feat_sink_ids[b] : (uint32_t)b;
                int src = 1;
                try { src = c->feat_sink(c->feat_sink_user, id, n, n ? c->liop_out.as<float>() + 144 * first[b] : nullptr, (*jobs)[(size_t)b].xy.data()); } catch (...) {}
                if (src != 0) { wrc[b] = R3DM_ERR_INVALID; werr[b] = "the features sink refused image " + std::to_string(id); }
                std::vector<float>().swap((*jobs)[(size_t)b].xy);
            });
            (void)hipSetDevice(c->device);
        }
        { const int wrc_prev = features_files_join(c); if (wrc_prev != R3DM_OK) return wrc_prev; }      // (a batch without keypoints has not joined it above)
        // a batch that failed here (text formatting, a sink that refused an image) is reported now and writes NO files: a writer started
        // for it would make the files of a failed call appear later, in the background
        bool batch_ok = true;
        for (uint32_t b = 0; b < B; ++b) if (feat_paths[b] && desc_paths[b] && wrc[b] != R3DM_OK) batch_ok = false;
        const int nice_value = c->background_nice;
        if (batch_ok) try {
            // (two threads per context: the writes have the caller's next phase to hide behind and must not take its cores)
            c->file_writer = std::thread([c, jobs, nice_value, writer_team = std::min(host_team, 2), wait_desc = n_total != 0]() {
                r3dm_background_thread(nice_value);
                const double t0 = now_ms();
                if (wait_desc && (hipSetDevice(c->device) != hipSuccess || hipEventSynchronize(c->ev_desc) != hipSuccess)) {
                    c->file_writer_rc = R3DM_ERR_HIP; c->file_writer_err = "the descriptors did not reach the host";
                    return;
                }
                std::vector<int> rcs(jobs->size(), R3DM_OK); std::vector<std::string> errs(jobs->size());
                r3dm_parallel_for((long)jobs->size(), writer_team, [&](long b) {
                    const Job& j = (*jobs)[(size_t)b];
                    if (j.feat.empty()) return;
                    try { rcs[(size_t)b] = write_feat_desc_files(errs[(size_t)b], j.feat.c_str(), j.desc.c_str(), j.txt.data(), j.len, j.rows, j.n); }
                    catch (...) { rcs[(size_t)b] = R3DM_ERR_NOMEM; }
                });
                // (the report comes late -- at the wait, or at this context's next batch: it names the file so that it can be traced to its batch)
                for (size_t b = 0; b < rcs.size(); ++b)
                    if (rcs[b] != R3DM_OK && c->file_writer_rc == R3DM_OK) {
                        c->file_writer_rc = rcs[b];
                        c->file_writer_err = "deferred feature files of " + (*jobs)[b].feat + ": " + (errs[b].empty() ? std::string("write failed") : errs[b]);
                    }
                c->file_writer_ms += now_ms() - t0;
            });
        } catch (...) { c->err = "cannot start the feature-file writer"; return R3DM_ERR_NOMEM; }
    } else if (to_files) {
        // ... and each thread hands its image to the sink (if any) as soon as its files are written: the sink of the facade registers the
        // view with the matcher (position classes, device-to-device copy, re-layout kernels) while the other threads still format theirs.
        // The descriptors of the batch are still in liop_out (this context's stream is idle: the copy above was waited for).
        r3dm_parallel_for((long)B, host_team, [&](long b) {
            if (!feat_paths[b] || !desc_paths[b]) return;
            const uint32_t n = (uint32_t)bo.recs[b].size();
            try {                                                   // nothing may leave an OpenMP region by exception
                std::vector<float> xy_written;
                if (c->feat_sink) xy_written.resize((size_t)n * 2 + 2);
                std::string e;
                wrc[b] = write_feat_desc(e, feat_paths[b], desc_paths[b], kps.data() + 4 * first[b], desc_host ? desc_host + 144 * first[b] : nullptr, n,
                                         c->feat_sink ? xy_written.data() : nullptr);
                werr[b] = e;
                if (wrc[b] == R3DM_OK && c->feat_sink) {
                    const uint32_t id = c->feat_sink_ids ? c->feat_sink_ids[b] : (uint32_t)b;
                    const int src = c->feat_sink(c->feat_sink_user, id, n, n ? c->liop_out.as<float>() + 144 * first[b] : nullptr, xy_written.data());
                    if (src != 0) { wrc[b] = R3DM_ERR_INVALID; werr[b] = "the features sink refused image " + std::to_string(id); }
                }
            } catch (...) { wrc[b] = R3DM_ERR_NOMEM; }
        });
        (void)hipSetDevice(c->device);                         // a sink may have worked on another device from this thread
    }
    if (deferred && desc_out && n_total) R3DM_HIP(c, hipEventSynchronize(c->ev_desc));
    for (uint32_t b = 0; b < B; ++b) {
        const uint32_t n = (uint32_t)bo.recs[b].size();
        if (to_files && feat_paths[b] && desc_paths[b] && wrc[b] != R3DM_OK) { c->err = werr[b].empty() ? "out of host memory" : werr[b]; return wrc[b]; }
        if (n_features) n_features[b] = n;
        if (kps_out) (*kps_out)[b].assign(kps.begin() + 4 * first[b], kps.begin() + 4 * first[b + 1]);
        if (desc_out) { if (n) (*desc_out)[b].assign(desc_host + 144 * first[b], desc_host + 144 * first[b + 1]); else (*desc_out)[b].clear(); }
    }
    c->stats.ms_feature_files = now_ms() - t_io;            // (the sink's time included)
    c->feat_totals.ms_liop_kernels += n_total ? c->stats.ms_liop_kernel : 0.0;
    c->feat_totals.ms_wall += c->stats.ms_liop_wall + c->stats.ms_feature_files;
    c->feat_totals.ms_files += c->stats.ms_feature_files;
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
    if (both_files_exist(feat_path, desc_path, n_features)) return R3DM_OK;
    return extract_features_batch_impl(c, 1, &gray, nullptr, width, height, threshold, &feat_path, &desc_path, n_features);
}

extern "C" int r3dm_extract_features_to_files(r3dm_ctx* c, const float* gray, uint32_t width, uint32_t height, float threshold,
                                              const char* feat_path, const char* desc_path, uint32_t* n_features)
{
    return r3dm_guarded(c, [&]() -> int { return r3dm_extract_features_to_files_impl(c, gray, width, height, threshold, feat_path, desc_path, n_features); });
}

// B same-size images through detector + LIOP + files in one pass (no skip rule: the caller decides what to compute).
// grays: B pointers to height x width floats, or NULL and bgrs: B pointers to height x width x 3 bytes (BGR, as cv::imread decodes).
extern "C" int r3dm_extract_features_batch(r3dm_ctx* c, uint32_t n_images, const float* const* grays, const unsigned char* const* bgrs,
                                           uint32_t width, uint32_t height, float threshold, const char* const* feat_paths,
                                           const char* const* desc_paths, uint32_t* n_features)
{
    return r3dm_guarded(c, [&]() -> int {
        if (!feat_paths || !desc_paths) return R3DM_ERR_INVALID;
        return extract_features_batch_impl(c, n_images, grays, bgrs, width, height, threshold, feat_paths, desc_paths, n_features); });
}


// ------------------------------------------------------------------------------------------------
// the features stage over a whole image list: R3DFeaturesThread::extractFeaturesAndDescriptors
// (src/threads/R3DFeaturesThread.cpp:38-89) -- a pool of CPUs + 1 worker threads pulling images off a work list (:93-121), each
// running processWorkItem (:123-210).  The reference admits ONE image at a time into the A-KAZE scale space
// (initAKAZESemaphore(1), src/R3DComputeMatches.cpp:1847; src/Regard3DFeatures.cpp:71-125) to bound host memory; HBM does not
// need that: every context of an r3dm_multi (r3dm_multi_create with a device id repeated K times = K streams + work buffers on
// that device; or one per GPU) is a worker that pulls BATCHES of same-size images off the list from its own host thread -- B images
// per pass of the detector, K passes in flight, the host part of one pass (angles, files) hidden behind the kernels of the others.
// An image whose .feat AND .desc both exist is skipped, exactly as processWorkItem does (:139-142: stale files of other parameters
// are reused; the reference wipes the matches directory instead, src/threads/R3DComputeMatchesThread.cpp:84-86); n_features then
// reports the row count of the existing .desc.
// ------------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <thread>
#include <tuple>

extern "C" {
int r3dm_multi_num_devices(const r3dm_multi* m);
r3dm_ctx* r3dm_multi_ctx(r3dm_multi* m, int k);
}

namespace {

// batch size for images of w x h on context c: 8 when HBM allows (the work buffers take ~125 bytes per pixel and image),
// fewer for very large images or a nearly full device
uint32_t ak_batch_for(r3dm_ctx* c, uint32_t w, uint32_t h, uint32_t want)
{
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(c->device);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1;
    const double per_image = (double)w * h * 4.0 * 32.0 + 64e6;
    const double budget = (double)free_b * 0.5 + (c->ak_w == (int)w && c->ak_h == (int)h ? (double)c->ak_B * per_image : 0.0);
    uint32_t b = want;
    while (b > 1 && b * per_image > budget) --b;
    return b;
}

int multi_extract_impl(r3dm_multi* m, uint32_t n_images, const float* const* grays, const unsigned char* const* bgrs,
                       const uint32_t* widths, const uint32_t* heights, float threshold, const char* const* feat_paths,
                       const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped, uint32_t batch, char* err, size_t err_cap)
{
    if (!m || (n_images && ((!grays && !bgrs) || !widths || !heights || !feat_paths || !desc_paths))) return R3DM_ERR_INVALID;
    // per image: gray floats if grays[i] is set, else 8-bit BGR
    auto is_gray = [&](uint32_t i) { return grays && grays[i]; };
    if (err && err_cap) err[0] = 0;
    const uint32_t concurrency = (uint32_t)r3dm_multi_num_devices(m);
    if (concurrency == 0) return R3DM_ERR_INVALID;
    if (batch == 0) batch = 8;
    int rc_all = R3DM_OK;
    std::vector<std::thread> th;
    try {
        // work list: the images still to compute (the skip rule applied up front), grouped by size so that a worker's batch is
        // a run of same-size images; the order inside a size class is the caller's
        std::vector<uint32_t> todo;
        for (uint32_t i = 0; i < n_images; ++i) {
            if (skipped) skipped[i] = 0;
            uint32_t rows = 0;
            if (both_files_exist(feat_paths[i], desc_paths[i], &rows)) {            // processWorkItem: already computed
                if (n_features) n_features[i] = rows;
                if (skipped) skipped[i] = 1;
            } else todo.push_back(i);
        }
        for (uint32_t i : todo) if (!is_gray(i) && !(bgrs && bgrs[i])) return R3DM_ERR_INVALID;      // an image to compute without pixels
        std::stable_sort(todo.begin(), todo.end(), [&](uint32_t a, uint32_t b) {
            return std::make_tuple(widths[a], heights[a], is_gray(a)) < std::make_tuple(widths[b], heights[b], is_gray(b)); });
        std::mutex mu;
        size_t next = 0;
        std::vector<int> rcs(concurrency, R3DM_OK);
        std::vector<std::string> errs(concurrency);
        // small lists: shrink the batch so that every worker gets something to do
        const uint32_t fair = (uint32_t)std::max<size_t>(1, (todo.size() + concurrency - 1) / concurrency);
        auto worker = [&](uint32_t k) noexcept {
            try {
                r3dm_ctx* c = r3dm_multi_ctx(m, (int)k);
                for (;;) {
                    std::vector<uint32_t> mine;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (next >= todo.size() || rcs[k] != R3DM_OK) return;
                        const uint32_t w = widths[todo[next]], h = heights[todo[next]];
                        const bool gk = is_gray(todo[next]);
                        const uint32_t bmax = ak_batch_for(c, w, h, std::min(batch, fair));
                        while (next < todo.size() && mine.size() < bmax && widths[todo[next]] == w && heights[todo[next]] == h && is_gray(todo[next]) == gk) mine.push_back(todo[next++]);
                    }
                    const uint32_t B = (uint32_t)mine.size();
                    std::vector<const float*> g(B); std::vector<const unsigned char*> bg(B);
                    std::vector<const char*> fp(B), dp(B); std::vector<uint32_t> nf(B, 0);
                    const bool gk = is_gray(mine[0]);
                    for (uint32_t j = 0; j < B; ++j) { if (gk) g[j] = grays[mine[j]]; else bg[j] = bgrs[mine[j]]; fp[j] = feat_paths[mine[j]]; dp[j] = desc_paths[mine[j]]; }
                    c->feat_sink_ids = mine.data();                    // the sink (if any) is told the caller's indices
                    const int rc = r3dm_extract_features_batch(c, B, gk ? g.data() : nullptr, gk ? nullptr : bg.data(), widths[mine[0]], heights[mine[0]],
                                                               threshold, fp.data(), dp.data(), nf.data());
                    c->feat_sink_ids = nullptr;
                    if (n_features) for (uint32_t j = 0; j < B; ++j) n_features[mine[j]] = nf[j];
                    if (rc != R3DM_OK) { rcs[k] = rc; errs[k] = std::string("image ") + std::to_string(mine[0]) + " (batch of " + std::to_string(B) + "): " + r3dm_last_error(c); }
                }
            } catch (...) { rcs[k] = R3DM_ERR_NOMEM; }          // nothing leaves a worker thread by exception (std::terminate)
        };
        for (uint32_t k = 1; k < concurrency; ++k) th.emplace_back(worker, k);
        worker(0);
        for (auto& t : th) t.join();
        th.clear();
        for (uint32_t k = 0; k < concurrency; ++k)
            if (rcs[k] != R3DM_OK && rc_all == R3DM_OK) {
                rc_all = rcs[k];
                if (err && err_cap) { strncpy(err, errs[k].c_str(), err_cap - 1); err[err_cap - 1] = 0; }
            }
    } catch (...) {
        for (auto& t : th) if (t.joinable()) t.join();          // thread creation failed half way: the started workers finish first
        rc_all = R3DM_ERR_NOMEM;
    }
    return rc_all;
}

}  // namespace

extern "C" int r3dm_set_features_sink(r3dm_ctx* c, r3dm_features_sink sink, void* user)
{
    if (!c) return R3DM_ERR_INVALID;
    c->feat_sink = sink; c->feat_sink_user = sink ? user : nullptr;
    return R3DM_OK;
}

extern "C" int r3dm_set_deferred_feature_files(r3dm_ctx* c, int on)
{
    if (!c) return R3DM_ERR_INVALID;
    c->defer_files = on != 0;
    return on ? R3DM_OK : features_files_join(c);
}

extern "C" int r3dm_set_background_nice(r3dm_ctx* c, int nice_value)
{
    if (!c || nice_value < 0 || nice_value > 19) return R3DM_ERR_INVALID;
    c->background_nice = nice_value;
    return R3DM_OK;
}

extern "C" int r3dm_multi_set_background_nice(r3dm_multi* m, int nice_value)
{
    if (!m) return R3DM_ERR_INVALID;
    int rc = R3DM_OK;
    for (int k = 0; k < r3dm_multi_num_devices(m); ++k) { const int r = r3dm_set_background_nice(r3dm_multi_ctx(m, k), nice_value); if (r != R3DM_OK && rc == R3DM_OK) rc = r; }
    return rc;
}

extern "C" int r3dm_features_files_wait(r3dm_ctx* c)
{
    if (!c) return R3DM_ERR_INVALID;
    return features_files_join(c);
}

extern "C" int r3dm_multi_set_deferred_feature_files(r3dm_multi* m, int on)
{
    if (!m) return R3DM_ERR_INVALID;
    int rc = R3DM_OK;
    for (int k = 0; k < r3dm_multi_num_devices(m); ++k) { const int r = r3dm_set_deferred_feature_files(r3dm_multi_ctx(m, k), on); if (r != R3DM_OK && rc == R3DM_OK) rc = r; }
    return rc;
}

extern "C" int r3dm_multi_features_files_wait(r3dm_multi* m, char* err, size_t err_cap)
{
    if (!m) return R3DM_ERR_INVALID;
    if (err && err_cap) err[0] = 0;
    int rc = R3DM_OK;
    for (int k = 0; k < r3dm_multi_num_devices(m); ++k) {
        r3dm_ctx* c = r3dm_multi_ctx(m, k);
        const int r = features_files_join(c);
        if (r != R3DM_OK && rc == R3DM_OK) { rc = r; if (err && err_cap) { strncpy(err, r3dm_last_error(c), err_cap - 1); err[err_cap - 1] = 0; } }
    }
    return rc;
}

extern "C" int r3dm_multi_set_features_sink(r3dm_multi* m, r3dm_features_sink sink, void* user)
{
    if (!m) return R3DM_ERR_INVALID;
    for (int k = 0; k < r3dm_multi_num_devices(m); ++k) (void)r3dm_set_features_sink(r3dm_multi_ctx(m, k), sink, user);
    return R3DM_OK;
}

extern "C" int r3dm_multi_extract_features(r3dm_multi* m, uint32_t n_images, const float* const* grays, const uint32_t* widths,
                                           const uint32_t* heights, float threshold, const char* const* feat_paths,
                                           const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped,
                                           char* err, size_t err_cap)
{
    return multi_extract_impl(m, n_images, grays, nullptr, widths, heights, threshold, feat_paths, desc_paths, n_features, skipped, 0, err, err_cap);
}

extern "C" int r3dm_multi_extract_features_ex(r3dm_multi* m, uint32_t n_images, const float* const* grays, const unsigned char* const* bgrs,
                                              const uint32_t* widths, const uint32_t* heights, float threshold, const char* const* feat_paths,
                                              const char* const* desc_paths, uint32_t* n_features, uint32_t* skipped, uint32_t batch,
                                              char* err, size_t err_cap)
{
    return multi_extract_impl(m, n_images, grays, bgrs, widths, heights, threshold, feat_paths, desc_paths, n_features, skipped, batch, err, err_cap);
}
