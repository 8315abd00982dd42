// level_head_emul.cpp -- the level-head marching kernel's control flow (regard3d_amd/csrc/level_head.inc) on the CPU, a 64-float
// struct standing in for a wavefront and a float array for its LDS, against the four per-pixel passes it replaces (Gaussian row +
// column pass with clamped borders, scaled Scharr derivatives with reflected borders, determinant, Scharr 3x3 + conductivity --
// the border forms of ak_gauss_rows_px / ak_gauss_cols_px / ak_sderiv_at / ak_scharr_g2_px in kernels_akaze.hip).
//   g++ -O1 -ffp-contract=off -o level_head_emul level_head_emul.cpp && ./level_head_emul
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#define FED_HD inline
#include "../../regard3d_amd/csrc/level_head.inc"

struct V64 { float v[64]; };
struct M64 { bool v[64]; };
struct I64 { int v[64]; };
struct EmuOps {
    using VF = V64; using VM = M64; using VI = I64; using Lds = float*;
    static VF zero() { VF r; for (int i = 0; i < 64; ++i) r.v[i] = 0.f; return r; }
    static VI lane_plus(int b) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = b + i; return r; }
    static VM gt(VI a, int b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] > b; return r; }
    static VM lt(VI a, int b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] < b; return r; }
    static VM land(VM a, VM b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] && b.v[i]; return r; }
    static VM core_lanes(int k) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = i >= k && i < 64 - k; return r; }
    static VI clampi(VI a, int lo, int hi) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = lh_clampi(a.v[i], lo, hi); return r; }
    static VI col_clamp(VI x, int off, int w, int x_first) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = lh_clampi(lh_clampi(x.v[i] + off, 0, w - 1) - x_first, -16, 79); return r; }
    static VI col_refl(VI x, int off, int w, int x_first) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = lh_clampi(lh_refl101(x.v[i] + off, w) - x_first, -16, 79); return r; }
    static VF load(const float* p, int row, int w, VI xc) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = p[(size_t)row * w + xc.v[i]]; return r; }
    static void store(float* p, int row, int w, VI x, VF v, VM m) { for (int i = 0; i < 64; ++i) if (m.v[i]) p[(size_t)row * w + x.v[i]] = v.v[i]; }
    static void lds_store(Lds l, int at, VF v) { for (int i = 0; i < 64; ++i) l[at + i] = v.v[i]; }
    static VF lds_load(Lds l, int at, VI idx) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = l[at + idx.v[i]]; return r; }
    static VF lds_load_off(Lds l, int at, int off) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = l[at + i + off]; return r; }
    static void wave_sync() {}
    static VF add(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
    static VF adds(VF a, float b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] + b; return r; }
    static VF sub(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
    static VF mul(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
    static VF muls(VF a, float b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] * b; return r; }
    static VF neg(VF a) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = -a.v[i]; return r; }
    static VF rcp_div(float a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a / b.v[i]; return r; }
};

// ---- the per-pixel reference passes (border forms of kernels_akaze.hip)
static void ref_gauss5(const float* src, float* dst, int w, int h, const float* k)
{
    std::vector<float> tmp((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* S = src + (size_t)y * w;
            tmp[(size_t)y * w + x] = S[x] * k[2] + (S[lh_clampi(x - 1, 0, w - 1)] + S[lh_clampi(x + 1, 0, w - 1)]) * k[3]
                                     + (S[lh_clampi(x - 2, 0, w - 1)] + S[lh_clampi(x + 2, 0, w - 1)]) * k[4];
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[2] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= 2; ++j)
                s += k[2 + j] * (tmp[(size_t)lh_clampi(y + j, 0, h - 1) * w + x] + tmp[(size_t)lh_clampi(y - j, 0, h - 1) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
}
static float ref_sd_row(const float* S, int x, int w, int s, int dx)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const float a = S[lh_refl101(x - s, w)], b = S[lh_refl101(x + s, w)];
    if (dx) return (-a) + b;
    if (s == 2) return S[x] * kc + (a + b) * norm;
    return (norm * a + kc * S[x]) + norm * b;
}
static float ref_sd_at(const float* src, int x, int y, int w, int h, int s, int dx)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (wgt + 2.0f));
    const float kc = wgt * norm;
    const float u = ref_sd_row(src + (size_t)lh_refl101(y - s, h) * w, x, w, s, dx);
    const float d = ref_sd_row(src + (size_t)lh_refl101(y + s, h) * w, x, w, s, dx);
    if (dx) { const float c = ref_sd_row(src + (size_t)y * w, x, w, s, dx); return kc * c + norm * (d + u); }
    return d - u;
}
static float ref_g2(const float* src, int w, int h, int x, int y, float inv_k2)
{
    const int xl = lh_refl101(x - 1, w), xr = lh_refl101(x + 1, w);
    const float* Su = src + (size_t)lh_refl101(y - 1, h) * w;
    const float* Sc = src + (size_t)y * w;
    const float* Sd = src + (size_t)lh_refl101(y + 1, h) * w;
    const float au = Su[xl], cu = Su[x], bu = Su[xr], ac = Sc[xl], bc = Sc[xr], ad = Sd[xl], cd = Sd[x], bd = Sd[xr];
    const float rdu = bu - au, rdc = bc - ac, rdd = bd - ad;
    const float rsu = cu * 10.0f + (au + bu) * 3.0f, rsd = cd * 10.0f + (ad + bd) * 3.0f;
    const float lx = (rdu + rdd) * 3.0f + rdc * 10.0f;
    const float ly = rsd - rsu;
    return 1.0f / (1.0f + ((lx * lx + ly * ly) * inv_k2));
}

static uint32_t rng_state = 777;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.0f / 16777216.0f); }

template <int S>
static int check(int w, int h, int R)
{
    using G = LevelHeadGeom<S>;
    const size_t n = (size_t)w * h;
    std::vector<float> src(n), sm(n), lx(n), ly(n), det(n), fl(n);
    for (auto& v : src) v = frand();
    const float k5[5] = {0.054488685f, 0.24420135f, 0.40261996f, 0.24420135f, 0.054488685f};
    const float inv_k2 = 1.0f / (0.02f * 0.02f);
    ref_gauss5(src.data(), sm.data(), w, h, k5);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            lx[(size_t)y * w + x] = ref_sd_at(sm.data(), x, y, w, h, S, 1);
            ly[(size_t)y * w + x] = ref_sd_at(sm.data(), x, y, w, h, S, 0);
            fl[(size_t)y * w + x] = ref_g2(sm.data(), w, h, x, y, inv_k2);
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float lxx = ref_sd_at(lx.data(), x, y, w, h, S, 1), lxy = ref_sd_at(lx.data(), x, y, w, h, S, 0), lyy = ref_sd_at(ly.data(), x, y, w, h, S, 0);
            det[(size_t)y * w + x] = lxx * lyy - lxy * lxy;
        }
    std::vector<float> mlx(n, -7.f), mly(n, -7.f), mdet(n, -7.f), mfl(n, -7.f), lds(G::FLOATS);
    const int n_strips = (w + G::VW - 1) / G::VW;
    for (int y0 = 0; y0 < h; y0 += R)
        for (int st = 0; st < n_strips; ++st) {
            for (auto& v : lds) v = 3.0e33f * (frand() - 0.5f);                    // whatever a previous workgroup left there
            const int x_first = st * G::VW - G::H, y1 = y0 + R < h ? y0 + R : h;
            const bool edge = x_first < 0 || x_first + 63 > w - 1;
            if (edge) level_head_strip<S, true, EmuOps>(src.data(), mlx.data(), mly.data(), mdet.data(), mfl.data(), w, h, k5, inv_k2, x_first, y0, y1, lds.data());
            else level_head_strip<S, false, EmuOps>(src.data(), mlx.data(), mly.data(), mdet.data(), mfl.data(), w, h, k5, inv_k2, x_first, y0, y1, lds.data());
        }
    int bad = 0;
    const char* names[4] = {"Lx", "Ly", "Ldet", "flow"};
    const std::vector<float>* A[4] = {&lx, &ly, &det, &fl};
    const std::vector<float>* B[4] = {&mlx, &mly, &mdet, &mfl};
    for (int p = 0; p < 4; ++p)
        if (memcmp(A[p]->data(), B[p]->data(), n * 4)) {
            size_t cnt = 0, first = (size_t)-1;
            for (size_t i = 0; i < n; ++i) if (memcmp(&(*A[p])[i], &(*B[p])[i], 4)) { if (first == (size_t)-1) first = i; ++cnt; }
            printf("S=%d %dx%d R=%d %s: %zu cells differ, first at (x %zu, y %zu): %.9g vs %.9g\n", S, w, h, R, names[p], cnt, first % w, first / w, (*A[p])[first], (*B[p])[first]);
            ++bad;
        }
    return bad;
}

int main()
{
    int bad = 0;
    const int sizes[][2] = {{200, 90}, {64, 64}, {131, 37}, {57, 41}, {45, 120}, {100, 19}, {300, 21}};
    for (auto& s : sizes)
        for (int R : {7, 16, 64, 256}) { bad += check<2>(s[0], s[1], R); bad += check<3>(s[0], s[1], R); bad += check<4>(s[0], s[1], R); }
    printf(bad ? "DIFFERENT (%d planes)\n" : "identical\n", bad);
    return bad ? 1 : 0;
}
