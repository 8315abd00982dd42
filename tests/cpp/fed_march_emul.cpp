// fed_march_emul.cpp -- the register-marching FED kernel's control flow (regard3d_amd/csrc/fed_march.inc) run on the CPU with a
// 64-float struct in place of a wavefront, against K plain FED steps (the per-pixel border rule of ak_fed_px_border =
// nld_step_scalarV2 + the update).  Exit code 0 and "identical" when every configuration matches bit for bit.
//   g++ -O1 -ffp-contract=off -o fed_march_emul fed_march_emul.cpp && ./fed_march_emul
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

#define FED_HD inline
#include "../../regard3d_amd/csrc/fed_march.inc"

struct V64 { float v[64]; };
struct M64 { bool v[64]; };
struct I64 { int v[64]; };
struct EmuOps {
    using VF = V64; using VM = M64; using VI = I64;
    static VF zero() { VF r; for (int i = 0; i < 64; ++i) r.v[i] = 0.f; return r; }
    static VI lane_plus(int b) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = b + i; return r; }
    static VM gt(VI a, int b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] > b; return r; }
    static VM lt(VI a, int b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] < b; return r; }
    static VM land(VM a, VM b) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] && b.v[i]; return r; }
    static VM core_lanes(int k) { VM r; for (int i = 0; i < 64; ++i) r.v[i] = i >= k && i < 64 - k; return r; }
    static VI clampi(VI a, int lo, int hi) { VI r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] < lo ? lo : (a.v[i] > hi ? hi : a.v[i]); return r; }
    static VF load(const float* p, int row, int w, VI xc) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = p[(size_t)row * w + xc.v[i]]; return r; }
    static void store(float* p, int row, int w, VI x, VF v, VM m) { for (int i = 0; i < 64; ++i) if (m.v[i]) p[(size_t)row * w + x.v[i]] = v.v[i]; }
    static VF add(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
    static VF sub(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
    static VF mul(VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
    static VF muls(VF a, float b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] * b; return r; }
    static VF sel(VM m, VF a, VF b) { VF r; for (int i = 0; i < 64; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    // garbage on purpose where a wavefront has no neighbour: whatever the kernel reads there must not reach a stored value
    static VF shr(VF a) { VF r; r.v[0] = 1.0e30f; for (int i = 1; i < 64; ++i) r.v[i] = a.v[i - 1]; return r; }
    static VF shl(VF a) { VF r; r.v[63] = -1.0e30f; for (int i = 0; i < 63; ++i) r.v[i] = a.v[i + 1]; return r; }
};

// one plain FED step (ak_fed_px_border)
static void fed_step(const float* Lt, const float* Lf, float* out, int w, int h, float step_size)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t p = (size_t)y * w + x;
            const bool has_l = x > 0, has_r = x < w - 1, has_a = y > 0, has_b = y < h - 1;
            const float tc = Lt[p], fc = Lf[p];
            float v;
            if (!has_a) {
                if (!has_l || !has_r) v = 0.0f;
                else v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc);
            } else if (!has_b) {
                if (!has_l || !has_r) v = 0.0f;
                else v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
            } else if (!has_l) {
                v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
            } else if (!has_r) {
                v = (fc + Lf[p - 1]) * (Lt[p - 1] - tc) + (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
            } else {
                v = (fc + Lf[p + 1]) * (Lt[p + 1] - tc) + (fc + Lf[p - 1]) * (Lt[p - 1] - tc) +
                    (fc + Lf[p + w]) * (Lt[p + w] - tc) + (fc + Lf[p - w]) * (Lt[p - w] - tc);
            }
            out[p] = tc + v * 0.5f * step_size;
        }
}

template <int K>
static void march_image(const float* Lt, const float* Lf, float* out, int w, int h, const float* tau, int R)
{
    const int VW = 64 - 2 * K;
    const int n_strips = (w + VW - 1) / VW;
    for (int y0 = 0; y0 < h; y0 += R)
        for (int st = 0; st < n_strips; ++st) {
            const int x_first = st * VW - K, y1 = y0 + R < h ? y0 + R : h;
            const bool edge = x_first < 1 || x_first + 63 > w - 2;
            if (edge) fed_march_strip<K, true, EmuOps>(Lt, Lf, out, w, h, tau, x_first, y0, y1);
            else fed_march_strip<K, false, EmuOps>(Lt, Lf, out, w, h, tau, x_first, y0, y1);
        }
}

static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.0f / 16777216.0f); }

template <int K>
static int check(int w, int h, int R)
{
    std::vector<float> Lt((size_t)w * h), Lf((size_t)w * h), a, b, m((size_t)w * h, -7.0f);
    for (auto& v : Lt) v = frand();
    for (auto& v : Lf) v = 0.05f + frand();
    float tau[K];
    for (int k = 0; k < K; ++k) tau[k] = 0.05f + 0.4f * frand();
    a = Lt; b.resize(a.size());
    for (int k = 0; k < K; ++k) { fed_step(a.data(), Lf.data(), b.data(), w, h, tau[k]); a.swap(b); }
    march_image<K>(Lt.data(), Lf.data(), m.data(), w, h, tau, R);
    const bool same = memcmp(a.data(), m.data(), a.size() * 4) == 0;
    if (!same) {
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < a.size(); ++i) if (memcmp(&a[i], &m[i], 4)) { if (first == (size_t)-1) first = i; ++bad; }
        printf("K=%d %dx%d R=%d: %zu cells differ, first at (x %zu, y %zu): %.9g vs %.9g\n", K, w, h, R, bad, first % w, first / w, a[first], m[first]);
    }
    return same ? 0 : 1;
}

int main()
{
    int bad = 0;
    const int sizes[][2] = {{200, 90}, {57, 33}, {64, 64}, {131, 17}, {56, 40}, {113, 7}, {5, 5}, {3, 3}, {300, 3}};
    for (auto& s : sizes)
        for (int R : {1, 5, 16, 64}) {
            bad += check<1>(s[0], s[1], R); bad += check<2>(s[0], s[1], R); bad += check<3>(s[0], s[1], R);
            bad += check<4>(s[0], s[1], R); bad += check<5>(s[0], s[1], R); bad += check<6>(s[0], s[1], R);
        }
    printf(bad ? "DIFFERENT (%d configurations)\n" : "identical\n", bad);
    return bad ? 1 : 0;
}
