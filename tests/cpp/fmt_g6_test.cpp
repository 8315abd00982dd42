// fmt_g6_test.cpp -- r3dm_fmt::put_g6 against std::to_chars(general, 6) = printf("%g") in the "C" locale: random floats of the
// ranges the .feat writer sees, every decade boundary, ties of the sixth digit, carries, zeros, negatives, values outside the fast
// range.  g++ -O2 -std=c++17 tests/cpp/fmt_g6_test.cpp -o /tmp/fmt_g6_test && /tmp/fmt_g6_test [n]
#include "../../regard3d_amd/csrc/fmt_g6.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

static long bad = 0, seen = 0;
static void check(float v)
{
    char a[64], b[64];
    float back = 0.0f;
    char* ea = r3dm_fmt::put_g6(a, a + 64, v, &back);
    char* eb = r3dm_fmt::put_g_general(b, b + 64, v);
    ++seen;
    {
        float want = v;                                     // (as the writer does: what from_chars leaves on failure is the value itself)
        (void)std::from_chars(b, eb, want);
        if (std::memcmp(&want, &back, 4) != 0 && !(want != want && back != back)) {
            if (bad < 10) std::printf("PARSE-BACK %.9g: %.9g vs %.9g ('%.*s')\n", (double)v, (double)back, (double)want, (int)(eb - b), b);
            ++bad;
        }
    }
    if (ea - a != eb - b || std::memcmp(a, b, (size_t)(ea - a)) != 0) {
        if (bad < 10) std::printf("MISMATCH %.9g: '%.*s' vs '%.*s'\n", (double)v, (int)(ea - a), a, (int)(eb - b), b);
        ++bad;
    }
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 20000000L;
    std::mt19937_64 rng(12345);
    // every float around the decade boundaries and around ties of the sixth digit
    for (double c : {1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, 10.0, 100.0, 1e3, 1e4, 1e5, 1e6, 1e7, 999999.5, 99999.95, 9999.995, 999.9995, 0.9999995, 0.09999995,
                     0.5, 0.25, 0.125, 1234.565, 1234.575, 2.5e-4, 123456.5, 123457.5, 0.000123456, 4000.0, 2999.5, 359.99999}) {
        float f = (float)c;
        for (int i = 0; i < 2000; ++i) f = std::nextafterf(f, 0.0f);
        for (int i = 0; i < 4000; ++i) { check(f); check(-f); f = std::nextafterf(f, 1e30f); }
    }
    for (float v : {0.0f, -0.0f, 1e-30f, 1e30f, INFINITY, -INFINITY, NAN, 1e-45f}) check(v);
    // exact ties: k + 0.5 scaled into every decade (representable ones)
    for (int e = -4; e <= 5; ++e)
        for (long m = 100000; m < 1000000; m += 997) {
            const double t = ((double)m + 0.5) * std::pow(10.0, e - 5);
            check((float)t); check(std::nextafterf((float)t, 0.0f)); check(std::nextafterf((float)t, 1e30f));
        }
    // the writer's ranges: positions 0 .. 4000, scales 0.5 .. 200, angles 0 .. 360, and a log-uniform sweep of 1e-6 .. 1e7
    std::uniform_real_distribution<float> pos(0.0f, 4000.0f), sc(0.5f, 200.0f), ang(0.0f, 360.0f), lg(-6.0f, 7.0f);
    for (long i = 0; i < n; ++i) {
        check(pos(rng)); check(sc(rng)); check(ang(rng)); check(std::pow(10.0f, lg(rng)));
        uint32_t b = (uint32_t)rng(); float f; std::memcpy(&f, &b, 4); check(f);          // any bit pattern
    }
    std::printf("%ld values, %ld mismatches\n", seen, bad);
    return bad ? 1 : 0;
}
