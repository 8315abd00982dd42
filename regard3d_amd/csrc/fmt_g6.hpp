// fmt_g6.hpp -- "%g" of a float (printf's default: six significant digits, shortest of fixed / exponent form, trailing zeros
// dropped) without going through the general-purpose std::to_chars, for the .feat writer: 28 k keypoints x 4 numbers per image at
// ~80 ns each were 9 ms of every batch's critical path on the host.
//
// Exactness.  A float has 24 significant bits; 10^k for k <= 10 has at most 24 (5^10 < 2^24); their product has at most 48 and is
// therefore EXACT in a double.  For 1e-4 <= |v| < 1e6 the six significant digits are round-half-even(|v| * 10^(5 - e)) with
// e = floor(log10 |v|) found by exact comparisons against powers of ten -- the same digits printf derives from the exact binary
// value.  For |v| < 1 the scale 10^(5 - e) exceeds 10^10 only below 1e-5, outside the range; between 1e-4 and 1 it is 10^6 ..
// 10^9: exact as well.  Everything else (0 is handled; negative zero, NaN, infinities, exponent forms) goes to std::to_chars, as
// before.  tests/cpp/fmt_g6_test.cpp compares the two on 10^8 values, ties and carries included.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace r3dm_fmt {

inline char* put_g_general(char* p, char* end, float v)
{
    const std::to_chars_result r = std::to_chars(p, end, v, std::chars_format::general, 6);
    return r.ptr;
}

// the text just written, read back as a float (what a reader of the file gets)
inline float parse_back(const char* b, const char* e, float fallback)
{
    float r = fallback;
    (void)std::from_chars(b, e, r);
    return r;
}

// writes at most 16 characters; returns the end.  parsed (optional): the value a reader of the text gets -- for the fast range the
// decimal is dig * 10^(e - 5) with dig < 2^24 and a power of ten that is exact in a float (<= 10^10), so ONE float multiplication
// or division is the correctly rounded result (Clinger's fast path), no parsing.
inline char* put_g6(char* p, char* end, float v, float* parsed = nullptr)
{
    if (end - p < 16) { char* q = put_g_general(p, end, v); if (parsed) *parsed = parse_back(p, q, v); return q; }
    uint32_t bits; std::memcpy(&bits, &v, 4);
    const bool neg = (bits >> 31) != 0;
    const float a = neg ? -v : v;
    if (bits == 0u) { *p++ = '0'; if (parsed) *parsed = 0.0f; return p; }
    if (!(a >= 1e-4f && a < 1e6f)) { char* q = put_g_general(p, end, v); if (parsed) *parsed = parse_back(p, q, v); return q; }   // (also NaN / inf / -0 / denormals)
    // e = floor(log10 a) by exact comparisons (the float constants below are the floats printf itself would compare the value with:
    // a power of ten up to 1e5 is exact in a float; 1e-1 .. 1e-4 are not, so those decades are decided in double on the exact value)
    const double d = (double)a;
    int e;
    if (d >= 1.0) e = d < 10.0 ? 0 : d < 100.0 ? 1 : d < 1000.0 ? 2 : d < 10000.0 ? 3 : d < 100000.0 ? 4 : 5;
    else {
        // d < 1: compare d * 10^k with 1 (products exact)
        e = d * 10.0 >= 1.0 ? -1 : d * 100.0 >= 1.0 ? -2 : d * 1000.0 >= 1.0 ? -3 : -4;
    }
    static const double p10[] = {1.0, 10.0, 100.0, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10};
    const int k = 5 - e;                                                    // 0 .. 9
    const double scaled = d * p10[k];                                       // exact (see above)
    int64_t dig = (int64_t)std::nearbyint(scaled);                          // round-half-even in the default rounding mode
    if (dig >= 1000000) { dig = 100000; e += 1; }                           // 999999.5 .. -> 1.00000e(e + 1)
    if (e >= 6) { char* q = put_g_general(p, end, v); if (parsed) *parsed = parse_back(p, q, v); return q; }
    if (parsed) {
        static const float p10f[] = {1.0f, 10.0f, 100.0f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
        const float r = e >= 5 ? (float)dig * p10f[e - 5] : (float)dig / p10f[5 - e];
        *parsed = neg ? -r : r;
    }
    if (neg) *p++ = '-';
    char ds[6];
    for (int i = 5; i >= 0; --i) { ds[i] = (char)('0' + dig % 10); dig /= 10; }
    int last = 5;
    while (last > 0 && ds[last] == '0') --last;                             // trailing zeros go (at least one digit stays)
    if (e >= 0) {
        // digits 0 .. e before the point
        for (int i = 0; i <= e; ++i) *p++ = ds[i];
        if (last > e) { *p++ = '.'; for (int i = e + 1; i <= last; ++i) *p++ = ds[i]; }
    } else {
        *p++ = '0'; *p++ = '.';
        for (int i = -1; i > e; --i) *p++ = '0';
        for (int i = 0; i <= last; ++i) *p++ = ds[i];
    }
    return p;
}

}  // namespace r3dm_fmt
