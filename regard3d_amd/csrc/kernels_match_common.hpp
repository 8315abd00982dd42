// kernels_match_common.hpp -- device helpers shared by the matching kernels (kernels_match.hip: f32 tiles; kernels_match_16bit.hip:
// bf16 / f16 nominators; kernels_match_hamming.hip: binary descriptors; kernels_match_exact.hip: exact scans + finalisation):
// the reference metric, the (best, runner-up, bound) lists, the buffer load of a fragment, and the shared tail of every L2 kernel
// (merge the lane halves, re-score in the reference arithmetic, certify, ratio test).
//
// Arithmetic contract (OpenMVG L2<float>; /root/reference/src/R3DComputeMatches.cpp:437-489): distances are the f32
// 4-way-unrolled sum of squared differences, NO fused multiply-add; equal distances -> lowest dataset row.  Every translation unit
// that includes this header is compiled with -ffp-contract=off; fused operations are spelled fmaf() / MFMA.
#pragma once
#include "r3dm_internal.hpp"

namespace r3dm {

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// pointers read out of ImgDev are generic; the hot loops cast them to the global address space so
// the compiler emits global_load (vmcnt only, SGPR base + lane offset) instead of flat_load
typedef const __attribute__((address_space(1))) f32x4* gf4p;
typedef const __attribute__((address_space(1))) float* gf1p;

// ------------------------------------------------------------------------------------------------
// exact squared L2 in the reference's arithmetic (OpenMVG L2<float>): 4-way unrolled, float
// accumulator, ((d0^2 + d1^2) + d2^2) + d3^2 added to the running result, scalar tail, no FMA.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float exact_l2sq(const float* __restrict__ a, const float* __restrict__ b, uint32_t dim)
{
    float result = 0.0f;
    uint32_t k = 0;
    if ((dim & 3u) == 0) {
        const f32x4* a4 = (const f32x4*)a;
        const f32x4* b4 = (const f32x4*)b;
        for (; k < dim; k += 4) {
            const f32x4 x = a4[k >> 2], y = b4[k >> 2];
            const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
            result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        return result;
    }
    for (; k + 3 < dim; k += 4) {
        const float d0 = a[k] - b[k], d1 = a[k + 1] - b[k + 1], d2 = a[k + 2] - b[k + 2], d3 = a[k + 3] - b[k + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (; k < dim; ++k) { const float d0 = a[k] - b[k]; result += d0 * d0; }
    return result;
}

// ------------------------------------------------------------------------------------------------
// running (best, runner-up, bound) list of one query column held by one lane
// ------------------------------------------------------------------------------------------------
struct Top2 {
    float d0, d1, d2;      // d0 <= d1 <= d2 ; d2 = smallest key NOT nominated (certification bound)
    uint32_t i0, i1;
};

__device__ __forceinline__ void top2_init(Top2& s)
{
    s.d0 = s.d1 = s.d2 = R3DM_INF; s.i0 = s.i1 = kNone;
}

__device__ __forceinline__ void top2_push(Top2& s, float key, uint32_t idx)
{
    // branch-free: locals first so every ?: is a plain select (v_cndmask), never control flow
    const float od0 = s.d0, od1 = s.d1, od2 = s.d2;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0;
    const bool c1 = key < od1;
    const uint32_t t1 = c1 ? idx : oi1;
    s.d2 = __builtin_amdgcn_fmed3f(od1, od2, key);     // min(d2, max(d1, key))
    s.d1 = __builtin_amdgcn_fmed3f(od0, od1, key);     // min(d1, max(d0, key))
    s.d0 = __builtin_amdgcn_fmed3f(-R3DM_INF, od0, key);   // min(d0, key) as one v_med3_f32 (no canonicalising v_max)
    s.i1 = c0 ? oi0 : t1;
    s.i0 = c0 ? idx : oi0;
}

// write the verdict for one query: ratio test, optional 2-NN dump
__device__ __forceinline__ void emit_result(const MatchParams& P, uint32_t pair, uint32_t q,
                                            float ea, uint32_t ia, float eb, uint32_t ib)
{
    const size_t o = (size_t)pair * P.q_stride + q;
    P.nn_idx[o] = (ib != kNone && ea < P.ratio_R * eb) ? ia : kNone;
    if (P.knn_idx) {
        P.knn_idx[2 * o] = (int32_t)ia; P.knn_idx[2 * o + 1] = (int32_t)ib;
        P.knn_dist[2 * o] = ea;         P.knn_dist[2 * o + 1] = eb;
    }
}

// 16-byte buffer load: wave-uniform descriptor + SGPR byte offset + per-lane 32-bit offset -- no 64-bit
// per-lane address registers in the hot loop (the pointer form spilled at 256 VGPRs)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// ---- per query: merge the two lane halves, re-score exactly, certify, ratio-test (tail of both L2 kernels).
// dpad = padded descriptor length, bf16_tiles = the keys come from the integer fast path
__device__ __forceinline__ void lex_push(Top2& s, float key, uint32_t idx)
{
    const float od0 = s.d0, od1 = s.d1;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0 || (key == od0 && idx < oi0);
    const bool c1 = key < od1 || (key == od1 && idx < oi1);
    s.d1 = c0 ? od0 : (c1 ? key : od1);
    s.i1 = c0 ? oi0 : (c1 ? idx : oi1);
    s.d0 = c0 ? key : od0;
    s.i0 = c0 ? idx : oi0;
}

// LEX: the lists are exact lexicographic (distance, index) top-2 lists without a bound (l2_knn2_int_kernel)
// SPLIT: the keys come from the split-f16 nominator (l2_knn2_split_kernel) in units of key_inv^-1; a query whose merged
//        top-2 cannot be certified gets a second chance with all four nominees of its two lane halves before it is sent to
//        the exact scan
// lane_key_inv (count tiles, l2_knn2_counts_kernel): the keys of query tile nj are in units of lane_key_inv[nj]^-1, a value per QUERY
//        (both lane halves of a column hold the same one)
template <int NJ, bool LEX = false, bool SPLIT = false>
__device__ __forceinline__ void l2_finish_queries(const MatchParams& P, uint32_t pair, const ImgDev* __restrict__ Ip,
                                                  const ImgDev* __restrict__ Jp, const Top2 (&st)[NJ], uint32_t qt0,
                                                  uint32_t h, uint32_t c, float dpad, bool bf16_tiles,
                                                  float key_inv = 1.0f, float slack_abs = 0.0f, const float* lane_key_inv = nullptr)
{
    const uint32_t nI = Ip->n, nJ = Jp->n, ntJ = Jp->n_tiles;
    const float maxnorm = __uint_as_float(Ip->max_norm_bits);
    const uint32_t dim = Ip->dim;
    // Exactness proof for integer-valued descriptors (e.g. SIFT bins 0..255): when every element of
    // both views is an integer and all partial sums stay below 2^24, the MFMA pass (norm init, fma
    // chain, + ||q||^2) and the reference's sum of squared differences are BOTH exact, hence equal:
    // no rounding slack is needed and only true ties with an un-nominated row need the exact scan.
    // Non-negative data: ||a||^2 <= D mI^2 and the running ||a||^2 - 2 sum(a q) stays within [-2 D mI mJ, D mI^2]; the distance
    // itself is at most D max(mI, mJ)^2.  With negative elements the partial sums reach D mI^2 + 2 D mI mJ and the distance
    // D (mI + mJ)^2, where the reference's own sum starts to round: one bound on the latter covers both.
    const float mI = __uint_as_float(Ip->max_abs_bits), mJ = __uint_as_float(Jp->max_abs_bits);
    const uint32_t fl = Ip->not_integer | Jp->not_integer;              // bit 0: non-integer, bit 1: negative elements
    const bool exact_pair = !SPLIT && (fl & 1u) == 0u &&
                            ((fl & 2u) ? dpad * (mI + mJ) * (mI + mJ) < 16777216.0f
                                       : (2.0f * dpad * mI * mJ < 16777216.0f && dpad * mI * mI < 16777216.0f && dpad * mJ * mJ < 16777216.0f)) &&
                            (!bf16_tiles || (mI <= 256.0f && mJ <= 256.0f));      // bf16 tiles hold the values exactly
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        Top2 s = st[nj];
        if constexpr (SPLIT) {
            const float ki = lane_key_inv ? lane_key_inv[nj] : key_inv;                 // positive scale: order unchanged
            s.d0 *= ki; s.d1 *= ki; s.d2 *= ki;
        }
        const Top2 own = s;                              // this lane half's list (rows 8 qd + 4 h + k of every tile)
        // partner half (same query column, the other 16 rows of every tile)
        const float pd0 = __shfl_xor(s.d0, 32), pd1 = __shfl_xor(s.d1, 32), pd2 = __shfl_xor(s.d2, 32);
        const uint32_t pi0 = __shfl_xor(s.i0, 32), pi1 = __shfl_xor(s.i1, 32);
        if constexpr (LEX) {
            lex_push(s, pd0, pi0);
            lex_push(s, pd1, pi1);
            s.d2 = R3DM_INF;                               // nothing un-nominated can tie or beat an exact top-2
        } else {
            top2_push(s, pd0, pi0);
            top2_push(s, pd1, pi1);
            s.d2 = fminf(s.d2, pd2);
        }
        // make both halves agree on the nominated pair (lane c's view)
        const uint32_t ci0 = __shfl(s.i0, (int)c), ci1 = __shfl(s.i1, (int)c);
        const float bound = __shfl(s.d2, (int)c);

        const uint32_t qt = qt0 + nj;
        const uint32_t q = qt * 32u + c;
        const bool valid = (qt < ntJ) && (q < nJ);
        const uint32_t cand = h ? ci1 : ci0;
        const float cd0 = __shfl(s.d0, (int)c), cd1 = __shfl(s.d1, (int)c);     // (both shuffles outside the lane-dependent select)
        const float ck = h ? cd1 : cd0;                                         // MFMA key ||a||^2 - 2 a.b of this lane's nominee
        float e = R3DM_INF;
        if (valid && cand != kNone) {
            // exact pairs (proof above): key + ||q||^2 IS the reference distance, bit for bit -- no need to fetch the two
            // nominated rows again (that re-read was 60 % of the kernel's HBM-side traffic: 3 x 512 B per query).
            // Otherwise re-score in the reference's summation order.
            if (exact_pair) e = ck + Jp->norms[q];
            else e = exact_l2sq(Ip->rows + (size_t)cand * dim, Jp->rows + (size_t)q * dim, dim);
        }
        const float eo = __shfl_xor(e, 32);
        float ea = h ? eo : e, eb = h ? e : eo;          // ea <-> ci0, eb <-> ci1
        uint32_t ia = ci0, ib = ci1;
        if (eb < ea || (eb == ea && ib < ia)) { const float tf = ea; ea = eb; eb = tf; const uint32_t tu = ia; ia = ib; ib = tu; }
        // certification (evaluated identically by both lane halves of a query)
        const float nb = valid ? Jp->norms[q] : 0.0f;
        const float slack = exact_pair ? 0.0f : P.err_scale * (maxnorm + nb) + slack_abs;
        const float A3 = bound + nb;                        // distance of the best un-nominated row (exact if exact_pair)
        // certified: every un-nominated row is strictly farther than the runner-up.  With exact
        // arithmetic a runner-up that merely TIES an un-nominated row still fixes the best row (ea < eb)
        // and the runner-up DISTANCE, which is all the ratio test needs; only the raw 2-NN dump
        // (r3dm_knn2) needs the tie's index resolved by the exact scan.
        bool certified = (eb < A3 - slack) || (exact_pair && P.knn_idx == nullptr && ea < eb && eb <= A3);
        if (bf16_tiles && !exact_pair) certified = false;      // bf16 keys of a non-exact pair mean nothing: exact scan
        // Match mode only needs the VERDICT of the ratio test.  The two re-scored nominees bound the true runner-up distance from
        // above (d2 <= eb: two rows are no farther than eb) and, with the un-nominated rows' lower bound L = bound + ||q||^2 -
        // slack, the true best distance from below (d1 >= min(ea, L)).  If min(ea, L) >= R eb then d1 >= R d2 whatever the exact
        // top-2 is: the query has no match, exactly as the exact scan would find -- and that is the fate of nearly every
        // uncertifiable query (descriptors without a counterpart sit at almost equal distances from their nearest rows).
        bool no_match = false;
        if (!certified && !exact_pair && !bf16_tiles && P.knn_idx == nullptr && valid && nI >= 2 && ib != kNone) {
            float L = A3 - slack;
            L -= fabsf(L) * 9.5367431640625e-07f;            // 2^-20: the float evaluation of L itself
            no_match = fminf(ea, L) >= P.ratio_R * eb;
        }
        if constexpr (SPLIT) {
            // second chance: the two lane halves of a query nominated up to four rows between them.  Re-score all four in the
            // reference arithmetic and certify against the smallest key that NONE of them holds (each half's third key):
            // the gap from the runner-up to the fifth-best row is what has to exceed the slack now, not the gap to the third.
            const bool need = valid && nI >= 2 && !certified && !no_match;
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                float f0 = R3DM_INF, f1 = R3DM_INF;
                if (need) {
                    const float* qrow = Jp->rows + (size_t)q * dim;
                    if (own.i0 != kNone) f0 = exact_l2sq(Ip->rows + (size_t)own.i0 * dim, qrow, dim);
                    if (own.i1 != kNone) f1 = exact_l2sq(Ip->rows + (size_t)own.i1 * dim, qrow, dim);
                }
                Top2 m4; top2_init(m4);
                lex_push(m4, f0, own.i0); lex_push(m4, f1, own.i1);
                const float g0 = __shfl_xor(f0, 32), g1 = __shfl_xor(f1, 32);
                const uint32_t j0 = __shfl_xor(own.i0, 32), j1 = __shfl_xor(own.i1, 32);
                lex_push(m4, g0, j0); lex_push(m4, g1, j1);
                const float bound4 = fminf(own.d2, __shfl_xor(own.d2, 32));
                if (need && m4.i1 != kNone && m4.d1 < (bound4 + nb) - slack) {
                    ea = m4.d0; ia = m4.i0; eb = m4.d1; ib = m4.i1; certified = true;
                }
            }
        }
        if (valid && h == 0) {
            if (nI < 2) {
                emit_result(P, pair, q, R3DM_INF, kNone, R3DM_INF, kNone);
            } else if (certified) {
                emit_result(P, pair, q, ea, ia, eb, ib);
            } else if (no_match) {
                P.nn_idx[(size_t)pair * P.q_stride + q] = kNone;
            } else {
                P.nn_idx[(size_t)pair * P.q_stride + q] = kFallback;
                const uint32_t pos = atomicAdd(P.fb_cnt + pair, 1u);
                atomicAdd(P.fb_total, 1u);
                if (pos < kFbPerPair) P.fb_q[(size_t)pair * kFbPerPair + pos] = q;
                else atomicAdd(P.fb_total + 1, 1u);
            }
        }
    }
}

// ---- helpers of the exact-key kernels (bf16 / i8 tiles): keys ARE distances' ranks, lists hold (best, runner-up) only
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__device__ __forceinline__ void tope_push(Top2& s, float key, uint32_t idx)
{
    const float od0 = s.d0, od1 = s.d1;
    const uint32_t oi0 = s.i0, oi1 = s.i1;
    const bool c0 = key < od0;
    const bool c1 = key < od1;
    const uint32_t t1 = c1 ? idx : oi1;
    s.d1 = __builtin_amdgcn_fmed3f(od0, od1, key);
    s.d0 = vmin2(od0, key);
    s.i1 = c0 ? oi0 : t1;
    s.i0 = c0 ? idx : oi0;
}

}  // namespace r3dm
