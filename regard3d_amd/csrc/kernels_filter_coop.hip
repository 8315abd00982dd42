// kernels_filter_coop.hip -- AC-RANSAC of a LONG pair spread over several workgroups (round 4).
//
// The one-workgroup-per-pair kernel (kernels_filter.hip) binds a pair to one CU: a pair of 10-20 k putative matches is bound by
// that CU's f64 rate, and a collection of few, long pairs (what Regard3D produces: tens of photographs) leaves most of the chip
// idle.  Here a pair with more than FilterParams::coop_min_m matches is evaluated by G workgroups:
//
//   * ACRANSAC stays sequential where it is sequential (the reference's iteration order, the pool that shrinks on every
//     meaningful improvement, the budget that is cut once -- SURVEY.md A.5); what is spread is the part that is not: the
//     residuals of a BATCH of up to 32 models (whole iterations of the current chunk of minimal samples) over the pair's matches.
//     Workgroup s takes the s-th slice of the match list, loads every point once for all models of the batch, and leaves a
//     residual histogram (the 1024 bins of the sort-skipping bound) and a count per model in its global slot.
//   * The workgroup that arrives last (an atomic ticket) merges the slots and walks the batch in the reference's order.  The NFA
//     bound of a model comes from the merged histogram alone; only a model whose bound can beat the best NFA so far -- a few per
//     cent -- is evaluated in full by that one workgroup (residuals of all matches, sort, NFA scan: the routines of the
//     one-workgroup kernel), so every decision is the one the sequential algorithm takes: same inlier sets, same models, same
//     iteration and model counts.  Models behind a pool change are discarded like the rest of a chunk always was; the batch size
//     starts small after a pool change and doubles (12, 24, 32) while nothing changes.
//   * Workgroups are not bound to pairs.  The kernel is a pool of workers over a queue of (pair, slice) tasks in global memory; the
//     last arriver of a batch becomes the pair's leader, prepares the next batch (drawing and solving a new chunk when the old one
//     is used up or void), publishes G - 1 tasks and takes slice 0 itself.  No workgroup ever waits for a particular other
//     workgroup to be scheduled -- only for the queue -- so nothing depends on co-residency, and F, E and H kernels can share the
//     device.  A worker that finds the queue empty while fewer tasks can exist than workers are alive retires, leaving its CU to
//     the other kernels.
//   * The bound: NFA_k >= loge0 + la(bin of the k-th residual) (k - SS) + T[k], T[k] = logc_n[k] + logc_k[k] (float tables).
//     T*(k) = log10(m! / ((m-k)! SS! (k-SS)!)) is concave in k, so over the k range of a bin la_b (k - SS) + T*(k) takes its minimum
//     at an end of the range: two evaluations per non-empty bin instead of one per k (the walk over k was half the evaluation time
//     of a long pair).  eps_T = max_k |T[k] - T*(k)|, computed per pair at start-up, makes the bound rigorous for the float
//     tables: bound = min over bins and ends - eps_T - 1e-6.
//
// Exchange through global memory between workgroups on different XCDs (eight L2s): plain stores, one agent-scope fence
// (__threadfence: L2 write-back / invalidate) per workgroup and BATCH on each side of the ticket or the queue -- not per model.
#define R3DM_FILTER_DEVICE_ONLY 1
#include "kernels_filter.hip"

namespace r3dm {

constexpr int kCoopNT = 512;                // threads per worker
constexpr uint32_t kCoopNW = kCoopNT / 64;
constexpr uint32_t kCoopInitSlice = 31u;    // slice code of a pair's first task
constexpr uint32_t kCoopNoTask = 0xFFFFFFFFu;
template <int KIND> struct CoopChunk { static constexpr int n = (KIND == 2) ? 2 * kE5Samples : kChunk; };   // E: 32 groups of 16 lanes

// State of one pair.  Lives in global memory (FilterParams::coop_state) between batches, in LDS while a leader works on it.
struct CoopS {
    double minNFA, errorMax, bestF[9];
    double eps_T;                          // max |T - T*| of the pair's tables
    double kinv[18];
    double red_v[8];
    double bnd[kCoopB];                    // NFA bound of model j of the batch (without loge0 / eps)
    uint32_t tot[kCoopB];                  // matches within the residual bound, model j of the batch
    uint32_t nIter, reserve, iter, pool_size, n_inl, acMode, n_models, iters_done;
    uint32_t chunk_iter0, chunk_n, chunk_c, chunk_valid;      // the chunk of solved samples: first iteration, iterations, next to decide
    uint32_t b_c0, b_c1, b_n, b_cap;       // the batch: iterations [c0, c1) of the chunk, models, slow-start capacity
    uint32_t cnt, flag, G, slice_len;
    uint32_t wave_cnt[8], red_k[8];
    uint32_t nm[64];                       // models of every hypothesis of the chunk
    uint8_t  bj_c[kCoopB], bj_k[kCoopB];   // model j of the batch = model bj_k[j] of hypothesis bj_c[j]
    uint32_t copy_end;                     // ---- fields below are not part of the LDS <-> global copy
    uint32_t arrived;                      // ticket of the batch in flight
};
static_assert(sizeof(CoopS) <= kCoopStateBytes, "CoopS must fit its global slot");
static_assert(sizeof(CoopS) <= 2048, "CoopS must fit the LDS header");
constexpr int kCoopHdr = 4608;             // LDS: [CoopS | batch models 32 x 9 doubles at 2048 | slice counts at 4352] [region R at 4608]
constexpr int kCoopBmOff = 2048, kCoopCntOff = 2048 + kCoopB * 72;
static_assert(kCoopCntOff + kCoopB * 4 <= kCoopHdr, "LDS header");

static inline size_t coop_lds_bytes_(int model_kind)
{
    const size_t hist = (size_t)kCoopB * 512 * 4;                                   // 1024 u16 bins per model
    const size_t work = model_kind == 2 ? (size_t)CoopChunk<2>::n * kE5Stride * 8 : 0;   // 5-point workspaces
    return (size_t)kCoopHdr + (hist > work ? hist : work);
}

// ---- the task queue (bounded ring, tickets): header words [head, tail, done, potential, active, n_pairs, cap - 1, -] + seq[cap] + data[cap]
#define QLOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ void coop_push(uint32_t* q, uint32_t v)
{
    const uint32_t mask = q[6];
    uint32_t* seq = q + 8;
    uint32_t* data = seq + mask + 1;
    const uint32_t t = atomicAdd(&q[1], 1u);
    const uint32_t slot = t & mask;
    // (capacity >= the tasks that can be outstanding: the slot is free; the wait is defensive)
    while (__hip_atomic_load(&seq[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != t) __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(&data[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&seq[slot], t + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool coop_pop(uint32_t* q, uint32_t& v)
{
    const uint32_t mask = q[6];
    uint32_t* seq = q + 8;
    uint32_t* data = seq + mask + 1;
    for (;;) {
        const uint32_t h = QLOAD(&q[0]);
        const uint32_t slot = h & mask;
        if (__hip_atomic_load(&seq[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != h + 1u) return false;   // empty, or its producer is still writing
        if (atomicCAS(&q[0], h, h + 1u) == h) {
            v = QLOAD(&data[slot]);
            __hip_atomic_store(&seq[slot], h + mask + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);        // free for the next lap
            return true;
        }
    }
}

// what a worker needs to know about the pair of its task
template <int KIND>
struct CoopCtx {
    uint32_t item, cp, m, G, slice_len, hoff;
    uint2 id;
    double s1, s2, t1x, t1y, t2x, t2y, logalpha0, maxThreshold, loge0;
    long long hist_base;
    double* pt; uint32_t* pool; uint32_t* inl; float* logc_n; double* tstar; double* la_tab;
    double* models; double* bm; CoopS* gs;
    unsigned long long* keys; uint32_t* sidx;
    const ImgDev* Ip; const ImgDev* Jp; const r3dm_match* mm;
};

template <int KIND>
__device__ __forceinline__ void coop_ctx(const FilterParams& P, uint32_t cp, CoopCtx<KIND>& C)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr double MAXM = (KIND == 0) ? 3.0 : (KIND == 1 ? 1.0 : 10.0);
    constexpr int MS = (KIND == 2) ? 90 : 27;
    C.cp = cp; C.item = P.coop_items[cp];
    const uint64_t begin = P.offsets[2 * C.item], end = P.offsets[2 * C.item + 1];
    C.m = (uint32_t)(end - begin);
    C.G = P.coop_G[cp]; C.slice_len = P.coop_slice[cp]; C.hoff = P.coop_hoff[cp];
    const uint2 sl = P.pairs[C.item];
    C.id = P.pair_ids[C.item];
    C.Ip = P.imgs + sl.x; C.Jp = P.imgs + sl.y; C.mm = P.matches + begin;
    const uint64_t so = P.soff[C.item];
    C.pt = P.pts_scratch + 4 * so; C.pool = P.pool_scratch + so; C.inl = P.inl_idx + so; C.logc_n = P.scratch_logc + so;
    C.tstar = P.coop_tstar + so; C.la_tab = P.coop_la + (size_t)cp * kHistBins;
    C.models = P.coop_models + (size_t)cp * CoopChunk<KIND>::n * MS;
    C.bm = P.coop_bm + (size_t)cp * kCoopB * 9;
    C.gs = reinterpret_cast<CoopS*>(P.coop_state + (size_t)cp * kCoopStateBytes);
    C.keys = P.spill_keys + P.spill_off[C.item]; C.sidx = P.spill_idx + P.spill_off[C.item];
    const int wI = (int)C.Ip->width, hI = (int)C.Ip->height, wJ = (int)C.Jp->width, hJ = (int)C.Jp->height;
    C.s1 = (KIND == 2) ? 1.0 : 1.0 / sqrt((double)(wI * hI));
    C.s2 = (KIND == 2) ? 1.0 : 1.0 / sqrt((double)(wJ * hJ));
    C.t1x = (KIND == 2) ? 0.0 : -0.5 * wI * C.s1; C.t1y = (KIND == 2) ? 0.0 : -0.5 * hI * C.s1;
    C.t2x = (KIND == 2) ? 0.0 : -0.5 * wJ * C.s2; C.t2y = (KIND == 2) ? 0.0 : -0.5 * hJ * C.s2;
    const double Dd = sqrt((double)wJ * (double)wJ + (double)hJ * (double)hJ);
    const double Aa = (double)wJ * (double)hJ;
    C.logalpha0 = (KIND == 0) ? log10(2.0 * Dd / Aa / C.s2)
                : (KIND == 1) ? log10(3.14159265358979323846 / Aa / (C.s2 * C.s2))
                              : log10(2.0 * Dd / Aa * 0.5);
    C.maxThreshold = P.precision_px * P.precision_px * C.s2 * C.s2;
    C.loge0 = log10(MAXM * (double)(C.m - SS));
    C.hist_base = (__double_as_longlong(fmin(C.maxThreshold, 1.0e6)) >> kHistShift) - (long long)(kHistBins - 1);
}

template <int KIND>
__device__ __forceinline__ double coop_residual(const double* M, double x1, double y1, double x2, double y2)
{
    return (KIND == 0) ? sym_epipolar_err(M, x1, y1, x2, y2) : (KIND == 1) ? h_asym_err(M, x1, y1, x2, y2) : epipolar_dist_err(M, x1, y1, x2, y2);
}

// ---- start-up of a pair: the prologue of acransac_body + the tables of the concave bound
template <int KIND>
__device__ void coop_init(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr double MULT_ERR = (KIND == 1) ? 1.0 : 0.5;
    const uint32_t m = C.m, lane = tid & 63u, wave = tid >> 6;
    if (KIND == 2 && tid < 18) {
        const uint2 sl = P.pairs[C.item];
        S.kinv[tid] = P.kinv[9 * (size_t)(tid < 9 ? sl.x : sl.y) + (tid < 9 ? tid : tid - 9)];
    }
    for (uint32_t p = tid; p < m; p += kCoopNT) {
        const r3dm_match q = C.mm[p];
        const double xi = (double)C.Ip->xy[2 * (size_t)q.i], yi = (double)C.Ip->xy[2 * (size_t)q.i + 1];
        const double xj = (double)C.Jp->xy[2 * (size_t)q.j], yj = (double)C.Jp->xy[2 * (size_t)q.j + 1];
        C.pt[4 * p + 0] = C.s1 * xi + C.t1x; C.pt[4 * p + 1] = C.s1 * yi + C.t1y;
        C.pt[4 * p + 2] = C.s2 * xj + C.t2x; C.pt[4 * p + 3] = C.s2 * yj + C.t2y;
        C.pool[p] = p;
    }
    if (tid == 0) {
        float pre = 0.0f;                                      // makelogcombi_n in the reference's float accumulation order (kernels_filter.hip)
        C.logc_n[0] = 0.0f; C.logc_n[m] = 0.0f;
        for (uint32_t i = 1; i <= m / 2; ++i) {
            pre = pre + (P.log10_tab[m - i + 1] - P.log10_tab[i]);
            C.logc_n[i] = pre;
            if (m - i > i) C.logc_n[m - i] = pre;
        }
        S.minNFA = __builtin_huge_val(); S.errorMax = __builtin_huge_val();
        for (int e = 0; e < 9; ++e) S.bestF[e] = 0.0;
        const uint32_t reserve = P.max_iter / 10;
        S.reserve = reserve; S.nIter = P.max_iter - reserve; S.iter = 0;
        S.pool_size = m; S.n_inl = 0; S.acMode = !(P.precision_px < __builtin_huge_val());
        S.n_models = 0; S.iters_done = 0; S.cnt = 0u; S.flag = 0u;
        S.chunk_iter0 = 0; S.chunk_n = 0; S.chunk_c = 0; S.chunk_valid = 0;
        S.b_c0 = S.b_c1 = S.b_n = 0; S.b_cap = 12;
        S.G = C.G; S.slice_len = C.slice_len;
    }
    // T*(k) = log10( m! / ((m - k)! SS! (k - SS)!) ), k >= SS: concave in k
    {
        const double inv_ln10 = 0.43429448190325182765;
        const double lgm = lgamma((double)m + 1.0), lgs = lgamma((double)SS + 1.0);
        for (uint32_t k = tid; k <= m; k += kCoopNT)
            C.tstar[k] = (k < SS) ? 0.0 : (lgm - lgamma((double)(m - k) + 1.0) - lgs - lgamma((double)(k - SS) + 1.0)) * inv_ln10;
    }
    for (uint32_t b = tid; b < (uint32_t)kHistBins; b += kCoopNT) {
        const double edge = b ? __longlong_as_double(((long long)b + C.hist_base) << kHistShift) : 0.0;
        C.la_tab[b] = C.logalpha0 + MULT_ERR * log10(edge + FLT_EPS_D);
    }
    wg_sync_global();                                          // logc_n (thread 0) and T* are read by everyone below
    double e = 0.0;
    for (uint32_t k = SS + 1 + tid; k <= m; k += kCoopNT) {
        const double d = fabs(((double)C.logc_n[k] + (double)P.logc_k[k]) - C.tstar[k]);
        e = d > e ? d : e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(e, off); e = o > e ? o : e; }
    if (lane == 0) S.red_v[wave] = e;
    r3dm_syncthreads();
    if (tid == 0) {
        double mx = 0.0;
        for (uint32_t w = 0; w < kCoopNW; ++w) mx = S.red_v[w] > mx ? S.red_v[w] : mx;
        S.eps_T = mx + 1.0e-9;                                 // (+ the rounding noise of T* itself)
    }
    r3dm_syncthreads();
}

// ---- draw + solve a chunk of minimal samples starting at iteration S.iter (the sampling block of acransac_body)
template <int KIND>
__device__ void coop_solve_chunk(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr int MS = (KIND == 2) ? 90 : 27;
    constexpr uint32_t CH = (uint32_t)CoopChunk<KIND>::n;
    const uint32_t iter0 = S.iter, nIter0 = S.nIter;
    const uint32_t chunk_n = (nIter0 - iter0 < CH) ? nIter0 - iter0 : CH;
    const uint32_t hyp = (KIND == 2) ? (tid >> 4) : tid;
    const uint32_t pool_size = S.pool_size;
    const double* K1i = S.kinv;
    const double* K2i = S.kinv + 9;
    if ((KIND == 2 || tid < CH) && hyp < chunk_n) {
        uint32_t pos[7];
        uint32_t cnt = 0, attempt = 0;
        while (cnt < SS) {
            const uint64_t r = rng_u64(P.seed, C.id.x, C.id.y, iter0 + hyp, attempt++);
            const uint32_t ps = (uint32_t)(((r >> 32) * (uint64_t)pool_size) >> 32);
            bool dup = false;
#pragma unroll
            for (int k = 0; k < 7; ++k) dup |= (k < (int)cnt) && (pos[k] == ps);
            if (!dup) {
#pragma unroll
                for (int k = 0; k < 7; ++k) if (k == (int)cnt) pos[k] = ps;
                ++cnt;
            }
        }
        double px1[7][2], px2[7][2];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint32_t sidx_ = C.pool[pos[k < (int)SS ? k : 0]];
            px1[k][0] = C.pt[4 * (size_t)sidx_ + 0]; px1[k][1] = C.pt[4 * (size_t)sidx_ + 1];
            px2[k][0] = C.pt[4 * (size_t)sidx_ + 2]; px2[k][1] = C.pt[4 * (size_t)sidx_ + 3];
            if (KIND == 2) {
                const double xa = px1[k][0], ya = px1[k][1], xb = px2[k][0], yb = px2[k][1];
                const double w1 = K1i[6] * xa + K1i[7] * ya + K1i[8];
                px1[k][0] = (K1i[0] * xa + K1i[1] * ya + K1i[2]) / w1;
                px1[k][1] = (K1i[3] * xa + K1i[4] * ya + K1i[5]) / w1;
                const double w2 = K2i[6] * xb + K2i[7] * yb + K2i[8];
                px2[k][0] = (K2i[0] * xb + K2i[1] * yb + K2i[2]) / w2;
                px2[k][1] = (K2i[3] * xb + K2i[4] * yb + K2i[5]) / w2;
            }
        }
        double* Fs = C.models + (size_t)hyp * MS;
        if constexpr (KIND == 2) {
            double* W = reinterpret_cast<double*>(smem + kCoopHdr) + (size_t)hyp * kE5Stride;
            const int l = (int)(tid & 15u);
            const int nm = five_point_coop(px1, px2, Fs, W, l);
            if (l == 0) S.nm[hyp] = (uint32_t)nm;
        } else {
            double F3[MS];
            int nm;
            if constexpr (KIND == 0) nm = seven_point(px1, px2, F3);
            else nm = four_point_h(px1, px2, F3);
            S.nm[tid] = (uint32_t)nm;
            for (int e = 0; e < 9 * nm; ++e) Fs[e] = F3[e];
        }
    }
    if (tid == 0) { S.chunk_iter0 = iter0; S.chunk_n = chunk_n; S.chunk_c = 0; S.chunk_valid = 1; }
    wg_sync_global();                                          // the models (global memory) are read by this workgroup's other waves
}

// ---- the next batch: whole iterations of the chunk from chunk_c on, at most b_cap models, inside the iteration budget
template <int KIND>
__device__ void coop_form_batch(const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr int MS = (KIND == 2) ? 90 : 27;
    if (tid == 0) {
        uint32_t c = S.chunk_c, n = 0;
        const uint32_t c0 = c;
        while (c < S.chunk_n && S.chunk_iter0 + c < S.nIter) {
            const uint32_t nm = S.nm[c];
            if (c > c0 && n + nm > S.b_cap) break;
            for (uint32_t k = 0; k < nm; ++k) { S.bj_c[n + k] = (uint8_t)c; S.bj_k[n + k] = (uint8_t)k; }
            n += nm; ++c;
        }
        S.b_c0 = c0; S.b_c1 = c; S.b_n = n;
    }
    r3dm_syncthreads();
    // the matrices the residuals are taken with: the model itself (F, H) or F = K2^-T E K1^-1 (every thread that needs it derives
    // the same bits from the same operations)
    if (tid < S.b_n) {
        const double* Mo = C.models + (size_t)S.bj_c[tid] * MS + 9 * (size_t)S.bj_k[tid];
        double M[9], FE[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) M[e] = Mo[e];
        if (KIND == 2) f_from_e(M, S.kinv, S.kinv + 9, FE);
        double* bm_l = reinterpret_cast<double*>(smem + kCoopBmOff) + 9 * tid;
#pragma unroll
        for (int e = 0; e < 9; ++e) { const double v = (KIND == 2) ? FE[e] : M[e]; bm_l[e] = v; C.bm[9 * tid + e] = v; }
    }
    r3dm_syncthreads();
}

// ---- one slice of the batch: residuals of the slice's matches for all models, histogram + count per model into the slice's slot
template <int KIND>
__device__ void coop_eval_slice(const FilterParams& P, const CoopCtx<KIND>& C, unsigned char* smem, uint32_t slice, uint32_t b_n,
                                bool bm_in_lds, uint32_t tid)
{
    const uint32_t lane = tid & 63u;
    double* bm_l = reinterpret_cast<double*>(smem + kCoopBmOff);
    uint32_t* cnt_l = reinterpret_cast<uint32_t*>(smem + kCoopCntOff);
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kCoopHdr);                    // [b_n][512]: bins 2w (low half), 2w + 1 (high half)
    if (!bm_in_lds) for (uint32_t e = tid; e < 9 * b_n; e += kCoopNT) bm_l[e] = C.bm[e];
    if (tid < (uint32_t)kCoopB) cnt_l[tid] = 0u;
    for (uint32_t e = tid; e < b_n * 512u; e += kCoopNT) hist[e] = 0u;
    r3dm_syncthreads();
    const uint32_t lo = slice * C.slice_len;
    uint32_t hi = lo + C.slice_len; if (hi > C.m) hi = C.m;
    const double maxThreshold = C.maxThreshold;
    for (uint32_t base = lo; base < hi; base += 2u * kCoopNT) {
        double px[2][4]; bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t p = base + (uint32_t)kCoopNT * (uint32_t)u + tid;
            valid[u] = p < hi;
            const size_t pp = 4 * (size_t)(valid[u] ? p : lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) px[u][e] = C.pt[pp + e];
        }
        for (uint32_t j = 0; j < b_n; ++j) {
            double M[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) M[e] = bm_l[9 * j + e];
            uint32_t n_new = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double r = coop_residual<KIND>(M, px[u][0], px[u][1], px[u][2], px[u][3]);
                const bool in = valid[u] && (r <= maxThreshold);
                if (in) {
                    long long bin = (__double_as_longlong(r) >> kHistShift) - C.hist_base;
                    bin = bin < 0 ? 0 : (bin > kHistBins - 1 ? kHistBins - 1 : bin);
                    atomicAdd(&hist[j * 512u + (uint32_t)(bin >> 1)], (bin & 1) ? 0x10000u : 1u);
                }
                n_new += (uint32_t)__builtin_popcountll(__ballot(in));
            }
            if (n_new != 0u && lane == 0) atomicAdd(&cnt_l[j], n_new);
        }
    }
    r3dm_syncthreads();
    uint32_t* gh = P.coop_hist + (size_t)(C.hoff + slice) * kCoopB * 512;
    uint32_t* gc = P.coop_cnt + (size_t)(C.hoff + slice) * kCoopB;
    if (tid < b_n) gc[tid] = cnt_l[tid];
    // (histograms of models without a match inside the bound are never read: their count says so)
    for (uint32_t j = 0; j < b_n; ++j) {
        if (cnt_l[j] == 0u) continue;
        gh[j * 512u + tid] = hist[j * 512u + tid];             // kCoopNT == 512 words per model
    }
    static_assert(kCoopNT == 512, "one histogram word per thread");
}

// ---- full evaluation of ONE model by one workgroup: residuals of all matches, compaction of those within the bound, sort, NFA scan
// (the evaluation block of acransac_body with its lists in global memory).  Returns the model's NFA and inlier count.
template <int KIND>
__device__ void coop_full_eval(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, const double* M /* residual matrix */,
                               uint32_t tid, double& nfa_out, uint32_t& kbest_out, uint32_t& total_out)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr double MULT_ERR = (KIND == 1) ? 1.0 : 0.5;
    constexpr int NT = kCoopNT;
    const uint32_t lane = tid & 63u, wave = tid >> 6, m = C.m;
    unsigned long long* keys = C.keys; uint32_t* sidx = C.sidx;
    const double* pt = C.pt;
    const double maxThreshold = C.maxThreshold;
    if (tid == 0) S.cnt = 0u;
    r3dm_syncthreads();
    for (uint32_t base = 0; base < m; base += 4u * NT) {
        double r[4]; bool in[4];
        double px[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = base + (uint32_t)NT * (uint32_t)u + tid;
            const size_t pp = 4 * (size_t)(p < m ? p : 0u);
#pragma unroll
            for (int e = 0; e < 4; ++e) px[u][e] = pt[pp + e];
        }
        unsigned long long bal[4];
        uint32_t n_new = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = base + (uint32_t)NT * (uint32_t)u + tid;
            r[u] = coop_residual<KIND>(M, px[u][0], px[u][1], px[u][2], px[u][3]);
            in[u] = (p < m) && (r[u] <= maxThreshold);
            bal[u] = __ballot(in[u]);
            n_new += (uint32_t)__builtin_popcountll(bal[u]);
        }
        if (n_new != 0u) {
            uint32_t woff = 0;
            if (lane == 0) woff = atomicAdd(&S.cnt, n_new);
            woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)woff);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t pos = woff + (uint32_t)__builtin_popcountll(bal[u] & ((1ull << lane) - 1ull));
                if (in[u]) { keys[pos] = (unsigned long long)__double_as_longlong(r[u]); sidx[pos] = base + (uint32_t)NT * (uint32_t)u + tid; }
                woff += (uint32_t)__builtin_popcountll(bal[u]);
            }
        }
    }
    wg_sync_t<true>();
    const uint32_t total = S.cnt;
    total_out = total;
    double nfa = __builtin_huge_val();
    uint32_t kbest = SS;
    if (total > SS) {
        uint32_t cap = 1; while (cap < total) cap <<= 1;
        bool sorted = true;
        switch (cap / (uint32_t)NT) {
            case 0: case 1: wg_sort_regs<1, true>(keys, sidx, cap, total, tid); break;
            case 2: wg_sort_regs<2, true>(keys, sidx, cap, total, tid); break;
            case 4: wg_sort_regs<4, true>(keys, sidx, cap, total, tid); break;
            case 8: wg_sort_regs<8, true>(keys, sidx, cap, total, tid); break;
            case 16: wg_sort_regs<16, true>(keys, sidx, cap, total, tid); break;
            case 32: wg_sort_regs<32, true>(keys, sidx, cap, total, tid); break;
            default: sorted = false; break;
        }
        if (!sorted) {                                          // longer lists: the plain network in global memory
            for (uint32_t q = total + tid; q < cap; q += NT) { keys[q] = ~0ull; sidx[q] = 0xFFFFFFFFu; }
            wg_sync_t<true>();
            for (uint32_t size = 2; size <= cap; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    for (uint32_t tI = tid; tI < (cap >> 1); tI += NT) {
                        const uint32_t lo = 2 * tI - (tI & (stride - 1));
                        const uint32_t hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const unsigned long long x = keys[lo], y = keys[hi];
                        const uint32_t xi = sidx[lo], yi = sidx[hi];
                        const bool gt = (x > y) || (x == y && xi > yi);
                        if (gt == up) { keys[lo] = y; keys[hi] = x; sidx[lo] = yi; sidx[hi] = xi; }
                    }
                    wg_sync_t<true>();
                }
            }
        }
        // bestNFA: k = SS + 1 .. total, first minimum wins
        double bv = __builtin_huge_val(); uint32_t bk = 0xFFFFFFFFu;
        for (uint32_t kk = SS + 1 + tid; kk <= total; kk += NT) {
            const double e = __longlong_as_double((long long)keys[kk - 1]);
            const double logalpha = C.logalpha0 + MULT_ERR * log10(e + FLT_EPS_D);
            const double v = C.loge0 + logalpha * (double)(kk - SS) + (double)C.logc_n[kk] + (double)P.logc_k[kk];
            if (v < bv) { bv = v; bk = kk; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off);
            const uint32_t ok = __shfl_xor(bk, off);
            if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        if (lane == 0) { S.red_v[wave] = bv; S.red_k[wave] = bk; }
        wg_sync_t<true>();
#pragma unroll
        for (uint32_t w = 0; w < kCoopNW; ++w) {
            const double ov = S.red_v[w]; const uint32_t ok = S.red_k[w];
            if (w == 0 || ov < nfa || (ov == nfa && ok < kbest)) { nfa = ov; kbest = ok; }
        }
        if (kbest == 0xFFFFFFFFu) { nfa = __builtin_huge_val(); kbest = SS; }
        r3dm_syncthreads();                                     // red_v / red_k are reused by the next evaluation
    }
    nfa_out = nfa; kbest_out = kbest;
}

// ---- merged histograms -> count and NFA bound of every model of the batch (one wave per model, eight at a time)
template <int KIND>
__device__ void coop_batch_bounds(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t* gh = P.coop_hist + (size_t)C.hoff * kCoopB * 512;
    const uint32_t* gc = P.coop_cnt + (size_t)C.hoff * kCoopB;
    for (uint32_t j = wave; j < S.b_n; j += kCoopNW) {
        uint32_t total = 0;
        for (uint32_t s = 0; s < C.G; ++s) total += gc[(size_t)s * kCoopB + j];
        double wmin = __builtin_huge_val();
        if (total > SS) {
            uint32_t h[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = 0u;
            for (uint32_t s = 0; s < C.G; ++s) {
                if (gc[(size_t)s * kCoopB + j] == 0u) continue;                       // (slot not written)
                const uint4* src = reinterpret_cast<const uint4*>(gh + ((size_t)s * kCoopB + j) * 512u + 8u * lane);
                const uint4 a = src[0], b = src[1];
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) { h[2 * i] += w[i] & 0xFFFFu; h[2 * i + 1] += w[i] >> 16; }
            }
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) run += h[i];
            uint32_t incl = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
            uint32_t k_prev = incl - run;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t k_hi = k_prev + h[i];
                uint32_t k_lo = k_prev + 1u; if (k_lo < SS + 1u) k_lo = SS + 1u;
                if (h[i] != 0u && k_hi >= k_lo) {
                    const double la = C.la_tab[16u * lane + (uint32_t)i];
                    const double v_lo = la * (double)(k_lo - SS) + C.tstar[k_lo];
                    const double v_hi = la * (double)(k_hi - SS) + C.tstar[k_hi];
                    const double v = v_lo < v_hi ? v_lo : v_hi;
                    wmin = v < wmin ? v : wmin;
                }
                k_prev = k_hi;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(wmin, off); wmin = o < wmin ? o : wmin; }
        }
        if (lane == 0) { S.tot[j] = total; S.bnd[j] = wmin; }
    }
    r3dm_syncthreads();
}

// ---- walk the batch in the reference's order (the evaluation loop of acransac_body, a model's evaluation replaced by its
// merged count / bound and, for the few models that can win, coop_full_eval)
template <int KIND>
__device__ void coop_decide(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr int MS = (KIND == 2) ? 90 : 27;
    constexpr int NT = kCoopNT;
    const uint32_t lane = tid & 63u, wave = tid >> 6, m = C.m;
    [[maybe_unused]] const uint32_t item = C.item;                 // (FCHECK reports it)
    if (S.b_n) coop_batch_bounds<KIND>(P, C, S, tid);
    bool pool_changed = false;
    uint32_t j = 0;
    uint32_t c = S.b_c0;
    const uint32_t c1 = S.b_c1;
    for (; c < c1 && !pool_changed; ++c) {
        const uint32_t it = S.chunk_iter0 + c;
        const uint32_t nm = S.nm[c];
        bool better = false;
        for (uint32_t k = 0; k < nm; ++k, ++j) {
            const uint32_t total = S.tot[j];
            bool ac = S.acMode != 0;
            if (!ac && (double)total > 2.5 * SS) ac = true;
            double nfa = __builtin_huge_val();
            uint32_t kbest = SS;
            const double* Mo = C.models + (size_t)c * MS + 9 * (size_t)k;       // the model itself (E for KIND 2)
            if (ac && total > SS) {
                const double bound = C.loge0 + S.bnd[j] - S.eps_T;
                const bool hopeless = !R3DM_DBG(P) && (S.minNFA < __builtin_huge_val()) && (bound - 1.0e-6 >= S.minNFA);
                if (!hopeless) {
                    double Mr[9];
                    const double* bm_g = C.bm + 9 * (size_t)j;
#pragma unroll
                    for (int e = 0; e < 9; ++e) Mr[e] = bm_g[e];
                    uint32_t total2 = 0;
                    coop_full_eval<KIND>(P, C, S, Mr, tid, nfa, kbest, total2);
                    FCHECK(total2 == total, 9, total2, total);                     // the slices and the full pass count the same matches
                    FCHECK(bound - 1.0e-6 <= nfa, 8, kbest, total);               // the sort-skipping bound really is one
                }
            }
            const bool improve = ac && (nfa < S.minNFA);
            if (improve) {
                for (uint32_t q = tid; q < kbest; q += NT) C.inl[q] = C.sidx[q];
                better = true;
            }
            wg_sync_t<true>();
            if (tid == 0) {
                S.acMode = ac ? 1u : 0u;
                S.n_models += 1;
                if (improve) {
                    S.minNFA = nfa;
                    S.n_inl = kbest;
                    S.errorMax = __longlong_as_double((long long)C.keys[kbest - 1]);
#pragma unroll
                    for (int e = 0; e < 9; ++e) S.bestF[e] = Mo[e];
                }
            }
            r3dm_syncthreads();
        }
        // ---- end of iteration `it`: ACRANSAC's pool / budget update.  Thread 0 is about to change the loop bounds the other waves
        // read right after the previous iteration's last barrier; a sample without real solutions (nm == 0) needs its own.
        if (nm == 0) r3dm_syncthreads();
        if (tid == 0) {
            S.iters_done = it + 1;
            S.flag = 0;
            const bool trigger = (better && S.minNFA < 0.0) || (it + 1 == S.nIter && S.reserve != 0);
            if (trigger) {
                if (S.n_inl == 0) { S.nIter += 1; S.reserve -= 1; }
                else {
                    S.flag = 1;
                    S.pool_size = S.n_inl;
                    if (S.reserve) { S.nIter = it + 1 + S.reserve; S.reserve = 0; }
                }
            }
        }
        r3dm_syncthreads();
        if (S.flag) {
            // new sampling pool = the inlier SET in ascending index order (same rule as acransac_body / oracle/acransac.c)
            const uint32_t ni = S.n_inl;
            uint32_t* flags = C.sidx;                                // sort scratch, free between models
            for (uint32_t q = tid; q < m; q += NT) flags[q] = 0u;
            wg_sync_t<true>();
            for (uint32_t q = tid; q < ni; q += NT) flags[C.inl[q]] = 1u;
            wg_sync_t<true>();
            uint32_t filled = 0;
            for (uint32_t base = 0; base < m; base += NT) {
                const uint32_t p = base + tid;
                const bool in = (p < m) && (flags[p] != 0u);
                const unsigned long long bal = __ballot(in);
                const uint32_t before = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) S.wave_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
                r3dm_syncthreads();
                uint32_t woff = 0, tot = 0;
#pragma unroll
                for (uint32_t w = 0; w < kCoopNW; ++w) { const uint32_t cw = S.wave_cnt[w]; if (w < wave) woff += cw; tot += cw; }
                if (in) C.pool[filled + woff + before] = p;
                filled += tot;
                r3dm_syncthreads();
            }
            pool_changed = true;
            wg_fence();
        }
        r3dm_syncthreads();
        if (it + 1 >= S.nIter) { ++c; break; }
    }
    if (tid == 0) {
        S.iter = S.chunk_iter0 + c;
        S.chunk_c = c;
        if (pool_changed) { S.chunk_valid = 0; S.b_cap = 12; }
        else { const uint32_t nb = S.b_cap * 2; S.b_cap = nb > (uint32_t)kCoopB ? (uint32_t)kCoopB : nb; }
    }
    wg_sync_global();
}

// ---- result of a pair (the epilogue of acransac_body)
template <int KIND>
__device__ void coop_finish(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, uint32_t tid)
{
    if (tid != 0) return;
    const uint32_t item = C.item;
    uint32_t n_inl = S.n_inl;
    if (!(S.minNFA < 0.0)) n_inl = 0;
    P.inl_count[item] = n_inl;
    double Fo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double thr = 0.0;
    if (n_inl > 0) {
        const double s1 = C.s1, s2 = C.s2;
        const double N1[9] = {s1, 0, C.t1x, 0, s1, C.t1y, 0, 0, 1};
        const double N2[9] = {s2, 0, C.t2x, 0, s2, C.t2y, 0, 0, 1};
        const double N2i[9] = {1.0 / s2, 0, -C.t2x / s2, 0, 1.0 / s2, -C.t2y / s2, 0, 0, 1};
        double T[9];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double v = 0.0;
                for (int k = 0; k < 3; ++k) v += ((KIND == 0) ? N2[3 * k + r] : N2i[3 * r + k]) * S.bestF[3 * k + cc];
                T[3 * r + cc] = v;
            }
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double v = 0.0;
                for (int k = 0; k < 3; ++k) v += T[3 * r + k] * N1[3 * k + cc];
                Fo[3 * r + cc] = v;
            }
        thr = sqrt(S.errorMax) / s2;
        if (KIND == 2) { for (int e = 0; e < 9; ++e) Fo[e] = S.bestF[e]; thr = S.errorMax; }
    }
    for (int e = 0; e < 9; ++e) P.F_out[9 * (size_t)item + e] = Fo[e];
    P.thr_nfa[2 * (size_t)item] = thr;
    P.thr_nfa[2 * (size_t)item + 1] = S.minNFA;
    P.iters[2 * (size_t)item] = S.iters_done;
    P.iters[2 * (size_t)item + 1] = S.n_models;
}

// LDS <-> global copies of the pair state (everything in front of CoopS::copy_end)
__device__ __forceinline__ void coop_state_copy(uint32_t* dst, const uint32_t* src, uint32_t tid)
{
    constexpr uint32_t W = (uint32_t)(offsetof(CoopS, copy_end) / 4);
    for (uint32_t e = tid; e < W; e += kCoopNT) dst[e] = src[e];
}

template <int KIND>
__global__ __launch_bounds__(kCoopNT, 1)
void acransac_coop_kernel(const FilterParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t sh_task, sh_last;
    CoopS& S = *reinterpret_cast<CoopS*>(smem);
    const uint32_t tid = threadIdx.x;
    uint32_t* q = P.coop_q;
    for (;;) {
        // ---- next task
        if (tid == 0) {
            uint32_t v = kCoopNoTask;
            for (;;) {
                if (coop_pop(q, v)) break;
                v = kCoopNoTask;
                if (QLOAD(&q[2]) == q[5]) break;                                   // every pair is finished
                const uint32_t a = QLOAD(&q[4]), pot = QLOAD(&q[3]);
                if (a > pot && atomicCAS(&q[4], a, a - 1u) == a) break;            // more workers than tasks can exist: retire
                __builtin_amdgcn_s_sleep(16);
            }
            sh_task = v;
        }
        r3dm_syncthreads();
        const uint32_t task = sh_task;
        r3dm_syncthreads();
        if (task == kCoopNoTask) return;
        const uint32_t cp = task >> 5, slice = task & 31u;
        __threadfence();                                                           // acquire: what the task's publisher wrote
        CoopCtx<KIND> C;
        coop_ctx<KIND>(P, cp, C);
        bool decide_first;
        if (slice == kCoopInitSlice) {
            coop_init<KIND>(P, C, S, tid);
            decide_first = false;
        } else {
            coop_eval_slice<KIND>(P, C, smem, slice, C.gs->b_n, false, tid);
            __threadfence();                                                       // release: the slot
            r3dm_syncthreads();
            if (tid == 0) sh_last = (atomicAdd(&C.gs->arrived, 1u) == C.G - 1u) ? 1u : 0u;
            r3dm_syncthreads();
            const bool last = sh_last != 0u;
            r3dm_syncthreads();
            if (!last) continue;
            __threadfence();                                                       // acquire: the other slices' slots, the pair state
            coop_state_copy(reinterpret_cast<uint32_t*>(&S), reinterpret_cast<const uint32_t*>(C.gs), tid);
            r3dm_syncthreads();
            decide_first = true;
        }
        // ---- this workgroup leads the pair until it hands a batch to the queue and is not the last to arrive
        for (;;) {
            if (decide_first) coop_decide<KIND>(P, C, S, tid);
            decide_first = true;
            if (S.iter >= S.nIter) {
                coop_finish<KIND>(P, C, S, tid);
                r3dm_syncthreads();
                if (tid == 0) { __threadfence(); atomicSub(&q[3], C.G); atomicAdd(&q[2], 1u); }
                break;
            }
            if (!S.chunk_valid || S.chunk_c >= S.chunk_n) coop_solve_chunk<KIND>(P, C, S, smem, tid);
            coop_form_batch<KIND>(C, S, smem, tid);
            if (S.b_n == 0) continue;                                              // iterations without a model: only their bookkeeping
            const uint32_t b_n = S.b_n;
            if (C.G > 1u) {
                coop_state_copy(reinterpret_cast<uint32_t*>(C.gs), reinterpret_cast<const uint32_t*>(&S), tid);
                if (tid == 0) C.gs->arrived = 0u;
                __threadfence();                                                   // release: state, batch models, (first batch) the pair's tables
                r3dm_syncthreads();
                if (tid == 0) for (uint32_t s = 1; s < C.G; ++s) coop_push(q, (cp << 5) | s);
            }
            coop_eval_slice<KIND>(P, C, smem, 0u, b_n, true, tid);
            bool last = true;
            if (C.G > 1u) {
                __threadfence();
                r3dm_syncthreads();
                if (tid == 0) sh_last = (atomicAdd(&C.gs->arrived, 1u) == C.G - 1u) ? 1u : 0u;
                r3dm_syncthreads();
                last = sh_last != 0u;
                r3dm_syncthreads();
                if (last) __threadfence();
            } else {
                wg_sync_global();                                                  // the slot is read back by this workgroup's other waves
            }
            if (!last) break;                                                      // another workgroup will lead (S in LDS is dropped)
        }
    }
}

template <int KIND>
static hipError_t launch_coop(hipStream_t st, const FilterParams& P, uint32_t n_workers)
{
    const size_t lds = coop_lds_bytes_(KIND);
    hipError_t e = hipFuncSetAttribute((const void*)acransac_coop_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((acransac_coop_kernel<KIND>), dim3(n_workers), dim3(kCoopNT), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_filter_coop_E(hipStream_t st, const FilterParams& P, uint32_t n_workers);
#ifdef R3DM_FILTER_COOP_ONLY_E
hipError_t launch_filter_coop_E(hipStream_t st, const FilterParams& P, uint32_t n_workers) { return launch_coop<2>(st, P, n_workers); }
#else
size_t filter_coop_lds_bytes(int model_kind) { return coop_lds_bytes_(model_kind); }
hipError_t launch_filter_coop(hipStream_t st, const FilterParams& P, uint32_t n_workers)
{
    if (P.n_coop == 0 || n_workers == 0) return hipSuccess;
    if (P.model_kind == 2) return launch_filter_coop_E(st, P, n_workers);
    return P.model_kind == 0 ? launch_coop<0>(st, P, n_workers) : launch_coop<1>(st, P, n_workers);
}
#endif

}  // namespace r3dm
