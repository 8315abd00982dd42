// kernels_filter_coop.hip -- AC-RANSAC of a LONG pair spread over several workgroups (round 4).
//
// The one-workgroup-per-pair kernel (kernels_filter.hip) binds a pair to one CU: a pair of 10-20 k putative matches is bound by
// that CU's f64 rate, and a collection of few, long pairs (what Regard3D produces: tens of photographs) leaves most of the chip
// idle.  Here a pair with more than a threshold of matches (api_filter.cpp: 4096) is evaluated by G workgroups:
//
//   * ACRANSAC stays sequential where it is sequential (the reference's iteration order, the pool that shrinks on every
//     meaningful improvement, the budget that is cut once -- SURVEY.md A.5); what is spread is the part that is not: the
//     residuals of a BATCH of up to 32 models (whole iterations of the current chunk of minimal samples) over the pair's matches.
//     A workgroup takes the s-th slice of the match list, loads every point once for all models of the batch, and leaves a
//     residual histogram (the 1024 bins of the sort-skipping bound) and a count per model in its global slot.
//   * The pair's LEADER (the workgroup that started the pair; its state stays in that workgroup's LDS for the pair's lifetime)
//     merges the slots and walks the batch in the reference's order.  The NFA bound of a model comes from the merged histogram
//     alone; only a model whose bound can beat the best NFA so far -- a few per cent -- is evaluated in full by the leader
//     (residuals of all matches, sort, NFA scan), so every decision is the one the sequential algorithm takes: same inlier sets,
//     same models, same iteration and model counts.  Models behind a pool change are discarded like the rest of a chunk always
//     was; the batch size starts small after a pool change and doubles (12, 24, 32) while nothing changes.
//   * Workgroups are not bound to slices, and ONE kernel serves the F, E and H filters of a call.  The kernel is a pool of workers: a
//     worker starts a pair that has no leader yet, or waits at its own mailbox for a slice.  A leader hands the slices of a batch to
//     the workers that are idle at that moment (one fetch-and on the idle bitmap claims them, one store each delivers the task), runs
//     slice 0 and every slice nobody was idle for itself, and then waits only for slices that ARE running somewhere.  No workgroup
//     ever waits on something a not-yet-scheduled workgroup would have to provide, so nothing depends on co-residency.  A worker that
//     waits while fewer slices can exist than workers are alive retires.
//   * The bound: NFA_k >= loge0 + la(bin of the k-th residual) (k - SS) + T[k], T[k] = logc_n[k] + logc_k[k] (float tables).
//     T*(k) = log10(m! / ((m-k)! SS! (k-SS)!)) is concave in k, so over the k range of a bin la_b (k - SS) + T*(k) takes its minimum
//     at an end of the range: two evaluations per non-empty bin instead of one per k (the walk over k was half the evaluation time
//     of a long pair).  eps_T = max_k |T[k] - T*(k)|, computed per pair at start-up, makes the bound rigorous for the float
//     tables: bound = min over bins and ends - eps_T - 1e-6.
//
// Hand-offs between workgroups (eight XCDs, private L2s, per-CU L1s that no other CU's store refreshes) follow the write-through
// form of /opt/skills/guides/cdna_hip_programming.md section 6 Guideline 16: every word another workgroup reads is STORED sc1
// (16-byte raw-buffer stores with aux = sc1, or 8-byte relaxed agent-scope atomics) and LOADED sc1; the storing waves drain
// (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, one lane bumps a relaxed agent-scope counter; pollers read that one
// word relaxed with s_sleep between reads.  No agent-scope fence anywhere: the first version fenced (__threadfence, every thread,
// four times per batch) and polled its queue with acquire loads -- correct, and 15x slower than the work on a collection of 66
// pairs (every fence a write-back + invalidate that the other 255 workgroups pay for).
#define R3DM_FILTER_DEVICE_ONLY 1
#include "kernels_filter.hip"

namespace r3dm {

constexpr int kCoopNT = 512;                // threads per worker
constexpr uint32_t kCoopNW = kCoopNT / 64;
constexpr uint32_t kCoopNoTask = 0xFFFFFFFFu, kCoopDone = 0xFFFFFFFEu, kCoopStall = 0xFFFFFFFDu;
template <int KIND> struct CoopChunk { static constexpr int n = (KIND == 2) ? 2 * kE5Samples : kChunk; };   // E: 32 groups of 16 lanes

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define QLOAD(p) __hip_atomic_load((p), RLX_AGENT)
#define DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// Every wait in this kernel is bounded: a poller that has waited kCoopStallTicks of the 100 MHz wall clock (2 s; a whole call takes
// milliseconds) records a stall code in the queue header, and every loop leaves as soon as that word is set -- the host then reports
// R3DM_ERR_HIP with the code instead of a device that never comes back.
constexpr unsigned long long kCoopStallTicks = 200000000ull;
constexpr int kAuxSc1 = 16;                 // raw-buffer aux bits: sc1 (write-through store / L1-bypassing load)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t coop_rsrc(const void* p)
{
    // descriptors from wave-uniform values only (readfirstlane), so no waterfall loop is emitted
    const uint64_t a = (uint64_t)p;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a)),
        0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t voff)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, kAuxSc1));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t voff, u32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, kAuxSc1);
}

// What a slice task reads about its pair.  One 64-byte record per pair in global memory, every word an 8-byte agent-scope atomic.
struct CoopPub {
    gu64 pt;             // the pair's normalised points (byte offset into FilterParams::pts_scratch)
    gu64 max_thr_bits;   // residual bound (bits of the double)
    gu64 hist_base;      // first histogram bin (bits of the signed value)
    gu64 m_slice;        // m | slice_len << 32
    gu64 hoff_bn;        // first slot of the pair | models of the batch in flight << 32
    gu64 arrived;        // slices of the batch that have been delivered (the leader's own is not counted)
    gu64 pad[2];
};
static_assert(sizeof(CoopPub) == 64, "one record per 64 bytes");

// State of one pair: in the LDS of its leader from start-up to the result.
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
    uint32_t cnt, flag;
    uint32_t wave_cnt[8], red_k[8];
    uint32_t nm[64];                       // models of every hypothesis of the chunk
    uint8_t  bj_c[kCoopB], bj_k[kCoopB];   // model j of the batch = model bj_k[j] of hypothesis bj_c[j]
    uint32_t sh_task, sh_aux;              // broadcast slots of thread 0 (all LDS in the one dynamic array: Guideline 17)
    uint32_t n_claim, kept;                // idle workers thread 0 claimed for the batch's slices; bitmask of the slices nobody took
    uint8_t  claim[32];
};
static_assert(sizeof(CoopS) <= 2048, "CoopS must fit the LDS header");
// LDS: [CoopS | slice scratch: batch matrices 32 x 9 doubles at 2048, counts at 4352 | slot counts of the merge at 4608 (30 x 32 u32)] [region R at 8704]
constexpr int kCoopBmOff = 2048, kCoopCntOff = 2048 + kCoopB * 72, kCoopAllCntOff = 4608, kCoopHdr = 4608 + 4096;
static_assert(kCoopCntOff + kCoopB * 4 <= kCoopAllCntOff && kCoopMaxG * kCoopB * 4 <= 4096, "LDS header");

static inline size_t coop_lds_bytes_(int model_kind)
{
    // region R serves, one at a time: the slice histograms (32 x 1024 u16 bins), the 5-point workspaces, the sort's image of a block
    const size_t hist = (size_t)kCoopB * 512 * 4;
    const size_t work = model_kind == 2 ? (size_t)CoopChunk<2>::n * kE5Stride * 8 : 0;
    const size_t sort = (size_t)8192 * 12;
    return (size_t)kCoopHdr + std::max(std::max(hist, work), sort);
}

// ---- scheduling state in global memory (FilterParams::coop_q), every access a relaxed agent-scope atomic by ONE lane.
// Three 128-byte lines of words, then the workers' mailboxes:
//   line 0: [5] pairs [7] next pair to start [8] stall code [9] stall info [10..15] what the staller saw [20] workers [21] pairs led at a time
//   line 1: [32 .. 39] idle bitmap, four 64-bit words: bit w = worker w waits at its mailbox
//   line 2: [64] pairs finished [65] potential (slices of unfinished pairs) [66] workers alive
//   [96 ..) mailboxes, 32 words (one line) per worker (at most 256 workers)
// A worker with nothing to do sets its bit in the idle bitmap and waits at ITS OWN mailbox line; a leader claims as many idle workers as
// its batch has slices with ONE fetch-and on a bitmap word and writes the tasks into their mailboxes; the slices nobody is idle for it
// runs itself (looking once more for an idle worker before each).  Nobody polls a shared word, and no task ever waits in a queue.
// What this replaced, on a collection of 66 pairs x 3 filters: a shared task ring that every idle worker polled (~200 pollers on one
// head word: each task delayed by ~125 us, 2.3x slower with 256 workers than with 100); an idle RING popped entry by entry (six
// lanes of a leader fighting over one head word: 55 us per batch); an overflow ring for the slices nobody was idle for, polled by
// the waiting leaders (tasks waited 0.7 ms on average: 62 ms for the call against 14 now).
constexpr uint32_t kQNextPair = 7, kQStall = 8, kQIdle = 32, kQDone = 64, kQPot = 65, kQActive = 66, kQArrays = 96;
constexpr uint32_t kMboxEmpty = 0u, kMboxRetired = 0xFFFFFFFEu;

struct CoopSched {
    gu64* idle;          // [4]
    uint32_t* mbox;
};
__device__ __forceinline__ CoopSched coop_sched(uint32_t* q)
{
    CoopSched Q;
    Q.idle = reinterpret_cast<gu64*>(q + kQIdle);
    Q.mbox = q + kQArrays;
    return Q;
}
__device__ __forceinline__ bool coop_stalled(uint32_t* q) { return QLOAD(&q[kQStall]) != 0u; }
__device__ __forceinline__ void coop_report_stall(uint32_t* q, uint32_t code, uint32_t info)
{
    uint32_t expect = 0u;
    if (__hip_atomic_compare_exchange_strong(&q[kQStall], &expect, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&q[9], info, RLX_AGENT);
        __hip_atomic_store(&q[10], (uint32_t)QLOAD(reinterpret_cast<gu64*>(q + kQIdle)), RLX_AGENT);                      // idle workers 0 .. 31
        __hip_atomic_store(&q[11], (uint32_t)(QLOAD(reinterpret_cast<gu64*>(q + kQIdle)) >> 32), RLX_AGENT);              // 32 .. 63
        __hip_atomic_store(&q[12], QLOAD(&q[kQDone]), RLX_AGENT); __hip_atomic_store(&q[13], QLOAD(&q[kQPot]), RLX_AGENT);
        __hip_atomic_store(&q[14], QLOAD(&q[kQActive]), RLX_AGENT); __hip_atomic_store(&q[15], QLOAD(&q[kQNextPair]), RLX_AGENT);
    }
}
__device__ __forceinline__ bool mbox_cas(uint32_t* word, uint32_t from, uint32_t to)
{
    uint32_t expect = from;
    return __hip_atomic_compare_exchange_strong(word, &expect, to, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what the leader knows about its pair
template <int KIND>
struct CoopCtx {
    uint32_t item, cp, m, G, slice_len, hoff;
    uint2 id;
    double s1, s2, t1x, t1y, t2x, t2y, logalpha0, maxThreshold, loge0;
    long long hist_base;
    double* pt; uint32_t* pool; uint32_t* inl; float* logc_n; double* tstar; double* la_tab;
    double* models; double* bm; CoopPub* pub;
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
    C.pub = reinterpret_cast<CoopPub*>(P.coop_pub) + cp;
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

// developer build: where a pair's time goes (R3DM_COOP_PROF=1 prints the table; FilterParams::coop_prof is null otherwise).  Ticks of the
// 100 MHz wall clock: [0] start-up [1] solves [2] batch forming + publishing [3] slice evaluations (all workgroups) [4] the leader
// waiting for / helping with slices [5] merged bounds [6] full evaluations [7] their number [8] rest of the walk [9] batches
// [10] start .. finish [11] models in batches
#ifdef R3DM_DEVTOOLS
#define PROF_NOW() ((P.coop_prof && threadIdx.x == 0) ? (unsigned long long)wall_clock64() : 0ull)
#define PROF_PUT(cp_, slot, v) do { if (P.coop_prof && threadIdx.x == 0) atomicAdd(&P.coop_prof[16 * (size_t)(cp_) + (slot)], (unsigned long long)(v)); } while (0)
#else
#define PROF_NOW() 0ull
#define PROF_PUT(cp_, slot, v) do { (void)(cp_); (void)(v); } while (0)
#endif

template <int KIND>
__device__ __forceinline__ double coop_residual(const double* M, double x1, double y1, double x2, double y2)
{
    return (KIND == 0) ? sym_epipolar_err(M, x1, y1, x2, y2) : (KIND == 1) ? h_asym_err(M, x1, y1, x2, y2) : epipolar_dist_err(M, x1, y1, x2, y2);
}

// ---- start-up of a pair: the prologue of acransac_body + the tables of the concave bound.  The points go out write-through (slice
// tasks on other XCDs read them); logc_n comes from the host (api_filter.cpp: its float running sum is order-bound, a serial loop
// of m / 2 steps was 1.0-1.6 ms of a 12 k pair's start-up here).
template <int KIND>
__device__ __attribute__((noinline)) void coop_init(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr double MULT_ERR = (KIND == 1) ? 1.0 : 0.5;
    const uint32_t m = C.m, lane = tid & 63u, wave = tid >> 6;
    if (KIND == 2 && tid < 18) {
        const uint2 sl = P.pairs[C.item];
        S.kinv[tid] = P.kinv[9 * (size_t)(tid < 9 ? sl.x : sl.y) + (tid < 9 ? tid : tid - 9)];
    }
    {
        const __amdgpu_buffer_rsrc_t rp = coop_rsrc(P.pts_scratch);
        const uint32_t pt_off = (uint32_t)((const char*)C.pt - (const char*)P.pts_scratch);
        for (uint32_t p = tid; p < m; p += kCoopNT) {
            const r3dm_match q = C.mm[p];
            const double xi = (double)C.Ip->xy[2 * (size_t)q.i], yi = (double)C.Ip->xy[2 * (size_t)q.i + 1];
            const double xj = (double)C.Jp->xy[2 * (size_t)q.j], yj = (double)C.Jp->xy[2 * (size_t)q.j + 1];
            const double a0 = C.s1 * xi + C.t1x, a1 = C.s1 * yi + C.t1y, b0 = C.s2 * xj + C.t2x, b1 = C.s2 * yj + C.t2y;
            const unsigned long long w0 = (unsigned long long)__double_as_longlong(a0), w1 = (unsigned long long)__double_as_longlong(a1);
            const unsigned long long w2 = (unsigned long long)__double_as_longlong(b0), w3 = (unsigned long long)__double_as_longlong(b1);
            u32x4 lo4, hi4;
            lo4.x = (uint32_t)w0; lo4.y = (uint32_t)(w0 >> 32); lo4.z = (uint32_t)w1; lo4.w = (uint32_t)(w1 >> 32);
            hi4.x = (uint32_t)w2; hi4.y = (uint32_t)(w2 >> 32); hi4.z = (uint32_t)w3; hi4.w = (uint32_t)(w3 >> 32);
            st16_sc1(rp, pt_off + 32u * p, lo4);
            st16_sc1(rp, pt_off + 32u * p + 16u, hi4);
            C.pool[p] = p;
        }
    }
    if (tid == 0) {
        S.minNFA = __builtin_huge_val(); S.errorMax = __builtin_huge_val();
        for (int e = 0; e < 9; ++e) S.bestF[e] = 0.0;
        const uint32_t reserve = P.max_iter / 10;
        S.reserve = reserve; S.nIter = P.max_iter - reserve; S.iter = 0;
        S.pool_size = m; S.n_inl = 0; S.acMode = !(P.precision_px < __builtin_huge_val());
        S.n_models = 0; S.iters_done = 0; S.cnt = 0u; S.flag = 0u;
        S.chunk_iter0 = 0; S.chunk_n = 0; S.chunk_c = 0; S.chunk_valid = 0;
        S.b_c0 = S.b_c1 = S.b_n = 0; S.b_cap = 12;
        CoopPub* pub = C.pub;
        __hip_atomic_store(&pub->pt, (gu64)((const char*)C.pt - (const char*)P.pts_scratch), RLX_AGENT);
        __hip_atomic_store(&pub->max_thr_bits, (gu64)__double_as_longlong(C.maxThreshold), RLX_AGENT);
        __hip_atomic_store(&pub->hist_base, (gu64)C.hist_base, RLX_AGENT);
        __hip_atomic_store(&pub->m_slice, (gu64)m | ((gu64)C.slice_len << 32), RLX_AGENT);
        __hip_atomic_store(&pub->hoff_bn, (gu64)C.hoff, RLX_AGENT);
        __hip_atomic_store(&pub->arrived, (gu64)0, RLX_AGENT);
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
    wg_sync_global();                                          // T* is read by everyone below (and the points are drained)
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
__device__ __attribute__((noinline)) void coop_solve_chunk(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr int MS = (KIND == 2) ? 90 : 27;
    constexpr uint32_t CH = (uint32_t)CoopChunk<KIND>::n;
    // While no model has any inlier yet, ACRANSAC extends its budget by one iteration at a time out of the reserve (nIter++, reserve--
    // at it + 1 == nIter): those iterations WILL run, from the same pool, unless one of them finds inliers (then the rest of the chunk
    // is void like behind any pool change).  Sizing the chunk by nIter alone drew them as ~200 chunks of ONE sample each at the end of
    // every pair without a model (5 ms of a homography filter's 7).
    const uint32_t iter0 = S.iter, nIter0 = S.nIter + (S.n_inl == 0 ? S.reserve : 0u);
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

// ---- the next batch: whole iterations of the chunk from chunk_c on, at most b_cap models, inside the iteration budget.  The matrices
// the residuals are taken with -- the model itself (F, H) or F = K2^-T E K1^-1 -- go to the pair's global array as 8-byte agent
// atomics (slice tasks elsewhere read them) and to this workgroup's slice scratch.
template <int KIND>
__device__ __attribute__((noinline)) void coop_form_batch(const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr int MS = (KIND == 2) ? 90 : 27;
    if (tid == 0) {
        uint32_t c = S.chunk_c, n = 0;
        const uint32_t c0 = c;
        const uint32_t budget = S.nIter + (S.n_inl == 0 ? S.reserve : 0u);       // (see coop_solve_chunk)
        while (c < S.chunk_n && S.chunk_iter0 + c < budget) {
            const uint32_t nm = S.nm[c];
            if (c > c0 && n + nm > S.b_cap) break;
            for (uint32_t k = 0; k < nm; ++k) { S.bj_c[n + k] = (uint8_t)c; S.bj_k[n + k] = (uint8_t)k; }
            n += nm; ++c;
        }
        S.b_c0 = c0; S.b_c1 = c; S.b_n = n;
    }
    r3dm_syncthreads();
    if (tid < S.b_n) {
        const double* Mo = C.models + (size_t)S.bj_c[tid] * MS + 9 * (size_t)S.bj_k[tid];
        double M[9], FE[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) M[e] = Mo[e];
        if (KIND == 2) f_from_e(M, S.kinv, S.kinv + 9, FE);
        double* bm_l = reinterpret_cast<double*>(smem + kCoopBmOff) + 9 * tid;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const double v = (KIND == 2) ? FE[e] : M[e];
            bm_l[e] = v;
            __hip_atomic_store(reinterpret_cast<gu64*>(C.bm + 9 * tid + e), (gu64)__double_as_longlong(v), RLX_AGENT);
        }
    }
    r3dm_syncthreads();
}

// ---- one slice of a batch: residuals of the slice's matches for all models, histogram + count per model into the slice's slot.
// Uses the slice scratch of the LDS only (never CoopS): a leader runs it for other pairs while it waits for its own slices.
struct CoopSliceArgs { uint32_t pt_off, lo, hi, slot, b_n; double maxThreshold; long long hist_base; const double* bm_g; };

template <int KIND>
__device__ __attribute__((noinline)) void coop_eval_slice(const FilterParams& P, const CoopSliceArgs& A, unsigned char* smem, bool bm_in_lds, uint32_t tid)
{
    const uint32_t lane = tid & 63u, b_n = A.b_n;
    double* bm_l = reinterpret_cast<double*>(smem + kCoopBmOff);
    uint32_t* cnt_l = reinterpret_cast<uint32_t*>(smem + kCoopCntOff);
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kCoopHdr);                    // [b_n][512]: bins 2w (low half), 2w + 1 (high half)
    if (!bm_in_lds)
        for (uint32_t e = tid; e < 9 * b_n; e += kCoopNT)
            bm_l[e] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const gu64*>(A.bm_g + e), RLX_AGENT));
    if (tid < (uint32_t)kCoopB) cnt_l[tid] = 0u;
    for (uint32_t e = tid; e < b_n * 512u; e += kCoopNT) hist[e] = 0u;
    r3dm_syncthreads();
    const __amdgpu_buffer_rsrc_t rp = coop_rsrc(P.pts_scratch);
    const double maxThreshold = A.maxThreshold;
    for (uint32_t base = A.lo; base < A.hi; base += 2u * kCoopNT) {
        double px[2][4]; bool valid[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t p = base + (uint32_t)kCoopNT * (uint32_t)u + tid;
            valid[u] = p < A.hi;
            const uint32_t off = A.pt_off + 32u * (valid[u] ? p : A.lo);
            const u32x4 a = ld16_sc1(rp, off), b = ld16_sc1(rp, off + 16u);
            px[u][0] = __longlong_as_double((long long)(((unsigned long long)a.y << 32) | a.x));
            px[u][1] = __longlong_as_double((long long)(((unsigned long long)a.w << 32) | a.z));
            px[u][2] = __longlong_as_double((long long)(((unsigned long long)b.y << 32) | b.x));
            px[u][3] = __longlong_as_double((long long)(((unsigned long long)b.w << 32) | b.z));
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
                    long long bin = (__double_as_longlong(r) >> kHistShift) - A.hist_base;
                    bin = bin < 0 ? 0 : (bin > kHistBins - 1 ? kHistBins - 1 : bin);
                    atomicAdd(&hist[j * 512u + (uint32_t)(bin >> 1)], (bin & 1) ? 0x10000u : 1u);
                }
                n_new += (uint32_t)__builtin_popcountll(__ballot(in));
            }
            if (n_new != 0u && lane == 0) atomicAdd(&cnt_l[j], n_new);
        }
    }
    r3dm_syncthreads();
    // the slot, write-through: counts as 8-byte atomics, histograms of the models that have a match inside the bound as 16-byte
    // sc1 stores (128 threads per model, four models per pass); histograms of the other models are never read
    gu64* gc = reinterpret_cast<gu64*>(P.coop_cnt + (size_t)A.slot * kCoopB);
    if (tid < (uint32_t)kCoopB / 2) __hip_atomic_store(&gc[tid], (gu64)cnt_l[2 * tid] | ((gu64)cnt_l[2 * tid + 1] << 32), RLX_AGENT);
    {
        const __amdgpu_buffer_rsrc_t rh = coop_rsrc(P.coop_hist + (size_t)A.slot * kCoopB * 512);
        const uint32_t sub = tid >> 7, w4 = tid & 127u;
        for (uint32_t j = sub; j < b_n; j += 4u) {
            if (cnt_l[j] == 0u) continue;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&hist[j * 512u + 4u * w4]);
            st16_sc1(rh, (j * 512u + 4u * w4) * 4u, v);
        }
    }
    DRAIN_VMEM();                                              // every storing wave, before the barrier in front of the arrival count
    r3dm_syncthreads();
}

// a slice task: kind << 30 | pair << 5 | slice; everything else about it comes from the pair's published record
template <int KIND>
__device__ __attribute__((noinline)) void coop_run_task_k(const FilterParams& P, unsigned char* smem, uint32_t task, uint32_t tid)
{
    const uint32_t cp = (task & 0x3FFFFFFFu) >> 5, slice = task & 31u;
    CoopPub* pub = reinterpret_cast<CoopPub*>(P.coop_pub) + cp;
    const unsigned long long t0 = PROF_NOW();
    CoopSliceArgs A;
    const gu64 ms = QLOAD(&pub->m_slice), hb = QLOAD(&pub->hoff_bn);
    const uint32_t m = (uint32_t)ms, slice_len = (uint32_t)(ms >> 32);
    A.pt_off = (uint32_t)QLOAD(&pub->pt);
    A.maxThreshold = __longlong_as_double((long long)QLOAD(&pub->max_thr_bits));
    A.hist_base = (long long)QLOAD(&pub->hist_base);
    A.lo = slice * slice_len; A.hi = A.lo + slice_len; if (A.hi > m) A.hi = m;
    A.slot = (uint32_t)hb + slice; A.b_n = (uint32_t)(hb >> 32);
    A.bm_g = P.coop_bm + (size_t)cp * kCoopB * 9;
#ifdef R3DM_DEVTOOLS
    const unsigned long long t_pub = (P.coop_prof && tid == 0) ? QLOAD(&pub->pad[0]) : 0ull;      // (before the arrival: the next batch overwrites it)
#endif
    coop_eval_slice<KIND>(P, A, smem, false, tid);
    if (tid == 0) __hip_atomic_fetch_add(&pub->arrived, (gu64)1, RLX_AGENT);
    PROF_PUT(cp, 3, PROF_NOW() - t0);
#ifdef R3DM_DEVTOOLS
    if (P.coop_prof && tid == 0) {                             // [12] sum and [14] max of publish -> start, [13] max slice duration
        const unsigned long long d = t0 > t_pub ? t0 - t_pub : 0ull, dur = PROF_NOW() - t0;
        atomicAdd(&P.coop_prof[16 * (size_t)cp + 12], d);
        atomicMax(&P.coop_prof[16 * (size_t)cp + 14], d);
        atomicMax(&P.coop_prof[16 * (size_t)cp + 13], dur);
    }
#endif
}
// (params: the three FilterParams of the call in global memory, indexed by model kind -- one pool of workers serves the F, E and H
// filters of a putative graph: with a kernel per kind, the first one launched held every CU with mostly idle helpers while the
// other two waited for its workers to retire)
__device__ __forceinline__ void coop_run_task(const FilterParams* params, unsigned char* smem, uint32_t task, uint32_t tid)
{
    switch (task >> 30) {
        case 0: coop_run_task_k<0>(params[0], smem, task, tid); break;
        case 1: coop_run_task_k<1>(params[1], smem, task, tid); break;
        default: coop_run_task_k<2>(params[2], smem, task, tid); break;
    }
}

// ---- bitonic sort of the (residual, index) lists (global memory), blocks of kCoopNT * E entries: E entries per thread in registers,
// exchanges between waves through an LDS image of the block (region R of the workgroup's LDS: 8192 x 12 bytes), global memory touched
// once to load the block and once to store it.  (The first form exchanged through the global lists themselves: ~20 round trips per
// sort, 150 us per full evaluation on an idle device and 3-4x that with the other 255 workgroups loading the memory system.)
// The stages `size_from .. size_to` of the network on the block that starts at entry i0 (directions from the GLOBAL index, so blocks
// sorted one after the other come out alternately ascending / descending, as the merge stages behind them need).  Strides >= the
// block are not this routine's business.  Same exchange rules as wg_sort_regs (kernels_filter.hip).  The caller has made the
// lists visible to the workgroup (wg_sync_t<true>) and finds them visible again on return.
template <int E>
__device__ __forceinline__ void coop_sort_block(unsigned long long* __restrict__ keys, uint32_t* __restrict__ sidx,
                                                unsigned long long* __restrict__ lk, uint32_t* __restrict__ lx, uint32_t i0,
                                                uint32_t size_from, uint32_t size_to, uint32_t total, uint32_t tid)
{
    constexpr uint32_t BLK = (uint32_t)kCoopNT * (uint32_t)E;
    unsigned long long k[E];
    uint32_t x[E];
    const uint32_t base = tid * (uint32_t)E;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const uint32_t i = i0 + base + (uint32_t)s;
        const bool live = i < total || size_from > 2u;          // (merge stages: the padding was materialised by the block sorts)
        k[s] = live ? keys[i] : ~0ull;
        x[s] = live ? sidx[i] : 0xFFFFFFFFu;
    }
    for (uint32_t size = size_from; size <= size_to; size <<= 1) {
        uint32_t stride = size >> 1;
        if (stride >= BLK) stride = BLK >> 1;
        // ---- distances that cross waves: every thread parks its entries in the LDS image, reads the partner's
        for (; stride >= 64u * E; stride >>= 1) {
            r3dm_syncthreads();                                    // earlier readers of the image are done
#pragma unroll
            for (int s = 0; s < E; ++s) { lk[base + s] = k[s]; lx[base + s] = x[s]; }
            r3dm_syncthreads();
#pragma unroll
            for (int s = 0; s < E; ++s) {
                const uint32_t li = base + (uint32_t)s, i = i0 + li;
                const unsigned long long ok = lk[li ^ stride];
                const uint32_t ox = lx[li ^ stride];
                const bool lower = (i & stride) == 0u, up = (i & size) == 0u;
                const bool mine_gt = pair_gt(k[s], x[s], ok, ox);
                if (mine_gt == (lower == up)) { k[s] = ok; x[s] = ox; }
            }
        }
        // ---- distances inside the wave
        for (; stride >= (uint32_t)E; stride >>= 1) {
            const int lane_xor = (int)(stride / (uint32_t)E);
#pragma unroll
            for (int s = 0; s < E; ++s) {
                const uint32_t i = i0 + base + (uint32_t)s;
                const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)k[s], lane_xor);
                const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(k[s] >> 32), lane_xor);
                const uint32_t ox = (uint32_t)__shfl_xor((int)x[s], lane_xor);
                const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;
                const bool lower = (i & stride) == 0u, up = (i & size) == 0u;
                const bool mine_gt = pair_gt(k[s], x[s], ok, ox);
                if (mine_gt == (lower == up)) { k[s] = ok; x[s] = ox; }
            }
        }
        // ---- distances inside the thread
#pragma unroll
        for (int ST = E / 2; ST >= 1; ST >>= 1) {
            if ((uint32_t)ST < size) {
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    if ((s & ST) == 0) {
                        const bool up = ((i0 + base + (uint32_t)s) & size) == 0u;
                        const bool gt = pair_gt(k[s], x[s], k[s | ST], x[s | ST]);
                        if (gt == up) {
                            const unsigned long long tk = k[s]; k[s] = k[s | ST]; k[s | ST] = tk;
                            const uint32_t tx = x[s]; x[s] = x[s | ST]; x[s | ST] = tx;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < E; ++s) { keys[i0 + base + s] = k[s]; sidx[i0 + base + s] = x[s]; }
    wg_sync_t<true>();
}

// the whole list: up to 8192 entries as one block of 1 .. 16 entries per thread; longer lists as blocks of 8192 (16 per thread: the
// 32-per-thread form of the one-workgroup kernel needs more registers than a 512-thread workgroup leaves without spilling into the
// loops around it) + merge stages whose block-crossing strides run on the plain network in global memory
__device__ __attribute__((noinline)) void coop_sort(unsigned long long* __restrict__ keys, uint32_t* __restrict__ sidx, unsigned char* smem,
                                                    uint32_t total, uint32_t tid)
{
    unsigned long long* lk = reinterpret_cast<unsigned long long*>(smem + kCoopHdr);          // [8192]
    uint32_t* lx = reinterpret_cast<uint32_t*>(smem + kCoopHdr + 8192 * 8);                    // [8192]
    uint32_t cap = 1; while (cap < total) cap <<= 1;
    constexpr uint32_t BLK = (uint32_t)kCoopNT * 16u;
    if (cap <= BLK) {
        switch (cap / (uint32_t)kCoopNT) {
            case 0: case 1: coop_sort_block<1>(keys, sidx, lk, lx, 0u, 2u, cap, total, tid); break;
            case 2: coop_sort_block<2>(keys, sidx, lk, lx, 0u, 2u, cap, total, tid); break;
            case 4: coop_sort_block<4>(keys, sidx, lk, lx, 0u, 2u, cap, total, tid); break;
            case 8: coop_sort_block<8>(keys, sidx, lk, lx, 0u, 2u, cap, total, tid); break;
            default: coop_sort_block<16>(keys, sidx, lk, lx, 0u, 2u, cap, total, tid); break;
        }
        return;
    }
    for (uint32_t i0 = 0; i0 < cap; i0 += BLK) coop_sort_block<16>(keys, sidx, lk, lx, i0, 2u, BLK, total, tid);
    for (uint32_t size = 2u * BLK; size <= cap; size <<= 1) {
        for (uint32_t stride = size >> 1; stride >= BLK; stride >>= 1) {
            for (uint32_t tI = tid; tI < (cap >> 1); tI += kCoopNT) {
                const uint32_t lo = 2 * tI - (tI & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                const uint32_t ai = sidx[lo], bi = sidx[hi];
                if (pair_gt(a, ai, b, bi) == up) { keys[lo] = b; keys[hi] = a; sidx[lo] = bi; sidx[hi] = ai; }
            }
            wg_sync_t<true>();
        }
        for (uint32_t i0 = 0; i0 < cap; i0 += BLK) coop_sort_block<16>(keys, sidx, lk, lx, i0, size, size, total, tid);
    }
}

// ---- the same ascending (residual, index) order by BUCKETS: residuals are >= 0, so their IEEE bit patterns order them; the top 17
// bits (exponent + 6 mantissa bits: 64 bins per octave) counted down from the pattern of the bound give 2048 bins that cover 32
// octaves below it (everything smaller shares bin 0).  Count, prefix, scatter into the spare half of the pair's sort arrays grouped
// by bin, then every entry ranks itself among the members of its own bin and goes to its final place.  ~12 entries per bin on a
// 12 k-match pair: three passes over the list and ~30 reads per entry where the bitonic network above makes 105 passes of which the
// block-crossing ones run through global memory -- 270 -> ~60 us per evaluation, and a long pair's AC-RANSAC spends half its time in
// its ~13 full evaluations.  The order is a pure function of the (key, index) pairs (distinct: the index breaks ties), so the
// result is the network's.  Returns false -- nothing moved -- when the residuals crowd few bins (sum of squared bin counts above
// 64 per entry: e.g. an exact synthetic scene whose inliers all have residual 0); the caller then runs the network.
constexpr uint32_t kBktBins = 2048;
__device__ __forceinline__ uint32_t coop_bkt_bin(unsigned long long key, uint32_t thr_top)
{
    const uint32_t top = (uint32_t)(key >> 46);
    const uint32_t lo = thr_top >= (kBktBins - 1u) ? thr_top - (kBktBins - 1u) : 0u;
    const uint32_t b = top > lo ? top - lo : 0u;
    return b < kBktBins ? b : kBktBins - 1u;
}
__device__ __attribute__((noinline)) bool coop_bucket_sort(unsigned long long* __restrict__ keys, uint32_t* __restrict__ sidx,
                                                           unsigned long long* __restrict__ tk, uint32_t* __restrict__ ti, unsigned char* smem,
                                                           CoopS& S, uint32_t total, double maxThreshold, uint32_t tid)
{
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + kCoopHdr);            // [kBktBins]
    uint32_t* start = cnt + kBktBins;                                         // [kBktBins]
    uint32_t* fill = start + kBktBins;                                        // [kBktBins]
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t thr_top = (uint32_t)((unsigned long long)__double_as_longlong(maxThreshold) >> 46);
    for (uint32_t b = tid; b < kBktBins; b += kCoopNT) { cnt[b] = 0u; fill[b] = 0u; }
    r3dm_syncthreads();
    for (uint32_t i = tid; i < total; i += kCoopNT) atomicAdd(&cnt[coop_bkt_bin(keys[i], thr_top)], 1u);
    r3dm_syncthreads();
    // exclusive prefix over the bins (kBktBins / kCoopNT consecutive bins per thread) + the crowding measure
    constexpr uint32_t PER = kBktBins / (uint32_t)kCoopNT;
    uint32_t c[PER], mine = 0u;
    double sq = 0.0;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) { c[u] = cnt[tid * PER + u]; mine += c[u]; sq += (double)c[u] * (double)c[u]; }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if ((int)lane >= off) incl += o; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 63u) S.red_k[wave] = incl;
    if (lane == 0u) S.red_v[wave] = sq;
    r3dm_syncthreads();
    uint32_t wbase = 0u; double sq_all = 0.0;
#pragma unroll
    for (uint32_t w = 0; w < kCoopNW; ++w) { if (w < wave) wbase += S.red_k[w]; sq_all += S.red_v[w]; }
    r3dm_syncthreads();                                                       // red_v / red_k are reused by the caller
    if (sq_all > 64.0 * (double)total) return false;                          // workgroup-uniform
    uint32_t run = wbase + incl - mine;
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) { start[tid * PER + u] = run; run += c[u]; }
    r3dm_syncthreads();
    for (uint32_t i = tid; i < total; i += kCoopNT) {
        const unsigned long long k = keys[i];
        const uint32_t b = coop_bkt_bin(k, thr_top);
        const uint32_t pos = start[b] + atomicAdd(&fill[b], 1u);
        tk[pos] = k; ti[pos] = sidx[i];
    }
    wg_sync_t<true>();
    for (uint32_t i = tid; i < total; i += kCoopNT) {
        const unsigned long long k = tk[i];
        const uint32_t x = ti[i];
        const uint32_t b = coop_bkt_bin(k, thr_top);
        const uint32_t s0 = start[b], e0 = s0 + cnt[b];
        uint32_t r = 0u;
        for (uint32_t j = s0; j < e0; ++j) r += pair_gt(k, x, tk[j], ti[j]) ? 1u : 0u;
        keys[s0 + r] = k; sidx[s0 + r] = x;
    }
    wg_sync_t<true>();
    return true;
}

// ---- full evaluation of ONE model by the leader: residuals of all matches, compaction of those within the bound, sort, NFA scan
// (the evaluation block of acransac_body with its lists in global memory).  Returns the model's NFA and inlier count.
template <int KIND>
__device__ void coop_full_eval(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, const double* M /* residual matrix */,
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
        uint32_t cap2 = 1u; while (cap2 < m) cap2 <<= 1;                     // the pair's arrays hold 2 x next_pow2(m) entries: [sort | spare]
        if (!coop_bucket_sort(keys, sidx, keys + cap2, sidx + cap2, smem, S, total, maxThreshold, tid))
            coop_sort(keys, sidx, smem, total, tid);
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

// ---- merged histograms -> count and NFA bound of every model of the batch (one wave per model, eight at a time).  The slot counts
// of all slices land in LDS first (one pass, every load in flight at once); a model's histograms are then fetched four slices at a
// time before any of them is used (the first form chained count -> branch -> histogram per slice: ~1.7 us per model in load latency).
template <int KIND>
__device__ void coop_batch_bounds(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    const uint32_t lane = tid & 63u, wave = tid >> 6, G = C.G;
    uint32_t* cnt_all = reinterpret_cast<uint32_t*>(smem + kCoopAllCntOff);          // [G][kCoopB]
    {
        const gu64* gc = reinterpret_cast<const gu64*>(P.coop_cnt + (size_t)C.hoff * kCoopB);
        for (uint32_t e = tid; e < G * (uint32_t)kCoopB / 2u; e += kCoopNT) {
            const gu64 v = QLOAD(&gc[e]);
            cnt_all[2 * e] = (uint32_t)v; cnt_all[2 * e + 1] = (uint32_t)(v >> 32);
        }
    }
    r3dm_syncthreads();
    const __amdgpu_buffer_rsrc_t rh = coop_rsrc(P.coop_hist + (size_t)C.hoff * kCoopB * 512);
    for (uint32_t j = wave; j < S.b_n; j += kCoopNW) {
        uint32_t total = 0;
        for (uint32_t s = 0; s < G; ++s) total += cnt_all[s * kCoopB + j];
        double wmin = __builtin_huge_val();
        if (total > SS) {
            uint32_t h[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = 0u;
            for (uint32_t s0 = 0; s0 < G; s0 += 4u) {
                u32x4 a[4], b[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t s = s0 + u;
                    const bool have = s < G && cnt_all[(s < G ? s : 0u) * kCoopB + j] != 0u;      // (slot not written otherwise)
                    const uint32_t off = (((have ? s : 0u) * (uint32_t)kCoopB + j) * 512u + 8u * lane) * 4u;
                    a[u] = ld16_sc1(rh, off); b[u] = ld16_sc1(rh, off + 16u);
                    if (!have) { a[u] = u32x4{0u, 0u, 0u, 0u}; b[u] = u32x4{0u, 0u, 0u, 0u}; }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t w[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) { h[2 * i] += w[i] & 0xFFFFu; h[2 * i + 1] += w[i] >> 16; }
                }
            }
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) run += h[i];
            uint32_t incl = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
            // every table read of the lane's 16 bins is issued before the first one is used (clamped indices, no branch around a load:
            // sixteen dependent trips to memory per model were most of this routine's time)
            const uint32_t k0 = incl - run;
            double la[16], tl[16], th[16];
            {
                const double2* lp = reinterpret_cast<const double2*>(C.la_tab + 16u * lane);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const double2 v = lp[i]; la[2 * i] = v.x; la[2 * i + 1] = v.y; }
                uint32_t k_prev = k0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t k_hi = k_prev + h[i];
                    uint32_t k_lo = k_prev + 1u; if (k_lo < SS + 1u) k_lo = SS + 1u;
                    tl[i] = C.tstar[k_lo <= C.m ? k_lo : C.m];
                    th[i] = C.tstar[k_hi <= C.m ? k_hi : C.m];
                    k_prev = k_hi;
                }
            }
            uint32_t k_prev = k0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t k_hi = k_prev + h[i];
                uint32_t k_lo = k_prev + 1u; if (k_lo < SS + 1u) k_lo = SS + 1u;
                if (h[i] != 0u && k_hi >= k_lo) {
                    const double v_lo = la[i] * (double)(k_lo - SS) + tl[i];
                    const double v_hi = la[i] * (double)(k_hi - SS) + th[i];
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
// merged count / bound and, for the few models that can win, coop_full_eval).  Every thread walks with its own copy of the scalars
// the walk changes (all threads hold the same values: everything comes from LDS that only barriers separate from its writer);
// a model that is skipped costs no barrier, thread 0 writes the scalars back where the workgroup meets anyway.
template <int KIND>
__device__ __attribute__((noinline)) void coop_decide(const FilterParams& P, const CoopCtx<KIND>& C, CoopS& S, unsigned char* smem, uint32_t tid)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    constexpr int MS = (KIND == 2) ? 90 : 27;
    constexpr int NT = kCoopNT;
    const uint32_t lane = tid & 63u, wave = tid >> 6, m = C.m;
    [[maybe_unused]] const uint32_t item = C.item;                 // (FCHECK reports it)
    const unsigned long long pt0 = PROF_NOW();
    if (S.b_n) coop_batch_bounds<KIND>(P, C, S, smem, tid);
    const unsigned long long pt1 = PROF_NOW();
    PROF_PUT(C.cp, 5, pt1 - pt0);
    [[maybe_unused]] unsigned long long prof_full = 0;
    // walk-local copies
    double minNFA = S.minNFA;
    uint32_t acMode = S.acMode, n_models = S.n_models, n_inl = S.n_inl, nIter = S.nIter, reserve = S.reserve, pool_size = S.pool_size;
    uint32_t iters_done = S.iters_done;
    const uint32_t chunk_iter0 = S.chunk_iter0, c1 = S.b_c1;
    bool pool_changed = false;
    uint32_t j = 0;
    uint32_t c = S.b_c0;
    for (; c < c1 && !pool_changed; ++c) {
        const uint32_t it = chunk_iter0 + c;
        const uint32_t nm = S.nm[c];
        bool better = false;
        for (uint32_t k = 0; k < nm; ++k, ++j) {
            const uint32_t total = S.tot[j];
            bool ac = acMode != 0;
            if (!ac && (double)total > 2.5 * SS) ac = true;
            acMode = ac ? 1u : 0u;
            n_models += 1;
            if (!(ac && total > SS)) continue;
            const double bound = C.loge0 + S.bnd[j] - S.eps_T;
            const bool hopeless = !R3DM_DBG(P) && (minNFA < __builtin_huge_val()) && (bound - 1.0e-6 >= minNFA);
            if (hopeless) continue;
            // ---- a model that can win: the full evaluation (the workgroup meets here)
            double Mr[9];
            const double* bm_g = C.bm + 9 * (size_t)j;
#pragma unroll
            for (int e = 0; e < 9; ++e) Mr[e] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const gu64*>(bm_g + e), RLX_AGENT));
            double nfa; uint32_t kbest, total2 = 0;
            const unsigned long long pf0 = PROF_NOW();
            coop_full_eval<KIND>(P, C, S, smem, Mr, tid, nfa, kbest, total2);
            prof_full += PROF_NOW() - pf0;
            PROF_PUT(C.cp, 7, 1);
            FCHECK(total2 == total, 9, total2, total);                 // the slices and the full pass count the same matches
            FCHECK(bound - 1.0e-6 <= nfa, 8, kbest, total);           // the sort-skipping bound really is one
            if (nfa < minNFA) {
                for (uint32_t q = tid; q < kbest; q += NT) C.inl[q] = C.sidx[q];
                better = true;
                minNFA = nfa; n_inl = kbest;
                wg_sync_t<true>();
                if (tid == 0) {
                    const double* Mo = C.models + (size_t)c * MS + 9 * (size_t)k;     // the model itself (E for KIND 2)
                    S.errorMax = __longlong_as_double((long long)C.keys[kbest - 1]);
#pragma unroll
                    for (int e = 0; e < 9; ++e) S.bestF[e] = Mo[e];
                }
            }
        }
        // ---- end of iteration `it`: ACRANSAC's pool / budget update
        iters_done = it + 1;
        bool rebuild = false;
        const bool trigger = (better && minNFA < 0.0) || (it + 1 == nIter && reserve != 0);
        if (trigger) {
            if (n_inl == 0) { nIter += 1; reserve -= 1; }
            else {
                rebuild = true;
                pool_size = n_inl;
                if (reserve) { nIter = it + 1 + reserve; reserve = 0; }
            }
        }
        if (rebuild) {
            // new sampling pool = the inlier SET in ascending index order (same rule as acransac_body / oracle/acransac.c)
            const uint32_t ni = n_inl;
            uint32_t* flags = C.sidx;                                // sort scratch, free between models
            wg_sync_t<true>();
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
        }
        if (it + 1 >= nIter) { ++c; break; }
    }
    r3dm_syncthreads();                                             // every thread has read the batch's LDS tables
    if (tid == 0) {
        S.minNFA = minNFA; S.acMode = acMode; S.n_models = n_models; S.n_inl = n_inl; S.nIter = nIter; S.reserve = reserve;
        S.pool_size = pool_size; S.iters_done = iters_done;
        S.iter = chunk_iter0 + c;
        S.chunk_c = c;
        if (pool_changed) { S.chunk_valid = 0; S.b_cap = 12; }
        else { const uint32_t nb = S.b_cap * 2; S.b_cap = nb > (uint32_t)kCoopB ? (uint32_t)kCoopB : nb; }
    }
    wg_sync_global();
    PROF_PUT(C.cp, 6, prof_full);
    PROF_PUT(C.cp, 8, PROF_NOW() - pt1 - prof_full);
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

// claim up to `need` idle workers (their bits leave the bitmap): one load + one fetch-and per bitmap word that has candidates
__device__ __forceinline__ uint32_t coop_claim_idle(const CoopSched& Q, uint32_t first_word, uint32_t need, uint8_t* ids)
{
    uint32_t got_n = 0;
    for (uint32_t k = 0; k < 4u && need; ++k) {
        const uint32_t w = (first_word + k) & 3u;
        gu64* word = &Q.idle[w];
        for (int tries = 0; tries < 2 && need; ++tries) {
            gu64 avail = QLOAD(word);
            if (!avail) break;
            gu64 mask = 0;
            for (uint32_t n = 0; n < need && avail; ++n) { const gu64 low = avail & (~avail + 1ull); mask |= low; avail ^= low; }
            gu64 mine = __hip_atomic_fetch_and(word, ~mask, RLX_AGENT) & mask;
            while (mine) {
                const uint32_t bit = (uint32_t)__builtin_ctzll(mine);
                mine &= mine - 1ull;
                ids[got_n++] = (uint8_t)(64u * w + bit);
                --need;
            }
        }
    }
    return got_n;
}

// ---- a pair from start-up to result, led by this workgroup
template <int KIND>
__device__ __attribute__((noinline)) void coop_lead_pair(const FilterParams* params, unsigned char* smem, uint32_t cp, uint32_t tid)
{
    const FilterParams& P = params[KIND];
    CoopS& S = *reinterpret_cast<CoopS*>(smem);
    uint32_t* q = P.coop_q;
    const CoopSched Q = coop_sched(q);
    CoopCtx<KIND> C;
    coop_ctx<KIND>(P, cp, C);
    const unsigned long long t_start = PROF_NOW();
    coop_init<KIND>(P, C, S, tid);
    PROF_PUT(cp, 0, PROF_NOW() - t_start);
    bool decide_first = false;
    for (uint32_t turn = 0;; ++turn) {
        if (decide_first) coop_decide<KIND>(P, C, S, smem, tid);
        decide_first = true;
        if (S.iter >= S.nIter) break;
        if (turn > 4u * P.max_iter + 64u) { if (tid == 0) coop_report_stall(q, 4u, cp); break; }   // (every turn consumes an iteration or draws a chunk)
        if (!S.chunk_valid || S.chunk_c >= S.chunk_n) {
            const unsigned long long t0 = PROF_NOW();
            coop_solve_chunk<KIND>(P, C, S, smem, tid);
            PROF_PUT(cp, 1, PROF_NOW() - t0);
        }
        const unsigned long long t1 = PROF_NOW();
        coop_form_batch<KIND>(C, S, smem, tid);
        const uint32_t b_n = S.b_n;
        if (b_n == 0) continue;                                    // iterations without a model: only their bookkeeping
        PROF_PUT(cp, 9, 1); PROF_PUT(cp, 11, b_n);
        // ---- the batch's slices: those that an idle worker can be found for are handed to those workers' mailboxes; the leader runs
        // slice 0 and every slice nobody is idle for (then every CU has work anyway).  Nothing ever waits in a queue, and a leader
        // waits only for slices that ARE running on some worker.
        uint32_t n_handed = 0;
        if (C.G > 1u) {
            DRAIN_VMEM();                                          // the batch matrices (and, first batch, the points and the record)
            r3dm_syncthreads();
            if (tid == 0) {
                __hip_atomic_store(&C.pub->hoff_bn, (gu64)C.hoff | ((gu64)b_n << 32), RLX_AGENT);
                __hip_atomic_store(&C.pub->arrived, (gu64)0, RLX_AGENT);
#ifdef R3DM_DEVTOOLS
                if (P.coop_prof) __hip_atomic_store(&C.pub->pad[0], (gu64)wall_clock64(), RLX_AGENT);
#endif
                DRAIN_VMEM();
                S.n_claim = coop_claim_idle(Q, cp, C.G - 1u, S.claim);
                S.kept = ((1u << C.G) - 1u) & ~1u;                  // slices 1 .. G - 1 (bit s), cleared as they are handed over
            }
            r3dm_syncthreads();
            if (tid < S.n_claim) {
                const uint32_t slice = tid + 1u;
                const uint32_t task = ((uint32_t)KIND << 30) | (cp << 5) | slice;
                if (mbox_cas(&Q.mbox[32u * S.claim[tid]], kMboxEmpty, task + 1u)) atomicAnd(&S.kept, ~(1u << slice));   // (fails on a worker that has just retired)
            }
            r3dm_syncthreads();
        }
        PROF_PUT(cp, 2, PROF_NOW() - t1);
        const uint32_t pt_off = (uint32_t)((const char*)C.pt - (const char*)P.pts_scratch);
        {
            const unsigned long long t2 = PROF_NOW();
            CoopSliceArgs A;
            A.pt_off = pt_off;
            A.lo = 0u; A.hi = C.slice_len < C.m ? C.slice_len : C.m;
            A.slot = C.hoff; A.b_n = b_n; A.maxThreshold = C.maxThreshold; A.hist_base = C.hist_base; A.bm_g = C.bm;
            coop_eval_slice<KIND>(P, A, smem, true, tid);
            PROF_PUT(cp, 3, PROF_NOW() - t2);
        }
        if (C.G > 1u) {
            uint32_t kept = S.kept;
            n_handed = C.G - 1u - (uint32_t)__builtin_popcount(kept);
            // the kept slices, one after the other; before each, one more look for a worker that has become idle meanwhile
            while (kept) {
                const uint32_t slice = (uint32_t)__builtin_ctz(kept);
                kept &= kept - 1u;
                r3dm_syncthreads();
                if (tid == 0) {
                    uint8_t id;
                    uint32_t handed = 0;
                    if (coop_claim_idle(Q, cp + slice, 1u, &id) == 1u)
                        handed = mbox_cas(&Q.mbox[32u * id], kMboxEmpty, (((uint32_t)KIND << 30) | (cp << 5) | slice) + 1u) ? 1u : 0u;
                    S.sh_aux = handed;
                }
                r3dm_syncthreads();
                if (S.sh_aux) { ++n_handed; continue; }
                const unsigned long long t2 = PROF_NOW();
                CoopSliceArgs A;
                A.pt_off = pt_off;
                A.lo = slice * C.slice_len; A.hi = A.lo + C.slice_len; if (A.hi > C.m) A.hi = C.m;
                A.slot = C.hoff + slice; A.b_n = b_n; A.maxThreshold = C.maxThreshold; A.hist_base = C.hist_base; A.bm_g = C.bm;
                coop_eval_slice<KIND>(P, A, smem, true, tid);
                PROF_PUT(cp, 3, PROF_NOW() - t2);
            }
            // wait for the slices that were handed over (each is running on a worker): only the pair's own arrival word is polled
            const unsigned long long t3 = PROF_NOW();
            if (tid == 0) {
                uint32_t v = kCoopDone;
                const unsigned long long w0 = wall_clock64();
                for (uint32_t spins = 1; QLOAD(&C.pub->arrived) < (gu64)n_handed; ++spins) {
                    __builtin_amdgcn_s_sleep(12);
                    if ((spins & 63u) == 0u) {
                        if (coop_stalled(q)) { v = kCoopStall; break; }
                        if ((unsigned long long)wall_clock64() - w0 > kCoopStallTicks) { coop_report_stall(q, 2u, cp); v = kCoopStall; break; }
                    }
                }
                S.sh_task = v;
            }
            r3dm_syncthreads();
            const uint32_t t = S.sh_task;
            r3dm_syncthreads();
            PROF_PUT(cp, 4, PROF_NOW() - t3);
            if (t == kCoopStall) break;                            // (the host reports the call as failed)
        }
    }
    coop_finish<KIND>(P, C, S, tid);
    PROF_PUT(cp, 10, PROF_NOW() - t_start);
    r3dm_syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_sub(&q[kQPot], C.G, RLX_AGENT);
        __hip_atomic_fetch_add(&q[kQDone], 1u, RLX_AGENT);
    }
}

__global__ __launch_bounds__(kCoopNT, 1)
void acransac_coop_kernel(const FilterParams* __restrict__ params /* [3], by model kind */, uint32_t* q, const uint32_t* __restrict__ start)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    CoopS& S = *reinterpret_cast<CoopS*>(smem);
    const uint32_t tid = threadIdx.x;
    const CoopSched Q = coop_sched(q);
    const uint32_t me = blockIdx.x;
    for (;;) {
        // ---- what next: a pair nobody leads yet; else wait at the mailbox for a slice, retire, or leave
        if (tid == 0) {
            uint32_t v = kCoopNoTask, what = 0;
            const uint32_t n_pairs = q[5];
            if (coop_stalled(q)) what = 0;
            else {
                // a new pair, unless enough pairs are being led already: q[21] = pairs in flight at most (about half of the workers, so
                // that the other half is there to take slices)
                const uint32_t started = QLOAD(&q[kQNextPair]);
                if (started < n_pairs && started - QLOAD(&q[kQDone]) < q[21]) {
                    const uint32_t t = __hip_atomic_fetch_add(&q[kQNextPair], 1u, RLX_AGENT);
                    if (t < n_pairs) { v = start[t]; what = 2; }
                }
                if (what == 0 && QLOAD(&q[kQDone]) != n_pairs) {
                    uint32_t* mine = &Q.mbox[32u * me];
                    __hip_atomic_store(mine, kMboxEmpty, RLX_AGENT);
                    DRAIN_VMEM();
                    __hip_atomic_fetch_or(&Q.idle[me >> 6], 1ull << (me & 63u), RLX_AGENT);
                    const unsigned long long w0 = wall_clock64();
                    for (uint32_t spins = 1;; ++spins) {
                        const uint32_t mb = QLOAD(mine);
                        if (mb != kMboxEmpty) { v = mb - 1u; what = 1; break; }
                        __builtin_amdgcn_s_sleep(24);                           // ~0.6 us between looks at the (private) line
                        if ((spins & 31u) != 0u) continue;
                        // every ~20 us: has the call ended, are there more workers than tasks can exist, has something stalled?
                        bool leave = coop_stalled(q) || QLOAD(&q[kQDone]) == n_pairs;
                        if (!leave) {
                            const uint32_t st = QLOAD(&q[kQNextPair]);
                            if (st < n_pairs && st - QLOAD(&q[kQDone]) < q[21] && mbox_cas(mine, kMboxEmpty, kMboxRetired)) {
                                // a pair can be started (one has finished since this worker went idle): back to the top
                                __hip_atomic_fetch_and(&Q.idle[me >> 6], ~(1ull << (me & 63u)), RLX_AGENT);
                                what = 3; break;
                            }
                        }
                        if (!leave) {
                            const uint32_t a = QLOAD(&q[kQActive]), pot = QLOAD(&q[kQPot]);
                            if (a > pot && mbox_cas(&q[kQActive], a, a - 1u)) leave = true;
                            if (!leave && (unsigned long long)wall_clock64() - w0 > kCoopStallTicks) { coop_report_stall(q, 3u, me); leave = true; }
                        }
                        if (leave && mbox_cas(mine, kMboxEmpty, kMboxRetired)) {       // (a task that arrived meanwhile is taken on the next look)
                            __hip_atomic_fetch_and(&Q.idle[me >> 6], ~(1ull << (me & 63u)), RLX_AGENT);
                            break;
                        }
                    }
                }
            }
            S.sh_task = v; S.sh_aux = what;
        }
        r3dm_syncthreads();
        const uint32_t task = S.sh_task, what = S.sh_aux;
        r3dm_syncthreads();
        if (what == 0) return;
        if (what == 3) continue;
        if (what == 1) coop_run_task(params, smem, task, tid);
        else {
            const uint32_t cp = task & 0x3FFFFFFFu;
            switch (task >> 30) {
                case 0: coop_lead_pair<0>(params, smem, cp, tid); break;
                case 1: coop_lead_pair<1>(params, smem, cp, tid); break;
                default: coop_lead_pair<2>(params, smem, cp, tid); break;
            }
        }
    }
}

size_t filter_coop_lds_bytes() { return std::max(std::max(coop_lds_bytes_(0), coop_lds_bytes_(1)), coop_lds_bytes_(2)); }
hipError_t launch_filter_coop(hipStream_t st, const FilterParams* dev_params, uint32_t* q, const uint32_t* start, uint32_t n_workers)
{
    if (n_workers == 0) return hipSuccess;
    const size_t lds = filter_coop_lds_bytes();
    hipError_t e = hipFuncSetAttribute((const void*)acransac_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(acransac_coop_kernel, dim3(n_workers), dim3(kCoopNT), lds, st, dev_params, q, start);
    return hipGetLastError();
}

}  // namespace r3dm
