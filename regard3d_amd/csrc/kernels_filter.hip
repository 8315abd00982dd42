// kernels_filter.hip -- batched a-contrario RANSAC fundamental-matrix filter on gfx950.
//
// Replaces, behind r3dm_filter_F, the reference's
//   ImageCollectionGeometricFilter::Robust_model_estimation(GeometricFilter_FMatrix_AC(4.0, 2048), ...)
//   (/root/reference/src/R3DComputeMatches.cpp:2099,2113-2115), whose arithmetic is OpenMVG 1.4's
//   ACRANSAC + ACKernelAdaptor<SevenPointSolver, SymmetricEpipolarDistanceError>.
//
// One workgroup per image pair.  AC-RANSAC is sequentially adaptive (the sampling pool shrinks to
// the inlier set on every meaningful improvement and the iteration budget is cut once), so the
// kernel keeps the reference's iteration ORDER and only parallelises inside it:
//   * a chunk of 64 minimal samples is drawn from the counter-based sample stream and solved
//     (7-point, f64, one lane per hypothesis, Householder QR null space + cubic);
//   * the <= 3 models of each iteration are then evaluated one after the other by all 256 lanes:
//     residuals of all m matches, wave-ballot compaction of those under the residual bound,
//     LDS bitonic sort of (residual, index), parallel NFA scan + argmin;
//   * when an iteration changes the pool, the rest of the chunk is discarded and re-drawn.
// Residuals beyond the bound never enter bestNFA's scan (its loop stops at the first one), so
// sorting only the sub-threshold set yields the same (NFA, k, inlier prefix) as the full sort.
//
// f64 throughout, compiled with -ffp-contract=off so residuals agree with the CPU restatement to
// rounding of the transcendental calls (acos/cos/cbrt/log10) only.

#include "r3dm_internal.hpp"
#include <cstdlib>

namespace r3dm {

#define FLT_EPS_D 1.1920928955078125e-07

// ---- sample stream: identical integer arithmetic to oracle/acransac.c orc_rng_u64 ----
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ uint64_t rng_u64(uint64_t seed, uint32_t I, uint32_t J, uint32_t iter, uint32_t attempt)
{
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    const uint64_t a = mix64(seed + G * (1ULL + (((uint64_t)I << 32) | (uint64_t)J)));
    return mix64(a + G * (1ULL + (((uint64_t)iter << 32) | (uint64_t)attempt)));
}

// ---- cubic: roots of c0 + c1 x + c2 x^2 + c3 x^3 (Cardano / trigonometric form) ----
__device__ int solve_cubic(const double* c, double* roots)
{
    if (c[3] == 0.0) return 0;
    const double a = c[2] / c[3], b = c[1] / c[3], cc = c[0] / c[3];
    const double q = a * a - 3.0 * b;
    const double r = 2.0 * a * a * a - 9.0 * a * b + 27.0 * cc;
    const double Q = q / 9.0, R = r / 54.0;
    const double Q3 = Q * Q * Q, R2 = R * R;
    const double CR2 = 729.0 * r * r, CQ3 = 2916.0 * q * q * q;
    const double a3 = a / 3.0;
    if (R == 0.0 && Q == 0.0) {
        roots[0] = roots[1] = roots[2] = -a3;
        return 3;
    }
    if (CR2 == CQ3) {
        const double sqrtQ = sqrt(Q);
        if (R > 0.0) { roots[0] = -2.0 * sqrtQ - a3; roots[1] = sqrtQ - a3; roots[2] = sqrtQ - a3; }
        else         { roots[0] = -sqrtQ - a3; roots[1] = -sqrtQ - a3; roots[2] = 2.0 * sqrtQ - a3; }
        return 3;
    }
    if (CR2 < CQ3) {
        const double sqrtQ = sqrt(Q);
        const double sqrtQ3 = sqrtQ * sqrtQ * sqrtQ;
        const double theta = acos(R / sqrtQ3);
        const double norm = -2.0 * sqrtQ;
        const double TWO_PI = 2.0 * 3.14159265358979323846;
        double x0 = norm * cos(theta / 3.0) - a3;
        double x1 = norm * cos((theta + TWO_PI) / 3.0) - a3;
        double x2 = norm * cos((theta - TWO_PI) / 3.0) - a3;
        double t;
        if (x0 > x1) { t = x0; x0 = x1; x1 = t; }
        if (x1 > x2) { t = x1; x1 = x2; x2 = t; if (x0 > x1) { t = x0; x0 = x1; x1 = t; } }
        roots[0] = x0; roots[1] = x1; roots[2] = x2;
        return 3;
    }
    const double sgnR = (R >= 0.0 ? 1.0 : -1.0);
    const double A = -sgnR * cbrt(fabs(R) + sqrt(R2 - Q3));
    const double B = Q / A;
    roots[0] = A + B - a3;
    return 1;
}

__device__ __forceinline__ double det3(const double* r0, const double* r1, const double* r2)
{
    return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1])
         - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0])
         + r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// 7-point solver: x1, x2 = 7 normalised correspondences.  Null space of the 7x9 system = last two
// columns of Q in the Householder QR of A^T (9x7); all indices static -> registers.
__device__ int seven_point(const double (&px1)[7][2], const double (&px2)[7][2], double* Fs /* 27 */)
{
    double M[9][7];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        const double ax = px1[p][0], ay = px1[p][1], bx = px2[p][0], by = px2[p][1];
        M[0][p] = bx * ax; M[1][p] = bx * ay; M[2][p] = bx;
        M[3][p] = by * ax; M[4][p] = by * ay; M[5][p] = by;
        M[6][p] = ax;      M[7][p] = ay;      M[8][p] = 1.0;
    }
    double beta[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        double nrm2 = 0.0;
#pragma unroll
        for (int r = j; r < 9; ++r) nrm2 += M[r][j] * M[r][j];
        const double nrm = sqrt(nrm2);
        double bj = 0.0;
        if (nrm != 0.0) {
            const double alpha = (M[j][j] > 0.0) ? -nrm : nrm;
            M[j][j] -= alpha;                       // column j now holds the reflector v_j (rows j..8)
            double vn2 = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) vn2 += M[r][j] * M[r][j];
            if (vn2 != 0.0) bj = 2.0 / vn2;
        } else {
#pragma unroll
            for (int r = j; r < 9; ++r) M[r][j] = 0.0;
        }
        beta[j] = bj;
#pragma unroll
        for (int c = j + 1; c < 7; ++c) {
            double dot = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) dot += M[r][j] * M[r][c];
            const double s = bj * dot;
#pragma unroll
            for (int r = j; r < 9; ++r) M[r][c] -= s * M[r][j];
        }
    }
    double f[2][9];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int r = 0; r < 9; ++r) f[e][r] = (r == 7 + e) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 6; j >= 0; --j) {
            double dot = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) dot += M[r][j] * f[e][r];
            const double s = beta[j] * dot;
#pragma unroll
            for (int r = j; r < 9; ++r) f[e][r] -= s * M[r][j];
        }
    }
    const double* A = f[0];
    const double* B = f[1];
    double P[4];
    P[0] = det3(A, A + 3, A + 6);
    P[1] = det3(B, A + 3, A + 6) + det3(A, B + 3, A + 6) + det3(A, A + 3, B + 6);
    P[2] = det3(A, B + 3, B + 6) + det3(B, A + 3, B + 6) + det3(B, B + 3, A + 6);
    P[3] = det3(B, B + 3, B + 6);
    double roots[3];
    const int n = solve_cubic(P, roots);
    for (int k = 0; k < n; ++k)
#pragma unroll
        for (int e = 0; e < 9; ++e) Fs[9 * k + e] = A[e] + roots[k] * B[e];
    return n;
}

// SymmetricEpipolarDistanceError (squared, /4)
__device__ __forceinline__ double sym_epipolar_err(const double* F, double x1, double y1, double x2, double y2)
{
    const double Fx0 = F[0] * x1 + F[1] * y1 + F[2];
    const double Fx1 = F[3] * x1 + F[4] * y1 + F[5];
    const double Fx2 = F[6] * x1 + F[7] * y1 + F[8];
    const double Fty0 = F[0] * x2 + F[3] * y2 + F[6];
    const double Fty1 = F[1] * x2 + F[4] * y2 + F[7];
    const double yFx = x2 * Fx0 + y2 * Fx1 + Fx2;
    return (yFx * yFx) * (1.0 / (Fx0 * Fx0 + Fx1 * Fx1) + 1.0 / (Fty0 * Fty0 + Fty1 * Fty1)) / 4.0;
}

// 4-point homography (OpenMVG homography::kernel::FourPointSolver, DLT): rows of L per correspondence
//   [x y 1 0 0 0 -x'x -x'y -x'] and [0 0 0 x y 1 -y'x -y'y -y'];  h = null vector of the 8x9 system,
// taken as the last column of Q in the Householder QR of L^T (9x8), all indices static.
__device__ int four_point_h(const double (&px1)[7][2], const double (&px2)[7][2], double* Hs)
{
    double M[9][8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double x = px1[p][0], y = px1[p][1], u = px2[p][0], v = px2[p][1];
        M[0][2 * p] = x;  M[1][2 * p] = y;  M[2][2 * p] = 1.0; M[3][2 * p] = 0.0; M[4][2 * p] = 0.0; M[5][2 * p] = 0.0;
        M[6][2 * p] = -u * x; M[7][2 * p] = -u * y; M[8][2 * p] = -u;
        M[0][2 * p + 1] = 0.0; M[1][2 * p + 1] = 0.0; M[2][2 * p + 1] = 0.0; M[3][2 * p + 1] = x; M[4][2 * p + 1] = y; M[5][2 * p + 1] = 1.0;
        M[6][2 * p + 1] = -v * x; M[7][2 * p + 1] = -v * y; M[8][2 * p + 1] = -v;
    }
    double beta[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double nrm2 = 0.0;
#pragma unroll
        for (int r = j; r < 9; ++r) nrm2 += M[r][j] * M[r][j];
        const double nrm = sqrt(nrm2);
        double bj = 0.0;
        if (nrm != 0.0) {
            const double alpha = (M[j][j] > 0.0) ? -nrm : nrm;
            M[j][j] -= alpha;
            double vn2 = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) vn2 += M[r][j] * M[r][j];
            if (vn2 != 0.0) bj = 2.0 / vn2;
        } else {
#pragma unroll
            for (int r = j; r < 9; ++r) M[r][j] = 0.0;
        }
        beta[j] = bj;
#pragma unroll
        for (int c = j + 1; c < 8; ++c) {
            double dot = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) dot += M[r][j] * M[r][c];
            const double s = bj * dot;
#pragma unroll
            for (int r = j; r < 9; ++r) M[r][c] -= s * M[r][j];
        }
    }
    double f[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) f[r] = (r == 8) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        double dot = 0.0;
#pragma unroll
        for (int r = j; r < 9; ++r) dot += M[r][j] * f[r];
        const double s = beta[j] * dot;
#pragma unroll
        for (int r = j; r < 9; ++r) f[r] -= s * M[r][j];
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) Hs[e] = f[e];
    return 1;
}

// homography AsymmetricError: || x2 - hnormalized(H x1) ||^2
__device__ __forceinline__ double h_asym_err(const double* H, double x1, double y1, double x2, double y2)
{
    const double w = H[6] * x1 + H[7] * y1 + H[8];
    const double ex = x2 - (H[0] * x1 + H[1] * y1 + H[2]) / w;
    const double ey = y2 - (H[3] * x1 + H[4] * y1 + H[5]) / w;
    return ex * ex + ey * ey;
}

// ------------------------------------------------------------------------------------------------
// 5-point essential matrix (OpenMVG essential::kernel::FivePointKernel behind GeometricFilter_EMatrix_AC,
// /root/reference/src/R3DComputeMatches.cpp:2169).  Same polynomial system as OpenMVG's FivePointsRelativePose,
// solved with Nister's hidden-variable elimination + a Sturm sequence; operation for operation the arithmetic of
// oracle/essential.c (only + - * / after the QR), so both produce the same bits.  Sixteen lanes per minimal sample
// (five_point_coop below), the sample's state in an LDS workspace.
// ------------------------------------------------------------------------------------------------
constexpr unsigned char kT11[4][4] = {{0, 2, 3, 4}, {2, 1, 5, 6}, {3, 5, 7, 8}, {4, 6, 8, 9}};
constexpr unsigned char kT21[10][4] = {{0, 2, 4, 5}, {3, 1, 6, 7}, {2, 3, 8, 9}, {4, 8, 10, 11}, {5, 9, 11, 12},
                                       {8, 6, 13, 14}, {9, 7, 14, 15}, {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}};

// products of small polynomials in (x, y, z); fully unrolled: the monomial tables are compile-time constants, so every
// index is static and the operands stay in registers
__device__ __forceinline__ void e_mul11(const double (&a)[4], const double (&b)[4], double (&out)[10])
{
#pragma unroll
    for (int k = 0; k < 10; ++k) out[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[kT11[i][j]] += a[i] * b[j];
}
__device__ __forceinline__ void e_mul21_acc(const double (&a)[10], const double (&b)[4], double (&out)[20])
{
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[kT21[i][j]] += a[i] * b[j];
}
template <int DA, int DB>
__device__ __forceinline__ void e_pmul(const double (&a)[DA + 1], const double (&b)[DB + 1], double (&out)[DA + DB + 1])
{
#pragma unroll
    for (int k = 0; k <= DA + DB; ++k) out[k] = 0.0;
#pragma unroll
    for (int i = 0; i <= DA; ++i)
#pragma unroll
        for (int j = 0; j <= DB; ++j) out[i + j] += a[i] * b[j];
}
template <int D>
__device__ __forceinline__ double e_peval(const double (&p)[D + 1], double t)
{
    double v = p[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) v = v * t + p[k];
    return v;
}

// ------------------------------------------------------------------------------------------------
// The 5-point solver as a COOPERATIVE routine: 16 lanes per minimal sample, 16 samples per workgroup pass.
// (Round 2 ran one sample per lane of wave 0: 512 registers + 256 spilled, two out-of-line calls from divergent flow that were
// only correct with SGPR spills forced to memory, 0.85 ms per chunk on one wave while 192 lanes waited, one workgroup per CU.)
// Here every quantity with an index -- matrix columns, constraint rows, polynomial coefficients, Sturm-chain evaluations, roots,
// models -- belongs to a lane of the sample's group, the sample's state lives in a 367-double LDS workspace, and the lanes of a
// group meet at wave-level fences only (a group never leaves its wavefront, so LDS program order is all the synchronisation there
// is; groups of one wave may diverge freely).  No register arrays with dynamic indices, no calls, nothing data-dependent crosses a
// workgroup barrier.  Every value is produced by exactly the operation sequence of the one-lane form (and of oracle/essential.c):
// the split is over INDEPENDENT outputs, sums keep their order.
// ------------------------------------------------------------------------------------------------
constexpr int kE5Lanes = 16;            // lanes per sample
constexpr int kE5Samples = 16;          // samples per workgroup pass = the essential-matrix kernel's chunk of hypotheses
constexpr int kE5Stride = 367;          // doubles per sample workspace (odd: the groups' accesses spread over the LDS banks)
// workspace map (doubles)
constexpr int kE5A = 0;                 // 10 x 20 elimination matrix A[r][k] at 20 r + k; later the Sturm chain f[k][c] at 11 k + c (k < 12)
constexpr int kE5Deg = 140;             //   ... degree of f[k] (inside the dead matrix)
constexpr int kE5Roots = 160;           //   ... real roots
constexpr int kE5Sgn = 172;             //   ... signs of the chain at a point (12)
constexpr int kE5N = 200;               // null space N[e][r] at 9 e + r (e < 4): E1[r][e] = N[e][r]
constexpr int kE5M = 236;               // 9 x 5 design matrix M[r][j] at 5 r + j, Householder vectors in place      (dead after N)
constexpr int kE5Beta = 281;            // beta[5]                                                                    (dead after N)
constexpr int kE5Eii = 236;             // EE^T diagonal polynomials eii[i][10] at 10 i (i < 3)                         (over M)
constexpr int kE5Tr = 266;              // trace polynomial [10]
constexpr int kE5L = 276;               // L[i][j][10] at 10 (3 i + j)  (90 doubles)                                   (dead after the rows)
constexpr int kE5B = 236;               // B[q][v][5] at 5 (3 q + v)   (45 doubles)                                    (over eii / tr / L)
constexpr int kE5Term = 290;            // the three terms of det B, [3][11]
constexpr int kE5P = 330;               // det B, degree 10 [11]
constexpr int kE5Flag = 345;            // model validity flags [10]
static_assert(kE5L + 90 <= kE5Stride && kE5Flag + 10 <= kE5Stride, "5-point workspace map");

// lanes of one sample meet here: compiler fence + scheduling barrier (LDS operations of one wavefront execute in program order)
#define E5_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ void e5_row(const double* __restrict__ W, int r, double (&a)[4])     // E1[r][0..3]
{
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = W[kE5N + 9 * e + r];
}
__device__ __forceinline__ void e5_mul11_rows(const double* __restrict__ W, int ra, int rb, double (&out)[10])
{
    double a[4], b[4];
    e5_row(W, ra, a); e5_row(W, rb, b);
    e_mul11(a, b, out);
}

// l = lane within the sample's group (0..15); W = the sample's workspace; Es = its model slots (9 doubles per model).
// Returns the number of models (the same value in every lane of the group).
__device__ __forceinline__ int five_point_coop(const double (&px1)[7][2], const double (&px2)[7][2], double* __restrict__ Es,
                                               double* __restrict__ W, const int l)
{
    // ---- design matrix: lane p < 5 writes the column of point p
    if (l < 5) {
        double ax = 0, ay = 0, bx = 0, by = 0;
#pragma unroll
        for (int p = 0; p < 5; ++p) if (p == l) { ax = px1[p][0]; ay = px1[p][1]; bx = px2[p][0]; by = px2[p][1]; }
        double* M = W + kE5M + l;
        M[0] = bx * ax; M[5] = bx * ay; M[10] = bx;
        M[15] = by * ax; M[20] = by * ay; M[25] = by;
        M[30] = ax;      M[35] = ay;      M[40] = 1.0;
    }
    E5_SYNC();
    // ---- Householder QR of the 9 x 5 matrix: every lane derives the reflection of column j from the column itself (identical
    // values everywhere), lane 0 stores the vector and beta, lanes c = j + 1 .. 4 apply it to their column
    for (int j = 0; j < 5; ++j) {
        double v[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) v[r] = W[kE5M + 5 * r + j];
        double nrm2 = 0.0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (r >= j) nrm2 += v[r] * v[r];
        const double nrm = sqrt(nrm2);
        double bj = 0.0;
        if (nrm != 0.0) {
            double vjj = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r == j) vjj = v[r];
            const double alpha = (vjj > 0.0) ? -nrm : nrm;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r == j) v[r] = vjj - alpha;
            double vn2 = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r >= j) vn2 += v[r] * v[r];
            if (vn2 != 0.0) bj = 2.0 / vn2;
        } else {
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r >= j) v[r] = 0.0;
        }
        E5_SYNC();                                                  // every lane has read column j
        if (l == 0) {
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r >= j) W[kE5M + 5 * r + j] = v[r];
            W[kE5Beta + j] = bj;
        }
        if (l > j && l < 5) {
            double col[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) col[r] = W[kE5M + 5 * r + l];
            double dot = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r >= j) dot += v[r] * col[r];
            const double sc = bj * dot;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (r >= j) W[kE5M + 5 * r + l] = col[r] - sc * v[r];
        }
        E5_SYNC();
    }
    // ---- null space: lane e < 4 reflects the unit vector e_{5 + e} back through the five reflections
    if (l < 4) {
        double n[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) n[r] = (r == 5 + l) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 4; j >= 0; --j) {
            double dot = 0.0;
#pragma unroll
            for (int r = j; r < 9; ++r) dot += W[kE5M + 5 * r + j] * n[r];
            const double sc = W[kE5Beta + j] * dot;
#pragma unroll
            for (int r = j; r < 9; ++r) n[r] -= sc * W[kE5M + 5 * r + j];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) W[kE5N + 9 * l + r] = n[r];
    }
    E5_SYNC();
    // ---- EE^T diagonal polynomials (lanes 0..2) and the determinant constraint, row 0 of A (lane 9)
    if (l < 3) {
        double a0[10], a1[10], a2[10];
        e5_mul11_rows(W, 3 * l, 3 * l, a0); e5_mul11_rows(W, 3 * l + 1, 3 * l + 1, a1); e5_mul11_rows(W, 3 * l + 2, 3 * l + 2, a2);
#pragma unroll
        for (int k = 0; k < 10; ++k) W[kE5Eii + 10 * l + k] = a0[k] + a1[k] + a2[k];
    }
    if (l == 9) {
        double row[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) row[k] = 0.0;
        double t1[10], t2[10], d2[10], e4[4];
        e5_mul11_rows(W, 1, 5, t1); e5_mul11_rows(W, 2, 4, t2);
#pragma unroll
        for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
        e5_row(W, 6, e4); e_mul21_acc(d2, e4, row);
        e5_mul11_rows(W, 2, 3, t1); e5_mul11_rows(W, 0, 5, t2);
#pragma unroll
        for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
        e5_row(W, 7, e4); e_mul21_acc(d2, e4, row);
        e5_mul11_rows(W, 0, 4, t1); e5_mul11_rows(W, 1, 3, t2);
#pragma unroll
        for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
        e5_row(W, 8, e4); e_mul21_acc(d2, e4, row);
#pragma unroll
        for (int k = 0; k < 20; ++k) W[kE5A + k] = row[k];
    }
    E5_SYNC();
    if (l < 10) W[kE5Tr + l] = 0.5 * ((W[kE5Eii + l] + W[kE5Eii + 10 + l]) + W[kE5Eii + 20 + l]);     // 0.5 * ((EET00 + EET11) + EET22)
    E5_SYNC();
    // ---- L[i][j] = (E E^T)_ij - delta_ij tr: lane 3 i + j
    if (l < 9) {
        const int i = l / 3, j = l % 3;
        double a0[10], a1[10], a2[10];
        e5_mul11_rows(W, 3 * i, 3 * j, a0); e5_mul11_rows(W, 3 * i + 1, 3 * j + 1, a1); e5_mul11_rows(W, 3 * i + 2, 3 * j + 2, a2);
#pragma unroll
        for (int k = 0; k < 10; ++k) { double v = a0[k] + a1[k] + a2[k]; if (i == j) v -= W[kE5Tr + k]; W[kE5L + 10 * l + k] = v; }
    }
    E5_SYNC();
    // ---- the nine trace constraints: row 1 + 3 i + j of A = sum over j' of L[i][j'] * E1[3 j' + j], lane 3 i + j
    if (l < 9) {
        const int i = l / 3, j = l % 3;
        double row[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) row[k] = 0.0;
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
            double Lp[10], e4[4];
#pragma unroll
            for (int k = 0; k < 10; ++k) Lp[k] = W[kE5L + 10 * (3 * i + jp) + k];
            e5_row(W, 3 * jp + j, e4);
            e_mul21_acc(Lp, e4, row);
        }
#pragma unroll
        for (int k = 0; k < 20; ++k) W[kE5A + 20 * (1 + l) + k] = row[k];
    }
    E5_SYNC();
    // ---- Gauss-Jordan on the first 10 columns, partial pivoting.  Lane l owns column l (and column 16 + l for l < 4).  Per
    // pivot column c: every lane reads column c (the pivot search and all ten multipliers come from it), then updates its own
    // columns: swap rows c / piv, scale the pivot row, eliminate -- the one-lane order of operations per entry.
    bool alive = true;
    for (int c = 0; c < 10; ++c) {
        double colc[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) colc[r] = W[kE5A + 20 * r + c];
        int piv = c;
        double best = 0.0, pval = 0.0, cval = 0.0;
#pragma unroll
        for (int r = 0; r < 10; ++r) if (r == c) { best = fabs(colc[r]); pval = colc[r]; cval = colc[r]; }
#pragma unroll
        for (int r = 0; r < 10; ++r) if (r > c) { const double v = fabs(colc[r]); if (v > best) { best = v; piv = r; pval = colc[r]; } }
        if (best == 0.0) alive = false;                             // (the same decision in every lane of the group)
        E5_SYNC();                                                  // every lane has read column c
        if (alive) {
            const double inv = 1.0 / pval;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = l + 16 * h;
                if (k < 20 && k >= c) {
                    const double old_c = W[kE5A + 20 * c + k], old_p = W[kE5A + 20 * piv + k];
                    const double prow = ((piv != c) ? old_p : old_c) * inv;       // pivot row after the swap, scaled
                    W[kE5A + 20 * c + k] = prow;
                    if (piv != c) W[kE5A + 20 * piv + k] = old_c;
#pragma unroll
                    for (int r = 0; r < 10; ++r) {
                        if (r == c) continue;
                        // row r after the swap: the old row c sits in row piv; its multiplier is its entry in column c
                        const bool moved = (piv != c) && (r == piv);
                        const double fct = moved ? cval : colc[r];
                        if (fct == 0.0) continue;
                        const double cur = moved ? old_c : W[kE5A + 20 * r + k];
                        W[kE5A + 20 * r + k] = cur - fct * prow;
                    }
                }
            }
        }
        E5_SYNC();
    }
    if (!alive) return 0;
    // ---- B(z): lane 3 q + v
    if (l < 9) {
        const int q = l / 3, v = l % 3;
        const int lo = kE5A + 20 * (4 + 2 * q), hi = kE5A + 20 * (5 + 2 * q);
        double b[5];
        if (v < 2) {
            const int o = 10 + 3 * v;
            b[0] = W[lo + o + 2];
            b[1] = W[lo + o + 1] - W[hi + o + 2];
            b[2] = W[lo + o] - W[hi + o + 1];
            b[3] = -W[hi + o];
            b[4] = 0.0;
        } else {
            b[0] = W[lo + 19];
            b[1] = W[lo + 18] - W[hi + 19];
            b[2] = W[lo + 17] - W[hi + 18];
            b[3] = W[lo + 16] - W[hi + 17];
            b[4] = -W[hi + 16];
        }
        E5_SYNC();                                                  // (B overlays nothing that is still read: eii / tr / L are dead)
#pragma unroll
        for (int e = 0; e < 5; ++e) W[kE5B + 5 * l + e] = b[e];
    } else {
        E5_SYNC();
    }
    E5_SYNC();
    // ---- det B(z) = sum of three products: lane q < 3 one term
    if (l < 3) {
        const int q = l, r1 = (q + 1) % 3, r2 = (q + 2) % 3;
        double a30[4], a31[4], b30[4], b31[4], c4[5];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a30[k] = W[kE5B + 5 * (3 * r1 + 0) + k]; a31[k] = W[kE5B + 5 * (3 * r1 + 1) + k];
            b30[k] = W[kE5B + 5 * (3 * r2 + 0) + k]; b31[k] = W[kE5B + 5 * (3 * r2 + 1) + k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) c4[k] = W[kE5B + 5 * (3 * q + 2) + k];
        double m1[7], m2[7], mn[7], term[11];
        e_pmul<3, 3>(a30, b31, m1);
        e_pmul<3, 3>(a31, b30, m2);
#pragma unroll
        for (int k = 0; k <= 6; ++k) mn[k] = m1[k] - m2[k];
        e_pmul<6, 4>(mn, c4, term);
#pragma unroll
        for (int k = 0; k <= 10; ++k) W[kE5Term + 11 * q + k] = term[k];
    }
    E5_SYNC();
    if (l < 11) { double pk = 0.0; pk += W[kE5Term + l]; pk += W[kE5Term + 11 + l]; pk += W[kE5Term + 22 + l]; W[kE5P + l] = pk; }
    E5_SYNC();
    // ---- real roots of det B (degree <= 10): Sturm chain in the dead matrix area, built coefficient-parallel
    int d = 10;
    while (d > 0 && W[kE5P + d] == 0.0) --d;
    if (d <= 0) return 0;
    // (the chain's 11 x 11 slots start as zeros: every coefficient above a polynomial's degree then IS zero -- copies, eliminations and
    //  normalisations below only write up to the degree and zero what they eliminate -- so the bisection can run Horner over fixed
    //  lengths: leading zeros change no bit of the value)
    for (int e = l; e < 121; e += kE5Lanes) W[kE5A + e] = 0.0;
    E5_SYNC();
    {
        const double lead = W[kE5P + d];
        if (l <= d) W[kE5A + l] = W[kE5P + l] / lead;
        if (l == 15) W[kE5Deg + 0] = (double)d;
    }
    E5_SYNC();
    if (l >= 1 && l <= d) W[kE5A + 11 + l - 1] = (double)l * W[kE5A + l];
    if (l == 15) W[kE5Deg + 1] = (double)(d - 1);
    E5_SYNC();
    int nf = 2;
    {
        int deg_prev = d, deg_cur = d - 1;
        while (deg_cur > 0) {
            // f[nf] = -rem(f[nf-2], f[nf-1]) / |leading coefficient|, remainder built in place in slot nf
            const int bo = kE5A + 11 * (nf - 1), ro = kE5A + 11 * nf;
            const int db = deg_cur;
            int dr = deg_prev;
            if (l <= dr) W[ro + l] = W[kE5A + 11 * (nf - 2) + l];
            E5_SYNC();
            const double blead = W[bo + db];
            while (dr >= db) {
                const double q = W[ro + dr] / blead;
                E5_SYNC();                                          // every lane has read the leading remainder coefficient
                if (l < db) W[ro + dr - db + l] -= q * W[bo + l];
                if (l == 15) W[ro + dr] = 0.0;
                E5_SYNC();
                --dr;
            }
            while (dr >= 0 && W[ro + dr] == 0.0) --dr;
            if (dr < 0) break;
            const double sc = fabs(W[ro + dr]);
            E5_SYNC();
            if (l <= dr) W[ro + l] = -W[ro + l] / sc;
            if (l == 15) W[kE5Deg + nf] = (double)dr;
            E5_SYNC();
            deg_prev = deg_cur; deg_cur = dr;
            ++nf;
        }
    }
    double bound = 0.0;
    for (int k = 0; k < d; ++k) { const double a = fabs(W[kE5A + k]); if (a > bound) bound = a; }
    bound += 1.0;
    // sign changes of the chain at -bound and +bound: lane k evaluates f[k], the count walks the signs in order
    int va = 0, vb = 0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const double t = side == 0 ? -bound : bound;
        if (l < nf) {
            const int dk = (int)W[kE5Deg + l];
            double v = W[kE5A + 11 * l + dk];
            for (int c = dk - 1; c >= 0; --c) v = v * t + W[kE5A + 11 * l + c];
            W[kE5Sgn + l] = (double)((v > 0.0) - (v < 0.0));
        }
        E5_SYNC();
        int changes = 0, last = 0;
        for (int k = 0; k < nf; ++k) {
            const int sg = (int)W[kE5Sgn + k];
            if (sg != 0) { if (last != 0 && sg != last) ++changes; last = sg; }
        }
        if (side == 0) va = changes; else vb = changes;
        E5_SYNC();
    }
    int nr = va - vb;
    if (nr > 10) nr = 10;
    if (nr <= 0) return 0;
    // ---- bisection: lane r < nr owns root r (the (r + 1)-th real root from below): its own midpoints and sign counts, 64 steps
    if (l < nr) {
        double lo = -bound, hi = bound;
        for (int it = 0; it < 64; ++it) {
            const double mid = 0.5 * (lo + hi);
            int changes = 0, last = 0;
            // f[k] has degree <= 10 - k (the chain's degrees fall by at least one per step) and zeros above its own: eleven Horner
            // chains of FIXED length, independent of one another, their 66 coefficients read from LDS at constant offsets -- the
            // reads pipeline and the chains interleave.  (With the degrees read from LDS the 66 reads and multiply-adds of a step were
            // one dependent sequence: 64 steps x ~66 LDS round trips per root, most of a chunk's 170 us.)  A polynomial beyond the
            // chain's end is all zeros: its sign 0 is skipped like any vanishing value.
            asm volatile("" ::: "memory");              // (the coefficients stay in LDS: 132 registers of them would spill)
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                double v = W[kE5A + 11 * k + (10 - k)];
#pragma unroll
                for (int c = 9 - k; c >= 0; --c) v = v * mid + W[kE5A + 11 * k + c];
                const int sg = (v > 0.0) - (v < 0.0);
                if (sg != 0) { if (last != 0 && sg != last) ++changes; last = sg; }
            }
            if (va - changes >= l + 1) hi = mid; else lo = mid;
        }
        W[kE5Roots + l] = 0.5 * (lo + hi);
    }
    E5_SYNC();
    // ---- one model per root: lane s < nr; models with a vanishing minor are skipped, the others keep their order
    double model[9];
    bool valid = false;
    if (l < nr) {
        const double z = W[kE5Roots + l];
        double b[3][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double p0[4], p1[4], p2[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) { p0[k] = W[kE5B + 5 * (3 * q) + k]; p1[k] = W[kE5B + 5 * (3 * q + 1) + k]; }
#pragma unroll
            for (int k = 0; k < 5; ++k) p2[k] = W[kE5B + 5 * (3 * q + 2) + k];
            b[q][0] = e_peval<3>(p0, z); b[q][1] = e_peval<3>(p1, z); b[q][2] = e_peval<4>(p2, z);
        }
        double bx = 0.0, by = 0.0, bw = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r1 = q, r2 = (q + 1) % 3;
            const double cx = b[r1][1] * b[r2][2] - b[r1][2] * b[r2][1];
            const double cy = b[r1][2] * b[r2][0] - b[r1][0] * b[r2][2];
            const double cw = b[r1][0] * b[r2][1] - b[r1][1] * b[r2][0];
            if (fabs(cw) > fabs(bw)) { bx = cx; by = cy; bw = cw; }
        }
        if (bw != 0.0) {
            valid = true;
            const double x = bx / bw, y = by / bw;
#pragma unroll
            for (int r = 0; r < 9; ++r) model[r] = x * W[kE5N + r] + y * W[kE5N + 9 + r] + z * W[kE5N + 18 + r] + W[kE5N + 27 + r];
        }
    }
    if (l < 10) W[kE5Flag + l] = (l < nr && valid) ? 1.0 : 0.0;
    E5_SYNC();
    int before = 0, n_out = 0;
    for (int s2 = 0; s2 < nr; ++s2) { const int fl = (int)W[kE5Flag + s2]; if (s2 < l) before += fl; n_out += fl; }
    if (valid) {
#pragma unroll
        for (int r = 0; r < 9; ++r) Es[9 * before + r] = model[r];
    }
    return n_out;
}
#undef E5_SYNC

// fundamental::kernel::EpipolarDistanceError: squared distance of x2 to the epipolar line F x1
__device__ __forceinline__ double epipolar_dist_err(const double* F, double x1, double y1, double x2, double y2)
{
    const double l0 = F[0] * x1 + F[1] * y1 + F[2];
    const double l1 = F[3] * x1 + F[4] * y1 + F[5];
    const double l2 = F[6] * x1 + F[7] * y1 + F[8];
    const double d = l0 * x2 + l1 * y2 + l2;
    return (d * d) / (l0 * l0 + l1 * l1);
}

// FundamentalFromEssential: F = K2^-T E K1^-1
__device__ __forceinline__ void f_from_e(const double* E, const double* K1i, const double* K2i, double* F)
{
    double T[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += K2i[3 * k + r] * E[3 * k + c];
            T[3 * r + c] = v;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += T[3 * r + k] * K1i[3 * k + c];
            F[3 * r + c] = v;
        }
}

// ---- per-workgroup shared state (lives at the front of the dynamic LDS region) ----
struct FState {
    double minNFA, errorMax;
    double bestF[9];
    double cur_nfa;          // scratch: NFA of the model under evaluation
    uint32_t nIter, reserve, iter, pool_size, n_inl, acMode, n_models, iters_done;
    uint32_t cnt, cur_k, flag, chunk_n;
    uint32_t wave_cnt[8];    // (up to 8 waves: the wide variant of the kernel runs 512 threads)
    double   red_v[8];
    double   red_b[8];       // reduction slots of the sort-skipping bound
    uint32_t red_k[8];
    uint32_t nm[64];
    uint32_t dbg_smp[64];    // debug trace: first sample index of each hypothesis of the chunk
    uint32_t dbg_pool;       // debug trace: pool size the chunk was drawn from
    double   kinv[18];       // essential matrix: K1^-1, K2^-1 of the pair (read from here at their two use sites: 36 wave-uniform
                             // doubles held in scalar registers for the whole kernel were most of its SGPR spills)
};
static_assert(sizeof(FState) <= 1024, "FState must fit the first 1024 bytes of the dynamic LDS region");

constexpr int kChunk = 64;                          // hypotheses per chunk, F and H: one lane of wave 0 per minimal sample
template <int KIND> struct ChunkOf { static constexpr int n = (KIND == 2) ? kE5Samples : kChunk; };   // E: 16 lanes per sample, 16 samples
// residual histogram of the model under evaluation (the sort-skipping bound below): 2^kHistSub bins per octave of the residual,
// kHistBins bins down from the bound on the residuals; LDS header = FState (padded to 1024) + the histogram
constexpr int kHistBins = 1024, kHistSub = 5, kHistShift = 52 - kHistSub;
constexpr int kHdr = 1024 + kHistBins * 4;
// the scout pass (round 6, acransac_body): per chunk the flat offsets of the hypotheses' models, and per scouted model the number of
// matches within the bound and the lower bound of its NFA; the per-wave histograms of the pass live in the (then idle) sort buffers
constexpr int kScoutModels = 192;                   // 64 x 3 (F), 64 x 1 (H), 16 x 10 (E)
constexpr int kScoutBytes = kScoutModels * 4 + kScoutModels * 8 + 80 * 4;      // totals, bounds, offsets [65] padded: 2624 bytes (16-byte multiple)
constexpr int kScoutHistBytes = 8 * kHistBins * 4;  // eight waves (the 512-thread variant) x 1024 u32 bins

#if defined(R3DM_FILTER_ONLY_E) || defined(R3DM_FILTER_DEVICE_ONLY)
size_t filter_F_lds_bytes(uint32_t m_cap, int model_kind);
static inline size_t filter_F_lds_bytes_unused_(uint32_t m_cap, int model_kind)
#else
size_t filter_F_lds_bytes(uint32_t m_cap, int model_kind)
#endif
{
    // [FState, padded to 1024][histogram: 1024 x u32][Fs: chunk x (9 x MAX_MODELS) doubles][scout: totals, bounds, offsets][keys: m_cap x u64][idx: m_cap x u32]
    const size_t ms = model_kind == 2 ? 90 : 27;
    const size_t chunk = model_kind == 2 ? kE5Samples : kChunk;
    size_t sort_bytes = (size_t)m_cap * 12;
    if (model_kind == 2 && sort_bytes < (size_t)kE5Samples * kE5Stride * 8) sort_bytes = (size_t)kE5Samples * kE5Stride * 8;   // 5-point workspaces
    if (sort_bytes < (size_t)kScoutHistBytes) sort_bytes = (size_t)kScoutHistBytes;                                                // the scout pass's per-wave histograms
    return (size_t)kHdr + chunk * ms * 8 + kScoutBytes + sort_bytes;
}

// KIND 0: fundamental matrix (7-point, <= 3 models, symmetric epipolar error, point-to-line NFA scale)
// KIND 1: homography (4-point DLT, 1 model, asymmetric transfer error, point-to-point NFA scale)
// KIND 2: essential matrix (5-point on K^-1 x, <= 10 models, epipolar distance in pixels through F = K2^-T E K1^-1,
//         no normalisation of the points: ACKernelAdaptorEssential)
// KeyT / IdxT: the (residual, index) sort buffers live in LDS, or -- for the rare pair with more putative matches than
// the LDS budget holds (near-duplicate views with > 8192 matches) -- in a slice of global scratch.
// debug aids of the developer build (-DR3DM_DEVTOOLS; r3dm_internal.hpp): R3DM_FILTER_CHECK=1 records the first violated
// invariant instead of running into a memory fault, R3DM_TRACE_PAIR dumps the per-model trace of one pair.  In the product
// build both pointers are compile-time null and the code below them disappears.
#if defined(R3DM_DEVTOOLS) || defined(R3DM_BISECT_DBG)
#define R3DM_DBG(P) ((P).dbg)
#else
#define R3DM_DBG(P) ((uint32_t*)nullptr)
#endif
#if defined(R3DM_DEVTOOLS) || defined(R3DM_BISECT_TRACE)
#define R3DM_TRACE(P) ((P).trace)
#else
#define R3DM_TRACE(P) ((double*)nullptr)
#endif
// bisect builds (tools/build_bisect.sh): each of the four trace sites can be kept as a runtime-null test on its own
#ifdef R3DM_BISECT_T1
#define R3DM_TRACE1(P) ((P).trace)
#else
#define R3DM_TRACE1(P) R3DM_TRACE(P)
#endif
#ifdef R3DM_BISECT_T2
#define R3DM_TRACE2(P) ((P).trace)
#else
#define R3DM_TRACE2(P) R3DM_TRACE(P)
#endif
#ifdef R3DM_BISECT_T3
#define R3DM_TRACE3(P) ((P).trace)
#else
#define R3DM_TRACE3(P) R3DM_TRACE(P)
#endif
#ifdef R3DM_BISECT_T4
#define R3DM_TRACE4(P) ((P).trace)
#else
#define R3DM_TRACE4(P) R3DM_TRACE(P)
#endif
#define FCHECK(cond, code, a, b)                                                                          \
    do {                                                                                                  \
        if (R3DM_DBG(P) && !(cond)) {                                                                           \
            if (atomicCAS(P.dbg, 0u, (uint32_t)(code)) == 0u) { P.dbg[1] = item; P.dbg[2] = (uint32_t)(a); P.dbg[3] = (uint32_t)(b); } \
        }                                                                                                 \
    } while (0)

// Workgroup barriers.  Waves of the workgroup exchange data through LDS almost everywhere (r3dm_syncthreads: barrier with an
// explicit lgkmcnt(0), see r3dm_internal.hpp -- the missing wait at the top of the chunk loop below is what made the
// homography kernel, the only one light enough for two workgroups per CU, return different results from run to run and
// eventually fault).  Through GLOBAL memory they exchange data in three places: the normalised points / pool / log-combinatorial
// table written at start-up, the pool rebuilt after an improvement, and -- spill variant only -- the sort buffers.  Every such
// slice belongs to ONE workgroup, whose waves share a CU and its vector L1 -- a WORKGROUP-scope fence plus a wait for the wave's
// outstanding stores and loads is what the exchange needs.  __threadfence() is agent scope: on this chip of eight L2s that is an L2
// write-back + invalidate per barrier, ~10 per evaluated model on a spilled pair, and it slows every other workgroup of the device
// with it.  (The explicit wait is not redundant: hipcc was seen to emit no vmcnt wait for a workgroup-scope fence on gfx950.)
// Two conditions the shortcut rests on, both enforced here rather than assumed: the s_waitcnt immediate below is the gfx9 encoding
// (vmcnt(0) lgkmcnt(0) = 0x0070), and the workgroup's waves must share one L1, i.e. the kernel is NOT compiled in threadgroup-split
// mode (-mtgsplit: the compiler defines no macro for it, so build.sh refuses the flag instead).
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(__gfx942__)
#error "wg_fence: the s_waitcnt encoding and the same-CU L1 argument are written for gfx942 / gfx950"
#endif
#endif
__device__ __forceinline__ void wg_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): this wave's global stores and loads have completed
}
__device__ __forceinline__ void wg_sync_global() { wg_fence(); r3dm_syncthreads(); wg_fence(); }
template <bool GLOBAL_BUFFERS>
__device__ __forceinline__ void wg_sync_t()
{
    if (GLOBAL_BUFFERS) { wg_fence(); r3dm_syncthreads(); wg_fence(); }
    else r3dm_syncthreads();
}

// Sort of the (residual, index) pairs of one model, ascending, for the workgroup's 256 threads: `total` compacted entries in
// keys / sidx (LDS), cap = next power of two >= total, E = max(1, cap / 256) entries per thread in registers.  Entry i lives in
// thread i / E, slot i % E; the bitonic network's exchanges at distance < E are register compare-exchanges, at distance < 64 E
// lane exchanges inside the wave (ds_bpermute), and only the two largest distances (three of the log2(cap)(log2(cap)+1)/2
// stages) cross waves through LDS with a barrier.  The pairs are distinct (index breaks ties), so the result is the same total
// order the plain LDS network produces -- 4.5x fewer cycles per model on C2's pairs.
__device__ __forceinline__ bool pair_gt(unsigned long long a, uint32_t ai, unsigned long long b, uint32_t bi)
{
    return (a > b) || (a == b && ai > bi);
}

// GLOBAL: keys / sidx are the global spill buffers of a pair with more matches than the LDS holds (the register part is the same,
// the few wave-crossing stages go through global memory with a fence at the barrier).
template <int E, bool GLOBAL = false>
__device__ __forceinline__ void wg_sort_regs(unsigned long long* __restrict__ keys, uint32_t* __restrict__ sidx,
                                             uint32_t cap, uint32_t total, uint32_t tid)
{
    unsigned long long k[E];
    uint32_t x[E];
    const uint32_t base = tid * (uint32_t)E;
#pragma unroll
    for (int s = 0; s < E; ++s) {
        const uint32_t i = base + (uint32_t)s;
        const bool live = i < total;
        k[s] = live ? keys[i] : ~0ull;
        x[s] = live ? sidx[i] : 0xFFFFFFFFu;
    }
    for (uint32_t size = 2; size <= cap; size <<= 1) {
        // ---- distances that cross waves: through LDS (every thread parks its entries, reads the partner's)
        for (uint32_t stride = size >> 1; stride >= 64u * E; stride >>= 1) {
            wg_sync_t<GLOBAL>();                                   // earlier readers of keys / sidx are done
            if (base < cap) {
#pragma unroll
                for (int s = 0; s < E; ++s) { keys[base + s] = k[s]; sidx[base + s] = x[s]; }
            }
            wg_sync_t<GLOBAL>();
            if (base < cap) {
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    const uint32_t i = base + (uint32_t)s;
                    const unsigned long long ok = keys[i ^ stride];
                    const uint32_t ox = sidx[i ^ stride];
                    const bool lower = (i & stride) == 0u, up = (i & size) == 0u;
                    const bool mine_gt = pair_gt(k[s], x[s], ok, ox);
                    if (mine_gt == (lower == up)) { k[s] = ok; x[s] = ox; }
                }
            }
        }
        // ---- distances inside the wave
        {
            uint32_t stride = size >> 1;
            if (stride >= 64u * E) stride = 32u * E;
            for (; stride >= (uint32_t)E; stride >>= 1) {
                const int lane_xor = (int)(stride / (uint32_t)E);
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    const uint32_t i = base + (uint32_t)s;
                    const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)k[s], lane_xor);
                    const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(k[s] >> 32), lane_xor);
                    const uint32_t ox = (uint32_t)__shfl_xor((int)x[s], lane_xor);
                    const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;
                    const bool lower = (i & stride) == 0u, up = (i & size) == 0u;
                    const bool mine_gt = pair_gt(k[s], x[s], ok, ox);
                    if (mine_gt == (lower == up)) { k[s] = ok; x[s] = ox; }
                }
            }
        }
        // ---- distances inside the thread
#pragma unroll
        for (int ST = E / 2; ST >= 1; ST >>= 1) {
            if ((uint32_t)ST < size) {
#pragma unroll
                for (int s = 0; s < E; ++s) {
                    if ((s & ST) == 0) {
                        const bool up = ((base + (uint32_t)s) & size) == 0u;
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
    wg_sync_t<GLOBAL>();
    if (base < cap) {
#pragma unroll
        for (int s = 0; s < E; ++s) { keys[base + s] = k[s]; sidx[base + s] = x[s]; }
    }
    wg_sync_t<GLOBAL>();
}

#ifndef R3DM_FILTER_DEVICE_ONLY      // (kernels_filter_coop.hip includes this file for the device routines above only)
// ------------------------------------------------------------------------------------------------
// The scout pass (round 6).  AC-RANSAC walks its models in order, but all a model needs from the matches to be passed over is how many of
// them lie within the bound and -- once some model has been accepted -- a lower bound of the NFA it could reach (the histogram bound
// of the full evaluation below).  Neither depends on the state of the walk, so the models of a chunk are scouted ahead of it, ONE
// WAVEFRONT PER MODEL (a workgroup scouts 4 or 8 models at a time): one sweep over the pair's matches, count by ballot, histogram in the
// wave's own 4 KiB of LDS, prefix and bound inside the wave -- no workgroup barrier, no compaction, no global atomics.  The walk then
// passes over a model without a barrier when the scout says it cannot be accepted (98 % of them), and runs the full
// evaluation -- unchanged -- on the others.  What was ~6 barriers + one pass per model becomes one barrier per sub-batch of models;
// inlier sets, models, iteration and model counts are those of the sequential rule (and of oracle/acransac.c): a skipped model is
// exactly one the full evaluation would have found hopeless or short of AC mode.
// la_g: per-bin NFA slope logalpha0 + mult log10(edge + eps) of the pair (filled once per pair, same expression as the full evaluation)
// ------------------------------------------------------------------------------------------------
// The scout only needs to know on which side of the bound a residual lies and in which histogram bin -- never the value -- so it
// trades the sweep's IEEE divisions (two per residual for F and H, ~11 dependent f64 instructions each) for v_rcp_f64 + one Newton step
// and carries an interval [lo, hi] that contains the residual the full evaluation would compute: v_rcp_f64 is good to 2^-23 (measured:
// tools/ubench/rcp_f64_accuracy.hip), one step squares that; the margins below are 2^-32 where the quotient enters as a factor and
// 2^-40 of the quotient where a difference follows it (H).  `hi <= bound` is surely within, `lo <= bound` possibly; the histogram takes
// `lo` of the possible ones, which can only lower the NFA bound.  Counts that the interval leaves open on either side of a threshold
// of the walk (SS, 2.5 SS) send the model to the full evaluation.  NaN / inf (a degenerate model) fail every comparison, as they do there.
__device__ __forceinline__ double rcp_newton(double a)
{
    const double r = __builtin_amdgcn_rcp(a);
    return __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
}
template <int KIND>
__device__ __forceinline__ void scout_residual(const double* F, double x1, double y1, double x2, double y2, double& lo, double& hi)
{
    if (KIND == 1) {
        const double rw = rcp_newton(F[6] * x1 + F[7] * y1 + F[8]);
        const double qx = (F[0] * x1 + F[1] * y1 + F[2]) * rw, qy = (F[3] * x1 + F[4] * y1 + F[5]) * rw;
        const double ex = __builtin_fabs(x2 - qx), ey = __builtin_fabs(y2 - qy);
        const double mx = 0x1p-40 * (__builtin_fabs(qx) + ex), my = 0x1p-40 * (__builtin_fabs(qy) + ey);
        const double lx = __builtin_fmax(ex - mx, 0.0), ly = __builtin_fmax(ey - my, 0.0), hx = ex + mx, hy = ey + my;
        lo = (lx * lx + ly * ly) * (1.0 - 0x1p-40);
        hi = (hx * hx + hy * hy) * (1.0 + 0x1p-40);
    } else {
        const double l0 = F[0] * x1 + F[1] * y1 + F[2];
        const double l1 = F[3] * x1 + F[4] * y1 + F[5];
        const double l2 = F[6] * x1 + F[7] * y1 + F[8];
        const double d = x2 * l0 + y2 * l1 + l2;
        double r;
        if (KIND == 0) {
            const double t0 = F[0] * x2 + F[3] * y2 + F[6];
            const double t1 = F[1] * x2 + F[4] * y2 + F[7];
            r = (d * d) * (rcp_newton(l0 * l0 + l1 * l1) + rcp_newton(t0 * t0 + t1 * t1)) * 0.25;
        } else {
            r = (d * d) * rcp_newton(l0 * l0 + l1 * l1);
        }
        lo = r * (1.0 - 0x1p-32);
        hi = r * (1.0 + 0x1p-32);
    }
}
constexpr uint32_t kScoutOpen = 0x80000000u;        // sc_total: the count is open on one side of a threshold of the walk
// lanes of a wavefront that hand values to each other through LDS: to the compiler they are separate threads, so the store of one and the
// load of another need an order it has to respect (found by the scout's check mode: the cross-lane reads of the running counts below had
// been scheduled ahead of the write-back and saw raw bin counts -- a bound that was not one)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int KIND>
__device__ __noinline__ void scout_model(const float4* __restrict__ pt, uint32_t m, const double* __restrict__ Fm, const double* __restrict__ kinv,
                                         double s1, double t1x, double t1y, double s2, double t2x, double t2y, double maxThreshold,
                                         long long hist_base, bool want_bound, const double* __restrict__ la_g, const float* __restrict__ logc_n,
                                         const float* __restrict__ logc_k, double loge0, uint32_t* __restrict__ whist, uint32_t lane,
                                         uint32_t* __restrict__ out_total, double* __restrict__ out_bound, bool exact, double slack)
{
    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);
    double F[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) F[e] = Fm[e];
    if (KIND == 2) { double FE[9]; f_from_e(F, kinv, kinv + 9, FE);
#pragma unroll
        for (int e = 0; e < 9; ++e) F[e] = FE[e]; }
    if (want_bound) {
        uint4* z = reinterpret_cast<uint4*>(whist + 16u * lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
    }
    uint32_t total = 0, total_sure = 0;
    for (uint32_t base = 0; base < m; base += 256u) {
        double px[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = base + 64u * (uint32_t)u + lane;
            const float4 f4 = pt[p < m ? p : 0u];
            px[u][0] = s1 * (double)f4.x + t1x; px[u][1] = s1 * (double)f4.y + t1y;
            px[u][2] = s2 * (double)f4.z + t2x; px[u][3] = s2 * (double)f4.w + t2y;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = base + 64u * (uint32_t)u + lane;
            double r_lo, r_hi;
            if (exact || KIND == 1) {           // (H: the interval costs what the two divisions by one denominator do -- measured 10.1 -> 11.0 ms; it keeps them)
                r_lo = (KIND == 0) ? sym_epipolar_err(F, px[u][0], px[u][1], px[u][2], px[u][3])
                     : (KIND == 1) ? h_asym_err(F, px[u][0], px[u][1], px[u][2], px[u][3])
                                   : epipolar_dist_err(F, px[u][0], px[u][1], px[u][2], px[u][3]);
                r_hi = r_lo;
            } else {
                scout_residual<KIND>(F, px[u][0], px[u][1], px[u][2], px[u][3], r_lo, r_hi);
            }
            const bool in = (p < m) && (r_lo <= maxThreshold);
            total += (uint32_t)__builtin_popcountll(__ballot(in));
            total_sure += (uint32_t)__builtin_popcountll(__ballot((p < m) && (r_hi <= maxThreshold)));
            if (want_bound && in) {
                long long bin = (__double_as_longlong(r_lo) >> kHistShift) - hist_base;
                bin = bin < 0 ? 0 : (bin > kHistBins - 1 ? kHistBins - 1 : bin);
                atomicAdd(&whist[bin], 1u);
            }
        }
    }
    const bool open = ((double)total > 2.5 * SS) != ((double)total_sure > 2.5 * SS) || (total > SS) != (total_sure > SS);
    double bound = -__builtin_huge_val();                  // "no bound": the walk then takes the full evaluation
    if (want_bound && total > SS) {
        // inclusive running counts over the bins: 16 consecutive bins per lane, written back in place
        wave_lds_sync();
        uint32_t cb[16];
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { run += whist[16u * lane + (uint32_t)j]; cb[j] = run; }
        uint32_t incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
        const uint32_t excl = incl - run;
#pragma unroll
        for (int j = 0; j < 16; ++j) whist[16u * lane + (uint32_t)j] = excl + cb[j];
        wave_lds_sync();
        // bins lane, lane + 64, ...: neighbouring (equally dense) bins go to different lanes.  Within a bin the NFA term of k is
        // la k + g(k) + const with g = logc_n + logc_k; g is concave in k (log-binomials) up to the rounding drift of its float
        // tables, so over the k of a bin the term is smallest at one END of the range, less `slack` (scout_slack below bounds twice the
        // drift): two table reads per bin, all of a lane's in flight at once, where one read pair per k made this loop -- a chain of
        // dependent global loads, up to the bin's whole count long -- the larger part of a model's cost (the scout's first form).
        double wmin = __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = lane + 64u * (uint32_t)j;
            const uint32_t k_hi = whist[b], k_prev = b ? whist[b - 1] : 0u;
            uint32_t k_lo = k_prev + 1u; if (k_lo < SS + 1u) k_lo = SS + 1u;
            const bool some = k_hi >= k_lo;
            const uint32_t ka = some ? k_lo : 0u, kb = some ? k_hi : 0u;
            const double la = la_g[b];
            const double wa = loge0 + la * (double)(ka - SS) + (double)logc_n[ka] + (double)logc_k[ka];
            const double wb = loge0 + la * (double)(kb - SS) + (double)logc_n[kb] + (double)logc_k[kb];
            const double w = wa < wb ? wa : wb;
            wmin = (some && w < wmin) ? w : wmin;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(wmin, off); wmin = o < wmin ? o : wmin; }
        bound = wmin - slack;
    }
    if (lane == 0) { *out_total = total | (open ? kScoutOpen : 0u); *out_bound = bound; }
}

// NT = threads of the workgroup: 256 (two workgroups per CU), or 512 for collections with long match lists (one workgroup per CU with
// the same registers per lane; every pass over the matches of a pair -- residuals, bound, sort, NFA scan -- takes half the trips)
template <int KIND, bool SPILL, int NT, class KeyT, class IdxT>
__device__ __forceinline__ void acransac_body(const FilterParams& P, double* __restrict__ pts, uint32_t* __restrict__ pool_g,
                                              float* __restrict__ logc_g, unsigned char* smem, KeyT keys, IdxT sidx, uint32_t item)
{
    FState& S = *reinterpret_cast<FState*>(smem);
    constexpr int MS = (KIND == 2) ? 90 : 27;                  // doubles per hypothesis: 9 x MAX_MODELS (27 also for H)
    double* Fs = reinterpret_cast<double*>(smem + kHdr);                                   // [64][MS]
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + 1024);                             // [kHistBins]

    constexpr uint32_t SS = (KIND == 0) ? 7u : (KIND == 1 ? 4u : 5u);    // Kernel::MINIMUM_SAMPLES
    constexpr double MAXM = (KIND == 0) ? 3.0 : (KIND == 1 ? 1.0 : 10.0); // Kernel::MAX_MODELS
    constexpr double MULT_ERR = (KIND == 1) ? 1.0 : 0.5;       // multError(): point-to-point vs point-to-line
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t NW = (uint32_t)NT / 64u;
    const uint64_t begin = P.offsets[2 * item], end = P.offsets[2 * item + 1];
    const uint32_t m = (uint32_t)(end - begin);
    const uint2 sl = P.pairs[item];
    const uint2 id = P.pair_ids[item];
    const ImgDev* __restrict__ Ip = P.imgs + sl.x;
    const ImgDev* __restrict__ Jp = P.imgs + sl.y;
    const r3dm_match* __restrict__ mm = P.matches + begin;
    // per-item slices of the global work arrays start at multiples of 32 elements (>= one 128-byte line for the narrowest
    // array): workgroups that share a CU -- and with it the vector L1 -- never share a cache line of data one of them is
    // still writing
    const uint64_t so = P.soff[item];
    // The positions of a pair's matches as the four FLOATS the views hold, 16 bytes per match (the first half of the slot a match has in
    // the points scratch): every reader re-forms the normalised f64 coordinates with the two operations that used to fill a 32-byte
    // record of doubles (bit-identical).  Round 4's PMC had this kernel fetch 7.4 GB per launch for 47 MB of input: two workgroups per
    // CU, 64 per XCD, each re-reading ~59 KB per model = the 4 MiB of an XCD's L2.  Half the footprint stays inside it.
    float4* __restrict__ pt = reinterpret_cast<float4*>(pts + 4 * so);
    uint32_t* __restrict__ pool = pool_g + so;
    uint32_t* __restrict__ inl = P.inl_idx + so;
    float* __restrict__ logc_n = logc_g + so;                 // m + 1 entries

    // ---- ACKernelAdaptor: normalisation N = [[s,0,-s w/2],[0,s,-s h/2],[0,0,1]], s = 1/sqrt(w h)
    const int wI = (int)Ip->width, hI = (int)Ip->height, wJ = (int)Jp->width, hJ = (int)Jp->height;
    const double s1 = (KIND == 2) ? 1.0 : 1.0 / sqrt((double)(wI * hI));
    const double s2 = (KIND == 2) ? 1.0 : 1.0 / sqrt((double)(wJ * hJ));
    const double t1x = (KIND == 2) ? 0.0 : -0.5 * wI * s1, t1y = (KIND == 2) ? 0.0 : -0.5 * hI * s1;
    const double t2x = (KIND == 2) ? 0.0 : -0.5 * wJ * s2, t2y = (KIND == 2) ? 0.0 : -0.5 * hJ * s2;
    const double* K1i = S.kinv;
    const double* K2i = S.kinv + 9;
    if (KIND == 2 && tid < 18) S.kinv[tid] = P.kinv[9 * (size_t)(tid < 9 ? sl.x : sl.y) + (tid < 9 ? tid : tid - 9)];
    for (uint32_t p = tid; p < m; p += NT) {
        const r3dm_match q = mm[p];
        pt[p] = make_float4(Ip->xy[2 * (size_t)q.i], Ip->xy[2 * (size_t)q.i + 1], Jp->xy[2 * (size_t)q.j], Jp->xy[2 * (size_t)q.j + 1]);
        pool[p] = p;
    }
    const double Dd = sqrt((double)wJ * (double)wJ + (double)hJ * (double)hJ);
    const double Aa = (double)wJ * (double)hJ;
    const double logalpha0 = (KIND == 0) ? log10(2.0 * Dd / Aa / s2)                       // 2 D / A / N2(0,0)
                           : (KIND == 1) ? log10(3.14159265358979323846 / Aa / (s2 * s2))   // pi / A / N2(0,0)^2
                                         : log10(2.0 * Dd / Aa * 0.5);                        // ACKernelAdaptorEssential
    const double maxThreshold = P.precision_px * P.precision_px * s2 * s2;
    const double loge0 = log10(MAXM * (double)(m - SS));
    // histogram bins: the top kHistSub + 12 bits of the residual's IEEE pattern (monotone for r >= 0), counted down from the
    // pattern of the largest residual that is kept; everything below the lowest bin shares bin 0 (lower edge 0)
    const long long hist_base = (__double_as_longlong(fmin(maxThreshold, 1.0e6)) >> kHistShift) - (long long)(kHistBins - 1);

    if (tid == 0) {
        // logcombi(k, m) as a running prefix in the reference's float accumulation order
        // (makelogcombi_n): pre[i] = pre[i-1] + (l10[m-i+1] - l10[i]); mirrored for k > m/2
        float pre = 0.0f;
        logc_n[0] = 0.0f; logc_n[m] = 0.0f;
        for (uint32_t i = 1; i <= m / 2; ++i) {
            pre = pre + (P.log10_tab[m - i + 1] - P.log10_tab[i]);
            logc_n[i] = pre;
            if (m - i > i) logc_n[m - i] = pre;
        }
        S.minNFA = __builtin_huge_val(); S.errorMax = __builtin_huge_val();
        for (int e = 0; e < 9; ++e) S.bestF[e] = 0.0;
        const uint32_t reserve = P.max_iter / 10;
        S.reserve = reserve; S.nIter = P.max_iter - reserve; S.iter = 0;
        S.pool_size = m; S.n_inl = 0; S.acMode = !(P.precision_px < __builtin_huge_val());
        S.n_models = 0; S.iters_done = 0;
        S.cnt = 0u;
    }
#pragma unroll
    for (int j = 0; j < kHistBins / NT; ++j) hist[tid + NT * j] = 0u;
    // the scout pass (scout_model above): on in the product; the developer build can switch it off (A/B, parity of the two walks) and
    // its traces / invariant checks take the full evaluation for every model
    // (developer build: R3DM_FILTER_SCOUT=2 -- the scout divides like the full evaluation; =3 with R3DM_FILTER_CHECK=1 -- the scout runs,
    // nothing is skipped, and every model's count and NFA are checked against what the scout promised: invariants 9 and 10)
    const uint32_t scout_mode = P.scout & 0xFFu;
    const bool scout_check = scout_mode >= 3u && R3DM_DBG(P) && !R3DM_TRACE(P);
    const bool scout_on = scout_mode != 0u && P.la_tab != nullptr && !R3DM_TRACE(P) && (!R3DM_DBG(P) || scout_check);
    const bool scout_exact = scout_mode == 2u || scout_mode == 5u;
    double* __restrict__ la_g = P.la_tab ? P.la_tab + (size_t)item * kHistBins : nullptr;
    if (scout_on) {
#pragma unroll
        for (int j = 0; j < kHistBins / NT; ++j) {
            const uint32_t b = tid + (uint32_t)NT * (uint32_t)j;
            const double edge = b ? __longlong_as_double(((long long)b + hist_base) << kHistShift) : 0.0;
            la_g[b] = logalpha0 + MULT_ERR * log10(edge + FLT_EPS_D);
        }
    }
    // walk state every thread keeps for itself (all threads take the same decisions on the same shared values): AC mode, models walked,
    // iterations done -- the full evaluation used to keep them in LDS behind a barrier per model
    bool ac_reg = !(P.precision_px < __builtin_huge_val());
    uint32_t nmod_reg = 0, iters_done_reg = 0;
    wg_sync_global();          // points, pool, logcombi and slope tables go through global memory
    // what the float accumulation of the log-binomial tables (orc_logcombi_tables: <= m / 2 additions of float log10 differences, each
    // rounded at the running sum's ulp) can move them away from a concave sequence, twice: drift <= (m / 2) (2^-25 max + 2^-21)
    const double scout_slack = scout_on ? 0x1p-24 * (double)m * ((double)logc_n[m / 2u] + 32.0) + 1.0e-4 : 0.0;

#ifdef R3DM_E_TIMING
    unsigned long long cyc_solve = 0, cyc_eval = 0, cyc_t0 = 0, cyc_res = 0, cyc_sort = 0;
#endif
    while (true) {
        wg_sync_t<SPILL>();
#ifdef R3DM_E_TIMING
        cyc_t0 = __builtin_readcyclecounter();
#endif
        const uint32_t iter0 = S.iter, nIter0 = S.nIter;
        uint32_t nIter_reg = nIter0, reserve_reg = S.reserve;
        if (iter0 >= nIter0) break;
        constexpr uint32_t CH = (uint32_t)ChunkOf<KIND>::n;
        const uint32_t chunk_n = (nIter0 - iter0 < CH) ? nIter0 - iter0 : CH;
        // the hypothesis of the chunk this lane works on: lane c of wave 0 for F / H, the 16-lane group tid / 16 for E
        const uint32_t hyp = (KIND == 2) ? (tid >> 4) : tid;             // (threads beyond the first 256 of the wide variant draw nothing: hyp >= 16)
        const uint32_t pool_size = S.pool_size;

        // ---- draw + solve one chunk of minimal samples (hypothesis c <-> iteration iter0 + c)
        if ((KIND == 2 || tid < (uint32_t)kChunk) && hyp < chunk_n) {
            uint32_t pos[7];
            uint32_t cnt = 0, attempt = 0;
            while (cnt < SS) {
                const uint64_t r = rng_u64(P.seed, id.x, id.y, iter0 + hyp, attempt++);
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
                uint32_t sidx_ = pool[pos[k < (int)SS ? k : 0]];
                FCHECK(pos[k < (int)SS ? k : 0] < pool_size, 1, pos[k < (int)SS ? k : 0], pool_size);
                FCHECK(sidx_ < m, 2, sidx_, m);
                if (R3DM_DBG(P) && sidx_ >= m) sidx_ = 0;
                const float4 f4 = pt[sidx_];
                px1[k][0] = s1 * (double)f4.x + t1x; px1[k][1] = s1 * (double)f4.y + t1y;
                px2[k][0] = s2 * (double)f4.z + t2x; px2[k][1] = s2 * (double)f4.w + t2y;
                if (KIND == 2) {                 // camera coordinates: hnormalized(K^-1 (x, y, 1))
                    const double xa = px1[k][0], ya = px1[k][1], xb = px2[k][0], yb = px2[k][1];
                    const double w1 = K1i[6] * xa + K1i[7] * ya + K1i[8];
                    px1[k][0] = (K1i[0] * xa + K1i[1] * ya + K1i[2]) / w1;
                    px1[k][1] = (K1i[3] * xa + K1i[4] * ya + K1i[5]) / w1;
                    const double w2 = K2i[6] * xb + K2i[7] * yb + K2i[8];
                    px2[k][0] = (K2i[0] * xb + K2i[1] * yb + K2i[2]) / w2;
                    px2[k][1] = (K2i[3] * xb + K2i[4] * yb + K2i[5]) / w2;
                }
            }
            if constexpr (KIND == 2) {
                // the 16 lanes of the group solve the sample together; the models go straight into the hypothesis' LDS slots, the
                // workspaces are the region behind the hypothesis buffer (the sort buffers of the evaluation phase, idle during the solves)
                double* W = reinterpret_cast<double*>(smem + kHdr + CH * MS * 8) + (size_t)hyp * kE5Stride;
                const int l = (int)(tid & 15u);
                const int nm = five_point_coop(px1, px2, Fs + hyp * MS, W, l);
                for (int e = 9 * nm + l; e < MS; e += kE5Lanes) Fs[hyp * MS + e] = 0.0;
                if (l == 0) {
                    S.nm[hyp] = (uint32_t)nm;
                    if (R3DM_TRACE1(P)) S.dbg_smp[hyp] = pool[pos[0]];
                    if (R3DM_TRACE2(P) && item == P.trace_item && iter0 + hyp == P.trace_iter) {
                        double* t = P.trace + 5 * (size_t)(P.trace_cap - 4);
                        for (int k = 0; k < 7; ++k) t[k] = (double)pool[pos[k < (int)SS ? k : 0]];
                        t[7] = nm; t[8] = pool_size; t[9] = iter0 + hyp;
                        for (int k = 0; k < 7; ++k) t[10 + k] = (double)pos[k < (int)SS ? k : 0];
                    }
                    if (R3DM_TRACE3(P) && hyp == 0) S.dbg_pool = pool_size;
                }
            } else {
                double F3[MS];
                int nm;
                if constexpr (KIND == 0) nm = seven_point(px1, px2, F3);
                else nm = four_point_h(px1, px2, F3);
                S.nm[tid] = (uint32_t)nm;
                if (R3DM_TRACE1(P)) S.dbg_smp[tid] = pool[pos[0]];
                if (R3DM_TRACE2(P) && item == P.trace_item && iter0 + tid == P.trace_iter) {
                    double* t = P.trace + 5 * (size_t)(P.trace_cap - 4);
                    for (int k = 0; k < 7; ++k) t[k] = (double)pool[pos[k]];
                    t[7] = nm; t[8] = pool_size; t[9] = iter0 + tid;
                    for (int k = 0; k < 7; ++k) t[10 + k] = (double)pos[k];
                }
                if (R3DM_TRACE3(P) && tid == 0) S.dbg_pool = pool_size;
                for (int e = 0; e < MS; ++e) Fs[tid * MS + e] = (e < 9 * nm) ? F3[e] : 0.0;
            }
        }
#ifdef R3DM_BISECT_VMWAIT
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
#endif
        wg_sync_t<SPILL>();
#ifdef R3DM_E_TIMING
        { const unsigned long long now = __builtin_readcyclecounter(); cyc_solve += now - cyc_t0; cyc_t0 = now; }
#endif

        // ---- flat offsets of the chunk's models (scout pass): moff[c] = models of the hypotheses before c
        uint32_t* __restrict__ sc_total = reinterpret_cast<uint32_t*>(smem + kHdr + CH * MS * 8);
        double* __restrict__ sc_bound = reinterpret_cast<double*>(sc_total + kScoutModels);
        uint32_t* __restrict__ moff = reinterpret_cast<uint32_t*>(sc_bound + kScoutModels);
        uint32_t* __restrict__ whist = reinterpret_cast<uint32_t*>(smem + kHdr + CH * MS * 8 + kScoutBytes) + wave * (uint32_t)kHistBins;   // (the idle LDS sort buffers)
        if (scout_on) {
            if (tid <= chunk_n) { uint32_t o = 0; for (uint32_t j = 0; j < tid; ++j) o += S.nm[j]; moff[tid] = o; }
            wg_sync_t<SPILL>();
        }
        uint32_t scouted = 0;                          // models [0, scouted) of the chunk hold scout results

        // ---- evaluate the chunk's iterations in order
        bool pool_changed = false;
        uint32_t c = 0;
        for (; c < chunk_n && !pool_changed; ++c) {
            const uint32_t it = iter0 + c;
            const uint32_t nm = S.nm[c];
            bool better = false;
            for (uint32_t k = 0; k < nm; ++k) {
                [[maybe_unused]] uint32_t chk_tot = 0u; [[maybe_unused]] double chk_bound = -__builtin_huge_val();
                if (scout_on) {
                    const uint32_t f = moff[c] + k;
                    if (f >= scouted) {
                        // scout the next sub-batch of models (a pool change discards what lies behind it)
                        const uint32_t n_chunk_models = moff[chunk_n];
                        // (two per wavefront: what lies behind an accepted model is scouted for nothing -- 16 per wavefront cost F 10.2 ms where 2 cost 6.9;
                        //  developer build: R3DM_FILTER_SCOUT_SUB)
                        const uint32_t sub_n = (P.scout >> 8) ? (P.scout >> 8) * NW : 2u * NW;
                        const uint32_t end_f = scouted + sub_n < n_chunk_models ? scouted + sub_n : n_chunk_models;
                        const bool want_bound = S.minNFA < __builtin_huge_val();
                        for (uint32_t fi = scouted + wave; fi < end_f; fi += NW) {
                            const bool hit = lane < chunk_n && moff[lane] <= fi && fi < moff[lane + 1u];
                            const uint32_t cc = (uint32_t)__builtin_ctzll(__ballot(hit));
                            const uint32_t kk = fi - moff[cc];
                            scout_model<KIND>(pt, m, Fs + cc * MS + kk * 9, S.kinv, s1, t1x, t1y, s2, t2x, t2y, maxThreshold, hist_base, want_bound,
                                              la_g, logc_n, P.logc_k, loge0, whist, lane, sc_total + fi, sc_bound + fi, scout_exact, scout_slack);
                        }
                        wg_sync_t<SPILL>();
                        scouted = end_f;
                    }
                    // can the full evaluation accept this model?  Not below AC mode's 2.5 x sample size, not with SS or fewer matches
                    // within the bound, not when the best NFA its residual histogram allows is above the best so far
                    const uint32_t tot_w = sc_total[f], tot_f = tot_w & ~kScoutOpen;
                    const double bound_f = sc_bound[f], minNFA_f = S.minNFA;
                    const bool ac_f = ac_reg || ((double)tot_f > 2.5 * SS);
                    const bool skip = !(tot_w & kScoutOpen) &&
                                      (!(ac_f && tot_f > SS) ||
                                       (bound_f > -__builtin_huge_val() && minNFA_f < __builtin_huge_val() && bound_f - 1.0e-6 >= minNFA_f));
                    if (skip && !scout_check) { ac_reg = ac_f; nmod_reg += 1; continue; }
                    chk_tot = tot_w; chk_bound = bound_f;
                }
                double F[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) F[e] = Fs[c * MS + k * 9 + e];
                double FE[9];
                if (KIND == 2) f_from_e(F, K1i, K2i, FE);
                // residuals + compaction of those within the bound
#ifdef R3DM_E_TIMING
                const unsigned long long cyc_m0 = __builtin_readcyclecounter();
#endif
                // (order of the compacted entries: whatever the atomics give -- the sort below fixes the order, `total` and the
                // set do not depend on it)
                // (S.cnt and the histogram are zero here: cleared once at kernel start and again at the end of every model, behind the
                // barrier that closes it -- one barrier less per model than clearing them here)
                // four batches of 256 matches per trip: the point loads (global memory, 16 bytes per match) of all four are in
                // flight before the first residual is needed
                for (uint32_t base = 0; base < m; base += 4u * NT) {
                    double r[4]; bool in[4];
                    double px[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t p = base + (uint32_t)NT * (uint32_t)u + tid;
                        const float4 f4 = pt[p < m ? p : 0u];
                        px[u][0] = s1 * (double)f4.x + t1x; px[u][1] = s1 * (double)f4.y + t1y;
                        px[u][2] = s2 * (double)f4.z + t2x; px[u][3] = s2 * (double)f4.w + t2y;
                    }
                    unsigned long long bal[4];
                    uint32_t n_new = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t p = base + (uint32_t)NT * (uint32_t)u + tid;
                        r[u] = (KIND == 0) ? sym_epipolar_err(F, px[u][0], px[u][1], px[u][2], px[u][3])
                             : (KIND == 1) ? h_asym_err(F, px[u][0], px[u][1], px[u][2], px[u][3])
                                           : epipolar_dist_err(FE, px[u][0], px[u][1], px[u][2], px[u][3]);
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
                            FCHECK(pos < m || !in[u], 3, pos, m);
                            if (in[u]) {
                                keys[pos] = (unsigned long long)__double_as_longlong(r[u]); sidx[pos] = base + (uint32_t)NT * (uint32_t)u + tid;
                                long long bin = (__double_as_longlong(r[u]) >> kHistShift) - hist_base;
                                bin = bin < 0 ? 0 : (bin > kHistBins - 1 ? kHistBins - 1 : bin);
                                atomicAdd(&hist[bin], 1u);
                            }
                            woff += (uint32_t)__builtin_popcountll(bal[u]);
                        }
                    }
                }
                wg_sync_t<SPILL>();
                const uint32_t total = S.cnt;
#ifdef R3DM_E_TIMING
                const unsigned long long cyc_m1 = __builtin_readcyclecounter();
                cyc_res += cyc_m1 - cyc_m0;
#endif
                // AC mode switches on with the first model that has > 2.5*7 points within the bound
                bool ac = ac_reg;
                if (!ac && (double)total > 2.5 * SS) ac = true;
                double nfa = __builtin_huge_val();
                uint32_t kbest = SS;
                [[maybe_unused]] double S_bound = -__builtin_huge_val();
                // ---- can this model beat the best NFA at all?  The k-th smallest residual lies in the histogram bin where the
                // running count reaches k, so it is at least that bin's lower edge; the NFA term of k grows with the residual
                // (k > SS), hence NFA_k >= the same expression on the edge, evaluated with the same operations.  A model whose
                // bound over all k stays above the best NFA so far cannot be committed: no sort, no NFA scan (98 % of the models
                // on C2's pairs).  The margin covers log10's last-bit non-monotonicity (effect < 1e-10).
                bool hopeless = false;
                if (ac && total > SS && S.minNFA < __builtin_huge_val() && !R3DM_TRACE(P)) {
                    // inclusive running counts, 4 consecutive bins per thread, written back in place
                    uint32_t cb[kHistBins / NT];
                    uint32_t run = 0;
#pragma unroll
                    for (int j = 0; j < kHistBins / NT; ++j) { run += hist[tid * (kHistBins / NT) + j]; cb[j] = run; }
                    uint32_t incl = run;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off); if (lane >= (uint32_t)off) incl += o; }
                    if (lane == 63) S.wave_cnt[wave] = incl;
                    wg_sync_t<SPILL>();
                    uint32_t excl = incl - run;
#pragma unroll
                    for (uint32_t w = 0; w < NW; ++w) if (w < wave) excl += S.wave_cnt[w];
#pragma unroll
                    for (int j = 0; j < kHistBins / NT; ++j) hist[tid * (kHistBins / NT) + j] = excl + cb[j];
                    wg_sync_t<SPILL>();
                    // bins tid, tid + 256, ...: neighbouring (equally dense) bins go to different threads
                    double wmin = __builtin_huge_val();
#pragma unroll
                    for (int j = 0; j < kHistBins / NT; ++j) {
                        const uint32_t b = tid + (uint32_t)NT * (uint32_t)j;
                        const uint32_t k_hi = hist[b], k_prev = b ? hist[b - 1] : 0u;
                        uint32_t k_lo = k_prev + 1u; if (k_lo < SS + 1u) k_lo = SS + 1u;
                        if (k_hi >= k_lo) {
                            const double edge = b ? __longlong_as_double(((long long)b + hist_base) << kHistShift) : 0.0;
                            const double la = logalpha0 + MULT_ERR * log10(edge + FLT_EPS_D);
                            for (uint32_t kk = k_lo; kk <= k_hi; ++kk) {
                                const double w = loge0 + la * (double)(kk - SS) + (double)logc_n[kk] + (double)P.logc_k[kk];
                                wmin = w < wmin ? w : wmin;
                            }
                        }
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(wmin, off); wmin = o < wmin ? o : wmin; }
                    if (lane == 0) S.red_b[wave] = wmin;
                    wg_sync_t<SPILL>();
                    wmin = S.red_b[0];
#pragma unroll
                    for (uint32_t w = 1; w < NW; ++w) wmin = fmin(wmin, S.red_b[w]);      // (own slots: the NFA reduction below uses red_v, no barrier in between)
                    hopeless = !R3DM_DBG(P) && (wmin - 1.0e-6 >= S.minNFA);
                    if (R3DM_DBG(P)) S_bound = wmin;              // developer build: nothing is skipped, the bound is checked against the NFA
                }
                if (ac && total > SS && !hopeless) {
                    // sort (residual, index) ascending: residuals are >= 0 so the u64 bit pattern orders them
                    uint32_t cap = 1; while (cap < total) cap <<= 1;
                    // (the 256-thread variant keeps its spilled lists on the plain network, as before: the register sort of a
                    // global list is instantiated for the wide variant only)
                    bool sorted = !SPILL || NT == 512;
                    if constexpr (!SPILL || NT == 512)
                    switch (cap / (uint32_t)NT) {
                        case 0: case 1: wg_sort_regs<1, SPILL>(keys, sidx, cap, total, tid); break;
                        case 2: wg_sort_regs<2, SPILL>(keys, sidx, cap, total, tid); break;
                        case 4: wg_sort_regs<4, SPILL>(keys, sidx, cap, total, tid); break;
                        case 8: wg_sort_regs<8, SPILL>(keys, sidx, cap, total, tid); break;
                        case 16: wg_sort_regs<16, SPILL>(keys, sidx, cap, total, tid); break;
                        case 32: wg_sort_regs<32, SPILL>(keys, sidx, cap, total, tid); break;
                        default: sorted = false; break;               // longer lists: the plain network below
                    }
                    if (!sorted) {
                    for (uint32_t q = total + tid; q < cap; q += NT) { keys[q] = ~0ull; sidx[q] = 0xFFFFFFFFu; }
                    wg_sync_t<SPILL>();
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
                            wg_sync_t<SPILL>();
                        }
                    }
                    }
#ifdef R3DM_E_TIMING
                    cyc_sort += __builtin_readcyclecounter() - cyc_m1;
#endif
                    // bestNFA: k = 8 .. total, first minimum wins
                    double bv = __builtin_huge_val(); uint32_t bk = 0xFFFFFFFFu;
                    for (uint32_t kk = SS + 1 + tid; kk <= total; kk += NT) {
                        const double e = __longlong_as_double((long long)keys[kk - 1]);
                        const double logalpha = logalpha0 + MULT_ERR * log10(e + FLT_EPS_D);
                        const double v = loge0 + logalpha * (double)(kk - SS) + (double)logc_n[kk] + (double)P.logc_k[kk];
                        if (v < bv) { bv = v; bk = kk; }
                    }
                    // wave argmin (value, then smaller k), then across the 4 waves
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const double ov = __shfl_xor(bv, off);
                        const uint32_t ok = __shfl_xor(bk, off);
                        if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
                    }
                    if (lane == 0) { S.red_v[wave] = bv; S.red_k[wave] = bk; }
                    wg_sync_t<SPILL>();
#pragma unroll
                    for (uint32_t w = 0; w < NW; ++w) {
                        const double ov = S.red_v[w]; const uint32_t ok = S.red_k[w];
                        if (w == 0 || ov < nfa || (ov == nfa && ok < kbest)) { nfa = ov; kbest = ok; }
                    }
                    if (kbest == 0xFFFFFFFFu) { nfa = __builtin_huge_val(); kbest = SS; }
                }
                // commit (every lane evaluates the same condition on the same shared values)
                const double minNFA = S.minNFA;
                const bool improve = ac && (nfa < minNFA);
                FCHECK(!improve || (kbest <= total && kbest > SS), 4, kbest, total);
                FCHECK(S_bound - 1.0e-6 <= nfa, 8, kbest, total);                 // the sort-skipping bound really is one
                if (scout_check) {
                    const uint32_t ct = chk_tot & ~kScoutOpen;
                    FCHECK(ct >= total && ((chk_tot & kScoutOpen) || (((double)ct > 2.5 * SS) == ((double)total > 2.5 * SS) && (ct > SS) == (total > SS))), 9, ct, total);
                    FCHECK(!(ac && total > SS) || !(S_bound > -__builtin_huge_val()) || chk_bound <= S_bound + 1.0e-9, 11, __float_as_uint((float)chk_bound), __float_as_uint((float)S_bound));   // ... the scout's bound is the full evaluation's, or below it
                    FCHECK(!(ac && total > SS) || chk_bound - 1.0e-6 <= nfa, 10, __float_as_uint((float)chk_bound), __float_as_uint((float)nfa));     // ... and so is the scout's
                }
                if (improve) {
                    for (uint32_t q = tid; q < kbest; q += NT) inl[q] = sidx[q];
                    better = true;
                }
                wg_sync_t<SPILL>();
                if (tid == 0) {
                    if (R3DM_TRACE4(P) && item == P.trace_item) {
                        const uint32_t row = *P.trace_rows;
                        if (row < P.trace_cap) {
                            double* t = P.trace + 5 * (size_t)row;
                            t[0] = it; t[1] = k + 10.0 * S.dbg_smp[c]; t[2] = total + 10000.0 * S.dbg_pool; t[3] = nfa; t[4] = (improve ? 1.0 : 0.0) + 2.0 * kbest;
                            *P.trace_rows = row + 1;
                        }
                    }
                    S.acMode = ac ? 1u : 0u;
                    if (improve) {
                        S.minNFA = nfa;
                        S.n_inl = kbest;
                        S.errorMax = __longlong_as_double((long long)keys[kbest - 1]);
#pragma unroll
                        for (int e = 0; e < 9; ++e) S.bestF[e] = F[e];
                    }
                    S.cnt = 0u;                                   // for the next model (every reader of this model's count is behind the barrier above)
                }
#pragma unroll
                for (int j = 0; j < kHistBins / NT; ++j) hist[tid + NT * j] = 0u;     // ... and its histogram: last read before the commit barrier
                ac_reg = ac; nmod_reg += 1;
                wg_sync_t<SPILL>();
            }
            // ---- end of iteration `it`, the common case: no model of it was accepted and the iteration budget does not end here --
            // nothing shared changes (ACRANSAC's pool / budget update below cannot trigger)
            if (scout_on && !better && !(it + 1 == nIter_reg && reserve_reg != 0u)) {
                iters_done_reg = it + 1;
                if (it + 1 >= nIter_reg) { ++c; break; }
                continue;
            }
            // ---- end of iteration `it`: ACRANSAC's pool / budget update.  Thread 0 is about to change the loop bounds that
            // the other waves read right after the previous iteration's last barrier; every evaluated model ends with a
            // barrier, a sample without real solutions (nm == 0, common for the 5-point solver) needs its own.
            if (nm == 0) wg_sync_t<SPILL>();
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
            wg_sync_t<SPILL>();
            if (S.flag) {
                // new sampling pool = the inlier SET in ascending index order.  (The residual order of the
                // inlier list is rounding noise among the 7 points the model was fitted to, so pool
                // positions must not depend on it -- same rule in oracle/acransac.c.)
                const uint32_t ni = S.n_inl;
                IdxT flags = sidx;                                       // sort scratch, free between models
                for (uint32_t q = tid; q < m; q += NT) flags[q] = 0u;
                wg_sync_t<SPILL>();
                FCHECK(ni <= m, 5, ni, m);
                for (uint32_t q = tid; q < ni; q += NT) { const uint32_t iq = inl[q]; FCHECK(iq < m, 6, iq, q); if (!R3DM_DBG(P) || iq < m) flags[iq] = 1u; }
                wg_sync_t<SPILL>();
                uint32_t filled = 0;
                for (uint32_t base = 0; base < m; base += NT) {
                    const uint32_t p = base + tid;
                    const bool in = (p < m) && (flags[p] != 0u);
                    const unsigned long long bal = __ballot(in);
                    const uint32_t before = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (lane == 0) S.wave_cnt[wave] = (uint32_t)__builtin_popcountll(bal);
                    wg_sync_t<SPILL>();
                    uint32_t woff = 0, tot = 0;
#pragma unroll
                    for (uint32_t w = 0; w < NW; ++w) { const uint32_t cw = S.wave_cnt[w]; if (w < wave) woff += cw; tot += cw; }
                    if (in) pool[filled + woff + before] = p;
                    filled += tot;
                    wg_sync_t<SPILL>();
                }
                FCHECK(filled == ni, 7, filled, ni);
                pool_changed = true;
                wg_fence();            // the rebuilt pool (global memory) is read by the sampling lanes of the next chunk (same workgroup)
            }
            wg_sync_t<SPILL>();
            iters_done_reg = it + 1;
            nIter_reg = S.nIter; reserve_reg = S.reserve;
            // the chunk was cut from the old budget: stop when the (possibly shrunk) budget is exhausted
            if (it + 1 >= S.nIter) { ++c; break; }
        }
        if (tid == 0) S.iter = iter0 + c;
#ifdef R3DM_E_TIMING
        cyc_eval += __builtin_readcyclecounter() - cyc_t0;
#endif
    }

    // ---- result
    wg_sync_t<SPILL>();
    if (tid == 0) {
        uint32_t n_inl = S.n_inl;
        if (!(S.minNFA < 0.0)) n_inl = 0;
        P.inl_count[item] = n_inl;
        double Fo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double thr = 0.0;
        if (n_inl > 0) {
            // Unnormalize: F = N2^T * F * N1 (UnnormalizerT) ; H = N2^-1 * H * N1 (UnnormalizerI)
            const double N1[9] = {s1, 0, t1x, 0, s1, t1y, 0, 0, 1};
            const double N2[9] = {s2, 0, t2x, 0, s2, t2y, 0, 0, 1};
            const double N2i[9] = {1.0 / s2, 0, -t2x / s2, 0, 1.0 / s2, -t2y / s2, 0, 0, 1};
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
            if (KIND == 2) { for (int e = 0; e < 9; ++e) Fo[e] = S.bestF[e]; thr = S.errorMax; }   // E itself; unormalizeError(e) = e
        }
        for (int e = 0; e < 9; ++e) P.F_out[9 * (size_t)item + e] = Fo[e];
        P.thr_nfa[2 * (size_t)item] = thr;
        P.thr_nfa[2 * (size_t)item + 1] = S.minNFA;
#ifdef R3DM_E_TIMING
        // timing builds only (tools/build_bisect.sh): cycles of wave 0 in the solve and the evaluation phases instead of (threshold, NFA)
        P.thr_nfa[2 * (size_t)item] = (double)cyc_solve + 1e-12 * (double)cyc_res;      // (integers below 2^40: both survive in one double pair)
        P.thr_nfa[2 * (size_t)item + 1] = (double)cyc_eval + 1e-12 * (double)cyc_sort;
#endif
        P.iters[2 * (size_t)item] = iters_done_reg;
        P.iters[2 * (size_t)item + 1] = nmod_reg;
    }
}

// Two workgroups per CU for F and H (256 registers per lane; the F kernel trades 42 spilled VGPRs of its cold solver path
// for the second workgroup: 40.0 -> 24.8 ms on the 790 pairs of C2, three per CU at 168 registers gains nothing more);
// E likewise since its 5-point solver became a cooperative LDS routine.  The wide variant (NT = 512, one workgroup per CU, the same
// registers per lane) serves collections with long match lists: FilterParams::wide, set by the host (api_filter.cpp).
template <int KIND, int NT>
__device__ __forceinline__ void acransac_item(const FilterParams& P, unsigned char* smem, uint32_t block)
{
    constexpr int MS = (KIND == 2) ? 90 : 27;
    const uint32_t item = P.order ? P.order[block] : block;
    const uint32_t m = (uint32_t)(P.offsets[2 * item + 1] - P.offsets[2 * item]);
    if (m <= P.m_cap) {
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem + kHdr + ChunkOf<KIND>::n * MS * 8 + kScoutBytes);
        uint32_t* sidx = reinterpret_cast<uint32_t*>(keys + P.m_cap);
        acransac_body<KIND, false, NT>(P, P.pts_scratch, P.pool_scratch, P.scratch_logc, smem, keys, sidx, item);
    } else {
        // slice of the global spill buffer, sized for the next power of two of m (host: spill_off[item])
        unsigned long long* keys = P.spill_keys + P.spill_off[item];
        uint32_t* sidx = P.spill_idx + P.spill_off[item];
        acransac_body<KIND, true, NT>(P, P.pts_scratch, P.pool_scratch, P.scratch_logc, smem, keys, sidx, item);
    }
}

template <int KIND, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1)
void acransac_kernel(const FilterParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    acransac_item<KIND, NT>(P, smem, blockIdx.x);
}

// The essential-matrix instantiation lives in its own translation unit (kernels_filter_e.hip = this file with
// R3DM_FILTER_ONLY_E) only to compile in parallel with F / H.  (Round 2 needed -mllvm -amdgpu-spill-sgpr-to-vgpr=0 there: the
// one-lane 5-point solver was called out of line from divergent control flow and SGPRs spilled into VGPR lanes across the calls
// gave run-to-run different inlier sets.  The cooperative solver above has no calls and no spills; the option is gone, and
// tests/test_gpu_fullsize.py::test_filters_are_deterministic_when_workgroups_share_a_cu runs against the default build.)
hipError_t launch_filter_E(hipStream_t st, const FilterParams& P, size_t lds);
template <int KIND, int NT>
static hipError_t launch_acransac(hipStream_t st, const FilterParams& P, size_t lds)
{
    hipError_t e = hipFuncSetAttribute((const void*)acransac_kernel<KIND, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((acransac_kernel<KIND, NT>), dim3(P.n_short), dim3(NT), lds, st, P);
    return hipGetLastError();
}
#ifdef R3DM_FILTER_ONLY_E
hipError_t launch_filter_E(hipStream_t st, const FilterParams& P, size_t lds)
{
    return P.wide ? launch_acransac<2, 512>(st, P, lds) : launch_acransac<2, 256>(st, P, lds);
}
#else
hipError_t launch_filter_F(hipStream_t st, const FilterParams& P)
{
    if (P.n_short == 0) return hipSuccess;
    const size_t lds = filter_F_lds_bytes(P.m_cap, P.model_kind);
    if (P.model_kind == 2) return launch_filter_E(st, P, lds);
    if (P.model_kind == 0) return P.wide ? launch_acransac<0, 512>(st, P, lds) : launch_acransac<0, 256>(st, P, lds);
    return P.wide ? launch_acransac<1, 512>(st, P, lds) : launch_acransac<1, 256>(st, P, lds);
}
#endif
#endif   // R3DM_FILTER_DEVICE_ONLY

}  // namespace r3dm
