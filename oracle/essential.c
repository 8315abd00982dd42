/*
 * essential.c -- CPU restatement of the five-point essential-matrix solver and the error the E filter uses.
 * TEST INFRASTRUCTURE ONLY (see r3d_oracle.h).
 *
 * Reference call site: GeometricFilter_EMatrix_AC(4.0, imax_iteration) at
 * /root/reference/src/R3DComputeMatches.cpp:2169 (inside `if(params.computeEssentialMatrix_)`, :2130-2200).  The
 * arithmetic lives in OpenMVG 1.4 (external, not vendored -- r3d_oracle.h): E_ACRobust.hpp builds
 *   ACKernelAdaptorEssential<essential::kernel::FivePointKernel, fundamental::kernel::EpipolarDistanceError, Mat3>
 * and runs ACRANSAC with precision 4.0^2; kept iff #inliers > 2.5 * 5.
 *
 * OpenMVG's FivePointsRelativePose (Stewenius' formulation) takes the 4-dim null space of the 5 epipolar equations,
 * builds the 10 cubic constraints (det E = 0 and 2 E E^T E - tr(E E^T) E = 0), Gauss-Jordan-eliminates the 10x20
 * coefficient matrix and reads the real solutions off the eigenvectors of a 10x10 action matrix (Eigen::EigenSolver).
 * This restatement solves the SAME polynomial system with Nister's elimination (PAMI 2004, section 3.2): after
 * the Gauss-Jordan step three rows combine into a 3x3 matrix B(z) of polynomials in z whose determinant is the
 * tenth-degree polynomial of the hidden variable; its real roots are isolated with a Sturm sequence + bisection and
 * (x, y) follow from the null vector of B(z).  The solution SET is the same (up to rounding); restatement decisions:
 *   - null space from a Householder QR of A^T (like the 7-point restatement), not an SVD -- any basis of the null
 *     space yields the same essential matrices up to scale;
 *   - models are emitted in ascending order of the root z (Eigen's eigenvalue order is an implementation detail);
 *   - only +, -, *, / and comparisons are used after the QR, so the HIP kernel reproduces every bit.
 */
#include "r3d_oracle.h"

#include <math.h>
#include <string.h>

/* monomial orders.  degree <= 1: x y z 1;  degree <= 2: xx yy xy xz x yz y zz z 1;
 * degree <= 3 (Nister's column order): xxx yyy xxy xyy xxz xx yyz yy xyz xy | xzz xz x yzz yz y zzz zz z 1 */
static const unsigned char T11[4][4] = {{0, 2, 3, 4}, {2, 1, 5, 6}, {3, 5, 7, 8}, {4, 6, 8, 9}};
static const unsigned char T21[10][4] = {{0, 2, 4, 5}, {3, 1, 6, 7}, {2, 3, 8, 9}, {4, 8, 10, 11}, {5, 9, 11, 12},
                                         {8, 6, 13, 14}, {9, 7, 14, 15}, {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}};

static void mul11(const double* a, const double* b, double* out /* 10 */)
{
    for (int k = 0; k < 10; ++k) out[k] = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[T11[i][j]] += a[i] * b[j];
}
static void mul21_acc(const double* a /* 10 */, const double* b /* 4 */, double* out /* 20, accumulated */)
{
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) out[T21[i][j]] += a[i] * b[j];
}

/* polynomials in z, ascending coefficients */
static void pmul(const double* a, int da, const double* b, int db, double* out)
{
    for (int k = 0; k <= da + db; ++k) out[k] = 0.0;
    for (int i = 0; i <= da; ++i)
        for (int j = 0; j <= db; ++j) out[i + j] += a[i] * b[j];
}
static double peval(const double* p, int d, double t)
{
    double v = p[d];
    for (int k = d - 1; k >= 0; --k) v = v * t + p[k];
    return v;
}

/* number of sign changes of the Sturm chain at t */
static int sturm_changes(const double (*f)[11], const int* deg, int nf, double t)
{
    int changes = 0, last = 0;
    for (int k = 0; k < nf; ++k) {
        const double v = peval(f[k], deg[k], t);
        const int s = (v > 0.0) - (v < 0.0);
        if (s != 0) { if (last != 0 && s != last) ++changes; last = s; }
    }
    return changes;
}

/* real roots of p (degree <= 10), ascending, distinct.  Returns their number. */
int orc_real_roots10(const double* p_in, int deg_in, double* roots)
{
    double f[12][11];
    int deg[12];
    int d = deg_in;
    while (d > 0 && p_in[d] == 0.0) --d;
    if (d <= 0) return 0;
    for (int k = 0; k <= d; ++k) f[0][k] = p_in[k] / p_in[d];
    deg[0] = d;
    for (int k = 1; k <= d; ++k) f[1][k - 1] = (double)k * f[0][k];
    deg[1] = d - 1;
    int nf = 2;
    while (deg[nf - 1] > 0) {
        /* f[nf] = -rem(f[nf-2], f[nf-1]), scaled by 1/|leading coefficient| */
        const double* b = f[nf - 1];
        const int db = deg[nf - 1];
        double r[11];
        int dr = deg[nf - 2];
        for (int k = 0; k <= dr; ++k) r[k] = f[nf - 2][k];
        while (dr >= db) {
            const double q = r[dr] / b[db];
            for (int k = 0; k < db; ++k) r[dr - db + k] -= q * b[k];
            r[dr] = 0.0;
            --dr;
        }
        while (dr >= 0 && r[dr] == 0.0) --dr;
        if (dr < 0) break;                       /* exact common factor: the chain ends */
        const double sc = fabs(r[dr]);
        for (int k = 0; k <= dr; ++k) f[nf][k] = -r[k] / sc;
        deg[nf] = dr;
        ++nf;
    }
    double bound = 0.0;
    for (int k = 0; k < d; ++k) { const double a = fabs(f[0][k]); if (a > bound) bound = a; }
    bound += 1.0;                                 /* Cauchy: every root lies in (-bound, bound) */
    const int va = sturm_changes(f, deg, nf, -bound);
    const int nr = va - sturm_changes(f, deg, nf, bound);
    int n_out = 0;
    for (int r = 1; r <= nr && n_out < 10; ++r) {
        double lo = -bound, hi = bound;
        for (int it = 0; it < 64; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (va - sturm_changes(f, deg, nf, mid) >= r) hi = mid; else lo = mid;
        }
        roots[n_out++] = 0.5 * (lo + hi);
    }
    return n_out;
}

/* x1, x2: 5 correspondences in camera (K^-1) coordinates, 2 doubles each.  Es: up to 10 row-major 3x3 matrices with
 * x2^T E x1 = 0.  Returns their number. */
int orc_five_point(const double* x1, const double* x2, double* Es)
{
    /* ---- 4-dim null space of the 5 x 9 epipolar system: last 4 columns of Q, A^T = Q R */
    double M[9][5];
    for (int p = 0; p < 5; ++p) {
        const double ax = x1[2 * p], ay = x1[2 * p + 1], bx = x2[2 * p], by = x2[2 * p + 1];
        M[0][p] = bx * ax; M[1][p] = bx * ay; M[2][p] = bx;
        M[3][p] = by * ax; M[4][p] = by * ay; M[5][p] = by;
        M[6][p] = ax;      M[7][p] = ay;      M[8][p] = 1.0;
    }
    double beta[5];
    for (int j = 0; j < 5; ++j) {
        double nrm2 = 0.0;
        for (int r = j; r < 9; ++r) nrm2 += M[r][j] * M[r][j];
        const double nrm = sqrt(nrm2);
        double bj = 0.0;
        if (nrm != 0.0) {
            const double alpha = (M[j][j] > 0.0) ? -nrm : nrm;
            M[j][j] -= alpha;
            double vn2 = 0.0;
            for (int r = j; r < 9; ++r) vn2 += M[r][j] * M[r][j];
            if (vn2 != 0.0) bj = 2.0 / vn2;
        } else {
            for (int r = j; r < 9; ++r) M[r][j] = 0.0;
        }
        beta[j] = bj;
        for (int c = j + 1; c < 5; ++c) {
            double dot = 0.0;
            for (int r = j; r < 9; ++r) dot += M[r][j] * M[r][c];
            const double s = bj * dot;
            for (int r = j; r < 9; ++r) M[r][c] -= s * M[r][j];
        }
    }
    double N[4][9];                                   /* X, Y, Z, W */
    for (int e = 0; e < 4; ++e) {
        for (int r = 0; r < 9; ++r) N[e][r] = (r == 5 + e) ? 1.0 : 0.0;
        for (int j = 4; j >= 0; --j) {
            double dot = 0.0;
            for (int r = j; r < 9; ++r) dot += M[r][j] * N[e][r];
            const double s = beta[j] * dot;
            for (int r = j; r < 9; ++r) N[e][r] -= s * M[r][j];
        }
    }

    /* ---- the 10 cubic constraints as a 10 x 20 coefficient matrix */
    double E1[9][4];                                  /* E_ij as a polynomial of degree 1: [x y z 1] */
    for (int r = 0; r < 9; ++r) for (int e = 0; e < 4; ++e) E1[r][e] = N[e][r];
    double A[10][20];
    for (int r = 0; r < 10; ++r) for (int c = 0; c < 20; ++c) A[r][c] = 0.0;
    double t1[10], t2[10], d2[10];
    /* det E = 0 */
    mul11(E1[1], E1[5], t1); mul11(E1[2], E1[4], t2); for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
    mul21_acc(d2, E1[6], A[0]);
    mul11(E1[2], E1[3], t1); mul11(E1[0], E1[5], t2); for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
    mul21_acc(d2, E1[7], A[0]);
    mul11(E1[0], E1[4], t1); mul11(E1[1], E1[3], t2); for (int k = 0; k < 10; ++k) d2[k] = t1[k] - t2[k];
    mul21_acc(d2, E1[8], A[0]);
    /* 2 E E^T E - trace(E E^T) E = 0  <=>  (E E^T - 1/2 trace(E E^T) I) E = 0 */
    double EET[3][3][10];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a0[10], a1[10], a2[10];
            mul11(E1[3 * i], E1[3 * j], a0); mul11(E1[3 * i + 1], E1[3 * j + 1], a1); mul11(E1[3 * i + 2], E1[3 * j + 2], a2);
            for (int k = 0; k < 10; ++k) EET[i][j][k] = a0[k] + a1[k] + a2[k];
        }
    double tr[10];
    for (int k = 0; k < 10; ++k) tr[k] = 0.5 * (EET[0][0][k] + EET[1][1][k] + EET[2][2][k]);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 10; ++k) EET[i][i][k] -= tr[k];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double* row = A[1 + 3 * i + j];
            mul21_acc(EET[i][0], E1[j], row); mul21_acc(EET[i][1], E1[3 + j], row); mul21_acc(EET[i][2], E1[6 + j], row);
        }

    /* ---- Gauss-Jordan on the first 10 columns, partial pivoting */
    for (int c = 0; c < 10; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 10; ++r) { const double v = fabs(A[r][c]); if (v > best) { best = v; piv = r; } }
        if (best == 0.0) return 0;
        if (piv != c) for (int k = 0; k < 20; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        const double inv = 1.0 / A[c][c];
        for (int k = c; k < 20; ++k) A[c][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double fct = A[r][c];
            if (fct == 0.0) continue;
            for (int k = c; k < 20; ++k) A[r][k] -= fct * A[c][k];
        }
    }

    /* ---- B(z): rows  <e> - z<f>,  <g> - z<h>,  <i> - z<j>  (rows 4..9); columns: x (deg 3), y (deg 3), 1 (deg 4) */
    double B[3][3][5];
    for (int q = 0; q < 3; ++q) {
        const double* lo = A[4 + 2 * q];     /* leading monomial  x^2 z / y^2 z / x y z */
        const double* hi = A[5 + 2 * q];     /* leading monomial  x^2   / y^2   / x y   */
        for (int v = 0; v < 2; ++v) {        /* x: columns 10..12, y: columns 13..15  (z^2, z, 1) */
            const int o = 10 + 3 * v;
            B[q][v][0] = lo[o + 2];
            B[q][v][1] = lo[o + 1] - hi[o + 2];
            B[q][v][2] = lo[o] - hi[o + 1];
            B[q][v][3] = -hi[o];
            B[q][v][4] = 0.0;
        }
        B[q][2][0] = lo[19];
        B[q][2][1] = lo[18] - hi[19];
        B[q][2][2] = lo[17] - hi[18];
        B[q][2][3] = lo[16] - hi[17];
        B[q][2][4] = -hi[16];
    }
    /* det B(z): expansion along the third column (degree 4) of 2x2 minors of the first two (degree 6) */
    double P[11];
    for (int k = 0; k <= 10; ++k) P[k] = 0.0;
    for (int q = 0; q < 3; ++q) {
        const int r1 = (q + 1) % 3, r2 = (q + 2) % 3;
        double m1[7], m2[7], mn[7], term[11];
        pmul(B[r1][0], 3, B[r2][1], 3, m1);
        pmul(B[r1][1], 3, B[r2][0], 3, m2);
        for (int k = 0; k <= 6; ++k) mn[k] = m1[k] - m2[k];
        pmul(mn, 6, B[q][2], 4, term);
        for (int k = 0; k <= 10; ++k) P[k] += term[k];
    }
    double roots[10];
    const int nr = orc_real_roots10(P, 10, roots);
    int n_out = 0;
    for (int s = 0; s < nr; ++s) {
        const double z = roots[s];
        double b[3][3];
        for (int q = 0; q < 3; ++q) { b[q][0] = peval(B[q][0], 3, z); b[q][1] = peval(B[q][1], 3, z); b[q][2] = peval(B[q][2], 4, z); }
        /* (x, y, 1) is orthogonal to every row: cross product of the best-conditioned pair of rows */
        double bx = 0.0, by = 0.0, bw = 0.0;
        for (int q = 0; q < 3; ++q) {
            const int r1 = q, r2 = (q + 1) % 3;
            const double cx = b[r1][1] * b[r2][2] - b[r1][2] * b[r2][1];
            const double cy = b[r1][2] * b[r2][0] - b[r1][0] * b[r2][2];
            const double cw = b[r1][0] * b[r2][1] - b[r1][1] * b[r2][0];
            if (fabs(cw) > fabs(bw)) { bx = cx; by = cy; bw = cw; }
        }
        if (bw == 0.0) continue;
        const double x = bx / bw, y = by / bw;
        for (int r = 0; r < 9; ++r) Es[9 * n_out + r] = x * N[0][r] + y * N[1][r] + z * N[2][r] + N[3][r];
        ++n_out;
    }
    return n_out;
}

/* fundamental::kernel::EpipolarDistanceError: squared distance of x2 to the epipolar line F x1 */
double orc_epipolar_dist_err(const double* F, double x1, double y1, double x2, double y2)
{
    const double l0 = F[0] * x1 + F[1] * y1 + F[2];
    const double l1 = F[3] * x1 + F[4] * y1 + F[5];
    const double l2 = F[6] * x1 + F[7] * y1 + F[8];
    const double d = l0 * x2 + l1 * y2 + l2;
    return (d * d) / (l0 * l0 + l1 * l1);
}

/* inverse of a 3x3 (row-major) by the adjugate */
void orc_inv3(const double* K, double* Ki)
{
    const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[5] * K[6] - K[3] * K[8], c02 = K[3] * K[7] - K[4] * K[6];
    const double det = K[0] * c00 + K[1] * c01 + K[2] * c02;
    const double id = 1.0 / det;
    Ki[0] = c00 * id; Ki[1] = (K[2] * K[7] - K[1] * K[8]) * id; Ki[2] = (K[1] * K[5] - K[2] * K[4]) * id;
    Ki[3] = c01 * id; Ki[4] = (K[0] * K[8] - K[2] * K[6]) * id; Ki[5] = (K[2] * K[3] - K[0] * K[5]) * id;
    Ki[6] = c02 * id; Ki[7] = (K[1] * K[6] - K[0] * K[7]) * id; Ki[8] = (K[0] * K[4] - K[1] * K[3]) * id;
}

/* FundamentalFromEssential: F = K2^-T E K1^-1 */
void orc_f_from_e(const double* E, const double* K1i, const double* K2i, double* F)
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
