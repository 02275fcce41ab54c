/*
 * pnp_oracle.c — CPU restatement of cv2.solvePnPRansac(obj, img, K, dist=0, flags=SOLVEPNP_ITERATIVE) and cv2.Rodrigues
 * as called at experiments/pnp_two_way_tracking.py:57-61,68-69 (and the dead branch slam/tracking.py:68-71).
 *
 * TEST INFRASTRUCTURE (see orb_oracle.c).  PARITY UNPINNED: OpenCV (calib3d) is absent from /root/reference and from this
 * image and no fixture exists.  What is restated from OpenCV's published algorithm (calib3d/src/solvepnp.cpp, ptsetreg.cpp,
 * epnp.cpp, core RNG):
 *   - inputs are rounded to float32 before the RANSAC stage (solvePnPRansac converts CV_64F -> CV_32F);
 *   - minimal model: 5 points solved with EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009): 4 control points from the PCA of the
 *     sample, barycentric coordinates, 12x12 M'M null-space, betas for N = 1,2,3 null vectors + 5 Gauss-Newton steps, rigid
 *     alignment (Arun) of the control-point camera coordinates, best of the three by reprojection error;
 *   - sampler: cv::RNG((uint64)-1) multiply-with-carry (x = (u32)x*4164903690 + (x>>32)), uniform(0,n) = next() % n, a draw
 *     is repeated until it differs from the earlier members of the subset;
 *   - scoring: squared reprojection error in float32 <= reprojErr^2; a model replaces the best when its inlier count is
 *     greater than max(best count, 4); after every improvement niters = RANSACUpdateNumIters(conf, outlier ratio, 5, niters);
 *   - small dense solves: the 6x4 Gauss-Newton systems by Householder QR (as epnp.cpp's qr_solve), the three initial 6xN systems by
 *     a pseudo-inverse (eigen-decomposition of the normal equations; OpenCV: cvSolve(SVD)), the refinement's 6x6 damped normal equations
 *     by Cholesky (pnp_small.h; OpenCV: CvLevMarq solves with SVD);
 *   - final pose: minimiser of the reprojection error over the inliers (OpenCV: solvePnP(..., ITERATIVE) = DLT initial guess +
 *     Levenberg-Marquardt; here Levenberg-Marquardt started from the RANSAC model — same minimum whenever both converge);
 *   - Rodrigues: R = cos(th) I + (1-cos(th)) rr' + sin(th) [r]x and its inverse.
 * Linear-algebra kernels inside (symmetric Jacobi eigen-solver, one-sided Jacobi SVD of 3x3) are this restatement's own:
 * OpenCV's cvSVD leaves eigenvector signs / orders unspecified, which EPnP is invariant to.
 * n == 4 (OpenCV switches to P3P) is not restated: needs n >= 5.
 * Inside orc_solve_pnp_ransac the five libm calls OpenCV makes (sin/cos of the refinement's rotation update, acos of the final
 * Rodrigues, pow and log of RANSACUpdateNumIters) go through detmath.h: fixed-operation-order routines within 2 ulp of libm, so
 * that the HIP kernel can reproduce this function bit for bit (the result feeds hard thresholds and the next frame's input).
 * The public orc_rodrigues_* keep libm, like cv2.Rodrigues.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "detmath.h"
#include "pnp_small.h"
#include "p3p_small.h"

/* ---------- small linear algebra ---------- */
/* cyclic Jacobi for symmetric A (n<=12), eigenvalues descending in d, eigenvectors in ROWS of Vt */
static void jacobi_eig(const double* Ain, int n, double* d, double* Vt)
{
    double A[144], V[144];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        int rot = 0;
        for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                /* standard threshold of the cyclic method: a rotation this small cannot change a double any more */
                if (fabs(apq) < 1e-300 || fabs(apq) <= 1e-17 * sqrt(fabs(A[p * n + p] * A[q * n + q]))) continue;
                rot = 1;
                double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
        if (!rot) break;
    }
    int order[12];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (A[order[j] * n + order[j]] > A[order[i] * n + order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int i = 0; i < n; i++) { d[i] = A[order[i] * n + order[i]]; for (int k = 0; k < n; k++) Vt[i * n + k] = V[k * n + order[i]]; }
}


/* The same eigen-problem with the rotations of a sweep arranged in ROUNDS of disjoint pairs (round-robin tournament ordering: with
 * M = n rounded up to even, round r pairs the fixed player M-1 with r and (r+k) mod (M-1) with (r-k) mod (M-1), k = 1 .. M/2-1; a pair
 * with the dummy player of an odd n is a bye).  All rotation angles of a round are taken from the matrix as it stands at the start of
 * the round (the diagonal and the pivot element of one pair are not touched by the others), then every column rotation of the round is
 * applied, then every row rotation.  Mathematically the classical Jacobi method with another pivot order; on the GPU a round is one
 * step of one wave instead of up to six dependent rotations, and this restatement fixes the operation order the kernel reproduces.
 * Used for the 12x12 M'M of EPnP and the small normal equations of lsq_solve. */
static void jacobi_eig_rr(const double* Ain, int n, double* d, double* Vt)
{
    double A[144], V[144];
    const int M = (n + 1) & ~1, NP = M / 2;
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        int rot = 0;
        for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int r = 0; r < M - 1; r++) {
            int P[6], Q[6], act[6]; double C[6], S[6];
            for (int i = 0; i < NP; i++) {
                int a = i == 0 ? M - 1 : (r + i) % (M - 1), b = i == 0 ? r : (r - i + (M - 1)) % (M - 1);
                int p = a < b ? a : b, q = a < b ? b : a;
                P[i] = p; Q[i] = q; act[i] = 0; C[i] = 1.0; S[i] = 0.0;
                if (q >= n) continue;                                   /* bye */
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300 || fabs(apq) <= 1e-17 * sqrt(fabs(A[p * n + p] * A[q * n + q]))) continue;
                double tau = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                C[i] = 1.0 / sqrt(1.0 + t * t); S[i] = t * C[i]; act[i] = 1; rot = 1;
            }
            for (int i = 0; i < NP; i++) if (act[i]) {
                const int p = P[i], q = Q[i]; const double c = C[i], s = S[i];
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
            for (int i = 0; i < NP; i++) if (act[i]) {
                const int p = P[i], q = Q[i]; const double c = C[i], s = S[i];
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
            }
        }
        if (!rot) break;
    }
    int order[12];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (A[order[j] * n + order[j]] > A[order[i] * n + order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int i = 0; i < n; i++) { d[i] = A[order[i] * n + order[i]]; for (int k = 0; k < n; k++) Vt[i * n + k] = V[k * n + order[i]]; }
}

/* least squares min |A x - b| (m x n, n <= 5) through the eigen-decomposition of A'A (pseudo-inverse, tiny modes dropped) */
static void lsq_solve(const double* A, const double* b, int m, int n, double* x)
{
    double AtA[25], Atb[5], d[5], Vt[25];
    for (int i = 0; i < n; i++) {
        Atb[i] = 0; for (int k = 0; k < m; k++) Atb[i] += A[k * n + i] * b[k];
        for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < m; k++) s += A[k * n + i] * A[k * n + j]; AtA[i * n + j] = s; }
    }
    jacobi_eig_rr(AtA, n, d, Vt);
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int e = 0; e < n; e++) {
        if (d[e] <= d[0] * 1e-14 || d[e] <= 0) continue;
        double c = 0; for (int i = 0; i < n; i++) c += Vt[e * n + i] * Atb[i];
        c /= d[e];
        for (int i = 0; i < n; i++) x[i] += c * Vt[e * n + i];
    }
}

/* 3x3 SVD A = U diag(s) V' by one-sided Jacobi (Hestenes) on the columns */
static void svd3(const double* A, double* U, double* s, double* V)
{
    double B[9]; memcpy(B, A, sizeof(B));
    for (int i = 0; i < 9; i++) V[i] = (i % 4) == 0;
    for (int sweep = 0; sweep < 40; sweep++) {
        double rot = 0;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            double al = 0, be = 0, ga = 0;
            for (int k = 0; k < 3; k++) { al += B[k * 3 + p] * B[k * 3 + p]; be += B[k * 3 + q] * B[k * 3 + q]; ga += B[k * 3 + p] * B[k * 3 + q]; }
            if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            rot = 1;
            double zeta = (be - al) / (2.0 * ga);
            double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
            for (int k = 0; k < 3; k++) {
                double bp = B[k * 3 + p], bq = B[k * 3 + q]; B[k * 3 + p] = c * bp - sn * bq; B[k * 3 + q] = sn * bp + c * bq;
                double vp = V[k * 3 + p], vq = V[k * 3 + q]; V[k * 3 + p] = c * vp - sn * vq; V[k * 3 + q] = sn * vp + c * vq;
            }
        }
        if (rot == 0) break;
    }
    for (int j = 0; j < 3; j++) {
        double nrm = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
        s[j] = nrm;
        for (int k = 0; k < 3; k++) U[k * 3 + j] = nrm > 1e-300 ? B[k * 3 + j] / nrm : 0.0;
    }
    /* complete U to an orthonormal basis if a singular value vanished (rank-deficient samples) */
    for (int j = 0; j < 3; j++) if (s[j] <= 1e-300) {
        int a = (j + 1) % 3, b = (j + 2) % 3;
        U[0 * 3 + j] = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
        U[1 * 3 + j] = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
        U[2 * 3 + j] = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
    }
}

static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static int inv3d(const double* m, double* o)
{
    double d = det3(m);
    if (fabs(d) < 1e-300) return 0;
    double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 1;
}

/* ---------- Rodrigues ---------- */
void orc_rodrigues_vec2mat(const double* r, double* R)
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0; return; }
    double c = cos(th), s = sin(th), c1 = 1.0 - c, it = 1.0 / th;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

void orc_rodrigues_mat2vec(const double* R, double* r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double th = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        double k = th / nr; r[0] *= k; r[1] *= k; r[2] *= k;
        return;
    }
    double vth = 1.0 / (2 * s) * th;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}


/* the same two conversions with detmath.h's sin/cos/acos (used inside orc_solve_pnp_ransac only) */
static void det_rodrigues_vec2mat(const double* r, double* R)
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0; return; }
    double c, s; dm_sincos(th, &s, &c);
    double c1 = 1.0 - c, it = 1.0 / th;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

static void det_rodrigues_mat2vec(const double* R, double* r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double th = dm_acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        double k = th / nr; r[0] *= k; r[1] *= k; r[2] *= k;
        return;
    }
    double vth = 1.0 / (2 * s) * th;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* ---------- EPnP ---------- */
typedef struct { double fu, fv, uc, vc; } epnp_K;

/* optional trace of the intermediates of every RANSAC hypothesis (tests: first differing operation vs the HIP kernel's trace,
 * same layout as bh_debug_pnp_trace: 512 doubles per hypothesis).  Not thread-safe; tests set it around one call. */
static double* g_trace = 0;     /* current hypothesis' slot, or NULL */
#define OTR(off, val) do { if (g_trace) g_trace[(off)] = (val); } while (0)

static double epnp_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double epnp_dist2(const double* a, const double* b) { double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2]; return d0 * d0 + d1 * d1 + d2 * d2; }

/* control-point camera coordinates from betas and the 4 null vectors (rows 11,10,9,8 of ut) */
static void epnp_ccs(const double* betas, const double* ut, double ccs[4][3])
{
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
}

static double epnp_R_t(const double* pws, const double* us, const double* alphas, int n, epnp_K K, const double* ut,
                       const double* betas, double* R, double* t)
{
    double ccs[4][3];
    epnp_ccs(betas, ut, ccs);
    double pcs[3 * 64];
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++)
        pcs[3 * i + j] = alphas[4 * i] * ccs[0][j] + alphas[4 * i + 1] * ccs[1][j] + alphas[4 * i + 2] * ccs[2][j] + alphas[4 * i + 3] * ccs[3][j];
    if (pcs[2] < 0.0) {                                   /* solve_for_sign */
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    }
    /* estimate_R_and_t: Arun's rigid alignment pws -> pcs */
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double ABt[9] = {0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
        ABt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
    double U[9], s[3], V[9];
    svd3(ABt, U, s, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    if (det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int i = 0; i < 3; i++) t[i] = pc0[i] - epnp_dot(R + 3 * i, pw0);
    /* reprojection_error */
    double sum2 = 0;
    for (int i = 0; i < n; i++) {
        const double* pw = pws + 3 * i;
        double Xc = epnp_dot(R, pw) + t[0], Yc = epnp_dot(R + 3, pw) + t[1], inv_Zc = 1.0 / (epnp_dot(R + 6, pw) + t[2]);
        double ue = K.uc + K.fu * Xc * inv_Zc, ve = K.vc + K.fv * Yc * inv_Zc;
        double u = us[2 * i], v = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

static void epnp_gauss_newton(const double* L, const double* rho, double* betas)
{
    for (int it = 0; it < 5; it++) {
        double A[24], b[6], x[4];
        for (int i = 0; i < 6; i++) {
            const double* r = L + 10 * i;
            A[4 * i]     = 2 * r[0] * betas[0] +     r[1] * betas[1] +     r[3] * betas[2] +     r[6] * betas[3];
            A[4 * i + 1] =     r[1] * betas[0] + 2 * r[2] * betas[1] +     r[4] * betas[2] +     r[7] * betas[3];
            A[4 * i + 2] =     r[3] * betas[0] +     r[4] * betas[1] + 2 * r[5] * betas[2] +     r[8] * betas[3];
            A[4 * i + 3] =     r[6] * betas[0] +     r[7] * betas[1] +     r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] +
                             r[3] * betas[0] * betas[2] + r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                             r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] + r[8] * betas[2] * betas[3] +
                             r[9] * betas[3] * betas[3]);
        }
        ps_qr_solve_6x4(A, b, x);                       /* Householder QR, as the published EPnP code (epnp.cpp qr_solve) */
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

/* pws n x 3, us n x 2 (pixels), n <= 64.  Output R (row-major), t: world -> camera. */
static void epnp_solve(const double* pws, const double* us, int n, epnp_K K, double* Rbest, double* tbest)
{
    double cws[4][3] = {{0}};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    double PtP[9] = {0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
        PtP[3 * j + k] += (pws[3 * i + j] - cws[0][j]) * (pws[3 * i + k] - cws[0][k]);
    double dc[3], uct[9];
    jacobi_eig(PtP, 3, dc, uct);
    for (int i = 1; i < 4; i++) {
        double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0) / n);
        for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
    double CC[9], CCi[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    if (!inv3d(CC, CCi)) { for (int i = 0; i < 9; i++) Rbest[i] = (i % 4) == 0; tbest[0] = tbest[1] = tbest[2] = 0; return; }
    double alphas[4 * 64];
    for (int i = 0; i < n; i++) {
        double* a = alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = CCi[3 * j] * (pws[3 * i] - cws[0][0]) + CCi[3 * j + 1] * (pws[3 * i + 1] - cws[0][1]) + CCi[3 * j + 2] * (pws[3 * i + 2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    for (int i = 0; i < 12; i++) OTR(8 + i, cws[i / 3][i % 3]);
    if (n == 5) for (int i = 0; i < 20; i++) OTR(20 + i, alphas[i]);
    double MtM[144] = {0};
    for (int i = 0; i < n; i++) {
        double m1[12], m2[12];
        const double* a = alphas + 4 * i;
        for (int j = 0; j < 4; j++) {
            m1[3 * j] = a[j] * K.fu; m1[3 * j + 1] = 0.0;          m1[3 * j + 2] = a[j] * (K.uc - us[2 * i]);
            m2[3 * j] = 0.0;         m2[3 * j + 1] = a[j] * K.fv;  m2[3 * j + 2] = a[j] * (K.vc - us[2 * i + 1]);
        }
        for (int p = 0; p < 12; p++) for (int q = 0; q < 12; q++) MtM[12 * p + q] += m1[p] * m1[q] + m2[p] * m2[q];
    }
    for (int i = 0; i < 144; i++) OTR(40 + i, MtM[i]);
    double dm[12], ut[144];
    jacobi_eig_rr(MtM, 12, dm, ut);
    for (int i = 0; i < 12; i++) OTR(184 + i, dm[i]);
    for (int i = 0; i < 144; i++) OTR(196 + i, ut[i]);
    /* L_6x10 and rho */
    double L[60], rho[6], dv[4][6][3];
    const double* v[4] = { ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8 };
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
            b++; if (b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* r = L + 10 * i;
        r[0] = epnp_dot(dv[0][i], dv[0][i]); r[1] = 2.0 * epnp_dot(dv[0][i], dv[1][i]); r[2] = epnp_dot(dv[1][i], dv[1][i]);
        r[3] = 2.0 * epnp_dot(dv[0][i], dv[2][i]); r[4] = 2.0 * epnp_dot(dv[1][i], dv[2][i]); r[5] = epnp_dot(dv[2][i], dv[2][i]);
        r[6] = 2.0 * epnp_dot(dv[0][i], dv[3][i]); r[7] = 2.0 * epnp_dot(dv[1][i], dv[3][i]); r[8] = 2.0 * epnp_dot(dv[2][i], dv[3][i]);
        r[9] = epnp_dot(dv[3][i], dv[3][i]);
    }
    rho[0] = epnp_dist2(cws[0], cws[1]); rho[1] = epnp_dist2(cws[0], cws[2]); rho[2] = epnp_dist2(cws[0], cws[3]);
    rho[3] = epnp_dist2(cws[1], cws[2]); rho[4] = epnp_dist2(cws[1], cws[3]); rho[5] = epnp_dist2(cws[2], cws[3]);
    for (int i = 0; i < 60; i++) OTR(340 + i, L[i]);
    for (int i = 0; i < 6; i++) OTR(400 + i, rho[i]);
    double best = DBL_MAX;
    for (int N = 1; N <= 3; N++) {
        double betas[4] = {0, 0, 0, 0}, A[30], x[5];
        if (N == 1) {            /* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]; approx_1 uses B11 B12 B13 B14 */
            for (int i = 0; i < 6; i++) { A[4 * i] = L[10 * i]; A[4 * i + 1] = L[10 * i + 1]; A[4 * i + 2] = L[10 * i + 3]; A[4 * i + 3] = L[10 * i + 6]; }
            lsq_solve(A, rho, 6, 4, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
        } else if (N == 2) {     /* approx_2 uses B11 B12 B22 */
            for (int i = 0; i < 6; i++) { A[3 * i] = L[10 * i]; A[3 * i + 1] = L[10 * i + 1]; A[3 * i + 2] = L[10 * i + 2]; }
            lsq_solve(A, rho, 6, 3, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
        } else {                 /* approx_3 uses B11 B12 B22 B13 B23 */
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) A[5 * i + k] = L[10 * i + k];
            lsq_solve(A, rho, 6, 5, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
            betas[2] = x[3] / betas[0];
        }
        for (int i = 0; i < 4; i++) OTR(406 + 21 * (N - 1) + i, betas[i]);
        epnp_gauss_newton(L, rho, betas);
        double R[9], t[3];
        double err = epnp_R_t(pws, us, alphas, n, K, ut, betas, R, t);
        for (int i = 0; i < 4; i++) OTR(406 + 21 * (N - 1) + 4 + i, betas[i]);
        for (int i = 0; i < 9; i++) OTR(406 + 21 * (N - 1) + 8 + i, R[i]);
        for (int i = 0; i < 3; i++) OTR(406 + 21 * (N - 1) + 17 + i, t[i]);
        OTR(406 + 21 * (N - 1) + 20, err);
        if (err < best || N == 1) { best = err; memcpy(Rbest, R, sizeof(R)); memcpy(tbest, t, sizeof(t)); }
    }
}

/* ---------- refinement: Levenberg-Marquardt on the reprojection error over the selected points ---------- */
static void pnp_refine(const double* obj, const double* img, const uint8_t* sel, int n, epnp_K K, double* R, double* t)
{
    double lambda = 1e-3;
    double cost = -1;
    for (int it = 0; it < 20; it++) {                       /* CvLevMarq: at most 20 iterations */
        double H[36] = {0}, g[6] = {0}, c0 = 0;
        for (int i = 0; i < n; i++) if (sel[i]) {
            const double* X = obj + 3 * i;
            double rx[3] = { epnp_dot(R, X), epnp_dot(R + 3, X), epnp_dot(R + 6, X) };
            double x = rx[0] + t[0], y = rx[1] + t[1], z = rx[2] + t[2], iz = 1.0 / z;
            double e0 = K.fu * x * iz + K.uc - img[2 * i], e1 = K.fv * y * iz + K.vc - img[2 * i + 1];
            double du[3] = { K.fu * iz, 0, -K.fu * x * iz * iz }, dv[3] = { 0, K.fv * iz, -K.fv * y * iz * iz };
            /* d(Xc)/d(omega) = -[RX]x = [[0,a2,-a1],[-a2,0,a0],[a1,-a0,0]], a = RX (left perturbation R <- exp(w) R); d(Xc)/dt = I */
            double J0[6], J1[6];
            J0[0] = du[1] * (-rx[2]) + du[2] * rx[1];
            J0[1] = du[0] * rx[2] + du[2] * (-rx[0]);
            J0[2] = du[0] * (-rx[1]) + du[1] * rx[0];
            J0[3] = du[0]; J0[4] = du[1]; J0[5] = du[2];
            J1[0] = dv[1] * (-rx[2]) + dv[2] * rx[1];
            J1[1] = dv[0] * rx[2] + dv[2] * (-rx[0]);
            J1[2] = dv[0] * (-rx[1]) + dv[1] * rx[0];
            J1[3] = dv[0]; J1[4] = dv[1]; J1[5] = dv[2];
            for (int a = 0; a < 6; a++) { g[a] += J0[a] * e0 + J1[a] * e1; for (int b = 0; b < 6; b++) H[6 * a + b] += J0[a] * J0[b] + J1[a] * J1[b]; }
            c0 += e0 * e0 + e1 * e1;
        }
        if (cost < 0) cost = c0;
        int improved = 0;
        for (int trial = 0; trial < 10 && !improved; trial++) {
            double A[36], ng[6], dx[6];
            memcpy(A, H, sizeof(A));
            for (int a = 0; a < 6; a++) { A[7 * a] += lambda * (H[7 * a] > 1e-12 ? H[7 * a] : 1e-12); ng[a] = -g[a]; }
            ps_chol6_solve(A, ng, dx);                      /* damped normal equations: SPD, Cholesky (a failed pivot gives dx = 0: the trial is rejected) */
            double dR[9], Rn[9], tn[3] = { t[0] + dx[3], t[1] + dx[4], t[2] + dx[5] };
            det_rodrigues_vec2mat(dx, dR);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rn[3 * a + b] = dR[3 * a] * R[b] + dR[3 * a + 1] * R[3 + b] + dR[3 * a + 2] * R[6 + b];
            double c1 = 0;
            for (int i = 0; i < n; i++) if (sel[i]) {
                const double* X = obj + 3 * i;
                double x = epnp_dot(Rn, X) + tn[0], y = epnp_dot(Rn + 3, X) + tn[1], z = epnp_dot(Rn + 6, X) + tn[2];
                double e0 = K.fu * x / z + K.uc - img[2 * i], e1 = K.fv * y / z + K.vc - img[2 * i + 1];
                c1 += e0 * e0 + e1 * e1;
            }
            if (c1 < c0 && isfinite(c1)) { memcpy(R, Rn, sizeof(Rn)); memcpy(t, tn, sizeof(tn)); lambda *= 0.2; improved = 1; cost = c1;
                double step = 0; for (int a = 0; a < 6; a++) step += dx[a] * dx[a]; if (step < 1e-18) it = 1000; }          /* |step| < 1e-9: OpenCV stops at a relative parameter change of FLT_EPSILON */
            else lambda *= 5.0;
        }
        if (!improved) break;
    }
}

/* ---------- RANSAC ---------- */
static unsigned rng_next(uint64_t* st) { *st = (uint64_t)(unsigned)(*st) * 4164903690U + (unsigned)(*st >> 32); return (unsigned)(*st); }

static int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : p > 1 ? 1 : p; ep = ep < 0 ? 0 : ep > 1 ? 1 : ep;
    double num = 1. - p; if (num < DBL_MIN) num = DBL_MIN;
    (void)model_points;                                   /* 5: the only minimal-set size of this path */
    double denom = 1. - dm_pow5(1. - ep);
    if (denom < DBL_MIN) return 0;
    num = dm_log(num); denom = dm_log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

static double* g_trace_all = 0; static int g_trace_hyp = 0;
void orc_pnp_set_trace(double* trace, int max_hyp) { g_trace_all = trace; g_trace_hyp = max_hyp; g_trace = 0; }

/* returns 1 on success.  K = {fx, fy, cx, cy}.  inlier_idx capacity n. */
int orc_solve_pnp_ransac(const double* obj_in, const double* img_in, int n, const double* Kin, int iterations_count,
                         double reproj_error, double confidence, double* rvec, double* tvec, int* inlier_idx, int* n_inliers)
{
    *n_inliers = 0;
    if (n < 4) return 0;
    epnp_K K = { Kin[0], Kin[1], Kin[2], Kin[3] };
    double* obj = malloc(sizeof(double) * 3 * n); double* img = malloc(sizeof(double) * 2 * n);
    for (int i = 0; i < 3 * n; i++) obj[i] = (double)(float)obj_in[i];          /* CV_64F -> CV_32F */
    for (int i = 0; i < 2 * n; i++) img[i] = (double)(float)img_in[i];
    uint8_t* mask = calloc(n, 1); uint8_t* best_mask = calloc(n, 1);
    double bestR[9], bestt[3];
    int max_good = 0, niters = iterations_count > 1 ? iterations_count : 1, ok = 0;
    const float thr2 = (float)(reproj_error * reproj_error);
    uint64_t st = (uint64_t)-1;
    if (n == 4) {
        /* OpenCV: npoints == 4 -> model_points = 4, SOLVEPNP_P3P; count == modelPoints -> ONE hypothesis from all four points, accepted only
         * when all four are inliers (goodCount > max(maxGoodCount, modelPoints - 1)).  p3p_small.h: pose of the first three, picked by the fourth. */
        if (p3_solve4(obj, img, K.fu, K.fv, K.uc, K.vc, bestR, bestt)) {
            int good = 0;
            for (int i = 0; i < 4; i++) {
                const double* X = obj + 3 * i;
                double x = epnp_dot(bestR, X) + bestt[0], y = epnp_dot(bestR + 3, X) + bestt[1], z = epnp_dot(bestR + 6, X) + bestt[2];
                float pu = (float)(K.fu * x / z + K.uc), pv = (float)(K.fv * y / z + K.vc);
                float du = (float)img[2 * i] - pu, dv = (float)img[2 * i + 1] - pv;
                best_mask[i] = du * du + dv * dv <= thr2;
                good += best_mask[i];
            }
            if (good == 4) max_good = 4;
        }
    } else if (n == 5) {
        epnp_solve(obj, img, 5, K, bestR, bestt);
        memset(best_mask, 1, n); max_good = 5;
    } else {
        for (int iter = 0; iter < niters; iter++) {
            int idx[5];
            for (int i = 0; i < 5; ) {
                int j;
                for (;;) { idx[i] = (int)(rng_next(&st) % (unsigned)n); for (j = 0; j < i; j++) if (idx[i] == idx[j]) break; if (j == i) break; }
                i++;
            }
            double so[15], si[10], R[9], t[3];
            for (int i = 0; i < 5; i++) { memcpy(so + 3 * i, obj + 3 * idx[i], 24); memcpy(si + 2 * i, img + 2 * idx[i], 16); }
            g_trace = g_trace_all && iter < g_trace_hyp ? g_trace_all + 512 * (size_t)iter : 0;
            epnp_solve(so, si, 5, K, R, t);
            int good = 0;
            for (int i = 0; i < n; i++) {
                const double* X = obj + 3 * i;
                double x = epnp_dot(R, X) + t[0], y = epnp_dot(R + 3, X) + t[1], z = epnp_dot(R + 6, X) + t[2];
                float pu = (float)(K.fu * x / z + K.uc), pv = (float)(K.fv * y / z + K.vc);
                float du = (float)img[2 * i] - pu, dv = (float)img[2 * i + 1] - pv;
                float e = du * du + dv * dv;
                mask[i] = e <= thr2;
                good += mask[i];
            }
            if (g_trace) { g_trace[0] = iter; for (int i = 0; i < 5; i++) g_trace[1 + i] = idx[i]; for (int i = 0; i < 9; i++) g_trace[469 + i] = R[i]; for (int i = 0; i < 3; i++) g_trace[478 + i] = t[i]; g_trace[481] = good; g_trace = 0; }
            if (good > (max_good > 4 ? max_good : 4)) {
                memcpy(best_mask, mask, n); memcpy(bestR, R, sizeof(R)); memcpy(bestt, t, sizeof(t));
                max_good = good;
                niters = ransac_update_iters(confidence, (double)(n - good) / n, 5, niters);
            }
        }
    }
    if (max_good > 0) {
        /* final pose on the inliers, from the original float32-rounded points widened to double (opoints.convertTo(CV_64F)) */
        pnp_refine(obj, img, best_mask, n, K, bestR, bestt);
        det_rodrigues_mat2vec(bestR, rvec);
        memcpy(tvec, bestt, sizeof(bestt));
        int k = 0;
        for (int i = 0; i < n; i++) if (best_mask[i]) inlier_idx[k++] = i;
        *n_inliers = k; ok = 1;
    }
    free(obj); free(img); free(mask); free(best_mask);
    return ok;
}

/* test access to detmath.h: out[0..4][n] = sin, cos, acos(clamped), log|x|, pow5 of x[i] */
void orc_detmath_eval(const double* x, int n, double* out)
{
    for (int i = 0; i < n; i++) {
        double s, c; dm_sincos(x[i], &s, &c);
        out[i] = s; out[n + i] = c;
        out[2 * n + i] = dm_acos(x[i] > 1 ? 1 : x[i] < -1 ? -1 : x[i]);
        out[3 * n + i] = x[i] != 0 ? dm_log(fabs(x[i])) : 0.0;
        out[4 * n + i] = dm_pow5(x[i]);
    }
}

/* Test hooks: the two small dense solves of pnp_small.h, callable on their own so that tests/test_oracle_cpu.py can check them
   against numpy.linalg (the header's text is shared with the kernel, so "HIP == oracle" says nothing about its correctness). */
void orc_ps_qr_solve_6x4(const double* A, const double* b, double* x)
{
    double Ac[24], bc[6];
    memcpy(Ac, A, sizeof(Ac)); memcpy(bc, b, sizeof(bc));
    ps_qr_solve_6x4(Ac, bc, x);
}
int orc_ps_chol6_solve(const double* A, const double* b, double* x) { return ps_chol6_solve(A, b, x); }

/* test access to p3p_small.h: roots of a quartic; the up-to-four poses of three correspondences; the four-point model */
int orc_p3_roots(const double* A, double* v) { return p3_roots(A, v); }
int orc_p3_solve3(const double* X, const double* f, double* Rs, double* ts) { return p3_solve3(X, f, Rs, ts); }
int orc_p3_solve4(const double* obj, const double* img, const double* K, double* R, double* t) { return p3_solve4(obj, img, K[0], K[1], K[2], K[3], R, t); }
