// pnp.hip — solvePnPRansac / Rodrigues on gfx950.
//
// Replaces cv2.solvePnPRansac(obj, img, K, dist=0, flags=SOLVEPNP_ITERATIVE) and cv2.Rodrigues as called at
// experiments/pnp_two_way_tracking.py:57-61,68-69 (and the dead branch slam/tracking.py:68-71), batched over B problems.
//
// One 256-lane workgroup per problem.  OpenCV's RANSAC loop is sequential (each improvement shrinks the iteration budget),
// but its sample stream does not depend on the models, so the loop is unrolled in space:
//   phase 0  lane 0 replays cv::RNG((uint64)-1) and writes every 5-point subset the sequential loop could ever draw;
//   phase 1  one lane per hypothesis solves EPnP on its subset (control points, 12x12 M'M null space by cyclic Jacobi,
//            betas for N = 1..3 + Gauss-Newton, Arun alignment) — f64, per-lane scratch;
//   phase 2  all lanes score all hypotheses against all points (float32 squared reprojection error, LDS integer counters);
//   phase 3  lane 0 replays the sequential "better than the best so far -> shrink niters" rule and names the winner;
//   phase 4  Levenberg-Marquardt on the winner's inliers (block-reduced 6x6 normal equations), ordered inlier compaction.
// Inputs are rounded to float32 first, as solvePnPRansac does (CV_64F -> CV_32F).
#include "bh_internal.h"
#include <math.h>
#include <float.h>
#include "detmath.h"
#include "pnp_small.h"
#include "p3p_small.h"

#define PNP_T 64            // one wave per problem: RANSAC iterations in sequence (as OpenCV runs them), each one wave-parallel
#define PNP_TRACE_HYP 32       // test hook: hypotheses traced per problem
#define PNP_TRACE_STRIDE 512   // doubles per traced hypothesis

struct pnp_K { double fu, fv, uc, vc; };

// ------------------------------------------------------------------------------------------------ small linear algebra
template <int N>
__device__ void pj_eig(const double* Ain, double* d, double* Vt)      // symmetric cyclic Jacobi; eigenvalues descending, vectors in rows
{
    // Every index below is a compile-time constant after unrolling, so A and V live in registers.  (With run-time indices the
    // two arrays sit in scratch memory and one 6x6 decomposition costs thousands of dependent scratch round trips.)
    double A[N * N], V[N * N];
#pragma unroll
    for (int i = 0; i < N * N; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) V[i * N + j] = i == j;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
#pragma unroll
        for (int p = 0; p < N; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) off += A[p * N + q] * A[p * N + q];
        if (off < 1e-300) break;
        bool rot = false;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                // standard threshold of the cyclic method: a rotation this small cannot change a double any more
                if (fabs(apq) < 1e-300 || fabs(apq) <= 1e-17 * sqrt(fabs(A[p * N + p] * A[q * N + q]))) continue;
                rot = true;
                const double tau = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
                for (int k = 0; k < N; k++) { const double akp = A[k * N + p], akq = A[k * N + q]; A[k * N + p] = c * akp - s * akq; A[k * N + q] = s * akp + c * akq; }
#pragma unroll
                for (int k = 0; k < N; k++) { const double apk = A[p * N + k], aqk = A[q * N + k]; A[p * N + k] = c * apk - s * aqk; A[q * N + k] = s * apk + c * aqk; }
#pragma unroll
                for (int k = 0; k < N; k++) { const double vkp = V[k * N + p], vkq = V[k * N + q]; V[k * N + p] = c * vkp - s * vkq; V[k * N + q] = s * vkp + c * vkq; }
            }
        if (!rot) break;
    }
    // the reference's exchange sort (descending, strict '>'), replayed on (value, column) pairs at compile-time positions
    double dv[N]; int oi[N];
#pragma unroll
    for (int i = 0; i < N; i++) { dv[i] = A[i * N + i]; oi[i] = i; }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            const bool sw = dv[j] > dv[i];
            const double tv = dv[i]; const int ti = oi[i];
            dv[i] = sw ? dv[j] : dv[i]; oi[i] = sw ? oi[j] : oi[i];
            dv[j] = sw ? tv : dv[j]; oi[j] = sw ? ti : oi[j];
        }
#pragma unroll
    for (int i = 0; i < N; i++) {
        d[i] = dv[i];
#pragma unroll
        for (int c = 0; c < N; c++)
            if (oi[i] == c) {
#pragma unroll
                for (int k = 0; k < N; k++) Vt[i * N + k] = V[k * N + c];
            }
    }
}

// Symmetric Jacobi eigen-decomposition of an N x N matrix (N <= 12) held in LDS, executed by one wave, with the rotations of a sweep
// arranged in ROUNDS of disjoint pairs (round-robin ordering; the CPU restatement's jacobi_eig_rr fixes the same operation order):
// per round lane i < N/2 derives the rotation of pair i from the matrix as it stands (a chain of 3 divisions and 2 square roots -
// once per round instead of once per rotation), then all column rotations of the round are applied (lane = pair x row, A and V),
// then all row rotations.  Eigenvalues descending in sd, eigenvectors in the rows of sVt.  srot: 6 x {c, s} + flags (>= 24 doubles).
// Every lane must call it with the same arguments (one-wave workgroup: __syncthreads is a wave barrier).
template <int N>
__device__ void pj_eig_wave(double* sA, double* sV, double* sd, double* sVt, double* srot)
{
    constexpr int M = (N + 1) & ~1, NP = M / 2;
    const int lane = threadIdx.x & 63;
    int* sact = reinterpret_cast<int*>(srot + 12);            // act[6], P[6], Q[6]
    for (int i = lane; i < N * N; i += 64) sV[i] = (i / N) == (i % N) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int p = 0; p < N; p++) for (int q = p + 1; q < N; q++) off += sA[p * N + q] * sA[p * N + q];
        if (off < 1e-300) break;
        bool rot = false;
        for (int r = 0; r < M - 1; r++) {
            __syncthreads();
            if (lane < NP) {
                const int i = lane;
                const int a = i == 0 ? M - 1 : (r + i) % (M - 1), b = i == 0 ? r : (r - i + (M - 1)) % (M - 1);
                const int p = a < b ? a : b, q = a < b ? b : a;
                double c = 1.0, s = 0.0; int act = 0;
                if (q < N) {
                    const double apq = sA[p * N + q];
                    if (!(fabs(apq) < 1e-300 || fabs(apq) <= 1e-17 * sqrt(fabs(sA[p * N + p] * sA[q * N + q])))) {
                        const double tau = (sA[q * N + q] - sA[p * N + p]) / (2.0 * apq);
                        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t); s = t * c; act = 1;
                    }
                }
                srot[2 * i] = c; srot[2 * i + 1] = s; sact[i] = act; sact[6 + i] = p; sact[12 + i] = q;
            }
            __syncthreads();
            // column rotations: work item w = pair * N + k for A, NP * N + the same for V
            for (int w = lane; w < 2 * NP * N; w += 64) {
                const bool isV = w >= NP * N;
                const int ww = isV ? w - NP * N : w;
                const int i = ww / N, k = ww - i * N;
                if (!sact[i]) continue;
                const int p = sact[6 + i], q = sact[12 + i];
                const double c = srot[2 * i], s = srot[2 * i + 1];
                double* Mx = isV ? sV : sA;
                const double xp = Mx[k * N + p], xq = Mx[k * N + q];
                Mx[k * N + p] = c * xp - s * xq; Mx[k * N + q] = s * xp + c * xq;
            }
            __syncthreads();
            for (int w = lane; w < NP * N; w += 64) {
                const int i = w / N, k = w - i * N;
                if (!sact[i]) continue;
                const int p = sact[6 + i], q = sact[12 + i];
                const double c = srot[2 * i], s = srot[2 * i + 1];
                const double apk = sA[p * N + k], aqk = sA[q * N + k];
                sA[p * N + k] = c * apk - s * aqk; sA[q * N + k] = s * apk + c * aqk;
            }
            for (int i = 0; i < NP; i++) rot = rot | (sact[i] != 0);
        }
        __syncthreads();
        if (!rot) break;
    }
    __syncthreads();
    int order[N];
#pragma unroll
    for (int i = 0; i < N; i++) order[i] = i;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = i + 1; j < N; j++) {                      // the reference's exchange sort (descending, strict '>'), on compile-time positions
            const bool sw = sA[order[j] * N + order[j]] > sA[order[i] * N + order[i]];
            const int oi = order[i], oj = order[j];
            order[i] = sw ? oj : oi; order[j] = sw ? oi : oj;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (lane < N) sVt[i * N + lane] = sV[lane * N + order[i]];
        if (lane == 0) sd[i] = sA[order[i] * N + order[i]];
    }
    __syncthreads();
}

// min |A x - b| (m x N, N <= 6) through the eigen-decomposition of A'A (pseudo-inverse, tiny modes dropped).  A, b and the result are
// wave-uniform registers; A'A, its eigen-decomposition and nothing else live in the LDS workspace ws (>= 3 N^2 + N doubles).
template <int N>
__device__ void p_lsq(const double* A, const double* b, int m, double* x, double* ws)
{
    double* sA = ws; double* sV = ws + N * N; double* sVt = ws + 2 * N * N; double* sd = ws + 3 * N * N;
    double Atb[N];
    __syncthreads();                                            // earlier readers of ws are done
#pragma unroll
    for (int i = 0; i < N; i++) {
        Atb[i] = 0; for (int k = 0; k < m; k++) Atb[i] += A[k * N + i] * b[k];
#pragma unroll
        for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < m; k++) s += A[k * N + i] * A[k * N + j]; sA[i * N + j] = s; }      // every lane writes the same value
    }
    __syncthreads();
    pj_eig_wave<N>(sA, sV, sd, sVt, ws + 3 * 36 + 6 + 66 + 20);
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = 0;
    const double d0 = sd[0];
#pragma unroll
    for (int e = 0; e < N; e++) {
        const double de = sd[e];
        if (de <= d0 * 1e-14 || de <= 0) continue;
        double c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) c += sVt[e * N + i] * Atb[i];
        c /= de;
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += c * sVt[e * N + i];
    }
}

__device__ void p_svd3(const double* A, double* U, double* s, double* V)       // one-sided Jacobi
{
    double B[9];
    for (int i = 0; i < 9; i++) { B[i] = A[i]; V[i] = (i % 4) == 0; }
    for (int sweep = 0; sweep < 40; sweep++) {
        int rot = 0;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            double al = 0, be = 0, ga = 0;
            for (int k = 0; k < 3; k++) { al += B[k * 3 + p] * B[k * 3 + p]; be += B[k * 3 + q] * B[k * 3 + q]; ga += B[k * 3 + p] * B[k * 3 + q]; }
            if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
            rot = 1;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
            for (int k = 0; k < 3; k++) {
                const double bp = B[k * 3 + p], bq = B[k * 3 + q]; B[k * 3 + p] = c * bp - sn * bq; B[k * 3 + q] = sn * bp + c * bq;
                const double vp = V[k * 3 + p], vq = V[k * 3 + q]; V[k * 3 + p] = c * vp - sn * vq; V[k * 3 + q] = sn * vp + c * vq;
            }
        }
        if (!rot) break;
    }
    for (int j = 0; j < 3; j++) {
        const double nrm = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
        s[j] = nrm;
        for (int k = 0; k < 3; k++) U[k * 3 + j] = nrm > 1e-300 ? B[k * 3 + j] / nrm : 0.0;
    }
    for (int j = 0; j < 3; j++) if (s[j] <= 1e-300) {
        const int a = (j + 1) % 3, b = (j + 2) % 3;
        U[0 * 3 + j] = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
        U[1 * 3 + j] = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
        U[2 * 3 + j] = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
    }
}

__host__ __device__ inline double p_det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ bool p_inv3(const double* m, double* o)
{
    const double d = p_det3(m);
    if (fabs(d) < 1e-300) return false;
    const double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}
__host__ __device__ inline double p_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline double p_dist2(const double* a, const double* b) { const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2]; return d0 * d0 + d1 * d1 + d2 * d2; }

// ------------------------------------------------------------------------------------------------ Rodrigues (host + device)
__host__ __device__ void p_rodrigues_v2m(const double* r, double* R)
{
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0; return; }
    const double c = cos(th), s = sin(th), c1 = 1.0 - c, it = 1.0 / th;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

__host__ __device__ void p_rodrigues_m2v(const double* R, double* r)
{
    const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double th = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double k = th / nr; r[0] *= k; r[1] *= k; r[2] *= k;
        return;
    }
    const double vth = 1.0 / (2 * s) * th;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}


// The same two conversions for the device code of solvePnPRansac, with detmath.h's sin/cos/acos: the CPU restatement uses the same
// statements, so the refinement and the final rvec agree with it bit for bit (bh_rodrigues above stays on libm, like cv2.Rodrigues).
__device__ void pd_rodrigues_v2m(const double* r, double* R)
{
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0; return; }
    double c, s; dm_sincos(th, &s, &c);
    const double c1 = 1.0 - c, it = 1.0 / th;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

__device__ void pd_rodrigues_m2v(const double* R, double* r)
{
    const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double th = dm_acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double k = th / nr; r[0] *= k; r[1] *= k; r[2] *= k;
        return;
    }
    const double vth = 1.0 / (2 * s) * th;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

extern "C" int bh_rodrigues(const double* in, int in_len, double* out)
{
    if (!in || !out) return BH_ERR_INVALID;
    if (in_len == 3) { p_rodrigues_v2m(in, out); return BH_OK; }
    if (in_len == 9) { p_rodrigues_m2v(in, out); return BH_OK; }
    return BH_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------ EPnP on a 5-point sample
#define NS 5

__device__ double p_epnp_R_t(const double* pws, const double* us, const double* alphas, pnp_K K, const double* ut,
                             const double* betas, double* R, double* t)
{
    double ccs[4][3];
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
    double pcs[3 * NS];
    for (int i = 0; i < NS; i++) for (int j = 0; j < 3; j++)
        pcs[3 * i + j] = alphas[4 * i] * ccs[0][j] + alphas[4 * i + 1] * ccs[1][j] + alphas[4 * i + 2] * ccs[2][j] + alphas[4 * i + 3] * ccs[3][j];
    if (pcs[2] < 0.0) for (int i = 0; i < 3 * NS; i++) pcs[i] = -pcs[i];
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < NS; i++) for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= NS; pw0[j] /= NS; }
    double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NS; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
        ABt[3 * j + k] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + k] - pw0[k]);
    double U[9], s[3], V[9];
    p_svd3(ABt, U, s, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    if (p_det3(R) < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    for (int i = 0; i < 3; i++) t[i] = pc0[i] - p_dot(R + 3 * i, pw0);
    double sum2 = 0;
    for (int i = 0; i < NS; i++) {
        const double* pw = pws + 3 * i;
        const double Xc = p_dot(R, pw) + t[0], Yc = p_dot(R + 3, pw) + t[1], inv_Zc = 1.0 / (p_dot(R + 6, pw) + t[2]);
        const double ue = K.uc + K.fu * Xc * inv_Zc, ve = K.vc + K.fv * Yc * inv_Zc;
        const double u = us[2 * i], v = us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / NS;
}

__device__ void p_gauss_newton(const double* L, const double* rho, double* betas, double* ws2)
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
        ps_qr_solve_6x4(A, b, x);                       // Householder QR in registers (wave-uniform), as the published EPnP code
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// Called by every lane of the wave with the same arguments (wave-uniform); ws = LDS workspace of 3*144 + 12 doubles for the 12x12
// problem (its eigenvectors stay live to the end), ws2 = 3*36 + 6 doubles for the small least-squares solves + 66 for L and rho + 20 for the barycentric coordinates
// trace (test hook, may be NULL): lane 0 writes the intermediates of this hypothesis (layout: tools/pnp_trace.py)
#define PTR(off, val) do { if (trace && (threadIdx.x & 63) == 0) trace[(off)] = (val); } while (0)
__device__ void p_epnp5(const double* pws, const double* us, pnp_K K, double* Rbest, double* tbest, double* ws, double* ws2, double* trace)
{
    double cws[4][3];
    for (int i = 0; i < 4; i++) cws[i][0] = cws[i][1] = cws[i][2] = 0;
    for (int i = 0; i < NS; i++) for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= NS;
    double PtP[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NS; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++)
        PtP[3 * j + k] += (pws[3 * i + j] - cws[0][j]) * (pws[3 * i + k] - cws[0][k]);
    double dc[3], uct[9];
    pj_eig<3>(PtP, dc, uct);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0) / NS);
        for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
    double CC[9], CCi[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    if (!p_inv3(CC, CCi)) { for (int i = 0; i < 9; i++) Rbest[i] = (i % 4) == 0; tbest[0] = tbest[1] = tbest[2] = 0; return; }
    double* alphas = ws2 + 3 * 36 + 6 + 66;          // [4 * NS] in LDS as well
    for (int i = 0; i < NS; i++) {
        double* a = alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = CCi[3 * j] * (pws[3 * i] - cws[0][0]) + CCi[3 * j + 1] * (pws[3 * i + 1] - cws[0][1]) + CCi[3 * j + 2] * (pws[3 * i + 2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    for (int i = 0; i < 12; i++) PTR(8 + i, cws[i / 3][i % 3]);
    for (int i = 0; i < 20; i++) PTR(20 + i, alphas[i]);
    // MtM (12x12) straight into LDS: lane e accumulates element e over the 5 points in point order (the per-lane loop's order)
    double* sA = ws; double* sV = ws + 144; double* ut = ws + 288; double* dm = ws + 432;
    {
        // the two rows each point contributes to M go through LDS (the eigenvector scratch sV is free until the decomposition
        // starts): lane e then reads its operands by index - a lane-indexed register array would live in scratch memory
        double* m1 = sV; double* m2 = sV + 12 * NS;
        __syncthreads();                              // the previous hypothesis' readers of ws are done
        for (int i = 0; i < NS; i++) {
            const double* a = alphas + 4 * i;
            for (int j = 0; j < 4; j++) {
                m1[12 * i + 3 * j] = a[j] * K.fu; m1[12 * i + 3 * j + 1] = 0.0;          m1[12 * i + 3 * j + 2] = a[j] * (K.uc - us[2 * i]);
                m2[12 * i + 3 * j] = 0.0;         m2[12 * i + 3 * j + 1] = a[j] * K.fv;  m2[12 * i + 3 * j + 2] = a[j] * (K.vc - us[2 * i + 1]);
            }
        }
        __syncthreads();
        double accs[3];
        for (int u = 0; u < 3; u++) {
            const int e = (int)(threadIdx.x & 63) + 64 * u;
            const int pp = min(e, 143) / 12, qq = min(e, 143) % 12;
            double acc = 0;
            for (int i = 0; i < NS; i++) acc += m1[12 * i + pp] * m1[12 * i + qq] + m2[12 * i + pp] * m2[12 * i + qq];
            accs[u] = acc;
        }
        for (int u = 0; u < 3; u++) {
            const int e = (int)(threadIdx.x & 63) + 64 * u;
            if (e < 144) { sA[e] = accs[u]; if (trace) trace[40 + e] = accs[u]; }
        }
        __syncthreads();
    }
    pj_eig_wave<12>(sA, sV, dm, ut, ws2 + 3 * 36 + 6 + 66 + 20);
    for (int i = 0; i < 12; i++) PTR(184 + i, dm[i]);
    for (int i = 0; i < 144; i++) PTR(196 + i, ut[i]);
    double* L = ws2 + 3 * 36 + 6; double* rho = L + 60;      // in LDS: read many times by the three beta solves and Gauss-Newton
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            for (int k = 0; k < 3; k++) dv[i][j][k] = v[3 * a + k] - v[3 * b + k];
            b++; if (b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* r = L + 10 * i;
        r[0] = p_dot(dv[0][i], dv[0][i]); r[1] = 2.0 * p_dot(dv[0][i], dv[1][i]); r[2] = p_dot(dv[1][i], dv[1][i]);
        r[3] = 2.0 * p_dot(dv[0][i], dv[2][i]); r[4] = 2.0 * p_dot(dv[1][i], dv[2][i]); r[5] = p_dot(dv[2][i], dv[2][i]);
        r[6] = 2.0 * p_dot(dv[0][i], dv[3][i]); r[7] = 2.0 * p_dot(dv[1][i], dv[3][i]); r[8] = 2.0 * p_dot(dv[2][i], dv[3][i]);
        r[9] = p_dot(dv[3][i], dv[3][i]);
    }
    rho[0] = p_dist2(cws[0], cws[1]); rho[1] = p_dist2(cws[0], cws[2]); rho[2] = p_dist2(cws[0], cws[3]);
    rho[3] = p_dist2(cws[1], cws[2]); rho[4] = p_dist2(cws[1], cws[3]); rho[5] = p_dist2(cws[2], cws[3]);
    for (int i = 0; i < 60; i++) PTR(340 + i, L[i]);
    for (int i = 0; i < 6; i++) PTR(400 + i, rho[i]);
    // The three beta approximations are three small least-squares problems solved one after the other by the whole wave; what follows
    // each of them - five Gauss-Newton steps and the (R, t) estimate with its 3 x 3 Jacobi SVD, ~50 us of wave-uniform f64 arithmetic
    // per variant - is the same code on different numbers, so lanes 0, 1, 2 run the three variants at once (every lane executes the
    // same instruction stream anyway) and the winner is picked by the reference's sequential rule afterwards.  Bit for bit the
    // same values as running them in turn.
    double b3[3][4];
#pragma unroll
    for (int N = 1; N <= 3; N++) {
        double betas[4] = {0, 0, 0, 0}, A[30], x[5];
        if (N == 1) {
            for (int i = 0; i < 6; i++) { A[4 * i] = L[10 * i]; A[4 * i + 1] = L[10 * i + 1]; A[4 * i + 2] = L[10 * i + 3]; A[4 * i + 3] = L[10 * i + 6]; }
            p_lsq<4>(A, rho, 6, x, ws2);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
        } else if (N == 2) {
            for (int i = 0; i < 6; i++) { A[3 * i] = L[10 * i]; A[3 * i + 1] = L[10 * i + 1]; A[3 * i + 2] = L[10 * i + 2]; }
            p_lsq<3>(A, rho, 6, x, ws2);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
        } else {
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) A[5 * i + k] = L[10 * i + k];
            p_lsq<5>(A, rho, 6, x, ws2);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
            betas[2] = x[3] / betas[0];
        }
        for (int i = 0; i < 4; i++) { PTR(406 + 21 * (N - 1) + i, betas[i]); b3[N - 1][i] = betas[i]; }
    }
    {
        const int ln = (int)(threadIdx.x & 63), var = ln < 2 ? ln : 2;      // lanes 3.. repeat the third variant
        double betas[4];
        for (int i = 0; i < 4; i++) betas[i] = var == 0 ? b3[0][i] : (var == 1 ? b3[1][i] : b3[2][i]);
        p_gauss_newton(L, rho, betas, ws2);
        double R[9], t[3];
        const double err = p_epnp_R_t(pws, us, alphas, K, ut, betas, R, t);
        double* sv = ws;                                                     // [3][17]: R, t, err, betas of the variants (the eigen-solver's A is dead)
        __syncthreads();
        if (ln < 3) {
            for (int i = 0; i < 9; i++) sv[17 * ln + i] = R[i];
            for (int i = 0; i < 3; i++) sv[17 * ln + 9 + i] = t[i];
            sv[17 * ln + 12] = err;
            for (int i = 0; i < 4; i++) sv[17 * ln + 13 + i] = betas[i];
        }
        __syncthreads();
        int pick = 0;
        double best = sv[12];
        for (int N = 2; N <= 3; N++) { const double e = sv[17 * (N - 1) + 12]; if (e < best) { best = e; pick = N - 1; } }
        for (int i = 0; i < 9; i++) Rbest[i] = sv[17 * pick + i];
        for (int i = 0; i < 3; i++) tbest[i] = sv[17 * pick + 9 + i];
        for (int N = 1; N <= 3; N++) {
            for (int i = 0; i < 4; i++) PTR(406 + 21 * (N - 1) + 4 + i, sv[17 * (N - 1) + 13 + i]);
            for (int i = 0; i < 9; i++) PTR(406 + 21 * (N - 1) + 8 + i, sv[17 * (N - 1) + i]);
            for (int i = 0; i < 3; i++) PTR(406 + 21 * (N - 1) + 17 + i, sv[17 * (N - 1) + 9 + i]);
            PTR(406 + 21 * (N - 1) + 20, sv[17 * (N - 1) + 12]);
        }
        __syncthreads();                                                     // sv is read before the next hypothesis reuses ws
    }
}

// ------------------------------------------------------------------------------------------------ RANSAC kernel
__device__ __forceinline__ unsigned p_rng_next(unsigned long long* st)
{
    *st = (unsigned long long)(unsigned)(*st) * 4164903690ull + (unsigned)(*st >> 32);
    return (unsigned)(*st);
}

__device__ int p_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : p > 1 ? 1 : p; ep = ep < 0 ? 0 : ep > 1 ? 1 : ep;
    double num = 1. - p; if (num < DBL_MIN) num = DBL_MIN;
    (void)model_points;                                   // 5: the only minimal-set size of this path
    double denom = 1. - dm_pow5(1. - ep);
    if (denom < DBL_MIN) return 0;
    num = dm_log(num); denom = dm_log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// The four-point problems of a batch (OpenCV: npoints == 4 -> SOLVEPNP_P3P on all four, one hypothesis): a lane per problem solves the
// perspective-three-point quartic of p3p_small.h and leaves the pose for k_pnp_ransac, which scores, refines and writes it like any other
// winner.  Its own launch: the quartic's scan and the closed-form alignment would cost the RANSAC kernel registers it does not have (434).
__global__ void k_pnp_p3p(const double* __restrict__ obj_in, const double* __restrict__ img_in, const int* __restrict__ n_arr, int stride, int B,
                          pnp_K K, double* __restrict__ pose0)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || n_arr[b] != 4) return;
    double obj[12], img[8], R[9], t[3];
    for (int i = 0; i < 12; i++) obj[i] = (double)(float)obj_in[(size_t)b * stride * 3 + i];        // CV_64F -> CV_32F, as the RANSAC path rounds its input
    for (int i = 0; i < 8; i++) img[i] = (double)(float)img_in[(size_t)b * stride * 2 + i];
    double* o = pose0 + (size_t)b * 13;
    const int ok = p3_solve4(obj, img, K.fu, K.fv, K.uc, K.vc, R, t);
    o[12] = ok ? 1.0 : 0.0;
    if (ok) { for (int k = 0; k < 9; k++) o[k] = R[k]; for (int k = 0; k < 3; k++) o[9 + k] = t[k]; }
}

__global__ __launch_bounds__(PNP_T) void k_pnp_ransac(const double* __restrict__ obj_in, const double* __restrict__ img_in,
                                                      const int* __restrict__ n_arr, int stride, pnp_K K, int iterations,
                                                      double reproj_error, double confidence, double* __restrict__ rvec,
                                                      double* __restrict__ tvec, int* __restrict__ inliers,
                                                      int* __restrict__ n_inliers, int* __restrict__ ok_out,
                                                      uint8_t* __restrict__ mask_ws, int lm_max, double* trace_all,
                                                      const double* __restrict__ pose0)
{
    __shared__ int s_idx[5];                                  // the current iteration's sample (drawn lazily: any iterationsCount works)
    __shared__ double s_ws[3 * 144 + 12];
    __shared__ double s_ws2[3 * 36 + 6 + 60 + 6 + 20 + 24];   // + rotation scratch of pj_eig_wave
    __shared__ double s_samp[25];
    __shared__ double s_red[28];
    __shared__ double s_j[14][PNP_T];                         // one 64-point chunk of the refinement: J0[6], J1[6], e0, e1 per point
    __shared__ double s_pose[12], s_try[12], s_c1;
    __shared__ int s_best, s_flag, s_w[PNP_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t b = blockIdx.x;
    const int n = n_arr[b];
    const double* obj = obj_in + b * (size_t)stride * 3;
    const double* img = img_in + b * (size_t)stride * 2;
    uint8_t* mask = mask_ws + b * (size_t)stride;
    int* out_idx = inliers + b * (size_t)stride;
    if (n < 4 || (n == 4 && !(pose0 && pose0[b * 13 + 12] > 0.0))) { if (tid == 0) { ok_out[b] = 0; n_inliers[b] = 0; } return; }
    const int niter0 = n == 5 ? 1 : max(1, iterations);
    // phase 0: the sample stream of cv::RNG((uint64)-1) is drawn LAZILY, five indices at the start of each iteration, by lane 0 (an
    // accepted model shrinks the iteration budget to one or two on a tracking problem; drawing all `iterations` samples up front was
    // 28 us of a single lane's 64-bit modulo arithmetic per problem)
    unsigned long long rng_st = ~0ull;
    // phases 1-3: the RANSAC iterations in sequence, exactly as the reference runs them (an accepted model shrinks niters, so a
    // typical tracking problem needs a handful of hypotheses, not all `iterations`); inside an iteration the wave cooperates:
    // EPnP's 12x12 eigen-decomposition in LDS, the reprojection test one point per lane
    const float thr2 = (float)(reproj_error * reproj_error);
    if (n == 4) {                                               // the four-point model of k_pnp_p3p: the one hypothesis of this problem
        if (tid == 0) { s_best = 0; for (int k = 0; k < 12; k++) s_pose[k] = pose0[b * 13 + k]; }
    } else {
        int best = -1, max_good = 0, niters = niter0;
        double bR[9], bt[3];
        for (int it = 0; it < niters; it++) {
            if (tid == 0) {
                int* idx = s_idx;
                if (n == 5) { for (int i = 0; i < 5; i++) idx[i] = i; }
                else
                    for (int i = 0; i < 5;) {
                        int j;
                        for (;;) { idx[i] = (int)(p_rng_next(&rng_st) % (unsigned)n); for (j = 0; j < i; j++) if (idx[i] == idx[j]) break; if (j == i) break; }
                        i++;
                    }
            }
            __syncthreads();
            double* so = s_samp; double* si = s_samp + 15;      // the sample, in LDS (every lane writes and reads the same values)
            double R[9], t[3];
            for (int i = 0; i < 5; i++) {
                const int p = s_idx[i];
                for (int k = 0; k < 3; k++) so[3 * i + k] = (double)(float)obj[3 * p + k];
                for (int k = 0; k < 2; k++) si[2 * i + k] = (double)(float)img[2 * p + k];
            }
            double* trace = trace_all && it < PNP_TRACE_HYP ? trace_all + (b * PNP_TRACE_HYP + it) * PNP_TRACE_STRIDE : nullptr;
            p_epnp5(so, si, K, R, t, s_ws, s_ws2, trace);
            int good;
            if (n == 5) good = 5;
            else {
                int cnt = 0;
                for (int i = tid; i < n; i += PNP_T) {
                    const double X[3] = { (double)(float)obj[3 * i], (double)(float)obj[3 * i + 1], (double)(float)obj[3 * i + 2] };
                    const float u = (float)img[2 * i], v = (float)img[2 * i + 1];
                    const double x = p_dot(R, X) + t[0], y = p_dot(R + 3, X) + t[1], z = p_dot(R + 6, X) + t[2];
                    const float pu = (float)(K.fu * x / z + K.uc), pv = (float)(K.fv * y / z + K.vc);
                    const float du = u - pu, dv = v - pv;
                    if (du * du + dv * dv <= thr2) cnt++;
                }
                for (int m = 32; m > 0; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
                good = cnt;
            }
            if (trace && tid == 0) { trace[0] = it; for (int i = 0; i < 5; i++) trace[1 + i] = s_idx[i]; for (int i = 0; i < 9; i++) trace[469 + i] = R[i]; for (int i = 0; i < 3; i++) trace[478 + i] = t[i]; trace[481] = good; }
            if (n == 5 || good > (max_good > 4 ? max_good : 4)) {
                best = it; max_good = good;
                for (int k = 0; k < 9; k++) bR[k] = R[k];
                for (int k = 0; k < 3; k++) bt[k] = t[k];
                if (n > 5) niters = p_update_iters(confidence, (double)(n - good) / n, 5, niters);
            }
        }
        if (tid == 0) {
            s_best = best;
            if (best >= 0) { for (int k = 0; k < 9; k++) s_pose[k] = bR[k]; for (int k = 0; k < 3; k++) s_pose[9 + k] = bt[k]; }
        }
    }
    __syncthreads();
    const int best = s_best;
    if (best < 0) { if (tid == 0) { ok_out[b] = 0; n_inliers[b] = 0; } return; }
    // inlier mask of the winner
    for (int i = tid; i < n; i += PNP_T) {
        uint8_t m = 1;
        if (n != 5) {
            const double X[3] = { (double)(float)obj[3 * i], (double)(float)obj[3 * i + 1], (double)(float)obj[3 * i + 2] };
            const double x = p_dot(s_pose, X) + s_pose[9], y = p_dot(s_pose + 3, X) + s_pose[10], z = p_dot(s_pose + 6, X) + s_pose[11];
            const float pu = (float)(K.fu * x / z + K.uc), pv = (float)(K.fv * y / z + K.vc);
            const float du = (float)img[2 * i] - pu, dv = (float)img[2 * i + 1] - pv;
            m = du * du + dv * dv <= thr2;
        }
        mask[i] = m;
    }
    __syncthreads();
    if (n == 4) {                                               // count == modelPoints: accepted only when all four points are inliers
        const bool all4 = mask[0] && mask[1] && mask[2] && mask[3];
        if (!all4) { if (tid == 0) { ok_out[b] = 0; n_inliers[b] = 0; } return; }
    }
    // phase 4: Levenberg-Marquardt on the inliers.  The normal equations are summed in POINT ORDER, like the sequential CPU
    // restatement (a tree reduction would differ in the last bits, and this pose feeds the next frame's float32-rounded input):
    // per 64-point chunk every lane writes its point's two Jacobian rows and residuals to LDS, then lane k < 28 walks the chunk
    // in order for accumulator k (21 upper-triangle entries of H, 6 of g, the cost).
    int r0, r1, r2, r3;                                       // accumulator k: acc += s_j[r0][j] * s_j[r1][j] + s_j[r2][j] * s_j[r3][j]
    {
        int k = lane, a = 0;
        if (k < 21) { while (k >= 6 - a) { k -= 6 - a; a++; } r0 = a; r1 = a + k; r2 = 6 + a; r3 = 6 + a + k; }
        else if (k < 27) { r0 = k - 21; r1 = 12; r2 = 6 + (k - 21); r3 = 13; }
        else { r0 = 12; r1 = 12; r2 = 13; r3 = 13; }
    }
    double lambda = 1e-3;
    for (int it = 0; it < lm_max; it++) {
        double acc = 0;
        for (int c0 = 0; c0 < n; c0 += PNP_T) {
            const int i = c0 + tid;
            const bool valid = i < n && mask[i];
            double J0[6] = {0, 0, 0, 0, 0, 0}, J1[6] = {0, 0, 0, 0, 0, 0}, e0 = 0, e1 = 0;
            if (valid) {
                const double X[3] = { (double)(float)obj[3 * i], (double)(float)obj[3 * i + 1], (double)(float)obj[3 * i + 2] };
                const double rx[3] = { p_dot(s_pose, X), p_dot(s_pose + 3, X), p_dot(s_pose + 6, X) };
                const double x = rx[0] + s_pose[9], y = rx[1] + s_pose[10], z = rx[2] + s_pose[11], iz = 1.0 / z;
                e0 = K.fu * x * iz + K.uc - (double)(float)img[2 * i]; e1 = K.fv * y * iz + K.vc - (double)(float)img[2 * i + 1];
                const double du[3] = { K.fu * iz, 0, -K.fu * x * iz * iz }, dv[3] = { 0, K.fv * iz, -K.fv * y * iz * iz };
                J0[0] = du[1] * (-rx[2]) + du[2] * rx[1]; J0[1] = du[0] * rx[2] + du[2] * (-rx[0]); J0[2] = du[0] * (-rx[1]) + du[1] * rx[0];
                J0[3] = du[0]; J0[4] = du[1]; J0[5] = du[2];
                J1[0] = dv[1] * (-rx[2]) + dv[2] * rx[1]; J1[1] = dv[0] * rx[2] + dv[2] * (-rx[0]); J1[2] = dv[0] * (-rx[1]) + dv[1] * rx[0];
                J1[3] = dv[0]; J1[4] = dv[1]; J1[5] = dv[2];
            }
            __syncthreads();                                  // the previous chunk's readers are done
#pragma unroll
            for (int a = 0; a < 6; a++) { s_j[a][tid] = J0[a]; s_j[6 + a][tid] = J1[a]; }
            s_j[12][tid] = e0; s_j[13][tid] = e1;
            __syncthreads();
            if (lane < 28) {
                // points that are not inliers (or beyond n) have zero rows: adding their +0.0 leaves the sum bit for bit what the
                // CPU restatement gets by skipping them, and the loop can be unrolled (loads in flight, only the adds are a chain)
#pragma unroll 16
                for (int j = 0; j < PNP_T; j++) acc += s_j[r0][j] * s_j[r1][j] + s_j[r2][j] * s_j[r3][j];
            }
        }
        __syncthreads();
        if (lane < 28) s_red[lane] = acc;
        __syncthreads();
        const double c0 = s_red[27];                          // H (21 upper-triangle entries), g (6) and the cost stay in s_red for the trials
        bool improved = false, tiny = false;
        for (int trial = 0; trial < 10 && !improved; trial++) {
            {   // damped normal equations (SPD): Cholesky in registers, the same statements as the CPU restatement (pnp_small.h)
                double A[36], ng[6], dx[6];
                {
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int c = a; c < 6; c++) { const double h = s_red[k]; A[6 * a + c] = h; A[6 * c + a] = h; k++; }
                }
#pragma unroll
                for (int a = 0; a < 6; a++) { const double h = A[7 * a]; A[7 * a] = h + lambda * (h > 1e-12 ? h : 1e-12); ng[a] = -s_red[21 + a]; }
                ps_chol6_solve(A, ng, dx);
                if (tid == 0) {
                    double dR[9];
                    pd_rodrigues_v2m(dx, dR);
                    for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) s_try[3 * a + c] = dR[3 * a] * s_pose[c] + dR[3 * a + 1] * s_pose[3 + c] + dR[3 * a + 2] * s_pose[6 + c];
                    for (int a = 0; a < 3; a++) s_try[9 + a] = s_pose[9 + a] + dx[3 + a];
                    double step = 0; for (int a = 0; a < 6; a++) step += dx[a] * dx[a];
                    s_flag = step < 1e-18 ? 1 : 0;
                }
            }
            __syncthreads();
            double c1 = 0;                                    // lane 0 sums the trial cost in point order
            for (int cb = 0; cb < n; cb += PNP_T) {
                const int i = cb + tid;
                const bool valid = i < n && mask[i];
                double e2 = 0;
                if (valid) {
                    const double X[3] = { (double)(float)obj[3 * i], (double)(float)obj[3 * i + 1], (double)(float)obj[3 * i + 2] };
                    const double x = p_dot(s_try, X) + s_try[9], y = p_dot(s_try + 3, X) + s_try[10], z = p_dot(s_try + 6, X) + s_try[11];
                    const double e0 = K.fu * x / z + K.uc - (double)(float)img[2 * i], e1 = K.fv * y / z + K.vc - (double)(float)img[2 * i + 1];
                    e2 = e0 * e0 + e1 * e1;
                }
                __syncthreads();
                s_j[0][tid] = e2;
                __syncthreads();
                if (tid == 0) {
#pragma unroll 16
                    for (int j = 0; j < PNP_T; j++) c1 += s_j[0][j];          // zeros for the points that do not count
                }
            }
            if (tid == 0) s_c1 = c1;
            __syncthreads();
            c1 = s_c1;
            __syncthreads();
            if (c1 < c0 && isfinite(c1)) {
                improved = true; tiny = s_flag != 0; lambda *= 0.2;
                if (tid < 12) s_pose[tid] = s_try[tid];
            } else lambda *= 5.0;
            __syncthreads();
        }
        if (!improved || tiny) break;
    }
    // outputs: ordered inlier indices, rvec, tvec
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += PNP_T) {
        const int i = c0 + tid;
        const bool keep = i < n && mask[i];
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) s_w[wv] = __popcll(m);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < PNP_T / 64; w++) { const int c = s_w[w]; if (w < wv) wbase += c; tot += c; }
        if (keep) out_idx[base + wbase + pre] = i;
        base += tot;
    }
    if (tid == 0) {
        double r[3];
        pd_rodrigues_m2v(s_pose, r);
        for (int k = 0; k < 3; k++) { rvec[3 * b + k] = r[k]; tvec[3 * b + k] = s_pose[9 + k]; }
        n_inliers[b] = base; ok_out[b] = 1;
    }
}

extern "C" int bh_solve_pnp_ransac(bh_ctx* ctx, const double* obj, const double* img, const int* n, int B, int stride,
                                   const double K[4], int iterations, double reproj_error, double confidence,
                                   double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!obj || !img || !n || !K || !rvec || !tvec || !inliers || !n_inliers || !ok || B < 0 || stride < 1 ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE) || !(confidence > 0 && confidence < 1))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_solve_pnp_ransac: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t np = (size_t)B * stride;
    const double *d_obj = obj, *d_img = img; const int* d_n = n;
    double *d_r = rvec, *d_t = tvec; int *d_in = inliers, *d_ni = n_inliers, *d_ok = ok;
    BH_TRY(bh_reserve(ctx, &ctx->st_h, np));                                    // inlier-mask workspace
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (n[b] < 0 || n[b] > stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_solve_pnp_ransac: n[b] outside [0, stride]");
        BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 3 * np)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 2 * np));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(int) * 3 * (size_t)B)); BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(double) * 6 * (size_t)B));
        BH_TRY(bh_reserve(ctx, &ctx->st_e, sizeof(int) * np));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, obj, sizeof(double) * 3 * np, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, img, sizeof(double) * 2 * np, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, n, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_obj = (const double*)ctx->st_a.p; d_img = (const double*)ctx->st_b.p; d_n = (const int*)ctx->st_c.p;
        d_ni = (int*)ctx->st_c.p + B; d_ok = (int*)ctx->st_c.p + 2 * (size_t)B;
        d_r = (double*)ctx->st_d.p; d_t = d_r + 3 * (size_t)B; d_in = (int*)ctx->st_e.p;
    }
    pnp_K Kd = { K[0], K[1], K[2], K[3] };
    BH_TRY(bh_reserve(ctx, &ctx->pnp_p0, sizeof(double) * 13 * (size_t)B));
    hipLaunchKernelGGL(k_pnp_p3p, dim3((B + 63) / 64), dim3(64), 0, st, d_obj, d_img, d_n, stride, B, Kd, (double*)ctx->pnp_p0.p);
    BH_LAUNCH(ctx, BH_K_PNP, k_pnp_ransac, dim3(B), dim3(PNP_T), st, d_obj, d_img, d_n, stride, Kd, iterations, reproj_error,
              confidence, d_r, d_t, d_in, d_ni, d_ok, (uint8_t*)ctx->st_h.p, 20, (double*)nullptr, (const double*)ctx->pnp_p0.p);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(rvec, d_r, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(tvec, d_t, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(inliers, d_in, sizeof(int) * np, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_inliers, d_ni, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(ok, d_ok, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ test hooks
// bh_debug_pnp_trace: bh_solve_pnp_ransac on host arrays that also returns, per problem, the intermediates of its first
// PNP_TRACE_HYP hypotheses (trace f64[B][32][512]; layout in tests/pnp_trace_layout.py) — used to locate the first differing
// operation against the oracle's own trace.
extern "C" int bh_debug_pnp_trace(bh_ctx* ctx, const double* obj, const double* img, const int* n, int B, int stride,
                                  const double K[4], int iterations, double reproj_error, double confidence,
                                  double* rvec, double* tvec, int* inliers, int* n_inliers, int* ok, double* trace)
{
    if (!ctx || !obj || !img || !n || !K || !rvec || !tvec || !inliers || !n_inliers || !ok || !trace || B < 1 || stride < 1)
        return BH_ERR_INVALID;
    for (int b = 0; b < B; b++) if (n[b] < 0 || n[b] > stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_debug_pnp_trace: n[b] outside [0, stride]");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t np = (size_t)B * stride, ntr = (size_t)B * PNP_TRACE_HYP * PNP_TRACE_STRIDE;
    BH_TRY(bh_reserve(ctx, &ctx->st_h, np));
    BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 3 * np)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 2 * np));
    BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(int) * 3 * (size_t)B)); BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(double) * 6 * (size_t)B));
    BH_TRY(bh_reserve(ctx, &ctx->st_e, sizeof(int) * np));
    BH_TRY(bh_reserve(ctx, &ctx->st_f, sizeof(double) * ntr));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, obj, sizeof(double) * 3 * np, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, img, sizeof(double) * 2 * np, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, n, sizeof(int) * B, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemsetAsync(ctx->st_f.p, 0, sizeof(double) * ntr, st));
    int* d_ni = (int*)ctx->st_c.p + B; int* d_ok = (int*)ctx->st_c.p + 2 * (size_t)B;
    double* d_r = (double*)ctx->st_d.p; double* d_t = d_r + 3 * (size_t)B;
    pnp_K Kd = { K[0], K[1], K[2], K[3] };
    hipLaunchKernelGGL(k_pnp_ransac, dim3(B), dim3(PNP_T), 0, st, (const double*)ctx->st_a.p, (const double*)ctx->st_b.p,
                       (const int*)ctx->st_c.p, stride, Kd, iterations, reproj_error, confidence, d_r, d_t, (int*)ctx->st_e.p,
                       d_ni, d_ok, (uint8_t*)ctx->st_h.p, 20, (double*)ctx->st_f.p, (const double*)nullptr);      // (the trace hook is for the RANSAC path: four-point problems report failure here)
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipMemcpyAsync(rvec, d_r, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(tvec, d_t, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(inliers, ctx->st_e.p, sizeof(int) * np, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(n_inliers, d_ni, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(ok, d_ok, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(trace, ctx->st_f.p, sizeof(double) * ntr, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}

// bh_debug_f64_ops: the f64 primitives the pose kernels lean on, evaluated on the device for n operand pairs, so a test can
// compare them bit for bit with the host's libm: out f64[8][n] = a/b, sqrt|a|, sin a, cos a, acos(clamp a), log|a|, pow(|a|,5), rint(a).
__global__ void k_debug_f64_ops(const double* a, const double* b, int n, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    out[0 * (size_t)n + i] = x / y;
    out[1 * (size_t)n + i] = sqrt(fabs(x));
    out[2 * (size_t)n + i] = sin(x);
    out[3 * (size_t)n + i] = cos(x);
    out[4 * (size_t)n + i] = acos(x > 1 ? 1 : x < -1 ? -1 : x);
    out[5 * (size_t)n + i] = log(fabs(x));
    out[6 * (size_t)n + i] = pow(fabs(x), 5.0);
    out[7 * (size_t)n + i] = rint(x);
}

extern "C" int bh_debug_f64_ops(bh_ctx* ctx, const double* a, const double* b, int n, double* out)
{
    if (!ctx || !a || !b || !out || n < 1) return BH_ERR_INVALID;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * n)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * n));
    BH_TRY(bh_reserve(ctx, &ctx->st_f, sizeof(double) * 8 * (size_t)n));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, a, sizeof(double) * n, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, b, sizeof(double) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_debug_f64_ops, dim3((n + 255) / 256), dim3(256), 0, st, (const double*)ctx->st_a.p, (const double*)ctx->st_b.p, n, (double*)ctx->st_f.p);
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipMemcpyAsync(out, ctx->st_f.p, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}

// bh_debug_p3p: the four-point model of k_pnp_p3p for B problems on host arrays (obj f64[B][4][3], img f64[B][4][2]): pose f64[B][13] =
// R (9), t (3), ok - compared bit for bit with the checker's p3p_small.h in the tests
extern "C" int bh_debug_p3p(bh_ctx* ctx, const double* obj, const double* img, int B, const double K[4], double* pose)
{
    if (!ctx || !obj || !img || !K || !pose || B < 1) return BH_ERR_INVALID;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 12 * (size_t)B)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 8 * (size_t)B));
    BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(int) * (size_t)B)); BH_TRY(bh_reserve(ctx, &ctx->pnp_p0, sizeof(double) * 13 * (size_t)B));
    std::vector<int> n4(B, 4);
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, obj, sizeof(double) * 12 * (size_t)B, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, img, sizeof(double) * 8 * (size_t)B, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, n4.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemsetAsync(ctx->pnp_p0.p, 0, sizeof(double) * 13 * (size_t)B, st));
    pnp_K Kd = { K[0], K[1], K[2], K[3] };
    hipLaunchKernelGGL(k_pnp_p3p, dim3((B + 63) / 64), dim3(64), 0, st, (const double*)ctx->st_a.p, (const double*)ctx->st_b.p, (const int*)ctx->st_c.p, 4, B, Kd,
                       (double*)ctx->pnp_p0.p);
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipMemcpyAsync(pose, ctx->pnp_p0.p, sizeof(double) * 13 * (size_t)B, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}
