/* se3_small.h - the closed-form rigid alignment behind bh_align_se3_ransac, SHARED AS TEXT between the HIP kernel
 * (boslam_amd/csrc/align.hip) and its CPU checker (align_oracle.c of the test tree; tests/test_host_cpu.py asserts the two copies are
 * identical).  Every routine is a fixed sequence of IEEE double operations (+ - * / sqrt fabs), both sides are compiled with
 * -ffp-contract=off, so kernel and checker agree bit for bit; tests/test_oracle_cpu.py checks the routines themselves against
 * numpy's SVD.
 *
 * Problem: correspondences (p_i, q_i) in R^3; find R in SO(3), t with q_i ~ R p_i + t in the least-squares sense
 * (Horn 1987 / Kabsch / Umeyama with the scale fixed to 1 - RGB-D maps are metric).  With the centroids pbar, qbar and
 * H = sum_i (p_i - pbar)(q_i - qbar)' = U diag(s) V':  R = V diag(1, 1, det(V U')) U' (the -1, if any, on the smallest
 * singular value), t = qbar - R pbar. */
/* ---- identical from here on ---- */
#ifndef BH_SE3_SMALL_H
#define BH_SE3_SMALL_H
#include <math.h>

#if defined(__HIPCC__)
#define SE_FN __host__ __device__ static inline
#else
#define SE_FN static inline
#endif

SE_FN double se_det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* one-sided Jacobi SVD of a 3 x 3 matrix (row-major): A = U diag(s) V', singular values unsorted, U's columns of zero singular
 * values completed to a right-handed frame */
SE_FN void se_svd3(const double* A, double* U, double* s, double* V)
{
    double B[9];
    for (int i = 0; i < 9; i++) { B[i] = A[i]; V[i] = (i % 4) == 0 ? 1.0 : 0.0; }
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
}

/* R, t from the centred cross-covariance H (row-major, H[a][b] = sum (p_a - pbar_a)(q_b - qbar_b)) and the centroids.
 * Returns 0 when the correspondences do not fix a rotation (fewer than two independent directions: collinear or coincident points). */
SE_FN int se_kabsch(const double* H, const double* pbar, const double* qbar, double* R, double* t)
{
    double U[9], s[3], V[9];
    se_svd3(H, U, s, V);
    int jmin = 0, jmax = 0;
    for (int j = 1; j < 3; j++) { if (s[j] < s[jmin]) jmin = j; if (s[j] > s[jmax]) jmax = j; }
    const int jmid = 3 - jmin - jmax;
    if (jmin == jmax || !(s[jmax] > 0.0) || !(s[jmid] > 1e-12 * s[jmax])) return 0;
    /* the column of U that belongs to the smallest singular value: the cross product of the other two (a rank-2 H leaves it undefined) */
    {
        const int a = jmax, b = jmid;
        double c0 = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
        double c1 = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
        double c2 = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
        const double n = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
        if (!(n > 0.0)) return 0;
        U[0 * 3 + jmin] = c0 / n; U[1 * 3 + jmin] = c1 / n; U[2 * 3 + jmin] = c2 / n;
    }
    /* det(V U') = det(V) det(U); the reflection, if any, goes onto the smallest singular value */
    const double d = se_det3(V) * se_det3(U) < 0.0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double acc = V[r * 3 + jmax] * U[c * 3 + jmax];
            acc = acc + V[r * 3 + jmid] * U[c * 3 + jmid];
            acc = acc + d * (V[r * 3 + jmin] * U[c * 3 + jmin]);
            R[r * 3 + c] = acc;
        }
    for (int r = 0; r < 3; r++) t[r] = qbar[r] - ((R[r * 3] * pbar[0] + R[r * 3 + 1] * pbar[1]) + R[r * 3 + 2] * pbar[2]);
    return 1;
}

/* squared distance of q from R p + t */
SE_FN double se_resid2(const double* R, const double* t, const double* p, const double* q)
{
    const double x = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
    const double y = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
    const double z = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
    const double dx = q[0] - x, dy = q[1] - y, dz = q[2] - z;
    return (dx * dx + dy * dy) + dz * dz;
}

/* natural logarithm of a positive normal double (the routine of detmath.h, restated so that this header stands alone) */
SE_FN double se_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 7.07106781186547572737e-01) { m = m * 2.0; e = e - 1; }
    const double tt = (m - 1.0) / (m + 1.0);
    const double w = tt * tt;
    double r = 1.0 / 27.0;
    r = r * w + 1.0 / 25.0; r = r * w + 1.0 / 23.0; r = r * w + 1.0 / 21.0; r = r * w + 1.0 / 19.0; r = r * w + 1.0 / 17.0; r = r * w + 1.0 / 15.0;
    r = r * w + 1.0 / 13.0; r = r * w + 1.0 / 11.0; r = r * w + 1.0 / 9.0; r = r * w + 1.0 / 7.0; r = r * w + 1.0 / 5.0; r = r * w + 1.0 / 3.0;
    const double tw = tt * w;
    const double lm = 2.0 * (tt + tw * r);
    const double de = (double)e;
    return (de * 6.93147180369123816490e-01 + lm) + de * 1.90821492927058770002e-10;
}

/* RANSAC's adaptive iteration count for 3-point minimal sets (the rule of OpenCV's RANSACUpdateNumIters) */
SE_FN int se_update_iters(double conf, double outlier_ratio, int max_iters)
{
    const double p = conf < 0 ? 0 : conf > 1 ? 1 : conf, ep = outlier_ratio < 0 ? 0 : outlier_ratio > 1 ? 1 : outlier_ratio;
    double num = 1.0 - p; if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    const double w = 1.0 - ep;
    double denom = 1.0 - (w * w) * w;
    if (denom < 2.2250738585072014e-308) return 0;
    num = se_log(num); denom = se_log(denom);
    if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
    const double qd = num / denom;
    return (int)(qd + 0.5);                                      /* qd > 0: round half up */
}
#endif
