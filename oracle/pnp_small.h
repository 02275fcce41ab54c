/*
 * pnp_small.h — the CPU oracle's copy of the two small dense solves of solvePnPRansac (TEST INFRASTRUCTURE).  See
 * boslam_amd/csrc/pnp_small.h; the two files are byte-identical below the marker (tests/test_host_cpu.py compares them).
 */
/* ---- identical from here on ---- */
#ifndef BH_PNP_SMALL_H
#define BH_PNP_SMALL_H
#include <math.h>

#if defined(__HIPCC__)
#define PS_FN __host__ __device__ static inline
#define PS_UNROLL _Pragma("unroll")
#else
#define PS_FN static inline
#define PS_UNROLL
#endif

/* A: 6 x 4 row-major (destroyed), b: 6 (destroyed), x: 4 */
PS_FN void ps_qr_solve_6x4(double* A, double* b, double* x)
{
    double A1[4], A2[4];
    PS_UNROLL
    for (int k = 0; k < 4; k++) {
        double eta = fabs(A[4 * k + k]);
        PS_UNROLL
        for (int i = k + 1; i < 6; i++) { const double e = fabs(A[4 * i + k]); if (e > eta) eta = e; }
        if (eta == 0.0) { x[0] = 0.0; x[1] = 0.0; x[2] = 0.0; x[3] = 0.0; return; }
        const double inv_eta = 1.0 / eta;
        double sum = 0.0;
        PS_UNROLL
        for (int i = k; i < 6; i++) { A[4 * i + k] = A[4 * i + k] * inv_eta; sum = sum + A[4 * i + k] * A[4 * i + k]; }
        double sigma = sqrt(sum);
        if (A[4 * k + k] < 0.0) sigma = -sigma;
        A[4 * k + k] = A[4 * k + k] + sigma;
        A1[k] = sigma * A[4 * k + k];
        A2[k] = -eta * sigma;
        PS_UNROLL
        for (int j = k + 1; j < 4; j++) {
            double s = 0.0;
            PS_UNROLL
            for (int i = k; i < 6; i++) s = s + A[4 * i + k] * A[4 * i + j];
            const double tau = s / A1[k];
            PS_UNROLL
            for (int i = k; i < 6; i++) A[4 * i + j] = A[4 * i + j] - tau * A[4 * i + k];
        }
    }
    PS_UNROLL
    for (int j = 0; j < 4; j++) {                       /* b <- Q' b */
        double s = 0.0;
        PS_UNROLL
        for (int i = j; i < 6; i++) s = s + A[4 * i + j] * b[i];
        const double tau = s / A1[j];
        PS_UNROLL
        for (int i = j; i < 6; i++) b[i] = b[i] - tau * A[4 * i + j];
    }
    x[3] = b[3] / A2[3];                                /* R x = Q' b, R's diagonal is A2 */
    PS_UNROLL
    for (int i = 2; i >= 0; i--) {
        double s = 0.0;
        PS_UNROLL
        for (int j = i + 1; j < 4; j++) s = s + A[4 * i + j] * x[j];
        x[i] = (b[i] - s) / A2[i];
    }
}

/* A: 6 x 6 row-major symmetric (lower part read), b: 6, x: 6.  Returns 1, or 0 with x = 0 when A is not positive definite. */
PS_FN int ps_chol6_solve(const double* A, const double* b, double* x)
{
    double L[36];
    PS_UNROLL
    for (int j = 0; j < 6; j++) {
        double d = A[6 * j + j];
        PS_UNROLL
        for (int k = 0; k < j; k++) d = d - L[6 * j + k] * L[6 * j + k];
        if (!(d > 0.0) || !(d < 1.0e300)) { for (int i = 0; i < 6; i++) x[i] = 0.0; return 0; }
        const double l = sqrt(d);
        L[6 * j + j] = l;
        PS_UNROLL
        for (int i = j + 1; i < 6; i++) {
            double s = A[6 * i + j];
            PS_UNROLL
            for (int k = 0; k < j; k++) s = s - L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = s / l;
        }
    }
    double y[6];
    PS_UNROLL
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        PS_UNROLL
        for (int k = 0; k < i; k++) s = s - L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    PS_UNROLL
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        PS_UNROLL
        for (int k = i + 1; k < 6; k++) s = s - L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return 1;
}
#endif
