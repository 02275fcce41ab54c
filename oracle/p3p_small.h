/* p3p_small.h - TEST INFRASTRUCTURE copy (oracle/): the four-point branch of cv2.solvePnPRansac as a fixed sequence of IEEE double
 * operations, the same text as boslam_amd/csrc/p3p_small.h from the marker on (tests/test_host_cpu.py asserts it), so that the CPU checker
 * (pnp_oracle.c) and the HIP kernel agree bit for bit.  What the routines are and where they come from: see the kernel-side copy's header.
 * Checked on its own against numpy.roots and generated poses in tests/test_oracle_cpu.py. */
/* ---- identical from here on ---- */
#ifndef BH_P3P_SMALL_H
#define BH_P3P_SMALL_H
#include <math.h>
#include "se3_small.h"

#if defined(__HIPCC__)
#define P3_FN __host__ __device__ static inline
#else
#define P3_FN static inline
#endif

/* the quartic (coefficients A[0] v^4 + .. + A[4]) at v = t / (1 - t), times (1 - t)^4 */
P3_FN double p3_q(const double* A, double t)
{
    const double w = 1.0 - t;
    const double t2 = t * t, w2 = w * w;
    double r = A[0] * (t2 * t2);
    r = r + A[1] * ((t2 * t) * w);
    r = r + A[2] * (t2 * w2);
    r = r + A[3] * (t * (w2 * w));
    r = r + A[4] * (w2 * w2);
    return r;
}

/* derivative of p3_q with respect to t from its monomial coefficients c[1..4] */
P3_FN double p3_dq(const double* c, double t)
{
    return ((4.0 * c[4] * t + 3.0 * c[3]) * t + 2.0 * c[2]) * t + c[1];
}

/* the root of p3_q between lo and hi (signs differ at the ends; qlo = the value at lo), as v = t / (1 - t) */
P3_FN double p3_bisect(const double* A, double lo, double hi, double qlo)
{
    for (int it = 0; it < 60; it++) {
        const double mid = 0.5 * (lo + hi), qm = p3_q(A, mid);
        if (qm == 0.0) { lo = mid; hi = mid; break; }
        if ((qlo < 0.0) == (qm < 0.0)) { lo = mid; qlo = qm; } else hi = mid;
    }
    const double tr = 0.5 * (lo + hi);
    return tr / (1.0 - tr);
}

/* positive real roots v of the quartic, ascending, at most 4; returns their number.  A step of the scan without a sign change can still hold
 * two roots (a close pair - the two poses of a nearly degenerate triangle): then the derivative changes sign inside it, the extremum is
 * found by bisection on the derivative, and if the quartic has the other sign there both halves are bisected. */
P3_FN int p3_roots(const double* A, double* v)
{
    int n = 0;
    double c[5];
    c[0] = A[4];
    c[1] = A[3] - 4.0 * A[4];
    c[2] = (A[2] - 3.0 * A[3]) + 6.0 * A[4];
    c[3] = ((A[1] - 2.0 * A[2]) + 3.0 * A[3]) - 4.0 * A[4];
    c[4] = (((A[0] - A[1]) + A[2]) - A[3]) + A[4];
    const double t0 = 1.0 / 1024.0, dt = (1.0 - 2.0 / 1024.0) / 512.0;
    double ta = t0, qa = p3_q(A, ta), da = p3_dq(c, ta);
    for (int k = 1; k <= 512 && n < 4; k++) {
        const double tb = t0 + dt * (double)k, qb = p3_q(A, tb), db = p3_dq(c, tb);
        if (qb == 0.0) v[n++] = tb / (1.0 - tb);
        else if (qa != 0.0 && (qa < 0.0) != (qb < 0.0)) v[n++] = p3_bisect(A, ta, tb, qa);
        else if (qa != 0.0 && (da < 0.0) != (db < 0.0)) {
            double lo = ta, hi = tb, dlo = da;
            for (int it = 0; it < 50; it++) {
                const double mid = 0.5 * (lo + hi), dm = p3_dq(c, mid);
                if ((dlo < 0.0) == (dm < 0.0)) { lo = mid; dlo = dm; } else hi = mid;
            }
            const double te = 0.5 * (lo + hi), qe = p3_q(A, te);
            if (qe == 0.0) v[n++] = te / (1.0 - te);
            else if ((qe < 0.0) != (qa < 0.0)) {
                v[n++] = p3_bisect(A, ta, te, qa);
                if (n < 4) v[n++] = p3_bisect(A, te, tb, qe);
            }
        }
        ta = tb; qa = qb; da = db;
    }
    return n;
}

/* up to four poses (R row-major 3x3, t) with s_i f_i = R X_i + t for the three correspondences X[3][3] (world), f[3][3] (unit rays);
 * returns their number */
P3_FN int p3_solve3(const double* X, const double* f, double* Rs, double* ts)
{
    double d12[3], d02[3], d01[3];
    for (int k = 0; k < 3; k++) { d12[k] = X[3 + k] - X[6 + k]; d02[k] = X[k] - X[6 + k]; d01[k] = X[k] - X[3 + k]; }
    const double a2 = (d12[0] * d12[0] + d12[1] * d12[1]) + d12[2] * d12[2];
    const double b2 = (d02[0] * d02[0] + d02[1] * d02[1]) + d02[2] * d02[2];
    const double c2 = (d01[0] * d01[0] + d01[1] * d01[1]) + d01[2] * d01[2];
    if (!(b2 > 1e-300) || !(a2 > 0.0) || !(c2 > 0.0)) return 0;
    const double ca = (f[3] * f[6] + f[4] * f[7]) + f[5] * f[8];
    const double cb = (f[0] * f[6] + f[1] * f[7]) + f[2] * f[8];
    const double cg = (f[0] * f[3] + f[1] * f[4]) + f[2] * f[5];
    const double K1 = (a2 - c2) / b2, Kc = c2 / b2;
    const double n2 = K1 - 1.0, n1 = (-2.0 * K1) * cb, n0 = 1.0 + K1;
    const double d1 = -2.0 * ca, d0 = 2.0 * cg;
    const double g2 = -Kc, g1 = (2.0 * Kc) * cb, g0 = 1.0 - Kc;
    const double NN[5] = { n2 * n2, (2.0 * n2) * n1, (2.0 * n2) * n0 + n1 * n1, (2.0 * n1) * n0, n0 * n0 };
    const double ND[5] = { 0.0, n2 * d1, n2 * d0 + n1 * d1, n1 * d0 + n0 * d1, n0 * d0 };
    const double DD[3] = { d1 * d1, (2.0 * d1) * d0, d0 * d0 };
    const double GD[5] = { g2 * DD[0], g2 * DD[1] + g1 * DD[0], (g2 * DD[2] + g1 * DD[1]) + g0 * DD[0], g1 * DD[2] + g0 * DD[1], g0 * DD[2] };
    double A[5];
    for (int i = 0; i < 5; i++) A[i] = (NN[i] - (2.0 * cg) * ND[i]) + GD[i];
    double vr[4];
    const int nr = p3_roots(A, vr);
    int ns = 0;
    for (int r = 0; r < nr; r++) {
        const double v = vr[r];
        const double D = d1 * v + d0;
        if (!(fabs(D) > 1e-12)) continue;
        const double u = ((n2 * v + n1) * v + n0) / D;
        const double den = (1.0 + v * v) - (2.0 * v) * cb;
        if (!(u > 0.0) || !(den > 0.0)) continue;
        const double s0 = sqrt(b2 / den), s1 = u * s0, s2 = v * s0;
        double P[9];
        for (int k = 0; k < 3; k++) { P[k] = s0 * f[k]; P[3 + k] = s1 * f[3 + k]; P[6 + k] = s2 * f[6 + k]; }
        double xb[3], pb[3], H[9];
        for (int k = 0; k < 3; k++) { xb[k] = ((X[k] + X[3 + k]) + X[6 + k]) / 3.0; pb[k] = ((P[k] + P[3 + k]) + P[6 + k]) / 3.0; }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double h = (X[a] - xb[a]) * (P[b] - pb[b]);
                h = h + (X[3 + a] - xb[a]) * (P[3 + b] - pb[b]);
                h = h + (X[6 + a] - xb[a]) * (P[6 + b] - pb[b]);
                H[3 * a + b] = h;
            }
        if (!se_kabsch(H, xb, pb, Rs + 9 * ns, ts + 3 * ns)) continue;
        ns++;
    }
    return ns;
}

/* solvePnPRansac's four-point model: obj[4][3], img[4][2] (pixels), intrinsics fx, fy, cx, cy; the solution of the first three correspondences that
 * reprojects the fourth best.  Returns 0 when there is none (degenerate triangle, no positive root, fourth point behind the camera everywhere). */
P3_FN int p3_solve4(const double* obj, const double* img, double fx, double fy, double cx, double cy, double* R, double* t)
{
    double f[9];
    for (int i = 0; i < 3; i++) {
        const double x = (img[2 * i] - cx) / fx, y = (img[2 * i + 1] - cy) / fy;
        const double nrm = sqrt((x * x + y * y) + 1.0);
        f[3 * i] = x / nrm; f[3 * i + 1] = y / nrm; f[3 * i + 2] = 1.0 / nrm;
    }
    double Rs[36], ts[12];
    const int ns = p3_solve3(obj, f, Rs, ts);
    int best = -1; double beste = 0.0;
    for (int s = 0; s < ns; s++) {
        const double* Rm = Rs + 9 * s; const double* tm = ts + 3 * s; const double* X = obj + 9;
        const double x = ((Rm[0] * X[0] + Rm[1] * X[1]) + Rm[2] * X[2]) + tm[0];
        const double y = ((Rm[3] * X[0] + Rm[4] * X[1]) + Rm[5] * X[2]) + tm[1];
        const double z = ((Rm[6] * X[0] + Rm[7] * X[1]) + Rm[8] * X[2]) + tm[2];
        if (!(z > 0.0)) continue;
        const double du = (fx * x / z + cx) - img[6], dv = (fy * y / z + cy) - img[7];
        const double e = du * du + dv * dv;
        if (best < 0 || e < beste) { best = s; beste = e; }
    }
    if (best < 0) return 0;
    for (int k = 0; k < 9; k++) R[k] = Rs[9 * best + k];
    for (int k = 0; k < 3; k++) t[k] = ts[3 * best + k];
    return 1;
}
#endif
