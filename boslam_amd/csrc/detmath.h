/*
 * detmath.h — elementary functions with a FIXED operation order (IEEE add/mul/div/sqrt only, no FMA contraction, no libm).
 *
 * Why: solvePnPRansac's refinement and iteration-count update call sin/cos/acos/pow/log.  libm results are not specified to
 * the last bit (glibc and the GPU's ocml differ by 1 ulp in a few per cent of the arguments, measured: tools/pnp_diag.py), and
 * this path feeds hard thresholds and the next frame's input (experiments/pnp_two_way_tracking.py:57-91), so the CPU oracle
 * and the HIP kernels evaluate these five functions with the same statements, one rounding per statement, and agree bit for
 * bit.  Accuracy: within 2 ulp of glibc over the ranges used (tests/test_oracle_cpu.py, test_detmath_tracks_libm).
 * The CPU checker keeps its own copy of this file, byte-identical below the marker line (the product includes nothing from the
 * checker's tree): tests/test_host_cpu.py compares the two.
 */
/* ---- identical from here on ---- */
#ifndef BH_DETMATH_H
#define BH_DETMATH_H
#include <math.h>

#if defined(__HIPCC__)
#define DM_FN __host__ __device__ static inline
#else
#define DM_FN static inline
#endif

/* sin/cos kernels on |x| <= pi/4: Taylor series in z = x*x, Horner form */
DM_FN double dm_sin_k(double x)
{
    const double z = x * x;
    double r = 1.0 / 51090942171709440000.0;            /* 1/21! */
    r = r * z - 1.0 / 121645100408832000.0;             /* 1/19! */
    r = r * z + 1.0 / 355687428096000.0;                /* 1/17! */
    r = r * z - 1.0 / 1307674368000.0;                  /* 1/15! */
    r = r * z + 1.0 / 6227020800.0;                     /* 1/13! */
    r = r * z - 1.0 / 39916800.0;                       /* 1/11! */
    r = r * z + 1.0 / 362880.0;                         /* 1/9! */
    r = r * z - 1.0 / 5040.0;                           /* 1/7! */
    r = r * z + 1.0 / 120.0;                            /* 1/5! */
    r = r * z - 1.0 / 6.0;                              /* 1/3! */
    const double xz = x * z;
    return x + xz * r;
}

DM_FN double dm_cos_k(double x)
{
    const double z = x * x;
    double r = 1.0 / 2432902008176640000.0;             /* 1/20! */
    r = r * z - 1.0 / 6402373705728000.0;               /* 1/18! */
    r = r * z + 1.0 / 20922789888000.0;                 /* 1/16! */
    r = r * z - 1.0 / 87178291200.0;                    /* 1/14! */
    r = r * z + 1.0 / 479001600.0;                      /* 1/12! */
    r = r * z - 1.0 / 3628800.0;                        /* 1/10! */
    r = r * z + 1.0 / 40320.0;                          /* 1/8! */
    r = r * z - 1.0 / 720.0;                            /* 1/6! */
    r = r * z + 1.0 / 24.0;                             /* 1/4! */
    const double zz = z * z;
    const double hz = 0.5 * z;
    return (1.0 - hz) + zz * r;
}

/* sin and cos of x (|x| < ~1e5 for full accuracy): Cody-Waite reduction by pi/2 in three pieces, then the kernels */
DM_FN void dm_sincos(double x, double* s, double* c)
{
    const double k = rint(x * 6.36619772367581382433e-01);                  /* x * 2/pi */
    double y = x - k * 1.57079632673412561417e+00;                         /* first 33 bits of pi/2 */
    y = y - k * 6.07710050630396597660e-11;                                /* next 33 bits */
    y = y - k * 2.02226624879595063154e-21;                                /* rest */
    const double sk = dm_sin_k(y), ck = dm_cos_k(y);
    const double q = k - 4.0 * floor(k * 0.25);                            /* k mod 4 in {0,1,2,3} */
    if (q == 0.0) { *s = sk; *c = ck; }
    else if (q == 1.0) { *s = ck; *c = -sk; }
    else if (q == 2.0) { *s = -sk; *c = -ck; }
    else { *s = -ck; *c = sk; }
}

/* asin(s) = s + s*z*R(z), z = s*s <= 0.25: Taylor coefficients a_k = (2k)! / (4^k k!^2 (2k+1)), built by their recurrence */
DM_FN double dm_asin_r(double z)
{
    double a[28];
    double t = 1.0;                                                        /* (2k)! / (4^k k!^2) */
    for (int k = 1; k <= 27; k++) {
        t = t * (double)(2 * k - 1) / (double)(2 * k);
        a[k] = t / (double)(2 * k + 1);
    }
    double r = a[27];
    for (int k = 26; k >= 1; k--) r = r * z + a[k];
    return r;
}

DM_FN double dm_acos(double c)
{
    const double pi = 3.14159265358979311600e+00, pio2 = 1.57079632679489655800e+00;
    if (c >= 1.0) return 0.0;
    if (c <= -1.0) return pi;
    if (c > 0.5) {
        const double z = (1.0 - c) * 0.5;
        const double s = sqrt(z);
        const double w = s * z * dm_asin_r(z);
        return 2.0 * (s + w);
    }
    if (c < -0.5) {
        const double z = (1.0 + c) * 0.5;
        const double s = sqrt(z);
        const double w = s * z * dm_asin_r(z);
        return pi - 2.0 * (s + w);
    }
    const double z = c * c;
    const double w = c * z * dm_asin_r(z);
    return pio2 - (c + w);
}

/* natural logarithm of a positive normal double: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh((m-1)/(m+1)) */
DM_FN double dm_log(double x)
{
    int e;
    double m = frexp(x, &e);                                               /* exact */
    if (m < 7.07106781186547572737e-01) { m = m * 2.0; e = e - 1; }
    const double t = (m - 1.0) / (m + 1.0);
    const double w = t * t;
    double r = 1.0 / 27.0;
    r = r * w + 1.0 / 25.0;
    r = r * w + 1.0 / 23.0;
    r = r * w + 1.0 / 21.0;
    r = r * w + 1.0 / 19.0;
    r = r * w + 1.0 / 17.0;
    r = r * w + 1.0 / 15.0;
    r = r * w + 1.0 / 13.0;
    r = r * w + 1.0 / 11.0;
    r = r * w + 1.0 / 9.0;
    r = r * w + 1.0 / 7.0;
    r = r * w + 1.0 / 5.0;
    r = r * w + 1.0 / 3.0;
    const double tw = t * w;
    const double lm = 2.0 * (t + tw * r);
    const double de = (double)e;
    return (de * 6.93147180369123816490e-01 + lm) + de * 1.90821492927058770002e-10;
}

DM_FN double dm_pow5(double x)
{
    const double x2 = x * x;
    const double x4 = x2 * x2;
    return x4 * x;
}
#endif
