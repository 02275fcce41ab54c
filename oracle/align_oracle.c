/* align_oracle.c - CPU checker of bh_align_se3_ransac (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/).
 *
 * What it restates: nothing the reference implements.  slam/loop_closing.py:17-19 declares
 * `_similiarity_transformation(self) -> (R, t)` ("R, t that transforms start of the loop with the end") with body `pass`; SURVEY 8(f)
 * rank 4 names the stage (Sim3 / SE3 alignment of a loop candidate).  The algorithm here is the textbook one for metric (RGB-D)
 * maps - RANSAC over 3-point minimal sets with Horn / Kabsch's closed form (se3_small.h), OpenCV's sampling stream and adaptive
 * iteration count as in the PnP path - stated by this project, so PARITY IS UNPINNED against any reference output; the closed form
 * is checked against numpy's SVD in tests/test_oracle_cpu.py and the kernel must reproduce this file bit for bit.
 *
 *   sampler    cv::RNG((uint64)-1) multiply-with-carry, uniform(0, n) = next() % n, a draw repeated until the three indices differ
 *   model      Kabsch on the three pairs (sums in index order); rejected when the points do not span a plane direction pair
 *   score      number of pairs with |q - (R p + t)|^2 <= thr^2; a model replaces the best one when it has more inliers than
 *              max(best, 2); niters = se_update_iters(confidence, outlier ratio, niters) after every improvement
 *   refinement two rounds of: Kabsch over the current inliers, inliers recomputed.  The sums over the inliers are taken the way a
 *              64-lane wave takes them - lane l adds pairs l, l + 64, ... in ascending order, then the 64 partial sums meet in an
 *              xor butterfly (32, 16, 8, 4, 2, 1) - so that the kernel can be parallel AND bit-identical.
 * status 0 = ok, 1 = fewer than 3 pairs, 2 = no model or fewer than min_inliers inliers. */
#include <stdint.h>
#include <string.h>
#include "se3_small.h"

static unsigned al_rng_next(uint64_t* st) { *st = (uint64_t)(unsigned)(*st) * 4164903690U + (unsigned)(*st >> 32); return (unsigned)(*st); }

/* K sums over the masked pairs in wave order: val(i, k) supplied by the callback */
typedef void (*al_term_fn)(const double* p, const double* q, const double* ctx, double* out);
static void al_wave_sums(const double* P, const double* Q, const unsigned char* mask, int n, int K, al_term_fn fn, const double* ctx, double* out)
{
    double part[64][9], term[9];
    for (int l = 0; l < 64; l++) for (int k = 0; k < K; k++) part[l][k] = 0.0;
    for (int i = 0; i < n; i++) {
        if (!mask[i]) continue;
        fn(P + 3 * (size_t)i, Q + 3 * (size_t)i, ctx, term);
        for (int k = 0; k < K; k++) part[i & 63][k] = part[i & 63][k] + term[k];
    }
    for (int m = 32; m > 0; m >>= 1) {
        double nx[64][9];
        for (int l = 0; l < 64; l++) for (int k = 0; k < K; k++) nx[l][k] = part[l][k] + part[l ^ m][k];
        memcpy(part, nx, sizeof(part));
    }
    for (int k = 0; k < K; k++) out[k] = part[0][k];
}
static void al_term_pq(const double* p, const double* q, const double* ctx, double* out)
{
    (void)ctx; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = q[0]; out[4] = q[1]; out[5] = q[2];
}
static void al_term_h(const double* p, const double* q, const double* c, double* out)      /* c = pbar | qbar */
{
    const double a0 = p[0] - c[0], a1 = p[1] - c[1], a2 = p[2] - c[2], b0 = q[0] - c[3], b1 = q[1] - c[4], b2 = q[2] - c[5];
    out[0] = a0 * b0; out[1] = a0 * b1; out[2] = a0 * b2; out[3] = a1 * b0; out[4] = a1 * b1; out[5] = a1 * b2; out[6] = a2 * b0; out[7] = a2 * b1; out[8] = a2 * b2;
}

static int al_model3(const double* P, const double* Q, const int* idx, double* R, double* t)
{
    double c[6], H[9];
    for (int k = 0; k < 3; k++) {
        c[k] = ((P[3 * (size_t)idx[0] + k] + P[3 * (size_t)idx[1] + k]) + P[3 * (size_t)idx[2] + k]) / 3.0;
        c[3 + k] = ((Q[3 * (size_t)idx[0] + k] + Q[3 * (size_t)idx[1] + k]) + Q[3 * (size_t)idx[2] + k]) / 3.0;
    }
    for (int k = 0; k < 9; k++) H[k] = 0.0;
    for (int s = 0; s < 3; s++) {
        double term[9];
        al_term_h(P + 3 * (size_t)idx[s], Q + 3 * (size_t)idx[s], c, term);
        for (int k = 0; k < 9; k++) H[k] = H[k] + term[k];
    }
    return se_kabsch(H, c, c + 3, R, t);
}

static int al_count(const double* P, const double* Q, int n, const double* R, const double* t, double thr2, unsigned char* mask)
{
    int good = 0;
    for (int i = 0; i < n; i++) { const int in = se_resid2(R, t, P + 3 * (size_t)i, Q + 3 * (size_t)i) <= thr2; if (mask) mask[i] = (unsigned char)in; good += in; }
    return good;
}

/* P, Q f64[n][3]; R row-major 3 x 3, t 3; mask u8[n].  Returns the status. */
int orc_align_se3_ransac(const double* P, const double* Q, int n, int iterations, double thr, double confidence, int min_inliers,
                         double* R, double* t, unsigned char* mask, int* n_inliers)
{
    static const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(R, I3, sizeof(I3)); t[0] = t[1] = t[2] = 0.0; *n_inliers = 0;
    for (int i = 0; i < n; i++) mask[i] = 0;
    if (n < 3) return 1;
    const double thr2 = thr * thr;
    uint64_t st = (uint64_t)-1;
    int best = 0, niters = iterations > 1 ? iterations : 1;
    double bR[9], bt[3];
    for (int it = 0; it < niters; it++) {
        int idx[3];
        for (int i = 0; i < 3; i++) {
            int j;
            for (;;) { idx[i] = (int)(al_rng_next(&st) % (unsigned)n); for (j = 0; j < i; j++) if (idx[i] == idx[j]) break; if (j == i) break; }
        }
        double mR[9], mt[3];
        if (!al_model3(P, Q, idx, mR, mt)) continue;
        const int good = al_count(P, Q, n, mR, mt, thr2, 0);
        if (good > (best > 2 ? best : 2)) {
            best = good; memcpy(bR, mR, sizeof(bR)); memcpy(bt, mt, sizeof(bt));
            niters = se_update_iters(confidence, (double)(n - good) / n, niters);
        }
    }
    if (best == 0) return 2;
    int cnt = al_count(P, Q, n, bR, bt, thr2, mask);
    for (int round = 0; round < 2 && cnt >= 3; round++) {
        double s6[6], c[6], H[9], nR[9], nt[3];
        al_wave_sums(P, Q, mask, n, 6, al_term_pq, 0, s6);
        for (int k = 0; k < 6; k++) c[k] = s6[k] / (double)cnt;
        al_wave_sums(P, Q, mask, n, 9, al_term_h, c, H);
        if (!se_kabsch(H, c, c + 3, nR, nt)) break;
        memcpy(bR, nR, sizeof(bR)); memcpy(bt, nt, sizeof(bt));
        cnt = al_count(P, Q, n, bR, bt, thr2, mask);
    }
    memcpy(R, bR, sizeof(bR)); memcpy(t, bt, sizeof(bt)); *n_inliers = cnt;
    return cnt >= min_inliers && cnt >= 3 ? 0 : 2;
}

/* test hook: the closed form on its own - H (9), pbar (3), qbar (3) -> R, t; returns se_kabsch's verdict */
int orc_se_kabsch(const double* H, const double* pbar, const double* qbar, double* R, double* t) { return se_kabsch(H, pbar, qbar, R, t); }
