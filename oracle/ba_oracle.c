/*
 * ba_oracle.c — CPU restatement of the g2o graph the reference builds in
 *   slam/optimizer.py:82-142   CamOnlyBA      (called at slam/tracking.py:73,128)
 *   slam/optimizer.py:8-79     BundleAdjustment (driven by slam/covisibility_graph.py:186-235 optimize_local)
 *
 * TEST INFRASTRUCTURE (see orb_oracle.c).  PARITY UNPINNED: the arithmetic lives in g2o (uoip/g2opy HEAD, unpinned,
 * docker/Dockerfile:84), absent from /root/reference and from this image; the reference has no BA known-answer
 * test (tests/ba_test.py is a visual demo).  This file restates g2o's published algorithm:
 *
 *   vertices  VertexCam (estimate SBACam = SE3Quat camera-in-world + Kcam; oplus: t += d[0:3];
 *             q = normalize(q * (d[3:6], sqrt(1-|d[3:6]|^2))); derived w2n=[R'|-R't], w2i=K*w2n, dRd{x,y,z}),
 *             VertexSBAPointXYZ (oplus: p += d)
 *   edge      EdgeProjectP2MC: e = (w2i*[X;1]).xy/z - meas; analytic Jacobians (types_sba.cpp linearizeOplus)
 *   robust    RobustKernelHuber(delta): rho'=1 if chi2<=delta^2 else delta/sqrt(chi2); H += J'(rho' I)J, b -= J' rho' e
 *   solver    BlockSolver<6,3>: Hpp/Hll/Hpl, Schur complement on the marginalised points, Cholesky of the reduced
 *             camera system (g2o: CSparse sparse Cholesky; here dense — same solution up to rounding),
 *             back-substitution for the points
 *   LM        OptimizationAlgorithmLevenberg: lambda0 = 1e-5 * max diag(H); per outer iteration <=10 trials with
 *             push / solve(H + lambda I) / oplus / chi2 ; rho = (chi - chi_new)/(sum dx (lambda dx + b) + 1e-3);
 *             accept: lambda *= max(1/3, min(2/3, 1-(2rho-1)^3)), ni=2 ; reject: pop, lambda *= ni, ni *= 2;
 *             loop while rho<0; Terminate (stop iterating) if 10 trials were used or rho == 0
 *   errors    after optimize() each edge holds the error of the LAST computeActiveErrors call: the last trial
 *             (CamOnlyBA, verbose off) or the current estimate (BundleAdjustment.optimize: flag_verbose=True makes
 *             SparseOptimizer::optimize recompute the active errors after every iteration)
 *
 * Information matrix is I2 (the only value the reference passes).  Vertices without an active edge, and edges
 * whose vertices are all fixed, take no part (g2o initializeOptimization).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

typedef struct {
    double q[4];            /* x y z w */
    double t[3];
    double w2n[12];         /* 3x4 row-major */
    double w2i[12];
    double dR[3][9];        /* dRdx, dRdy, dRdz row-major */
} orc_cam;

static void quat_from_R(const double* m /* row-major 3x3 */, double* q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i*3+i] - m[j*3+j] - m[k*3+k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[k*3+j] - m[j*3+k]) * t;
        q[j] = (m[j*3+i] + m[i*3+j]) * t;
        q[k] = (m[k*3+i] + m[i*3+k]) * t;
    }
}
static void quat_normalize(double* q)
{
    double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat_to_R(const double* q, double* R)
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2*x, ty = 2*y, tz = 2*z;
    double twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    R[0] = 1-(tyy+tzz); R[1] = txy-twz;     R[2] = txz+twy;
    R[3] = txy+twz;     R[4] = 1-(txx+tzz); R[5] = tyz-twx;
    R[6] = txz-twy;     R[7] = tyz+twx;     R[8] = 1-(txx+tyy);
}

static void cam_refresh(orc_cam* c, const double* K)
{
    double R[9]; quat_to_R(c->q, R);
    double* n = c->w2n;
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) n[r*4+k] = R[k*3+r];            /* R' */
    for (int r = 0; r < 3; r++) n[r*4+3] = -(n[r*4+0]*c->t[0] + n[r*4+1]*c->t[1] + n[r*4+2]*c->t[2]);
    double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    for (int k = 0; k < 4; k++) {
        c->w2i[k]   = fx * n[k] + cx * n[8+k];
        c->w2i[4+k] = fy * n[4+k] + cy * n[8+k];
        c->w2i[8+k] = n[8+k];
    }
    /* dRdx = [[0,0,0],[0,0,2],[0,-2,0]] * R' ; dRdy = [[0,0,-2],[0,0,0],[2,0,0]] * R' ; dRdz = [[0,2,0],[-2,0,0],[0,0,0]] * R' */
    for (int k = 0; k < 3; k++) {
        double r0 = n[k], r1 = n[4+k], r2 = n[8+k];
        c->dR[0][k] = 0;       c->dR[0][3+k] = 2*r2;  c->dR[0][6+k] = -2*r1;
        c->dR[1][k] = -2*r2;   c->dR[1][3+k] = 0;     c->dR[1][6+k] = 2*r0;
        c->dR[2][k] = 2*r1;    c->dR[2][3+k] = -2*r0; c->dR[2][6+k] = 0;
    }
}

static void cam_init(orc_cam* c, const double* R, const double* t, const double* K)
{
    quat_from_R(R, c->q);
    if (c->q[3] < 0) for (int i = 0; i < 4; i++) c->q[i] = -c->q[i];       /* SE3Quat::normalizeRotation */
    quat_normalize(c->q);
    memcpy(c->t, t, sizeof(double) * 3);
    cam_refresh(c, K);
}

static void cam_oplus(orc_cam* c, const double* d, const double* K)
{
    c->t[0] += d[0]; c->t[1] += d[1]; c->t[2] += d[2];
    double bx = d[3], by = d[4], bz = d[5];
    double bw = sqrt(1.0 - (bx*bx + by*by + bz*bz));           /* NaN if the step is longer than 1: propagates */
    double ax = c->q[0], ay = c->q[1], az = c->q[2], aw = c->q[3];
    c->q[3] = aw*bw - ax*bx - ay*by - az*bz;
    c->q[0] = aw*bx + ax*bw + ay*bz - az*by;
    c->q[1] = aw*by + ay*bw + az*bx - ax*bz;
    c->q[2] = aw*bz + az*bw + ax*by - ay*bx;
    quat_normalize(c->q);
    cam_refresh(c, K);
}

/* error of one edge */
static void edge_error(const orc_cam* c, const double* X, const double* z, double* e)
{
    const double* w = c->w2i;
    double p0 = w[0]*X[0] + w[1]*X[1] + w[2]*X[2] + w[3];
    double p1 = w[4]*X[0] + w[5]*X[1] + w[6]*X[2] + w[7];
    double p2 = w[8]*X[0] + w[9]*X[1] + w[10]*X[2] + w[11];
    e[0] = p0 / p2 - z[0];
    e[1] = p1 / p2 - z[1];
}

/* Jacobians: Jc 2x6 (tx ty tz qx qy qz), Jp 2x3 */
static void edge_jac(const orc_cam* c, const double* X, const double* K, double* Jc, double* Jp)
{
    const double* n = c->w2n;
    double px = n[0]*X[0] + n[1]*X[1] + n[2]*X[2] + n[3];
    double py = n[4]*X[0] + n[5]*X[1] + n[6]*X[2] + n[7];
    double pz = n[8]*X[0] + n[9]*X[1] + n[10]*X[2] + n[11];
    double ipz2 = 1.0 / (pz * pz);
    double ipz2fx = ipz2 * K[0], ipz2fy = ipz2 * K[1];
    double pwt[3] = { X[0] - c->t[0], X[1] - c->t[1], X[2] - c->t[2] };
    for (int a = 0; a < 3; a++) {
        const double* D = c->dR[a];
        double dp0 = D[0]*pwt[0] + D[1]*pwt[1] + D[2]*pwt[2];
        double dp1 = D[3]*pwt[0] + D[4]*pwt[1] + D[5]*pwt[2];
        double dp2 = D[6]*pwt[0] + D[7]*pwt[1] + D[8]*pwt[2];
        Jc[3+a]   = (pz*dp0 - px*dp2) * ipz2fx;
        Jc[6+3+a] = (pz*dp1 - py*dp2) * ipz2fy;
        double d0 = -n[a], d1 = -n[4+a], d2 = -n[8+a];
        Jc[a]   = (pz*d0 - px*d2) * ipz2fx;
        Jc[6+a] = (pz*d1 - py*d2) * ipz2fy;
        d0 = n[a]; d1 = n[4+a]; d2 = n[8+a];
        Jp[a]   = (pz*d0 - px*d2) * ipz2fx;
        Jp[3+a] = (pz*d1 - py*d2) * ipz2fy;
    }
}

static void huber(double e2, double delta, double* rho0, double* rho1)
{
    if (delta <= 0) { *rho0 = e2; *rho1 = 1.0; return; }
    double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { double s = sqrt(e2); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

/* dense Cholesky solve A x = b (A n x n SPD, row-major, destroyed); returns 0 on failure */
static int chol_solve(double* A, double* b, int n)
{
    for (int j = 0; j < n; j++) {
        double d = A[j*n+j];
        for (int k = 0; k < j; k++) d -= A[j*n+k] * A[j*n+k];
        if (!(d > 0) || !isfinite(d)) return 0;
        d = sqrt(d); A[j*n+j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i*n+j];
            for (int k = 0; k < j; k++) s -= A[i*n+k] * A[j*n+k];
            A[i*n+j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i*n+k] * b[k]; b[i] = s / A[i*n+i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[k*n+i] * b[k]; b[i] = s / A[i*n+i]; }
    return 1;
}

static int inv3(const double* m, double* o)
{
    double c00 = m[4]*m[8]-m[5]*m[7], c01 = m[5]*m[6]-m[3]*m[8], c02 = m[3]*m[7]-m[4]*m[6];
    double det = m[0]*c00 + m[1]*c01 + m[2]*c02;
    double id = 1.0 / det;
    o[0] = c00*id; o[1] = (m[2]*m[7]-m[1]*m[8])*id; o[2] = (m[1]*m[5]-m[2]*m[4])*id;
    o[3] = c01*id; o[4] = (m[0]*m[8]-m[2]*m[6])*id; o[5] = (m[2]*m[3]-m[0]*m[5])*id;
    o[6] = c02*id; o[7] = (m[1]*m[6]-m[0]*m[7])*id; o[8] = (m[0]*m[4]-m[1]*m[3])*id;
    return isfinite(id);
}

/*
 * R,t: C poses (row-major R), in/out.  cam_fixed u8[C].  pts P x 3 in/out, pt_fixed u8[P] or NULL (all free).
 * edges: ecam/ept/meas.  K={fx,fy,cx,cy,baseline}.  huber_delta<=0: no kernel.
 * chi2_out f64[E]; log f64[iters][4] = {robust chi2 after the iteration, lambda, trials, accepted}; may be NULL.
 * returns 0 ok, -1 bad input, -5 if the first linearisation is not finite.
 */
/* 0 (default): (2 rho - 1)^3 by pow() as g2o's OptimizationAlgorithmLevenberg does; 1: by two rounded multiplications. */
static int orc_cube_two_mul = 0;
int orc_ba_cube_mode(int two_mul) { const int old = orc_cube_two_mul; if (two_mul >= 0) orc_cube_two_mul = two_mul ? 1 : 0; return old; }

int orc_ba_optimize(int C, double* R, double* t, const uint8_t* cam_fixed, int P, double* pts, const uint8_t* pt_fixed,
                    int E, const int* ecam, const int* ept, const double* meas, const double* K,
                    int iters, double huber_delta, int verbose_errors, double* chi2_out, double* log, int* iters_done)
{
    if (C < 0 || P < 0 || E < 0) return -1;
    orc_cam* cams = calloc(C > 0 ? C : 1, sizeof(orc_cam));
    orc_cam* cams_bak = calloc(C > 0 ? C : 1, sizeof(orc_cam));
    double* pts_bak = malloc(sizeof(double) * 3 * (P > 0 ? P : 1));
    for (int i = 0; i < C; i++) cam_init(&cams[i], R + 9*i, t + 3*i, K);
    /* active structure */
    int* cidx = malloc(sizeof(int) * (C + 1)); int* pidx = malloc(sizeof(int) * (P + 1));
    uint8_t* eact = malloc(E + 1);
    for (int i = 0; i < C; i++) cidx[i] = -1;
    for (int j = 0; j < P; j++) pidx[j] = -1;
    for (int e = 0; e < E; e++) {
        if (ecam[e] < 0 || ecam[e] >= C || ept[e] < 0 || ept[e] >= P) { free(cams); free(cams_bak); free(pts_bak); free(cidx); free(pidx); free(eact); return -1; }
        int cf = cam_fixed && cam_fixed[ecam[e]], pf = pt_fixed && pt_fixed[ept[e]];
        eact[e] = !(cf && pf);
        if (eact[e]) { if (!cf) cidx[ecam[e]] = 0; if (!pf) pidx[ept[e]] = 0; }
    }
    int nc = 0, np = 0;
    for (int i = 0; i < C; i++) if (cidx[i] == 0) cidx[i] = nc++;
    for (int j = 0; j < P; j++) if (pidx[j] == 0) pidx[j] = np++;
    int n6 = 6 * nc;
    double* Hpp = calloc((size_t)(n6 > 0 ? n6 * n6 : 1), sizeof(double));     /* only diagonal blocks are filled */
    double* bp = calloc(n6 + 1, sizeof(double));
    double* Hll = calloc((size_t)9 * (np + 1), sizeof(double));
    double* bl = calloc((size_t)3 * (np + 1), sizeof(double));
    double* Hpl = calloc((size_t)18 * (E + 1), sizeof(double));               /* per edge, 6x3 */
    double* S = malloc(sizeof(double) * (size_t)(n6 > 0 ? n6 * n6 : 1));
    double* dxp = malloc(sizeof(double) * (n6 + 1)); double* dxl = malloc(sizeof(double) * 3 * (np + 1));
    double* Dinv = malloc(sizeof(double) * 9 * (np + 1));
    double* err = calloc((size_t)2 * (E + 1), sizeof(double));
    /* per-point edge lists */
    int* pstart = calloc(np + 2, sizeof(int)); int* plist = malloc(sizeof(int) * (E + 1));
    for (int e = 0; e < E; e++) if (eact[e] && pidx[ept[e]] >= 0) pstart[pidx[ept[e]] + 1]++;
    for (int j = 0; j < np; j++) pstart[j+1] += pstart[j];
    { int* fill = calloc(np + 1, sizeof(int));
      for (int e = 0; e < E; e++) if (eact[e] && pidx[ept[e]] >= 0) { int j = pidx[ept[e]]; plist[pstart[j] + fill[j]++] = e; }
      free(fill); }

    double lambda = 0, ni = 2;
    int it_done = 0, rc = 0;
    for (int it = 0; it < iters; it++) {
        /* computeActiveErrors + activeRobustChi2 */
        double chi = 0;
        for (int e = 0; e < E; e++) if (eact[e]) {
            edge_error(&cams[ecam[e]], pts + 3*ept[e], meas + 2*e, err + 2*e);
            double r0, r1; huber(err[2*e]*err[2*e] + err[2*e+1]*err[2*e+1], huber_delta, &r0, &r1); chi += r0;
        }
        /* buildSystem */
        memset(Hpp, 0, sizeof(double) * (size_t)n6 * n6); memset(bp, 0, sizeof(double) * n6);
        memset(Hll, 0, sizeof(double) * 9 * np); memset(bl, 0, sizeof(double) * 3 * np);
        memset(Hpl, 0, sizeof(double) * 18 * (size_t)E);
        for (int e = 0; e < E; e++) if (eact[e]) {
            double Jc[12], Jp[6];
            edge_jac(&cams[ecam[e]], pts + 3*ept[e], K, Jc, Jp);
            double e0 = err[2*e], e1 = err[2*e+1], r0, r1;
            huber(e0*e0 + e1*e1, huber_delta, &r0, &r1);
            int ci = cidx[ecam[e]], pj = pidx[ept[e]];
            if (ci >= 0) {
                for (int a = 0; a < 6; a++) {
                    bp[6*ci+a] -= r1 * (Jc[a]*e0 + Jc[6+a]*e1);
                    for (int b = 0; b < 6; b++) Hpp[(size_t)(6*ci+a)*n6 + 6*ci+b] += r1 * (Jc[a]*Jc[b] + Jc[6+a]*Jc[6+b]);
                }
            }
            if (pj >= 0) {
                for (int a = 0; a < 3; a++) {
                    bl[3*pj+a] -= r1 * (Jp[a]*e0 + Jp[3+a]*e1);
                    for (int b = 0; b < 3; b++) Hll[9*pj + 3*a + b] += r1 * (Jp[a]*Jp[b] + Jp[3+a]*Jp[3+b]);
                }
            }
            if (ci >= 0 && pj >= 0)
                for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++) Hpl[18*(size_t)e + 3*a + b] = r1 * (Jc[a]*Jp[b] + Jc[6+a]*Jp[3+b]);
        }
        if (it == 0) {
            double md = 0;
            for (int i = 0; i < n6; i++) md = fmax(fabs(Hpp[(size_t)i*n6+i]), md);
            for (int j = 0; j < np; j++) for (int a = 0; a < 3; a++) md = fmax(fabs(Hll[9*j+4*a]), md);
            if (!isfinite(md)) { rc = -5; break; }
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0; int qmax = 0, accepted = 0;
        double tempChi = chi;
        do {
            memcpy(cams_bak, cams, sizeof(orc_cam) * C); memcpy(pts_bak, pts, sizeof(double) * 3 * P);   /* push */
            /* solve (H + lambda I) dx = b with Schur */
            int ok2 = 1;
            for (size_t i = 0; i < (size_t)n6 * n6; i++) S[i] = Hpp[i];
            for (int i = 0; i < n6; i++) S[(size_t)i*n6+i] += lambda;
            for (int i = 0; i < n6; i++) dxp[i] = bp[i];
            for (int j = 0; j < np; j++) {
                double D[9]; memcpy(D, Hll + 9*j, sizeof(D));
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                if (!inv3(D, Dinv + 9*j)) ok2 = 0;
                const double* Di = Dinv + 9*j;
                double db[3];
                for (int a = 0; a < 3; a++) db[a] = Di[3*a]*bl[3*j] + Di[3*a+1]*bl[3*j+1] + Di[3*a+2]*bl[3*j+2];
                for (int u = pstart[j]; u < pstart[j+1]; u++) {
                    int e1 = plist[u]; int c1 = cidx[ecam[e1]]; if (c1 < 0) continue;
                    const double* B1 = Hpl + 18*(size_t)e1;
                    double BD[18];
                    for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++)
                        BD[3*a+b] = B1[3*a]*Di[b] + B1[3*a+1]*Di[3+b] + B1[3*a+2]*Di[6+b];
                    for (int a = 0; a < 6; a++) dxp[6*c1+a] -= B1[3*a]*db[0] + B1[3*a+1]*db[1] + B1[3*a+2]*db[2];
                    for (int v = pstart[j]; v < pstart[j+1]; v++) {
                        int e2 = plist[v]; int c2 = cidx[ecam[e2]]; if (c2 < 0) continue;
                        const double* B2 = Hpl + 18*(size_t)e2;
                        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++)
                            S[(size_t)(6*c1+a)*n6 + 6*c2+b] -= BD[3*a]*B2[3*b] + BD[3*a+1]*B2[3*b+1] + BD[3*a+2]*B2[3*b+2];
                    }
                }
            }
            if (n6 > 0 && ok2) ok2 = chol_solve(S, dxp, n6);
            for (int j = 0; j < np; j++) {
                double c[3] = { bl[3*j], bl[3*j+1], bl[3*j+2] };
                for (int u = pstart[j]; u < pstart[j+1]; u++) {
                    int e1 = plist[u]; int c1 = cidx[ecam[e1]]; if (c1 < 0) continue;
                    const double* B1 = Hpl + 18*(size_t)e1;
                    for (int b = 0; b < 3; b++) for (int a = 0; a < 6; a++) c[b] -= B1[3*a+b] * dxp[6*c1+a];
                }
                const double* Di = Dinv + 9*j;
                for (int a = 0; a < 3; a++) dxl[3*j+a] = Di[3*a]*c[0] + Di[3*a+1]*c[1] + Di[3*a+2]*c[2];
            }
            /* update */
            for (int i = 0; i < C; i++) if (cidx[i] >= 0) cam_oplus(&cams[i], dxp + 6*cidx[i], K);
            for (int j = 0; j < P; j++) if (pidx[j] >= 0) for (int a = 0; a < 3; a++) pts[3*j+a] += dxl[3*pidx[j]+a];
            /* computeActiveErrors */
            tempChi = 0;
            for (int e = 0; e < E; e++) if (eact[e]) {
                edge_error(&cams[ecam[e]], pts + 3*ept[e], meas + 2*e, err + 2*e);
                double r0, r1; huber(err[2*e]*err[2*e] + err[2*e+1]*err[2*e+1], huber_delta, &r0, &r1); tempChi += r0;
            }
            if (!ok2) tempChi = DBL_MAX;
            rho = chi - tempChi;
            double scale = 0;
            for (int i = 0; i < n6; i++) scale += dxp[i] * (lambda * dxp[i] + bp[i]);
            for (int i = 0; i < 3*np; i++) scale += dxl[i] * (lambda * dxl[i] + bl[i]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                /* g2o: alpha = 1. - pow((2*rho-1), 3) - libm's pow, the default.  orc_ba_cube_mode(1) selects two rounded
                   multiplications instead (<= 1 ulp away): tests/test_oracle_cpu.py checks that the LM decisions do not depend on it. */
                const double q1_ = 2*rho - 1;
                double q3_;
                if (orc_cube_two_mul) { volatile double q2_ = q1_ * q1_; volatile double q3v_ = q2_ * q1_; q3_ = q3v_; }
                else q3_ = pow(q1_, 3);
                double alpha = 1. - q3_;
                alpha = fmin(alpha, 2./3.);
                double sf = fmax(1./3., alpha);
                lambda *= sf; ni = 2; chi = tempChi; accepted = 1;
            } else {
                lambda *= ni; ni *= 2;
                memcpy(cams, cams_bak, sizeof(orc_cam) * C); memcpy(pts, pts_bak, sizeof(double) * 3 * P);   /* pop */
                accepted = 0;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (verbose_errors)
            for (int e = 0; e < E; e++) if (eact[e]) edge_error(&cams[ecam[e]], pts + 3*ept[e], meas + 2*e, err + 2*e);
        if (log) { log[4*it] = accepted ? chi : tempChi; log[4*it+1] = lambda; log[4*it+2] = qmax; log[4*it+3] = accepted; }
        it_done = it + 1;
        if (qmax == 10 || rho == 0) break;       /* OptimizationAlgorithm::Terminate */
    }
    for (int e = 0; e < E; e++) chi2_out[e] = eact[e] ? err[2*e]*err[2*e] + err[2*e+1]*err[2*e+1] : 0.0;
    for (int i = 0; i < C; i++) { quat_to_R(cams[i].q, R + 9*i); memcpy(t + 3*i, cams[i].t, sizeof(double) * 3); }
    if (iters_done) *iters_done = it_done;
    free(cams); free(cams_bak); free(pts_bak); free(cidx); free(pidx); free(eact); free(Hpp); free(bp); free(Hll); free(bl);
    free(Hpl); free(S); free(dxp); free(dxl); free(Dinv); free(err); free(pstart); free(plist);
    return rc;
}

/* CamOnlyBA (optimizer.py:82-142): one free camera initialised to identity, n fixed points; verbose off. */
int orc_pose_only_ba(const double* pix, const double* pts, int n, const double* K, int iters, double huber_delta,
                     double chi2_thr, double* R, double* t, double* chi2, uint8_t* inlier)
{
    double Rt[9] = {1,0,0, 0,1,0, 0,0,1}, tt[3] = {0,0,0};
    uint8_t cf = 0;
    uint8_t* pf = malloc(n + 1); memset(pf, 1, n + 1);
    int* ec = calloc(n + 1, sizeof(int)); int* ep = malloc(sizeof(int) * (n + 1));
    for (int i = 0; i < n; i++) ep[i] = i;
    double* p = malloc(sizeof(double) * 3 * (n + 1)); memcpy(p, pts, sizeof(double) * 3 * n);
    int done = 0;
    int rc = orc_ba_optimize(1, Rt, tt, &cf, n, p, pf, n, ec, ep, pix, K, iters, huber_delta, 0, chi2, NULL, &done);
    memcpy(R, Rt, sizeof(Rt)); memcpy(t, tt, sizeof(tt));
    for (int i = 0; i < n; i++) inlier[i] = chi2[i] <= chi2_thr;
    free(pf); free(ec); free(ep); free(p);
    return rc;
}
