// loop.hip - the two geometric stages behind bh_align_se3_ransac in the loop-closure thread: GlobalMapManager._loop_fusion and
// GlobalMapManager._essential_graph_optimization (slam/loop_closing.py:21-29).
//
// Reference: both bodies are `pass`: PARITY UNPINNED BY CONSTRUCTION - there is no reference behaviour, the
// algorithms are specified by this project and checked against CPU restatements of that specification
// (oracle.loop_fuse_pairs, oracle.pose_graph_optimize).
//
// bh_loop_fuse_pairs ("finds duplicated mappoints and fuses them together", :21-26): the map points seen from the loop's two ends - side A
// around the current keyframe, side B around the loop candidate - with (R, t) of _similiarity_transformation (a ~ R b + t).  A pair
// (i, j) is a duplicate when the cross-check Hamming matcher pairs feat_a[i] with feat_b[j] (cv2.BFMatcher(NORM_HAMMING, crossCheck=True), the
// matcher every other stage of the reference uses), their distance is below d_max (config.d_hamming_max, `<` as slam/tracking.py:57) and
// |pt_a[i] - (R pt_b[j] + t)| <= radius.  The survivor of a pair is the point with more observations (A on a tie): keep_a.  Pairs come
// out in ascending i.  The graph surgery itself (re-pointing edges_kf2mps / edges_mp2kfs) stays with the host's graph.
#include "bh_internal.h"
#include <vector>

__global__ __launch_bounds__(256) void k_fuse_filter(const int* __restrict__ rq, const int* __restrict__ rt, const int* __restrict__ rd, const int* __restrict__ rn, int raw_cap,
                                                     const double* __restrict__ pt_a, const double* __restrict__ pt_b, const int* __restrict__ nobs_a,
                                                     const int* __restrict__ nobs_b, const double* __restrict__ Rt /* R[9] t[3] */, int d_max, double radius2,
                                                     int* __restrict__ pair_a, int* __restrict__ pair_b, uint8_t* __restrict__ keep_a, int* __restrict__ n_pairs, int cap)
{
    __shared__ int s_wsum[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(rn[0], raw_cap);
    const double R0 = Rt[0], R1 = Rt[1], R2 = Rt[2], R3 = Rt[3], R4 = Rt[4], R5 = Rt[5], R6 = Rt[6], R7 = Rt[7], R8 = Rt[8], t0 = Rt[9], t1 = Rt[10], t2 = Rt[11];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        int a = 0, b = 0;
        bool keep = false;
        if (i < n) {
            a = rq[i]; b = rt[i];
            if (rd[i] < d_max) {
                const double bx = pt_b[3 * (size_t)b], by = pt_b[3 * (size_t)b + 1], bz = pt_b[3 * (size_t)b + 2];
                const double dx = pt_a[3 * (size_t)a] - (((R0 * bx) + (R1 * by)) + (R2 * bz) + t0);
                const double dy = pt_a[3 * (size_t)a + 1] - (((R3 * bx) + (R4 * by)) + (R5 * bz) + t1);
                const double dz = pt_a[3 * (size_t)a + 2] - (((R6 * bx) + (R7 * by)) + (R8 * bz) + t2);
                keep = ((dx * dx) + (dy * dy)) + (dz * dz) <= radius2;
            }
        }
        const unsigned long long bm = __ballot(keep);
        const int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wv] = __popcll(bm);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; w++) off += s_wsum[w];
        const int pos = off + before;
        if (keep && pos < cap) { pair_a[pos] = a; pair_b[pos] = b; keep_a[pos] = nobs_a[a] >= nobs_b[b] ? 1 : 0; }
        __syncthreads();
        if (tid == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    if (tid == 0) n_pairs[0] = min(s_base, cap);
}

extern "C" int bh_loop_fuse_pairs(bh_ctx* ctx, const uint8_t* feat_a, const double* pt_a, const int* nobs_a, int Na,
                                  const uint8_t* feat_b, const double* pt_b, const int* nobs_b, int Nb,
                                  const double R[9], const double t[3], int d_max, double radius,
                                  int* pair_a, int* pair_b, uint8_t* keep_a, int* n_pairs, int cap, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!feat_a || !pt_a || !nobs_a || !feat_b || !pt_b || !nobs_b || !R || !t || !pair_a || !pair_b || !keep_a || !n_pairs || Na < 0 || Nb < 0 || cap < 1 ||
        !(radius >= 0) || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_loop_fuse_pairs: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t na = (size_t)(Na > 0 ? Na : 1), nb = (size_t)(Nb > 0 ? Nb : 1), raw = na < nb ? na : nb;
    const size_t w_b = al(sizeof(int) * 2) + 3 * al(sizeof(int) * raw) + al(sizeof(int)) + al(sizeof(double) * 12);
    const size_t in_b = mem == BH_MEM_HOST ? al(na * 32) + al(nb * 32) + al(sizeof(double) * 3 * na) + al(sizeof(double) * 3 * nb) + al(sizeof(int) * na) + al(sizeof(int) * nb) +
                                             2 * al(sizeof(int) * (size_t)cap) + al((size_t)cap) + al(sizeof(int)) : 0;
    BH_TRY(bh_reserve(ctx, &ctx->st_h, w_b + in_b));
    char* p = (char*)ctx->st_h.p;
    int* d_cnt = (int*)p; p += al(sizeof(int) * 2);                // nq, nt of the one matcher pair
    int* rq = (int*)p; p += al(sizeof(int) * raw); int* rt = (int*)p; p += al(sizeof(int) * raw); int* rd = (int*)p; p += al(sizeof(int) * raw);
    int* rn = (int*)p; p += al(sizeof(int));
    double* d_Rt = (double*)p; p += al(sizeof(double) * 12);
    const int hc[2] = { Na, Nb };
    double hRt[12]; memcpy(hRt, R, sizeof(double) * 9); memcpy(hRt + 9, t, sizeof(double) * 3);
    BH_HIP(ctx, hipMemcpyAsync(d_cnt, hc, sizeof(hc), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(d_Rt, hRt, sizeof(hRt), hipMemcpyHostToDevice, st));
    const uint8_t *d_fa = feat_a, *d_fb = feat_b; const double *d_pa = pt_a, *d_pb = pt_b; const int *d_na = nobs_a, *d_nb = nobs_b;
    int *d_qa = pair_a, *d_qb = pair_b, *d_np = n_pairs; uint8_t* d_ka = keep_a;
    if (mem == BH_MEM_HOST) {
        auto up = [&](const void* src, size_t bytes, size_t room) -> char* { char* d = p; p += al(room); return (bytes == 0 || hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess) ? d : nullptr; };
        char* q;
#define LF_UP(dst, type, src, bytes, room) if (!(q = up(src, bytes, room))) return bh_set_err(ctx, BH_ERR_HIP, "bh_loop_fuse_pairs: upload failed"); dst = (type)q
        LF_UP(d_fa, const uint8_t*, feat_a, (size_t)Na * 32, na * 32); LF_UP(d_fb, const uint8_t*, feat_b, (size_t)Nb * 32, nb * 32);
        LF_UP(d_pa, const double*, pt_a, sizeof(double) * 3 * (size_t)Na, sizeof(double) * 3 * na); LF_UP(d_pb, const double*, pt_b, sizeof(double) * 3 * (size_t)Nb, sizeof(double) * 3 * nb);
        LF_UP(d_na, const int*, nobs_a, sizeof(int) * (size_t)Na, sizeof(int) * na); LF_UP(d_nb, const int*, nobs_b, sizeof(int) * (size_t)Nb, sizeof(int) * nb);
#undef LF_UP
        d_qa = (int*)p; p += al(sizeof(int) * (size_t)cap); d_qb = (int*)p; p += al(sizeof(int) * (size_t)cap); d_ka = (uint8_t*)p; p += al((size_t)cap); d_np = (int*)p;
    }
    BH_TRY(bh_match_hamming_crosscheck(ctx, d_fa, d_cnt, (int)na, d_fb, d_cnt + 1, (int)nb, 1, rq, rt, rd, rn, (int)raw, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_MATCH, k_fuse_filter, dim3(1), dim3(256), st, (const int*)rq, (const int*)rt, (const int*)rd, (const int*)rn, (int)raw, d_pa, d_pb, d_na, d_nb,
              (const double*)d_Rt, d_max, radius * radius, d_qa, d_qb, d_ka, d_np, cap);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(n_pairs, d_np, sizeof(int), hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(pair_a, d_qa, sizeof(int) * (size_t)cap, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(pair_b, d_qb, sizeof(int) * (size_t)cap, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(keep_a, d_ka, (size_t)cap, hipMemcpyDeviceToHost, st));
        return bh_ctx_status(ctx);
    }
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ essential-graph optimisation
// bh_pose_graph_optimize (GlobalMapManager._essential_graph_optimization, slam/loop_closing.py:28-29, body `pass`; specified by this
// project, checked against oracle.pose_graph_optimize): keyframe poses (R_i, t_i) (camera in world, the convention of every other
// stage) and relative-pose edges Z_ij ~ T_i^-1 T_j - the spanning tree and the strong covisibility edges measured before the loop was
// closed, plus the loop edge from bh_align_se3_ransac.  Minimise sum_e w_e |r_e|^2 with r_e = (t_E, log R_E), E = Z^-1 T_i^-1 T_j, over
// the free poses by Gauss-Newton on T <- T (+) d (R <- R Exp(d_w), t <- t + R d_v); every step solves (J'WJ + 1e-9 I) d = -J'W r by
// conjugate gradients with the 6 x 6 block-Jacobi preconditioner to a relative residual of 1e-10; a step is kept only if it lowers the
// cost.  The whole iteration - linearisation, preconditioner, the CG loop, the pose update - is ONE workgroup of 1 024 lanes in ONE launch:
// an essential graph has a few thousand edges, so the work per CG step is a few microseconds and anything spread over launches would be
// launch-bound (the reductions use a fixed order: results are reproducible run to run).
#define PG_T 1024
struct pg_args {
    int K, E, iters, max_cg;
    double* R; double* t; const uint8_t* fixed;                    // [K][9], [K][3], [K]
    const int* ei; const int* ej; const double* Zr; const double* Zt; const double* w;      // edges
    const int* n_start; const int* n_edge;                          // CSR node -> incident edges (edge index << 1 | side: 0 = i, 1 = j), in edge order
    double* Rb; double* tb;                                         // backup of the poses (a rejected step is undone)
    double* res; double* JA; double* JB; double* u;                // [E][6], [E][36], [E][36], [E][6]
    double* b; double* x; double* rr; double* z; double* p; double* Ap; double* Dinv;      // [K][6] each, Dinv [K][36]
    double* log; int* info;                                         // cost per evaluation; info[0] = Gauss-Newton steps kept, info[1] = CG iterations in total
};
__device__ __forceinline__ void pg_so3_log(const double* R, double* w)
{
    const double c = fmin(1.0, fmax(-1.0, 0.5 * ((R[0] + R[4] + R[8]) - 1.0)));
    const double th = acos(c);
    const double k = th < 1e-7 ? 0.5 + th * th / 12.0 : th / (2.0 * sin(th));
    w[0] = k * (R[7] - R[5]); w[1] = k * (R[2] - R[6]); w[2] = k * (R[3] - R[1]);
}
__device__ __forceinline__ void pg_so3_exp(const double* w, double* R)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    const double a = th < 1e-7 ? 1.0 - th2 / 6.0 : sin(th) / th, b = th < 1e-7 ? 0.5 - th2 / 24.0 : (1.0 - cos(th)) / th2;
    const double x = w[0], y = w[1], z = w[2];
    R[0] = 1.0 - b * (y * y + z * z); R[1] = -a * z + b * x * y;      R[2] = a * y + b * x * z;
    R[3] = a * z + b * x * y;         R[4] = 1.0 - b * (x * x + z * z); R[5] = -a * x + b * y * z;
    R[6] = -a * y + b * x * z;        R[7] = a * x + b * y * z;       R[8] = 1.0 - b * (x * x + y * y);
}
// inverse right Jacobian of SO(3): I + [w]x / 2 + c [w]x^2, c = 1/th^2 - (1 + cos th) / (2 th sin th)
__device__ __forceinline__ void pg_jr_inv(const double* w, double* J)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    const double c = th < 1e-4 ? 1.0 / 12.0 + th2 / 720.0 : 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    const double x = w[0], y = w[1], z = w[2];
    J[0] = 1.0 - c * (y * y + z * z); J[1] = -0.5 * z + c * x * y;      J[2] = 0.5 * y + c * x * z;
    J[3] = 0.5 * z + c * x * y;       J[4] = 1.0 - c * (x * x + z * z); J[5] = -0.5 * x + c * y * z;
    J[6] = -0.5 * y + c * x * z;      J[7] = 0.5 * x + c * y * z;       J[8] = 1.0 - c * (x * x + y * y);
}
__device__ __forceinline__ void pg_mm(const double* A, const double* B, double* C)      // 3x3 C = A B
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void pg_mtm(const double* A, const double* B, double* C)     // C = A' B
{
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}
// block-wide sum in a fixed order: every lane adds its strided elements in sequence, then a binary tree over the 1 024 partial sums
__device__ __forceinline__ double pg_block_sum(double v, double* s_red)
{
    const int tid = threadIdx.x;
    __syncthreads();
    s_red[tid] = v;
    __syncthreads();
    for (int m = PG_T / 2; m > 0; m >>= 1) { if (tid < m) s_red[tid] += s_red[tid + m]; __syncthreads(); }
    return s_red[0];
}
// residual and Jacobians of edge e at the current poses: r = (t_E, log R_E); rows 0-2 translation, 3-5 rotation; columns (d_v, d_w)
__device__ __forceinline__ double pg_edge(const pg_args& G, int e, bool jac)
{
    const int i = G.ei[e], j = G.ej[e];
    const double* Ri = G.R + 9 * (size_t)i; const double* Rj = G.R + 9 * (size_t)j; const double* ti = G.t + 3 * (size_t)i; const double* tj = G.t + 3 * (size_t)j;
    const double* Zr = G.Zr + 9 * (size_t)e; const double* Zt = G.Zt + 3 * (size_t)e;
    double M[9], RE[9], a[3], d[3] = { tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2] }, tE[3], ph[3];
    pg_mtm(Ri, Rj, M);                                             // Ri' Rj
    pg_mtm(Zr, M, RE);                                             // R_E = Zr' Ri' Rj
    for (int k = 0; k < 3; k++) a[k] = Ri[k] * d[0] + Ri[3 + k] * d[1] + Ri[6 + k] * d[2];      // a = Ri' (tj - ti)
    for (int k = 0; k < 3; k++) tE[k] = Zr[k] * (a[0] - Zt[0]) + Zr[3 + k] * (a[1] - Zt[1]) + Zr[6 + k] * (a[2] - Zt[2]);
    pg_so3_log(RE, ph);
    double* r = G.res + 6 * (size_t)e;
    r[0] = tE[0]; r[1] = tE[1]; r[2] = tE[2]; r[3] = ph[0]; r[4] = ph[1]; r[5] = ph[2];
    const double c = G.w[e] * (((tE[0] * tE[0] + tE[1] * tE[1]) + tE[2] * tE[2]) + ((ph[0] * ph[0] + ph[1] * ph[1]) + ph[2] * ph[2]));
    if (!jac) return c;
    double Ji[9], ax[9] = { 0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0 }, T1[9], T3[9];
    pg_jr_inv(ph, Ji);
    double* A = G.JA + 36 * (size_t)e; double* B = G.JB + 36 * (size_t)e;
    for (int k = 0; k < 36; k++) { A[k] = 0.0; B[k] = 0.0; }
    pg_mtm(Zr, ax, T1);                                            // d tE / d w_i = Zr' [a]x
    // d phi / d w_i = -Jr^-1 RE' Zr'   ((RE' Zr')[r][c] = sum_k RE[k][r] Zr[c][k])
    double REtZt[9];
    for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) REtZt[3 * rr + cc] = RE[rr] * Zr[3 * cc] + RE[3 + rr] * Zr[3 * cc + 1] + RE[6 + rr] * Zr[3 * cc + 2];
    pg_mm(Ji, REtZt, T3);
    for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) {
        A[6 * rr + cc] = -Zr[3 * cc + rr];                         // d tE / d v_i = -Zr'
        A[6 * rr + 3 + cc] = T1[3 * rr + cc];
        A[6 * (3 + rr) + 3 + cc] = -T3[3 * rr + cc];
        B[6 * rr + cc] = RE[3 * rr + cc];                          // d tE / d v_j = R_E
        B[6 * (3 + rr) + 3 + cc] = Ji[3 * rr + cc];                // d phi / d w_j = Jr^-1
    }
    return c;
}
__global__ __launch_bounds__(PG_T) void k_pose_graph(pg_args G)
{
    __shared__ double s_red[PG_T];
    const int tid = threadIdx.x;
    const double lam = 1e-9;
    double prev = 0.0;
    int kept = 0, cg_total = 0;
    for (int it = 0; it <= G.iters; it++) {
        // ---- cost (and, unless this is the closing evaluation, the linearisation) at the current poses
        double c = 0.0;
        for (int e = tid; e < G.E; e += PG_T) c += pg_edge(G, e, it < G.iters);
        const double cost = pg_block_sum(c, s_red);
        if (it > 0 && !(cost < prev)) {                             // the last step did not help: undo it and stop
            for (int k = tid; k < 9 * G.K; k += PG_T) G.R[k] = G.Rb[k];
            for (int k = tid; k < 3 * G.K; k += PG_T) G.t[k] = G.tb[k];
            break;
        }
        if (tid == 0) G.log[it] = cost;
        if (it > 0) kept++;
        prev = cost;
        if (it == G.iters) break;
        __syncthreads();
        // ---- right-hand side b = -J'W r and the block-Jacobi preconditioner (inverse of the node's 6 x 6 diagonal block + lam I)
        for (int n = tid; n < G.K; n += PG_T) {
            double bn[6] = { 0, 0, 0, 0, 0, 0 }, D[36];
            for (int k = 0; k < 36; k++) D[k] = 0.0;
            if (!G.fixed[n])
                for (int q = G.n_start[n]; q < G.n_start[n + 1]; q++) {
                    const int e = G.n_edge[q] >> 1; const double* J = (G.n_edge[q] & 1) ? G.JB + 36 * (size_t)e : G.JA + 36 * (size_t)e;
                    const double* r = G.res + 6 * (size_t)e; const double we = G.w[e];
                    for (int a = 0; a < 6; a++) {
                        double s = 0.0;
                        for (int k = 0; k < 6; k++) s += J[6 * k + a] * r[k];
                        bn[a] -= we * s;
                        for (int b2 = 0; b2 <= a; b2++) { double h = 0.0; for (int k = 0; k < 6; k++) h += J[6 * k + a] * J[6 * k + b2]; D[6 * a + b2] += we * h; }
                    }
                }
            double* Di = G.Dinv + 36 * (size_t)n;
            if (G.fixed[n]) { for (int k = 0; k < 36; k++) Di[k] = 0.0; }
            else {
                // Cholesky of D + lam I (lower), then the inverse column by column
                double L[36];
                for (int a = 0; a < 6; a++) for (int b2 = 0; b2 <= a; b2++) {
                    double s = D[6 * a + b2] + (a == b2 ? lam : 0.0);
                    for (int k = 0; k < b2; k++) s -= L[6 * a + k] * L[6 * b2 + k];
                    L[6 * a + b2] = a == b2 ? sqrt(fmax(s, 1e-300)) : s / L[6 * b2 + b2];
                }
                for (int col = 0; col < 6; col++) {
                    double y[6];
                    for (int a = 0; a < 6; a++) { double s = a == col ? 1.0 : 0.0; for (int k = 0; k < a; k++) s -= L[6 * a + k] * y[k]; y[a] = s / L[6 * a + a]; }
                    for (int a = 5; a >= 0; a--) { double s = y[a]; for (int k = a + 1; k < 6; k++) s -= L[6 * k + a] * y[k]; y[a] = s / L[6 * a + a]; }
                    for (int a = 0; a < 6; a++) Di[6 * a + col] = y[a];
                }
            }
            for (int a = 0; a < 6; a++) { G.b[6 * (size_t)n + a] = bn[a]; G.x[6 * (size_t)n + a] = 0.0; G.rr[6 * (size_t)n + a] = bn[a]; }
        }
        __syncthreads();
        // z = Dinv r; p = z; rz = <r, z>; |b|^2
        double l_rz = 0.0, l_bb = 0.0;
        for (int n = tid; n < G.K; n += PG_T) {
            const double* Di = G.Dinv + 36 * (size_t)n; const double* r = G.rr + 6 * (size_t)n;
            for (int a = 0; a < 6; a++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += Di[6 * a + k] * r[k];
                G.z[6 * (size_t)n + a] = s; G.p[6 * (size_t)n + a] = s; l_rz += r[a] * s; l_bb += r[a] * r[a];
            }
        }
        double rz = pg_block_sum(l_rz, s_red);
        const double bb = pg_block_sum(l_bb, s_red);
        if (bb > 0.0)
            for (int cg = 0; cg < G.max_cg; cg++) {
                cg_total++;
                for (int e = tid; e < G.E; e += PG_T) {                  // u_e = J_i p_i + J_j p_j
                    const double* A = G.JA + 36 * (size_t)e; const double* B = G.JB + 36 * (size_t)e;
                    const double* pi = G.p + 6 * (size_t)G.ei[e]; const double* pj = G.p + 6 * (size_t)G.ej[e];
                    for (int a = 0; a < 6; a++) { double s = 0.0; for (int k = 0; k < 6; k++) s += A[6 * a + k] * pi[k] + B[6 * a + k] * pj[k]; G.u[6 * (size_t)e + a] = s; }
                }
                __syncthreads();
                double l_pap = 0.0;
                for (int n = tid; n < G.K; n += PG_T) {                  // Ap_n = sum_inc J' w u + lam p_n
                    double acc[6] = { 0, 0, 0, 0, 0, 0 };
                    if (!G.fixed[n]) {
                        for (int q = G.n_start[n]; q < G.n_start[n + 1]; q++) {
                            const int e = G.n_edge[q] >> 1; const double* J = (G.n_edge[q] & 1) ? G.JB + 36 * (size_t)e : G.JA + 36 * (size_t)e;
                            const double* ue = G.u + 6 * (size_t)e; const double we = G.w[e];
                            for (int a = 0; a < 6; a++) { double s = 0.0; for (int k = 0; k < 6; k++) s += J[6 * k + a] * ue[k]; acc[a] += we * s; }
                        }
                        for (int a = 0; a < 6; a++) acc[a] += lam * G.p[6 * (size_t)n + a];
                    }
                    for (int a = 0; a < 6; a++) { G.Ap[6 * (size_t)n + a] = acc[a]; l_pap += G.p[6 * (size_t)n + a] * acc[a]; }
                }
                const double pap = pg_block_sum(l_pap, s_red);
                if (!(pap > 0.0)) break;
                const double alpha = rz / pap;
                double l_rr = 0.0;
                for (int k = tid; k < 6 * G.K; k += PG_T) { G.x[k] += alpha * G.p[k]; const double v = G.rr[k] - alpha * G.Ap[k]; G.rr[k] = v; l_rr += v * v; }
                const double rr2 = pg_block_sum(l_rr, s_red);
                if (rr2 <= 1e-20 * bb) break;
                double l_rz2 = 0.0;
                for (int n = tid; n < G.K; n += PG_T) {
                    const double* Di = G.Dinv + 36 * (size_t)n; const double* r = G.rr + 6 * (size_t)n;
                    for (int a = 0; a < 6; a++) { double s = 0.0; for (int k = 0; k < 6; k++) s += Di[6 * a + k] * r[k]; G.z[6 * (size_t)n + a] = s; l_rz2 += r[a] * s; }
                }
                const double rz2 = pg_block_sum(l_rz2, s_red);
                const double beta = rz2 / rz;
                rz = rz2;
                for (int k = tid; k < 6 * G.K; k += PG_T) G.p[k] = G.z[k] + beta * G.p[k];
                __syncthreads();
            }
        __syncthreads();
        // ---- T <- T (+) x (after a backup)
        for (int n = tid; n < G.K; n += PG_T) {
            double* R = G.R + 9 * (size_t)n; double* t = G.t + 3 * (size_t)n;
            for (int k = 0; k < 9; k++) G.Rb[9 * (size_t)n + k] = R[k];
            for (int k = 0; k < 3; k++) G.tb[3 * (size_t)n + k] = t[k];
            if (G.fixed[n]) continue;
            const double* d = G.x + 6 * (size_t)n;
            double Ex[9], Rn[9];
            pg_so3_exp(d + 3, Ex);
            pg_mm(R, Ex, Rn);
            for (int k = 0; k < 3; k++) t[k] += R[3 * k] * d[0] + R[3 * k + 1] * d[1] + R[3 * k + 2] * d[2];
            for (int k = 0; k < 9; k++) R[k] = Rn[k];
        }
        __syncthreads();
    }
    if (tid == 0) { G.info[0] = kept; G.info[1] = cg_total; }
}

extern "C" int bh_pose_graph_optimize(bh_ctx* ctx, int K, double* R, double* t, const uint8_t* fixed, int E, const int* ei, const int* ej,
                                      const double* Zr, const double* Zt, const double* w, int iters, double* cost_log, int* steps_kept, int* cg_iters)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!R || !t || !fixed || !ei || !ej || !Zr || !Zt || !w || !cost_log || K < 1 || E < 0 || iters < 0 || iters > 1000)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_pose_graph_optimize: bad argument");
    bool any_fixed = false;
    for (int n = 0; n < K; n++) any_fixed |= fixed[n] != 0;
    if (!any_fixed) return bh_set_err(ctx, BH_ERR_INVALID, "bh_pose_graph_optimize: at least one pose must be fixed (the gauge)");
    for (int e = 0; e < E; e++) if (ei[e] < 0 || ei[e] >= K || ej[e] < 0 || ej[e] >= K || ei[e] == ej[e] || !(w[e] >= 0)) return bh_set_err(ctx, BH_ERR_INVALID, "bh_pose_graph_optimize: bad edge %d", e);
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // node -> incident edges, in edge order (the order of every sum over a node's edges)
    std::vector<int> ns((size_t)K + 1, 0), ne((size_t)2 * E);
    for (int e = 0; e < E; e++) { ns[ei[e] + 1]++; ns[ej[e] + 1]++; }
    for (int n = 0; n < K; n++) ns[n + 1] += ns[n];
    { std::vector<int> cur(ns.begin(), ns.end() - 1); for (int e = 0; e < E; e++) { ne[cur[ei[e]]++] = e << 1; ne[cur[ej[e]]++] = (e << 1) | 1; } }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t k = (size_t)K, ee = (size_t)(E > 0 ? E : 1), D = sizeof(double), I = sizeof(int);
    const size_t total = 2 * al(D * 9 * k) + 2 * al(D * 3 * k) + al(k) + 2 * al(I * ee) + al(D * 9 * ee) + al(D * 3 * ee) + al(D * ee) + al(I * (k + 1)) + al(I * 2 * ee) +
                         al(D * 6 * ee) * 2 + al(D * 36 * ee) * 2 + 6 * al(D * 6 * k) + al(D * 36 * k) + al(D * (size_t)(iters + 1)) + al(I * 2);
    BH_TRY(bh_reserve(ctx, &ctx->st_h, total));
    char* p = (char*)ctx->st_h.p;
    auto take = [&](size_t bytes) { char* q = p; p += al(bytes); return q; };
    pg_args G; memset(&G, 0, sizeof(G));
    G.K = K; G.E = E; G.iters = iters; G.max_cg = 6 * K + 50;
    G.R = (double*)take(D * 9 * k); G.Rb = (double*)take(D * 9 * k); G.t = (double*)take(D * 3 * k); G.tb = (double*)take(D * 3 * k);
    uint8_t* d_fixed = (uint8_t*)take(k); int* d_ei = (int*)take(I * ee); int* d_ej = (int*)take(I * ee);
    double* d_Zr = (double*)take(D * 9 * ee); double* d_Zt = (double*)take(D * 3 * ee); double* d_w = (double*)take(D * ee);
    int* d_ns = (int*)take(I * (k + 1)); int* d_ne = (int*)take(I * 2 * ee);
    G.res = (double*)take(D * 6 * ee); G.u = (double*)take(D * 6 * ee); G.JA = (double*)take(D * 36 * ee); G.JB = (double*)take(D * 36 * ee);
    G.b = (double*)take(D * 6 * k); G.x = (double*)take(D * 6 * k); G.rr = (double*)take(D * 6 * k); G.z = (double*)take(D * 6 * k); G.p = (double*)take(D * 6 * k); G.Ap = (double*)take(D * 6 * k);
    G.Dinv = (double*)take(D * 36 * k); G.log = (double*)take(D * (size_t)(iters + 1)); G.info = (int*)take(I * 2);
    G.fixed = d_fixed; G.ei = d_ei; G.ej = d_ej; G.Zr = d_Zr; G.Zt = d_Zt; G.w = d_w; G.n_start = d_ns; G.n_edge = d_ne;
    BH_HIP(ctx, hipMemcpyAsync(G.R, R, D * 9 * k, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(G.t, t, D * 3 * k, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(d_fixed, fixed, k, hipMemcpyHostToDevice, st));
    if (E > 0) {
        BH_HIP(ctx, hipMemcpyAsync(d_ei, ei, I * ee, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_ej, ej, I * ee, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(d_Zr, Zr, D * 9 * ee, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_Zt, Zt, D * 3 * ee, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(d_w, w, D * ee, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_ne, ne.data(), I * 2 * ee, hipMemcpyHostToDevice, st));
    }
    BH_HIP(ctx, hipMemcpyAsync(d_ns, ns.data(), I * (k + 1), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemsetAsync(G.log, 0, D * (size_t)(iters + 1), st));
    BH_LAUNCH(ctx, BH_K_ALIGN, k_pose_graph, dim3(1), dim3(PG_T), st, G);
    BH_HIP(ctx, hipGetLastError());
    int info[2] = { 0, 0 };
    BH_HIP(ctx, hipMemcpyAsync(R, G.R, D * 9 * k, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(t, G.t, D * 3 * k, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(cost_log, G.log, D * (size_t)(iters + 1), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(info, G.info, sizeof(info), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    if (steps_kept) *steps_kept = info[0];
    if (cg_iters) *cg_iters = info[1];
    return BH_OK;
}
