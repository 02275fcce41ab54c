// align.hip - bh_align_se3_ransac: the rigid transform between two sets of matched 3-D points (loop-closure verification).
//
// Reference: slam/loop_closing.py:17-19 declares GlobalMapManager._similiarity_transformation(self) -> (R, t) - "R, t that transforms
// start of the loop with the end" - with body `pass`; there is no reference behaviour to match, the algorithm is
// the textbook one for metric RGB-D maps and is specified by this project: RANSAC over 3-point minimal sets + the Horn / Kabsch closed
// form (se3_small.h, shared as text with the CPU checker align_oracle.c of the test tree), OpenCV's sampling stream and adaptive iteration count
// as in the PnP path.  The kernel reproduces the checker bit for bit.
//
// One wave per problem.  The RANSAC iterations run in sequence (each may shorten the loop, as in OpenCV); inside an iteration the
// 3-point model is wave-uniform register arithmetic and the 64 lanes share the scoring pass (point i on lane i mod 64, a ballot per
// 64 points).  The refinement's sums over the inliers are lane-strided partial sums joined by an xor butterfly - the order the
// checker restates - so they are parallel AND reproducible.
#include "bh_internal.h"
#include "se3_small.h"

__device__ __forceinline__ double al_shfl_xor(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned al_rng_next(unsigned long long* st)
{
    *st = (unsigned long long)(unsigned)(*st) * 4164903690ull + (unsigned)(*st >> 32);
    return (unsigned)(*st);
}
__device__ __forceinline__ void al_term_h(const double* p, const double* q, const double* c, double* out)
{
    const double a0 = p[0] - c[0], a1 = p[1] - c[1], a2 = p[2] - c[2], b0 = q[0] - c[3], b1 = q[1] - c[4], b2 = q[2] - c[5];
    out[0] = a0 * b0; out[1] = a0 * b1; out[2] = a0 * b2; out[3] = a1 * b0; out[4] = a1 * b1; out[5] = a1 * b2; out[6] = a2 * b0; out[7] = a2 * b1; out[8] = a2 * b2;
}
// number of pairs within thr of the model; mask != nullptr: also written
__device__ __forceinline__ int al_count(const double* __restrict__ P, const double* __restrict__ Q, int n, const double* R, const double* t, double thr2,
                                        uint8_t* __restrict__ mask, int lane)
{
    int good = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < n) {
            const double p[3] = { P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2] }, q[3] = { Q[3 * (size_t)i], Q[3 * (size_t)i + 1], Q[3 * (size_t)i + 2] };
            in = se_resid2(R, t, p, q) <= thr2;
            if (mask) mask[i] = in ? 1 : 0;
        }
        good += __popcll(__ballot(in));
    }
    return good;
}

__global__ __launch_bounds__(64) void k_align_se3(const double* __restrict__ Pall, const double* __restrict__ Qall, const int* __restrict__ n_arr, int stride,
                                                  int iterations, double thr, double confidence, int min_inliers, double* __restrict__ Rout,
                                                  double* __restrict__ tout, uint8_t* __restrict__ mask_all, int* __restrict__ n_inl, int* __restrict__ status)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = min(max(n_arr[b], 0), stride);
    const double* P = Pall + 3 * (size_t)b * stride; const double* Q = Qall + 3 * (size_t)b * stride;
    uint8_t* mask = mask_all + (size_t)b * stride;
    for (int i = lane; i < stride; i += 64) mask[i] = 0;
    double bR[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, bt[3] = { 0, 0, 0 };
    int best = 0, cnt = 0, st_out = 2;
    if (n < 3) st_out = 1;
    else {
        const double thr2 = thr * thr;
        unsigned long long rs = ~0ull;
        int niters = iterations > 1 ? iterations : 1;
        for (int it = 0; it < niters; it++) {
            int idx[3];
            for (int i = 0; i < 3; i++) {
                int j;
                for (;;) { idx[i] = (int)(al_rng_next(&rs) % (unsigned)n); for (j = 0; j < i; j++) if (idx[i] == idx[j]) break; if (j == i) break; }
            }
            double c[6], H[9], mR[9], mt[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                c[k] = ((P[3 * (size_t)idx[0] + k] + P[3 * (size_t)idx[1] + k]) + P[3 * (size_t)idx[2] + k]) / 3.0;
                c[3 + k] = ((Q[3 * (size_t)idx[0] + k] + Q[3 * (size_t)idx[1] + k]) + Q[3 * (size_t)idx[2] + k]) / 3.0;
            }
#pragma unroll
            for (int k = 0; k < 9; k++) H[k] = 0.0;
            for (int s = 0; s < 3; s++) {
                const double p[3] = { P[3 * (size_t)idx[s]], P[3 * (size_t)idx[s] + 1], P[3 * (size_t)idx[s] + 2] };
                const double q[3] = { Q[3 * (size_t)idx[s]], Q[3 * (size_t)idx[s] + 1], Q[3 * (size_t)idx[s] + 2] };
                double term[9];
                al_term_h(p, q, c, term);
#pragma unroll
                for (int k = 0; k < 9; k++) H[k] = H[k] + term[k];
            }
            if (!se_kabsch(H, c, c + 3, mR, mt)) continue;
            const int good = al_count(P, Q, n, mR, mt, thr2, nullptr, lane);
            if (good > (best > 2 ? best : 2)) {
                best = good;
#pragma unroll
                for (int k = 0; k < 9; k++) bR[k] = mR[k];
#pragma unroll
                for (int k = 0; k < 3; k++) bt[k] = mt[k];
                niters = se_update_iters(confidence, (double)(n - good) / n, niters);
            }
        }
        if (best > 0) {
            cnt = al_count(P, Q, n, bR, bt, thr2, mask, lane);
            for (int round = 0; round < 2 && cnt >= 3; round++) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // centroids, then the centred cross-covariance: lane-strided partial sums, xor butterfly
                double s6[6] = { 0, 0, 0, 0, 0, 0 };
                for (int i = lane; i < n; i += 64) if (mask[i]) {
#pragma unroll
                    for (int k = 0; k < 3; k++) { s6[k] = s6[k] + P[3 * (size_t)i + k]; s6[3 + k] = s6[3 + k] + Q[3 * (size_t)i + k]; }
                }
#pragma unroll
                for (int m = 32; m > 0; m >>= 1)
#pragma unroll
                    for (int k = 0; k < 6; k++) s6[k] = s6[k] + al_shfl_xor(s6[k], m);
                double c[6], H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, nR[9], nt[3];
#pragma unroll
                for (int k = 0; k < 6; k++) c[k] = s6[k] / (double)cnt;
                for (int i = lane; i < n; i += 64) if (mask[i]) {
                    const double p[3] = { P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2] }, q[3] = { Q[3 * (size_t)i], Q[3 * (size_t)i + 1], Q[3 * (size_t)i + 2] };
                    double term[9];
                    al_term_h(p, q, c, term);
#pragma unroll
                    for (int k = 0; k < 9; k++) H[k] = H[k] + term[k];
                }
#pragma unroll
                for (int m = 32; m > 0; m >>= 1)
#pragma unroll
                    for (int k = 0; k < 9; k++) H[k] = H[k] + al_shfl_xor(H[k], m);
                if (!se_kabsch(H, c, c + 3, nR, nt)) break;
#pragma unroll
                for (int k = 0; k < 9; k++) bR[k] = nR[k];
#pragma unroll
                for (int k = 0; k < 3; k++) bt[k] = nt[k];
                cnt = al_count(P, Q, n, bR, bt, thr2, mask, lane);
            }
            st_out = (cnt >= min_inliers && cnt >= 3) ? 0 : 2;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) Rout[9 * (size_t)b + k] = bR[k];
#pragma unroll
        for (int k = 0; k < 3; k++) tout[3 * (size_t)b + k] = bt[k];
        n_inl[b] = cnt; status[b] = st_out;
    }
}

extern "C" int bh_align_se3_ransac(bh_ctx* ctx, const double* P, const double* Q, const int* n, int B, int stride, int iterations, double inlier_thr,
                                   double confidence, int min_inliers, double* R, double* t, uint8_t* inlier, int* n_inlier, int* status, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!P || !Q || !n || !R || !t || !inlier || !n_inlier || !status || B < 0 || stride < 1 || !(inlier_thr > 0) || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_align_se3_ransac: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const double *dP = P, *dQ = Q; const int* dn = n; double *dR = R, *dt = t; uint8_t* din = inlier; int *dni = n_inlier, *dst = status;
    const size_t pb = sizeof(double) * 3 * (size_t)B * stride, mb = (size_t)B * stride;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (n[b] < 0 || n[b] > stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_align_se3_ransac: row count outside [0, stride]");
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        BH_TRY(bh_reserve(ctx, &ctx->st_a, 2 * al(pb) + al(sizeof(int) * B) + al(sizeof(double) * 12 * B) + al(mb) + 2 * al(sizeof(int) * B)));
        char* ip = (char*)ctx->st_a.p;
        BH_HIP(ctx, hipMemcpyAsync(ip, P, pb, hipMemcpyHostToDevice, st)); dP = (const double*)ip; ip += al(pb);
        BH_HIP(ctx, hipMemcpyAsync(ip, Q, pb, hipMemcpyHostToDevice, st)); dQ = (const double*)ip; ip += al(pb);
        BH_HIP(ctx, hipMemcpyAsync(ip, n, sizeof(int) * B, hipMemcpyHostToDevice, st)); dn = (const int*)ip; ip += al(sizeof(int) * B);
        dR = (double*)ip; dt = dR + 9 * (size_t)B; ip += al(sizeof(double) * 12 * B);
        din = (uint8_t*)ip; ip += al(mb);
        dni = (int*)ip; ip += al(sizeof(int) * B); dst = (int*)ip;
    }
    BH_LAUNCH(ctx, BH_K_ALIGN, k_align_se3, dim3(B), dim3(64), st, dP, dQ, dn, stride, iterations, inlier_thr, confidence, min_inliers, dR, dt, din, dni, dst);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(R, dR, sizeof(double) * 9 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(t, dt, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(inlier, din, mb, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_inlier, dni, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(status, dst, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}
