// ba.hip — bundle adjustment on gfx950: the g2o graph of slam/optimizer.py as batched f64 HIP kernels.
//
//   bh_pose_only_ba   CamOnlyBA (slam/optimizer.py:82-142; callers slam/tracking.py:73,128): B independent problems,
//                     one workgroup each, the whole Levenberg loop (10 iterations x <=10 trials) inside one launch:
//                     per-edge residual/Jacobian -> 28-term block reduction -> 6x6 Cholesky -> SBACam oplus.
//   bh_ba_*           BundleAdjustment (slam/optimizer.py:8-79) as driven by CovisibilityGraph.optimize_local
//                     (slam/covisibility_graph.py:186-235): EdgeProjectP2MC linearisation, Hpp/Hll/Hpl assembly, Schur
//                     complement on the marginalised points, dense Cholesky of the reduced camera system,
//                     back-substitution, g2o's Levenberg lambda schedule.
//
// Design notes
//   * f64 throughout (the reference's g2o path is double); every reduction has a fixed order (CSR by point / by camera,
//     tree reductions), so results are bitwise reproducible run to run — no floating-point atomics anywhere.
//   * per LM iteration: k_ba_point (8 lanes per point: residual, Huber weight, Jacobians, Hll/bl, per-edge Hpl),
//     k_ba_cam (workgroup per free camera: Hpp/bp), k_ba_prep; per trial: k_ba_dinv, k_ba_schur_S (workgroup per
//     camera pair, binary-search join of the two cameras' point-sorted edge lists), k_ba_schur_b, k_ba_chol (one
//     workgroup), k_ba_update (back-substitution + oplus into the trial state), k_ba_err, k_ba_decide.  The host reads
//     back one 16-byte verdict per trial (accept / retry / terminate) — the only synchronisation.
//   * the reduced camera system of a local window is tiny (6 x #free keyframes <= a few hundred): a single-workgroup
//     right-looking Cholesky is used.  The MFMA-f64 tiled factorisation for global-BA sizes is a later row.
#include "bh_internal.h"
#include <math.h>
#include <float.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <utility>
#include "bh_scan.h"

struct ba_cam {
    double q[4];            // x y z w
    double t[3];
    double w2n[12];
    double w2i[12];
    double dR[27];
};

struct ba_K { double fx, fy, cx, cy; };

// ------------------------------------------------------------------------------------------------ SBACam math (device)
__device__ __forceinline__ void d_quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__device__ void d_cam_refresh(ba_cam* c, const ba_K K)
{
    double R[9]; d_quat_to_R(c->q, R);
    double* n = c->w2n;
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) n[r * 4 + k] = R[k * 3 + r];
    for (int r = 0; r < 3; r++) n[r * 4 + 3] = -(n[r * 4 + 0] * c->t[0] + n[r * 4 + 1] * c->t[1] + n[r * 4 + 2] * c->t[2]);
    for (int k = 0; k < 4; k++) {
        c->w2i[k] = K.fx * n[k] + K.cx * n[8 + k];
        c->w2i[4 + k] = K.fy * n[4 + k] + K.cy * n[8 + k];
        c->w2i[8 + k] = n[8 + k];
    }
    for (int k = 0; k < 3; k++) {
        const double r0 = n[k], r1 = n[4 + k], r2 = n[8 + k];
        c->dR[k] = 0;            c->dR[3 + k] = 2 * r2;        c->dR[6 + k] = -2 * r1;
        c->dR[9 + k] = -2 * r2;  c->dR[9 + 3 + k] = 0;         c->dR[9 + 6 + k] = 2 * r0;
        c->dR[18 + k] = 2 * r1;  c->dR[18 + 3 + k] = -2 * r0;  c->dR[18 + 6 + k] = 0;
    }
}

__device__ void d_cam_oplus(ba_cam* c, const double* d, const ba_K K)
{
    c->t[0] += d[0]; c->t[1] += d[1]; c->t[2] += d[2];
    const double bx = d[3], by = d[4], bz = d[5];
    const double bw = sqrt(1.0 - (bx * bx + by * by + bz * bz));
    const double ax = c->q[0], ay = c->q[1], az = c->q[2], aw = c->q[3];
    double q[4];
    q[3] = aw * bw - ax * bx - ay * by - az * bz;
    q[0] = aw * bx + ax * bw + ay * bz - az * by;
    q[1] = aw * by + ay * bw + az * bx - ax * bz;
    q[2] = aw * bz + az * bw + ax * by - ay * bx;
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) c->q[i] = q[i] / nrm;
    d_cam_refresh(c, K);
}

__device__ __forceinline__ void d_edge_error(const double* w2i, const double* X, const double* z, double* e)
{
    const double p0 = w2i[0] * X[0] + w2i[1] * X[1] + w2i[2] * X[2] + w2i[3];
    const double p1 = w2i[4] * X[0] + w2i[5] * X[1] + w2i[6] * X[2] + w2i[7];
    const double p2 = w2i[8] * X[0] + w2i[9] * X[1] + w2i[10] * X[2] + w2i[11];
    e[0] = p0 / p2 - z[0];
    e[1] = p1 / p2 - z[1];
}

__device__ __forceinline__ void d_edge_jac(const double* n, const double* dR, const double* ct, const double* X, const ba_K K,
                                           double* Jc, double* Jp)
{
    const double px = n[0] * X[0] + n[1] * X[1] + n[2] * X[2] + n[3];
    const double py = n[4] * X[0] + n[5] * X[1] + n[6] * X[2] + n[7];
    const double pz = n[8] * X[0] + n[9] * X[1] + n[10] * X[2] + n[11];
    const double ipz2 = 1.0 / (pz * pz);
    const double ipz2fx = ipz2 * K.fx, ipz2fy = ipz2 * K.fy;
    const double pwt[3] = { X[0] - ct[0], X[1] - ct[1], X[2] - ct[2] };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double* D = dR + 9 * a;
        const double dp0 = D[0] * pwt[0] + D[1] * pwt[1] + D[2] * pwt[2];
        const double dp1 = D[3] * pwt[0] + D[4] * pwt[1] + D[5] * pwt[2];
        const double dp2 = D[6] * pwt[0] + D[7] * pwt[1] + D[8] * pwt[2];
        Jc[3 + a] = (pz * dp0 - px * dp2) * ipz2fx;
        Jc[6 + 3 + a] = (pz * dp1 - py * dp2) * ipz2fy;
        double d0 = -n[a], d1 = -n[4 + a], d2 = -n[8 + a];
        Jc[a] = (pz * d0 - px * d2) * ipz2fx;
        Jc[6 + a] = (pz * d1 - py * d2) * ipz2fy;
        d0 = n[a]; d1 = n[4 + a]; d2 = n[8 + a];
        Jp[a] = (pz * d0 - px * d2) * ipz2fx;
        Jp[3 + a] = (pz * d1 - py * d2) * ipz2fy;
    }
}

// x^3 as the reference's pow(x, 3) returns it (g2o OptimizationAlgorithmLevenberg: alpha = 1. - pow((2*rho-1), 3)): the cube is carried
// as an unevaluated sum of two doubles (error-free products by fma) and rounded once, i.e. correctly rounded except in
// ~2^-50 of the arguments - where libm's pow (< 1 ulp) may round either way itself.  Two plain multiplications are 1 ulp off in ~25 %.
__device__ __forceinline__ double d_cube(double x)
{
    const double h = __dmul_rn(x, x), l = __fma_rn(x, x, -h);      // x^2 = h + l
    const double p = __dmul_rn(h, x), e = __fma_rn(h, x, -p);      // h x = p + e
    return __dadd_rn(p, __fma_rn(l, x, e));
}

__device__ __forceinline__ void d_huber(double e2, double delta, double* rho0, double* rho1)
{
    if (delta <= 0) { *rho0 = e2; *rho1 = 1.0; return; }
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double s = sqrt(e2); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

__device__ __forceinline__ double d_shfl_xor(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double d_shfl_idx(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}
template <int SH> __device__ __forceinline__ double d_row_shr(double v)          // value of lane - SH inside the 16-lane DPP row, 0 at the row's start
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SH, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SH, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double d_readlane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// value of lane ^ M (M = 8, 4, 2, 1) on the vector ALU: DPP row rotate / two bank-masked row shifts / quad permutes
template <int M> __device__ __forceinline__ int i_xor_dpp(int x)
{
    if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);                       // row_ror:8
    else if constexpr (M == 4) { const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);          // row_shl:4 into banks 0, 2
                                 return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false); }               // row_shr:4 into banks 1, 3
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4e, 0xf, 0xf, false);                    // quad_perm [2,3,0,1]
    else return __builtin_amdgcn_update_dpp(0, x, 0xb1, 0xf, 0xf, false);                                          // quad_perm [1,0,3,2]
}
// value of lane M of the lane's quad (DPP quad_perm [M, M, M, M])
template <int M> __device__ __forceinline__ double d_quad_bcast(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), M * 0x55, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), M * 0x55, 0xf, 0xf, false));
}
template <int M> __device__ __forceinline__ double d_xor_dpp(double v)
{
    return __hiloint2double(i_xor_dpp<M>(__double2hiint(v)), i_xor_dpp<M>(__double2loint(v)));
}
// value of lane ^ 32 / lane ^ 16 by one v_permlane32_swap / v_permlane16_swap of two copies (instead of a ds_bpermute and its address)
template <int M> __device__ __forceinline__ double d_xor_swap(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (M == 32) {
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const int lane = threadIdx.x & 63;
        return lane < 32 ? __hiloint2double(rh[1], rl[1]) : __hiloint2double(rh[0], rl[0]);
    } else {
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const int lane = threadIdx.x & 63;
        return (lane & 16) ? __hiloint2double(rh[0], rl[0]) : __hiloint2double(rh[1], rl[1]);
    }
}
__device__ __forceinline__ double d_wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += d_shfl_xor(v, m);
    return v;
}

// Wave sums of NV values per lane as a REDUCE-SCATTER: the pairs are those of d_wave_sum's xor butterfly (32, 16, 8, 4, 2, 1), so every
// sum is the one d_wave_sum gives bit for bit (an IEEE addition commutes), but the first two steps halve the number of values a
// lane carries - gfx950's v_permlane32_swap / v_permlane16_swap exchange the halves the partner keeps in ONE instruction - and
// the last four run on DPP.  36 values: 207 vector instructions and no LDS traffic where 36 butterflies were 432 ds_bpermutes plus as
// many address and add instructions.  Afterwards v[i], i < N2, of a lane in 16-lane row q is value i + N2 (q & 1) + N1 (q >> 1)
// (N1 = ceil(NV / 2), N2 = ceil(N1 / 2); combinations that run past N1 / NV are padding).
template <int NV> __device__ __forceinline__ void d_wave_sum_rs(double* v)
{
    constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2;
#pragma unroll
    for (int i = 0; i < N1; i++) {
        int alo = __double2loint(v[i]), ahi = __double2hiint(v[i]);
        int blo = i + N1 < NV ? __double2loint(v[i + N1]) : 0, bhi = i + N1 < NV ? __double2hiint(v[i + N1]) : 0;
        const auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false), rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        v[i] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
    }
#pragma unroll
    for (int i = 0; i < N2; i++) {
        int alo = __double2loint(v[i]), ahi = __double2hiint(v[i]);
        int blo = i + N2 < N1 ? __double2loint(v[i + N2]) : 0, bhi = i + N2 < N1 ? __double2hiint(v[i + N2]) : 0;
        const auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false), rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        v[i] = __hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]);
    }
#pragma unroll
    for (int i = 0; i < N2; i++) { v[i] += d_xor_dpp<8>(v[i]); v[i] += d_xor_dpp<4>(v[i]); v[i] += d_xor_dpp<2>(v[i]); v[i] += d_xor_dpp<1>(v[i]); }
}
// the four lanes 0, 16, 32, 48 of a wave store what d_wave_sum_rs left them: out[value index] (NV doubles)
template <int NV> __device__ __forceinline__ void d_wave_sum_rs_store(const double* v, double* out)
{
    constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2;
    const int lane = threadIdx.x & 63, q = lane >> 4;
    if (lane & 15) return;
#pragma unroll
    for (int i = 0; i < N2; i++) {
        const int sub = i + N2 * (q & 1), idx = sub + N1 * (q >> 1);
        if (sub < N1 && idx < NV) out[idx] = v[i];
    }
}

// block-wide sum of NV doubles per thread, fixed order; result valid in every thread.  s_red: [NV * nwaves] doubles.
template <int NV, int NT>
__device__ void d_block_sum(double* v, double* s_red)
{
    const int wv = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    d_wave_sum_rs<NV>(v);
    __syncthreads();
    d_wave_sum_rs_store<NV>(v, s_red + wv * NV);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) { double s = 0; for (int w = 0; w < NW; w++) s += s_red[w * NV + i]; v[i] = s; }
}

// ---- fan-in of per-workgroup partial sums (round 5).  The Levenberg control needs three big sums per trial - chi over the edges at the
// linearisation point and at the trial state, dx'(lambda dx + b) over the landmark steps - and a maximum over the diagonal once.  They used
// to be separate launches behind the kernels that produce their terms (k_ba_partial x 2, k_ba_prep, k_ba_decide: four of a local window's ten
// launches per iteration, ~7 us each with the gap in front).  Now every workgroup of the producing kernel leaves the sum of ITS terms (fixed
// order) in a slot of its own, and the workgroup that arrives LAST - an agent-scope counter - adds the slots up in slot order and runs the
// control step.  No workgroup waits for another; the result does not depend on which one is last.
// Visibility across the XCDs' private L2s: a slot is written by ONE write-through (agent-scope atomic) store, drained before that lane's counter
// add; the last arriver acquires at agent scope and reads the slots with agent-scope loads (never through the scalar cache).
__device__ __forceinline__ void fan_store(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fan_load(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// one wave (all 64 lanes): the n slots summed in a fixed order - lane l takes slots l, l + 64, .. one after the other, then the xor tree;
// the same bits whichever workgroup (of whatever size) runs it, which is what keeps a batched window bitwise equal to the same window alone
__device__ __forceinline__ double fan_sum(const double* part, int n)
{
    const int lane = threadIdx.x & 63;
    double t = 0;
    for (int i0 = lane; i0 < n; i0 += 64 * 8) {                  // eight loads in flight per lane (one after the other they cost a memory round trip each)
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + 64 * k < n ? fan_load(part + i0 + 64 * k) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) t += v[k];                   // (+ 0.0 beyond the end: the sums are sums of squares and of dx'(lambda dx + b) terms, never -0.0)
    }
    for (int m = 32; m > 0; m >>= 1) t += d_shfl_xor(t, m);
    return t;
}
__device__ __forceinline__ double fan_max(const double* part, int n, double t)
{
    const int lane = threadIdx.x & 63;
    for (int i0 = lane; i0 < n; i0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + 64 * k < n ? fan_load(part + i0 + 64 * k) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) t = fmax(t, v[k]);
    }
    for (int m = 32; m > 0; m >>= 1) t = fmax(t, d_shfl_xor(t, m));
    return t;
}
// every thread of the workgroup calls it (lane 0 has stored the workgroup's slots): true, in every thread, in the workgroup that arrives last of `total`.
// Two levels: workgroup `bid` counts into the counter of its group of 64 (grp[bid / 64]), the last of a group into `top` - the 15 000 workgroups of a
// configs[4] launch adding to ONE word serialise in the L2 (~25 ns each: the error kernel took 430 us instead of 80).  The last arrivers put the
// counters back to zero for the next launch.
__device__ __forceinline__ bool fan_arrive(int* top, int* grp, int bid, int total)
{
    __shared__ int s_tk;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slot stores of this lane have left the CU (write-through)
        const int g = bid >> 6, ngroups = (total + 63) >> 6, gsize = min(64, total - (g << 6));
        int last = 0;
        if (__hip_atomic_fetch_add(grp + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(grp + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(top, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
                __hip_atomic_store(top, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_tk = last;
    }
    __syncthreads();
    const bool last = s_tk != 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last;
}
// sum of one value per thread over a 256-thread workgroup, fixed order (valid in every thread); s4: 4 doubles of LDS
__device__ __forceinline__ double fan_block_sum256(double v, double* s4)
{
    double a[1] = { v };
    d_block_sum<1, 256>(a, s4);
    return a[0];
}
// The sum of n slots by a workgroup, the same bits whoever runs it: up to FAN_WAVE_MAX slots by wave 0 alone (fan_sum: what a batched window's
// 1024-thread control kernels and a single window's 256-thread last arriver both do); beyond that - configs[4] has 15 000 slots, one wave would walk
// them in 30 dependent round trips - thread t of a 256-thread workgroup takes slots t, t + 256, .. and the workgroup adds up in fixed order (only
// bh_ba_optimize's own kernels see such sizes).  Valid in thread 0.  Every thread of the workgroup calls it.
#define FAN_WAVE_MAX 2048
__device__ __forceinline__ double fan_sum_wg(const double* part, int n, double* s4)
{
    if (n <= FAN_WAVE_MAX) return threadIdx.x < 64 ? fan_sum(part, n) : 0.0;
    double t = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + 256 * k < n ? fan_load(part + i0 + 256 * k) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) t += v[k];
    }
    return fan_block_sum256(t, s4);
}
__device__ __forceinline__ double fan_max_wg(const double* part, int n, double t0, double* s4)
{
    if (n <= FAN_WAVE_MAX) return threadIdx.x < 64 ? fan_max(part, n, t0) : t0;
    double t = t0;
    for (int i = threadIdx.x; i < n; i += 256) t = fmax(t, fan_load(part + i));
    for (int m = 32; m > 0; m >>= 1) t = fmax(t, d_shfl_xor(t, m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = t;
    __syncthreads();
    return fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
}

// in-register 6x6 Cholesky solve; returns false if not positive definite
__device__ bool d_chol6(double* A /*36*/, double* b /*6*/)
{
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) return false;
        d = sqrt(d); A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * 6 + k] * b[k]; b[i] = s / A[i * 6 + i]; }
    for (int i = 5; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < 6; k++) s -= A[k * 6 + i] * b[k]; b[i] = s / A[i * 6 + i]; }
    return true;
}

// ================================================================================================ pose-only BA
#define PB_T 256

__global__ __launch_bounds__(PB_T) void k_pose_ba(const double* __restrict__ pix, const double* __restrict__ pts,
                                                  const int* __restrict__ n_arr, int stride, ba_K K, int iters,
                                                  double delta, double chi2_thr, double* __restrict__ Rout,
                                                  double* __restrict__ tout, double* __restrict__ chi2,
                                                  uint8_t* __restrict__ inlier, int* __restrict__ status)
{
    __shared__ ba_cam s_cam, s_bak;
    __shared__ double s_red[28 * (PB_T / 64)];
    __shared__ double s_dx[6];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const int n = n_arr[b];
    const double* px = pix + b * (size_t)stride * 2;
    const double* X = pts + b * (size_t)stride * 3;
    double* c2 = chi2 + b * (size_t)stride;

    if (tid == 0) {
        s_cam.q[0] = s_cam.q[1] = s_cam.q[2] = 0; s_cam.q[3] = 1; s_cam.t[0] = s_cam.t[1] = s_cam.t[2] = 0;     // optimizer.py:90-91
        d_cam_refresh(&s_cam, K);
    }
    __syncthreads();
    double lambda = 0, ni = 2;
    bool numeric_fail = false;
    for (int it = 0; it < iters; it++) {
        // linearise: H (21 unique), b (6), chi
        double acc[28];
#pragma unroll
        for (int i = 0; i < 28; i++) acc[i] = 0;
        for (int e = tid; e < n; e += PB_T) {
            double er[2], Jc[12], Jp[6], r0, r1;
            d_edge_error(s_cam.w2i, X + 3 * e, px + 2 * e, er);
            d_edge_jac(s_cam.w2n, s_cam.dR, s_cam.t, X + 3 * e, K, Jc, Jp);
            d_huber(er[0] * er[0] + er[1] * er[1], delta, &r0, &r1);
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int bb = a; bb < 6; bb++) acc[k++] += r1 * (Jc[a] * Jc[bb] + Jc[6 + a] * Jc[6 + bb]);
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] -= r1 * (Jc[a] * er[0] + Jc[6 + a] * er[1]);
            acc[27] += r0;
        }
        d_block_sum<28, PB_T>(acc, s_red);
        double H[36], bvec[6];
        { int k = 0; for (int a = 0; a < 6; a++) for (int bb = a; bb < 6; bb++) { H[a * 6 + bb] = acc[k]; H[bb * 6 + a] = acc[k]; k++; } }
        for (int a = 0; a < 6; a++) bvec[a] = acc[21 + a];
        double chi = acc[27];
        if (it == 0) {
            double md = 0;
            for (int a = 0; a < 6; a++) md = fmax(fabs(H[a * 6 + a]), md);
            if (!isfinite(md)) { numeric_fail = true; break; }
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            if (tid == 0) {
                s_bak = s_cam;
                double A[36], dx[6];
                for (int i = 0; i < 36; i++) A[i] = H[i];
                for (int a = 0; a < 6; a++) { A[a * 6 + a] += lambda; dx[a] = bvec[a]; }
                s_ok = d_chol6(A, dx) ? 1 : 0;
                for (int a = 0; a < 6; a++) s_dx[a] = dx[a];
                d_cam_oplus(&s_cam, dx, K);
            }
            __syncthreads();
            double tc[1] = { 0 };
            for (int e = tid; e < n; e += PB_T) {
                double er[2], r0, r1;
                d_edge_error(s_cam.w2i, X + 3 * e, px + 2 * e, er);
                const double e2 = er[0] * er[0] + er[1] * er[1];
                c2[e] = e2;                                   // edge error of the LAST computeActiveErrors call
                d_huber(e2, delta, &r0, &r1);
                tc[0] += r0;
            }
            d_block_sum<1, PB_T>(tc, s_red);
            double tempChi = tc[0];
            if (!s_ok) tempChi = DBL_MAX;
            rho = chi - tempChi;
            double scale = 0;
            for (int a = 0; a < 6; a++) scale += s_dx[a] * (lambda * s_dx[a] + bvec[a]);
            scale += 1e-3;
            rho /= scale;
            __syncthreads();
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - d_cube(2 * rho - 1);         // g2o: 1. - pow((2*rho-1), 3)
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; chi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                if (tid == 0) s_cam = s_bak;
            }
            __syncthreads();
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;
    }
    __syncthreads();
    if (iters <= 0 || numeric_fail)          // no trial ran: report the errors at the initial estimate
        for (int e = tid; e < n; e += PB_T) { double er[2]; d_edge_error(s_cam.w2i, X + 3 * e, px + 2 * e, er); c2[e] = er[0] * er[0] + er[1] * er[1]; }
    __syncthreads();
    for (int e = tid; e < n; e += PB_T) inlier[b * (size_t)stride + e] = c2[e] <= chi2_thr ? 1 : 0;
    if (tid == 0) {
        double R[9]; d_quat_to_R(s_cam.q, R);
        bool fin = true;
        for (int i = 0; i < 9; i++) { Rout[b * 9 + i] = R[i]; fin = fin && isfinite(R[i]); }
        for (int i = 0; i < 3; i++) { tout[b * 3 + i] = s_cam.t[i]; fin = fin && isfinite(s_cam.t[i]); }
        status[b] = (fin && !numeric_fail) ? BH_OK : BH_ERR_NUMERIC;
    }
}

extern "C" int bh_pose_only_ba(bh_ctx* ctx, const double* pix, const double* pts, const int* n, int B, int stride,
                               const double K[5], int iters, double huber_delta, double chi2_thr,
                               double* R, double* t, double* chi2, uint8_t* inlier, int* status, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!pix || !pts || !n || !K || !R || !t || !chi2 || !inlier || !status || B < 0 || stride < 1 ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_pose_only_ba: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t ne = (size_t)B * stride;
    const double* d_pix = pix; const double* d_pts = pts; const int* d_n = n;
    double *d_R = R, *d_t = t, *d_chi = chi2; uint8_t* d_in = inlier; int* d_st = status;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (n[b] < 0 || n[b] > stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_pose_only_ba: n[b] outside [0, stride]");
        BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 2 * ne)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 3 * ne));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(int) * 2 * B)); BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(double) * 12 * B));
        BH_TRY(bh_reserve(ctx, &ctx->st_e, sizeof(double) * ne)); BH_TRY(bh_reserve(ctx, &ctx->st_f, ne));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, pix, sizeof(double) * 2 * ne, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, pts, sizeof(double) * 3 * ne, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, n, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_pix = (const double*)ctx->st_a.p; d_pts = (const double*)ctx->st_b.p; d_n = (const int*)ctx->st_c.p; d_st = (int*)ctx->st_c.p + B;
        d_R = (double*)ctx->st_d.p; d_t = d_R + 9 * (size_t)B; d_chi = (double*)ctx->st_e.p; d_in = (uint8_t*)ctx->st_f.p;
    }
    ba_K Kd = { K[0], K[1], K[2], K[3] };
    BH_LAUNCH(ctx, BH_K_POSE_BA, k_pose_ba, dim3(B), dim3(PB_T), st, d_pix, d_pts, d_n, stride, Kd, iters, huber_delta, chi2_thr,
              d_R, d_t, d_chi, d_in, d_st);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(R, d_R, sizeof(double) * 9 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(t, d_t, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(chi2, d_chi, sizeof(double) * ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(inlier, d_in, ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(status, d_st, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

// ================================================================================================ local BA
struct ba_state {                 // device-resident LM control block
    double lambda, ni, chi_cur, chi_trial, rho, scale, max_diag;
    int qmax, accepted, again, terminate, chol_ok, pad;
    double red[4];                // partial sums handed to the all-reduce hook (multi-GPU global BA)
    long long dbg[8];             // k_ba_chol_sw phase clocks (bh_ba_debug_band_solve)
    // batched local BA (bh_ba_batch_optimize): the Levenberg control flow of a window, kept on the device
    int b_it, b_iters, b_need_lin, b_done, b_cur, b_last_err, b_verbose, b_fail;
    int tick[2];                  // fan-in arrival counters: [0] linearisation -> prep, [1] trial errors -> decide (zero between launches)
};

struct bh_ba {
    bh_ctx* ctx = nullptr;
    ba_K K = {};
    int C = 0, P = 0, E = 0;           // user sizes
    int Cf = 0, Pa = 0, Ea = 0;        // free+active cameras, active points, active edges
    std::vector<int> cam_slot;         // user cam -> free slot or -1
    std::vector<int> pt_slot;          // user point -> active point or -1
    std::vector<int> edge_user;        // active edge -> user edge
    std::vector<double> log;           // per iteration {chi, lambda, trials, accepted}
    int iters_done = 0;
    int have_problem = 0, verbose_last = 0;
    // device
    bh_buf cams[2], pts[2];            // state double buffer: [cur], [trial]
    int cur = 0;
    bh_buf e_cam, e_pt, e_meas;        // per active edge: user cam index, active point index, measurement
    bh_buf cslot;                      // per user cam: free slot or -1
    bh_buf p_start, p_edges, c_start, c_edges;
    bh_buf err[2], chi_e[2];           // [0] at the current state (iteration start), [1] at the trial state
    bh_buf Hpl, Hll, bl, Hpp, bp, pdiag, cdiag;
    bh_buf Dinv, db, S, bs, dxp, dxl, state, pairs, part, xch, fan;
    bh_buf join, j_start;              // per camera pair: the (edge, edge) pairs that share a landmark (Schur complement's work list)
    bh_buf c_pt, c_meas;               // row form: landmark slot / measurement of every entry of c_edges (camera-major copies)
    int cam_major = 0;                 // c_pt / c_meas are valid for the current problem
    bh_buf row_start, join_slot;       // row form (k_ba_schur_row): first pair of every row; LDS slot of every join entry's first edge
    int schur_row = 0;
    bh_buf pair_rec, join_pt;          // wave-per-pair form only: (s1, s2, first, end) per pair in one 16-B record; landmark slot per join entry
    long long join_len = 0;
    int chol_split = 0;                // two-way window solve: first block column of the separator (0 = one workgroup)
    int npairs = 0;
    int bw = 0;                        // scalar half-bandwidth of the reduced camera system (6*max|s1-s2| + 5)
    int chol_nb = 32;                  // panel width chosen for the LDS budget
    int chol_window = 0;               // on-chip sliding-window banded Cholesky (k_ba_chol_sw) instead of the panel kernel
    int chol_dense = 0;                // whole factor on chip (k_ba_chol_dense): 6 Cf <= 288, the local-BA window
    int chol_bcr = 0;                  // block cyclic reduction over the whole chip (long narrow systems: global BA)
    ba_state* h_state = nullptr;       // pinned, device-writable: k_ba_decide's verdict (polled by bh_ba_optimize)
    struct bh_ba_batch* self = nullptr; // a local window alone is a batch of one (bh_ba_optimize: device-side control flow, the host one round ahead)
    int seq = 0;
    struct bcr_dev* bcr = nullptr;
    struct dbcr_dev* dsol = nullptr;   // distributed reduced-camera solve (sharded mode, long banded systems)
    long long ar_bytes = 0;            // of comm_bytes: plain all-reduces
    int cam_parts = 1;                 // workgroups per free camera in the linearisation (k_ba_linearize: long edge lists of a dense local window are cut into parts)
    int schur_wave = 0;                // one wave per camera pair (many short edge lists) instead of a 256-thread workgroup
    int last_err_sel = 0;              // which err buffer holds "the last computeActiveErrors"
    // landmark-sharded multi-GPU mode (bh_ba_set_allreduce)
    bh_allreduce_fn ar = nullptr;
    void* ar_user = nullptr;
    int ar_ordered = 0;                // the hook enqueues on the context's stream: no host synchronisation around it
    int rank = 0, world = 1;
    double comm_ms = 0; long long comm_bytes = 0;
};

// fan-in slots of a window (ba->fan): [l_chi nbp][l_max nbp][c_max Cf][u_sum nbp][e_chi ceil(E / 256)]
#define BA_CAM_PARTS_MAX 8
static size_t ba_fan_doubles(size_t Pa, size_t Cf, size_t E)
{
    const size_t nbp = (Pa * 8 + 255) / 256, ne = (E + 255) / 256;
    return 3 * nbp + Cf + ne + 8 + (nbp + BA_CAM_PARTS_MAX * Cf + ne) / 128 + 8 + 27 * BA_CAM_PARTS_MAX * Cf + Cf / 2 + 2;      // .. + camera-part sums + camera arrival counters
}

// ---- k_ba_point: 8 lanes per active point; residual + Huber + Jacobians; Hll, bl, per-edge Hpl, per-edge error/chi
#define BP_G 8
__device__ __forceinline__ void ba_point_body(int bid, const ba_cam* __restrict__ cams, const double* __restrict__ pts,
                                                  const int* __restrict__ e_cam, const double* __restrict__ e_meas,
                                                  const int* __restrict__ cslot, const int* __restrict__ p_start,
                                                  const int* __restrict__ p_edges, int Pa, ba_K K, double delta,
                                                  double* __restrict__ Hpl, double* __restrict__ Hll, double* __restrict__ bl,
                                                  double* __restrict__ err, double* __restrict__ chi_e, double* __restrict__ pdiag,
                                                  double* __restrict__ mine = nullptr /* [2]: this thread's sum of chi, its landmark's max |diag| */)
{
    double chi_sum = 0, md_mine = 0;
    const int gid = (bid * 256 + threadIdx.x) / BP_G, sub = threadIdx.x & (BP_G - 1);
    const bool live = gid < Pa;
    const int j = live ? gid : 0;
    const double X[3] = { pts[3 * j], pts[3 * j + 1], pts[3 * j + 2] };
    double acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = 0;       // Hll upper (6) + bl (3)
    if (live) {
        for (int u = p_start[j] + sub; u < p_start[j + 1]; u += BP_G) {
            const int e = p_edges[u];
            const ba_cam* c = cams + e_cam[e];
            double er[2], Jc[12], Jp[6], r0, r1;
            d_edge_error(c->w2i, X, e_meas + 2 * e, er);
            d_edge_jac(c->w2n, c->dR, c->t, X, K, Jc, Jp);
            d_huber(er[0] * er[0] + er[1] * er[1], delta, &r0, &r1);
            err[2 * e] = er[0]; err[2 * e + 1] = er[1]; chi_e[e] = r0;
            chi_sum += r0;
            acc[0] += r1 * (Jp[0] * Jp[0] + Jp[3] * Jp[3]); acc[1] += r1 * (Jp[0] * Jp[1] + Jp[3] * Jp[4]); acc[2] += r1 * (Jp[0] * Jp[2] + Jp[3] * Jp[5]);
            acc[3] += r1 * (Jp[1] * Jp[1] + Jp[4] * Jp[4]); acc[4] += r1 * (Jp[1] * Jp[2] + Jp[4] * Jp[5]); acc[5] += r1 * (Jp[2] * Jp[2] + Jp[5] * Jp[5]);
            acc[6] -= r1 * (Jp[0] * er[0] + Jp[3] * er[1]); acc[7] -= r1 * (Jp[1] * er[0] + Jp[4] * er[1]); acc[8] -= r1 * (Jp[2] * er[0] + Jp[5] * er[1]);
            double* B = Hpl + 18 * (size_t)e;
            if (cslot[e_cam[e]] >= 0) {
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int bb = 0; bb < 3; bb++) B[3 * a + bb] = r1 * (Jc[a] * Jp[bb] + Jc[6 + a] * Jp[3 + bb]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) { acc[i] += d_xor_dpp<1>(acc[i]); acc[i] += d_xor_dpp<2>(acc[i]); acc[i] += d_xor_dpp<4>(acc[i]); }      // (DPP: 54 ds_bpermutes per wave before)
    if (live && sub == 0) {
        double* H = Hll + 9 * (size_t)j;
        H[0] = acc[0]; H[1] = acc[1]; H[2] = acc[2]; H[3] = acc[1]; H[4] = acc[3]; H[5] = acc[4]; H[6] = acc[2]; H[7] = acc[4]; H[8] = acc[5];
        bl[3 * j] = acc[6]; bl[3 * j + 1] = acc[7]; bl[3 * j + 2] = acc[8];
        md_mine = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
        pdiag[j] = md_mine;
    }
    if (mine) { mine[0] = chi_sum; mine[1] = md_mine; }
}

// ---- k_ba_cam: workgroup per free camera: Hpp (6x6) and bp (6) over the camera's edges
__device__ __forceinline__ bool ba_cam_body(int s, const ba_cam* __restrict__ cams, const double* __restrict__ pts,
                                                const int* __restrict__ e_cam, const int* __restrict__ e_pt,
                                                const double* __restrict__ e_meas, const int* __restrict__ c_start,
                                                const int* __restrict__ c_edges, ba_K K, double delta,
                                                double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ cdiag,
                                                const int* __restrict__ c_pt = nullptr, const double* __restrict__ c_meas = nullptr,
                                                double* __restrict__ md_out = nullptr /* thread 0: the camera's max |diag| */,
                                                int part = 0, int cp = 1, double* c_part = nullptr, int* c_cnt = nullptr)
{
    // Returns true (in every thread) in the workgroup that wrote the camera's blocks.  cp > 1 (a dense local window: 1 600 observations per
    // keyframe at configs[2], 48 workgroups of six rounds each while the landmark workgroups finish in one): the camera's edge list is cut into
    // cp contiguous parts, a workgroup each; a part leaves its 27 sums in c_part and counts itself in c_cnt[s], the part that arrives last adds the
    // cp partial sums IN PART ORDER (the result does not depend on which one is last) and writes Hpp / bp / cdiag.
    __shared__ double s_red[27 * 4];
    const int tid = threadIdx.x;
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    int u0 = c_start[s], u1 = c_start[s + 1];
    if (cp > 1) { const int per = (u1 - u0 + cp - 1) / cp, a = min(u1, u0 + part * per); u1 = min(u1, a + per); u0 = a; }
    // c_pt / c_meas (global-BA form): landmark slot and measurement of every c_edges entry in camera-major order, copied once per problem -
    // the loop then gathers the landmark only (the edge's camera, landmark slot and measurement were three more scattered lines per edge);
    // every edge of the list has this camera
    const ba_cam* c0 = (c_meas && u0 < u1) ? cams + e_cam[c_edges[u0]] : nullptr;
    for (int u = u0 + tid; u < u1; u += 256) {
        const int e = c_meas ? 0 : c_edges[u];
        const ba_cam* c = c_meas ? c0 : cams + e_cam[e];
        const double* X = pts + 3 * (size_t)(c_meas ? c_pt[u] : e_pt[e]);
        double er[2], Jc[12], Jp[6], r0, r1;
        d_edge_error(c->w2i, X, c_meas ? c_meas + 2 * (size_t)u : e_meas + 2 * e, er);
        d_edge_jac(c->w2n, c->dR, c->t, X, K, Jc, Jp);
        d_huber(er[0] * er[0] + er[1] * er[1], delta, &r0, &r1);
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int bb = a; bb < 6; bb++) acc[k++] += r1 * (Jc[a] * Jc[bb] + Jc[6 + a] * Jc[6 + bb]);
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] -= r1 * (Jc[a] * er[0] + Jc[6 + a] * er[1]);
    }
    d_block_sum<27, 256>(acc, s_red);
    if (cp > 1) {
        __shared__ int s_last;
        __shared__ double s_fin[27];
        if (tid < 27) { double v = 0; for (int w = 0; w < 4; w++) v += s_red[w * 27 + tid]; fan_store(c_part + ((size_t)s * cp + part) * 27 + tid, v); }      // (= acc[tid]: the same sum in the same order)
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // wave 0's 27 slot stores have left the CU
            const int prev = __hip_atomic_fetch_add(c_cnt + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = prev == cp - 1 ? 1 : 0;
            if (prev == cp - 1) __hip_atomic_store(c_cnt + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid < 27) {
            double v[BA_CAM_PARTS_MAX];
#pragma unroll
            for (int q = 0; q < BA_CAM_PARTS_MAX; q++) v[q] = q < cp ? fan_load(c_part + ((size_t)s * cp + q) * 27 + tid) : 0.0;
            double t = v[0];
#pragma unroll
            for (int q = 1; q < BA_CAM_PARTS_MAX; q++) if (q < cp) t += v[q];
            s_fin[tid] = t;
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 27; i++) acc[i] = s_fin[i];
        }
    }
    if (tid == 0) {
        double* H = Hpp + 36 * (size_t)s;
        int k = 0; double md = 0;
        for (int a = 0; a < 6; a++) for (int bb = a; bb < 6; bb++) { H[a * 6 + bb] = acc[k]; H[bb * 6 + a] = acc[k]; k++; }
        for (int a = 0; a < 6; a++) { bp[6 * s + a] = acc[21 + a]; md = fmax(md, fabs(H[a * 6 + a])); }
        cdiag[s] = md;
        if (md_out) *md_out = md;
    }
    return true;
}

// ---- k_ba_partial: first stage of the big sums (global-BA sizes): block b sums its contiguous slice of the edges / points
// in a fixed order into part[b] = {sum chi, sum dx (lambda dx + b)}; k_ba_prep / k_ba_decide add the BA_NPART partials.
// ---- k_ba_linearize: the per-landmark and the per-camera halves of buildSystem in ONE launch (workgroups [0, nbp) take landmarks,
// the rest one free camera each): they are independent, and as two launches of 18 us each the second - 48 workgroups on a local
// window - left most of the chip idle.
// What a workgroup of the linearisation leaves for the fan-in: slot bx of chi / max (landmark workgroups), slot s of cmax (camera workgroups)
struct ba_fan {
    double *l_chi, *l_max, *c_max;      // linearisation: [nbp], [nbp], [Cf]
    double *u_sum, *e_chi;              // trial: [landmark workgroups of the update], [workgroups of the error kernel]
    int* grp;                           // arrival counters per group of 64 workgroups (zero between launches)
    double* c_part;                     // [Cf][cp][27] sums of a camera's parts (cp > 1)
    int* c_cnt;                         // [Cf] parts of a camera that have arrived (zero between launches)
    int cp;                             // workgroups per free camera
};
static ba_fan ba_fan_of(const bh_ba* ba)
{
    const size_t nbp = ((size_t)ba->Pa * BP_G + 255) / 256;
    double* f = (double*)ba->fan.p;
    ba_fan F; F.l_chi = f; F.l_max = f + nbp; F.c_max = f + 2 * nbp; F.u_sum = f + 2 * nbp + ba->Cf; F.e_chi = f + 3 * nbp + ba->Cf;
    const size_t ne = ((size_t)ba->Ea + 255) / 256;
    F.grp = (int*)(f + 3 * nbp + ba->Cf + ne + 1);
    F.c_part = f + 3 * nbp + ba->Cf + ne + 8 + (nbp + BA_CAM_PARTS_MAX * (size_t)ba->Cf + ne) / 128 + 8;
    F.c_cnt = (int*)(F.c_part + 27 * BA_CAM_PARTS_MAX * (size_t)ba->Cf);
    F.cp = ba->cam_parts;
    return F;
}
__device__ __forceinline__ void ba_lin_fan_store(bool point_block, int slot, const double* mine, double md_cam, const ba_fan& F, double* s8)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (point_block) {
        const double cs = fan_block_sum256(mine[0], s8);
        double m = mine[1];
        for (int k = 32; k > 0; k >>= 1) m = fmax(m, d_shfl_xor(m, k));
        if (lane == 0) s8[4 + wv] = m;
        __syncthreads();
        if (tid == 0) { fan_store(F.l_chi + slot, cs); fan_store(F.l_max + slot, fmax(fmax(s8[4], s8[5]), fmax(s8[6], s8[7]))); }
    } else if (tid == 0 && slot >= 0) fan_store(F.c_max + slot, md_cam);      // (slot < 0: a camera part that was not the last to arrive)
}
// k_ba_prep's work from the slots: one wave
// (phase 1, multi-GPU: this rank's chi and the maximum over ITS landmarks go to S->red for the all-reduce hook; k_ba_prep's phase 2 folds in the
// cameras' maxima - replicated, known only after the all-reduce of Hpp - and takes the decision)
__device__ __forceinline__ void ba_prep_fan(const ba_fan& F, int nbp, int Cf, int first, int phase, ba_state* __restrict__ S)
{
    __shared__ double s_w[4];
    const double chi = fan_sum_wg(F.l_chi, nbp, s_w);
    double md = 0;
    if (first) { md = fan_max_wg(F.l_max, nbp, 0.0, s_w); if (phase == 0) md = fan_max_wg(F.c_max, Cf, md, s_w); }
    if (threadIdx.x == 0) {
        if (phase == 1) { S->red[0] = chi; S->red[1] = md; return; }
        S->chi_cur = chi;
        if (first) { S->max_diag = md; S->lambda = 1e-5 * md; S->ni = 2; }
        S->qmax = 0; S->accepted = 0; S->again = 0; S->terminate = 0; S->rho = 0; S->chol_ok = 1;
    }
}
__global__ __launch_bounds__(256) void k_ba_linearize(int nbp, const ba_cam* __restrict__ cams, const double* __restrict__ pts,
                                                      const int* __restrict__ e_cam, const int* __restrict__ e_pt, const double* __restrict__ e_meas,
                                                      const int* __restrict__ cslot, const int* __restrict__ p_start, const int* __restrict__ p_edges,
                                                      const int* __restrict__ c_start, const int* __restrict__ c_edges, int Pa, ba_K K, double delta,
                                                      double* __restrict__ Hpl, double* __restrict__ Hll, double* __restrict__ bl,
                                                      double* __restrict__ err, double* __restrict__ chi_e, double* __restrict__ pdiag,
                                                      double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ cdiag,
                                                      const int* __restrict__ c_pt, const double* __restrict__ c_meas,
                                                      ba_fan F, ba_state* __restrict__ S, int first, int phase)
{
    __shared__ double s_f[8];
    double mine[2] = { 0, 0 }, md_cam = 0;
    const bool pb = (int)blockIdx.x < nbp;
    const int cp = F.cp > 1 ? F.cp : 1, c = (int)blockIdx.x - nbp, cs = cp > 1 ? c / cp : c;      // camera workgroups: camera cs, part c - cs * cp
    bool fin = true;
    if (pb) ba_point_body(blockIdx.x, cams, pts, e_cam, e_meas, cslot, p_start, p_edges, Pa, K, delta, Hpl, Hll, bl, err, chi_e, pdiag, mine);
    else fin = ba_cam_body(cs, cams, pts, e_cam, e_pt, e_meas, c_start, c_edges, K, delta, Hpp, bp, cdiag, c_pt, c_meas, &md_cam, c - cs * cp, cp, F.c_part, F.c_cnt);
    if (!F.l_chi) return;                                        // (BH_BA_FAN=0: the sums as separate launches, the A/B reference)
    ba_lin_fan_store(pb, pb ? (int)blockIdx.x : (fin ? cs : -1), mine, md_cam, F, s_f);
    if (fan_arrive(&S->tick[0], F.grp, (int)blockIdx.x, (int)gridDim.x)) ba_prep_fan(F, nbp, ((int)gridDim.x - nbp) / cp, first, phase, S);
}

#define BA_NPART 256
__device__ __forceinline__ void ba_partial_body(const double* __restrict__ chi_e, int Ea, const double* __restrict__ dxl,
                                                     const double* __restrict__ bl, int n3, const ba_state* __restrict__ S,
                                                     double* __restrict__ part)
{
    __shared__ double s_red[2 * 16];
    const int tid = threadIdx.x, b = blockIdx.x;
    double v[2] = { 0, 0 };
    { const int per = (Ea + BA_NPART - 1) / BA_NPART, lo = b * per, hi = min(Ea, lo + per);
      for (int e = lo + tid; e < hi; e += 1024) v[0] += chi_e[e]; }
    if (dxl) {
        const double lambda = S->lambda;
        const int per = (n3 + BA_NPART - 1) / BA_NPART, lo = b * per, hi = min(n3, lo + per);
        for (int i = lo + tid; i < hi; i += 1024) v[1] += dxl[i] * (lambda * dxl[i] + bl[i]);
    }
    d_block_sum<2, 1024>(v, s_red);
    if (tid == 0) { part[2 * b] = v[0]; part[2 * b + 1] = v[1]; }
}
__global__ __launch_bounds__(1024) void k_ba_partial(const double* __restrict__ chi_e, int Ea, const double* __restrict__ dxl,
                                                     const double* __restrict__ bl, int n3, const ba_state* __restrict__ S,
                                                     double* __restrict__ part)
{
    ba_partial_body(chi_e, Ea, dxl, bl, n3, S, part);
}

// ---- k_ba_prep: chi at the current state; lambda init on the first iteration
__device__ __forceinline__ void ba_prep_body(const double* __restrict__ chi_e, int Ea, const double* __restrict__ pdiag, int Pa,
                                                  const double* __restrict__ cdiag, int Cf, int first, int phase,
                                                  const double* __restrict__ part, ba_state* __restrict__ S)
{
    // phase 0: single GPU (sums + decision in one launch); 1: partial sums -> S->red; 2: decision from the all-reduced S->red
    __shared__ double s_red[16];
    __shared__ double s_max[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double v[1] = { 0 };
    if (phase < 2) {
        if (part) { if (tid < BA_NPART) v[0] = part[2 * tid]; }
        else for (int e = tid; e < Ea; e += 1024) v[0] += chi_e[e];
    }
    d_block_sum<1, 1024>(v, s_red);
    double md = 0;
    if (first && phase != 2) {
        if (phase != 3) for (int j = tid; j < Pa; j += 1024) md = fmax(md, pdiag[j]);
        for (int c = tid; c < Cf; c += 1024) md = fmax(md, cdiag[c]);
        for (int m = 32; m > 0; m >>= 1) md = fmax(md, d_shfl_xor(md, m));
        if (lane == 0) s_max[wv] = md;
        __syncthreads();
        md = 0; for (int w = 0; w < 16; w++) md = fmax(md, s_max[w]);
    }
    if (tid == 0) {
        if (phase == 1) { S->red[0] = v[0]; S->red[1] = md; return; }
        if (phase == 2) { v[0] = S->red[0]; md = S->red[1]; }
        if (phase == 3) { v[0] = S->red[0]; md = fmax(S->red[1], md); }      // fan-in form: red[1] holds the landmarks' maximum, md the cameras'
        S->chi_cur = v[0];
        if (first) { S->max_diag = md; S->lambda = 1e-5 * md; S->ni = 2; }
        S->qmax = 0; S->accepted = 0; S->again = 0; S->terminate = 0; S->rho = 0; S->chol_ok = 1;
    }
}
__global__ __launch_bounds__(1024) void k_ba_prep(const double* __restrict__ chi_e, int Ea, const double* __restrict__ pdiag, int Pa,
                                                  const double* __restrict__ cdiag, int Cf, int first, int phase,
                                                  const double* __restrict__ part, ba_state* __restrict__ S)
{
    ba_prep_body(chi_e, Ea, pdiag, Pa, cdiag, Cf, first, phase, part, S);
}

__device__ __forceinline__ bool d_inv3(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return isfinite(id);
}

// ---- k_ba_dinv: per point (Hll + lambda I)^-1 and Dinv * bl
__device__ __forceinline__ void ba_dinv_body(int bid, const double* __restrict__ Hll, const double* __restrict__ bl, int Pa,
                                                 ba_state* __restrict__ S, double* __restrict__ Dinv, double* __restrict__ db)
{
    const int j = bid * 256 + threadIdx.x;
    if (j >= Pa) return;
    const double lambda = S->lambda;
    double D[9], Di[9];
#pragma unroll
    for (int i = 0; i < 9; i++) D[i] = Hll[9 * (size_t)j + i];
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    if (!d_inv3(D, Di)) S->chol_ok = 0;              // benign race: every writer stores 0; k_ba_prep / k_ba_decide re-arm it to 1
#pragma unroll
    for (int i = 0; i < 9; i++) Dinv[9 * (size_t)j + i] = Di[i];
#pragma unroll
    for (int a = 0; a < 3; a++) db[3 * j + a] = Di[3 * a] * bl[3 * j] + Di[3 * a + 1] * bl[3 * j + 1] + Di[3 * a + 2] * bl[3 * j + 2];
}
// (the workgroups behind the landmarks clear the reduced camera system of a local window: one launch less per trial than a memset)
__global__ __launch_bounds__(256) void k_ba_dinv(const double* __restrict__ Hll, const double* __restrict__ bl, int Pa,
                                                 ba_state* __restrict__ S, double* __restrict__ Dinv, double* __restrict__ db,
                                                 int pblocks, double* __restrict__ Sm, long long nS)
{
    if ((int)blockIdx.x < pblocks) { ba_dinv_body(blockIdx.x, Hll, bl, Pa, S, Dinv, db); return; }
    const long long step = (long long)((int)gridDim.x - pblocks) * 256;
    for (long long i = (long long)((int)blockIdx.x - pblocks) * 256 + threadIdx.x; i < nS; i += step) Sm[i] = 0.0;
}

// The wave-per-pair form walks pairs -> j_start -> join -> e_pt -> Dinv: five dependent global loads before the first block arrives, for
// ~84 join entries per pair (1.3 rounds of a wave) - the chain, not the bytes, set its 1.2 ms at config 5.  Built once per problem:
// pair_rec[p] = (s1, s2, first, end) in one 16-B load, join_pt[t] = landmark slot of join entry t (read next to the entry, coalesced).
__global__ void k_ba_schur_aux(const int2* __restrict__ pairs, const int* __restrict__ j_start, int npairs, const int2* __restrict__ join,
                               const int* __restrict__ e_pt, long long M, int4* __restrict__ pair_rec, int* __restrict__ join_pt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < npairs) pair_rec[i] = make_int4(pairs[i].x, pairs[i].y, j_start[i], j_start[i + 1]);
    if (i < M) join_pt[i] = e_pt[join[i].x];
}

// ---- k_ba_schur_S: workgroup per camera pair (s1 <= s2): S[s1][s2] = [s1==s2](Hpp + lambda I) - sum_j Hpl_1j Dinv_j Hpl_2j'
template <int NT>
__device__ __forceinline__ void ba_schur_S_body(const int bid, const int ngrid, const int2* __restrict__ pairs, const int* __restrict__ j_start,
                                                    const int2* __restrict__ join, const int* __restrict__ e_pt,
                                                    const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                                    const double* __restrict__ Hpp, const ba_state* __restrict__ St,
                                                    int bw, int add_diag, double* __restrict__ Sm, int npairs,
                                                    const int4* __restrict__ pair_rec = nullptr, const int* __restrict__ join_pt = nullptr)
{
    // NT = 256 for a few long lists (local BA), one wave per pair for many short ones (global BA).  join[j_start[p] .. j_start[p+1])
    // lists the (edge of s1, edge of s2) pairs that observe the same landmark, in landmark order: built once per problem
    // (ba_build_device); round 1 re-derived it in every trial by a binary search per edge, 90 % of which found nothing.
    __shared__ double s_red[36 * (NT / 64)];
    const int tid = threadIdx.x;
    // workgroups are dealt to the 8 XCDs round-robin; each XCD takes one contiguous eighth of the (s1, s2)-sorted pair list, so that
    // the blocks of camera s1 (shared by its ~2 x band pairs in a row) and of s2 (met again at (s1 + 1, s2), one band later) are still
    // in that XCD's own L2 - dealt pair by pair, every reuse was a miss in seven other L2s
    const int chunk = (ngrid + 7) >> 3;
    const int pid = (bid & 7) * chunk + (bid >> 3);
    if (pid >= npairs) return;
    int s1, s2, t0, t1;
    if (NT == 64 && pair_rec) { const int4 pr = pair_rec[pid]; s1 = pr.x; s2 = pr.y; t0 = pr.z; t1 = pr.w; }
    else { s1 = pairs[pid].x; s2 = pairs[pid].y; t0 = j_start[pid]; t1 = j_start[pid + 1]; }
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
    for (int t = t0 + tid; t < t1; t += NT) {
        const int2 ee = join[t];
        const int e1 = ee.x, e2 = ee.y, j = (NT == 64 && join_pt) ? join_pt[t] : e_pt[e1];
        // the 144-B blocks are 16-B aligned (hipMalloc base, 18 doubles each): nine 16-B loads per block instead of eighteen 8-B ones -
        // every lane reads a different block, so the load count is what the texture path is billed for
        double B1[18], B2[18];
        {
            const double2* q1 = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)e1); const double2* q2 = reinterpret_cast<const double2*>(Hpl + 18 * (size_t)e2);
#pragma unroll
            for (int q = 0; q < 9; q++) { const double2 u = q1[q], w = q2[q]; B1[2 * q] = u.x; B1[2 * q + 1] = u.y; B2[2 * q] = w.x; B2[2 * q + 1] = w.y; }
        }
        const double* Di = Dinv + 9 * (size_t)j;
        double BD[18];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int bb = 0; bb < 3; bb++) BD[3 * a + bb] = B1[3 * a] * Di[bb] + B1[3 * a + 1] * Di[3 + bb] + B1[3 * a + 2] * Di[6 + bb];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int bb = 0; bb < 6; bb++) acc[6 * a + bb] += BD[3 * a] * B2[3 * bb] + BD[3 * a + 1] * B2[3 * bb + 1] + BD[3 * a + 2] * B2[3 * bb + 2];
    }
    double v = 0;
    if constexpr (NT == 64) {
        // one wave: the 36 x 64 partial sums are transposed through LDS (pitch 65: lanes k read column j of rows k without bank conflicts)
        // and lane k < 36 adds up element k in lane order, four interleaved chains - 36 butterfly reductions were 432 ds_bpermutes.
        // (18.7 KB: two workgroups per SIMD, which is also what the 188 VGPRs allow; forcing 3 or 4 spills and was measured slower)
        __shared__ double s_tr[36 * 65];
#pragma unroll
        for (int i = 0; i < 36; i++) s_tr[i * 65 + tid] = acc[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < 36) {
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (int l = 0; l < 64; l += 4) { p0 += s_tr[tid * 65 + l]; p1 += s_tr[tid * 65 + l + 1]; p2 += s_tr[tid * 65 + l + 2]; p3 += s_tr[tid * 65 + l + 3]; }
            v = (p0 + p1) + (p2 + p3);
        }
    } else {
        // wave sums, then thread e < 36 adds the waves' partials of element e in wave order (same order as d_block_sum; reading the
        // result as acc[tid] would index the accumulator array dynamically and move all 36 accumulators to scratch memory)
        d_wave_sum_rs<36>(acc);
        d_wave_sum_rs_store<36>(acc, s_red + (tid >> 6) * 36);
        __syncthreads();
        if (tid < 36)
            for (int w = 0; w < NT / 64; w++) v += s_red[w * 36 + tid];
    }
    if (tid < 36) {
        const int a = tid / 6, bb = tid - a * 6;
        v = -v;
        if (s1 == s2 && add_diag) { v += Hpp[36 * (size_t)s1 + tid]; if (a == bb) v += St->lambda; }
        // lower band storage: element (row, col), row >= col, at Sm[row*LD + (col - row + bw)]; s1 <= s2
        const int row = 6 * s2 + bb, col = 6 * s1 + a;
        if (col <= row) Sm[(size_t)row * (bw + 1) + (col - row + bw)] = v;
    }
}

// ---- k_ba_schur_b: per free camera  b~ = bp - sum_e Hpl_e (Dinv bl)_pt(e)
__device__ __forceinline__ void ba_schur_b_body(const int s, const int* __restrict__ c_start, const int* __restrict__ c_edges,
                                                    const int* __restrict__ e_pt, const double* __restrict__ Hpl,
                                                    const double* __restrict__ db, const double* __restrict__ bp,
                                                    int use_bp, double* __restrict__ bs)
{
    __shared__ double s_red[6 * 4];
    const int tid = threadIdx.x;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int u = c_start[s] + tid; u < c_start[s + 1]; u += 256) {
        const int e = c_edges[u];
        const double* B = Hpl + 18 * (size_t)e; const double* d = db + 3 * (size_t)e_pt[e];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] += B[3 * a] * d[0] + B[3 * a + 1] * d[1] + B[3 * a + 2] * d[2];
    }
    d_block_sum<6, 256>(acc, s_red);
    if (tid < 6) bs[6 * s + tid] = (use_bp ? bp[6 * s + tid] : 0.0) - acc[tid];
}

// ---- k_ba_schur_row: the global-BA form of S and b~, one workgroup per free camera s1 (row of the block-banded S).
// What bounds the wave-per-pair kernel at config 5 (1.1 ms) is cache lines, not bytes or flops: every join entry gathers three
// scattered blocks - Hpl[e1], Hpl[e2] (144 B: one or two 128-B lines each) and Dinv[j] - 4.2 lines per entry, 17.3 M entries, and the
// e2 blocks are met again only ~40 rows (15 MB of gathers) later, i.e. from HBM (tools/vmem_rate.hip: a line costs a CU 3.4 cycles from
// L2, 13-16 from HBM).  Here the ~784 edges of camera s1 are staged ONCE per row as BD_e = Hpl_e Dinv_pt(e) in LDS (144 B x SR_CAP slots),
// so an entry of the row's 40 off-diagonal pairs gathers only Hpl[e2] (1.6 lines) and reads BD from LDS by its slot (join_slot, built
// once per problem); b~ of the row needs the same blocks and rides along (k_ba_schur_b re-read all of Hpl: 0.33 ms).
//   staging       thread per edge of s1: BD -> LDS, b~ += Hpl_e db_pt(e)
//   diagonal pair all 512 threads stride its join list (duplicate observations of a landmark by one camera give e1 != e2 entries)
//   other pairs   a wave per pair, lanes stride the pair's join list, reduce-scatter on v_permlane*_swap + DPP, lanes 0/16/32/48 store
// All sums in fixed order (results do not depend on timing); the order differs from k_ba_schur_S's, the values agree to rounding.
#define SR_T 512
#define SR_CAP 1024
__device__ __forceinline__ void sr_load18(const double* __restrict__ p, double* B)
{
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 9; i++) { const double2 u = q[i]; B[2 * i] = u.x; B[2 * i + 1] = u.y; }
}
__device__ __forceinline__ void sr_bd(const double* B1, const double* Di, double* BD)
{
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int bb = 0; bb < 3; bb++) BD[3 * a + bb] = B1[3 * a] * Di[bb] + B1[3 * a + 1] * Di[3 + bb] + B1[3 * a + 2] * Di[6 + bb];
}
__global__ __launch_bounds__(SR_T) void k_ba_schur_row(const int* __restrict__ row_start, const int4* __restrict__ pair_rec, const int2* __restrict__ join,
                                                       const int* __restrict__ join_slot, const int* __restrict__ c_start, const int* __restrict__ c_edges,
                                                       const int* __restrict__ c_pt, const int* __restrict__ e_pt, const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                                       const double* __restrict__ db, const double* __restrict__ Hpp, const double* __restrict__ bp,
                                                       const ba_state* __restrict__ St, int bw, int add_diag, int Cf, double* __restrict__ Sm, double* __restrict__ bs)
{
    extern __shared__ double s_bd[];                             // [SR_CAP][18]
    __shared__ double s_red[42 * (SR_T / 64)];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // contiguous eighths of the rows per XCD (workgroups b and b + 8 share one): the e2 blocks of neighbouring rows overlap
    const int chunk = ((int)gridDim.x + 7) >> 3;
    const int s1 = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    if (s1 >= Cf) return;
    const int u0 = c_start[s1], n1 = c_start[s1 + 1] - u0;
    const int first = row_start[s1], last = row_start[s1 + 1];
    // no camera observes a landmark twice (row_start[Cf + 1], found once per problem): the diagonal pair's join list is (e, e) for every edge
    // of s1, and its block is summed while the blocks are staged - one phase of dependent gathers less per row
    const bool simple_diag = row_start[Cf + 1] == 0;
    // (the records of this wave's pairs are needed two barriers from here: asked for now)
    const int p_first = first + 1 + wv;
    int4 pr_first = p_first < last ? pair_rec[p_first] : make_int4(0, 0, 0, 0);
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0;
    for (int u = tid; u < n1; u += SR_T) {
        const int e = c_edges[u0 + u], j = c_pt[u0 + u];
        double B1[18], BD[18];
        sr_load18(Hpl + 18 * (size_t)e, B1);
        const double* Di = Dinv + 9 * (size_t)j; const double* d = db + 3 * (size_t)j;
        sr_bd(B1, Di, BD);
        if (u < SR_CAP) {
            double2* o = reinterpret_cast<double2*>(s_bd + 18 * (size_t)u);
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = make_double2(BD[2 * i], BD[2 * i + 1]);
        }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[36 + a] += B1[3 * a] * d[0] + B1[3 * a + 1] * d[1] + B1[3 * a + 2] * d[2];
        if (simple_diag) {
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int bb = 0; bb < 6; bb++) acc[6 * a + bb] += BD[3 * a] * B1[3 * bb] + BD[3 * a + 1] * B1[3 * bb + 1] + BD[3 * a + 2] * B1[3 * bb + 2];
        }
    }
    __syncthreads();
    // one join entry: acc += BD(e1) Hpl(e2)'
    auto entry = [&](int slot, int2 ee, double* a36) {
        double BD[18], B2[18];
        if (slot < SR_CAP) {
            const double2* q = reinterpret_cast<const double2*>(s_bd + 18 * (size_t)slot);
#pragma unroll
            for (int i = 0; i < 9; i++) { const double2 u = q[i]; BD[2 * i] = u.x; BD[2 * i + 1] = u.y; }
        } else {                                                 // a camera with more than SR_CAP observations: the tail comes from global memory
            double B1[18];
            sr_load18(Hpl + 18 * (size_t)ee.x, B1);
            sr_bd(B1, Dinv + 9 * (size_t)e_pt[ee.x], BD);
        }
        sr_load18(Hpl + 18 * (size_t)ee.y, B2);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int bb = 0; bb < 6; bb++) a36[6 * a + bb] += BD[3 * a] * B2[3 * bb] + BD[3 * a + 1] * B2[3 * bb + 1] + BD[3 * a + 2] * B2[3 * bb + 2];
    };
    // the diagonal pair (the first of the row: pairs are sorted and every camera has its diagonal block)
    int p0 = first;
    if (first < last) {
        const int4 pr = pair_rec[first];
        if (pr.y == s1) { p0 = first + 1; if (!simple_diag) for (int t = pr.z + tid; t < pr.w; t += SR_T) entry(join_slot[t], join[t], acc); }
    }
    d_wave_sum_rs<42>(acc);
    d_wave_sum_rs_store<42>(acc, s_red + wv * 42);
    __syncthreads();
    if (tid < 42) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < SR_T / 64; w++) v += s_red[w * 42 + tid];
        v = -v;
        if (tid < 36) {
            const int a = tid / 6, bb = tid - a * 6;
            if (add_diag) { v += Hpp[36 * (size_t)s1 + tid]; if (a == bb) v += St->lambda; }
            const int row = 6 * s1 + bb, col = 6 * s1 + a;       // lower band storage: element (row, col), row >= col, at Sm[row * LD + (col - row + bw)]
            if (col <= row) Sm[(size_t)row * (bw + 1) + (col - row + bw)] = v;
        } else bs[6 * s1 + (tid - 36)] = (add_diag ? bp[6 * s1 + (tid - 36)] : 0.0) + v;
    }
    // the other pairs of the row: a wave each.  The indices of a round (pair record, LDS slot, second edge) are fetched one round ahead:
    // what a round then waits for is the gather of Hpl[e2] alone.
    int p = p0 + wv;
    int4 pr = p == p_first ? pr_first : (p < last ? pair_rec[p] : make_int4(0, 0, 0, 0));
    int t = pr.z + lane;
    int slot = 0; int2 ee = make_int2(0, 0);
    if (p < last && t < pr.w) { slot = join_slot[t]; ee = join[t]; }
    while (p < last) {
        double a36[36];
#pragma unroll
        for (int i = 0; i < 36; i++) a36[i] = 0;
        const int4 cur = pr;
        const int pn = p + SR_T / 64;
        int4 nx = make_int4(0, 0, 0, 0);
        if (pn < last) nx = pair_rec[pn];
        for (;;) {
            const bool live = t < cur.w;
            const int cslot = slot; const int2 cee = ee;
            const int tn = t + 64;
            const bool more = __any(tn < cur.w);                 // wave-uniform: another round of this pair
            if (more) { if (tn < cur.w) { slot = join_slot[tn]; ee = join[tn]; } }
            else if (pn < last) { const int t2 = nx.z + lane; if (t2 < nx.w) { slot = join_slot[t2]; ee = join[t2]; } }
            if (live) entry(cslot, cee, a36);
            if (!more) break;
            t = tn;
        }
        d_wave_sum_rs<36>(a36);
        if ((lane & 15) == 0) {                                  // (the layout d_wave_sum_rs leaves: see d_wave_sum_rs_store)
            const int q = lane >> 4;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int idx = i + 9 * (q & 1) + 18 * (q >> 1);
                const int a = idx / 6, bb = idx - a * 6;
                const int row = 6 * cur.y + bb, col = 6 * cur.x + a;
                if (col <= row) Sm[(size_t)row * (bw + 1) + (col - row + bw)] = -a36[i];
            }
        }
        p = pn; pr = nx; t = nx.z + lane;
    }
}

// join_slot[t] = position of join entry t's first edge inside its camera's c_edges list (the LDS slot k_ba_schur_row stages it in);
// row_start[s] = first pair of row s in the (s1, s2)-sorted pair list.  Once per problem.
__global__ void k_ba_schur_rank(const int* __restrict__ c_start, const int* __restrict__ c_edges, const int* __restrict__ e_pt, const double* __restrict__ e_meas,
                                int Cf, int* __restrict__ e_rank, int* __restrict__ c_pt, double* __restrict__ c_meas)
{
    const int s = blockIdx.x;
    for (int u = c_start[s] + threadIdx.x; u < c_start[s + 1]; u += 256) {
        const int e = c_edges[u];
        e_rank[e] = u - c_start[s]; c_pt[u] = e_pt[e];
        c_meas[2 * (size_t)u] = e_meas[2 * (size_t)e]; c_meas[2 * (size_t)u + 1] = e_meas[2 * (size_t)e + 1];
    }
}
__global__ void k_ba_schur_rows(const int2* __restrict__ pairs, int npairs, int Cf, const int2* __restrict__ join, const int* __restrict__ e_rank, long long M,
                                const int* __restrict__ e_cam, int* __restrict__ row_start, int* __restrict__ join_slot)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < npairs) { const int s = pairs[i].x; if (i == 0 || pairs[i - 1].x != s) row_start[s] = (int)i; }
    if (i == 0) row_start[Cf] = npairs;
    if (i < M) {
        const int2 ee = join[i];
        join_slot[i] = e_rank[ee.x];
        if (ee.x != ee.y && e_cam[ee.x] == e_cam[ee.y]) row_start[Cf + 1] = 1;      // one camera, one landmark, two observations (every writer stores 1)
    }
}

// The kernels: S alone (one wave per pair: the global-BA form), b~ alone, and both in one launch for the 256-thread form (a local
// window: 48 workgroups of b~ next to 1 176 of S instead of after them).
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_ba_schur_S(const int2* __restrict__ pairs, const int* __restrict__ j_start,
                                                    const int2* __restrict__ join, const int* __restrict__ e_pt,
                                                    const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                                    const double* __restrict__ Hpp, const ba_state* __restrict__ St,
                                                    int bw, int add_diag, double* __restrict__ Sm, int npairs,
                                                    const int4* __restrict__ pair_rec, const int* __restrict__ join_pt)
{
    ba_schur_S_body<NT>(blockIdx.x, gridDim.x, pairs, j_start, join, e_pt, Hpl, Dinv, Hpp, St, bw, add_diag, Sm, npairs, pair_rec, join_pt);
}
__global__ __launch_bounds__(256) void k_ba_schur_b(const int* __restrict__ c_start, const int* __restrict__ c_edges,
                                                    const int* __restrict__ e_pt, const double* __restrict__ Hpl,
                                                    const double* __restrict__ db, const double* __restrict__ bp,
                                                    int use_bp, double* __restrict__ bs)
{
    ba_schur_b_body(blockIdx.x, c_start, c_edges, e_pt, Hpl, db, bp, use_bp, bs);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_ba_schur_Sb(int ngS, const int2* __restrict__ pairs, const int* __restrict__ j_start,
                                                    const int2* __restrict__ join, const int* __restrict__ e_pt,
                                                    const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                                    const double* __restrict__ Hpp, const ba_state* __restrict__ St,
                                                    int bw, int add_diag, double* __restrict__ Sm, int npairs,
                                                    const int* __restrict__ c_start, const int* __restrict__ c_edges, const double* __restrict__ db,
                                                    const double* __restrict__ bp, int use_bp, double* __restrict__ bs)
{
    if ((int)blockIdx.x < ngS) ba_schur_S_body<256>(blockIdx.x, ngS, pairs, j_start, join, e_pt, Hpl, Dinv, Hpp, St, bw, add_diag, Sm, npairs);
    else ba_schur_b_body(blockIdx.x - ngS, c_start, c_edges, e_pt, Hpl, db, bp, use_bp, bs);
}

// ---- k_ba_chol: one workgroup, blocked right-looking Cholesky of the BANDED reduced camera system + both substitutions.
// S is stored as a lower band: row i holds columns i-bw .. i at A[i*LD + (c - i + bw)], LD = bw+1 (bw = n-1 is the dense
// case).  Per NB-column panel: the in-band part of the panel (min(n-j0, NB+bw) rows x NB) is staged in LDS; wave 0 factors
// the NB x NB diagonal block in registers; one lane per row does the triangular solve of the rows below; the trailing update
// A22 -= L21 L21' runs on the matrix cores (v_mfma_f64_16x16x4_f64, one 16x16 tile per wave and step, operands straight
// from the LDS panel); the forward substitution rides along panel by panel, the backward one re-stages the panels in
// reverse.  The right-hand side / solution lives in global memory (x).  All sums in fixed order.
typedef double ba_d4 __attribute__((ext_vector_type(4)));
#define AB(i, c) A[(size_t)(i) * LD + ((c) - (i) + bw)]

template <int NB>
__global__ __launch_bounds__(1024) void k_ba_chol(double* __restrict__ A, const double* __restrict__ bs, int n, int bw,
                                                  double* __restrict__ x, ba_state* __restrict__ St)
{
    extern __shared__ double sm[];
    constexpr int PP = NB + 1;
    const int LD = bw + 1;
    const int mmax = min(n, NB + bw);
    double* P = sm;                                   // panel, row pitch NB+1
    double* part = sm + (size_t)mmax * PP;            // [1024/NB][NB] partial sums of the backward pass / exchange buffer
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_fail = St->chol_ok ? 0 : 1;
    for (int i = tid; i < n; i += 1024) x[i] = bs[i];
    __syncthreads();
    for (int j0 = 0; j0 < n && !s_fail; j0 += NB) {
        const int nbw = min(NB, n - j0), m = min(n - j0, NB + bw);
        for (int idx = tid; idx < m * NB; idx += 1024) {
            const int r = idx / NB, c = idx - r * NB;
            P[r * PP + c] = (c < nbw && c <= r && c >= r - bw) ? AB(j0 + r, j0 + c) : 0.0;      // in-band lower part only
        }
        __syncthreads();
        if (wv == 0) {                                // diagonal block in registers: lane l owns row l; column c is broadcast with
            double row[NB];                           // v_readlane (no LDS round trip, no fences), sqrt and 1/sqrt by a Goldschmidt chain
#pragma unroll
            for (int k = 0; k < NB; k++) row[k] = lane < nbw ? P[lane * PP + k] : (k == lane ? 1.0 : 0.0);     // padding rows: identity
            bool bad = false;
            double myinv = 0.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const double dg = d_readlane(row[c], c);
                bad = bad | (c < nbw && (!(dg > 0) | !(dg < 1.0e300)));
                const double y0 = __builtin_amdgcn_rsq(dg);
                double g = dg * y0, h = 0.5 * y0;
                double rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
                rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
                const double id = h + h;
                myinv = lane == c ? id : myinv;
                const double lc = lane == c ? g : row[c] * id;
                row[c] = lane >= c ? lc : row[c];
#pragma unroll
                for (int c2 = c + 1; c2 < NB; c2++) {
                    const double l2 = d_readlane(lc, c2);
                    row[c2] = lane >= c2 ? __builtin_fma(-lc, l2, row[c2]) : row[c2];
                }
            }
            if (lane < nbw) {
#pragma unroll
                for (int k = 0; k < NB; k++) P[lane * PP + k] = row[k];
            }
            if (lane < NB) part[128 + lane] = myinv;  // 1 / L[c][c] for the rows below and the forward substitution
            if (bad && lane == 0) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) break;
        for (int r = nbw + tid; r < m; r += 1024) {   // rows below the diagonal block: L21 = A21 * L11^-T, row in registers
            double rw[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) rw[k] = P[r * PP + k];
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (c < nbw) {
                    double s2 = rw[c];
#pragma unroll
                    for (int k = 0; k < c; k++) s2 -= rw[k] * P[c * PP + k];
                    rw[c] = s2 * part[128 + c];
                } else rw[c] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < NB; k++) P[r * PP + k] = rw[k];
        }
        if (wv == 0) {                                // forward substitution inside the panel: lane l owns y[j0+l]
            double yl = lane < nbw ? x[j0 + lane] : 0.0;
            for (int c = 0; c < nbw; c++) {
                const double yc = d_shfl_idx(yl, c) * part[128 + c];
                if (lane == c) yl = yc;
                else if (lane > c && lane < nbw) yl -= P[lane * PP + c] * yc;
            }
            if (lane < nbw) x[j0 + lane] = yl;
        }
        __syncthreads();
        for (int idx = tid; idx < m * nbw; idx += 1024) {
            const int r = idx / nbw, c = idx - r * nbw;
            if (c <= r && c >= r - bw) AB(j0 + r, j0 + c) = P[r * PP + c];
        }
        for (int r = nbw + tid; r < m; r += 1024) {   // right-hand side below the panel
            double s2 = x[j0 + r];
            for (int c = 0; c < nbw; c++) s2 -= P[r * PP + c] * x[j0 + c];
            x[j0 + r] = s2;
        }
        // trailing update on the matrix cores: 16x16 tiles of the lower triangle, K = NB in steps of 4
        const int mt = m - nbw, nt = (mt + 15) >> 4;
        for (int t = wv; t < nt * (nt + 1) / 2; t += 16) {
            int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) ti--;
            while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
            const int tj = t - ti * (ti + 1) / 2;
            const int ra = nbw + 16 * ti + (lane & 15), rb = nbw + 16 * tj + (lane & 15), kq = lane >> 4;
            const int col = nbw + 16 * tj + (lane & 15);
            double old[4];
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {           // the tile's current values: in flight while the MFMAs run
                const int row = nbw + 16 * ti + (lane >> 4) + 4 * rg;
                old[rg] = (row < m && col < m && col <= row) ? AB(j0 + row, j0 + col) : 0.0;
            }
            ba_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int ks = 0; ks < NB / 4; ks++) {
                const double av = ra < m ? P[ra * PP + 4 * ks + kq] : 0.0;
                const double bv = rb < m ? P[rb * PP + 4 * ks + kq] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int row = nbw + 16 * ti + (lane >> 4) + 4 * rg;
                if (row < m && col < m && col <= row) AB(j0 + row, j0 + col) = old[rg] - acc[rg];
            }
        }
        __syncthreads();
    }
    if (s_fail) { if (tid == 0) St->chol_ok = 0; for (int i = tid; i < n; i += 1024) x[i] = 0; return; }
    // backward substitution L' x = y, panels in reverse
    const int last = ((n - 1) / NB) * NB;
    for (int j0 = last; j0 >= 0; j0 -= NB) {
        const int nbw = min(NB, n - j0), m = min(n - j0, NB + bw);
        for (int idx = tid; idx < m * nbw; idx += 1024) {
            const int r = idx / nbw, c = idx - r * nbw;
            P[r * PP + c] = (c <= r && c >= r - bw) ? AB(j0 + r, j0 + c) : 0.0;
        }
        __syncthreads();
        {   // part[p][c] = sum over rows r = nbw + p, nbw + p + NP, ... of L[r][c] * x[r]
            const int c = tid & (NB - 1), p = tid / NB;             // 1024 / NB parts
            constexpr int NP = 1024 / NB;
            double s2 = 0;
            if (c < nbw) for (int r = nbw + p; r < m; r += NP) s2 += P[r * PP + c] * x[j0 + r];
            part[p * NB + c] = s2;
            __syncthreads();
            if (tid < nbw) { double tot = 0; for (int q = 0; q < NP; q++) tot += part[q * NB + tid]; x[j0 + tid] -= tot; }
            __syncthreads();
        }
        if (wv == 0) {                                // backward substitution inside the panel
            double yl = lane < nbw ? x[j0 + lane] : 0.0;
            for (int c = nbw - 1; c >= 0; c--) {
                const double xc = d_shfl_idx(yl, c) / P[c * PP + c];
                if (lane == c) yl = xc;
                else if (lane < c) yl -= P[c * PP + lane] * xc;
            }
            if (lane < nbw) x[j0 + lane] = yl;
        }
        __syncthreads();
    }
}
#undef AB


// ---- k_ba_chol_dense: dense SPD solve S x = b with the WHOLE factor on chip (local-BA windows: n <= 288, the case bw = n - 1;
// also the 256 x 256 separator system of the two-way window solve).  The panel kernel above streams the trailing matrix through
// L2 once per panel and spends most of a panel in one wave's triangular solves; here the strictly lower tiles (16x16, MFMA
// accumulator layout) live in the registers of one 1024-thread workgroup, the diagonal tiles in LDS, and every O(n^3) and O(n^2)
// step is a matrix-core instruction.  Wave 0 is the panel wave (no tiles: its registers hold the 16x16 factor and its inverse);
// waves 1..15 own the tiles, at most 11 each: the tiles of the column-major enumeration of the strict lower triangle are dealt to the waves once
// (round-robin, the three waves on the panel wave's SIMD fewer: see the tile-id set-up below) and never move; the diagonal tiles are updated in LDS
// by the tile waves that own them.
//   per block column J:  [2] wave 0 factors diagonal tile J (row per lane, v_readlane broadcasts), inverts the 16x16 triangle
//        (column per lane) and finishes y_J = Linv y_J;  [3] the tile waves turn their tiles of column J into
//        L(i,J) = T(i,J) Linv' with 4 MFMAs (operands via LDS), publish them and take L(i,J) y_J out of y_i;  [4] they update
//        their tiles T(i,j) -= L(i,J) L(j,J)' with 4 MFMAs each, operands from the published column.  Three barriers per column.
//   backward substitution: the factor is still in the registers; per block column the owners form L(i,J)' x_i (lane sums), wave 0
//        adds the partials in row order and applies Linv'.  Fixed summation order throughout (bitwise reproducible).
// Lanes of one wave exchanging data through LDS: the hardware executes a wave's LDS operations in order, but the compiler reasons
// per thread and may move a lane's read above another lane's (non-aliasing, for that thread) write: this keeps them apart.
#define CD_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define CD_MAXT 18                                   // tile rows: n <= 288 (153 strict-lower tiles on 15 waves x 11)
#define CD_NTW 11
#define CD_P 17                                      // LDS tile pitch in doubles
#define CD_TILE (16 * CD_P)
#define CD_LDS_DOUBLES (3 * CD_MAXT * CD_TILE /*Lc, Linv, diagonal tiles*/ + 3 * CD_MAXT * 16 /*y, x, partials*/ + 16)

// BCR = true: the batched factor-only form used by the block cyclic reduction below - workgroup b factors the 256 x 256 block
// el[b].e of Dblk (row-major, lower part) in place (strictly lower tiles -> L, diagonal tiles -> the INVERSES of their factors) and
// leaves y = L^-1 g in yv; no backward pass.
struct bcr_elim { int e, l, r, sl, sr; };            // block to eliminate, its chain neighbours (-1: none), the coupling slots (e x l), (e x r)
template <bool BCR>
__device__ __forceinline__ void ba_chol_dense_body(const double* __restrict__ A_, const double* __restrict__ bs_, int n, int bw,
                                                        double* __restrict__ x, ba_state* __restrict__ St, const bcr_elim* __restrict__ el,
                                                        double* Dblk, const double* gv, double* yv)
{
    const double* A = A_; const double* bs = bs_;
    double* Aout = nullptr; double* yout = nullptr;
    if constexpr (BCR) {
        const int e = el[blockIdx.x].e;
        Aout = Dblk + (size_t)e * 65536; A = Aout; bs = gv + 256 * (size_t)e; yout = yv + 256 * (size_t)e;
    }
#define CD_A(r, c) (BCR ? A[(size_t)(r) * 256 + (c)] : A[(size_t)(r) * LD + ((c) - (r) + bw)])
    extern __shared__ double sm[];
    double* Lc = sm;                                  // [NT][16][17] the factored block column J (operands of [4])
    double* Li = Lc + CD_MAXT * CD_TILE;              // [NT][16][17] inverses of the diagonal tiles' factors
    double* Dg = Li + CD_MAXT * CD_TILE;              // [NT][16][17] diagonal tiles (lower part; updated in place until factored)
    double* sy = Dg + CD_MAXT * CD_TILE;              // [NT*16] right-hand side -> y
    double* sx = sy + CD_MAXT * 16;                   // [NT*16] solution
    double* spart = sx + CD_MAXT * 16;                // [NT][16] partial L' x of the backward pass
    double* sz = spart + CD_MAXT * 16;                // [16]
    __shared__ int s_fail, s_flag;                    // s_flag: the highest diagonal tile that has received its last update
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NT = (n + 15) >> 4, T = NT * (NT - 1) / 2, LD = bw + 1;
#ifdef CD_CLOCKS                                     // phase clocks of the panel wave (tools/chol_clocks.py; build ba.hip with -DCD_CLOCKS)
    const long long tk0 = wall_clock64();
    long long tkA = 0, tkB = 0, tkW = 0, tkF = 0, tkF1 = 0, tkF2 = 0;
#endif
    if (tid == 0) { s_fail = St->chol_ok ? 0 : 1; s_flag = 0; }
    for (int i = tid; i < NT * 16; i += 1024) sy[i] = i < n ? bs[i] : 0.0;
    for (int i = tid; i < NT * CD_TILE; i += 1024) Li[i] = 0.0;      // the blocked inverse writes the lower blocks only
    for (int e = tid; e < NT * 256; e += 1024) {      // diagonal tiles: lower part, identity padding
        const int j = e >> 8, rr = (e >> 4) & 15, cc = e & 15, r = 16 * j + rr, c = 16 * j + cc;
        double v = 0.0;
        if (r < n && c < n) { if (c <= r && r - c <= bw) v = CD_A(r, c); }
        else if (r == c) v = 1.0;
        Dg[j * CD_TILE + rr * CD_P + cc] = v;
    }
    __syncthreads();
    if (wv == 0) {
#ifdef CD_CLOCKS
        tkA = wall_clock64();
#endif
        // ================================================================ panel wave
        const int rl = lane & 15;                     // lanes 16..63 mirror lanes 0..15 (duplicates keep every v_readlane source defined)
        for (int J = 0; J < NT; J++) {
            // look-ahead: diagonal tile J received its last update from the wave that owns it at the START of that wave's step [4] of
            // column J - 1; the panel wave factors it while the tile waves are still updating their tiles of the trailing matrix
#ifdef CD_CLOCKS
            const long long w0 = wall_clock64();
#endif
            if (J > 0) while (__atomic_load_n(&s_flag, __ATOMIC_ACQUIRE) < J) __builtin_amdgcn_s_sleep(1);
#ifdef CD_CLOCKS
            const long long f0 = wall_clock64(); tkW += f0 - w0;
#endif
            double row[16];
#pragma unroll
            for (int k = 0; k < 16; k++) row[k] = Dg[J * CD_TILE + rl * CD_P + k];
            bool bad = false;
            double myinv = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const double dg = d_readlane(row[c], c);
                bad = bad | (!(dg > 0) | !(dg < 1.0e300));
                const double y0 = __builtin_amdgcn_rsq(dg);
                double g = dg * y0, h = 0.5 * y0;
                double rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
                rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
                const double id = h + h;
                myinv = rl == c ? id : myinv;
                // (no masks: the entries above the diagonal - lanes rl < c / rl < c2 - take meaningless finite values that nothing reads: the pivots,
                // the column broadcasts and the inverse below only use the lower triangle.  The 64-bit selects were a quarter of this loop's
                // instructions, and the loop is bound by how fast one wave issues: 2.5 -> 1.0 us per 16 x 16 tile.)
                const double lc = rl == c ? g : row[c] * id;
                row[c] = lc;
#pragma unroll
                for (int c2 = c + 1; c2 < 16; c2++) {
                    const double l2 = d_readlane(lc, c2);
                    row[c2] = __builtin_fma(-lc, l2, row[c2]);
                }
            }
#ifdef CD_CLOCKS
            const long long f1 = wall_clock64(); tkF1 += f1 - f0;
#endif
            // Inverse of the lower-triangular factor, on all 64 lanes in 4 x 4 blocks (round 4; column rl per lane by forward substitution was
            // 120 dependent fused multiply-adds fed by 240 v_readlane on 16 lanes: half of the panel wave's time per tile).  X[p][c] =
            // -(1 / L[p][p]) sum_{q < p} L[p][q] X[q][c], X[c][c] = 1 / L[c][c].  L and the reciprocal diagonal go to LDS (the diagonal tile's own
            // slot: the tile waves read Linv only).  Step d = 0..3 makes the blocks (g + d, g), g = lane >> 4: lane = (column c of the block,
            // row r of the block) with the four rows of a column in one quad - the part of the sum over rows ABOVE the block comes from LDS
            // (results of the earlier steps), the part inside the block runs down the quad through DPP quad broadcasts.
            double* Lt = Dg + J * CD_TILE; double* Xt = Li + J * CD_TILE;
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) Lt[rl * CD_P + k] = row[k];
                sz[rl] = myinv;
            }
            CD_WAVE_SYNC();
            {
                const int g4 = lane >> 4, cq = (lane >> 2) & 3, rq = lane & 3;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int bi = g4 + d, bj = g4;
                    const bool act = bi < 4;
                    const int pp = act ? 4 * bi + rq : 0, colx = 4 * bj + cq;
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 4 * d; q++) acc = __builtin_fma(Lt[pp * CD_P + 4 * bj + q], Xt[(4 * bj + q) * CD_P + colx], acc);
                    const double ip = sz[pp];
                    double Lm[3];
#pragma unroll
                    for (int m = 0; m < 3; m++) Lm[m] = Lt[pp * CD_P + 4 * (act ? bi : 0) + m];
                    double xv = 0.0;
#define CD_INV_STEP(m)                                                                                                     \
                    {                                                                                                          \
                        const double cand = d == 0 ? (rq == cq ? ip : (rq > cq ? -ip * acc : 0.0)) : -ip * acc;                \
                        const double xm = d_quad_bcast<m>(cand);                                                               \
                        xv = rq == m ? cand : xv;                                                                              \
                        if (m < 3) acc = rq > m ? __builtin_fma(Lm[m < 3 ? m : 0], xm, acc) : acc;                             \
                    }
                    CD_INV_STEP(0) CD_INV_STEP(1) CD_INV_STEP(2) CD_INV_STEP(3)
#undef CD_INV_STEP
                    if (act) Xt[pp * CD_P + colx] = xv;
                    CD_WAVE_SYNC();
                }
            }
#ifdef CD_CLOCKS
            tkF2 += wall_clock64() - f1;
#endif
            if (bad && lane == 0) s_fail = 1;
            CD_WAVE_SYNC();
            // y_J = Linv y_J (row rl of Linv through LDS)
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) acc = __builtin_fma(Li[J * CD_TILE + rl * CD_P + c], sy[16 * J + c], acc);
            CD_WAVE_SYNC();                           // every lane has read all 16 old values
            if (lane < 16) sy[16 * J + rl] = acc;
#ifdef CD_CLOCKS
            tkF += wall_clock64() - f0;
#endif
            __syncthreads();                          // B2: Linv_J and y_J are ready
            if (s_fail) break;
            __syncthreads();                          // B3: column J published, y updated
        }
#ifdef CD_CLOCKS
        tkB = wall_clock64();
#endif
        if (!BCR && !s_fail)
            for (int J = NT - 1; J >= 0; J--) {
                __syncthreads();                      // the partials of column J are in spart
                double z = sy[16 * J + rl];
                for (int i = J + 1; i < NT; i++) z -= spart[i * 16 + rl];
                if (lane < 16) sz[rl] = z;
                CD_WAVE_SYNC();
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 16; r++) acc = __builtin_fma(Li[J * CD_TILE + r * CD_P + rl], sz[r], acc);      // (Linv')[rl][r] = Linv[r][rl]
                if (lane < 16) sx[16 * J + rl] = acc;
                __syncthreads();                      // x_J is ready
            }
    } else {
        // ================================================================ tile waves
        const int colq = lane & 15, r0q = lane >> 4;
        uint32_t tq[6] = { 0, 0, 0, 0, 0, 0 };        // (i | j << 8) of my tiles, two per register; 0xffff = no tile
#define CD_TIJ(k) ((int)((tq[(k) >> 1] >> (16 * ((k) & 1))) & 0xffffu))
        ba_d4 Tl[CD_NTW];
#pragma unroll
        for (int k = 0; k < 12; k++) {
#ifndef CD_EVEN_TILES
            // Round 5: waves 4, 8, 12 share SIMD 0 with the panel wave (waves go round the four SIMDs), and the panel wave's f64 multiply-add chains - the factorisation
            // of a diagonal tile - wait behind their f64 MFMAs, which hold the SIMD's f64 pipe for 32 cycles each: they take 7 tiles, the other twelve 11 (rounds 0 .. 6
            // deal a tile to all fifteen tile waves, rounds 7 .. 10 to the twelve others).  The panel wave's factor loop: 1.63 -> 0.94 us per tile, the solve 127 -> 122.6 us
            // (tools/chol_clocks.py); the same tiles with the same arithmetic, so the same bits.  -DCD_EVEN_TILES: tile t to wave 1 + t mod 15 as before.
            const bool simd0 = (wv & 3) == 0;
            const int t = k < 7 ? (wv - 1) + 15 * k : 105 + 12 * (k - 7) + (wv - 1 - (wv >> 2));
            int j = 0, rem = t;
            const bool have = k < CD_NTW && t < T && !(simd0 && k >= 7);
#else
            const int t = (wv - 1) + 15 * k;
            int j = 0, rem = t;
            const bool have = k < CD_NTW && t < T;
#endif
            if (have) { while (rem >= NT - 1 - j) { rem -= NT - 1 - j; j++; } }
            const int i = j + 1 + rem;
            tq[k >> 1] |= (uint32_t)(have ? (i | (j << 8)) : 0xffff) << (16 * (k & 1));
            if (k < CD_NTW) {
                ba_d4 v = { 0.0, 0.0, 0.0, 0.0 };
                if (have) {
                    const int c = 16 * j + colq;
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int r = 16 * i + r0q + 4 * rg;
                        v[rg] = (r < n && c < n && r - c <= bw) ? CD_A(r, c) : 0.0;      // strictly below the diagonal tiles: r > c
                    }
                }
                Tl[k] = v;
            }
        }
        bool failed = false;
        for (int J = 0; J < NT; J++) {
            // per-lane LDS addresses are recomputed inside the step: left alone, LICM hoists the loop-invariant tile addresses out of
            // the J loop and the tiles spill (same cure as in k_ba_chol_sw)
            int col = colq, r0 = r0q;
            asm volatile("" : "+v"(col), "+v"(r0), "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]));
            __syncthreads();                          // B2
            if (s_fail) { failed = true; break; }
            // [3] column J: L(i,J) = T(i,J) Linv'.  The raw tile goes to its Lc slot, is read back as the A operand, and the slot is
            // then overwritten with the result (same wave, LDS operations in order).
            const double yj = sy[16 * J + col];
#pragma unroll
            for (int k = 0; k < CD_NTW; k++) {
                const int id = CD_TIJ(k);
                if (id == 0xffff || (id >> 8) != J) continue;
                const int i = id & 255;
                double* slot = Lc + i * CD_TILE;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) slot[(r0 + 4 * rg) * CD_P + col] = Tl[k][rg];
                CD_WAVE_SYNC();
                ba_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(slot[col * CD_P + 4 * ks + r0], Li[J * CD_TILE + col * CD_P + 4 * ks + r0], acc, 0, 0, 0);
                Tl[k] = acc;
                if constexpr (BCR) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) Aout[(size_t)(16 * i + r0 + 4 * rg) * 256 + 16 * J + col] = acc[rg];
                }
                CD_WAVE_SYNC();                       // the operand reads above precede the overwrite below in every lane
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    slot[(r0 + 4 * rg) * CD_P + col] = acc[rg];
                    // y_i -= L(i,J) y_J : the sum over the 16 columns is a sum over the 16 lanes of a DPP row
                    double v = acc[rg] * yj;
                    v += d_row_shr<1>(v); v += d_row_shr<2>(v); v += d_row_shr<4>(v); v += d_row_shr<8>(v);      // lane 15 of the row holds the sum
                    if (col == 15) sy[16 * i + r0 + 4 * rg] -= v;
                }
            }
            __syncthreads();                          // B3
            // [4] first the diagonal tiles this wave owns (tile wv, and tile wv + 15 when n > 256), in LDS: D_j -= L(j,J) L(j,J)' - the
            // next one to be factored before anything else, so that the panel wave can start on it
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int jd = wv + 15 * q;                   // (left transparent: made opaque, or scalar, the tiles spill)
                if (jd > J && jd < NT) {
                    ba_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        const double o = Lc[jd * CD_TILE + col * CD_P + 4 * ks + r0];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o, o, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) Dg[jd * CD_TILE + (r0 + 4 * rg) * CD_P + col] -= acc[rg];
                    if (jd == J + 1) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __atomic_store_n(&s_flag, J + 1, __ATOMIC_RELEASE);
                    }
                }
            }
            // then the trailing update T(i,j) -= L(i,J) L(j,J)' of my register tiles
#pragma unroll
            for (int k = 0; k < CD_NTW; k++) {
                const int id = CD_TIJ(k);
                if (id == 0xffff || (id >> 8) <= J) continue;
                const int i = id & 255, j = id >> 8;
                ba_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lc[i * CD_TILE + col * CD_P + 4 * ks + r0], Lc[j * CD_TILE + col * CD_P + 4 * ks + r0], acc, 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; rg++) Tl[k][rg] -= acc[rg];
            }
        }
        if (!BCR && !failed)
            for (int J = NT - 1; J >= 0; J--) {
                int col = colq, r0 = r0q;
                asm volatile("" : "+v"(col), "+v"(r0), "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]));
#pragma unroll
                for (int k = 0; k < CD_NTW; k++) {
                    const int id = CD_TIJ(k);
                    if (id == 0xffff || (id >> 8) != J) continue;
                    const int i = id & 255;
                    double v = 0.0;
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) v = __builtin_fma(Tl[k][rg], sx[16 * i + r0 + 4 * rg], v);      // column `col`, my 4 rows
                    v += d_xor_swap<16>(v); v += d_xor_swap<32>(v);                                               // the other 12 rows
                    if (lane < 16) spart[i * 16 + col] = v;
                }
                __syncthreads();
                __syncthreads();
            }
#undef CD_TIJ
    }
    if constexpr (BCR) {
        __syncthreads();
        if (s_fail) { if (tid == 0) St->chol_ok = 0; return; }
        for (int e = tid; e < NT * 256; e += 1024) {
            const int j = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            Aout[(size_t)(16 * j + rr) * 256 + 16 * j + cc] = Li[j * CD_TILE + rr * CD_P + cc];
        }
        for (int i = tid; i < 256; i += 1024) yout[i] = sy[i];
        return;
    }
    if (s_fail) { if (tid == 0) St->chol_ok = 0; for (int i = tid; i < n; i += 1024) x[i] = 0; return; }
    for (int i = tid; i < n; i += 1024) x[i] = sx[i];
#ifdef CD_CLOCKS
    if (tid == 0) { St->dbg[0] = tkA - tk0; St->dbg[1] = tkB - tkA; St->dbg[2] = wall_clock64() - tkB; St->dbg[3] = tkW; St->dbg[4] = tkF; St->dbg[5] = tkF1; St->dbg[6] = tkF2; }
#endif
#undef CD_A
}
template <bool BCR>
__global__ __launch_bounds__(1024) void k_ba_chol_dense(const double* __restrict__ A_, const double* __restrict__ bs_, int n, int bw,
                                                        double* __restrict__ x, ba_state* __restrict__ St, const bcr_elim* __restrict__ el,
                                                        double* Dblk, const double* gv, double* yv)
{
    ba_chol_dense_body<BCR>(A_, bs_, n, bw, x, St, el, Dblk, gv, yv);
}

// ---- k_ba_chol_sw: banded Cholesky with the whole trailing window ON CHIP (global-BA sizes: n >> bw, bw <= 256).
// The panel kernel above streams the bw x bw trailing window through global memory once per panel; here it never leaves
// the CU.  16x16 tiles; block column J is owned by wave J mod 16.  Tile (J+d, J), d = 0..D (D = ceil-band in blocks <= 16):
//   d < 8   lives in the owner's registers (MFMA accumulator layout) from the moment the wave adopts the column,
//   d >= 8  lives in an LDS ring slot, allocated when the first update reaches it (step J+d-D) - 45 slots in all.
// Step Jc: [A] the owner dumps its column into the LDS operand buffer Lc, factors the diagonal tile in registers (row per
// lane, v_readlane broadcasts) and forward-substitutes the rhs block; [B] one thread per row below does the triangular
// solve against the diagonal tile, the rhs update and the store of the final L row; [C] every wave updates the tiles of
// its own column on the matrix cores: T(J+d, J) -= L(J+d, Jc) L(J, Jc)' (v_mfma_f64_16x16x4_f64, operands from Lc), while
// the former owner adopts column Jc+16.  The backward substitution re-reads L column by column.  Fixed summation order.
#define SW_P 17
template <int C> __device__ __forceinline__ double d_row_bcast(double v)      // lane (lane & ~15) + C of every 16-lane row
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + C, 0xf, 0xf, false);    // row_newbcast:C
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <typename F, int... Is> __device__ __forceinline__ void sw_static_for(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
// One workgroup's logical matrix: lower element (r, c), r >= c, of a band matrix padded to npad = 16*ceil(n/16) rows.  rev = 0: the
// stored matrix itself.  rev = 1: the same matrix with rows and columns in reverse order (r -> npad-1-r), still banded and SPD;
// its lower element (r, c) is the stored lower element (npad-1-c, npad-1-r).  Rows >= nl (the leading principal submatrix this
// workgroup works on) and rows that map beyond n behave as identity padding.
struct sw_view {
    long long base, sr, sc;                  // band-array index of (r, c) = base + r*sr + c*sc
    int nl, npad, n, bw, rev;
    __device__ __forceinline__ long long idx(int r, int c) const { return base + (long long)r * sr + (long long)c * sc; }
    __device__ __forceinline__ int vec(int r) const { return rev ? npad - 1 - r : r; }            // index into rhs / solution
    __device__ __forceinline__ bool vec_ok(int r) const { return r < nl && vec(r) < n; }
    __device__ __forceinline__ bool stored(int r, int c) const                                    // r >= c: inside matrix and band
    {
        return r < nl && r - c <= bw && (rev ? npad - 1 - c : r) < n;
    }
    __device__ __forceinline__ bool padding(int r, int c) const { return r >= nl || (rev ? npad - 1 - c : r) >= n; }
};
__device__ __forceinline__ sw_view sw_make_view(int n, int bw, int nl, int rev)
{
    sw_view v;
    v.n = n; v.bw = bw; v.nl = nl; v.rev = rev; v.npad = ((n + 15) >> 4) << 4;
    if (!rev) { v.base = bw; v.sr = bw; v.sc = 1; }                      // r*(bw+1) + c - r + bw
    else { v.base = (long long)(v.npad - 1) * (bw + 1) + bw; v.sr = -1; v.sc = -bw; }      // (npad-1-c)*(bw+1) + c - r + bw
    return v;
}
__device__ __forceinline__ ba_d4 sw_load_tile(const double* __restrict__ A, const sw_view& V, int Ri, int Rj, int lane)
{
    ba_d4 v;
    const int c = 16 * Rj + (lane & 15), r0 = 16 * Ri + (lane >> 4);
    // interior tile (wave-uniform test): strictly below the diagonal tile, inside the matrix and the band -> four loads a
    // constant stride apart
    if (Ri > Rj && 16 * Ri + 16 <= V.nl && 16 * (Ri - Rj) + 15 <= V.bw && (V.rev ? 16 * Rj >= V.npad - V.n : 16 * Ri + 16 <= V.n)) {
        const double* p = A + V.idx(r0, c);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) v[rg] = p[4 * rg * V.sr];
        return v;
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        const int r = r0 + 4 * rg;
        const int rr = r > c ? r : c, cc = r > c ? c : r;  // the diagonal tile is delivered symmetric (both triangles)
        double xv = 0.0;
        if (!V.padding(rr, cc)) { if (rr - cc <= V.bw) xv = A[V.idx(rr, cc)]; }
        else if (r == c) xv = 1.0;                         // padding rows/columns: identity
        v[rg] = xv;
    }
    return v;
}
__device__ __forceinline__ int sw_slot(int d, int J)      // ring slot of LDS tile (J+d, J), 8 <= d <= 16
{
    const int base = (d - 8) * 9 - (d - 8) * (d - 9) / 2;  // sum_{e=8}^{d-1} (17 - e)
    return base + J % (17 - d);
}

#define SW_NT 512                                     // 8 waves: 256 VGPRs each, two block columns (16 register tiles) per wave
#define SW_LDS_DOUBLES (272 * SW_P + 45 * 256 + 288 + 16 + 16 + 512 + 512 + 7 * 256)
// Two-way solve (split > 0, D == 16, three launches on one stream instead of one):
//   phase 1, grid 2: workgroup 0 eliminates block columns 0 .. split-1 of the matrix top-down, workgroup 1 the first
//            NB2 = NBLK-split-16 block columns of the REVERSED matrix (the stored matrix bottom-up).  The 16 block columns in
//            between (>= the bandwidth) separate the two eliminations completely; each side stops with its trailing window =
//            A22 minus its own Schur update, and writes window + rhs rows to xch;
//   k_sw_separator + k_ba_chol_dense: the 256 x 256 separator system  W_top + W_bottom - A22  is assembled and solved;
//   phase 2, grid 2: both sides back-substitute their own part with the separator's solution in place.
// No extra flops, two CUs instead of one.  phase 0 = everything in one workgroup.
#define SW_XCH_TILE(side, q, d) (65536 * (side) + 256 * (16 * (q) + (d)))       // window tile (column q, distance d), row-major 16x16
#define SW_XCH_RHS(side) (2 * 65536 + 256 * (side))
#define SW_XCH_DOUBLES (2 * 65536 + 512)
// REV is a template parameter: with the ordering known at compile time the tile addresses fold into immediates; as a run-time
// flag the generic strides cost ~100 address registers and spill the window tiles.
template <int REV>
__device__ __forceinline__ void sw_body(double* __restrict__ sm, double* __restrict__ A, const double* __restrict__ bs, int n, int bw,
                                        double* __restrict__ x, ba_state* __restrict__ St, int phase, int split, int side,
                                        double* __restrict__ xch)
{
    const int D = (bw + 15) >> 4, NB_ALL = (n + 15) >> 4;
    const int NFACT = !phase ? NB_ALL : split;        // block columns this workgroup eliminates (`split` = nfact of this side)
    const int NBLK = !phase ? NB_ALL : NFACT + 16;    // ... out of a leading submatrix of this many
    const sw_view V = sw_make_view(n, bw, 16 * NBLK, REV);
    double* Lc = sm;                                  // [(D+1)*16][SW_P] operand buffer: the factored block column
    double* slots = Lc + 272 * SW_P;                  // 45 tiles x 256 (accumulator layout [reg][lane])
    double* xr = slots + 45 * 256;                    // rhs ring: global row mod 288
    double* invd = xr + 288;                          // 1 / L[c][c] of the current diagonal tile
    double* ybuf = invd + 16;                         // forward-substituted rhs of the current block
    double* part = ybuf + 16;                         // [32][16] partial dot products of the backward pass
    double* lastb = part + 512;                       // [2][256] outermost tile (d = D) of the next two columns to be factored
    double* rdump = lastb + 512;                      // [7][256] the owner's register tiles d = 1..7, accumulator layout
    __shared__ int s_fail;
    __shared__ long long dbgl[4];
    const int tid = threadIdx.x, wv = tid >> 6;
    // lane_v is made opaque once per step (asm below): every per-lane LDS address is then recomputed where it is used instead
    // of being hoisted out of the step loop as ~100 loop-invariant VGPRs, which is what pushed the tiles into scratch
    int lane_v = tid & 63;
#define SW_LANE const int lane = lane_v, col = lane_v & 15, r0 = lane_v >> 4; (void)lane; (void)col; (void)r0
    if (tid == 0) { s_fail = St->chol_ok ? 0 : 1; dbgl[0] = dbgl[1] = dbgl[2] = 0; }
    // block column J lives in wave J & 7, register set (J >> 3) & 1
    ba_d4 T0[8], T1[8];
    int J0 = wv, J1 = wv + 8;

    auto load_col = [&](ba_d4 (&T)[8], int J) {
        SW_LANE;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            if (J < NBLK && d <= D) T[d] = sw_load_tile(A, V, J + d, J, lane);
            else { T[d][0] = 0; T[d][1] = 0; T[d][2] = 0; T[d][3] = 0; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto init_slots = [&](int J) {                    // LDS tiles that are live before the loop's first-touch rule applies
        SW_LANE;
#pragma unroll
        for (int d = 8; d <= 16; d++)
            if (d < D && J < NBLK && J + d - D < 0) {
                const ba_d4 v = sw_load_tile(A, V, J + d, J, lane);
                double* sl = slots + 256 * sw_slot(d, J);
#pragma unroll
                for (int rg = 0; rg < 4; rg++) sl[64 * rg + lane] = v[rg];
            }
    };
    if (phase != 2) { load_col(T0, J0); load_col(T1, J1); init_slots(J0); init_slots(J1); }
    if (phase != 2 && wv == 0 && D >= 8) {
        SW_LANE;
        const ba_d4 v = sw_load_tile(A, V, D, 0, lane);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) lastb[64 * rg + lane] = v[rg];
    }
    for (int i = tid; i < 16 * (D + 1); i += SW_NT) xr[i] = V.vec_ok(i) ? bs[V.vec(i)] : 0.0;
    __syncthreads();

    // while the owner factors: bring in the tiles of column J (q = J - Jc >= 1) that the window reaches next
    auto prefetch = [&](int J, int q) {
        if (J >= NBLK) return;
        SW_LANE;
        const int dnew = D - q;                       // the tile of the column that this step's update touches first
        if (dnew >= 8 && dnew < D) {
            const ba_d4 v = sw_load_tile(A, V, J + dnew, J, lane);
            double* sl = slots + 256 * sw_slot(dnew, J);
#pragma unroll
            for (int rg = 0; rg < 4; rg++) sl[64 * rg + lane] = v[rg];
        }
        if (q == 1 && D >= 8) {                       // next step's owner: its outermost tile (never updated) for the dump
            const ba_d4 v = sw_load_tile(A, V, J + D, J, lane);
#pragma unroll
            for (int rg = 0; rg < 4; rg++) lastb[256 * (J & 1) + 64 * rg + lane] = v[rg];
        }
    };

    // ---- [A] owner: factor the diagonal tile in place, dump the column into Lc, adopt column Jc+16, forward-substitute
    auto phase_a = [&](ba_d4 (&T)[8], int Jc) {
        SW_LANE;
        long long a0 = wall_clock64();
        bool bad = false;
        double myinv = 0.0;
        // right-looking on the symmetric tile (accumulator layout: lane holds column lane&15, rows (lane>>4)+4*rg): column C by
        // row comes from a DPP row broadcast, row C by column from one ds_bpermute; both are issued before the pivot's rsqrt
        // chain completes
        sw_static_for([&](auto cc) {
            constexpr int Cc = decltype(cc)::value;
            const double dg = d_readlane(T[0][Cc >> 2], 16 * (Cc & 3) + Cc);
            const double rowc = d_shfl_idx(T[0][Cc >> 2], 16 * (Cc & 3) + col);
            double bc[4];
#pragma unroll
            for (int rg = 0; rg < 4; rg++) bc[rg] = d_row_bcast<Cc>(T[0][rg]);
            bad = bad | !(dg > 0) | !(dg < 1.0e300);
            // sqrt and 1/sqrt together (Goldschmidt from the hardware seed): g -> sqrt(dg), h -> 1/(2 sqrt(dg)); 5 dependent ops
            const double y0 = __builtin_amdgcn_rsq(dg);
            double g = dg * y0, h = 0.5 * y0;
            double rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
            rr = __builtin_fma(-g, h, 0.5); g = __builtin_fma(g, rr, g); h = __builtin_fma(h, rr, h);
            const double sd = g, id = h + h;
            myinv = lane == Cc ? id : myinv;
            // branch-free: masked operands instead of divergent control flow (the chain is latency-bound)
            const double lcolm = col > Cc ? rowc * id : 0.0;
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int r = r0 + 4 * rg;
                const double lr = bc[rg] * id;
                const double upd = __builtin_fma(-(r > Cc ? lr : 0.0), lcolm, T[0][rg]);
                const double fin = r == Cc ? sd : (r > Cc ? lr : T[0][rg]);
                T[0][rg] = col == Cc ? fin : upd;
            }
        }, std::make_integer_sequence<int, 16>{});
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {                            // final L of the diagonal tile to global
            const int r = r0 + 4 * rg, grow = 16 * Jc + r;
            if (col <= r && V.stored(grow, 16 * Jc + col)) A[V.idx(grow, 16 * Jc + col)] = T[0][rg];
        }
        { const long long a1 = wall_clock64(); if (lane == 0) dbgl[0] += a1 - a0; a0 = a1; }
        // share the column: the diagonal tile (lower triangle) goes to Lc, the register tiles below it to rdump; the LDS tiles
        // (slots, lastb) are read in place by [B]
#pragma unroll
        for (int rg = 0; rg < 4; rg++) { const int r = r0 + 4 * rg; Lc[r * SW_P + col] = col > r ? 0.0 : T[0][rg]; }
#pragma unroll
        for (int d = 1; d < 8; d++)
            if (d <= D) {
#pragma unroll
                for (int rg = 0; rg < 4; rg++) rdump[256 * (d - 1) + 64 * rg + lane] = T[d][rg];
            }
        load_col(T, Jc + 16);                          // adopt the next column: its register tiles start from S
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        { const long long a1 = wall_clock64(); if (lane == 0) dbgl[1] += a1 - a0; a0 = a1; }
        // forward substitution of the rhs block: lane l < 16 owns y[16 Jc + l]
        double yl = xr[(16 * Jc + col) % 288];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const double yc = d_readlane(yl, c) * d_readlane(myinv, c);
            yl = col == c ? yc : __builtin_fma(-(col > c ? Lc[col * SW_P + c] : 0.0), yc, yl);
            if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 16) {
            invd[lane] = myinv; ybuf[lane] = yl; xr[(16 * Jc + lane) % 288] = yl;
            if (V.vec_ok(16 * Jc + lane)) x[V.vec(16 * Jc + lane)] = yl;
        }
        if (bad && lane == 0) s_fail = 1;
        { const long long a1 = wall_clock64(); if (lane == 0) dbgl[2] += a1 - a0; }
    };

    // ---- [C] trailing update of column Jm (q blocks ahead of the factored one) on the matrix cores
    auto phase_c = [&](ba_d4 (&T)[8], int Jm, int qq) {
        if (Jm >= NBLK || qq > D) return;
        SW_LANE;
        double bv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bv[ks] = Lc[(16 * qq + col) * SW_P + 4 * ks + r0];
#pragma unroll
        for (int d = 0; d < 8; d++)
            if (qq + d <= D) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    T[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lc[(16 * (qq + d) + col) * SW_P + 4 * ks + r0], bv[ks], T[d], 0, 0, 0);
                if (d & 1) __builtin_amdgcn_sched_barrier(0);     // two tiles' operands in flight at most: the scheduler otherwise
            }                                                     // hoists every LDS read of the phase and spills
#pragma unroll
        for (int d = 8; d <= 16; d++)
            if (qq + d <= D && d < D) {
                double* sl = slots + 256 * sw_slot(d, Jm);
                ba_d4 acc;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) acc[rg] = sl[64 * rg + lane];
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lc[(16 * (qq + d) + col) * SW_P + 4 * ks + r0], bv[ks], acc, 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; rg++) sl[64 * rg + lane] = acc[rg];
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    long long tA = 0, tB = 0, tC = 0, tBack = 0, t0 = 0;
    for (int Jc = 0; Jc < (phase == 2 ? 0 : NFACT); Jc++) {
        const int wo = Jc & 7, so = (Jc >> 3) & 1;
        asm volatile("" : "+v"(lane_v));
        SW_LANE;
        if (tid == 0) t0 = wall_clock64();
        if (wv == ((wo + 2) & 7) && Jc >= 1 && lane < 16) {    // rhs block entering the window (kept off the owner's critical path)
            const int row = 16 * (Jc + D) + lane; xr[row % 288] = V.vec_ok(row) ? bs[V.vec(row)] : 0.0;
        }
        if (!(wv == wo && so == 0)) prefetch(J0, J0 - Jc);
        if (!(wv == wo && so == 1)) prefetch(J1, J1 - Jc);
        if (wv == wo) { if (so == 0) phase_a(T0, Jc); else phase_a(T1, Jc); }
        __syncthreads();
        if (tid == 0) { const long long t1 = wall_clock64(); tA += t1 - t0; t0 = t1; }
        if (s_fail) break;
        // ---- [B] tiles below the diagonal: L21 = A21 L11^-T, tile by tile in the accumulator layout (column C scaled, broadcast
        // along the rows by DPP, eliminated from the columns to its right), rhs update, final L to global
        for (int d = wv + 1; d <= D; d += 8) {
            ba_d4 t;
            const double* src = d < 8 ? rdump + 256 * (d - 1) : (d == D ? lastb + 256 * (Jc & 1) : slots + 256 * sw_slot(d, Jc));
#pragma unroll
            for (int rg = 0; rg < 4; rg++) t[rg] = src[64 * rg + lane];
            sw_static_for([&](auto cc) {
                constexpr int Cc = decltype(cc)::value;
                const double lm = col > Cc ? Lc[col * SW_P + Cc] : 0.0, sc = col == Cc ? invd[Cc] : 1.0;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) t[rg] *= sc;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) t[rg] = __builtin_fma(-d_row_bcast<Cc>(t[rg]), lm, t[rg]);
                if ((Cc & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, 16>{});
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int r = 16 * d + r0 + 4 * rg, grow = 16 * Jc + r, gc = 16 * Jc + col;
                Lc[r * SW_P + col] = t[rg];
                if (V.stored(grow, gc)) A[V.idx(grow, gc)] = t[rg];
            }
        }
        __syncthreads();
        if (tid == 0) { const long long t1 = wall_clock64(); tB += t1 - t0; t0 = t1; }
        if (tid >= 16 && tid < 16 * (D + 1)) {        // rhs below the block: b -= L21 y, one row per thread
            double s2 = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) s2 += Lc[tid * SW_P + k] * ybuf[k];
            xr[(16 * Jc + tid) % 288] -= s2;
        }
        if (wv == wo) { if (so == 0) J0 = Jc + 16; else J1 = Jc + 16; }
        phase_c(T0, J0, J0 - Jc);
        phase_c(T1, J1, J1 - Jc);
        __syncthreads();
        if (tid == 0) { const long long t1 = wall_clock64(); tC += t1 - t0; t0 = t1; }
    }
    if (s_fail) { if (tid == 0) St->chol_ok = 0; if (side == 0) for (int i = tid; i < n; i += SW_NT) x[i] = 0; return; }
    SW_LANE;
    if (phase == 1) {
        // hand the window (block columns NFACT .. NFACT+15: every tile inside the separator is live when D == 16) and the rhs over
        auto dump_col = [&](ba_d4 (&T)[8], int J) {
            if (J < NFACT || J >= NFACT + 16) return;
            const int qp = J - NFACT;
#pragma unroll
            for (int d = 0; d < 16; d++) {
                if (qp + d > 15) continue;
                double* w = xch + SW_XCH_TILE(side, qp, d);
                if (d < 8) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) w[16 * (r0 + 4 * rg) + col] = T[d < 8 ? d : 0][rg];
                } else {
                    const double* sl = slots + 256 * sw_slot(d, J);
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) w[16 * (r0 + 4 * rg) + col] = sl[64 * rg + lane];
                }
            }
        };
        dump_col(T0, J0); dump_col(T1, J1);
        if (tid < 256) xch[SW_XCH_RHS(side) + tid] = xr[(16 * NFACT + tid) % 288];
        return;
    }

    // ---- backward substitution L' x = y, block columns in reverse; the next column's tiles are prefetched into registers
    if (tid == 0) t0 = wall_clock64();
    for (int i = tid; i < 288; i += SW_NT) xr[i] = 0.0;
    __syncthreads();
    if (phase == 2 && tid < 256) {                    // the separator's solution sits "below" this side's part
        const int r = 16 * NFACT + tid;
        xr[r % 288] = V.vec_ok(r) ? x[V.vec(r)] : 0.0;
    }
    const int rows = 16 * (D + 1);
    auto gload = [&](int Jb, int e) -> double {
        const int r = e >> 4, k = e & 15, grow = 16 * Jb + r, gc = 16 * Jb + k;
        if (r >= rows) return 0.0;
        if (gc > grow) return 0.0;
        if (!V.padding(grow, gc)) return grow - gc <= bw ? A[V.idx(grow, gc)] : 0.0;
        return grow == gc ? 1.0 : 0.0;
    };
    double nxt[9];
#pragma unroll
    for (int i = 0; i < 9; i++) nxt[i] = gload(NFACT - 1, tid + SW_NT * i);
    double ynext = (tid < 16 && V.vec_ok(16 * (NFACT - 1) + tid)) ? x[V.vec(16 * (NFACT - 1) + tid)] : 0.0;
    __syncthreads();
    for (int Jc = NFACT - 1; Jc >= 0; Jc--) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int e = tid + SW_NT * i; if (e < rows * 16) Lc[(e >> 4) * SW_P + (e & 15)] = nxt[i]; }
        const double ycur = ynext;
        if (Jc > 0) {
#pragma unroll
            for (int i = 0; i < 9; i++) nxt[i] = gload(Jc - 1, tid + SW_NT * i);
            ynext = (tid < 16 && V.vec_ok(16 * (Jc - 1) + tid)) ? x[V.vec(16 * (Jc - 1) + tid)] : 0.0;
        }
        __syncthreads();
        {
            const int k = tid & 15, p = tid >> 4;
            double s2 = 0;
            for (int r = 16 + p; r < rows; r += 32) s2 += Lc[r * SW_P + k] * xr[(16 * Jc + r) % 288];
            part[p * 16 + k] = s2;
        }
        __syncthreads();
        if (wv == 0) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) t += part[(8 * r0 + i) * 16 + col];
            t += d_xor_swap<16>(t); t += d_xor_swap<32>(t);
            double yl = d_shfl_idx(ycur, col) - t;               // ycur lives in lanes 0..15 of wave 0
            const double myinv = 1.0 / Lc[col * SW_P + col];
#pragma unroll
            for (int c = 15; c >= 0; c--) {
                const double xc = d_readlane(yl, c) * d_readlane(myinv, c);
                if (col == c) yl = xc; else if (col < c) yl -= Lc[c * SW_P + col] * xc;
            }
            if (lane < 16) { xr[(16 * Jc + lane) % 288] = yl; if (V.vec_ok(16 * Jc + lane)) x[V.vec(16 * Jc + lane)] = yl; }
        }
        __syncthreads();
    }
#undef SW_LANE
    if (tid == 0 && side == 0 && phase != 2) { tBack = wall_clock64() - t0; St->dbg[0] = tA; St->dbg[1] = tB; St->dbg[2] = tC; St->dbg[3] = tBack; St->dbg[4] = dbgl[0]; St->dbg[5] = dbgl[1]; St->dbg[6] = dbgl[2]; }
}

__global__ __launch_bounds__(SW_NT) void k_ba_chol_sw(double* __restrict__ A, const double* __restrict__ bs, int n, int bw,
                                                      double* __restrict__ x, ba_state* __restrict__ St, int phase, int nfact0,
                                                      double* __restrict__ A1, const double* __restrict__ bs1, int n1,
                                                      double* __restrict__ x1, int nfact1, double* __restrict__ xch)
{
    // phase 0: one workgroup, the whole system (A, bs, n, x).  phases 1 / 2 (two-way solve): workgroup 0 works on (A, bs, n, x) and
    // eliminates / back-substitutes its first nfact0 block columns; workgroup 1 does the same on (A1, bs1, n1, x1), the physically
    // reversed copy of the other end of the matrix (k_sw_reverse), with nfact1 block columns.
    extern __shared__ double sm[];
    if (phase && blockIdx.x == 1) sw_body<0>(sm, A1, bs1, n1, bw, x1, St, phase, nfact1, 1, xch);
    else sw_body<0>(sm, A, bs, n, bw, x, St, phase, nfact0, 0, xch);
}

// ---- k_sw_separator: the 256 x 256 separator system of the two-way solve, in the band layout of k_ba_chol (n = 256, bw = 255):
// S2 = W_top + W_bottom - A22 and b2 likewise.  The bottom window is in reversed coordinates: element (i, j) of the top tile
// (column q, distance d) is element (15-j, 15-i) of the bottom tile (column 15-q-d, same d).
__global__ __launch_bounds__(256) void k_sw_separator(const double* __restrict__ A, const double* __restrict__ bs, int bw, int split,
                                                      const double* __restrict__ xch, double* __restrict__ Ssep, double* __restrict__ bsep)
{
    const int r = blockIdx.x, c = threadIdx.x;       // r >= c only
    if (c == 0) bsep[r] = xch[SW_XCH_RHS(0) + r] + xch[SW_XCH_RHS(1) + 255 - r] - bs[16 * split + r];
    if (c > r) return;
    const int q = c >> 4, d = (r >> 4) - q, i = r & 15, j = c & 15;
    const double wt = xch[SW_XCH_TILE(0, q, d) + 16 * i + j];
    const double wb = xch[SW_XCH_TILE(1, 15 - q - d, d) + 16 * (15 - j) + (15 - i)];
    const size_t gr = 16 * (size_t)split + r, gc = 16 * (size_t)split + c;
    const double a = r - c <= bw ? A[gr * (bw + 1) + (gc - gr + bw)] : 0.0;
    Ssep[(size_t)r * 256 + (c - r + 255)] = wt + wb - a;
}

// ---- two-way solve helpers: the far end of the system in reversed order, physically (row r' of the copy = row npad-1-r' of
// the stored matrix read backwards; rows that map beyond n become identity padding), so that both workgroups run the same code
__global__ __launch_bounds__(256) void k_sw_reverse(const double* __restrict__ S, const double* __restrict__ bs, int n, int bw, int npad,
                                                    int n1, double* __restrict__ S1, double* __restrict__ bs1)
{
    const int rp = blockIdx.x;                       // row of the reversed copy, < n1
    const int co = npad - 1 - rp;                    // stored column of all its elements' mirror images
    for (int k = threadIdx.x; k <= bw; k += 256) {   // element (rp, rp - k)  <-  stored (co + k, co)
        const int ro = co + k;
        double v = 0.0;
        if (rp - k >= 0) v = ro < n ? S[(size_t)ro * (bw + 1) + (bw - k)] : (k == 0 ? 1.0 : 0.0);
        S1[(size_t)rp * (bw + 1) + (bw - k)] = v;
    }
    if (threadIdx.x == 0) bs1[rp] = co < n ? bs[co] : 0.0;
}
// mode 0: separator solution x[16 split .. +256) -> x1[n1-256 .. n1) (reversed);  mode 1: x1[0 .. n1-256) -> x (reversed)
__global__ void k_sw_rev_vec(double* __restrict__ x, double* __restrict__ x1, int n, int npad, int n1, int mode)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (mode == 0) { if (i < 256) x1[n1 - 256 + i] = x[npad - 1 - (n1 - 256 + i)]; }
    else if (i < n1 - 256 && npad - 1 - i < n) x[npad - 1 - i] = x1[i];
}

// host: one reduced-camera-system solve with the window kernel; split > 0 selects the two-way schedule (scratch: xch)
#define SW_SCRATCH_DOUBLES(n, bw) ((size_t)SW_XCH_DOUBLES + 65536 + 256 + ((size_t)(n) / 2 + 512) * ((bw) + 3))
static int sw_solve(bh_ctx* ctx, hipStream_t st, double* S, const double* bs, int n, int bw, double* x, ba_state* state, int split,
                    double* xch)
{
    const size_t lds = sizeof(double) * SW_LDS_DOUBLES;
    if (!split) {
        hipLaunchKernelGGL(k_ba_chol_sw, dim3(1), dim3(SW_NT), lds, st, S, bs, n, bw, x, state, 0, 0, (double*)nullptr, (const double*)nullptr, 0,
                           (double*)nullptr, 0, xch);
        BH_HIP(ctx, hipGetLastError());
        return BH_OK;
    }
    const int nblk = (n + 15) / 16, npad = 16 * nblk, nb2 = nblk - split - 16, n1 = 16 * (nb2 + 16);
    double* Ssep = xch + SW_XCH_DOUBLES; double* bsep = Ssep + 65536;
    double* S1 = bsep + 256; double* bs1 = S1 + (size_t)n1 * (bw + 1); double* x1 = bs1 + n1;
    hipLaunchKernelGGL(k_sw_reverse, dim3(n1), dim3(256), 0, st, (const double*)S, bs, n, bw, npad, n1, S1, bs1);
    hipLaunchKernelGGL(k_ba_chol_sw, dim3(2), dim3(SW_NT), lds, st, S, bs, n, bw, x, state, 1, split, S1, (const double*)bs1, n1, x1, nb2, xch);
    hipLaunchKernelGGL(k_sw_separator, dim3(256), dim3(256), 0, st, (const double*)S, bs, bw, split, (const double*)xch, Ssep, bsep);
    hipLaunchKernelGGL(k_ba_chol_dense<false>, dim3(1), dim3(1024), sizeof(double) * CD_LDS_DOUBLES, st, (const double*)Ssep, (const double*)bsep, 256, 255,
                       x + 16 * (size_t)split, state, (const bcr_elim*)nullptr, (double*)nullptr, (const double*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(k_sw_rev_vec, dim3(1), dim3(256), 0, st, x, x1, n, npad, n1, 0);
    hipLaunchKernelGGL(k_ba_chol_sw, dim3(2), dim3(SW_NT), lds, st, S, bs, n, bw, x, state, 2, split, S1, (const double*)bs1, n1, x1, nb2, xch);
    hipLaunchKernelGGL(k_sw_rev_vec, dim3((n1 + 255) / 256), dim3(256), 0, st, x, x1, n, npad, n1, 1);
    BH_HIP(ctx, hipGetLastError());
    return BH_OK;
}

// ---- block cyclic reduction of the banded reduced camera system (the solve on many CUs).
// bw <= 256, so in 256 x 256 blocks the matrix is block tridiagonal: a chain of N = ceil(n / 256) diagonal blocks D_i and the
// couplings between neighbours.  Every other block of the chain (the even positions) is eliminated AT THE SAME TIME - none of them
// touches another - which leaves a chain of half the length over the odd positions with fill between what were second neighbours;
// log2 N levels later one block is left.  That is a Cholesky factorisation in nested-dissection order (no pivoting needed: SPD),
// with every O(m^3) step on the matrix cores and the level's work spread over the chip:
//   k_ba_chol_dense<true>   one workgroup per eliminated block e: D_e = L L', y_e = L^-1 g_e            (in place)
//   k_bcr_trsm              U = L^-1 M for both couplings M (e x neighbour), one wave per 16-column strip; the solved tiles stay
//                           in the accumulator layout, which IS the B-operand layout of v_mfma_f64_16x16x4_f64 (row 4 rg + l/16)
//   k_bcr_update            survivor s: D_s -= U'U (both eliminated neighbours, lower tiles), g_s -= U'y; and the fill between the
//                           two survivors either side of e: -U_a' U_b, written with its rows on the block the NEXT level eliminates
//   k_bcr_back              levels in reverse: x_e = L^-T (y_e - U_l x_l - U_r x_r)
// Couplings are stored (eliminated block) x (neighbour) at every level, so every product has the form X'Y with X, Y row-major and
// the reduction index on the rows: both MFMA operands are coalesced 128-B row segments straight from L2.
#define BCR_BLK 65536
struct bcr_work { int out, kind, a0, b0, a1, ye0, ye1; };     // kind 0: D[out] -= U[a0]'U[a0] + U[a1]'U[a1], g[out] -= U[a0]'y[ye0] + U[a1]'y[ye1]
                                                              // kind 1: C[out] = -U[a0]'U[b0]
struct bcr_plan {
    int N = 0, nslots = 0, nlev = 0;
    std::vector<int> e_off, w_off;                            // per level: first entry of elim / work
    std::vector<bcr_elim> elim;
    std::vector<bcr_work> work;
};
static void bcr_make_plan(int N, bcr_plan* P)
{
    P->N = N; P->elim.clear(); P->work.clear(); P->e_off.assign(1, 0); P->w_off.assign(1, 0);
    std::vector<int> chain(N), next;
    for (int i = 0; i < N; i++) chain[i] = i;
    int base = 0, nslots = N - 1;                             // level-0 coupling p joins blocks p and p + 1
    for (;;) {
        const int k = (int)chain.size();
        for (int p = 0; p < k; p += 2)
            P->elim.push_back({ chain[p], p > 0 ? chain[p - 1] : -1, p + 1 < k ? chain[p + 1] : -1, p > 0 ? base + p - 1 : -1, p + 1 < k ? base + p : -1 });
        next.clear();
        for (int p = 1; p < k; p += 2) {
            next.push_back(chain[p]);
            P->work.push_back({ chain[p], 0, base + p - 1, -1, p + 1 < k ? base + p : -1, chain[p - 1], p + 1 < k ? chain[p + 1] : -1 });
        }
        const int nb = nslots;
        for (int q = 0; q + 1 < (int)next.size(); q++) {      // fill through the eliminated block at position 2q + 2
            const int sa = base + 2 * q + 1, sb = base + 2 * q + 2;        // (e x next[q]), (e x next[q + 1])
            P->work.push_back({ nb + q, 1, (q & 1) ? sb : sa, (q & 1) ? sa : sb, -1, -1, -1 });      // rows on the even position of the new chain
        }
        if (next.size() > 1) nslots += (int)next.size() - 1;
        P->e_off.push_back((int)P->elim.size()); P->w_off.push_back((int)P->work.size());
        if (k == 1) break;
        base = nb; chain = next;
    }
    P->nslots = nslots; P->nlev = (int)P->e_off.size() - 1;
}

// band storage -> diagonal blocks (lower part, identity padding), level-0 couplings (rows on the even block), right-hand side
__global__ __launch_bounds__(256) void k_bcr_extract(const double* __restrict__ A, const double* __restrict__ bs, int n, int bw, int N,
                                                     double* __restrict__ Dblk, double* __restrict__ Cpl, double* __restrict__ g)
{
    const int b = blockIdx.x, r = blockIdx.y, c = threadIdx.x;
    const size_t LD = (size_t)bw + 1;
    if (b < N) {
        if ((c >> 4) > (r >> 4)) return;                      // tiles above the diagonal are never read
        const int R = 256 * b + r, C = 256 * b + c;
        double v = 0.0;
        if (R < n && C < n) { if (c <= r && r - c <= bw) v = A[R * LD + (C - R + bw)]; }
        else if (R == C) v = 1.0;
        Dblk[(size_t)b * BCR_BLK + r * 256 + c] = v;
        if (c == 0) g[256 * b + r] = R < n ? bs[R] : 0.0;
    } else {
        const int p = b - N;
        const int R = 256 * (p + 1) + r, C = 256 * p + c;     // element (R, C) of the lower triangle, read along the band row
        const double v = (R < n && R - C <= bw) ? A[R * LD + (C - R + bw)] : 0.0;
        Cpl[(size_t)p * BCR_BLK + ((p & 1) ? r * 256 + c : c * 256 + r)] = v;      // odd p: rows on block p + 1; even p: rows on block p
    }
}

// U = L^-1 M.  A workgroup takes four 16-column strips of one coupling (a wave each); the 16 x 16(i+1) tile row i of L is staged
// through LDS by all four waves (coalesced 16-B loads, double-buffered: row i + 1 is in flight while row i feeds the matrix cores -
// read straight from L2, every one of the 544 dependent MFMAs of a strip waited for its own operand load: 101 us per level).
#define TR_P 262                                     // LDS row pitch in doubles: 16 rows x 4 k-offsets land in 64 distinct banks
__global__ __launch_bounds__(256) void k_bcr_trsm(const bcr_elim* __restrict__ el, const double* __restrict__ Dblk, double* __restrict__ Cpl)
{
    __shared__ double sL[2][16 * TR_P];
    const bcr_elim E = el[blockIdx.z];
    const int slot = blockIdx.y ? E.sr : E.sl;
    if (slot < 0) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, col = lane & 15, r0 = lane >> 4;
    const int lr = tid >> 4, lc = (tid & 15) * 2;     // staging: row lr of the tile row, columns lc + 32 m
    const double* __restrict__ Lg = Dblk + (size_t)E.e * BCR_BLK + (size_t)lr * 256 + lc;
    double* __restrict__ M = Cpl + (size_t)slot * BCR_BLK + 16 * (blockIdx.x * 4 + wv) + (size_t)r0 * 256 + col;
    ba_d4 U[16];
    {
        const double2 v = *reinterpret_cast<const double2*>(Lg);      // tile row 0 is the first diagonal tile alone
        if (lc < 16) { sL[0][lr * TR_P + lc] = v.x; sL[0][lr * TR_P + lc + 1] = v.y; }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        __syncthreads();                              // tile row i is in sL[i & 1]; every wave is done with sL[(i + 1) & 1]
        double2 nx[8];
        if (i < 15) {
#pragma unroll
            for (int m = 0; m < (16 * (i + 2) + 31) / 32; m++)
                if (lc + 32 * m < 16 * (i + 2)) nx[m] = *reinterpret_cast<const double2*>(Lg + (size_t)(16 * (i + 1)) * 256 + 32 * m);
        }
        ba_d4 acc;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) acc[rg] = M[(size_t)(16 * i + 4 * rg) * 256];
        const double* __restrict__ sa = sL[i & 1] + col * TR_P + r0;
#pragma unroll
        for (int k = 0; k < i; k++) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sa[16 * k + 4 * ks], U[k][ks], acc, 0, 0, 0);
        }
        ba_d4 u = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int ks = 0; ks < 4; ks++)                        // the diagonal tile holds the inverse of its factor
            u = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[16 * i + 4 * ks], acc[ks], u, 0, 0, 0);
        U[i] = u;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) M[(size_t)(16 * i + 4 * rg) * 256] = u[rg];
        if (i < 15) {
            double* __restrict__ dst = sL[(i + 1) & 1] + lr * TR_P + lc;
#pragma unroll
            for (int m = 0; m < (16 * (i + 2) + 31) / 32; m++)
                if (lc + 32 * m < 16 * (i + 2)) { dst[32 * m] = nx[m].x; dst[32 * m + 1] = nx[m].y; }
        }
    }
}

// X'Y products of one level.  A workgroup owns a 32 x 32 block of the result (2 x 2 tiles: two A and two B operand loads feed four
// MFMAs) and splits the reduction index over its four waves (64 rows each - at the small levels the length of a wave's MFMA chain
// is what the launch waits for); the four partial blocks meet in LDS and wave w adds up tile w in wave order (fixed summation order).
__global__ __launch_bounds__(256) void k_bcr_update(const bcr_work* __restrict__ wk, double* __restrict__ Dblk, double* __restrict__ Cpl,
                                                    double* __restrict__ g, const double* __restrict__ yv)
{
    __shared__ double red[4 * 4 * 4 * 64];            // [wave][tile][rg][lane]
    __shared__ double gred[4][2][16];
    const bcr_work W = wk[blockIdx.y];
    const int bi = blockIdx.x >> 3, bj = blockIdx.x & 7;
    if (W.kind == 0 && bj > bi) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, col = lane & 15, r0 = lane >> 4;
    ba_d4 acc[4] = { { 0.0, 0.0, 0.0, 0.0 }, { 0.0, 0.0, 0.0, 0.0 }, { 0.0, 0.0, 0.0, 0.0 }, { 0.0, 0.0, 0.0, 0.0 } };
    double g0 = 0.0, g1 = 0.0;
    const bool do_g = W.kind == 0 && bi == bj;
#pragma unroll 1
    for (int side = 0; side < 2; side++) {
        const int sa = side ? W.a1 : W.a0;
        if (sa < 0) continue;
        const int sb = W.kind ? W.b0 : sa;
        const double* __restrict__ Ua = Cpl + (size_t)sa * BCR_BLK + (size_t)(64 * wv + r0) * 256 + 32 * bi + col;
        const double* __restrict__ Ub = Cpl + (size_t)sb * BCR_BLK + (size_t)(64 * wv + r0) * 256 + 32 * bj + col;
        const int yb = side ? W.ye1 : W.ye0;
        const double* __restrict__ ye = yv + 256 * (size_t)(yb < 0 ? 0 : yb) + 64 * wv + r0;
#pragma unroll 8
        for (int kk = 0; kk < 16; kk++) {
            const double a0 = Ua[(size_t)kk * 1024], a1 = Ua[(size_t)kk * 1024 + 16], b0 = Ub[(size_t)kk * 1024], b1 = Ub[(size_t)kk * 1024 + 16];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
            if (do_g) { const double yk = ye[4 * kk]; g0 = __builtin_fma(a0, yk, g0); g1 = __builtin_fma(a1, yk, g1); }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) red[((wv * 4 + t) * 4 + rg) * 64 + lane] = acc[t][rg];
    if (do_g) {
        g0 += d_xor_swap<16>(g0); g0 += d_xor_swap<32>(g0); g1 += d_xor_swap<16>(g1); g1 += d_xor_swap<32>(g1);
        if (lane < 16) { gred[wv][0][col] = g0; gred[wv][1][col] = g1; }
    }
    __syncthreads();
    const int ti = 2 * bi + (wv >> 1), tj = 2 * bj + (wv & 1);
    if (do_g && tid < 32) {
        const int h = tid >> 4, c = tid & 15;
        g[256 * (size_t)W.out + 32 * bi + 16 * h + c] -= ((gred[0][h][c] + gred[1][h][c]) + gred[2][h][c]) + gred[3][h][c];
    }
    if (W.kind == 0 && tj > ti) return;
    double v[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
        v[rg] = ((red[((0 * 4 + wv) * 4 + rg) * 64 + lane] + red[((1 * 4 + wv) * 4 + rg) * 64 + lane]) + red[((2 * 4 + wv) * 4 + rg) * 64 + lane]) + red[((3 * 4 + wv) * 4 + rg) * 64 + lane];
    double* __restrict__ O = (W.kind == 0 ? Dblk : Cpl) + (size_t)W.out * BCR_BLK + (size_t)(16 * ti + r0) * 256 + 16 * tj + col;
    if (W.kind == 0) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) O[(size_t)rg * 1024] -= v[rg];
    } else {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) O[(size_t)rg * 1024] = -v[rg];
    }
}

// x_e = L^-T (y_e - U_l x_l - U_r x_r).  1024 threads stream the two couplings (16 rows per wave, every load in flight at once:
// a 256-thread version spent most of its 47 us waiting for 1 MB through four waves); the first 256 then walk the factor upwards,
// the next block row already in registers when the previous one's barrier falls.
__global__ __launch_bounds__(1024) void k_bcr_back(const bcr_elim* __restrict__ el, const double* __restrict__ Dblk, const double* __restrict__ Cpl,
                                                   const double* __restrict__ yv, double* __restrict__ xp, double* __restrict__ x, int n,
                                                   const ba_state* __restrict__ St)
{
    __shared__ double z[256], xi[16];
    const bcr_elim E = el[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t o = 256 * (size_t)E.e;
    if (!St->chol_ok) { if (tid < 256) { xp[o + tid] = 0.0; if (o + tid < (size_t)n) x[o + tid] = 0.0; } return; }
    const double* L = Dblk + (size_t)E.e * BCR_BLK;
    const bool walker = tid < 256;
    double ln[16], ld[16];
    if (walker) {
#pragma unroll
        for (int r = 0; r < 16; r++) { ln[r] = L[(size_t)(240 + r) * 256 + (tid < 240 ? tid : 0)]; ld[r] = L[(size_t)(240 + r) * 256 + 240 + (tid & 15)]; }
    }
    {   // z = y_e - U_l x_l - U_r x_r: 16 lanes per row (DPP row sums), 4 rows per wave instruction, 16 rows per wave
        const int cl = lane & 15, rq = lane >> 4;
        const double* Ul = Cpl + (size_t)(E.sl < 0 ? 0 : E.sl) * BCR_BLK + 4 * cl;
        const double* Ur = Cpl + (size_t)(E.sr < 0 ? 0 : E.sr) * BCR_BLK + 4 * cl;
        const double* xlp = xp + 256 * (size_t)(E.l < 0 ? 0 : E.l) + 4 * cl;
        const double* xrp = xp + 256 * (size_t)(E.r < 0 ? 0 : E.r) + 4 * cl;
        double s2[4] = { 0.0, 0.0, 0.0, 0.0 };
        if (E.sl >= 0) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const double4 xv = *reinterpret_cast<const double4*>(xlp + 64 * m);
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const double4 u = *reinterpret_cast<const double4*>(Ul + (size_t)(16 * wv + 4 * it + rq) * 256 + 64 * m);
                    s2[it] = __builtin_fma(u.x, xv.x, __builtin_fma(u.y, xv.y, __builtin_fma(u.z, xv.z, __builtin_fma(u.w, xv.w, s2[it]))));
                }
            }
        }
        if (E.sr >= 0) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const double4 xv = *reinterpret_cast<const double4*>(xrp + 64 * m);
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const double4 u = *reinterpret_cast<const double4*>(Ur + (size_t)(16 * wv + 4 * it + rq) * 256 + 64 * m);
                    s2[it] = __builtin_fma(u.x, xv.x, __builtin_fma(u.y, xv.y, __builtin_fma(u.z, xv.z, __builtin_fma(u.w, xv.w, s2[it]))));
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            double v = s2[it];
            v += d_row_shr<1>(v); v += d_row_shr<2>(v); v += d_row_shr<4>(v); v += d_row_shr<8>(v);      // lane 15 of the row holds the sum
            const int row = 16 * wv + 4 * it + rq;
            if (cl == 15) z[row] = yv[o + row] - v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int J = 15; J >= 0; J--) {
        if (tid < 16) {                                       // x_J = Linv_JJ' z_J
            double s3 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r++) s3 = __builtin_fma(ld[r], z[16 * J + r], s3);
            xi[tid] = s3;
            xp[o + 16 * J + tid] = s3;
            if (o + 16 * J + tid < (size_t)n) x[o + 16 * J + tid] = s3;
            if (J > 0) {                                      // the next diagonal tile: in flight across both barriers
#pragma unroll
                for (int r = 0; r < 16; r++) ld[r] = L[(size_t)(16 * (J - 1) + r) * 256 + 16 * (J - 1) + tid];
            }
        }
        __syncthreads();
        if (tid < 16 * J) {                                   // z_j -= L(J, j)' x_J for the block rows above
            double s3 = z[tid];
#pragma unroll
            for (int r = 0; r < 16; r++) s3 = __builtin_fma(-ln[r], xi[r], s3);
            z[tid] = s3;
        }
        if (walker && J > 0) {                                // block row J - 1 of the factor: in flight across the barrier and the next x_J
#pragma unroll
            for (int r = 0; r < 16; r++) ln[r] = L[(size_t)(16 * (J - 1) + r) * 256 + (tid < 16 * (J - 1) ? tid : 0)];
        }
        __syncthreads();
    }
}

#define BCR_SCRATCH_BYTES(N) (sizeof(double) * ((size_t)(N) * BCR_BLK + (size_t)(2 * (N) + 2) * BCR_BLK + 3 * (size_t)(N) * 256) + 64 * (size_t)(4 * (N) + 8))
struct bcr_dev { bcr_plan plan; double *D = nullptr, *C = nullptr, *g = nullptr, *y = nullptr, *xp = nullptr; bcr_elim* elim = nullptr; bcr_work* work = nullptr; };
// carve the scratch buffer and upload the elimination plan (on the stream; the plan's host vectors live in `dv`)
static int bcr_prepare(bh_ctx* ctx, hipStream_t st, bcr_dev* dv, int n, void* scratch)
{
    const int N = (n + 255) / 256;
    bcr_make_plan(N, &dv->plan);
    dv->D = (double*)scratch; dv->C = dv->D + (size_t)N * BCR_BLK; dv->g = dv->C + (size_t)(2 * N + 2) * BCR_BLK;
    dv->y = dv->g + (size_t)N * 256; dv->xp = dv->y + (size_t)N * 256;
    dv->elim = (bcr_elim*)(dv->xp + (size_t)N * 256); dv->work = (bcr_work*)(dv->elim + 2 * N + 4);
    if (dv->plan.nslots > 2 * N + 2 || (int)dv->plan.elim.size() > 2 * N + 4 || (int)dv->plan.work.size() > 2 * N + 4)
        return bh_set_err(ctx, BH_ERR_CAPACITY, "bcr_prepare: plan larger than its scratch");
    BH_HIP(ctx, hipMemcpyAsync(dv->elim, dv->plan.elim.data(), sizeof(bcr_elim) * dv->plan.elim.size(), hipMemcpyHostToDevice, st));
    if (!dv->plan.work.empty())
        BH_HIP(ctx, hipMemcpyAsync(dv->work, dv->plan.work.data(), sizeof(bcr_work) * dv->plan.work.size(), hipMemcpyHostToDevice, st));
    return BH_OK;
}
static void bcr_forward(hipStream_t st, const bcr_dev* dv, ba_state* state)
{
    const bcr_plan& P = dv->plan;
    for (int lv = 0; lv < P.nlev; lv++) {
        const int ne = P.e_off[lv + 1] - P.e_off[lv], nw = P.w_off[lv + 1] - P.w_off[lv];
        hipLaunchKernelGGL(k_ba_chol_dense<true>, dim3(ne), dim3(1024), sizeof(double) * CD_LDS_DOUBLES, st, (const double*)nullptr, (const double*)nullptr, 256, 255,
                           (double*)nullptr, state, (const bcr_elim*)(dv->elim + P.e_off[lv]), dv->D, (const double*)dv->g, dv->y);
        if (nw) {
            hipLaunchKernelGGL(k_bcr_trsm, dim3(4, 2, ne), dim3(256), 0, st, (const bcr_elim*)(dv->elim + P.e_off[lv]), (const double*)dv->D, dv->C);
            hipLaunchKernelGGL(k_bcr_update, dim3(64, nw), dim3(256), 0, st, (const bcr_work*)(dv->work + P.w_off[lv]), dv->D, dv->C, dv->g, (const double*)dv->y);
        }
    }
}
static void bcr_backward(hipStream_t st, const bcr_dev* dv, double* x, int n, ba_state* state)
{
    const bcr_plan& P = dv->plan;
    for (int lv = P.nlev - 1; lv >= 0; lv--)
        if (P.e_off[lv + 1] > P.e_off[lv])
            hipLaunchKernelGGL(k_bcr_back, dim3(P.e_off[lv + 1] - P.e_off[lv]), dim3(1024), 0, st, (const bcr_elim*)(dv->elim + P.e_off[lv]), (const double*)dv->D,
                               (const double*)dv->C, (const double*)dv->y, dv->xp, x, n, (const ba_state*)state);
}
static int bcr_solve(bh_ctx* ctx, hipStream_t st, const bcr_dev* dv, const double* S, const double* bs, int n, int bw, double* x, ba_state* state)
{
    const bcr_plan& P = dv->plan;
    hipLaunchKernelGGL(k_bcr_extract, dim3(2 * P.N - 1, 256), dim3(256), 0, st, S, bs, n, bw, P.N, dv->D, dv->C, dv->g);
    bcr_forward(st, dv, state);
    bcr_backward(st, dv, x, n, state);
    BH_HIP(ctx, hipGetLastError());
    return BH_OK;
}

// ================================================================================================ distributed reduced-camera solve
// Global BA over R GPUs (BASELINE configs[4]).  Round 2 all-reduced the whole band of S (59 MB at configs[4]) to every rank once per
// Levenberg trial and every rank then ran the same 1.6 ms block cyclic reduction: at most 1.8x on 8 GPUs before any communication
// (VERDICT r02).  Now the chain of 256-row blocks is cut into R contiguous segments, one per rank:
//   1. the partial S of every rank is REDUCE-SCATTERED: rank r receives the summed band rows of its own segment only (59 MB / R);
//   2. rank r eliminates the interior blocks of its segment with the cyclic-reduction kernels (same kernels, a plan in which the
//      segment's first and last block are never eliminated) - no communication; what is left is the Schur complement of its
//      interior on its two end blocks: D_first, D_last, the fill between them, their right-hand sides;
//   3. those, with the original coupling to the previous segment's last block, are ALL-GATHERED (4 blocks = 2 MB per rank) and form a
//      chain of 2R blocks, which every rank reduces and solves - the only replicated arithmetic left (4 levels at R = 8);
//   4. the end blocks' solution in hand, rank r back-substitutes through its interior, and the step of the keyframes is all-gathered
//      (240 KB at configs[4]).
// This is the nested-dissection factorisation of the single-GPU solver with the separators chosen at the rank boundaries: the same
// solution up to rounding (the tests ask 1e-9 and the same Levenberg decisions).
struct dbcr_dev {
    int active = 0;
    int seg = 0, B0 = 0, nb = 0;       // blocks per rank (equal chunks; the last rank's may reach past N), first block, blocks this rank holds (>= 2)
    bcr_dev loc, glob;                  // loc: the segment (local block ids 0 .. nb-1, ends pinned); glob: the chain of 2R end blocks
    std::vector<uint8_t> orient;        // level-0 coupling p of the segment (local blocks p, p+1): 1 = rows on block p, 0 = rows on block p+1
    int fslot = 0;                      // slot of the coupling between the two end blocks once the interior is gone
    bh_buf buf_loc, buf_glob, gather, orient_d;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double ms_replicated = 0; int n_replicated = 0;
    long long bytes_rs = 0, bytes_ag = 0;
};
static void dbcr_release(dbcr_dev* d);
#define DBCR_CH (4 * BCR_BLK + 512)     // one rank's share of the gathered end blocks: D_first | D_last | F | C_prev | g_first | g_last (doubles)

// Plan for a chain 0 .. nb-1 whose two ends are pinned: per level every other block of what is left of the interior.  T[b] = level at
// which block b goes (pinned: never); every coupling is stored with its rows on the end that goes first.
static void dbcr_make_plan(int nb, bcr_plan* P, std::vector<uint8_t>* orient0, int* fslot)
{
    const int NEVER = 1 << 30;
    std::vector<int> T(nb, NEVER);
    {   // pass 1: elimination levels
        std::vector<int> chain(nb); for (int i = 0; i < nb; i++) chain[i] = i;
        for (int lv = 0;; lv++) {
            std::vector<int> next; bool any = false, prev = false;
            for (int i = 0; i < (int)chain.size(); i++) {
                const bool pinned = chain[i] == 0 || chain[i] == nb - 1;
                const bool mark = !pinned && !prev;
                if (mark) { T[chain[i]] = lv; any = true; } else next.push_back(chain[i]);
                prev = mark;
            }
            if (!any) break;
            chain = next;
        }
    }
    P->N = nb; P->elim.clear(); P->work.clear(); P->e_off.assign(1, 0); P->w_off.assign(1, 0);
    orient0->assign(std::max(nb - 1, 1), 0);
    for (int p = 0; p + 1 < nb; p++) (*orient0)[p] = T[p] < T[p + 1] || (T[p] == NEVER && T[p + 1] == NEVER) ? 1 : 0;      // both pinned (nb == 2): rows on the first
    std::vector<int> chain(nb), cslot(std::max(nb - 1, 0));
    for (int i = 0; i < nb; i++) chain[i] = i;
    for (int p = 0; p + 1 < nb; p++) cslot[p] = p;
    int nslots = nb - 1;
    for (int lv = 0;; lv++) {
        const int k = (int)chain.size();
        std::vector<char> mark(k, 0);
        bool any = false;
        for (int i = 0; i < k; i++) if (T[chain[i]] == lv) { mark[i] = 1; any = true; }
        if (!any) break;
        std::vector<int> fill(k, -1);                         // slot of the new coupling through eliminated position i
        for (int i = 0; i < k; i++) if (mark[i])
            P->elim.push_back({ chain[i], i > 0 ? chain[i - 1] : -1, i + 1 < k ? chain[i + 1] : -1, i > 0 ? cslot[i - 1] : -1, i + 1 < k ? cslot[i] : -1 });
        for (int i = 0; i < k; i++) if (!mark[i]) {
            int a[2] = { -1, -1 }, ye[2] = { -1, -1 }, m = 0;
            if (i > 0 && mark[i - 1]) { a[m] = cslot[i - 1]; ye[m] = chain[i - 1]; m++; }
            if (i + 1 < k && mark[i + 1]) { a[m] = cslot[i]; ye[m] = chain[i + 1]; m++; }
            if (m) P->work.push_back({ chain[i], 0, a[0], -1, a[1], ye[0], ye[1] });
        }
        for (int i = 0; i < k; i++) if (mark[i] && i > 0 && i + 1 < k) {
            const int l = chain[i - 1], r = chain[i + 1];
            const bool rows_on_l = T[l] < T[r] || (T[l] == NEVER && T[r] == NEVER);
            fill[i] = nslots++;
            P->work.push_back({ fill[i], 1, rows_on_l ? cslot[i - 1] : cslot[i], rows_on_l ? cslot[i] : cslot[i - 1], -1, -1, -1 });
        }
        std::vector<int> nchain, ncslot;
        for (int i = 0; i < k; i++) if (!mark[i]) {
            if (!nchain.empty()) ncslot.push_back(mark[i - 1] ? fill[i - 1] : cslot[i - 1]);
            nchain.push_back(chain[i]);
        }
        P->e_off.push_back((int)P->elim.size()); P->w_off.push_back((int)P->work.size());
        chain = nchain; cslot = ncslot;
    }
    P->nslots = nslots; P->nlev = (int)P->e_off.size() - 1;
    *fslot = cslot.empty() ? 0 : cslot[0];                    // the chain is (first, last) now
}

// band rows of this rank's segment -> its diagonal blocks, the couplings inside the segment (orientation per plan), the right-hand
// side, and - straight into this rank's share of the gather buffer - the coupling to the previous segment's last block (rows on this
// segment's first block: the band's own orientation).  grid (2 nb, 256): blocks [0, nb) diagonal, [nb, 2nb-1) couplings, 2nb-1: C_prev
__global__ __launch_bounds__(256) void k_dbcr_extract(const double* __restrict__ A, const double* __restrict__ bs, int n, int bw, int B0, int nb,
                                                      const uint8_t* __restrict__ orient, double* __restrict__ Dblk, double* __restrict__ Cpl,
                                                      double* __restrict__ g, double* __restrict__ cprev)
{
    const int b = blockIdx.x, r = blockIdx.y, c = threadIdx.x;
    const size_t LD = (size_t)bw + 1;
    if (b < nb) {
        if ((c >> 4) > (r >> 4)) return;                      // tiles above the diagonal are never read
        const long long R = 256LL * (B0 + b) + r, C = 256LL * (B0 + b) + c;
        double v = 0.0;
        if (R < n && C < n) { if (c <= r && r - c <= bw) v = A[R * LD + (C - R + bw)]; }
        else if (R == C) v = 1.0;
        Dblk[(size_t)b * BCR_BLK + r * 256 + c] = v;
        if (c == 0) g[256 * b + r] = R < n ? bs[R] : 0.0;
    } else if (b < 2 * nb - 1) {
        const int p = b - nb;
        const long long R = 256LL * (B0 + p + 1) + r, C = 256LL * (B0 + p) + c;
        const double v = (R < n && R - C <= bw) ? A[R * LD + (C - R + bw)] : 0.0;
        Cpl[(size_t)p * BCR_BLK + (orient[p] ? c * 256 + r : r * 256 + c)] = v;
    } else {
        const long long R = 256LL * B0 + r, C = 256LL * (B0 - 1) + c;
        cprev[r * 256 + c] = (B0 > 0 && R < n && R - C <= bw) ? A[R * LD + (C - R + bw)] : 0.0;
    }
}
// the segment's end blocks after the interior is gone -> this rank's share of the gather buffer
__global__ __launch_bounds__(256) void k_dbcr_pack(const double* __restrict__ Dblk, const double* __restrict__ Cpl, const double* __restrict__ g, int nb, int fslot,
                                                   double* __restrict__ out)
{
    const int which = blockIdx.y;                             // 0 D_first, 1 D_last, 2 F
    const double* src = which == 0 ? Dblk : (which == 1 ? Dblk + (size_t)(nb - 1) * BCR_BLK : Cpl + (size_t)fslot * BCR_BLK);
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[(size_t)which * BCR_BLK + i] = src[i];
    if (which == 0 && i < 256) { out[4 * (size_t)BCR_BLK + i] = g[i]; out[4 * (size_t)BCR_BLK + 256 + i] = g[256 * (size_t)(nb - 1) + i]; }
}
// gathered shares -> the chain of 2R blocks in the layout the cyclic-reduction kernels take (coupling p: even p = F of rank p/2,
// rows on block p; odd p = C_prev of rank (p+1)/2, rows on block p+1)
__global__ __launch_bounds__(256) void k_dbcr_unpack(const double* __restrict__ gat, int Rk, double* __restrict__ Dblk, double* __restrict__ Cpl, double* __restrict__ g)
{
    const int r = blockIdx.z, which = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const double* src = gat + (size_t)r * DBCR_CH;
    const double v = src[(size_t)which * BCR_BLK + i];
    if (which == 0) Dblk[(size_t)(2 * r) * BCR_BLK + i] = v;
    else if (which == 1) Dblk[(size_t)(2 * r + 1) * BCR_BLK + i] = v;
    else if (which == 2) Cpl[(size_t)(2 * r) * BCR_BLK + i] = v;
    else if (r > 0) Cpl[(size_t)(2 * r - 1) * BCR_BLK + i] = v;
    if (which == 0 && i < 256) { g[256 * (size_t)(2 * r) + i] = src[4 * (size_t)BCR_BLK + i]; g[256 * (size_t)(2 * r + 1) + i] = src[4 * (size_t)BCR_BLK + 256 + i]; }
}
// solution of the two end blocks -> the segment's padded solution and its output rows
__global__ __launch_bounds__(256) void k_dbcr_ends(const double* __restrict__ xg, int rank, int nb, double* __restrict__ xp, double* __restrict__ x, int nloc)
{
    const int i = threadIdx.x, which = blockIdx.x;
    const double v = xg[256 * (size_t)(2 * rank + which) + i];
    const size_t o = which ? 256 * (size_t)(nb - 1) : 0;
    xp[o + i] = v;
    if ((long long)(o + i) < nloc) x[o + i] = v;
}

static int dbcr_carve(bh_ctx* ctx, hipStream_t st, bcr_dev* dv, int N, void* scratch)
{
    dv->D = (double*)scratch; dv->C = dv->D + (size_t)N * BCR_BLK; dv->g = dv->C + (size_t)(2 * N + 2) * BCR_BLK;
    dv->y = dv->g + (size_t)N * 256; dv->xp = dv->y + (size_t)N * 256;
    dv->elim = (bcr_elim*)(dv->xp + (size_t)N * 256); dv->work = (bcr_work*)(dv->elim + 2 * N + 4);
    if (dv->plan.nslots > 2 * N + 2 || (int)dv->plan.elim.size() > 2 * N + 4 || (int)dv->plan.work.size() > 2 * N + 4)
        return bh_set_err(ctx, BH_ERR_CAPACITY, "dbcr: plan larger than its scratch");
    if (!dv->plan.elim.empty()) BH_HIP(ctx, hipMemcpyAsync(dv->elim, dv->plan.elim.data(), sizeof(bcr_elim) * dv->plan.elim.size(), hipMemcpyHostToDevice, st));
    if (!dv->plan.work.empty()) BH_HIP(ctx, hipMemcpyAsync(dv->work, dv->plan.work.data(), sizeof(bcr_work) * dv->plan.work.size(), hipMemcpyHostToDevice, st));
    return BH_OK;
}

// ---- k_ba_update: trial state = current state (+) step.  blocks [0, pblocks): points (back-substitution); rest: cameras
__device__ __forceinline__ void ba_update_body(int bid, const ba_cam* __restrict__ cams, ba_cam* __restrict__ cams_t,
                                                   const double* __restrict__ pts, double* __restrict__ pts_t,
                                                   const int* __restrict__ cslot, int C, int Pa, int pblocks,
                                                   const int* __restrict__ p_start, const int* __restrict__ p_edges,
                                                   const int* __restrict__ e_cam, const double* __restrict__ Hpl,
                                                   const double* __restrict__ Dinv, const double* __restrict__ bl,
                                                   const double* __restrict__ dxp, double* __restrict__ dxl, ba_K K,
                                                   double lambda = 0.0, double* __restrict__ mine = nullptr /* this thread's dx'(lambda dx + b) */)
{
    if (mine) *mine = 0.0;
    if (bid < pblocks) {
        // BP_G lanes per landmark (as the linearisation): a lane takes every BP_G-th edge - one thread per landmark walked its ~8 edges one
        // dependent load chain after the other (k_ba_update: 25 us at configs[2], 13 us for a live window) - and the partial sums of
        // Hpl' dx meet in an xor tree on DPP; lane 0 of the group finishes the landmark
        const int j = (bid * 256 + (int)threadIdx.x) / BP_G, sub = threadIdx.x & (BP_G - 1);
        const bool live = j < Pa;
        double acc[3] = { 0, 0, 0 };
        if (live)
            for (int u = p_start[j] + sub; u < p_start[j + 1]; u += BP_G) {
                const int e = p_edges[u];
                const int s = cslot[e_cam[e]];
                if (s < 0) continue;
                const double* B = Hpl + 18 * (size_t)e; const double* d = dxp + 6 * s;
#pragma unroll
                for (int bb = 0; bb < 3; bb++)
#pragma unroll
                    for (int a = 0; a < 6; a++) acc[bb] += B[3 * a + bb] * d[a];
            }
#pragma unroll
        for (int i = 0; i < 3; i++) { acc[i] += d_xor_dpp<1>(acc[i]); acc[i] += d_xor_dpp<2>(acc[i]); acc[i] += d_xor_dpp<4>(acc[i]); }
        if (!live || sub) return;
        const double c[3] = { bl[3 * j] - acc[0], bl[3 * j + 1] - acc[1], bl[3 * j + 2] - acc[2] };
        const double* Di = Dinv + 9 * (size_t)j;
        double q = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double dl = Di[3 * a] * c[0] + Di[3 * a + 1] * c[1] + Di[3 * a + 2] * c[2];
            dxl[3 * j + a] = dl;
            pts_t[3 * j + a] = pts[3 * j + a] + dl;
            q += dl * (lambda * dl + bl[3 * j + a]);
        }
        if (mine) *mine = q;
    } else {
        const int i = (bid - pblocks) * 256 + threadIdx.x;
        if (i >= C) return;
        ba_cam c = cams[i];
        const int s = cslot[i];
        if (s >= 0) d_cam_oplus(&c, dxp + 6 * s, K);
        cams_t[i] = c;
    }
}
__global__ __launch_bounds__(256) void k_ba_update(const ba_cam* __restrict__ cams, ba_cam* __restrict__ cams_t,
                                                   const double* __restrict__ pts, double* __restrict__ pts_t,
                                                   const int* __restrict__ cslot, int C, int Pa, int pblocks,
                                                   const int* __restrict__ p_start, const int* __restrict__ p_edges,
                                                   const int* __restrict__ e_cam, const double* __restrict__ Hpl,
                                                   const double* __restrict__ Dinv, const double* __restrict__ bl,
                                                   const double* __restrict__ dxp, double* __restrict__ dxl, ba_K K,
                                                   ba_fan F, const ba_state* __restrict__ S)
{
    __shared__ double s_f[4];
    double mine = 0;
    ba_update_body(blockIdx.x, cams, cams_t, pts, pts_t, cslot, C, Pa, pblocks, p_start, p_edges, e_cam, Hpl, Dinv, bl, dxp, dxl, K, F.u_sum ? S->lambda : 0.0, &mine);
    if (!F.u_sum || (int)blockIdx.x >= pblocks) return;
    const double t = fan_block_sum256(mine, s_f);
    if (threadIdx.x == 0) fan_store(F.u_sum + blockIdx.x, t);    // read by the last workgroup of the NEXT launch (k_ba_err)
}

// ---- k_ba_err: per-edge error and robust chi at the trial state
__host__ __device__ __forceinline__ int ba_err_ept(int Ea) { return Ea > (1 << 18) ? 4 : 1; }
__host__ __device__ __forceinline__ int ba_err_blocks(int Ea) { const int per = 256 * ba_err_ept(Ea); return (Ea + per - 1) / per; }
__device__ __forceinline__ void ba_err_body(int bid, const ba_cam* __restrict__ cams, const double* __restrict__ pts,
                                                const int* __restrict__ e_cam, const int* __restrict__ e_pt,
                                                const double* __restrict__ e_meas, int Ea, double delta,
                                                double* __restrict__ err, double* __restrict__ chi_e, double* __restrict__ mine = nullptr)
{
    // a workgroup takes 256 x ept consecutive observations: one per thread for a local window, four beyond 2^18 observations (configs[4]: a quarter of the
    // workgroups, whose fixed cost - the workgroup's sum, its slot, its arrival - was more than the 1 us of work a workgroup had)
    const int ept = ba_err_ept(Ea);
    double acc = 0.0;
    for (int k = 0; k < ept; k++) {
        const int e = (bid * ept + k) * 256 + threadIdx.x;
        if (e >= Ea) break;
        double er[2], r0, r1;
        d_edge_error(cams[e_cam[e]].w2i, pts + 3 * (size_t)e_pt[e], e_meas + 2 * e, er);
        d_huber(er[0] * er[0] + er[1] * er[1], delta, &r0, &r1);
        err[2 * e] = er[0]; err[2 * e + 1] = er[1]; chi_e[e] = r0;
        acc += r0;
    }
    if (mine) *mine = acc;
}
__device__ __forceinline__ void ba_decide_fan(const ba_fan& F, int n_e, int n_u, const double* __restrict__ dxp, const double* __restrict__ bp, int n6,
                                              int phase, int cam_part, ba_state* __restrict__ S, ba_state* host_out, int seq);
__global__ __launch_bounds__(256) void k_ba_err(const ba_cam* __restrict__ cams, const double* __restrict__ pts,
                                                const int* __restrict__ e_cam, const int* __restrict__ e_pt,
                                                const double* __restrict__ e_meas, int Ea, double delta,
                                                double* __restrict__ err, double* __restrict__ chi_e,
                                                ba_fan F, int n_u, const double* __restrict__ dxp, const double* __restrict__ bp, int n6,
                                                int phase, int cam_part, ba_state* __restrict__ S, ba_state* host_out, int seq)
{
    __shared__ double s_f[4];
    double mine = 0;
    ba_err_body(blockIdx.x, cams, pts, e_cam, e_pt, e_meas, Ea, delta, err, chi_e, &mine);
    if (!F.e_chi) return;
    const double t = fan_block_sum256(mine, s_f);
    if (threadIdx.x == 0) fan_store(F.e_chi + blockIdx.x, t);
    if (fan_arrive(&S->tick[1], F.grp, (int)blockIdx.x, (int)gridDim.x)) ba_decide_fan(F, (int)gridDim.x, n_u, dxp, bp, n6, phase, cam_part, S, host_out, seq);
}

// ---- k_ba_decide: Levenberg gain ratio and lambda schedule (OptimizationAlgorithmLevenberg::solve)
__device__ __forceinline__ void ba_verdict(double chi_trial, double denom, double lambda, ba_state* __restrict__ S, ba_state* host_out, int seq)
{
    double tempChi = chi_trial;
    if (!S->chol_ok) tempChi = DBL_MAX;
    double rho = S->chi_cur - tempChi;
    const double scale = denom + 1e-3;
    rho /= scale;
    S->chi_trial = tempChi; S->scale = scale; S->rho = rho;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - d_cube(2 * rho - 1);             // g2o: 1. - pow((2*rho-1), 3)
        alpha = fmin(alpha, 2. / 3.);
        S->lambda = lambda * fmax(1. / 3., alpha); S->ni = 2; S->chi_cur = tempChi; S->accepted = 1;
    } else {
        S->lambda = lambda * S->ni; S->ni *= 2; S->accepted = 0;
    }
    S->qmax += 1;
    S->again = (rho < 0 && S->qmax < 10) ? 1 : 0;
    S->terminate = (!S->again && (S->qmax == 10 || rho == 0)) ? 1 : 0;
    S->chol_ok = 1;                                  // re-arm for the next trial
    if (host_out) {                                  // the host polls pinned memory for the verdict instead of waiting on a copy + stream sync
        ba_state t = *S; t.pad = 0;
        *host_out = t;
        __threadfence_system();
        __atomic_store_n(&host_out->pad, seq, __ATOMIC_RELEASE);
    }
}
__device__ __forceinline__ void ba_decide_body(const double* __restrict__ chi_e, int Ea, const double* __restrict__ dxp,
                                                    const double* __restrict__ bp, int n6, const double* __restrict__ dxl,
                                                    const double* __restrict__ bl, int n3, int phase, int cam_part,
                                                    const double* __restrict__ part, ba_state* __restrict__ S, ba_state* host_out, int seq)
{
    // phases as in k_ba_prep; cam_part: this rank contributes the (replicated) camera part of the gain-ratio denominator
    __shared__ double s_red[2 * 16];
    const int tid = threadIdx.x;
    const double lambda = S->lambda;
    double v[2] = { 0, 0 };
    if (phase != 2) {
        if (cam_part) for (int i = tid; i < n6; i += 1024) v[1] += dxp[i] * (lambda * dxp[i] + bp[i]);
        if (part) { if (tid < BA_NPART) { v[0] = part[2 * tid]; v[1] += part[2 * tid + 1]; } }
        else {
            for (int e = tid; e < Ea; e += 1024) v[0] += chi_e[e];
            for (int i = tid; i < n3; i += 1024) v[1] += dxl[i] * (lambda * dxl[i] + bl[i]);
        }
    }
    d_block_sum<2, 1024>(v, s_red);
    if (tid == 0) {
        if (phase == 1) { S->red[0] = v[0]; S->red[1] = v[1]; S->red[2] = S->chol_ok ? 0.0 : 1.0; return; }
        if (phase == 2) { v[0] = S->red[0]; v[1] = S->red[1]; if (S->red[2] > 0) S->chol_ok = 0; }
        ba_verdict(v[0], v[1], lambda, S, host_out, seq);
    }
}
// the fan-in form (single GPU): chi of the trial from the error kernel's slots, the landmark half of the denominator from the update's, the
// camera half summed here; one wave
// (phase 1, multi-GPU: this rank's sums go to S->red for the all-reduce hook, k_ba_decide's phase 2 takes the decision; cam_part: this rank
// contributes the replicated camera half of the denominator)
__device__ __forceinline__ void ba_decide_fan(const ba_fan& F, int n_e, int n_u, const double* __restrict__ dxp, const double* __restrict__ bp, int n6,
                                              int phase, int cam_part, ba_state* __restrict__ S, ba_state* host_out, int seq)
{
    __shared__ double s_w[4];
    const int lane = threadIdx.x;
    const double lambda = S->lambda;
    const double chi = fan_sum_wg(F.e_chi, n_e, s_w);
    double q = fan_sum_wg(F.u_sum, n_u, s_w);
    if (cam_part) {                                              // (the same rule for the camera half: one wave up to FAN_WAVE_MAX unknowns, the 256-thread workgroup beyond)
        double c = 0;
        if (n6 <= FAN_WAVE_MAX) {
            if (lane < 64) {
                for (int i = lane; i < n6; i += 64) c += dxp[i] * (lambda * dxp[i] + bp[i]);
                for (int m = 32; m > 0; m >>= 1) c += d_shfl_xor(c, m);
            }
        } else {
            for (int i = lane; i < n6; i += 256) c += dxp[i] * (lambda * dxp[i] + bp[i]);
            c = fan_block_sum256(c, s_w);
        }
        q += c;
    }
    if (lane == 0) {
        if (phase == 1) { S->red[0] = chi; S->red[1] = q; S->red[2] = S->chol_ok ? 0.0 : 1.0; return; }
        ba_verdict(chi, q, lambda, S, host_out, seq);
    }
}
__global__ __launch_bounds__(1024) void k_ba_decide(const double* __restrict__ chi_e, int Ea, const double* __restrict__ dxp,
                                                    const double* __restrict__ bp, int n6, const double* __restrict__ dxl,
                                                    const double* __restrict__ bl, int n3, int phase, int cam_part,
                                                    const double* __restrict__ part, ba_state* __restrict__ S, ba_state* host_out, int seq)
{
    ba_decide_body(chi_e, Ea, dxp, bp, n6, dxl, bl, n3, phase, cam_part, part, S, host_out, seq);
}

// ================================================================================================ batched local BA
// W independent windows (different sizes) advance in lockstep: one launch of every kernel per Levenberg trial covers all of them,
// gridded over (item, window); the window's arrays come from a descriptor (ba_view), its control flow - which state buffer is current,
// whether the next round starts a new iteration (linearisation) or retries with a larger lambda, when it is finished - lives in its
// ba_state on the device.  The kernels are the bodies of the single-window kernels, so a window's result is bitwise the one
// bh_ba_optimize gives.  slam/local_mapping.py:46-47 optimises one window at a time; SURVEY 8(e) "replicas only".
struct ba_view {
    ba_cam* cams[2]; double* pts[2];
    const int *e_cam, *e_pt; const double* e_meas; const int* cslot;
    const int *p_start, *p_edges, *c_start, *c_edges;
    double *err[2], *chi_e[2];
    double *Hpl, *Hll, *bl, *Hpp, *bp, *pdiag, *cdiag, *Dinv, *db, *S, *bs, *dxp, *dxl;
    const int2* pairs; const int* j_start; const int2* join;
    ba_state* state; ba_state* hstate;
    int C, Cf, Pa, Ea, npairs, bw, n6, pad;
    ba_K K;
    ba_fan fan;                   // the window's fan-in slots (the same sums, in the same order, as bh_ba_optimize)
};

// Which (window, block) a workgroup of a (blocks, windows-rounded-up-to-8) grid works on.  Consecutive workgroup ids go round the 8 XCDs,
// each with its own 4 MB L2; all workgroups of window w are sent to XCD w % 8, so a live window's Hpl blocks (0.7 MB, written by the
// linearisation, gathered ~36 times by the Schur products) stay in ONE L2 instead of being pulled into all eight: 64 windows x 0.7 MB
// thrashed every L2 and the Schur launch ran at HBM speed (87 us; the same windows one at a time: 12.5 us each).
__device__ __forceinline__ bool bb_where(int W, int& w, int& bid)
{
    if (gridDim.y == 1) { w = 0; bid = blockIdx.x; return true; }      // a batch of ONE window (bh_ba_optimize): its workgroups go round all XCDs
    const int nb = gridDim.x, lin = blockIdx.y * nb + blockIdx.x, v = lin >> 3, g = v / nb;
    w = 8 * g + (lin & 7); bid = v - g * nb;
    return w < W;
}

// The control steps of a round (round 5: run by the LAST workgroup of the window's linearisation / error launch to arrive, as in bh_ba_optimize's own
// kernels, instead of two more launches per round; every thread of that workgroup calls them).  seq = the round's tag for the pinned verdict slot.
__device__ __forceinline__ void bb_report(ba_state* S, ba_state* hout, int seq)
{
    ba_state t = *S; t.pad = 0;
    *hout = t;
    __threadfence_system();
    __atomic_store_n(&hout->pad, seq, __ATOMIC_RELEASE);
}
__device__ __forceinline__ void bb_prep_tail(const ba_view& V, int seq, bool report_fail)
{
    ba_state* S = V.state;
    const int first = S->b_it == 0 ? 1 : 0;
    ba_prep_fan(V.fan, (V.Pa * BP_G + 255) / 256, V.Cf, first, 0, S);
    if (threadIdx.x == 0) {
        S->b_last_err = 0;
        if (first && !isfinite(S->max_diag)) {                   // non-finite Hessian at the initial estimate
            S->b_done = 1; S->b_fail = 1;
            if (report_fail) { S->b_fail = 2; bb_report(S, V.hstate + (seq & 1), seq); }      // (no error launch of this window will run: tell the host now)
        }
    }
}
__device__ __forceinline__ void bb_decide_tail(const ba_view& V, int seq)
{
    ba_state* S = V.state;
    ba_decide_fan(V.fan, ba_err_blocks(V.Ea), (V.Pa * BP_G + 255) / 256, V.dxp, V.bp, V.n6, 0, 1, S, nullptr, 0);
    if (threadIdx.x == 0) {                                      // the control flow of bh_ba_optimize's host loop
        if (S->accepted) S->b_cur ^= 1;
        S->b_last_err = 1;                                       // the trial's errors are the last computeActiveErrors
        if (!S->again) {                                         // the iteration is over
            if (S->b_verbose && !S->accepted) S->b_last_err = 0; // verbose branch of SparseOptimizer::optimize: errors at the restored estimate
            S->b_it += 1; S->b_need_lin = 1;
            if (S->terminate || S->b_it >= S->b_iters) S->b_done = 1;
        } else S->b_need_lin = 0;
        bb_report(S, V.hstate + (seq & 1), seq);                 // the verdict slot of this round's parity (the host is at most one round behind)
    }
}

__global__ void k_bb_init(const ba_view* __restrict__ views, int W, int iters, int verbose)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= W) return;
    ba_state* S = views[w].state;
    S->tick[0] = 0; S->tick[1] = 0;
    {   // arrival counters of the window's fan-in (the last arrivers leave them at zero; a buffer that grew starts with whatever it held)
        const ba_view& V = views[w];
        const int ng = ((V.Pa * BP_G + 255) / 256 + BA_CAM_PARTS_MAX * V.Cf + ba_err_blocks(V.Ea) + 63) / 64 + 8;
        if (V.fan.grp) for (int i = 0; i < ng; i++) V.fan.grp[i] = 0;
    }
    S->b_it = 0; S->b_iters = iters; S->b_need_lin = 1; S->b_done = (views[w].Ea == 0 || iters <= 0) ? 1 : 0; S->b_last_err = 0; S->b_verbose = verbose; S->b_fail = 0;
}
__global__ void k_bb_setcur(const ba_view* __restrict__ views, const int* __restrict__ cur, int W)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w < W) views[w].state->b_cur = cur[w];
}

__global__ __launch_bounds__(256) void k_bb_linearize(const ba_view* __restrict__ views, int W, double delta, int seq)
{
    __shared__ double s_f[8];
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    ba_state* S = V.state;
    if (S->b_done || !S->b_need_lin) return;
    const int nbp = (V.Pa * BP_G + 255) / 256, cur = S->b_cur;
    double mine[2] = { 0, 0 }, md_cam = 0;
    if (bx < nbp) ba_point_body(bx, V.cams[cur], V.pts[cur], V.e_cam, V.e_meas, V.cslot, V.p_start, V.p_edges, V.Pa, V.K, delta, V.Hpl, V.Hll, V.bl,
                                             V.err[0], V.chi_e[0], V.pdiag, mine);
    const int cp = V.fan.cp > 1 ? V.fan.cp : 1, c = bx - nbp, cs = cp > 1 ? c / cp : c;
    bool fin = true;
    if (bx >= nbp + V.Cf * cp) return;
    if (bx >= nbp) fin = ba_cam_body(cs, V.cams[cur], V.pts[cur], V.e_cam, V.e_pt, V.e_meas, V.c_start, V.c_edges, V.K, delta, V.Hpp, V.bp, V.cdiag, nullptr, nullptr, &md_cam,
                                                     c - cs * cp, cp, V.fan.c_part, V.fan.c_cnt);
    ba_lin_fan_store(bx < nbp, bx < nbp ? bx : (fin ? cs : -1), mine, md_cam, V.fan, s_f);      // (read by the last arriver, or by k_bb_prep, the next launch)
    if (seq > 0 && fan_arrive(&S->tick[0], V.fan.grp, bx, nbp + V.Cf * cp)) bb_prep_tail(V, seq, true);
}
__global__ __launch_bounds__(1024) void k_bb_prep(const ba_view* __restrict__ views)
{
    const ba_view& V = views[blockIdx.y];
    ba_state* S = V.state;
    if (S->b_done || !S->b_need_lin) return;
    bb_prep_tail(V, 0, false);                                   // (BH_BA_BB_TAIL=0: the control steps as launches of their own; k_bb_decide reports a failure)
}
__global__ __launch_bounds__(256) void k_bb_dinv(const ba_view* __restrict__ views, int W)
{
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    if (V.state->b_done) return;
    const int pb = (int)gridDim.x - 8;                           // the last 8 workgroups of a window clear its reduced camera system (was k_bb_zeroS)
    if (bx >= pb) {
        const size_t n = (size_t)V.n6 * (V.bw + 1);
        for (size_t i = (size_t)(bx - pb) * 256 + threadIdx.x; i < n; i += 8 * 256) V.S[i] = 0.0;
        return;
    }
    if (bx * 256 >= V.Pa) return;
    ba_dinv_body(bx, V.Hll, V.bl, V.Pa, V.state, V.Dinv, V.db);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_bb_schur(const ba_view* __restrict__ views, int W)
{
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    if (V.state->b_done) return;
    const int ngS = (V.npairs + 7) / 8 * 8;
    if (bx < ngS) { if (!V.pad) ba_schur_S_body<256>(bx, ngS, V.pairs, V.j_start, V.join, V.e_pt, V.Hpl, V.Dinv, V.Hpp, V.state, V.bw, 1, V.S, V.npairs); }
    else if (bx < ngS + V.Cf) ba_schur_b_body(bx - ngS, V.c_start, V.c_edges, V.e_pt, V.Hpl, V.db, V.bp, 1, V.bs);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_bb_schur64(const ba_view* __restrict__ views, int W)
{
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    if (V.state->b_done || !V.pad) return;                       // (pad: the window's Schur products go one wave per camera pair)
    const int ngS = (V.npairs + 7) / 8 * 8;
    if (bx < ngS) ba_schur_S_body<64>(bx, ngS, V.pairs, V.j_start, V.join, V.e_pt, V.Hpl, V.Dinv, V.Hpp, V.state, V.bw, 1, V.S, V.npairs);
}
__global__ __launch_bounds__(1024) void k_bb_chol(const ba_view* __restrict__ views)
{
    const ba_view& V = views[blockIdx.x];
    if (V.state->b_done || V.Cf == 0) return;
    ba_chol_dense_body<false>(V.S, V.bs, V.n6, V.bw, V.dxp, V.state, nullptr, nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(256) void k_bb_update(const ba_view* __restrict__ views, int W)
{
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    const ba_state* S = V.state;
    const int pblocks = (V.Pa * BP_G + 255) / 256, cblocks = (V.C + 255) / 256;
    if (S->b_done || bx >= pblocks + cblocks) return;
    const int cur = S->b_cur;
    __shared__ double s_f[4];
    double mine = 0;
    ba_update_body(bx, V.cams[cur], V.cams[cur ^ 1], V.pts[cur], V.pts[cur ^ 1], V.cslot, V.C, V.Pa, pblocks, V.p_start, V.p_edges, V.e_cam, V.Hpl, V.Dinv, V.bl, V.dxp, V.dxl, V.K,
                   S->lambda, &mine);
    if (bx >= pblocks) return;
    const double t = fan_block_sum256(mine, s_f);
    if (threadIdx.x == 0) fan_store(V.fan.u_sum + bx, t);
}
__global__ __launch_bounds__(256) void k_bb_err(const ba_view* __restrict__ views, int W, double delta, int seq)
{
    int w, bx;
    if (!bb_where(W, w, bx)) return;
    const ba_view& V = views[w];
    ba_state* S = V.state;
    if (S->b_done || bx >= ba_err_blocks(V.Ea)) return;
    const int cur = S->b_cur;
    __shared__ double s_f[4];
    double mine = 0;
    ba_err_body(bx, V.cams[cur ^ 1], V.pts[cur ^ 1], V.e_cam, V.e_pt, V.e_meas, V.Ea, delta, V.err[1], V.chi_e[1], &mine);
    const double t = fan_block_sum256(mine, s_f);
    if (threadIdx.x == 0) fan_store(V.fan.e_chi + bx, t);
    if (seq > 0 && fan_arrive(&S->tick[1], V.fan.grp, bx, ba_err_blocks(V.Ea))) bb_decide_tail(V, seq);
}
__global__ __launch_bounds__(1024) void k_bb_decide(const ba_view* __restrict__ views, int seq)
{
    const ba_view& V = views[blockIdx.y];
    ba_state* S = V.state;
    ba_state* const hout = V.hstate + (seq & 1);             // the verdict slot of this round's parity (the host is at most one round behind)
    if (S->b_done) {
        if (threadIdx.x == 0 && S->b_fail == 1) {                // a window that failed in k_bb_prep: tell the host once
            S->b_fail = 2;
            ba_state t = *S; t.pad = 0;
            *hout = t;
            __threadfence_system();
            __atomic_store_n(&hout->pad, seq, __ATOMIC_RELEASE);
        }
        return;
    }
    bb_decide_tail(V, seq);
}

// multi-GPU: per-camera max |diag| of the all-reduced Hpp
__global__ void k_ba_cdiag(const double* __restrict__ Hpp, int Cf, double* __restrict__ cdiag)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= Cf) return;
    double md = 0;
    for (int a = 0; a < 6; a++) md = fmax(md, fabs(Hpp[36 * (size_t)s + 7 * a]));
    cdiag[s] = md;
}

__global__ void k_ba_cam_init(ba_cam* cams, int C, ba_K K)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < C) d_cam_refresh(&cams[i], K);
}

__global__ void k_ba_export(const ba_cam* cams, int C, double* R, double* t)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= C) return;
    double Rm[9]; d_quat_to_R(cams[i].q, Rm);
    for (int k = 0; k < 9; k++) R[9 * (size_t)i + k] = Rm[k];
    for (int k = 0; k < 3; k++) t[3 * (size_t)i + k] = cams[i].t[k];
}

// ------------------------------------------------------------------------------------------------ host side
static void h_quat_from_R(const double* m, double* q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t; q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
    if (q[3] < 0) for (int a = 0; a < 4; a++) q[a] = -q[a];                 // SE3Quat::normalizeRotation
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= n;
}

static void ba_builder_attr();
extern "C" int bh_ba_create(bh_ctx* ctx, const double K[5], bh_ba** out)
{
    if (!ctx || !K || !out) return BH_ERR_INVALID;
    (void)hipSetDevice(ctx->device);                         // the attributes below belong to the context's device
    (void)hipFuncSetAttribute((const void*)k_ba_chol<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_ba_chol<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_ba_chol<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)k_ba_chol_dense<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CD_LDS_DOUBLES));
    (void)hipFuncSetAttribute((const void*)k_ba_chol_dense<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CD_LDS_DOUBLES));
    (void)hipFuncSetAttribute((const void*)k_ba_chol_sw, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    ba_builder_attr();                                       // (k_bld_small: > 64 KB of dynamic LDS)
    bh_ba* ba = new bh_ba();
    ba->ctx = ctx; ba->K = { K[0], K[1], K[2], K[3] };
    (void)hipSetDevice(ctx->device);
    if (hipHostMalloc((void**)&ba->h_state, sizeof(ba_state), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { ba->h_state = nullptr; (void)hipGetLastError(); }
    else memset(ba->h_state, 0, sizeof(ba_state));
    *out = ba;
    return BH_OK;
}

extern "C" void bh_ba_destroy(bh_ba* ba)
{
    if (!ba) return;
    dbcr_release(ba->dsol); ba->dsol = nullptr;
    (void)hipSetDevice(ba->ctx->device);
    (void)hipStreamSynchronize(ba->ctx->stream);
    bh_buf* all[] = { &ba->cams[0], &ba->cams[1], &ba->pts[0], &ba->pts[1], &ba->e_cam, &ba->e_pt, &ba->e_meas, &ba->cslot,
                      &ba->p_start, &ba->p_edges, &ba->c_start, &ba->c_edges, &ba->err[0], &ba->err[1], &ba->chi_e[0],
                      &ba->chi_e[1], &ba->Hpl, &ba->Hll, &ba->bl, &ba->Hpp, &ba->bp, &ba->pdiag, &ba->cdiag, &ba->Dinv,
                      &ba->db, &ba->S, &ba->bs, &ba->dxp, &ba->dxl, &ba->state, &ba->pairs, &ba->part, &ba->xch, &ba->fan, &ba->join, &ba->j_start, &ba->pair_rec, &ba->join_pt, &ba->row_start, &ba->join_slot, &ba->c_pt, &ba->c_meas };
    for (bh_buf* b : all) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (ba->h_state) (void)hipHostFree(ba->h_state);
    if (ba->self) bh_ba_batch_destroy(ba->self);
    delete ba->bcr;
    delete ba;
}

extern "C" int bh_ba_set_allreduce(bh_ba* ba, bh_allreduce_fn fn, void* user, int rank, int world)
{
    if (!ba) return BH_ERR_INVALID;
    if (fn && (world < 1 || rank < 0 || rank >= world)) return bh_set_err(ba->ctx, BH_ERR_INVALID, "bh_ba_set_allreduce: bad rank/world");
    ba->ar = fn; ba->ar_user = user; ba->rank = fn ? rank : 0; ba->world = fn ? world : 1;
    ba->have_problem = 0;                                  // the camera slots depend on the mode
    return BH_OK;
}

extern "C" int bh_ba_set_allreduce_ordered(bh_ba* ba, int ordered)
{
    if (!ba) return BH_ERR_INVALID;
    ba->ar_ordered = ordered ? 1 : 0;
    return BH_OK;
}

extern "C" int bh_ba_get_comm(bh_ba* ba, double* ms, long long* bytes)
{
    if (!ba) return BH_ERR_INVALID;
    if (ms) *ms = ba->comm_ms;
    if (bytes) *bytes = ba->comm_bytes;
    return BH_OK;
}

// stream idle -> hook -> reduced values visible to later work on the stream
static int ba_allreduce(bh_ba* ba, void* dev, long long count, int op)
{
    bh_ctx* ctx = ba->ctx;
    if (!ba->ar_ordered) BH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (stream-ordered hooks enqueue behind the kernels instead: bh_ba_set_allreduce_ordered)
    const auto t0 = std::chrono::steady_clock::now();
    const int r = ba->ar(ba->ar_user, dev, count, op);
    ba->comm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // bytes a rank RECEIVES: all-reduce = the buffer; reduce-scatter (op 2) = its chunk; all-gather (op 3) = the other ranks' chunks
    const long long rb = op == 2 ? 8 * count / ba->world : (op == 3 ? 8 * count / ba->world * (ba->world - 1) : 8 * count);
    ba->comm_bytes += rb;
    if (op < 2) ba->ar_bytes += rb;
    else if (ba->dsol) { if (op == 2) ba->dsol->bytes_rs += rb; else ba->dsol->bytes_ag += rb; }
    if (r != 0) return bh_set_err(ctx, BH_ERR_INVALID, "global BA: the all-reduce hook failed (%d)", r);
    return BH_OK;
}


static void dbcr_release(dbcr_dev* d)
{
    if (!d) return;
    for (bh_buf* b : { &d->buf_loc, &d->buf_glob, &d->gather, &d->orient_d }) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (d->e0) (void)hipEventDestroy(d->e0);
    if (d->e1) (void)hipEventDestroy(d->e1);
    delete d;
}

// decide whether the sharded solve applies (every rank takes the same decision from n, bw and the world size) and set it up
static int dbcr_setup(bh_ba* ba, int n6, int bw, long long* rows_pad)
{
    bh_ctx* ctx = ba->ctx;
    *rows_pad = n6;
    if (ba->dsol) ba->dsol->active = 0;
    const char* ev = getenv("BH_BA_DSOLVE");
    if (!ba->ar || ba->world < 2 || !ba->chol_bcr || (ev && !strcmp(ev, "0"))) return BH_OK;
    const int N = (n6 + 255) / 256, R = ba->world, seg = (N + R - 1) / R;
    if (seg < 2 || N - (R - 1) * seg < 2) return BH_OK;       // a rank would hold fewer than two blocks: the replicated solve stays
    if (!ba->dsol) ba->dsol = new dbcr_dev();
    dbcr_dev* d = ba->dsol;
    d->seg = seg; d->B0 = ba->rank * seg; d->nb = std::min(N, (ba->rank + 1) * seg) - d->B0;
    dbcr_make_plan(d->nb, &d->loc.plan, &d->orient, &d->fslot);
    bcr_make_plan(2 * R, &d->glob.plan);
    hipStream_t st = ctx->stream;
    BH_TRY(bh_reserve(ctx, &d->buf_loc, BCR_SCRATCH_BYTES(d->nb) + 4096));
    BH_TRY(bh_reserve(ctx, &d->buf_glob, BCR_SCRATCH_BYTES(2 * R) + sizeof(double) * 512 * (size_t)R + 4096));
    BH_TRY(bh_reserve(ctx, &d->gather, sizeof(double) * (size_t)R * DBCR_CH));
    BH_TRY(bh_reserve(ctx, &d->orient_d, d->orient.size() + 16));
    BH_TRY(dbcr_carve(ctx, st, &d->loc, d->nb, d->buf_loc.p));
    BH_TRY(dbcr_carve(ctx, st, &d->glob, 2 * R, d->buf_glob.p));
    BH_HIP(ctx, hipMemcpyAsync(d->orient_d.p, d->orient.data(), d->orient.size(), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemsetAsync(d->gather.p, 0, sizeof(double) * (size_t)R * DBCR_CH, st));
    if (!d->e0) { BH_HIP(ctx, hipEventCreate(&d->e0)); BH_HIP(ctx, hipEventCreate(&d->e1)); }
    BH_HIP(ctx, hipStreamSynchronize(st));
    d->active = 1;
    *rows_pad = 256LL * seg * R;
    return BH_OK;
}

static int ba_allreduce(bh_ba* ba, void* dev, long long count, int op);
// S: this rank's band rows are summed (after the reduce-scatter); bs: complete on every rank.  x receives the whole step.
static int dbcr_solve(bh_ba* ba, const double* S, const double* bs, int n, int bw, double* x, ba_state* state)
{
    bh_ctx* ctx = ba->ctx;
    hipStream_t st = ctx->stream;
    dbcr_dev* d = ba->dsol;
    const int R = ba->world, nb = d->nb;
    double* gat = (double*)d->gather.p; double* mine = gat + (size_t)ba->rank * DBCR_CH;
    double* xg = (double*)((char*)d->buf_glob.p + BCR_SCRATCH_BYTES(2 * R));
    double* xloc = x + 256 * (size_t)d->B0;
    const int nloc = std::max(0, std::min(n - 256 * d->B0, 256 * nb));
    hipLaunchKernelGGL(k_dbcr_extract, dim3(2 * nb, 256), dim3(256), 0, st, S, bs, n, bw, d->B0, nb, (const uint8_t*)d->orient_d.p, d->loc.D, d->loc.C, d->loc.g,
                       mine + 3 * (size_t)BCR_BLK);
    bcr_forward(st, &d->loc, state);                          // the interior of the segment goes; its Schur complement lands on the two end blocks
    hipLaunchKernelGGL(k_dbcr_pack, dim3(256, 3), dim3(256), 0, st, (const double*)d->loc.D, (const double*)d->loc.C, (const double*)d->loc.g, nb, d->fslot, mine);
    BH_HIP(ctx, hipGetLastError());
    BH_TRY(ba_allreduce(ba, gat, (long long)R * DBCR_CH, 3));
    BH_HIP(ctx, hipEventRecord(d->e0, st));                   // the replicated part: the chain of 2R end blocks
    hipLaunchKernelGGL(k_dbcr_unpack, dim3(256, 4, R), dim3(256), 0, st, (const double*)gat, R, d->glob.D, d->glob.C, d->glob.g);
    bcr_forward(st, &d->glob, state);
    bcr_backward(st, &d->glob, xg, 512 * R, state);
    BH_HIP(ctx, hipEventRecord(d->e1, st));
    hipLaunchKernelGGL(k_dbcr_ends, dim3(2), dim3(256), 0, st, (const double*)d->glob.xp, ba->rank, nb, d->loc.xp, xloc, nloc);
    bcr_backward(st, &d->loc, xloc, nloc, state);
    BH_HIP(ctx, hipGetLastError());
    BH_TRY(ba_allreduce(ba, x, 256LL * d->seg * R, 3));      // every rank's rows of the step -> everyone
    float ms = 0;
    if (hipEventElapsedTime(&ms, d->e0, d->e1) == hipSuccess) { d->ms_replicated += ms; d->n_replicated++; } else (void)hipGetLastError();
    return BH_OK;
}

extern "C" int bh_ba_get_comm_detail(bh_ba* ba, double* out)
{
    if (!ba || !out) return BH_ERR_INVALID;
    const dbcr_dev* d = ba->dsol;
    out[0] = (double)ba->ar_bytes; out[1] = d ? (double)d->bytes_rs : 0.0; out[2] = d ? (double)d->bytes_ag : 0.0;
    out[3] = d ? d->ms_replicated : 0.0; out[4] = d ? (double)d->n_replicated : 0.0; out[5] = d && d->active ? 1.0 : 0.0;
    out[6] = d && d->active ? (double)d->nb : 0.0; out[7] = d && d->active ? (double)d->seg : 0.0;
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ problem construction on the device
// The structure g2o builds in initializeOptimization - which vertices take part, the edges by point and by free camera, the camera
// pairs that share a landmark (the blocks of the reduced camera system) - used to be computed by host loops (1.3 ms at config 3,
// 0.11 s at config 5, untimed).  It is now a handful of kernels on the context's stream with the same outputs bit for bit - counting,
// exclusive scans (bh_scan.h) and RANKING, no sort (rounds 2 - 3 used three library radix sorts here; round 4 builds every list the way
// the one-workgroup builder does, for any size):
//   slots      cam_slot = rank among the used, non-fixed cameras; pt_slot = rank among the points that have an edge (exclusive scans)
//   p_edges    per-point lists: p_start by counting + scan, filled in arbitrary order (integer atomics), then ranked inside the point by
//              (camera, input position) - the keys are unique, the order is total, nothing depends on the fill order
//   c_edges    the p_edges sequence filtered by free camera: positions filled per camera in arbitrary order, then ranked by position
//              (a workgroup per camera, keys through LDS); c_pt = landmark slot of every entry
//   join       per camera pair (counted per landmark with the pair's index = rank of its bit in the bitmap): camera s1's list looks its
//              landmarks up in s2's (both sorted by landmark) and writes the hits in order - the kernel of the small builder
//   pairs      camera pairs sharing a landmark: one thread per point marks its pairs in a Cf x Cf bitmap (integer atomicOr), the
//              diagonal is always set; compacted in lexicographic order by a scan over the bitmap's popcounts
__global__ void k_bld_mark(const int* __restrict__ ep, const int* __restrict__ pp, int E, int C, int P, int* __restrict__ cam_used,
                           int* __restrict__ pt_used, int* __restrict__ info)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int c = ep[e], q = pp[e];
    if (c < 0 || c >= C || q < 0 || q >= P) { atomicMin(&info[0], e); return; }      // info[0]: first edge that references a missing vertex
    cam_used[c] = 1; pt_used[q] = 1;
}
__global__ void k_bld_camflag(const int* __restrict__ cam_used, const uint8_t* __restrict__ fixed, int C, int all_used, int* __restrict__ flag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C) flag[i] = ((all_used || cam_used[i]) && !fixed[i]) ? 1 : 0;
}
__global__ void k_bld_slot(const int* __restrict__ flag, const int* __restrict__ scan, int n, int* __restrict__ slot, int* __restrict__ total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) slot[i] = flag[i] ? scan[i] : -1;
    if (i == n - 1) *total = scan[i] + flag[i];
}
#define BLD_CC 1024
__global__ void k_bld_edges(const int* __restrict__ ep, const int* __restrict__ pp, int E, const int* __restrict__ pt_slot,
                            const int* __restrict__ cam_slot, int* __restrict__ e_pt, unsigned long long* __restrict__ key, int* __restrict__ idx,
                            int* __restrict__ p_cnt, int* __restrict__ c_cnt, int Cf)
{
    // per-camera counts through a workgroup-private LDS histogram when the free cameras fit (a local window: 48 counters would take
    // all 80 k global atomics), straight to global otherwise
    __shared__ int s_cc[BLD_CC];
    const bool priv = Cf <= BLD_CC;
    if (priv) for (int i = threadIdx.x; i < Cf; i += 256) s_cc[i] = 0;
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < E) {
        const int c = ep[e], j = pt_slot[pp[e]];
        e_pt[e] = j;
        if (key) { key[e] = ((unsigned long long)(unsigned)j << 32) | (unsigned)c; idx[e] = e; }
        atomicAdd(&p_cnt[j], 1);
        const int sl = cam_slot[c];
        if (sl >= 0) { if (priv) atomicAdd(&s_cc[sl], 1); else atomicAdd(&c_cnt[sl], 1); }
    }
    __syncthreads();
    if (priv) for (int i = threadIdx.x; i < Cf; i += 256) { const int v = s_cc[i]; if (v) atomicAdd(&c_cnt[i], v); }
}
__global__ void k_bld_pairs_mark(const int* __restrict__ p_start, const int* __restrict__ p_edges, const int* __restrict__ ep,
                                 const int* __restrict__ cam_slot, int Pa, int Cf, unsigned* __restrict__ bits, int* __restrict__ info)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    int bw = 0;
    if (j < Pa) {
        const int a0 = p_start[j], a1 = p_start[j + 1];
        for (int u = a0; u < a1; u++) {
            const int s1 = cam_slot[ep[p_edges[u]]]; if (s1 < 0) continue;
            for (int v = u; v < a1; v++) {
                const int s2 = cam_slot[ep[p_edges[v]]]; if (s2 < 0) continue;
                const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
                const unsigned long long bit = (unsigned long long)lo * Cf + hi;
                // test before set: in a dense window every point marks the same few words (48 x 48 bits: 280 k atomics on 72 words
                // took 0.65 ms of a 1.7 ms set-up); a stale read only costs one redundant atomic
                const unsigned m = 1u << (bit & 31);
                if (!(__atomic_load_n(&bits[bit >> 5], __ATOMIC_RELAXED) & m)) atomicOr(&bits[bit >> 5], m);
                bw = max(bw, hi - lo);
            }
        }
    } else if (j - Pa < Cf) {                                                     // the diagonal blocks exist whatever the edges say
        const unsigned long long bit = (unsigned long long)(j - Pa) * Cf + (j - Pa);
        atomicOr(&bits[bit >> 5], 1u << (bit & 31));
    }
    for (int mm = 32; mm > 0; mm >>= 1) bw = max(bw, __shfl_xor(bw, mm, 64));      // one atomic per wave
    if (bw && (threadIdx.x & 63) == 0) atomicMax(&info[1], bw);                   // info[1]: max |s1 - s2| over the pairs
}
__global__ void k_bld_popc(const unsigned* __restrict__ bits, long long nw, int* __restrict__ cnt)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w < nw) cnt[w] = __popc(bits[w]);
}
__global__ void k_bld_pairs_emit(const unsigned* __restrict__ bits, const int* __restrict__ off, long long nw, int Cf, int2* __restrict__ pairs)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    unsigned m = bits[w];
    int o = off[w];
    while (m) {
        const int b = __ffs(m) - 1; m &= m - 1;
        const unsigned long long bit = (unsigned long long)w * 32 + b;
        pairs[o++] = make_int2((int)(bit / Cf), (int)(bit % Cf));
    }
}
__global__ void k_bld_gather_pts(const double* __restrict__ points, const int* __restrict__ pt_slot, int P, double* __restrict__ pts)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const int s = pt_slot[j];
    if (s >= 0) { pts[3 * (size_t)s] = points[3 * (size_t)j]; pts[3 * (size_t)s + 1] = points[3 * (size_t)j + 1]; pts[3 * (size_t)s + 2] = points[3 * (size_t)j + 2]; }
}

// ---- the general builder's own kernels
__global__ void k_g_pfill(const int* __restrict__ e_pt, int E, const int* __restrict__ p_start, int* __restrict__ pfill, int* __restrict__ tmp)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int j = e_pt[e];
    tmp[p_start[j] + atomicAdd(&pfill[j], 1)] = e;
}
// rank inside the point by (camera, input position).  Lane per point for the usual short lists; a point with more than 32 edges is
// ranked by its whole wave afterwards (lane-strided entries, every lane walks the list).
__global__ __launch_bounds__(256) void k_g_prank(const int* __restrict__ p_start, const int* __restrict__ tmp, const int* __restrict__ ep, int Pa,
                                                 int* __restrict__ p_edges)
{
    const int j = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    int a0 = 0, k = 0;
    if (j < Pa) { a0 = p_start[j]; k = p_start[j + 1] - a0; }
    if (k <= 32)
        for (int i = 0; i < k; i++) {
            const int ei = tmp[a0 + i], ci = ep[ei];
            int r = 0;
            for (int i2 = 0; i2 < k; i2++) { const int ex = tmp[a0 + i2], c2 = ep[ex]; r += (c2 < ci || (c2 == ci && ex < ei)) ? 1 : 0; }
            p_edges[a0 + r] = ei;
        }
    unsigned long long heavy = __ballot(k > 32);
    while (heavy) {
        const int src = __ffsll((long long)heavy) - 1; heavy &= heavy - 1;
        const int b0 = __shfl(a0, src, 64), kk = __shfl(k, src, 64);
        for (int i = lane; i < kk; i += 64) {
            const int ei = tmp[b0 + i], ci = ep[ei];
            int r = 0;
            for (int i2 = 0; i2 < kk; i2++) { const int ex = tmp[b0 + i2], c2 = ep[ex]; r += (c2 < ci || (c2 == ci && ex < ei)) ? 1 : 0; }
            p_edges[b0 + r] = ei;
        }
    }
}
__global__ void k_g_cfill(const int* __restrict__ p_edges, const int* __restrict__ ep, const int* __restrict__ cam_slot, int E, const int* __restrict__ c_start,
                          int* __restrict__ cfill, int* __restrict__ ctmp)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= E) return;
    const int sl = cam_slot[ep[p_edges[k]]];
    if (sl >= 0) ctmp[c_start[sl] + atomicAdd(&cfill[sl], 1)] = k;
}
// a workgroup per free camera: its positions in the p_edges sequence (unique keys) are ranked against all of them, 2 048 keys at a time through
// LDS (four keys per broadcast read).  NQ = keys a thread ranks per pass: 2 / 4 / 8 for lists up to 512 / 1 024 / any length.
#define G_CT 2048
template <int NQ>
__device__ __forceinline__ void g_crank_body(int* s_k, const int a0, const int n, const int* __restrict__ ctmp, const int* __restrict__ p_edges,
                                             const int* __restrict__ e_pt, int* __restrict__ c_edges, int* __restrict__ c_pt)
{
    const int tid = threadIdx.x;
    for (int base = 0; base < n; base += 256 * NQ) {
        int kk[NQ], r[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int i = base + tid + 256 * q; kk[q] = i < n ? ctmp[a0 + i] : 0x7fffffff; r[q] = 0; }
        for (int tb = 0; tb < n; tb += G_CT) {
            const int m = min(G_CT, n - tb), m4 = (m + 3) & ~3;
            __syncthreads();
            for (int i = tid; i < m4; i += 256) s_k[i] = i < m ? ctmp[a0 + tb + i] : 0x7fffffff;      // (padding keys rank nobody)
            __syncthreads();
            for (int t = 0; t < m4; t += 4) {
                const int4 v = *reinterpret_cast<const int4*>(s_k + t);
#pragma unroll
                for (int q = 0; q < NQ; q++) r[q] += (v.x < kk[q] ? 1 : 0) + (v.y < kk[q] ? 1 : 0) + (v.z < kk[q] ? 1 : 0) + (v.w < kk[q] ? 1 : 0);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int i = base + tid + 256 * q;
            if (i < n) { const int e = p_edges[kk[q]]; c_edges[a0 + r[q]] = e; c_pt[a0 + r[q]] = e_pt[e]; }
        }
    }
}
__global__ __launch_bounds__(256) void k_g_crank(const int* __restrict__ c_start, const int* __restrict__ ctmp, const int* __restrict__ p_edges,
                                                 const int* __restrict__ e_pt, int* __restrict__ c_edges, int* __restrict__ c_pt)
{
    __shared__ __attribute__((aligned(16))) int s_k[G_CT];
    const int s = blockIdx.x, a0 = c_start[s], n = c_start[s + 1] - a0;
    if (n <= 512) g_crank_body<2>(s_k, a0, n, ctmp, p_edges, e_pt, c_edges, c_pt);
    else if (n <= 1024) g_crank_body<4>(s_k, a0, n, ctmp, p_edges, e_pt, c_edges, c_pt);
    else g_crank_body<8>(s_k, a0, n, ctmp, p_edges, e_pt, c_edges, c_pt);
}
// ---- camera pairs and their join counts in ONE pass over the landmarks when the band is narrow: counters cnt[lo][hi - lo] (Cf x (band + 1) ints,
// the band = largest slot distance inside a landmark, found first) instead of a Cf x Cf bitmap, its ranks and a second pass for the counts
__global__ void k_g_band(const int* __restrict__ p_start, const int* __restrict__ p_edges, const int* __restrict__ ep, const int* __restrict__ cam_slot, int Pa,
                         int* __restrict__ info)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    int bw = 0;
    if (j < Pa) {
        int lo = 0x7fffffff, hi = -1;                            // edges of a point are sorted by camera, slots grow with the camera index
        for (int u = p_start[j]; u < p_start[j + 1]; u++) { const int sl = cam_slot[ep[p_edges[u]]]; if (sl >= 0) { lo = min(lo, sl); hi = max(hi, sl); } }
        if (hi >= 0) bw = hi - lo;
    }
    for (int mm = 32; mm > 0; mm >>= 1) bw = max(bw, __shfl_xor(bw, mm, 64));
    if (bw && (threadIdx.x & 63) == 0) atomicMax(&info[1], bw);
}
__global__ void k_g_bandcount(const int* __restrict__ p_start, const int* __restrict__ p_edges, const int* __restrict__ ep, const int* __restrict__ cam_slot,
                              int Pa, int W, int* __restrict__ cnt, unsigned long long* __restrict__ total)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (j < Pa) {
        const int a0 = p_start[j], a1 = p_start[j + 1];
        for (int u = a0; u < a1; u++) {
            const int s1 = cam_slot[ep[p_edges[u]]]; if (s1 < 0) continue;
            for (int v = u; v < a1; v++) {
                const int s2 = cam_slot[ep[p_edges[v]]]; if (s2 < 0) continue;
                const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
                atomicAdd(&cnt[(size_t)lo * W + (hi - lo)], 1);
                mine++;
            }
        }
    }
    for (int mm = 32; mm > 0; mm >>= 1) mine += __shfl_xor(mine, mm, 64);
    if (mine && (threadIdx.x & 63) == 0) atomicAdd(total, mine);
}
__global__ void k_g_bandflag(const int* __restrict__ cnt, long long n, int W, int Cf, int* __restrict__ flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { flag[i] = 0; return; }
    const int lo = (int)(i / W), d = (int)(i - (long long)lo * W);
    flag[i] = (lo + d < Cf && (cnt[i] > 0 || d == 0)) ? 1 : 0;  // the diagonal blocks exist whatever the edges say
}
__global__ void k_g_bandemit(const int* __restrict__ cnt, const int* __restrict__ flag, const int* __restrict__ pos, long long n, int W, int2* __restrict__ pairs,
                             int* __restrict__ j_cnt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const int lo = (int)(i / W), d = (int)(i - (long long)lo * W);
    pairs[pos[i]] = make_int2(lo, lo + d); j_cnt[pos[i]] = cnt[i];
}
// join entries per camera pair: one thread per landmark adds its (u <= v) pairs to the pair's counter; the pair's index is the rank of
// its bit in the bitmap (pairs are emitted in bit order)
__global__ void k_g_jcount(const int* __restrict__ p_start, const int* __restrict__ p_edges, const int* __restrict__ ep, const int* __restrict__ cam_slot,
                           int Pa, int Cf, const unsigned* __restrict__ bits, const int* __restrict__ woff, int* __restrict__ j_cnt,
                           unsigned long long* __restrict__ total)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (j < Pa) {
        const int a0 = p_start[j], a1 = p_start[j + 1];
        for (int u = a0; u < a1; u++) {
            const int s1 = cam_slot[ep[p_edges[u]]]; if (s1 < 0) continue;
            for (int v = u; v < a1; v++) {
                const int s2 = cam_slot[ep[p_edges[v]]]; if (s2 < 0) continue;
                const int lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
                const unsigned long long bit = (unsigned long long)lo * Cf + hi;
                const unsigned w = bits[bit >> 5];
                atomicAdd(&j_cnt[woff[bit >> 5] + __popc(w & ((1u << (bit & 31)) - 1u))], 1);
                mine++;
            }
        }
    }
    for (int mm = 32; mm > 0; mm >>= 1) mine += __shfl_xor(mine, mm, 64);
    if (mine && (threadIdx.x & 63) == 0) atomicAdd(total, mine);
}
__global__ __launch_bounds__(256) void k_g_join_merge(const int2* __restrict__ pairs, int npairs, const int* __restrict__ c_start, const int* __restrict__ c_edges,
                                                      const int* __restrict__ c_pt, const int* __restrict__ j_start, int2* __restrict__ join);

struct ba_build_out { int Cf, Pa, c_total, npairs, bwb; long long M; };

// Builds every structural array of the problem in ba's device buffers from the uploaded edge list.  scratch: ctx->st_a .. st_g.
static int ba_build_device(bh_ba* ba, int C, const uint8_t* fixed, int P, const double* points, int E, const int* edge_pose,
                           const int* edge_point, ba_build_out* out)
{
    bh_ctx* ctx = ba->ctx;
    hipStream_t st = ctx->stream;
    const int Ee = E > 0 ? E : 1, Ce = C > 0 ? C : 1, Pe = P > 0 ? P : 1;
    // scratch layout (ints unless noted): ep[E] pp[E] tmp[E] ctmp[E] | cam_used[C] camflag[C] camscan[C] | pt_used[P] ptscan[P] | info[8] | fixed u8[C]
    BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(int) * (4 * (size_t)Ee + 3 * (size_t)Ce + 2 * (size_t)Pe + 8) + Ce + 64));
    int* ep = (int*)ctx->st_a.p; int* pp = ep + Ee; int* tmp = pp + Ee; int* ctmp = tmp + Ee;
    int* cam_used = ctmp + Ee; int* camflag = cam_used + Ce; int* camscan = camflag + Ce;
    int* pt_used = camscan + Ce; int* ptscan = pt_used + Pe; int* info = ptscan + Pe; uint8_t* d_fixed = (uint8_t*)(info + 8);
    BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(double) * 3 * (size_t)Pe));        // user points
    BH_TRY(bh_reserve(ctx, &ba->cslot, sizeof(int) * (size_t)Ce));
    BH_TRY(bh_reserve(ctx, &ba->e_cam, sizeof(int) * (size_t)Ee)); BH_TRY(bh_reserve(ctx, &ba->e_pt, sizeof(int) * (size_t)Ee));
    BH_TRY(bh_reserve(ctx, &ba->p_edges, sizeof(int) * (size_t)Ee)); BH_TRY(bh_reserve(ctx, &ba->c_edges, sizeof(int) * (size_t)Ee));
    BH_TRY(bh_reserve(ctx, &ba->c_pt, sizeof(int) * (size_t)Ee));
    BH_TRY(bh_reserve(ctx, &ba->p_start, sizeof(int) * ((size_t)Pe + 2))); BH_TRY(bh_reserve(ctx, &ba->c_start, sizeof(int) * ((size_t)Ce + 2)));
    // pt_slot[P] p_cnt[P + 1] c_cnt[C + 1] pfill[P] cfill[C]
    BH_TRY(bh_reserve(ctx, &ctx->st_d, sizeof(int) * (3 * (size_t)Pe + 2 * (size_t)Ce + 8)));
    int* pt_slot = (int*)ctx->st_d.p; int* p_cnt = pt_slot + Pe; int* c_cnt = p_cnt + Pe + 1; int* pfill = c_cnt + Ce + 1; int* cfill = pfill + Pe;
    int* cam_slot = (int*)ba->cslot.p;
    const long long nbits0 = (long long)Ce * Ce, nw0 = (nbits0 + 31) / 32;
    BH_TRY(bh_reserve(ctx, &ctx->st_e, bh_exscan_scratch<int>(std::max<long long>(std::max<long long>(Pe, Ce), nw0) + 2) + 256));
    void* scan_tmp = ctx->st_e.p;
    const int hinfo[8] = { 0x7fffffff, 0, 0, 0, 0, 0, 0, 0 };
    BH_HIP(ctx, hipMemcpyAsync(info, hinfo, sizeof(hinfo), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemsetAsync(cam_used, 0, sizeof(int) * (size_t)Ce, st)); BH_HIP(ctx, hipMemsetAsync(pt_used, 0, sizeof(int) * (size_t)Pe, st));
    BH_HIP(ctx, hipMemsetAsync(p_cnt, 0, sizeof(int) * (2 * (size_t)Pe + 2 * (size_t)Ce + 2), st));      // p_cnt, c_cnt, pfill, cfill
    if (E) { BH_HIP(ctx, hipMemcpyAsync(ep, edge_pose, sizeof(int) * (size_t)E, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(pp, edge_point, sizeof(int) * (size_t)E, hipMemcpyHostToDevice, st)); }
    if (C) BH_HIP(ctx, hipMemcpyAsync(d_fixed, fixed, (size_t)C, hipMemcpyHostToDevice, st));
    if (P) BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, points, sizeof(double) * 3 * (size_t)P, hipMemcpyHostToDevice, st));
    const dim3 T(256);
    auto G = [](long long n) { return dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)); };
    if (E) hipLaunchKernelGGL(k_bld_mark, G(E), T, 0, st, (const int*)ep, (const int*)pp, E, C, P, cam_used, pt_used, info);
    if (C) {
        hipLaunchKernelGGL(k_bld_camflag, G(C), T, 0, st, (const int*)cam_used, (const uint8_t*)d_fixed, C, ba->ar ? 1 : 0, camflag);
        BH_TRY((bh_exscan<int, int>(ctx, st, camflag, camscan, C, scan_tmp)));
        hipLaunchKernelGGL(k_bld_slot, G(C), T, 0, st, (const int*)camflag, (const int*)camscan, C, cam_slot, info + 2);      // info[2] = Cf
    }
    if (P) {
        BH_TRY((bh_exscan<int, int>(ctx, st, pt_used, ptscan, P, scan_tmp)));
        hipLaunchKernelGGL(k_bld_slot, G(P), T, 0, st, (const int*)pt_used, (const int*)ptscan, P, pt_slot, info + 3);        // info[3] = Pa
    }
    int h[8];
    BH_HIP(ctx, hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    if (h[0] != 0x7fffffff) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: edge %d references a missing vertex", h[0]);
    const int Cf = C ? h[2] : 0, Pa = P ? h[3] : 0;
    out->Cf = Cf; out->Pa = Pa;
    int* e_pt = (int*)ba->e_pt.p;
    if (E) {
        BH_HIP(ctx, hipMemcpyAsync(ba->e_cam.p, ep, sizeof(int) * (size_t)E, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_bld_edges, G(E), T, 0, st, (const int*)ep, (const int*)pp, E, (const int*)pt_slot, (const int*)cam_slot, e_pt,
                           (unsigned long long*)nullptr, (int*)nullptr, p_cnt, c_cnt, Cf);
    }
    // p_start / c_start: exclusive scans of the counts (one extra element gives the total; p_cnt[Pa] = c_cnt[Cf] = 0 from the memset)
    BH_TRY((bh_exscan<int, int>(ctx, st, p_cnt, (int*)ba->p_start.p, Pa + 1, scan_tmp)));
    BH_TRY((bh_exscan<int, int>(ctx, st, c_cnt, (int*)ba->c_start.p, Cf + 1, scan_tmp)));
    if (E) {
        hipLaunchKernelGGL(k_g_pfill, G(E), T, 0, st, (const int*)e_pt, E, (const int*)ba->p_start.p, pfill, tmp);
        if (Pa) hipLaunchKernelGGL(k_g_prank, G(Pa), T, 0, st, (const int*)ba->p_start.p, (const int*)tmp, (const int*)ep, Pa, (int*)ba->p_edges.p);
        hipLaunchKernelGGL(k_g_cfill, G(E), T, 0, st, (const int*)ba->p_edges.p, (const int*)ep, (const int*)cam_slot, E, (const int*)ba->c_start.p, cfill, ctmp);
        if (Cf) hipLaunchKernelGGL(k_g_crank, dim3(Cf), T, 0, st, (const int*)ba->c_start.p, (const int*)ctmp, (const int*)ba->p_edges.p, (const int*)e_pt,
                                   (int*)ba->c_edges.p, (int*)ba->c_pt.p);
    }
    // camera pairs
    const long long nbits = (long long)Cf * Cf, nw = (nbits + 31) / 32;
    if (nw >= (1ll << 28)) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: %d free keyframes: the camera-pair bitmap would need %lld MB", Cf, nw >> 18);
    int npairs = 0, bwb = 0, c_total = 0;
    long long M_total = 0;
    if (Cf) {
        // the band (largest slot distance inside a landmark) and the number of free-camera edges come back in one read
        if (Pa) hipLaunchKernelGGL(k_g_band, G(Pa), T, 0, st, (const int*)ba->p_start.p, (const int*)ba->p_edges.p, (const int*)ep, (const int*)cam_slot, Pa, info);
        int hv[8];
        BH_HIP(ctx, hipMemcpyAsync(info + 5, (int*)ba->c_start.p + Cf, sizeof(int), hipMemcpyDeviceToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(hv, info, sizeof(hv), hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
        bwb = hv[1]; c_total = hv[5];
        const long long W = (long long)bwb + 1, nb = (long long)Cf * W;
        unsigned long long M = 0;
        int* j_cnt = nullptr;
        const char* pm = getenv("BH_BA_PAIRS");                  // A/B + test switch: "bitmap" forces the wide-band path
        if (nb <= (1ll << 26) && !(pm && !strcmp(pm, "bitmap"))) {
            // narrow band: counters cnt[lo][hi - lo] mark the pairs AND count their join entries in one pass over the landmarks
            BH_TRY(bh_reserve(ctx, &ctx->st_f, sizeof(int) * (3 * (size_t)nb + 4) + 64));
            int* cnt = (int*)ctx->st_f.p; int* flag = cnt + nb; int* pos = flag + nb + 1;
            unsigned long long* tot = (unsigned long long*)(((uintptr_t)(pos + nb + 1) + 15) & ~(uintptr_t)15);
            BH_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nb, st)); BH_HIP(ctx, hipMemsetAsync(tot, 0, sizeof(*tot), st));
            if (Pa) hipLaunchKernelGGL(k_g_bandcount, G(Pa), T, 0, st, (const int*)ba->p_start.p, (const int*)ba->p_edges.p, (const int*)ep, (const int*)cam_slot, Pa, (int)W, cnt, tot);
            hipLaunchKernelGGL(k_g_bandflag, G(nb + 1), T, 0, st, (const int*)cnt, nb, (int)W, Cf, flag);
            BH_TRY(bh_reserve(ctx, &ctx->st_e, bh_exscan_scratch<int>(nb + 2) + 256));
            BH_TRY((bh_exscan<int, int>(ctx, st, flag, pos, nb + 1, ctx->st_e.p)));
            BH_HIP(ctx, hipMemcpyAsync(&npairs, pos + nb, sizeof(int), hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipMemcpyAsync(&M, tot, sizeof(M), hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipStreamSynchronize(st));
            BH_TRY(bh_reserve(ctx, &ba->pairs, sizeof(int2) * ((size_t)npairs + 1)));
            BH_TRY(bh_reserve(ctx, &ctx->st_g, sizeof(int) * ((size_t)npairs + 2) + 64));
            j_cnt = (int*)ctx->st_g.p;
            BH_HIP(ctx, hipMemsetAsync(j_cnt + npairs, 0, sizeof(int) * 2, st));
            hipLaunchKernelGGL(k_g_bandemit, G(nb), T, 0, st, (const int*)cnt, (const int*)flag, (const int*)pos, nb, (int)W, (int2*)ba->pairs.p, j_cnt);
        } else {
            // wide band (a dense window with many free cameras): Cf x Cf bitmap, pairs in bit order, a second pass counts by the rank of the pair's bit
            BH_TRY(bh_reserve(ctx, &ctx->st_f, sizeof(unsigned) * (size_t)nw + sizeof(int) * 2 * ((size_t)nw + 1)));
            unsigned* bits = (unsigned*)ctx->st_f.p; int* wcnt = (int*)(bits + nw); int* woff = wcnt + nw + 1;
            BH_HIP(ctx, hipMemsetAsync(bits, 0, sizeof(unsigned) * (size_t)nw, st));
            BH_HIP(ctx, hipMemsetAsync(wcnt + nw, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_bld_pairs_mark, G((long long)Pa + Cf), T, 0, st, (const int*)ba->p_start.p, (const int*)ba->p_edges.p, (const int*)ep,
                               (const int*)cam_slot, Pa, Cf, bits, info);
            hipLaunchKernelGGL(k_bld_popc, G(nw), T, 0, st, (const unsigned*)bits, nw, wcnt);
            BH_TRY((bh_exscan<int, int>(ctx, st, wcnt, woff, nw + 1, scan_tmp)));
            BH_HIP(ctx, hipMemcpyAsync(&npairs, woff + nw, sizeof(int), hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipStreamSynchronize(st));
            BH_TRY(bh_reserve(ctx, &ba->pairs, sizeof(int2) * ((size_t)npairs + 1)));
            hipLaunchKernelGGL(k_bld_pairs_emit, G(nw), T, 0, st, (const unsigned*)bits, (const int*)woff, nw, Cf, (int2*)ba->pairs.p);
            BH_TRY(bh_reserve(ctx, &ctx->st_g, sizeof(int) * ((size_t)npairs + 2) + 64));
            j_cnt = (int*)ctx->st_g.p; unsigned long long* tot = (unsigned long long*)(j_cnt + (((size_t)npairs + 2 + 1) & ~(size_t)1));
            BH_HIP(ctx, hipMemsetAsync(j_cnt, 0, sizeof(int) * ((size_t)npairs + 2) + 64, st));
            if (Pa) {
                hipLaunchKernelGGL(k_g_jcount, G(Pa), T, 0, st, (const int*)ba->p_start.p, (const int*)ba->p_edges.p, (const int*)ep, (const int*)cam_slot, Pa, Cf,
                                   (const unsigned*)bits, (const int*)woff, j_cnt, tot);
                BH_HIP(ctx, hipMemcpyAsync(&M, tot, sizeof(M), hipMemcpyDeviceToHost, st));
                BH_HIP(ctx, hipStreamSynchronize(st));
            }
        }
        // the Schur complement's work list: (edge, edge) pairs per camera pair, landmarks ascending inside a pair
        if (M >= (1ull << 31)) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: %llu landmark / camera-pair incidences (limit 2^31)", M);
        M_total = (long long)M;
        BH_TRY(bh_reserve(ctx, &ba->j_start, sizeof(int) * ((size_t)npairs + 2)));
        BH_TRY(bh_reserve(ctx, &ctx->st_e, bh_exscan_scratch<int>((long long)npairs + 2) + 256));
        BH_TRY((bh_exscan<int, int>(ctx, st, j_cnt, (int*)ba->j_start.p, (long long)npairs + 1, ctx->st_e.p)));
        BH_TRY(bh_reserve(ctx, &ba->join, sizeof(int2) * (size_t)(M_total + 1)));
        if (npairs && M_total) hipLaunchKernelGGL(k_g_join_merge, dim3(npairs), T, 0, st, (const int2*)ba->pairs.p, npairs, (const int*)ba->c_start.p, (const int*)ba->c_edges.p,
                                                  (const int*)ba->c_pt.p, (const int*)ba->j_start.p, (int2*)ba->join.p);
    } else {
        BH_TRY(bh_reserve(ctx, &ba->j_start, sizeof(int) * 2)); BH_TRY(bh_reserve(ctx, &ba->join, sizeof(int2)));
        BH_HIP(ctx, hipMemsetAsync(ba->j_start.p, 0, sizeof(int) * 2, st));
    }
    out->npairs = npairs; out->bwb = bwb; out->c_total = c_total; out->M = M_total;
    // active points in slot order; host mirrors of the slots (get_points / get_poses map back through them)
    BH_TRY(bh_reserve(ctx, &ba->pts[0], sizeof(double) * 3 * ((size_t)Pa + 1)));
    if (P) hipLaunchKernelGGL(k_bld_gather_pts, G(P), T, 0, st, (const double*)ctx->st_c.p, (const int*)pt_slot, P, (double*)ba->pts[0].p);
    ba->cam_slot.assign(C, -1); ba->pt_slot.assign(P, -1);
    if (C) BH_HIP(ctx, hipMemcpyAsync(ba->cam_slot.data(), cam_slot, sizeof(int) * (size_t)C, hipMemcpyDeviceToHost, st));
    if (P) BH_HIP(ctx, hipMemcpyAsync(ba->pt_slot.data(), pt_slot, sizeof(int) * (size_t)P, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One-workgroup builder for local windows (round 3).  ba_build_device above is ~30 launches (~60 while it used library sorts)
// and four host reads - 0.4 ms for a live window whose seven LM iterations take 0.6 ms, and the part of local mapping that the
// batched optimiser does not cover.  A window with at most BLD_SF free keyframes is built by ONE 1024-thread workgroup
// (k_bld_small: slots, edges by point, edges by camera, camera pairs, join offsets) plus one launch of a wave per camera pair
// (k_bld_join_merge), from user arrays that went up in one copy; what the host needs comes back in one.  Same arrays bit for bit as
// the general builder and the host loops (test_device_built_problem_equals_host_built runs all three), built without any sort:
//   p_edges   per-point lists are filled in arbitrary order (integer atomics) and then RANKED inside the point by (camera, input
//             position) - the keys are unique, so the order is total and nothing depends on the fill order;
//   c_edges   = the p_edges sequence filtered by camera: a stable partition.  Wave w owns a contiguous chunk; lane s of the wave keeps
//             camera s's running position in a register (hence <= 64 free cameras), a batch of 64 edges is placed by one ballot per
//             distinct camera in it;
//   pairs     counted per (s1, s2) in LDS by integer atomics, compacted in lexicographic order by a block scan; j_start = scan of the counts;
//   join      camera s1's and s2's lists are both sorted by landmark: a wave per pair walks s1's list, finds each landmark in s2's by
//             binary search and writes the hits in order (ballot prefix) - landmark order without a sort.
#define BLD_SF 64                    // free cameras a window may have (lane per camera; 64 x 64 pair counters in LDS)
#define BLD_SC 1024                  // user cameras
#define BLD_LE 12288                 // edges: four 16-bit arrays of this length live in LDS
#define BLD_LP 4096                  // landmarks
#define BLD_MJ 4096                  // k_bld_join_merge: entries of camera s2's list staged in LDS
#define BLD_E2 (1 << 18)             // tier 2 (grid-wide kernels, global scratch): edges, landmarks
#define BLD_P2 (1 << 16)
struct bld_view {
    int C, P, E, all_used;
    const int* ep; const int* pp; const uint8_t* fixed; const double* points; const double* meas;      // staged user arrays (device)
    int* e_cam; int* e_pt; double* e_meas; int* cslot; int* pslot;
    int* p_start; int* p_edges; int* c_start; int* c_edges; int* c_pt;                                  // c_pt: landmark of every c_edges entry (scratch for the join)
    int2* pairs; int* j_start; int2* join; long long join_cap;
    double* pts;
    int* info;                       // out: [0] first bad edge, [1] max |s1 - s2|, [2] Cf, [3] Pa, [4] npairs, [5] c_total, [6] M, [7] refused (1: Cf > BLD_SF, 2: join list too long)
    int* back_cslot;                 // copy of cslot next to info and pslot: what the host mirrors comes back in ONE read
    // what bh_ba_set_problem does after the build, here in the same launch: camera blocks, cleared control block / residuals / S / step
    const ba_cam* cams_src; ba_cam* cams_dst; ba_K K;
    double* S; double* dxp; double* err0; ba_state* state;
    // tier 2 (a local window too big for LDS: the same steps as grid-wide kernels over zero-initialised global scratch)
    int tier, nwaves;
    int* cam_used; int* pcnt; int* pfill; int* ccnt; int* jc; int* wh; int* tmp;
};
__host__ __device__ inline size_t bld_lds_bytes(int E, int P)
{
    const size_t e2 = ((size_t)E + 1) & ~(size_t)1, p2 = ((size_t)P + 2) & ~(size_t)1;
    return 2 * (4 * e2 + 3 * p2) + 16;
}

// exclusive prefix of one int per thread over the 1024-thread workgroup; *total = the sum (s_w: 16 ints of LDS)
__device__ __forceinline__ int bld_exscan(int v, int* s_w, int* total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();                                             // s_w may still be read from the previous scan
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const int t = s_w[w]; if (w < wv) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}
// Stable partition step for one batch of 64 entries: lane l holds camera slot sl (or -1) of entry l, and - as camera `lane` - that
// camera's next output position in `run`.  Returns the output position of the lane's entry and advances the counters.  Six ballots
// (one per bit of the slot) give every lane the set of lanes with ITS slot and, as a camera, the set of lanes that are its entries.
__device__ __forceinline__ int bld_place64(int sl, int lane, int& run)
{
    const unsigned long long valid = __ballot(sl >= 0);
    unsigned long long same = valid, mine = valid;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const unsigned long long mb = __ballot((sl >> b) & 1);
        same &= ((sl >> b) & 1) ? mb : ~mb;
        mine &= ((lane >> b) & 1) ? mb : ~mb;
    }
    const int base = __shfl(run, sl & 63, 64);
    run += __popcll(mine);
    return base + __popcll(same & ((1ull << lane) - 1));
}
// 16-bit counters, two per LDS word (LDS atomics are 32 bits wide; a half never reaches 65 536 here)
__device__ __forceinline__ int bld_add16(uint16_t* a, int i, int v)
{
    const unsigned sh = 16u * ((unsigned)i & 1u);
    const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(a) + (i >> 1), (unsigned)v << sh);
    return (int)((old >> sh) & 0xffffu);
}

__global__ __launch_bounds__(1024) void k_bld_small(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.tier != 1) return;                         // a window of the batch that is not a local one, or one too big for LDS
#ifdef BLD_CLOCKS
    int clk_i = 0;
#define BLD_CLK() do { if (threadIdx.x == 0 && clk_i < 10) V.info[8 + clk_i] = (int)wall_clock64(); clk_i++; } while (0)
#else
#define BLD_CLK() do {} while (0)
#endif
    BLD_CLK();
    extern __shared__ __attribute__((aligned(16))) uint16_t s_dyn[];
    __shared__ int8_t s_cs[BLD_SC];                              // user camera -> free slot or -1 (first: used flag)
    __shared__ int s_jc[BLD_SF * BLD_SF];                        // (s1, s2) -> number of (edge, edge) pairs on a common landmark
    __shared__ int s_cc[BLD_SF + 1];
    __shared__ int s_w[16];
    __shared__ int s_bad, s_bw;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = V.C, P = V.P, E = V.E;
    const int e2 = (E + 1) & ~1, p2 = (P + 2) & ~1;
    uint16_t* s_ep = s_dyn;                                      // user camera of edge e
    uint16_t* s_pt = s_ep + e2;                                  // user landmark, then landmark slot
    uint16_t* s_tmp = s_pt + e2;                                 // per-point lists in fill order
    uint16_t* s_pe = s_tmp + e2;                                 // p_edges
    uint16_t* s_ps = s_pe + e2;                                  // user landmark -> slot (0xffff: none); first: used flag
    uint16_t* s_st = s_ps + p2;                                  // counts, then p_start
    uint16_t* s_fl = s_st + p2;                                  // fill counters
    for (int i = tid; i < C; i += 1024) s_cs[i] = 0;
    for (int i = tid; i < BLD_SF * BLD_SF; i += 1024) s_jc[i] = 0;
    for (int i = tid; i < 3 * p2 / 2; i += 1024) reinterpret_cast<uint32_t*>(s_ps)[i] = 0;
    if (tid <= BLD_SF) s_cc[tid] = 0;
    if (tid == 0) { s_bad = 0x7fffffff; s_bw = 0; }
    __syncthreads();
    BLD_CLK();
    // which vertices take part
    for (int e = tid; e < E; e += 1024) {
        const int c = V.ep[e], q = V.pp[e];
        if (c < 0 || c >= C || q < 0 || q >= P) atomicMin(&s_bad, e);
        else { s_ep[e] = (uint16_t)c; s_pt[e] = (uint16_t)q; s_cs[c] = 1; s_ps[q] = 1; V.e_cam[e] = c; }
    }
    __syncthreads();
    if (s_bad != 0x7fffffff) { if (tid == 0) { V.info[0] = s_bad; V.info[7] = 0; } return; }
    BLD_CLK();
    // slots: rank among the used, non-fixed cameras / among the points that have an edge
    int Cf, Pa;
    {
        const int f = (tid < C && (V.all_used || s_cs[tid]) && !V.fixed[tid]) ? 1 : 0;
        const int ex = bld_exscan(f, s_w, &Cf);
        if (tid < C) { const int sl = (f && ex < BLD_SF) ? ex : -1; s_cs[tid] = (int8_t)sl; V.cslot[tid] = f ? ex : -1; V.back_cslot[tid] = f ? ex : -1; }
    }
    if (Cf > BLD_SF) { if (tid == 0) { V.info[0] = 0x7fffffff; V.info[7] = 1; } return; }
    {
        const int q0 = tid * 4;
        int fl[4], n = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { fl[i] = (q0 + i < P) ? s_ps[q0 + i] : 0; n += fl[i]; }
        int run = bld_exscan(n, s_w, &Pa);
#pragma unroll
        for (int i = 0; i < 4; i++) if (q0 + i < P) { const int sl = fl[i] ? run : -1; s_ps[q0 + i] = (uint16_t)sl; V.pslot[q0 + i] = sl; run += fl[i]; }
    }
    __syncthreads();
    BLD_CLK();
    // counts per point and per free camera
    for (int e = tid; e < E; e += 1024) {
        const int j = s_ps[s_pt[e]];
        s_pt[e] = (uint16_t)j; V.e_pt[e] = j;
        bld_add16(s_st, j, 1);
        const int sl = s_cs[s_ep[e]];
        if (sl >= 0) atomicAdd(&s_cc[sl], 1);
    }
    __syncthreads();
    int c_total;
    {
        const int j0 = tid * 4;
        int cn[4], n = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { cn[i] = (j0 + i < Pa) ? s_st[j0 + i] : 0; n += cn[i]; }
        int tot, run = bld_exscan(n, s_w, &tot);
#pragma unroll
        for (int i = 0; i < 4; i++) if (j0 + i <= Pa) { s_st[j0 + i] = (uint16_t)run; V.p_start[j0 + i] = run; run += cn[i]; }
        if (tid == 0) { s_st[Pa] = (uint16_t)tot; V.p_start[Pa] = tot; }              // (index 4096 has no owner thread)
        const int cv = tid < Cf ? s_cc[tid] : 0;
        const int cex = bld_exscan(cv, s_w, &c_total);
        if (tid <= Cf) { V.c_start[tid] = cex; s_cc[tid] = cex; }
    }
    __syncthreads();
    BLD_CLK();
    // per-point lists in fill order, then ranked by (camera, input position)
    for (int e = tid; e < E; e += 1024) {
        const int j = s_pt[e];
        const int pos = bld_add16(s_fl, j, 1);
        s_tmp[s_st[j] + pos] = (uint16_t)e;
    }
    __syncthreads();
    for (int j = tid; j < Pa; j += 1024) {
        const int a0 = s_st[j], k = s_st[j + 1] - a0;
        for (int i = 0; i < k; i++) {
            const int ei = s_tmp[a0 + i], ci = s_ep[ei];
            int r = 0;
            for (int i2 = 0; i2 < k; i2++) { const int ex = s_tmp[a0 + i2], c2 = s_ep[ex]; r += (c2 < ci || (c2 == ci && ex < ei)) ? 1 : 0; }
            s_pe[a0 + r] = (uint16_t)ei;
        }
    }
    __syncthreads();
    BLD_CLK();
    for (int k = tid; k < E; k += 1024) V.p_edges[k] = s_pe[k];
    // camera pairs: how many (edge, edge) pairs of a landmark fall on (s1, s2), s1 <= s2 (a landmark's edges are sorted by camera)
    for (int j = tid; j < Pa; j += 1024) {
        const int a0 = s_st[j], a1 = s_st[j + 1];
        for (int u = a0; u < a1; u++) {
            const int s1 = s_cs[s_ep[s_pe[u]]]; if (s1 < 0) continue;
            for (int v = u; v < a1; v++) { const int sb = s_cs[s_ep[s_pe[v]]]; if (sb >= 0) atomicAdd(&s_jc[s1 * Cf + sb], 1); }
        }
    }
    __syncthreads();
    BLD_CLK();
    int npairs, M;
    {   // lexicographic compaction: thread t owns the four cells 4 t .. 4 t + 3 of the Cf x Cf table
        int fl[4], cn[4], nf = 0, nc = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = 4 * tid + i, s1 = idx < Cf * Cf ? idx / Cf : 0, sb = idx - s1 * Cf;
            const int c = idx < Cf * Cf ? s_jc[idx] : 0;
            fl[i] = (idx < Cf * Cf && sb >= s1 && (c > 0 || s1 == sb)) ? 1 : 0; cn[i] = fl[i] ? c : 0;
            nf += fl[i]; nc += cn[i];
            if (fl[i] && sb - s1 > 0) atomicMax(&s_bw, sb - s1);
        }
        int pi = bld_exscan(nf, s_w, &npairs);
        int jo = bld_exscan(nc, s_w, &M);
#pragma unroll
        for (int i = 0; i < 4; i++) if (fl[i]) {
            const int idx = 4 * tid + i, s1 = idx / Cf, sb = idx - s1 * Cf;
            V.pairs[pi] = make_int2(s1, sb); V.j_start[pi] = jo; pi++; jo += cn[i];
        }
        if (tid == 0) V.j_start[npairs] = M;
    }
    BLD_CLK();
    // c_edges: stable partition of the p_edges sequence by free camera.  Lane s holds camera s's next position.
    {
        const int ch = ((E + 15) / 16 + 63) & ~63, k0 = wv * ch, k1 = min(E, k0 + ch);
        __syncthreads();                                         // s_jc is dead: it takes the per-wave histograms
        int* s_wh = s_jc;                                        // [16][BLD_SF]
        for (int i = tid; i < 16 * BLD_SF; i += 1024) s_wh[i] = 0;
        __syncthreads();
        for (int k = k0 + lane; k < k1; k += 64) { const int sl = s_cs[s_ep[s_pe[k]]]; if (sl >= 0) atomicAdd(&s_wh[wv * BLD_SF + sl], 1); }
        __syncthreads();
        int run = 0;
        if (lane < Cf) { run = s_cc[lane]; for (int w = 0; w < wv; w++) run += s_wh[w * BLD_SF + lane]; }
        for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane;
            const int e = k < k1 ? s_pe[k] : 0;
            const int sl = k < k1 ? s_cs[s_ep[e]] : -1;
            const int o = bld_place64(sl, lane, run);
            if (sl >= 0) { V.c_edges[o] = e; V.c_pt[o] = s_pt[e]; }
        }
    }
    BLD_CLK();
    if (tid == 0) {
        V.info[0] = 0x7fffffff; V.info[1] = s_bw; V.info[2] = Cf; V.info[3] = Pa; V.info[4] = npairs; V.info[5] = c_total; V.info[6] = M;
        V.info[7] = (long long)M > V.join_cap ? 2 : 0;
    }
}

// ---- tier 2: the steps of k_bld_small as grid-wide kernels (blockIdx.y = window) for local windows that do not fit LDS (configs[2]:
// 80 k observations).  Scratch is zero-initialised by one memset; info[0] counts as 0x7fffffff - (first bad edge) until k_b2_compact.
#define B2_CH 512                    // p_edges entries per wave in the camera partition
#define B2_LIVE(V) ((V).C != 0 && (V).tier == 2 && (V).info[0] == 0 && (V).info[7] == 0)
__global__ __launch_bounds__(256) void k_b2_mark(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.tier != 2) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= V.E) return;
    const int c = V.ep[e], q = V.pp[e];
    if (c < 0 || c >= V.C || q < 0 || q >= V.P) { atomicMax(&V.info[0], 0x7fffffff - e); return; }
    V.cam_used[c] = 1; V.pslot[q] = 1; V.e_cam[e] = c;
}
__global__ __launch_bounds__(1024) void k_b2_slots(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (!B2_LIVE(V)) return;
    __shared__ int s_w[16];
    const int tid = threadIdx.x, C = V.C, P = V.P;
    int Cf, Pa;
    const int f = (tid < C && (V.all_used || V.cam_used[tid]) && !V.fixed[tid]) ? 1 : 0;
    const int ex = bld_exscan(f, s_w, &Cf);
    if (tid < C) { V.cslot[tid] = f ? ex : -1; V.back_cslot[tid] = f ? ex : -1; }
    const int ch = (P + 1023) >> 10, q0 = tid * ch, q1 = min(P, q0 + ch);
    int n = 0;
    for (int q = q0; q < q1; q++) n += V.pslot[q];
    int run = bld_exscan(n, s_w, &Pa);
    for (int q = q0; q < q1; q++) { const int fl = V.pslot[q]; V.pslot[q] = fl ? run : -1; run += fl; }
    if (tid == 0) { V.info[2] = Cf; V.info[3] = Pa; if (Cf > BLD_SF) V.info[7] = 1; }
}
__global__ __launch_bounds__(256) void k_b2_count(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (!B2_LIVE(V)) return;
    __shared__ int s_cc[BLD_SF];
    if (threadIdx.x < BLD_SF) s_cc[threadIdx.x] = 0;
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < V.E) {
        const int j = V.pslot[V.pp[e]];
        V.e_pt[e] = j;
        atomicAdd(&V.pcnt[j], 1);
        const int sl = V.cslot[V.ep[e]];
        if (sl >= 0) atomicAdd(&s_cc[sl], 1);
    }
    __syncthreads();
    if (threadIdx.x < BLD_SF) { const int v = s_cc[threadIdx.x]; if (v) atomicAdd(&V.ccnt[threadIdx.x], v); }
}
__global__ __launch_bounds__(1024) void k_b2_scan(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (!B2_LIVE(V)) return;
    __shared__ int s_w[16];
    const int tid = threadIdx.x, Cf = V.info[2], Pa = V.info[3];
    const int* __restrict__ cnt = V.pcnt; int* __restrict__ ps = V.p_start;
    const int ch = (Pa + 1 + 1023) >> 10, j0 = tid * ch, j1 = min(Pa + 1, j0 + ch);
    int n = 0;
    for (int j = j0; j < j1; j++) n += cnt[j];
    int tot, run = bld_exscan(n, s_w, &tot);
    for (int j = j0; j < j1; j++) { ps[j] = run; run += cnt[j]; }
    int c_total;
    const int cv = tid < Cf ? V.ccnt[tid] : 0;
    const int cex = bld_exscan(cv, s_w, &c_total);
    if (tid <= Cf) V.c_start[tid] = cex;
    if (tid == 0) V.info[5] = c_total;
}
__global__ __launch_bounds__(256) void k_b2_fill(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (!B2_LIVE(V)) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= V.E) return;
    const int j = V.e_pt[e];
    const int pos = atomicAdd(&V.pfill[j], 1);
    V.tmp[V.p_start[j] + pos] = e;
}
#define B2_MK 16                     // landmark degree up to which k_b2_rank works from an LDS copy of the landmark's list
__global__ __launch_bounds__(256) void k_b2_rank(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (!B2_LIVE(V)) return;
    __shared__ int s_jc[BLD_SF * BLD_SF];
    __shared__ int s_e[B2_MK][256];                              // [entry][thread]: conflict-free
    __shared__ uint16_t s_c[B2_MK][256];
    __shared__ int8_t s_s[B2_MK][256];
    const int Cf = V.info[2], Pa = V.info[3], tid = threadIdx.x;
    for (int i = tid; i < Cf * Cf; i += 256) s_jc[i] = 0;
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j < Pa) {
        const int a0 = V.p_start[j], k = V.p_start[j + 1] - a0;
        const int* __restrict__ tm = V.tmp + a0; const int* __restrict__ ep = V.ep; const int* __restrict__ cs = V.cslot;
        if (k <= B2_MK) {
            // three dependent levels of loads for the whole list (entry, camera, slot), then everything from LDS.  A pair of entries
            // lands on (min slot, max slot): slots grow with the camera index, so no order is needed for the count.
            for (int i = 0; i < k; i++) s_e[i][tid] = tm[i];
            for (int i = 0; i < k; i++) s_c[i][tid] = (uint16_t)ep[s_e[i][tid]];
            for (int i = 0; i < k; i++) s_s[i][tid] = (int8_t)cs[s_c[i][tid]];
            for (int i = 0; i < k; i++) {
                const int ei = s_e[i][tid], ci = s_c[i][tid], si = s_s[i][tid];
                int r = 0;
                for (int i2 = 0; i2 < k; i2++) { const int ex = s_e[i2][tid], c2 = s_c[i2][tid]; r += (c2 < ci || (c2 == ci && ex < ei)) ? 1 : 0; }
                V.p_edges[a0 + r] = ei;
                if (si >= 0) for (int i2 = i; i2 < k; i2++) { const int s2 = s_s[i2][tid]; if (s2 >= 0) atomicAdd(&s_jc[min(si, s2) * Cf + max(si, s2)], 1); }
            }
        } else {
            for (int i = 0; i < k; i++) {
                const int ei = tm[i], ci = ep[ei];
                int r = 0;
                for (int i2 = 0; i2 < k; i2++) { const int ex = tm[i2], c2 = ep[ex]; r += (c2 < ci || (c2 == ci && ex < ei)) ? 1 : 0; }
                V.p_edges[a0 + r] = ei;
                const int si = cs[ci];
                if (si >= 0) for (int i2 = i; i2 < k; i2++) { const int s2 = cs[ep[tm[i2]]]; if (s2 >= 0) atomicAdd(&s_jc[min(si, s2) * Cf + max(si, s2)], 1); }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < Cf * Cf; i += 256) { const int v = s_jc[i]; if (v) atomicAdd(&V.jc[i], v); }
}
__global__ __launch_bounds__(1024) void k_b2_compact(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.tier != 2) return;
    const int tid = threadIdx.x;
    if (V.info[0] != 0 || V.info[7] != 0) { if (tid == 0) V.info[0] = V.info[0] ? 0x7fffffff - V.info[0] : 0x7fffffff; return; }
    __shared__ int s_w[16];
    __shared__ int s_bw;
    if (tid == 0) s_bw = 0;
    __syncthreads();
    const int Cf = V.info[2];
    int fl[4], cn[4], nf = 0, nc = 0, npairs, M;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int idx = 4 * tid + i, s1 = idx < Cf * Cf ? idx / Cf : 0, sb = idx - s1 * Cf;
        const int c = idx < Cf * Cf ? V.jc[idx] : 0;
        fl[i] = (idx < Cf * Cf && sb >= s1 && (c > 0 || s1 == sb)) ? 1 : 0; cn[i] = fl[i] ? c : 0;
        nf += fl[i]; nc += cn[i];
        if (fl[i] && sb - s1 > 0) atomicMax(&s_bw, sb - s1);
    }
    int pi = bld_exscan(nf, s_w, &npairs);
    int jo = bld_exscan(nc, s_w, &M);
#pragma unroll
    for (int i = 0; i < 4; i++) if (fl[i]) {
        const int idx = 4 * tid + i, s1 = idx / Cf, sb = idx - s1 * Cf;
        V.pairs[pi] = make_int2(s1, sb); V.j_start[pi] = jo; pi++; jo += cn[i];
    }
    if (tid == 0) {
        V.j_start[npairs] = M;
        V.info[0] = 0x7fffffff; V.info[1] = s_bw; V.info[4] = npairs; V.info[6] = M; V.info[7] = (long long)M > V.join_cap ? 2 : 0;
    }
}
// camera partition of the p_edges sequence: wave w owns entries [w B2_CH, (w + 1) B2_CH); pass 1 counts per camera, pass 2 places
__global__ __launch_bounds__(256) void k_b2_chist(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.tier != 2 || V.info[0] != 0x7fffffff || V.info[7]) return;
    __shared__ int s_h[4][BLD_SF];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, w = blockIdx.x * 4 + wv;
    s_h[wv][lane] = 0;
    __syncthreads();
    if (w < V.nwaves) {
        const int k0 = w * B2_CH, k1 = min(V.E, k0 + B2_CH);
        for (int k = k0 + lane; k < k1; k += 64) { const int sl = V.cslot[V.ep[V.p_edges[k]]]; if (sl >= 0) atomicAdd(&s_h[wv][sl], 1); }
    }
    __syncthreads();
    if (w < V.nwaves) V.wh[w * BLD_SF + lane] = s_h[wv][lane];
}
__global__ __launch_bounds__(256) void k_b2_cplace(const bld_view* __restrict__ views)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.tier != 2 || V.info[0] != 0x7fffffff || V.info[7]) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, w = blockIdx.x * 4 + wv;
    if (w >= V.nwaves) return;
    const int Cf = V.info[2];
    int run = 0;
    if (lane < Cf) { run = V.c_start[lane]; const int* __restrict__ wh = V.wh; for (int x = 0; x < w; x++) run += wh[x * BLD_SF + lane]; }
    const int k0 = w * B2_CH, k1 = min(V.E, k0 + B2_CH);
    constexpr int NB = B2_CH / 64;
    int e[NB], sl[NB], jp[NB];                                   // all loads of the chunk first (three dependent levels), then the placement
#pragma unroll
    for (int i = 0; i < NB; i++) { const int k = k0 + 64 * i + lane; e[i] = k < k1 ? V.p_edges[k] : -1; }
#pragma unroll
    for (int i = 0; i < NB; i++) { sl[i] = e[i] >= 0 ? V.ep[e[i]] : -1; jp[i] = e[i] >= 0 ? V.e_pt[e[i]] : 0; }
#pragma unroll
    for (int i = 0; i < NB; i++) sl[i] = sl[i] >= 0 ? V.cslot[sl[i]] : -1;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        if (k0 + 64 * i >= k1) break;
        const int o = bld_place64(sl[i], lane, run);
        if (sl[i] >= 0) { V.c_edges[o] = e[i]; V.c_pt[o] = jp[i]; }
    }
}

// One camera pair of the join: the (edge of s1, edge of s2) pairs on a common landmark, landmarks ascending (256 threads).
__device__ __forceinline__ void bld_join_pair(const int p, const int npairs, const int2* __restrict__ pairs, const int* __restrict__ c_start,
                                              const int* __restrict__ c_edges, const int* __restrict__ c_pt, const int* __restrict__ j_start, int2* __restrict__ join)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (p >= npairs) return;
    __shared__ int s_l2[BLD_MJ];
    __shared__ int s_w[4];
    const int2 pr = pairs[p];
    const int a0 = c_start[pr.x], n1 = c_start[pr.x + 1] - a0, b0 = c_start[pr.y], n2 = c_start[pr.y + 1] - b0;
    const int* q2 = c_pt + b0;                                 // landmarks of s2's list
    const bool staged = n2 <= BLD_MJ;
    if (staged) for (int i = tid; i < n2; i += 256) s_l2[i] = q2[i];
    __syncthreads();
    const int* l2 = c_edges + b0;
    int out = j_start[p];
    for (int i0 = 0; i0 < n1; i0 += 256) {
        const int i = i0 + tid;
        int e1 = 0, first = 0, cnt = 0;
        if (i < n1) {
            e1 = c_edges[a0 + i];
            const int j = c_pt[a0 + i];
            int lo = 0, hi = n2;
            if (staged) {
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_l2[mid] < j) lo = mid + 1; else hi = mid; }
                if (pr.x == pr.y) while (lo < n2 && s_l2[lo] == j && l2[lo] < e1) lo++;      // the diagonal block takes the pairs u <= v only
                first = lo;
                while (lo < n2 && s_l2[lo] == j) lo++;
            } else {
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (q2[mid] < j) lo = mid + 1; else hi = mid; }
                if (pr.x == pr.y) while (lo < n2 && q2[lo] == j && l2[lo] < e1) lo++;
                first = lo;
                while (lo < n2 && q2[lo] == j) lo++;
            }
            cnt = lo - first;
        }
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        __syncthreads();
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int t = s_w[w]; if (w < wv) base += t; tot += t; }
        const int wr = out + base + inc - cnt;
        for (int k = 0; k < cnt; k++) join[wr + k] = make_int2(e1, l2[first + k]);
        out += tot;
    }
}

// One workgroup per camera pair: the (edge of s1, edge of s2) pairs on a common landmark, landmarks ascending - both cameras' lists are
// sorted by landmark, so s1's entries look themselves up in s2's (staged in LDS) and a block scan puts the hits in order.  The
// workgroups behind the pairs copy the measurements and gather the active landmarks (independent of everything else in the build).
__global__ __launch_bounds__(256) void k_bld_join_merge(const bld_view* __restrict__ views, int pair_blocks)
{
    const bld_view V = views[blockIdx.y];
    if (V.C == 0 || V.info[0] != 0x7fffffff || V.info[7]) return;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= pair_blocks) {
        const int nx = (int)gridDim.x - pair_blocks, i0 = ((int)blockIdx.x - pair_blocks) * 256 + tid, step = nx * 256;
        for (int i = i0; i < V.E; i += step) { V.e_meas[2 * i] = V.meas[2 * i]; V.e_meas[2 * i + 1] = V.meas[2 * i + 1]; V.err0[2 * i] = 0; V.err0[2 * i + 1] = 0; }
        for (int i = i0; i < V.P; i += step) {
            const int sl = V.pslot[i];
            if (sl >= 0) { V.pts[3 * (size_t)sl] = V.points[3 * (size_t)i]; V.pts[3 * (size_t)sl + 1] = V.points[3 * (size_t)i + 1]; V.pts[3 * (size_t)sl + 2] = V.points[3 * (size_t)i + 2]; }
        }
        for (int i = i0; i < V.C; i += step) { V.cams_dst[i] = V.cams_src[i]; d_cam_refresh(&V.cams_dst[i], V.K); }      // (in place, as k_ba_cam_init: the same code, the same bits)
        // cleared: the LM control block, the residual slot past the last edge, the reduced camera system and the step (as bh_ba_set_problem's memsets)
        const int n6 = 6 * V.info[2], bw = n6 > 0 ? min(n6 - 1, 6 * V.info[1] + 5) : 0;
        const long long nS = (long long)n6 * (bw + 1) + 1;
        for (long long i = i0; i < nS; i += step) V.S[i] = 0;
        for (int i = i0; i < n6 + 6; i += step) V.dxp[i] = 0;
        if (i0 < 2) V.err0[2 * (size_t)V.E + i0] = 0;
        for (int i = i0; i < (int)(sizeof(ba_state) / 4); i += step) reinterpret_cast<int*>(V.state)[i] = 0;
        return;
    }
    bld_join_pair(blockIdx.x, V.info[4], V.pairs, V.c_start, V.c_edges, V.c_pt, V.j_start, V.join);
}
__global__ __launch_bounds__(256) void k_g_join_merge(const int2* __restrict__ pairs, int npairs, const int* __restrict__ c_start, const int* __restrict__ c_edges,
                                                      const int* __restrict__ c_pt, const int* __restrict__ j_start, int2* __restrict__ join)
{
    bld_join_pair(blockIdx.x, npairs, pairs, c_start, c_edges, c_pt, j_start, join);
}

static void ba_builder_attr() { (void)hipFuncSetAttribute((const void*)k_bld_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bld_lds_bytes(BLD_LE, BLD_LP)); }

// Host side of the one-workgroup builder, for W windows at once (W = 1: bh_ba_set_problem).  taken[w] = 0: window w is not a local
// window (or the kernel refused it) and nothing was built for it; taken[w] = 1: structure, measurements, landmarks, camera blocks and
// the cleared buffers are on the device, out[w] holds the counts.  One copy up, two launches, one copy back, one synchronisation.
typedef bh_ba_problem ba_prob;
static void h_quat_from_R(const double* m, double* q);
static int ba_small_many(bh_ctx* ctx, int W, bh_ba* const* bas, const ba_prob* pr, ba_build_out* out, int* taken)
{
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    struct lay { size_t ep, pp, fx, pt, ms, cm, back, cpt, cu, pc, pf, cc, jc, wh, tm; int nfree, tier, nwaves; long long join_cap; size_t maxpairs; };
    std::vector<lay> L(W);
    size_t up = al(sizeof(bld_view) * (size_t)W), back = 0, scr = 0;
    int n_el = 0, max_blocks = 1, n_t1 = 0, n_t2 = 0, maxE2 = 1, maxP2 = 1, maxW2 = 1;
    size_t max_lds = 0;
    for (int w = 0; w < W; w++) {
        taken[w] = 0;
        const ba_prob& q = pr[w]; bh_ba* ba = bas[w];
        if (ba->ar || q.C < 1 || q.C > BLD_SC || q.E < 1 || q.E > BLD_E2 || q.P < 1 || q.P > BLD_P2) continue;
        int nfree = 0;
        for (int i = 0; i < q.C; i++) nfree += q.fixed[i] ? 0 : 1;
        if (nfree > BLD_SF) continue;
        lay& l = L[w];
        l.nfree = nfree;
        l.join_cap = (long long)q.E * (nfree + 1) / 2 + q.E + 1;                   // exact bound without duplicate edges; the kernel checks
        l.maxpairs = (size_t)nfree * (nfree + 1) / 2 + 1;
        l.ep = up; up += al(sizeof(int) * (size_t)q.E); l.pp = up; up += al(sizeof(int) * (size_t)q.E); l.fx = up; up += al((size_t)q.C);
        l.pt = up; up += al(sizeof(double) * 3 * (size_t)q.P); l.ms = up; up += al(sizeof(double) * 2 * (size_t)q.E); l.cm = up; up += al(sizeof(ba_cam) * (size_t)q.C);
        l.back = back; back += sizeof(int) * (8 + (size_t)q.C + (size_t)q.P);
        l.cpt = scr; scr += sizeof(int) * (size_t)q.E;
        l.tier = (q.E <= BLD_LE && q.P <= BLD_LP) ? 1 : 2;
        l.nwaves = (q.E + B2_CH - 1) / B2_CH;
        if (l.tier == 2) {
            l.cu = scr; scr += sizeof(int) * (size_t)q.C; l.pc = scr; scr += sizeof(int) * ((size_t)q.P + 1); l.pf = scr; scr += sizeof(int) * ((size_t)q.P + 1);
            l.cc = scr; scr += sizeof(int) * BLD_SF; l.jc = scr; scr += sizeof(int) * BLD_SF * BLD_SF; l.wh = scr; scr += sizeof(int) * BLD_SF * (size_t)l.nwaves;
            l.tm = scr; scr += sizeof(int) * (size_t)q.E;
            n_t2++; maxE2 = std::max(maxE2, q.E); maxP2 = std::max(maxP2, q.P); maxW2 = std::max(maxW2, l.nwaves);
        } else { n_t1++; max_lds = std::max(max_lds, bld_lds_bytes(q.E, q.P)); }
        taken[w] = 2;                                                              // staged
        n_el++;
        max_blocks = std::max(max_blocks, std::min(64, (std::max(q.E, q.P) + 255) / 256 + 1));      // workgroups behind the pairs (grid-stride copies)
    }
    if (!n_el) return BH_OK;
    const size_t back_al = al(back);
    BH_TRY(bh_pin_reserve(ctx, up + back_al));
    BH_TRY(bh_reserve(ctx, &ctx->st_a, up));
    BH_TRY(bh_reserve(ctx, &ctx->st_d, back_al + scr + 256));
    for (int w = 0; w < W; w++) {                                                  // everything the two kernels write, sized by bounds (Cf <= nfree, Pa <= P)
        if (!taken[w]) continue;
        const ba_prob& q = pr[w]; bh_ba* ba = bas[w]; const lay& l = L[w];
        const size_t E = q.E, P = q.P, C = q.C, n6m = 6 * (size_t)l.nfree;
        BH_TRY(bh_reserve(ctx, &ba->cslot, sizeof(int) * C));
        BH_TRY(bh_reserve(ctx, &ba->e_cam, sizeof(int) * E)); BH_TRY(bh_reserve(ctx, &ba->e_pt, sizeof(int) * E));
        BH_TRY(bh_reserve(ctx, &ba->e_meas, sizeof(double) * 2 * (E + 1)));
        BH_TRY(bh_reserve(ctx, &ba->p_edges, sizeof(int) * E)); BH_TRY(bh_reserve(ctx, &ba->c_edges, sizeof(int) * E));
        BH_TRY(bh_reserve(ctx, &ba->p_start, sizeof(int) * (P + 2))); BH_TRY(bh_reserve(ctx, &ba->c_start, sizeof(int) * (C + 2)));
        BH_TRY(bh_reserve(ctx, &ba->pairs, sizeof(int2) * l.maxpairs)); BH_TRY(bh_reserve(ctx, &ba->j_start, sizeof(int) * (l.maxpairs + 2)));
        BH_TRY(bh_reserve(ctx, &ba->join, sizeof(int2) * (size_t)l.join_cap));
        BH_TRY(bh_reserve(ctx, &ba->pts[0], sizeof(double) * 3 * (P + 1)));
        BH_TRY(bh_reserve(ctx, &ba->cams[0], sizeof(ba_cam) * (C + 1)));
        BH_TRY(bh_reserve(ctx, &ba->state, sizeof(ba_state)));
        BH_TRY(bh_reserve(ctx, &ba->err[0], sizeof(double) * 2 * (E + 1)));
        BH_TRY(bh_reserve(ctx, &ba->S, sizeof(double) * (n6m * n6m + 1))); BH_TRY(bh_reserve(ctx, &ba->dxp, sizeof(double) * (n6m + 6)));
    }
    char* hp = (char*)ctx->pin; char* dp = (char*)ctx->st_a.p; char* ds = (char*)ctx->st_d.p;
    bld_view* hv = (bld_view*)hp;
    bool all_piecewise = true;
    // The VALUES of a window - landmarks, measurements, camera blocks: two thirds of what goes up - are read by the last launch only (the workgroups
    // behind the pairs of k_bld_join_merge).  When every window is a big one (tier 2: configs[2], alone or batched) they are packed and sent AFTER the
    // structure kernels are enqueued, so the host's packing and the transfer run under those kernels instead of in front of them
    // (BH_BA_BUILD_VALUES_FIRST=1: the former order).
    const bool values_later = n_t1 == 0 && !(getenv("BH_BA_BUILD_VALUES_FIRST") && atoi(getenv("BH_BA_BUILD_VALUES_FIRST")) == 1);
    // ... and on a stream of their own (the first auxiliary stream of the context, forked behind whatever the context's stream holds - the staging
    // buffer may still be in use there - and joined in front of the last launch): in stream order the transfer would wait for the structure kernels
    // instead of running beside them.
    hipStream_t vst = (values_later && ctx->aux_stream[0] && ctx->ev_fork && ctx->ev_join[0]) ? ctx->aux_stream[0] : st;
    if (vst != st) { BH_HIP(ctx, hipEventRecord(ctx->ev_fork, st)); BH_HIP(ctx, hipStreamWaitEvent(vst, ctx->ev_fork, 0)); }
    // (an error return between the fork and the join must not leave a transfer from the pinned block in flight on the auxiliary stream: the next call packs into that block)
    struct vjoin_guard { hipStream_t v; bool armed; ~vjoin_guard() { if (armed) (void)hipStreamSynchronize(v); } } vguard{ vst, vst != st };
    auto put_values = [&](int w) -> int {
        const ba_prob& q = pr[w]; const lay& l = L[w];
        const bool piecewise = l.tier == 2;
        auto put = [&](size_t off, const void* src, size_t bytes) -> int {
            memcpy(hp + off, src, bytes);
            if (piecewise) BH_HIP(ctx, hipMemcpyAsync(dp + off, hp + off, bytes, hipMemcpyHostToDevice, vst));
            return BH_OK;
        };
        BH_TRY(put(l.ms, q.meas, sizeof(double) * 2 * (size_t)q.E));
        BH_TRY(put(l.pt, q.points, sizeof(double) * 3 * (size_t)q.P));
        ba_cam* hc = (ba_cam*)(hp + l.cm);
        memset(hc, 0, sizeof(ba_cam) * (size_t)q.C);
        for (int i = 0; i < q.C; i++) { h_quat_from_R(q.R + 9 * (size_t)i, hc[i].q); for (int k = 0; k < 3; k++) hc[i].t[k] = q.t[3 * (size_t)i + k]; }
        if (piecewise) BH_HIP(ctx, hipMemcpyAsync(dp + l.cm, hp + l.cm, sizeof(ba_cam) * (size_t)q.C, hipMemcpyHostToDevice, vst));
        return BH_OK;
    };
    for (int w = 0; w < W; w++) {
        bld_view V = {};
        V.info = (int*)ds;                                                         // (a window that is not staged: C = 0 makes both kernels leave at once)
        if (taken[w]) {
            const ba_prob& q = pr[w]; bh_ba* ba = bas[w]; const lay& l = L[w];
            V.C = q.C; V.P = q.P; V.E = q.E; V.all_used = 0;
            V.ep = (const int*)(dp + l.ep); V.pp = (const int*)(dp + l.pp); V.fixed = (const uint8_t*)(dp + l.fx);
            V.points = (const double*)(dp + l.pt); V.meas = (const double*)(dp + l.ms); V.cams_src = (const ba_cam*)(dp + l.cm);
            int* bk = (int*)(ds + l.back);
            V.info = bk; V.back_cslot = bk + 8; V.pslot = bk + 8 + q.C; V.c_pt = (int*)(ds + back_al + l.cpt);
            V.tier = l.tier; V.nwaves = l.nwaves;
            if (l.tier == 2) {
                char* sc = ds + back_al;
                V.cam_used = (int*)(sc + l.cu); V.pcnt = (int*)(sc + l.pc); V.pfill = (int*)(sc + l.pf); V.ccnt = (int*)(sc + l.cc);
                V.jc = (int*)(sc + l.jc); V.wh = (int*)(sc + l.wh); V.tmp = (int*)(sc + l.tm);
            }
            V.e_cam = (int*)ba->e_cam.p; V.e_pt = (int*)ba->e_pt.p; V.e_meas = (double*)ba->e_meas.p; V.cslot = (int*)ba->cslot.p;
            V.p_start = (int*)ba->p_start.p; V.p_edges = (int*)ba->p_edges.p; V.c_start = (int*)ba->c_start.p; V.c_edges = (int*)ba->c_edges.p;
            V.pairs = (int2*)ba->pairs.p; V.j_start = (int*)ba->j_start.p; V.join = (int2*)ba->join.p; V.join_cap = l.join_cap;
            V.pts = (double*)ba->pts[0].p; V.cams_dst = (ba_cam*)ba->cams[0].p; V.K = ba->K;
            V.S = (double*)ba->S.p; V.dxp = (double*)ba->dxp.p; V.err0 = (double*)ba->err[0].p; V.state = (ba_state*)ba->state.p;
            // a big window (configs[2]: 2.2 MB) goes up array by array, each copy enqueued as soon as its array is packed - the transfer of
            // one runs under the packing of the next; small windows wait for the one copy of everything below
            const bool piecewise = l.tier == 2;
            auto put = [&](size_t off, const void* src, size_t bytes) -> int {
                memcpy(hp + off, src, bytes);
                if (piecewise) BH_HIP(ctx, hipMemcpyAsync(dp + off, hp + off, bytes, hipMemcpyHostToDevice, st));
                return BH_OK;
            };
            BH_TRY(put(l.ep, q.edge_pose, sizeof(int) * (size_t)q.E)); BH_TRY(put(l.pp, q.edge_point, sizeof(int) * (size_t)q.E));
            BH_TRY(put(l.fx, q.fixed, (size_t)q.C));
            if (!values_later) BH_TRY(put_values(w));
            all_piecewise &= piecewise;
        }
        hv[w] = V;
    }
    if (all_piecewise) BH_HIP(ctx, hipMemcpyAsync(dp, hp, al(sizeof(bld_view) * (size_t)W), hipMemcpyHostToDevice, st));      // only the view table is left
    else BH_HIP(ctx, hipMemcpyAsync(dp, hp, up, hipMemcpyHostToDevice, st));
    int max_pairs = 1;
    for (int w = 0; w < W; w++) if (taken[w]) max_pairs = std::max(max_pairs, (int)L[w].maxpairs);
    const bld_view* dv = (const bld_view*)dp;
    if (n_t2) BH_HIP(ctx, hipMemsetAsync(ds, 0, back_al + scr, st));             // tier 2 counts into zero-initialised scratch (tier 1 does not care)
    if (n_t1) {
        bh_time_begin(ctx, BH_K_BA_OTHER);
        hipLaunchKernelGGL(k_bld_small, dim3(1, W), dim3(1024), max_lds, st, dv);
        bh_time_end(ctx);
    }
    if (n_t2) {
        const dim3 gE((maxE2 + 255) / 256, W), gP((maxP2 + 255) / 256, W), g1(1, W), gW((maxW2 + 3) / 4, W);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_mark, gE, dim3(256), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_slots, g1, dim3(1024), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_count, gE, dim3(256), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_scan, g1, dim3(1024), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_fill, gE, dim3(256), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_rank, gP, dim3(256), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_compact, g1, dim3(1024), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_chist, gW, dim3(256), st, dv);
        BH_LAUNCH(ctx, BH_K_BA_OTHER, k_b2_cplace, gW, dim3(256), st, dv);
    }
    if (values_later) {
        for (int w = 0; w < W; w++) if (taken[w]) BH_TRY(put_values(w));
        if (vst != st) { BH_HIP(ctx, hipEventRecord(ctx->ev_join[0], vst)); BH_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join[0], 0)); vguard.armed = false; }
    }
    BH_LAUNCH(ctx, BH_K_BA_OTHER, k_bld_join_merge, dim3((unsigned)(max_pairs + max_blocks), W), dim3(256), st, (const bld_view*)dp, max_pairs);
    char* hb0 = hp + up;
    BH_HIP(ctx, hipMemcpyAsync(hb0, ds, back, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipStreamSynchronize(st));
    for (int w = 0; w < W; w++) {
        if (!taken[w]) continue;
        const ba_prob& q = pr[w]; bh_ba* ba = bas[w];
        const int* hb = (const int*)(hb0 + L[w].back);
        if (hb[0] != 0x7fffffff) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: edge %d references a missing vertex", hb[0]);
        if (hb[7]) { taken[w] = 0; continue; }                                     // refused (duplicate-heavy edge list): the general builder takes it
        out[w].bwb = hb[1]; out[w].Cf = hb[2]; out[w].Pa = hb[3]; out[w].npairs = hb[4]; out[w].c_total = hb[5]; out[w].M = hb[6];
        ba->cam_slot.assign(hb + 8, hb + 8 + q.C); ba->pt_slot.assign(hb + 8 + q.C, hb + 8 + q.C + q.P);
        taken[w] = 1;
    }
    return BH_OK;
}

template <typename T>
static int upload(bh_ctx* ctx, bh_buf* b, const std::vector<T>& v)
{
    BH_TRY(bh_reserve(ctx, b, sizeof(T) * (v.size() + 1)));
    if (!v.empty()) BH_HIP(ctx, hipMemcpyAsync(b->p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx->stream));
    return BH_OK;
}

static int ba_set_problem_rest(bh_ba* ba, int C, const double* R, const double* t, const uint8_t* fixed, int P, const double* points, int E,
                               const int* edge_pose, const int* edge_point, const double* meas, bool host_build, int small, ba_build_out bo);

extern "C" int bh_ba_set_problem(bh_ba* ba, int C, const double* R, const double* t, const uint8_t* fixed,
                                 int P, const double* points, int E, const int* edge_pose, const int* edge_point,
                                 const double* meas, int mem)
{
    if (!ba) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (mem != BH_MEM_HOST) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: the graph is built on the host side (BH_MEM_HOST only)");
    if (C < 0 || P < 0 || E < 0 || (C && (!R || !t || !fixed)) || (P && !points) || (E && (!edge_pose || !edge_point || !meas)))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: bad argument");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    // The structure is built on the device (ba_build_device).  BH_BA_BUILD=host selects the former host loops, kept as the A/B
    // reference the tests compare against (same arrays bit for bit).
    const char* bmode = getenv("BH_BA_BUILD");
    const bool host_build = bmode && !strcmp(bmode, "host");
    if (ba->ar && E == 0) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: a rank of a sharded problem needs at least one edge");
    ba->have_problem = 0;                                    // an error below leaves no half-built problem behind
    int small = 0;                                           // local windows: one workgroup builds the structure (BH_BA_BUILD=general: the chip-wide builder)
    ba_build_out bo = {};
    if (!host_build && !(bmode && !strcmp(bmode, "general"))) {
        const ba_prob q = { C, R, t, fixed, P, points, E, edge_pose, edge_point, meas };
        bh_ba* one = ba;
        BH_TRY(ba_small_many(ctx, 1, &one, &q, &bo, &small));
    }
    return ba_set_problem_rest(ba, C, R, t, fixed, P, points, E, edge_pose, edge_point, meas, host_build, small, bo);
}

// bh_ba_set_problem behind the one-workgroup builder: the chip-wide / host builders for everything that is not a local window, then the
// solver choice and the working buffers.  small != 0: structure, camera blocks and cleared buffers are already on the device.
static int ba_set_problem_rest(bh_ba* ba, int C, const double* R, const double* t, const uint8_t* fixed, int P, const double* points, int E,
                               const int* edge_pose, const int* edge_point, const double* meas, bool host_build, int small, ba_build_out bo)
{
    bh_ctx* ctx = ba->ctx;
    int Cf = 0, Pa = 0, bwb = 0, c_total = 0;
    std::vector<ba_cam> cams(small ? 0 : C);
    for (int i = 0; i < C && !small; i++) {
        memset(&cams[i], 0, sizeof(ba_cam));
        h_quat_from_R(R + 9 * i, cams[i].q);
        for (int k = 0; k < 3; k++) cams[i].t[k] = t[3 * i + k];
    }
    if (!host_build) {
        ba->C = C; ba->P = P; ba->E = E; ba->edge_user.clear();
        if (!small) {
            BH_TRY(ba_build_device(ba, C, fixed, P, points, E, edge_pose, edge_point, &bo));
            BH_TRY(bh_reserve(ctx, &ba->e_meas, sizeof(double) * 2 * ((size_t)E + 1)));
            if (E) BH_HIP(ctx, hipMemcpyAsync(ba->e_meas.p, meas, sizeof(double) * 2 * (size_t)E, hipMemcpyHostToDevice, ctx->stream));
        }
        Cf = bo.Cf; Pa = bo.Pa; bwb = bo.bwb; c_total = bo.c_total; ba->npairs = bo.npairs;
        ba->Cf = Cf; ba->Pa = Pa; ba->Ea = E;
    } else {
        for (int e = 0; e < E; e++)
            if (edge_pose[e] < 0 || edge_pose[e] >= C || edge_point[e] < 0 || edge_point[e] >= P)
                return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: edge %d references a missing vertex", e);
        ba->C = C; ba->P = P; ba->E = E;
        // active structure (g2o initializeOptimization): points are free; a vertex takes part iff it has an edge
        ba->cam_slot.assign(C, -1); ba->pt_slot.assign(P, -1); ba->edge_user.clear();
        std::vector<char> cam_used(C, 0);
        for (int e = 0; e < E; e++) { cam_used[edge_pose[e]] = 1; ba->pt_slot[edge_point[e]] = 0; }
        if (ba->ar) {                                          // sharded: every rank must number the free cameras identically
            if (E == 0) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: a rank of a sharded problem needs at least one edge");
            for (int i = 0; i < C; i++) cam_used[i] = 1;
        }
        Cf = 0; Pa = 0;
        for (int i = 0; i < C; i++) if (cam_used[i] && !fixed[i]) ba->cam_slot[i] = Cf++;
        for (int j = 0; j < P; j++) if (ba->pt_slot[j] == 0) ba->pt_slot[j] = Pa++;
        ba->Cf = Cf; ba->Pa = Pa; ba->Ea = E;
        std::vector<double> pts(3 * (size_t)Pa);
        for (int j = 0; j < P; j++) if (ba->pt_slot[j] >= 0) for (int k = 0; k < 3; k++) pts[3 * (size_t)ba->pt_slot[j] + k] = points[3 * (size_t)j + k];
        std::vector<int> e_cam(E), e_pt(E); std::vector<double> e_meas(2 * (size_t)E);
        for (int e = 0; e < E; e++) { e_cam[e] = edge_pose[e]; e_pt[e] = ba->pt_slot[edge_point[e]]; e_meas[2 * e] = meas[2 * e]; e_meas[2 * e + 1] = meas[2 * e + 1]; }
        // CSR by point (edges ordered by camera) and by free camera (edges ordered by point)
        // order = edges sorted by (point, camera, input position): two stable counting sorts, O(E + C + Pa) instead of a comparison sort
        std::vector<int> order(E), p_start(Pa + 1, 0);
        {
            std::vector<int> by_cam(E), cnt(C + 1, 0);
            for (int e = 0; e < E; e++) cnt[e_cam[e] + 1]++;
            for (int i = 0; i < C; i++) cnt[i + 1] += cnt[i];
            for (int e = 0; e < E; e++) by_cam[cnt[e_cam[e]]++] = e;
            for (int e = 0; e < E; e++) p_start[e_pt[e] + 1]++;
            for (int j = 0; j < Pa; j++) p_start[j + 1] += p_start[j];
            std::vector<int> fill(p_start.begin(), p_start.end() - 1);
            for (int k = 0; k < E; k++) { const int e = by_cam[k]; order[fill[e_pt[e]]++] = e; }
        }
        std::vector<int> p_edges(order);
        std::vector<int> c_start(Cf + 1, 0), c_edges;
        for (int e = 0; e < E; e++) { const int s = ba->cam_slot[e_cam[e]]; if (s >= 0) c_start[s + 1]++; }
        for (int s = 0; s < Cf; s++) c_start[s + 1] += c_start[s];
        c_edges.resize(c_start[Cf]);
        { std::vector<int> fill(Cf, 0);
          for (int k = 0; k < E; k++) { const int e = order[k]; const int s = ba->cam_slot[e_cam[e]]; if (s >= 0) c_edges[c_start[s] + fill[s]++] = e; } }
        // camera pairs that share at least one point (upper triangle incl. diagonal)
        std::vector<char> cov((size_t)Cf * Cf, 0);
        for (int j = 0; j < Pa; j++)
            for (int u = p_start[j]; u < p_start[j + 1]; u++) {
                const int s1 = ba->cam_slot[e_cam[p_edges[u]]]; if (s1 < 0) continue;
                for (int v = u; v < p_start[j + 1]; v++) { const int s2 = ba->cam_slot[e_cam[p_edges[v]]]; if (s2 >= 0) cov[(size_t)(s1 < s2 ? s1 : s2) * Cf + (s1 < s2 ? s2 : s1)] = 1; }
            }
        std::vector<int2> pairs;
        for (int a = 0; a < Cf; a++) for (int b = a; b < Cf; b++) if (cov[(size_t)a * Cf + b] || a == b) pairs.push_back(make_int2(a, b));
        ba->npairs = (int)pairs.size();
        c_total = c_start[Cf];
        for (const int2& pr : pairs) bwb = (pr.y - pr.x) > bwb ? (pr.y - pr.x) : bwb;
        BH_TRY(upload(ctx, &ba->pts[0], pts));
        BH_TRY(upload(ctx, &ba->e_cam, e_cam)); BH_TRY(upload(ctx, &ba->e_pt, e_pt)); BH_TRY(upload(ctx, &ba->e_meas, e_meas));
        BH_TRY(upload(ctx, &ba->cslot, ba->cam_slot));
        BH_TRY(upload(ctx, &ba->p_start, p_start)); BH_TRY(upload(ctx, &ba->p_edges, p_edges));
        BH_TRY(upload(ctx, &ba->c_start, c_start)); BH_TRY(upload(ctx, &ba->c_edges, c_edges));
        BH_TRY(upload(ctx, &ba->pairs, pairs));
        {   // the join list, as the device build produces it: per camera pair (lexicographic), landmarks ascending
            std::vector<int> j_start(pairs.size() + 1, 0);
            std::vector<long long> pair_of((size_t)Cf * Cf, -1);
            for (size_t q = 0; q < pairs.size(); q++) pair_of[(size_t)pairs[q].x * Cf + pairs[q].y] = (long long)q;
            for (int j = 0; j < Pa; j++)
                for (int u = p_start[j]; u < p_start[j + 1]; u++) {
                    const int s1 = ba->cam_slot[e_cam[p_edges[u]]]; if (s1 < 0) continue;
                    for (int v = u; v < p_start[j + 1]; v++) { const int s2 = ba->cam_slot[e_cam[p_edges[v]]]; if (s2 >= 0) j_start[pair_of[(size_t)s1 * Cf + s2] + 1]++; }
                }
            for (size_t q = 0; q < pairs.size(); q++) j_start[q + 1] += j_start[q];
            std::vector<int2> join(j_start[pairs.size()] + 1);
            std::vector<int> fill(j_start.begin(), j_start.end() - 1);
            for (int j = 0; j < Pa; j++)
                for (int u = p_start[j]; u < p_start[j + 1]; u++) {
                    const int e1 = p_edges[u], s1 = ba->cam_slot[e_cam[e1]]; if (s1 < 0) continue;
                    for (int v = u; v < p_start[j + 1]; v++) { const int e2 = p_edges[v], s2 = ba->cam_slot[e_cam[e2]]; if (s2 >= 0) join[fill[pair_of[(size_t)s1 * Cf + s2]]++] = make_int2(e1, e2); }
                }
            BH_TRY(upload(ctx, &ba->join, join)); BH_TRY(upload(ctx, &ba->j_start, j_start));
            bo.M = j_start[pairs.size()];
        }
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the host vectors above go out of scope
    }
    ba->schur_wave = (Cf > 0 && ba->npairs >= 4096 && c_total / Cf < 1024) ? 1 : 0;
    {   // A/B switch (off by default): a wave per camera pair also for short join lists (BH_BA_SCHUR_WAVE_AVG = largest average list length
        // that takes it).  Measured at 512: 64 batched live windows +7 % (the 256-thread form keeps 188 VGPRs per lane, two workgroups
        // per CU, 8 rounds of ~10 us), but one live window -10 % and one configs[2] window -6 %: not kept as the default.
        int avg_max = 0;
        if (const char* ev = getenv("BH_BA_SCHUR_WAVE_AVG")) avg_max = atoi(ev);
        if (Cf > 0 && ba->npairs > 0 && !ba->ar && bo.M / ba->npairs <= avg_max && avg_max > 0) ba->schur_wave = 1;
    }
    ba->join_len = bo.M;
    if (ba->schur_wave && ba->npairs > 0) {
        BH_TRY(bh_reserve(ctx, &ba->pair_rec, sizeof(int4) * (size_t)ba->npairs)); BH_TRY(bh_reserve(ctx, &ba->join_pt, sizeof(int) * (size_t)(bo.M + 1)));
        const long long n = std::max<long long>(ba->npairs, bo.M);
        hipLaunchKernelGGL(k_ba_schur_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const int2*)ba->pairs.p, (const int*)ba->j_start.p, ba->npairs,
                           (const int2*)ba->join.p, (const int*)ba->e_pt.p, (long long)bo.M, (int4*)ba->pair_rec.p, (int*)ba->join_pt.p);
        BH_HIP(ctx, hipGetLastError());
        // the row form: every free camera has its diagonal pair, so every row exists; BH_BA_SCHUR_ROW=0 keeps the wave-per-pair kernels
        const char* ev = getenv("BH_BA_SCHUR_ROW");
        ba->schur_row = (ev && atoi(ev) == 0) ? 0 : 1;
        if (ba->schur_row) {
            BH_TRY(bh_reserve(ctx, &ba->row_start, sizeof(int) * ((size_t)Cf + 2))); BH_TRY(bh_reserve(ctx, &ba->join_slot, sizeof(int) * (size_t)(bo.M + 1)));
            BH_TRY(bh_reserve(ctx, &ba->c_pt, sizeof(int) * (size_t)(E > 0 ? E : 1)));
            BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(int) * (size_t)(E > 0 ? E : 1)));
            int* e_rank = (int*)ctx->st_a.p;
            BH_HIP(ctx, hipMemsetAsync((int*)ba->row_start.p + Cf + 1, 0, sizeof(int), ctx->stream));
            BH_TRY(bh_reserve(ctx, &ba->c_meas, sizeof(double) * 2 * (size_t)(E > 0 ? E : 1)));
            hipLaunchKernelGGL(k_ba_schur_rank, dim3(Cf), dim3(256), 0, ctx->stream, (const int*)ba->c_start.p, (const int*)ba->c_edges.p, (const int*)ba->e_pt.p,
                               (const double*)ba->e_meas.p, Cf, e_rank, (int*)ba->c_pt.p, (double*)ba->c_meas.p);
            ba->cam_major = 1;
            hipLaunchKernelGGL(k_ba_schur_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const int2*)ba->pairs.p, ba->npairs, Cf, (const int2*)ba->join.p,
                               (const int*)e_rank, (long long)bo.M, (const int*)ba->e_cam.p, (int*)ba->row_start.p, (int*)ba->join_slot.p);
            BH_HIP(ctx, hipGetLastError());
            (void)hipFuncSetAttribute((const void*)k_ba_schur_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 18 * SR_CAP));
        }
    } else ba->schur_row = 0;
    if (!ba->schur_row) ba->cam_major = 0;
    BH_TRY(bh_reserve(ctx, &ba->state, sizeof(ba_state)));
    if (ba->ar) {                                          // the band must be the same everywhere: max over ranks
        double v = (double)bwb;
        double* red = (double*)((char*)ba->state.p + offsetof(ba_state, red));
        BH_HIP(ctx, hipMemcpyAsync(red, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        BH_TRY(ba_allreduce(ba, red, 1, 1));
        BH_HIP(ctx, hipMemcpyAsync(&v, red, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bwb = (int)v;
    }
    const int n6i = 6 * Cf;
    ba->bw = n6i > 0 ? std::min(n6i - 1, 6 * bwb + 5) : 0;
    {   // narrow band, long matrix: the trailing window stays on chip (BH_BA_CHOL=panel|window overrides)
        const char* ev = getenv("BH_BA_CHOL");
        const bool fits = (ba->bw + 15) / 16 <= 16 && ba->bw >= 15;
        ba->chol_window = (fits && n6i >= 256) ? 1 : 0;
        if (ev && !strcmp(ev, "panel")) ba->chol_window = 0;
        if (ev && !strcmp(ev, "window") && fits) ba->chol_window = 1;
        // two workgroups eliminating from both ends when the band spans exactly 16 tiles and the matrix is long (BH_BA_CHOL_TWOWAY=0 disables)
        const int nblk = (n6i + 15) / 16;
        const char* tw = getenv("BH_BA_CHOL_TWOWAY");
        ba->chol_split = (ba->chol_window && (ba->bw + 15) / 16 == 16 && nblk >= 96 && !(tw && !strcmp(tw, "0"))) ? (nblk - 16) / 2 : 0;
    }
    {   // long narrow systems: block cyclic reduction on the whole chip (BH_BA_CHOL_BCR=0 keeps the window kernels)
        const char* eb = getenv("BH_BA_CHOL_BCR");
        ba->chol_bcr = (ba->chol_window && n6i >= 8 * 256 && !(eb && !strcmp(eb, "0"))) ? 1 : 0;
    }
    BH_TRY(bh_reserve(ctx, &ba->xch, std::max(sizeof(double) * SW_SCRATCH_DOUBLES(n6i, ba->bw), ba->chol_bcr ? BCR_SCRATCH_BYTES((n6i + 255) / 256) : (size_t)0)));
    if (ba->chol_bcr) {
        if (!ba->bcr) ba->bcr = new bcr_dev();
        BH_TRY(bcr_prepare(ctx, ctx->stream, ba->bcr, n6i, ba->xch.p));
    }
    long long rows_pad = n6i;                                  // sharded mode: S and the step are padded to equal row chunks per rank
    BH_TRY(dbcr_setup(ba, n6i, ba->bw, &rows_pad));
    {
        const char* ev = getenv("BH_BA_CHOL");
        ba->chol_dense = (!ba->chol_window && n6i >= 1 && n6i <= 16 * CD_MAXT && !(ev && !strcmp(ev, "panel"))) ? 1 : 0;
    }
    if (!ba->chol_window) {   // panel width by LDS budget: min(n, NB+bw) x (NB+1) panel + 1024 partial sums, f64, <= 150 KB
        int nb = 16;                                              // 16 measured fastest (0.37 vs 0.47 ms at n = 288: <32> spills 152 VGPRs)
        while (nb >= 8 && sizeof(double) * ((size_t)std::min(n6i, nb + ba->bw) * (nb + 1) + 1100) > 150 * 1024) nb >>= 1;
        if (nb < 8) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: reduced camera system too wide for the single-workgroup banded solver "
                                      "(%d free keyframes, covisibility half-bandwidth %d keyframes)", Cf, bwb);
        if (const char* e = getenv("BH_BA_CHOL_NB")) { const int v = atoi(e); if ((v == 8 || v == 16 || v == 32) && v <= nb) nb = v; }      // A/B switch
        ba->chol_nb = nb;
    }

    if (!small) BH_TRY(upload(ctx, &ba->cams[0], cams));
    BH_TRY(bh_reserve(ctx, &ba->cams[1], sizeof(ba_cam) * (C + 1)));
    BH_TRY(bh_reserve(ctx, &ba->pts[1], sizeof(double) * 3 * (Pa + 1)));
    const size_t n6 = 6 * (size_t)Cf;
    for (int k = 0; k < 2; k++) { BH_TRY(bh_reserve(ctx, &ba->err[k], sizeof(double) * 2 * (E + 1))); BH_TRY(bh_reserve(ctx, &ba->chi_e[k], sizeof(double) * (E + 1))); }
    BH_TRY(bh_reserve(ctx, &ba->Hpl, sizeof(double) * 18 * ((size_t)E + 1)));
    BH_TRY(bh_reserve(ctx, &ba->Hll, sizeof(double) * 9 * ((size_t)Pa + 1))); BH_TRY(bh_reserve(ctx, &ba->bl, sizeof(double) * 3 * ((size_t)Pa + 1)));
    BH_TRY(bh_reserve(ctx, &ba->Hpp, sizeof(double) * 36 * ((size_t)Cf + 1))); BH_TRY(bh_reserve(ctx, &ba->bp, sizeof(double) * (n6 + 6)));
    BH_TRY(bh_reserve(ctx, &ba->pdiag, sizeof(double) * (Pa + 1))); BH_TRY(bh_reserve(ctx, &ba->cdiag, sizeof(double) * (Cf + 1)));
    BH_TRY(bh_reserve(ctx, &ba->Dinv, sizeof(double) * 9 * ((size_t)Pa + 1))); BH_TRY(bh_reserve(ctx, &ba->db, sizeof(double) * 3 * ((size_t)Pa + 1)));
    BH_TRY(bh_reserve(ctx, &ba->S, sizeof(double) * ((size_t)rows_pad * (size_t)(ba->bw + 1) + 1))); BH_TRY(bh_reserve(ctx, &ba->bs, sizeof(double) * (n6 + 6)));
    BH_TRY(bh_reserve(ctx, &ba->dxp, sizeof(double) * ((size_t)rows_pad + 6))); BH_TRY(bh_reserve(ctx, &ba->dxl, sizeof(double) * 3 * ((size_t)Pa + 1)));
    BH_TRY(bh_reserve(ctx, &ba->part, sizeof(double) * 2 * BA_NPART));
    BH_TRY(bh_reserve(ctx, &ba->fan, sizeof(double) * ba_fan_doubles(Pa, Cf, E)));
    {   // workgroups per free camera in the linearisation: a function of the problem's sizes alone (a window gives the same bits alone and in a batch)
        const int per_cam = C > 0 ? E / C : 0;
        ba->cam_parts = (Cf > 0 && Cf <= 64 && per_cam >= 512) ? std::min(BA_CAM_PARTS_MAX, (per_cam + 511) / 512) : 1;      // configs[2]: 1 600 per keyframe -> 4 (measured 1 / 4 / 7 parts: 26.0 / 21.9 / 24.4 us)
        if (const char* e = getenv("BH_BA_CAM_PARTS")) { const int v = atoi(e); if (v >= 1 && v <= BA_CAM_PARTS_MAX) ba->cam_parts = Cf > 0 ? v : 1; }      // A/B switch
    }
    if (!small) {
        BH_HIP(ctx, hipMemsetAsync(ba->state.p, 0, sizeof(ba_state), ctx->stream));
        BH_HIP(ctx, hipMemsetAsync(ba->err[0].p, 0, sizeof(double) * 2 * (E + 1), ctx->stream));
        BH_HIP(ctx, hipMemsetAsync(ba->S.p, 0, sizeof(double) * (n6 * (size_t)(ba->bw + 1) + 1), ctx->stream));
        BH_HIP(ctx, hipMemsetAsync(ba->dxp.p, 0, sizeof(double) * (n6 + 6), ctx->stream));
        if (C) hipLaunchKernelGGL(k_ba_cam_init, dim3((C + 63) / 64), dim3(64), 0, ctx->stream, (ba_cam*)ba->cams[0].p, C, ba->K);
        BH_HIP(ctx, hipGetLastError());
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // the host vectors above go out of scope
    }
    ba->cur = 0; ba->have_problem = 1; ba->iters_done = 0; ba->log.clear(); ba->last_err_sel = 0;
    return BH_OK;
}

extern "C" int bh_ba_batch_set_problems(bh_ctx* ctx, int W, bh_ba* const* windows, const bh_ba_problem* problems)
{
    if (!ctx) return BH_ERR_INVALID;
    if (W < 0 || (W && (!windows || !problems))) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_set_problems: bad argument");
    for (int w = 0; w < W; w++) {
        const bh_ba_problem& q = problems[w];
        if (!windows[w] || windows[w]->ctx != ctx) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_set_problems: window %d is not a handle of this context", w);
        if (q.C < 0 || q.P < 0 || q.E < 0 || (q.C && (!q.R || !q.t || !q.fixed)) || (q.P && !q.points) || (q.E && (!q.edge_pose || !q.edge_point || !q.meas)))
            return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_set_problems: bad argument (window %d)", w);
        if (windows[w]->ar && q.E == 0) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_set_problem: a rank of a sharded problem needs at least one edge");
    }
    BH_HIP(ctx, hipSetDevice(ctx->device));
    for (int w = 0; w < W; w++) windows[w]->have_problem = 0;         // an error below leaves no half-built window behind
    const char* bmode = getenv("BH_BA_BUILD");
    const bool host_build = bmode && !strcmp(bmode, "host");
    std::vector<ba_build_out> bo(W);
    std::vector<int> taken(W, 0);
    if (W && !host_build && !(bmode && !strcmp(bmode, "general"))) BH_TRY(ba_small_many(ctx, W, windows, problems, bo.data(), taken.data()));
    for (int w = 0; w < W; w++) {
        const bh_ba_problem& q = problems[w];
        BH_TRY(ba_set_problem_rest(windows[w], q.C, q.R, q.t, q.fixed, q.P, q.points, q.E, q.edge_pose, q.edge_point, q.meas, host_build, taken[w], bo[w]));
    }
    return BH_OK;
}

// Test hook: one structural array of the problem as bh_ba_set_problem left it (tests compare the three builders array by array).
// which: 0 e_pt[E], 1 p_start[Pa+1], 2 p_edges[E], 3 c_start[Cf+1], 4 c_edges[c_total], 5 pairs[npairs] (int2), 6 j_start[npairs+1],
// 7 join[M] (int2), 8 cslot[C].  *n_ints = number of 32-bit values written (at most cap_ints).
extern "C" int bh_ba_debug_structure(bh_ba* ba, int which, int* out, long long cap_ints, long long* n_ints)
{
    if (!ba || !out || !n_ints) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (!ba->have_problem) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_debug_structure: no problem set");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int c_total = 0, M = 0;
    if (ba->Cf) BH_HIP(ctx, hipMemcpy(&c_total, (int*)ba->c_start.p + ba->Cf, sizeof(int), hipMemcpyDeviceToHost));
    BH_HIP(ctx, hipMemcpy(&M, (int*)ba->j_start.p + ba->npairs, sizeof(int), hipMemcpyDeviceToHost));
    const void* src = nullptr; long long n = 0;
    switch (which) {
        case 0: src = ba->e_pt.p; n = ba->Ea; break;
        case 1: src = ba->p_start.p; n = ba->Pa + 1; break;
        case 2: src = ba->p_edges.p; n = ba->Ea; break;
        case 3: src = ba->c_start.p; n = ba->Cf + 1; break;
        case 4: src = ba->c_edges.p; n = c_total; break;
        case 5: src = ba->pairs.p; n = 2LL * ba->npairs; break;
        case 6: src = ba->j_start.p; n = ba->npairs + 1; break;
        case 7: src = ba->join.p; n = 2LL * M; break;
        case 8: src = ba->cslot.p; n = ba->C; break;
        default: return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_debug_structure: which = %d", which);
    }
    if (n > cap_ints) return bh_set_err(ctx, BH_ERR_CAPACITY, "bh_ba_debug_structure: %lld values, room for %lld", n, cap_ints);
    if (n) BH_HIP(ctx, hipMemcpy(out, src, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    *n_ints = n;
    return BH_OK;
}

extern "C" int bh_ba_optimize(bh_ba* ba, int iters, double huber_delta, int verbose_errors, int* iters_done)
{
    if (!ba) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (!ba->have_problem) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_optimize: no problem set");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ba->log.clear(); ba->iters_done = 0;
    if (iters_done) *iters_done = 0;
    const int C = ba->C, Cf = ba->Cf, Pa = ba->Pa, Ea = ba->Ea, n6 = 6 * Cf;
    if (Ea == 0 || iters <= 0) return BH_OK;
    // A local window (dense on-chip solve) runs as a batch of ONE: the same kernel bodies - bitwise the same result (asserted for batches) -
    // with the Levenberg control flow on the device, so the host enqueues trial t + 1 before it has seen the verdict of trial t and the
    // device never idles through the host's poll -> launch turnaround (~25 us of a 260 us iteration at configs[2], of a 60 us one at the live
    // size).  The host-driven loop below stays for everything else (banded / cyclic-reduction solves, the multi-GPU form) and as the A/B
    // reference (BH_BA_SELF_BATCH=0).
    const bool self_off = getenv("BH_BA_SELF_BATCH") && atoi(getenv("BH_BA_SELF_BATCH")) == 0;
    const bool fan_small = ((size_t)Ea + 255) / 256 <= FAN_WAVE_MAX && ((size_t)Pa * BP_G + 255) / 256 <= FAN_WAVE_MAX;      // (the batch's 1024-thread control kernels sum their slots with one wave)
    if (!ba->ar && ba->chol_dense && Cf > 0 && fan_small && !self_off) {
        if (!ba->self) { bh_ba* one = ba; BH_TRY(bh_ba_batch_create(ctx, 1, &one, &ba->self)); }
        int itd = 0;
        const int brc = bh_ba_batch_optimize(ba->self, iters, huber_delta, verbose_errors, &itd);
        if (iters_done) *iters_done = itd;
        return brc;
    }
    const int* e_cam = (const int*)ba->e_cam.p; const int* e_pt = (const int*)ba->e_pt.p; const double* e_meas = (const double*)ba->e_meas.p;
    const int* cslot = (const int*)ba->cslot.p;
    const int* p_start = (const int*)ba->p_start.p; const int* p_edges = (const int*)ba->p_edges.p;
    const int* c_start = (const int*)ba->c_start.p; const int* c_edges = (const int*)ba->c_edges.p;
    double *Hpl = (double*)ba->Hpl.p, *Hll = (double*)ba->Hll.p, *bl = (double*)ba->bl.p, *Hpp = (double*)ba->Hpp.p, *bp = (double*)ba->bp.p;
    double *pdiag = (double*)ba->pdiag.p, *cdiag = (double*)ba->cdiag.p, *Dinv = (double*)ba->Dinv.p, *db = (double*)ba->db.p;
    double *S = (double*)ba->S.p, *bs = (double*)ba->bs.p, *dxp = (double*)ba->dxp.p, *dxl = (double*)ba->dxl.p;
    ba_state* state = (ba_state*)ba->state.p;
    double* red = (double*)((char*)ba->state.p + offsetof(ba_state, red));
    const bool dist = ba->ar != nullptr;
    double* part = Ea > 16384 ? (double*)ba->part.p : nullptr;       // (multi-GPU form) two-stage sums above this size (one workgroup streams ~50 GB/s: 80 k residuals took 22 + 31 us in k_ba_prep / k_ba_decide)
    // the sums of the Levenberg control ride in the kernels that produce their terms (fan-in, see fan_store); in the multi-GPU form the last
    // workgroup leaves this rank's sums in state->red for the all-reduce hook and phase 2 of k_ba_prep / k_ba_decide takes the decision
    const bool fan_off = getenv("BH_BA_FAN") && atoi(getenv("BH_BA_FAN")) == 0;       // A/B switch
    const bool use_fan = !fan_off;
    ba_fan F = {}; if (use_fan) F = ba_fan_of(ba);
    if (use_fan) {
        BH_HIP(ctx, hipMemsetAsync(&state->tick[0], 0, sizeof(int) * 2, st));
        BH_HIP(ctx, hipMemsetAsync(F.grp, 0, sizeof(int) * ((((size_t)Pa * BP_G + 255) / 256 + (size_t)BA_CAM_PARTS_MAX * Cf + ((size_t)Ea + 255) / 256 + 63) / 64 + 8), st));
        if (F.cp > 1) BH_HIP(ctx, hipMemsetAsync(F.c_cnt, 0, sizeof(int) * Cf, st));
    }
    const int lead = ba->rank == 0 ? 1 : 0;              // the rank that contributes the replicated camera terms
    ba->comm_ms = 0; ba->comm_bytes = 0; ba->ar_bytes = 0;
    if (ba->dsol) { ba->dsol->bytes_rs = ba->dsol->bytes_ag = 0; ba->dsol->ms_replicated = 0; ba->dsol->n_replicated = 0; }
    const int pblocks = (Pa + 255) / 256, cblocks = (C + 255) / 256;
    int rc = BH_OK;
    for (int it = 0; it < iters; it++) {
        const ba_cam* cams = (const ba_cam*)ba->cams[ba->cur].p; const double* pts = (const double*)ba->pts[ba->cur].p;
        {
            const int nbp = (Pa * BP_G + 255) / 256;
            BH_LAUNCH(ctx, BH_K_BA_LINEARIZE, k_ba_linearize, dim3(nbp + Cf * (F.cp > 1 ? F.cp : 1)), dim3(256), st, nbp, cams, pts, e_cam, e_pt, e_meas, cslot, p_start, p_edges,
                      c_start, c_edges, Pa, ba->K, huber_delta, Hpl, Hll, bl, (double*)ba->err[0].p, (double*)ba->chi_e[0].p, pdiag, Hpp, bp, cdiag,
                      ba->cam_major ? (const int*)ba->c_pt.p : (const int*)nullptr, ba->cam_major ? (const double*)ba->c_meas.p : (const double*)nullptr,
                      F, state, it == 0 ? 1 : 0, dist ? 1 : 0);
        }
        if (dist && Cf) {                                  // Hpp, bp: partial over this rank's edges -> full on every rank
            BH_TRY(ba_allreduce(ba, Hpp, 36LL * Cf, 0)); BH_TRY(ba_allreduce(ba, bp, n6, 0));
            hipLaunchKernelGGL(k_ba_cdiag, dim3((Cf + 63) / 64), dim3(64), 0, st, (const double*)Hpp, Cf, cdiag);
        }
        if (!use_fan) {
            if (part) BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_partial, dim3(BA_NPART), dim3(1024), st, (const double*)ba->chi_e[0].p, Ea, (const double*)nullptr,
                                (const double*)nullptr, 0, (const ba_state*)state, part);
            BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_prep, dim3(1), dim3(1024), st, (const double*)ba->chi_e[0].p, Ea, (const double*)pdiag, Pa,
                      (const double*)cdiag, Cf, it == 0 ? 1 : 0, dist ? 1 : 0, (const double*)part, state);
        }
        if (dist) {
            BH_TRY(ba_allreduce(ba, red, 1, 0));
            if (it == 0) BH_TRY(ba_allreduce(ba, red + 1, 1, 1));
            BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_prep, dim3(1), dim3(1024), st, (const double*)ba->chi_e[0].p, Ea, (const double*)pdiag, Pa,
                      (const double*)cdiag, Cf, it == 0 ? 1 : 0, use_fan ? 3 : 2, (const double*)part, state);
        }
        ba_state hs;
        if (it == 0) {
            BH_HIP(ctx, hipMemcpyAsync(&hs, state, sizeof(hs), hipMemcpyDeviceToHost, st));
            BH_HIP(ctx, hipStreamSynchronize(st));
            if (!std::isfinite(hs.max_diag)) { rc = bh_set_err(ctx, BH_ERR_NUMERIC, "bh_ba_optimize: non-finite Hessian at the initial estimate"); break; }
        }
        ba->last_err_sel = 0;
        do {
            ba_cam* cams_t = (ba_cam*)ba->cams[ba->cur ^ 1].p; double* pts_t = (double*)ba->pts[ba->cur ^ 1].p;
            const bool zero_in_dinv = Cf > 0 && ba->chol_dense && !dist;      // a local window: S is cleared by extra workgroups of k_ba_dinv
            const long long nS0 = zero_in_dinv ? (long long)n6 * (ba->bw + 1) : 0;
            BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_ba_dinv, dim3(pblocks + (zero_in_dinv ? 32 : 0)), dim3(256), st, (const double*)Hll, (const double*)bl, Pa, state, Dinv, db,
                      pblocks, S, nS0);
            if (Cf) {
                const bool dsolve = dist && ba->dsol && ba->dsol->active;
                const size_t s_rows = dsolve ? 256 * (size_t)ba->dsol->seg * ba->world : (size_t)n6;       // sharded solve: equal row chunks per rank
                if (!zero_in_dinv) BH_HIP(ctx, hipMemsetAsync(S, 0, sizeof(double) * s_rows * (ba->bw + 1), st));     // blocks without a common point stay zero
                if (ba->schur_wave && ba->schur_row)
                {
                    bh_time_begin(ctx, BH_K_BA_SCHUR);
                    hipLaunchKernelGGL(k_ba_schur_row, dim3((Cf + 7) / 8 * 8), dim3(SR_T), sizeof(double) * 18 * SR_CAP, st, (const int*)ba->row_start.p, (const int4*)ba->pair_rec.p,
                                       (const int2*)ba->join.p, (const int*)ba->join_slot.p, c_start, c_edges, (const int*)ba->c_pt.p, e_pt, (const double*)Hpl, (const double*)Dinv, (const double*)db,
                                       (const double*)Hpp, (const double*)bp, (const ba_state*)state, ba->bw, dist ? lead : 1, Cf, S, bs);
                    bh_time_end(ctx);
                }
                else if (ba->schur_wave)
                    BH_LAUNCH(ctx, BH_K_BA_SCHUR, (k_ba_schur_S<64>), dim3((ba->npairs + 7) / 8 * 8), dim3(64), st, (const int2*)ba->pairs.p, (const int*)ba->j_start.p, (const int2*)ba->join.p, e_pt,
                              (const double*)Hpl, (const double*)Dinv, (const double*)Hpp, (const ba_state*)state, ba->bw, dist ? lead : 1, S, ba->npairs,
                              (const int4*)ba->pair_rec.p, (const int*)ba->join_pt.p);
                else {
                    const int ngS = (ba->npairs + 7) / 8 * 8;
                    BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_ba_schur_Sb, dim3(ngS + Cf), dim3(256), st, ngS, (const int2*)ba->pairs.p, (const int*)ba->j_start.p, (const int2*)ba->join.p, e_pt,
                              (const double*)Hpl, (const double*)Dinv, (const double*)Hpp, (const ba_state*)state, ba->bw, dist ? lead : 1, S, ba->npairs,
                              c_start, c_edges, (const double*)db, (const double*)bp, dist ? lead : 1, bs);
                }
                if (ba->schur_wave && !ba->schur_row)
                    BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_ba_schur_b, dim3(Cf), dim3(256), st, c_start, c_edges, e_pt, (const double*)Hpl, (const double*)db,
                              (const double*)bp, dist ? lead : 1, bs);
                if (dsolve) { BH_TRY(ba_allreduce(ba, S, (long long)s_rows * (ba->bw + 1), 2)); BH_TRY(ba_allreduce(ba, bs, n6, 0)); }      // every rank keeps the sum of ITS rows
                else if (dist) { BH_TRY(ba_allreduce(ba, S, (long long)n6 * (ba->bw + 1), 0)); BH_TRY(ba_allreduce(ba, bs, n6, 0)); }
                {
                    const int nbsel = ba->chol_nb, bw = ba->bw;
                    const size_t lds = sizeof(double) * ((size_t)std::min(n6, nbsel + bw) * (nbsel + 1) + 1100);
                    int solve_rc = BH_OK;
                    bh_time_begin(ctx, BH_K_BA_SOLVE);
                    if (dsolve) solve_rc = dbcr_solve(ba, (const double*)S, (const double*)bs, n6, bw, dxp, state);
                    else if (ba->chol_bcr) solve_rc = bcr_solve(ctx, st, ba->bcr, (const double*)S, (const double*)bs, n6, bw, dxp, state);
                    else if (ba->chol_window) solve_rc = sw_solve(ctx, st, S, (const double*)bs, n6, bw, dxp, state, ba->chol_split, (double*)ba->xch.p);
                    else if (ba->chol_dense) hipLaunchKernelGGL(k_ba_chol_dense<false>, dim3(1), dim3(1024), sizeof(double) * CD_LDS_DOUBLES, st, (const double*)S, (const double*)bs, n6, bw, dxp, state,
                                                                 (const bcr_elim*)nullptr, (double*)nullptr, (const double*)nullptr, (double*)nullptr);
                    else if (nbsel == 32) hipLaunchKernelGGL(k_ba_chol<32>, dim3(1), dim3(1024), lds, st, S, (const double*)bs, n6, bw, dxp, state);
                    else if (nbsel == 16) hipLaunchKernelGGL(k_ba_chol<16>, dim3(1), dim3(1024), lds, st, S, (const double*)bs, n6, bw, dxp, state);
                    else hipLaunchKernelGGL(k_ba_chol<8>, dim3(1), dim3(1024), lds, st, S, (const double*)bs, n6, bw, dxp, state);
                    bh_time_end(ctx);
                    BH_TRY(solve_rc);                       // a failed launch inside the multi-kernel solvers (the error text is set there)
                }
            }
            const int upblocks = (Pa * BP_G + 255) / 256;       // the update's landmark workgroups: BP_G lanes per landmark
            const int seq = ++ba->seq;
            ba_state* hpin = ba->h_state;                // may be null (allocation failed): then the copy + sync path below
            BH_LAUNCH(ctx, BH_K_BA_UPDATE, k_ba_update, dim3(upblocks + cblocks), dim3(256), st, cams, cams_t, pts, pts_t, cslot, C, Pa, upblocks,
                      p_start, p_edges, e_cam, (const double*)Hpl, (const double*)Dinv, (const double*)bl, (const double*)dxp, dxl, ba->K, F, (const ba_state*)state);
            BH_LAUNCH(ctx, BH_K_BA_UPDATE, k_ba_err, dim3(ba_err_blocks(Ea)), dim3(256), st, (const ba_cam*)cams_t, (const double*)pts_t, e_cam, e_pt,
                      e_meas, Ea, huber_delta, (double*)ba->err[1].p, (double*)ba->chi_e[1].p, F, upblocks, (const double*)dxp, (const double*)bp, n6,
                      dist ? 1 : 0, dist ? lead : 1, state, dist ? (ba_state*)nullptr : hpin, seq);
            if (!use_fan) {
                if (part) BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_partial, dim3(BA_NPART), dim3(1024), st, (const double*)ba->chi_e[1].p, Ea, (const double*)dxl,
                                    (const double*)bl, 3 * Pa, (const ba_state*)state, part);
                BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_decide, dim3(1), dim3(1024), st, (const double*)ba->chi_e[1].p, Ea, (const double*)dxp, (const double*)bp,
                          n6, (const double*)dxl, (const double*)bl, 3 * Pa, dist ? 1 : 0, dist ? lead : 1, (const double*)part, state, dist ? (ba_state*)nullptr : hpin, seq);
            }
            if (dist) {
                BH_TRY(ba_allreduce(ba, red, 3, 0));
                BH_LAUNCH(ctx, BH_K_BA_OTHER, k_ba_decide, dim3(1), dim3(1024), st, (const double*)ba->chi_e[1].p, Ea, (const double*)dxp, (const double*)bp,
                          n6, (const double*)dxl, (const double*)bl, 3 * Pa, 2, lead, (const double*)part, state, hpin, seq);
            }
            BH_HIP(ctx, hipGetLastError());
            bool polled = false;
            if (hpin) {                                  // poll the pinned verdict: a copy + stream sync costs ~40 us of wake-up per trial
                const auto tpoll = std::chrono::steady_clock::now();
                for (long spin = 0; ; spin++) {              // bounded by time, not by a spin count (a slow trial is not a fault: the copy + sync path below takes over)
                    if (__atomic_load_n(&hpin->pad, __ATOMIC_ACQUIRE) == seq) { polled = true; break; }
                    if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - tpoll > std::chrono::milliseconds(200)) break;
                }
                if (polled) memcpy(&hs, hpin, sizeof(hs));
            }
            if (!polled) {                               // a kernel fault never delivers the verdict: the stream reports it here
                BH_HIP(ctx, hipMemcpyAsync(&hs, state, sizeof(hs), hipMemcpyDeviceToHost, st));
                BH_HIP(ctx, hipStreamSynchronize(st));
            }
            ba->last_err_sel = 1;                       // the trial's errors are the last computeActiveErrors
            if (hs.accepted) { ba->cur ^= 1; cams = (const ba_cam*)ba->cams[ba->cur].p; pts = (const double*)ba->pts[ba->cur].p; }
        } while (hs.again);
        // verbose branch of SparseOptimizer::optimize: errors recomputed at the (possibly restored) current estimate
        if (verbose_errors && !hs.accepted) ba->last_err_sel = 0;
        ba->log.push_back(hs.accepted ? hs.chi_cur : hs.chi_trial); ba->log.push_back(hs.lambda);
        ba->log.push_back((double)hs.qmax); ba->log.push_back((double)hs.accepted);
        ba->iters_done = it + 1;
        if (hs.terminate) break;
    }
    if (iters_done) *iters_done = ba->iters_done;
    ba->verbose_last = verbose_errors;
    return rc;
}

// ---- batched local BA: W windows that each hold a problem (bh_ba_set_problem) advance together, one launch per kernel and trial
struct bh_ba_batch {
    bh_ctx* ctx = nullptr;
    std::vector<bh_ba*> win;
    bh_buf views, curs;
    ba_state* hstates = nullptr;      // pinned, TWO per window (round parity): k_bb_decide's verdicts - the host runs one round ahead of them
    ba_view* hviews = nullptr;        // pinned staging of the view table and the windows' state selectors (behind hstates in the same allocation): they go up
    int* hcurs = nullptr;             // without a synchronisation (as host vectors they had to be waited for before going out of scope: a host round trip per call)
    int seq = 0;
};

extern "C" int bh_ba_batch_create(bh_ctx* ctx, int W, bh_ba* const* windows, bh_ba_batch** out)
{
    if (!ctx || !windows || !out || W < 1) return ctx ? bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_create: bad argument") : BH_ERR_INVALID;
    for (int w = 0; w < W; w++)
        if (!windows[w] || windows[w]->ctx != ctx) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_create: window %d belongs to another context", w);
    BH_HIP(ctx, hipSetDevice(ctx->device));
    (void)hipFuncSetAttribute((const void*)k_bb_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CD_LDS_DOUBLES));
    bh_ba_batch* b = new bh_ba_batch();
    b->ctx = ctx; b->win.assign(windows, windows + W);
    const size_t hs_bytes = (sizeof(ba_state) * 2 * W + 255) & ~(size_t)255, hv_bytes = (sizeof(ba_view) * W + 255) & ~(size_t)255;
    if (hipHostMalloc((void**)&b->hstates, hs_bytes + hv_bytes + sizeof(int) * W, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError(); delete b;
        return bh_set_err(ctx, BH_ERR_HIP, "bh_ba_batch_create: pinned allocation failed");
    }
    memset(b->hstates, 0, hs_bytes + hv_bytes + sizeof(int) * W);
    b->hviews = (ba_view*)((char*)b->hstates + hs_bytes); b->hcurs = (int*)((char*)b->hstates + hs_bytes + hv_bytes);
    *out = b;
    return BH_OK;
}

extern "C" void bh_ba_batch_destroy(bh_ba_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->hstates) (void)hipHostFree(b->hstates);
    for (bh_buf* q : { &b->views, &b->curs }) { if (q->p) (void)hipFree(q->p); q->p = nullptr; q->cap = 0; }
    delete b;
}

extern "C" int bh_ba_batch_optimize(bh_ba_batch* b, int iters, double huber_delta, int verbose_errors, int* iters_done)
{
    if (!b) return BH_ERR_INVALID;
    bh_ctx* ctx = b->ctx;
    const int W = (int)b->win.size(), W8 = W == 1 ? 1 : (W + 7) & ~7;      // grids are (blocks, windows rounded up to 8): see bb_where
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // (the pinned staging is free: the previous call returned after reading its last verdicts, which the device wrote behind these copies)
    ba_view* const hv = b->hviews; int* const hcur = b->hcurs;
    int g_lin = 1, g_pb = 1, g_schur = 1, g_upd = 1, g_eb = 1, g_s64 = 0;
    for (int w = 0; w < W; w++) {
        bh_ba* ba = b->win[w];
        if (!ba->have_problem) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_optimize: window %d has no problem set", w);
        if (ba->ar) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_optimize: window %d is a sharded (multi-GPU) problem", w);
        if (ba->Cf > 0 && !ba->chol_dense)
            return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_optimize: window %d has %d free keyframes; the batch takes local windows (<= %d unknowns): use bh_ba_optimize", w, ba->Cf, 16 * CD_MAXT);
        if (((size_t)ba->Ea + 255) / 256 > FAN_WAVE_MAX || ((size_t)ba->Pa * BP_G + 255) / 256 > FAN_WAVE_MAX)
            return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_batch_optimize: window %d has %d observations / %d landmarks; the batch takes windows of at most %d: use bh_ba_optimize", w, ba->Ea, ba->Pa, 256 * FAN_WAVE_MAX);
        ba_view& V = hv[w];
        memset(&V, 0, sizeof(V));
        for (int k = 0; k < 2; k++) { V.cams[k] = (ba_cam*)ba->cams[k].p; V.pts[k] = (double*)ba->pts[k].p; V.err[k] = (double*)ba->err[k].p; V.chi_e[k] = (double*)ba->chi_e[k].p; }
        V.e_cam = (const int*)ba->e_cam.p; V.e_pt = (const int*)ba->e_pt.p; V.e_meas = (const double*)ba->e_meas.p; V.cslot = (const int*)ba->cslot.p;
        V.p_start = (const int*)ba->p_start.p; V.p_edges = (const int*)ba->p_edges.p; V.c_start = (const int*)ba->c_start.p; V.c_edges = (const int*)ba->c_edges.p;
        V.Hpl = (double*)ba->Hpl.p; V.Hll = (double*)ba->Hll.p; V.bl = (double*)ba->bl.p; V.Hpp = (double*)ba->Hpp.p; V.bp = (double*)ba->bp.p;
        V.pdiag = (double*)ba->pdiag.p; V.cdiag = (double*)ba->cdiag.p; V.Dinv = (double*)ba->Dinv.p; V.db = (double*)ba->db.p;
        V.S = (double*)ba->S.p; V.bs = (double*)ba->bs.p; V.dxp = (double*)ba->dxp.p; V.dxl = (double*)ba->dxl.p;
        V.fan = ba_fan_of(ba);
        V.pairs = (const int2*)ba->pairs.p; V.j_start = (const int*)ba->j_start.p; V.join = (const int2*)ba->join.p;
        V.state = (ba_state*)ba->state.p; V.hstate = b->hstates + 2 * w;
        V.C = ba->C; V.Cf = ba->Cf; V.Pa = ba->Pa; V.Ea = ba->Ea; V.npairs = ba->npairs; V.bw = ba->bw; V.n6 = 6 * ba->Cf; V.K = ba->K; V.pad = ba->schur_wave;
        if (ba->schur_wave) g_s64 = std::max(g_s64, (ba->npairs + 7) / 8 * 8);
        hcur[w] = ba->cur;
        ba->log.clear(); ba->iters_done = 0;
        g_lin = std::max(g_lin, (ba->Pa * BP_G + 255) / 256 + ba->Cf * std::max(1, ba->cam_parts));
        if (ba->cam_parts > 1) BH_HIP(ctx, hipMemsetAsync(V.fan.c_cnt, 0, sizeof(int) * ba->Cf, st));      // (a faulted launch could have left a count behind)
        g_pb = std::max(g_pb, (ba->Pa + 255) / 256);
        g_schur = std::max(g_schur, (ba->npairs + 7) / 8 * 8 + ba->Cf);
        g_upd = std::max(g_upd, (ba->Pa * BP_G + 255) / 256 + (ba->C + 255) / 256);
        g_eb = std::max(g_eb, ba_err_blocks(ba->Ea));
        if (iters_done) iters_done[w] = 0;
    }
    BH_TRY(bh_reserve(ctx, &b->views, sizeof(ba_view) * W)); BH_TRY(bh_reserve(ctx, &b->curs, sizeof(int) * W));
    BH_HIP(ctx, hipMemcpyAsync(b->views.p, hv, sizeof(ba_view) * W, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(b->curs.p, hcur, sizeof(int) * W, hipMemcpyHostToDevice, st));
    const ba_view* views = (const ba_view*)b->views.p;
    hipLaunchKernelGGL(k_bb_init, dim3((W + 63) / 64), dim3(64), 0, st, views, W, iters, verbose_errors ? 1 : 0);
    hipLaunchKernelGGL(k_bb_setcur, dim3((W + 63) / 64), dim3(64), 0, st, views, (const int*)b->curs.p, W);
    std::vector<char> done(W, 0);
    int active = 0;
    for (int w = 0; w < W; w++) { done[w] = (b->win[w]->Ea == 0 || iters <= 0) ? 1 : 0; active += !done[w]; b->win[w]->last_err_sel = 0; }
    int rc = BH_OK;
    // One round = one Levenberg trial of every live window.  Which kernels a window runs in a round is decided ON THE DEVICE (b_need_lin, b_done,
    // b_cur in its control block), so the host does not have to know a round's verdicts before it enqueues the next one: it stays ONE round
    // ahead (round r + 1 is in the queue while the verdicts of round r are polled; two pinned verdict slots per window, by round parity).  The
    // device never waits for the host's poll -> launch turnaround, which was ~25 us of an iteration (a tenth of a configs[2] window's, half of a
    // live window's); the one surplus round at the end finds every window done and returns at once.
    // The control steps of a round run in the last workgroup of the window's linearisation / error launch to arrive (six launches per round; BH_BA_BB_TAIL=0:
    // as launches of their own, eight per round - the A/B reference)
    const bool tail_off = getenv("BH_BA_BB_TAIL") && atoi(getenv("BH_BA_BB_TAIL")) == 0;
    auto enqueue_round = [&](int seq) {
        BH_LAUNCH(ctx, BH_K_BA_LINEARIZE, k_bb_linearize, dim3(g_lin, W8), dim3(256), st, views, W, huber_delta, tail_off ? 0 : seq);      // (windows that retry with a larger lambda skip it)
        if (tail_off) BH_LAUNCH(ctx, BH_K_BA_OTHER, k_bb_prep, dim3(1, W), dim3(1024), st, views);
        BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_bb_dinv, dim3(g_pb + 8, W8), dim3(256), st, views, W);
        BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_bb_schur, dim3(g_schur, W8), dim3(256), st, views, W);
        if (g_s64) BH_LAUNCH(ctx, BH_K_BA_SCHUR, k_bb_schur64, dim3(g_s64, W8), dim3(64), st, views, W);
        bh_time_begin(ctx, BH_K_BA_SOLVE);
        hipLaunchKernelGGL(k_bb_chol, dim3(W), dim3(1024), sizeof(double) * CD_LDS_DOUBLES, st, views);
        bh_time_end(ctx);
        BH_LAUNCH(ctx, BH_K_BA_UPDATE, k_bb_update, dim3(g_upd, W8), dim3(256), st, views, W);
        BH_LAUNCH(ctx, BH_K_BA_UPDATE, k_bb_err, dim3(g_eb, W8), dim3(256), st, views, W, huber_delta, tail_off ? 0 : seq);
        if (tail_off) BH_LAUNCH(ctx, BH_K_BA_OTHER, k_bb_decide, dim3(1, W), dim3(1024), st, views, seq);
    };
    const bool look_off = getenv("BH_BA_LOOKAHEAD") && atoi(getenv("BH_BA_LOOKAHEAD")) == 0;      // A/B switch: enqueue a round only after the last one's verdicts
    int seq_cur = 0;
    if (active > 0) { seq_cur = ++b->seq; enqueue_round(seq_cur); }
    std::vector<int> it_known(W, 0);                           // iterations a window had completed at the last verdict the host has read
    bool polled = false;
    for (int round = 0; active > 0 && round < iters * 10 + 2; round++) {
        int seq_next = 0;
        // no look-ahead behind a round that ends the run if its trials are accepted (every live window is in its last iteration): the surplus
        // round - eight launches that find everything done - would sit in front of the caller's next call; a retry there costs one host turn-around
        bool may_end = true;
        for (int w = 0; w < W && may_end; w++) if (!done[w] && it_known[w] + 1 < iters) may_end = false;
        if (!look_off && !may_end) { seq_next = ++b->seq; enqueue_round(seq_next); }
        BH_HIP(ctx, hipGetLastError());
        const int seq = seq_cur;
        for (int w = 0; w < W; w++) {
            if (done[w]) continue;
            ba_state* hp = b->hstates + 2 * w + (seq & 1);
            // Poll the pinned verdict for a bounded TIME (a spin count would turn a round that is merely slow - many or large windows, a
            // profiler attached, a busy host core - into an error); after that the stream itself decides: a successful synchronise means
            // the round has finished, so the verdict must be there; only a failing synchronise, or a verdict that is still missing after
            // it, is a fault.  No window may be read back after such an exit (the device may have flipped its ping-pong state).
            bool got = false;
            const auto tpoll = std::chrono::steady_clock::now();
            for (long spin = 0; !got; spin++) {
                if (__atomic_load_n(&hp->pad, __ATOMIC_ACQUIRE) == seq) { got = true; break; }
                if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - tpoll > std::chrono::milliseconds(200)) break;
            }
            if (!got) {
                const hipError_t e = hipStreamSynchronize(st);
                got = e == hipSuccess && __atomic_load_n(&hp->pad, __ATOMIC_ACQUIRE) == seq;
                if (!got) {
                    for (int v = 0; v < W; v++) b->win[v]->have_problem = 0;
                    if (e != hipSuccess) return bh_set_err(ctx, BH_ERR_HIP, "bh_ba_batch_optimize: %s", hipGetErrorString(e));
                    return bh_set_err(ctx, BH_ERR_HIP, "bh_ba_batch_optimize: no verdict from window %d", w);
                }
            }
            polled = true;
            ba_state hs; memcpy(&hs, hp, sizeof(hs));
            bh_ba* ba = b->win[w];
            if (hs.b_fail) { done[w] = 1; active--; rc = bh_set_err(ctx, BH_ERR_NUMERIC, "bh_ba_batch_optimize: non-finite Hessian at the initial estimate of window %d", w); continue; }
            ba->cur = hs.b_cur; ba->last_err_sel = hs.b_last_err;
            it_known[w] = hs.b_it;
            if (!hs.again) {
                ba->log.push_back(hs.accepted ? hs.chi_cur : hs.chi_trial); ba->log.push_back(hs.lambda);
                ba->log.push_back((double)hs.qmax); ba->log.push_back((double)hs.accepted);
                ba->iters_done = hs.b_it;
                if (iters_done) iters_done[w] = hs.b_it;
            }
            if (hs.b_done) { done[w] = 1; active--; }
        }
        if (seq_next) seq_cur = seq_next;
        else if (active > 0) { seq_cur = ++b->seq; enqueue_round(seq_cur); }
    }
    for (int w = 0; w < W; w++) b->win[w]->verbose_last = verbose_errors;
    if (!polled) BH_HIP(ctx, hipStreamSynchronize(st));        // no verdict was read (nothing to optimise): the pinned view staging must not be rewritten under its copy
    return rc;
}

// Test hook: solve one banded SPD system with the BA solver kernels (mode 0 = panel kernel, 1 = sliding-window kernel).
extern "C" int bh_ba_debug_band_solve(bh_ctx* ctx, const double* band, const double* rhs, int n, int bw, int mode, double* x, long long* clocks)
{
    if (!ctx || !band || !rhs || !x || n <= 0 || bw < 0 || bw >= n) return BH_ERR_INVALID;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t nb = sizeof(double) * (size_t)n * (bw + 1);
    BH_TRY(bh_reserve(ctx, &ctx->st_a, nb)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(double) * 2 * (size_t)n + sizeof(ba_state)));
    double* dA = (double*)ctx->st_a.p; double* db = (double*)ctx->st_b.p; double* dx = db + n; ba_state* dS = (ba_state*)(dx + n);
    ba_state hs; memset(&hs, 0, sizeof(hs)); hs.chol_ok = 1;
    BH_HIP(ctx, hipMemcpyAsync(dA, band, nb, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(db, rhs, sizeof(double) * n, hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(dS, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    (void)hipFuncSetAttribute((const void*)k_ba_chol_dense<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CD_LDS_DOUBLES));
    (void)hipFuncSetAttribute((const void*)k_ba_chol_dense<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CD_LDS_DOUBLES));
    bcr_dev bdv;
    if (mode == 5) {
        if (bw > 256) return bh_set_err(ctx, BH_ERR_INVALID, "band solve: block cyclic reduction needs bw <= 256");
        BH_TRY(bh_reserve(ctx, &ctx->st_c, BCR_SCRATCH_BYTES((n + 255) / 256)));
        BH_TRY(bcr_prepare(ctx, st, &bdv, n, ctx->st_c.p));
        BH_TRY(bcr_solve(ctx, st, &bdv, (const double*)dA, (const double*)db, n, bw, dx, dS));
    } else if (mode == 4) {
        if (n > 16 * CD_MAXT) return bh_set_err(ctx, BH_ERR_INVALID, "band solve: the on-chip dense kernel needs n <= %d", 16 * CD_MAXT);
        hipLaunchKernelGGL(k_ba_chol_dense<false>, dim3(1), dim3(1024), sizeof(double) * CD_LDS_DOUBLES, st, (const double*)dA, (const double*)db, n, bw, dx, dS,
                           (const bcr_elim*)nullptr, (double*)nullptr, (const double*)nullptr, (double*)nullptr);
    } else if (mode >= 1) {
        if ((bw + 15) / 16 > 16 || bw < 15) return bh_set_err(ctx, BH_ERR_INVALID, "band solve: the window kernel needs 15 <= bw <= 256");
        const int nblk = (n + 15) / 16;
        if (mode == 3 && ((bw + 15) / 16 != 16 || nblk < 40)) return bh_set_err(ctx, BH_ERR_INVALID, "band solve: two-way mode needs a 16-tile band and >= 40 block columns");
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(double) * SW_SCRATCH_DOUBLES(n, bw)));
        (void)hipFuncSetAttribute((const void*)k_ba_chol_sw, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void*)k_ba_chol<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        BH_TRY(sw_solve(ctx, st, dA, (const double*)db, n, bw, dx, dS, mode == 3 ? (nblk - 16) / 2 : 0, (double*)ctx->st_c.p));
    } else {
        int nbsel = 16;
        while (nbsel >= 8 && sizeof(double) * ((size_t)std::min(n, nbsel + bw) * (nbsel + 1) + 1100) > 150 * 1024) nbsel >>= 1;
        if (nbsel < 8) return bh_set_err(ctx, BH_ERR_INVALID, "band solve: band too wide");
        const size_t lds = sizeof(double) * ((size_t)std::min(n, nbsel + bw) * (nbsel + 1) + 1100);
        (void)hipFuncSetAttribute((const void*)k_ba_chol<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void*)k_ba_chol<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        (void)hipFuncSetAttribute((const void*)k_ba_chol<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        if (nbsel == 32) hipLaunchKernelGGL(k_ba_chol<32>, dim3(1), dim3(1024), lds, st, dA, (const double*)db, n, bw, dx, dS);
        else if (nbsel == 16) hipLaunchKernelGGL(k_ba_chol<16>, dim3(1), dim3(1024), lds, st, dA, (const double*)db, n, bw, dx, dS);
        else hipLaunchKernelGGL(k_ba_chol<8>, dim3(1), dim3(1024), lds, st, dA, (const double*)db, n, bw, dx, dS);
    }
    BH_HIP(ctx, hipGetLastError());
    BH_HIP(ctx, hipMemcpyAsync(x, dx, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipMemcpyAsync(&hs, dS, sizeof(hs), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    if (clocks) for (int i = 0; i < 8; i++) clocks[i] = hs.dbg[i];
    return hs.chol_ok ? BH_OK : bh_set_err(ctx, BH_ERR_NUMERIC, "band solve: not positive definite");
}

extern "C" int bh_ba_get_poses(bh_ba* ba, double* R, double* t, int mem)
{
    if (!ba || !R || !t) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (!ba->have_problem || ba->C == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    double* dR = R; double* dt = t;
    if (mem == BH_MEM_HOST) { BH_TRY(bh_reserve(ctx, &ctx->st_a, sizeof(double) * 12 * ba->C)); dR = (double*)ctx->st_a.p; dt = dR + 9 * (size_t)ba->C; }
    hipLaunchKernelGGL(k_ba_export, dim3((ba->C + 63) / 64), dim3(64), 0, ctx->stream, (const ba_cam*)ba->cams[ba->cur].p, ba->C, dR, dt);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(R, dR, sizeof(double) * 9 * ba->C, hipMemcpyDeviceToHost, ctx->stream));
        BH_HIP(ctx, hipMemcpyAsync(t, dt, sizeof(double) * 3 * ba->C, hipMemcpyDeviceToHost, ctx->stream));
        BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return BH_OK;
}

extern "C" int bh_ba_get_points(bh_ba* ba, double* points, int mem)
{
    if (!ba || !points) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (mem != BH_MEM_HOST) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_get_points: BH_MEM_HOST only (inactive points keep the caller's values)");
    if (!ba->have_problem || ba->Pa == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> tmp(3 * (size_t)ba->Pa);
    BH_HIP(ctx, hipMemcpyAsync(tmp.data(), ba->pts[ba->cur].p, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < ba->P; j++) if (ba->pt_slot[j] >= 0) for (int k = 0; k < 3; k++) points[3 * (size_t)j + k] = tmp[3 * (size_t)ba->pt_slot[j] + k];
    return BH_OK;
}

extern "C" int bh_ba_get_chi2(bh_ba* ba, double* chi2, int mem)
{
    if (!ba || !chi2) return BH_ERR_INVALID;
    bh_ctx* ctx = ba->ctx;
    if (mem != BH_MEM_HOST) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_get_chi2: BH_MEM_HOST only");
    if (!ba->have_problem || ba->E == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> err(2 * (size_t)ba->Ea);
    BH_HIP(ctx, hipMemcpyAsync(err.data(), ba->err[ba->last_err_sel].p, sizeof(double) * err.size(), hipMemcpyDeviceToHost, ctx->stream));
    BH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int e = 0; e < ba->E; e++) chi2[e] = err[2 * (size_t)e] * err[2 * (size_t)e] + err[2 * (size_t)e + 1] * err[2 * (size_t)e + 1];
    return BH_OK;
}

extern "C" int bh_ba_solver(bh_ba* ba)
{
    if (!ba || ba->C <= 0) return BH_ERR_INVALID;
    if (ba->dsol && ba->dsol->active) return 5;
    if (ba->chol_bcr) return 4;
    if (ba->chol_window) return ba->chol_split ? 3 : 2;
    return ba->chol_dense ? 1 : 0;
}

extern "C" int bh_ba_get_log(bh_ba* ba, double* log, int max_rows)
{
    if (!ba || !log) return BH_ERR_INVALID;
    const int rows = ba->iters_done < max_rows ? ba->iters_done : max_rows;
    for (int i = 0; i < rows * 4; i++) log[i] = ba->log[i];
    return BH_OK;
}
