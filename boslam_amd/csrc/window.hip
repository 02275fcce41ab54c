// window.hip — local-BA window selection and problem construction on the device.
//
// Replaces CovisibilityGraph.get_local_map(kf, flag_inner_outer=True, flag_with_input_kf=False) — slam/covisibility_graph.py:51-78 —
// and the graph walk of optimize_local — :188-218 — up to (not including) the add_pose / add_point / add_edge calls: which
// keyframes are optimised or held fixed, which map points are seen by more than one of them, and the (keyframe, point, pixel)
// edge list.  The reference walks Python sets (arbitrary order); here every set is enumerated in ascending id, the chain order
// is [kf] + inner + outer as in :203, and points are numbered in order of first appearance along that chain.
#include "bh_internal.h"
#include <limits.h>

#define WIN_T 1024

// ---- k_win_sets: inner / outer flags.  One thread per (neighbour kf1 of kf, neighbour kf2 of kf1) walk.
__global__ __launch_bounds__(256) void k_win_sets(const int* __restrict__ nbr_start, const int* __restrict__ nbr_id,
                                                  const int* __restrict__ nbr_common, int kf, int min_common,
                                                  unsigned char* __restrict__ inner_f, unsigned char* __restrict__ outer_f)
{
    const int a0 = nbr_start[kf], a1 = nbr_start[kf + 1];
    for (int e = a0 + (int)blockIdx.x; e < a1; e += gridDim.x) {
        if (nbr_common[e] < min_common) continue;                          // :65
        const int kf1 = nbr_id[e];
        if (threadIdx.x == 0) inner_f[kf1] = 1;                            // :66
        for (int e2 = nbr_start[kf1] + (int)threadIdx.x; e2 < nbr_start[kf1 + 1]; e2 += 256) {
            const int kf2 = nbr_id[e2];
            if (kf2 == kf) continue;                                       // :68
            int c = 0;                                                     // kf2kf_num_common_mps[(kf2, kf)], 0 when absent (defaultdict, :39)
            for (int e3 = a0; e3 < a1; e3++) if (nbr_id[e3] == kf2) { c = nbr_common[e3]; break; }
            if (c >= min_common) outer_f[kf2] = 1;                         // :69-70
        }
    }
}

__device__ __forceinline__ int win_block_excl_scan(int v, int* s_w, int* total)      // WIN_T threads; exclusive prefix of v
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < WIN_T / 64; w++) { const int c = s_w[w]; if (w < wv) base += c; tot += c; }
    *total = tot;
    return base + inc - v;
}

// ---- k_win_list: [kf] + inner (ascending) + outer minus inner (ascending), fixed flags (:190-201), per-keyframe observation offsets
__global__ __launch_bounds__(WIN_T) void k_win_list(const unsigned char* __restrict__ inner_f, const unsigned char* __restrict__ outer_f, int K,
                                                    int kf, const int* __restrict__ obs_start, int cap_kf, int* __restrict__ win_kf,
                                                    unsigned char* __restrict__ win_fixed, int* __restrict__ win_off, int* __restrict__ counts)
{
    __shared__ int s_w[WIN_T / 64];
    __shared__ int s_n[2];
    const int tid = threadIdx.x;
    int n_in = 0, n_out = 0;
    for (int pass = 0; pass < 2; pass++) {
        int base = 0;
        for (int c0 = 0; c0 < K; c0 += WIN_T) {
            const int i = c0 + tid;
            const bool f = i < K && i != kf && (pass == 0 ? inner_f[i] != 0 : (outer_f[i] != 0 && inner_f[i] == 0));
            int tot;
            const int pos = win_block_excl_scan(f ? 1 : 0, s_w, &tot);
            const int slot = 1 + (pass == 0 ? 0 : n_in) + base + pos;
            if (f && slot < cap_kf) win_kf[slot] = i;
            base += tot;
        }
        if (pass == 0) n_in = base; else n_out = base;
    }
    if (tid == 0) { s_n[0] = n_in; s_n[1] = n_out; }
    __syncthreads();
    const int nw = (n_in == 0 && n_out == 0) ? 0 : 1 + n_in + n_out;          // :191-194 "nothing to optimize"
    if (tid == 0) {
        if (nw > 0 && cap_kf > 0) win_kf[0] = kf;
        counts[0] = nw; counts[1] = n_in; counts[2] = n_out;
        counts[3] = nw > cap_kf ? 1 : 0;                                      // capacity flag
    }
    __syncthreads();
    const int nwc = nw < cap_kf ? nw : cap_kf;
    // fixed flags: outer empty -> kf fixed (:195); else kf free and outer fixed (:197-199); inner always free (:201)
    for (int i = tid; i < nwc; i += WIN_T) win_fixed[i] = (i == 0) ? (n_out == 0) : (i > n_in);
    __syncthreads();
    // observation offsets along the chain (sequential: a window holds tens of keyframes)
    if (tid == 0) {
        int off = 0;
        for (int i = 0; i < nwc; i++) { win_off[i] = off; off += obs_start[win_kf[i] + 1] - obs_start[win_kf[i]]; }
        win_off[nwc] = off;
    }
}

// ---- k_win_count: per map point, how many window keyframes see it (Counter, :205-208) and where it appears first along the chain
__global__ __launch_bounds__(256) void k_win_count(const int* __restrict__ win_kf, const int* __restrict__ win_off, const int* __restrict__ counts,
                                                   int cap_kf, const int* __restrict__ obs_start, const int* __restrict__ obs_mp,
                                                   int* __restrict__ cnt, int* __restrict__ first)
{
    const int nw = counts[0] < cap_kf ? counts[0] : cap_kf;
    for (int w = blockIdx.x; w < nw; w += gridDim.x) {
        const int k = win_kf[w], o0 = obs_start[k], n = obs_start[k + 1] - o0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int mp = obs_mp[o0 + i];
            atomicAdd(&cnt[mp], 1);
            atomicMin(&first[mp], win_off[w] + i);
        }
    }
}

// ---- k_win_emit: ordered compaction of the observations whose map point is seen more than once (:211-216); points numbered by
// first appearance.  One workgroup walks the chain (a window has a few 10^4 observations).
__global__ __launch_bounds__(WIN_T) void k_win_emit(const int* __restrict__ win_kf, const int* __restrict__ win_off, const int* __restrict__ counts,
                                                    int cap_kf, const int* __restrict__ obs_start, const int* __restrict__ obs_mp,
                                                    const long long* __restrict__ obs_px, const int* __restrict__ cnt, const int* __restrict__ first,
                                                    int* __restrict__ slot_of, int cap_pt, int cap_e, int* __restrict__ pt_mp,
                                                    int* __restrict__ edge_pose, int* __restrict__ edge_mp, long long* __restrict__ edge_px,
                                                    int* __restrict__ counts2)
{
    __shared__ int s_w[WIN_T / 64];
    const int tid = threadIdx.x;
    const int nw = counts[0] < cap_kf ? counts[0] : cap_kf;
    const int T = nw > 0 ? win_off[nw] : 0;
    int ebase = 0, pbase = 0;
    for (int c0 = 0; c0 < T; c0 += WIN_T) {
        const int pos = c0 + tid;
        bool keep = false, isfirst = false;
        int w = 0, mp = 0, src = 0;
        if (pos < T) {
            int lo = 0, hi = nw - 1;                                        // which window keyframe holds chain position pos
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (win_off[mid] <= pos) lo = mid; else hi = mid - 1; }
            w = lo; src = obs_start[win_kf[w]] + (pos - win_off[w]); mp = obs_mp[src];
            keep = cnt[mp] > 1;                                             // :212
            isfirst = keep && first[mp] == pos;                             // :213-215
        }
        int etot, ptot;
        const int e = ebase + win_block_excl_scan(keep ? 1 : 0, s_w, &etot);
        const int p = pbase + win_block_excl_scan(isfirst ? 1 : 0, s_w, &ptot);
        if (keep && e < cap_e) { edge_pose[e] = w; edge_mp[e] = mp; edge_px[2 * (size_t)e] = obs_px[2 * (size_t)src]; edge_px[2 * (size_t)e + 1] = obs_px[2 * (size_t)src + 1]; }
        if (isfirst) { slot_of[mp] = p; if (p < cap_pt) pt_mp[p] = mp; }
        ebase += etot; pbase += ptot;
    }
    if (tid == 0) { counts2[0] = pbase; counts2[1] = ebase; counts2[2] = (pbase > cap_pt || ebase > cap_e) ? 1 : 0; }
}

__global__ void k_win_edges(const int* __restrict__ counts2, int cap_e, const int* __restrict__ slot_of, int* __restrict__ edge_mp_to_point)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int ne = counts2[1] < cap_e ? counts2[1] : cap_e;
    if (e < ne) edge_mp_to_point[e] = slot_of[edge_mp_to_point[e]];
}

extern "C" int bh_ba_window_problem(bh_ctx* ctx, int K, const int* nbr_start, const int* nbr_id, const int* nbr_common,
                                    const int* obs_start, const int* obs_mp, const long long* obs_px, int n_mp, int kf, int min_common,
                                    int cap_kf, int* win_kf, uint8_t* win_fixed, int cap_pt, int* pt_mp, int cap_e, int* edge_pose,
                                    int* edge_point, long long* edge_px, int* counts /* [5]: n_window, n_inner, n_outer, n_points, n_edges */)
{
    if (!ctx) return BH_ERR_INVALID;
    if (K <= 0 || kf < 0 || kf >= K || n_mp < 0 || !nbr_start || !obs_start || !win_kf || !win_fixed || !pt_mp || !edge_pose || !edge_point ||
        !edge_px || !counts || cap_kf < 1 || cap_pt < 0 || cap_e < 0)
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_window_problem: bad argument");
    const int n_nbr = nbr_start[K], n_obs = obs_start[K];
    if (n_nbr < 0 || n_obs < 0 || (n_nbr && (!nbr_id || !nbr_common)) || (n_obs && (!obs_mp || !obs_px)))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_window_problem: bad CSR");
    for (int i = 0; i < n_nbr; i++) if (nbr_id[i] < 0 || nbr_id[i] >= K) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_window_problem: neighbour id out of range");
    for (int i = 0; i < n_obs; i++) if (obs_mp[i] < 0 || obs_mp[i] >= n_mp) return bh_set_err(ctx, BH_ERR_INVALID, "bh_ba_window_problem: map point id out of range");
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // device staging: graph, observations, scratch, outputs
    const size_t b_graph = sizeof(int) * ((size_t)2 * (K + 1) + 2 * (size_t)n_nbr + (size_t)n_obs) + 64;
    const size_t b_px = sizeof(long long) * 2 * ((size_t)n_obs + 1);
    const size_t b_scr = (size_t)2 * K + sizeof(int) * ((size_t)3 * n_mp + cap_kf + 2 + 16) + 64;
    const size_t b_out = (size_t)cap_kf * 5 + sizeof(int) * ((size_t)cap_pt + 2 * (size_t)cap_e) + sizeof(long long) * 2 * ((size_t)cap_e + 1) + 64;
    BH_TRY(bh_reserve(ctx, &ctx->st_a, b_graph)); BH_TRY(bh_reserve(ctx, &ctx->st_b, b_px)); BH_TRY(bh_reserve(ctx, &ctx->st_c, b_scr));
    BH_TRY(bh_reserve(ctx, &ctx->st_d, b_out));
    int* d_ns = (int*)ctx->st_a.p; int* d_os = d_ns + (K + 1); int* d_nid = d_os + (K + 1); int* d_nc = d_nid + n_nbr; int* d_omp = d_nc + n_nbr;
    long long* d_opx = (long long*)ctx->st_b.p;
    int* d_cnt = (int*)ctx->st_c.p; int* d_first = d_cnt + n_mp; int* d_slot = d_first + n_mp; int* d_woff = d_slot + n_mp; int* d_counts = d_woff + cap_kf + 2;
    unsigned char* d_inner = (unsigned char*)(d_counts + 16); unsigned char* d_outer = d_inner + K;
    long long* d_epx = (long long*)ctx->st_d.p; int* d_wkf = (int*)(d_epx + 2 * ((size_t)cap_e + 1)); int* d_ptmp = d_wkf + cap_kf;
    int* d_epose = d_ptmp + cap_pt; int* d_ept = d_epose + cap_e; unsigned char* d_wfix = (unsigned char*)(d_ept + cap_e);
    BH_HIP(ctx, hipMemcpyAsync(d_ns, nbr_start, sizeof(int) * (K + 1), hipMemcpyHostToDevice, st));
    BH_HIP(ctx, hipMemcpyAsync(d_os, obs_start, sizeof(int) * (K + 1), hipMemcpyHostToDevice, st));
    if (n_nbr) { BH_HIP(ctx, hipMemcpyAsync(d_nid, nbr_id, sizeof(int) * n_nbr, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_nc, nbr_common, sizeof(int) * n_nbr, hipMemcpyHostToDevice, st)); }
    if (n_obs) { BH_HIP(ctx, hipMemcpyAsync(d_omp, obs_mp, sizeof(int) * n_obs, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_opx, obs_px, sizeof(long long) * 2 * n_obs, hipMemcpyHostToDevice, st)); }
    BH_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)n_mp, st));
    BH_HIP(ctx, hipMemsetAsync(d_first, 0x7f, sizeof(int) * (size_t)n_mp, st));
    BH_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int) * 16, st));
    BH_HIP(ctx, hipMemsetAsync(d_inner, 0, (size_t)2 * K, st));
    const int deg = nbr_start[kf + 1] - nbr_start[kf];
    if (deg > 0) hipLaunchKernelGGL(k_win_sets, dim3(deg < 1024 ? deg : 1024), dim3(256), 0, st, d_ns, d_nid, d_nc, kf, min_common, d_inner, d_outer);
    hipLaunchKernelGGL(k_win_list, dim3(1), dim3(WIN_T), 0, st, d_inner, d_outer, K, kf, d_os, cap_kf, d_wkf, d_wfix, d_woff, d_counts);
    hipLaunchKernelGGL(k_win_count, dim3(cap_kf < 256 ? cap_kf : 256), dim3(256), 0, st, d_wkf, d_woff, d_counts, cap_kf, d_os, d_omp, d_cnt, d_first);
    hipLaunchKernelGGL(k_win_emit, dim3(1), dim3(WIN_T), 0, st, d_wkf, d_woff, d_counts, cap_kf, d_os, d_omp, d_opx, d_cnt, d_first, d_slot, cap_pt,
                       cap_e, d_ptmp, d_epose, d_ept, d_epx, d_counts + 8);
    if (cap_e > 0) hipLaunchKernelGGL(k_win_edges, dim3((cap_e + 255) / 256), dim3(256), 0, st, d_counts + 8, cap_e, d_slot, d_ept);
    BH_HIP(ctx, hipGetLastError());
    int hc[16];
    BH_HIP(ctx, hipMemcpyAsync(hc, d_counts, sizeof(hc), hipMemcpyDeviceToHost, st));
    BH_HIP(ctx, hipStreamSynchronize(st));
    counts[0] = hc[0]; counts[1] = hc[1]; counts[2] = hc[2]; counts[3] = hc[8]; counts[4] = hc[9];
    if (hc[3] || hc[10]) return bh_set_err(ctx, BH_ERR_CAPACITY, "bh_ba_window_problem: output capacity too small (%d keyframes, %d points, %d edges needed)", hc[0], hc[8], hc[9]);
    const int nw = hc[0], np = hc[8], ne = hc[9];
    if (nw) { BH_HIP(ctx, hipMemcpyAsync(win_kf, d_wkf, sizeof(int) * nw, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(win_fixed, d_wfix, nw, hipMemcpyDeviceToHost, st)); }
    if (np) BH_HIP(ctx, hipMemcpyAsync(pt_mp, d_ptmp, sizeof(int) * np, hipMemcpyDeviceToHost, st));
    if (ne) {
        BH_HIP(ctx, hipMemcpyAsync(edge_pose, d_epose, sizeof(int) * ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(edge_point, d_ept, sizeof(int) * ne, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(edge_px, d_epx, sizeof(long long) * 2 * ne, hipMemcpyDeviceToHost, st));
    }
    BH_HIP(ctx, hipStreamSynchronize(st));
    return BH_OK;
}
