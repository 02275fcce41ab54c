// track.hip — one tracking step of slam/tracking.py on the device, batched over (frame, reference) pairs.
//
// Tracker._track_wrt_refkf (slam/tracking.py:54-89) and the second half of Tracker._track_local_map (:121-136) are the same chain:
//     matches = matcher.match(frame.des, ref_des)                      cross-check Hamming matcher
//     matches = [m for m in matches if m.distance < d_hamming_max]     (<= in the local-map leg, :121)
//     need >= min_matches (min_matches_cg = 15, :62 / 15, :123)
//     CamOnlyBA(frame.kp_arr[inds_f], ref_points[inds_ref], cam)       pose-only BA, identity start
//     need >= 15 inliers (:83, :136)
// Round 2 ran the two kernels with the filter / index extraction / gathers as host Python between them.  Here the
// whole chain is enqueued on the context's stream: k_match -> k_track_gather (distance filter with ordered compaction, gather of the
// int64 pixels and f64 points into the pose-BA layout) -> k_pose_ba -> k_track_finish (inlier count, verdict).  No host hop.
#include "bh_internal.h"

// One workgroup per pair: ordered compaction of the raw match list by the distance test, gather of the BA inputs.
__global__ __launch_bounds__(256) void k_track_gather(const int* __restrict__ rq, const int* __restrict__ rt, const int* __restrict__ rd,
                                                      const int* __restrict__ rn, int raw_stride, int d_max, int inclusive, int min_matches,
                                                      const long long* __restrict__ q_xy, int q_stride, const double* __restrict__ t_pts, int t_stride,
                                                      int* __restrict__ qidx, int* __restrict__ tidx, int* __restrict__ dist, int* __restrict__ n_match,
                                                      int out_stride, double* __restrict__ pix, double* __restrict__ pts, int* __restrict__ n_ba)
{
    __shared__ int s_wsum[4], s_base;
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(rn[b], raw_stride);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        int q = 0, t = 0, d = 0;
        bool keep = false;
        if (i < n) {
            q = rq[b * raw_stride + i]; t = rt[b * raw_stride + i]; d = rd[b * raw_stride + i];
            keep = inclusive ? d <= d_max : d < d_max;
        }
        const unsigned long long bm = __ballot(keep);
        const int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wv] = __popcll(bm);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; w++) off += s_wsum[w];
        const int pos = off + before;
        if (keep && pos < out_stride) {
            const size_t o = b * out_stride + pos;
            qidx[o] = q; tidx[o] = t; dist[o] = d;
            const long long* xy = q_xy + ((size_t)b * q_stride + q) * 2;
            pix[2 * o] = (double)xy[0]; pix[2 * o + 1] = (double)xy[1];
            const double* p = t_pts + ((size_t)b * t_stride + t) * 3;
            pts[3 * o] = p[0]; pts[3 * o + 1] = p[1]; pts[3 * o + 2] = p[2];
        }
        __syncthreads();
        if (tid == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        const int m = min(s_base, out_stride);
        n_match[b] = m;
        n_ba[b] = m >= min_matches && m > 0 ? m : 0;             // too few matches: CamOnlyBA is not called (tracking.py:58-65, :123)
    }
}

__global__ __launch_bounds__(256) void k_track_finish(const int* __restrict__ n_match, const int* __restrict__ n_ba, const uint8_t* __restrict__ inlier,
                                                      const int* __restrict__ ba_status, int out_stride, int min_inliers,
                                                      int* __restrict__ n_inlier, int* __restrict__ status)
{
    __shared__ int s_cnt;
    const size_t b = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int n = n_ba[b];
    int c = 0;
    for (int i = tid; i < n; i += 256) c += inlier[b * out_stride + i] ? 1 : 0;
    for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((tid & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (tid == 0) {
        n_inlier[b] = s_cnt;
        status[b] = n == 0 ? BH_TRACK_FEW_MATCHES : (ba_status[b] != BH_OK ? BH_TRACK_BA_FAILED : (s_cnt < min_inliers ? BH_TRACK_FEW_INLIERS : BH_TRACK_OK));
    }
}

extern "C" int bh_track_refkf(bh_ctx* ctx, const uint8_t* q_desc, const int* nq, const int64_t* q_xy, int q_stride,
                              const uint8_t* t_desc, const int* nt, const double* t_pts, int t_stride, int B,
                              int d_max, int inclusive, int min_matches, int min_inliers,
                              const double K[5], int iters, double huber_delta, double chi2_thr,
                              int* qidx, int* tidx, int* dist, int* n_match, double* R, double* t, uint8_t* inlier, int* n_inlier,
                              int* status, int out_stride, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!q_desc || !nq || !q_xy || !t_desc || !nt || !t_pts || !K || !qidx || !tidx || !dist || !n_match || !R || !t || !inlier || !n_inlier || !status ||
        B < 0 || q_stride < 1 || t_stride < 1 || out_stride < 1 || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_track_refkf: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t no = (size_t)B * out_stride;
    // intermediates: raw matches (3 ints per slot + counts), gathered pixels / points, per-edge chi2, BA status / sizes
    const size_t w_raw = sizeof(int) * (3 * no + B), w_pix = sizeof(double) * 2 * no, w_pts = sizeof(double) * 3 * no, w_chi = sizeof(double) * no;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    BH_TRY(bh_reserve(ctx, &ctx->tr_w, al(w_raw) + al(w_pix) + al(w_pts) + al(w_chi) + al(sizeof(int) * 2 * B)));
    char* wp = (char*)ctx->tr_w.p;
    int* rq = (int*)wp; int* rt = rq + no; int* rd = rt + no; int* rn = rd + no; wp += al(w_raw);
    double* pix = (double*)wp; wp += al(w_pix);
    double* pts = (double*)wp; wp += al(w_pts);
    double* chi = (double*)wp; wp += al(w_chi);
    int* n_ba = (int*)wp; int* ba_st = n_ba + B;
    const uint8_t* d_q = q_desc; const uint8_t* d_t = t_desc; const int* d_nq = nq; const int* d_nt = nt; const long long* d_xy = (const long long*)q_xy; const double* d_tp = t_pts;
    int *d_qi = qidx, *d_ti = tidx, *d_di = dist, *d_nm = n_match, *d_ni = n_inlier, *d_st = status;
    double *d_R = R, *d_tt = t; uint8_t* d_in = inlier;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (nq[b] < 0 || nt[b] < 0 || nq[b] > q_stride || nt[b] > t_stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_track_refkf: row count outside [0, stride]");
        const size_t qb = (size_t)B * q_stride * 32, tb = (size_t)B * t_stride * 32, xb = sizeof(long long) * 2 * (size_t)B * q_stride, pb = sizeof(double) * 3 * (size_t)B * t_stride;
        const size_t ob = al(sizeof(int) * no) * 3 + al(sizeof(int) * B) * 3 + al(sizeof(double) * 12 * B) + al(no);
        BH_TRY(bh_reserve(ctx, &ctx->tr_in, al(qb) + al(tb) + al(xb) + al(pb) + al(sizeof(int) * 2 * B) + ob));
        char* ip = (char*)ctx->tr_in.p;
        BH_HIP(ctx, hipMemcpyAsync(ip, q_desc, qb, hipMemcpyHostToDevice, st)); d_q = (const uint8_t*)ip; ip += al(qb);
        BH_HIP(ctx, hipMemcpyAsync(ip, t_desc, tb, hipMemcpyHostToDevice, st)); d_t = (const uint8_t*)ip; ip += al(tb);
        BH_HIP(ctx, hipMemcpyAsync(ip, q_xy, xb, hipMemcpyHostToDevice, st)); d_xy = (const long long*)ip; ip += al(xb);
        BH_HIP(ctx, hipMemcpyAsync(ip, t_pts, pb, hipMemcpyHostToDevice, st)); d_tp = (const double*)ip; ip += al(pb);
        BH_HIP(ctx, hipMemcpyAsync(ip, nq, sizeof(int) * B, hipMemcpyHostToDevice, st)); d_nq = (const int*)ip;
        BH_HIP(ctx, hipMemcpyAsync(ip + sizeof(int) * B, nt, sizeof(int) * B, hipMemcpyHostToDevice, st)); d_nt = d_nq + B; ip += al(sizeof(int) * 2 * B);
        d_qi = (int*)ip; ip += al(sizeof(int) * no); d_ti = (int*)ip; ip += al(sizeof(int) * no); d_di = (int*)ip; ip += al(sizeof(int) * no);
        d_nm = (int*)ip; ip += al(sizeof(int) * B); d_ni = (int*)ip; ip += al(sizeof(int) * B); d_st = (int*)ip; ip += al(sizeof(int) * B);
        d_R = (double*)ip; d_tt = d_R + 9 * (size_t)B; ip += al(sizeof(double) * 12 * B);
        d_in = (uint8_t*)ip;
    }
    BH_TRY(bh_match_hamming_crosscheck(ctx, d_q, d_nq, q_stride, d_t, d_nt, t_stride, B, rq, rt, rd, rn, out_stride, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_MATCH, k_track_gather, dim3(B), dim3(256), st, (const int*)rq, (const int*)rt, (const int*)rd, (const int*)rn, out_stride, d_max, inclusive ? 1 : 0,
              min_matches, d_xy, q_stride, d_tp, t_stride, d_qi, d_ti, d_di, d_nm, out_stride, pix, pts, n_ba);
    BH_TRY(bh_pose_only_ba(ctx, pix, pts, n_ba, B, out_stride, K, iters, huber_delta, chi2_thr, d_R, d_tt, chi, d_in, ba_st, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_POSE_BA, k_track_finish, dim3(B), dim3(256), st, (const int*)d_nm, (const int*)n_ba, (const uint8_t*)d_in, (const int*)ba_st, out_stride, min_inliers, d_ni, d_st);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(qidx, d_qi, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(tidx, d_ti, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(dist, d_di, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_match, d_nm, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_inlier, d_ni, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(status, d_st, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(R, d_R, sizeof(double) * 9 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(t, d_tt, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(inlier, d_in, no, hipMemcpyDeviceToHost, st));
        return bh_ctx_status(ctx);                               // synchronises; the matcher's capacity flags
    }
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ the local-map half of Tracker.update
// Tracker._track_local_map (slam/tracking.py:91-162) with no host hop: visibility gate over the local map's (keyframe, map point)
// candidates -> ordered gather of the survivors' medoid descriptors and points into the matcher's train layout -> the chain of
// bh_track_refkf with the `<=` distance test of :121 -> the candidate index of every match (what ids_matching_mps[inds] needs).

// see_vector of a frame whose pose has just been set (camera.py:26-29 on a fresh Frame: R (0, 0, 1) normalised), and the candidate count
// the gate works on: Tracker.update runs _track_local_map only `if is_ok` (slam/tracking.py:205-212) - a frame whose first stage did not
// return BH_TRACK_OK gets zero candidates (nothing is gated, matched or optimised for it) and the verdict BH_TRACK_SKIPPED
__global__ void k_tlm_prep(const double* __restrict__ R, double* __restrict__ see, const int* __restrict__ stage1, const int* __restrict__ M,
                           int* __restrict__ M_eff, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool run = !stage1 || stage1[b] == BH_TRACK_OK;
    M_eff[b] = run ? M[b] : 0;
    if (see) {
        const double x = R[9 * b + 2], y = R[9 * b + 5], z = R[9 * b + 8];
        const double n = sqrt(x * x + y * y + z * z);
        const bool ok = run && n > 0.0;                            // a skipped frame's R may still be all zero
        see[3 * b] = ok ? x / n : 0.0; see[3 * b + 1] = ok ? y / n : 0.0; see[3 * b + 2] = ok ? z / n : 1.0;
    }
}

__global__ __launch_bounds__(256) void k_tlm_gather(const int* __restrict__ gate_idx, const int* __restrict__ n_gate, int m_stride,
                                                    const uint8_t* __restrict__ mp_feat, const double* __restrict__ mp_pts,
                                                    uint8_t* __restrict__ g_desc, double* __restrict__ g_pts)
{
    const size_t b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(n_gate[b], m_stride)) return;
    const size_t j = b * m_stride + (size_t)gate_idx[b * m_stride + i], o = b * m_stride + i;
    const uint4* src = reinterpret_cast<const uint4*>(mp_feat + j * 32);
    uint4* dst = reinterpret_cast<uint4*>(g_desc + o * 32);
    dst[0] = src[0]; dst[1] = src[1];
    g_pts[3 * o] = mp_pts[3 * j]; g_pts[3 * o + 1] = mp_pts[3 * j + 1]; g_pts[3 * o + 2] = mp_pts[3 * j + 2];
}

__global__ __launch_bounds__(256) void k_tlm_finish(const int* __restrict__ gate_idx, const int* __restrict__ n_gate, int m_stride,
                                                    const int* __restrict__ tidx, const int* __restrict__ n_match, int out_stride,
                                                    const int* __restrict__ stage1, int* __restrict__ cand, int* __restrict__ status)
{
    const size_t b = blockIdx.x;
    const int n = min(n_match[b], out_stride);
    for (int i = threadIdx.x; i < n; i += 256) cand[b * out_stride + i] = gate_idx[b * m_stride + tidx[b * out_stride + i]];
    if (threadIdx.x == 0) {
        if (stage1 && stage1[b] != BH_TRACK_OK) status[b] = BH_TRACK_SKIPPED;         // `if is_ok:` around the call (:205-212)
        else if (n_gate[b] == 0) status[b] = BH_TRACK_NO_CANDIDATES;                  // `if len(feats) == 0: return False` (:116-117)
    }
}

extern "C" int bh_track_local_map(bh_ctx* ctx, const uint8_t* q_desc, const int* nq, const int64_t* q_xy, int q_stride,
                                  const double* R_in, const double* t_in, const double* see, const int* stage1_status,
                                  const double* mp_pts, const double* mp_normals, const uint8_t* mp_feat, const int* M, int m_stride, int B,
                                  int width, int height, double cos_thr, int d_max, int min_matches, int min_inliers,
                                  const double K[5], int iters, double huber_delta, double chi2_thr,
                                  int* gate_idx, int* n_gate, int* qidx, int* tidx, int* cand, int* dist, int* n_match,
                                  double* R, double* t, uint8_t* inlier, int* n_inlier, int* status, int out_stride, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!q_desc || !nq || !q_xy || !R_in || !t_in || !mp_pts || !mp_normals || !mp_feat || !M || !K || !gate_idx || !n_gate || !qidx || !tidx || !cand ||
        !dist || !n_match || !R || !t || !inlier || !n_inlier || !status || B < 0 || q_stride < 1 || m_stride < 1 || out_stride < 1 ||
        (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_track_local_map: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t nm = (size_t)B * m_stride, no = (size_t)B * out_stride, nqs = (size_t)B * q_stride;
    // workspace: gathered train descriptors / points, the see vectors when the caller leaves them to the pose
    BH_TRY(bh_reserve(ctx, &ctx->tl_w, al(nm * 32) + al(sizeof(double) * 3 * nm) + al(sizeof(double) * 3 * B) + al(sizeof(int) * B)));
    char* wp = (char*)ctx->tl_w.p;
    uint8_t* g_desc = (uint8_t*)wp; wp += al(nm * 32);
    double* g_pts = (double*)wp; wp += al(sizeof(double) * 3 * nm);
    double* d_see_w = (double*)wp; wp += al(sizeof(double) * 3 * B);
    int* d_Meff = (int*)wp;
    const int* d_s1 = stage1_status;
    const uint8_t* d_q = q_desc; const int* d_nq = nq; const long long* d_xy = (const long long*)q_xy; const double* d_Ri = R_in; const double* d_ti = t_in;
    const double* d_see = see; const double* d_mp = mp_pts; const double* d_mn = mp_normals; const uint8_t* d_mf = mp_feat; const int* d_M = M;
    int *d_gi = gate_idx, *d_ng = n_gate, *d_qi = qidx, *d_ti2 = tidx, *d_ca = cand, *d_di = dist, *d_nm = n_match, *d_ni = n_inlier, *d_st = status;
    double *d_R = R, *d_t = t; uint8_t* d_in = inlier;
    if (mem == BH_MEM_HOST) {
        for (int b = 0; b < B; b++) if (nq[b] < 0 || nq[b] > q_stride || M[b] < 0 || M[b] > m_stride) return bh_set_err(ctx, BH_ERR_INVALID, "bh_track_local_map: count outside [0, stride]");
        const size_t in_b = al(nqs * 32) + al(sizeof(long long) * 2 * nqs) + al(sizeof(int) * 3 * B) + al(sizeof(double) * 15 * B) + 2 * al(sizeof(double) * 3 * nm) + al(nm * 32);
        const size_t out_b = al(sizeof(int) * nm) + 4 * al(sizeof(int) * no) + 4 * al(sizeof(int) * B) + al(sizeof(double) * 12 * B) + al(no);
        BH_TRY(bh_reserve(ctx, &ctx->tl_in, in_b + out_b));
        char* ip = (char*)ctx->tl_in.p;
        auto up = [&](const void* src, size_t bytes) -> char* { char* d = ip; ip += al(bytes); return hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess ? d : nullptr; };
        char* p;
        if (!(p = up(q_desc, nqs * 32))) return bh_set_err(ctx, BH_ERR_HIP, "bh_track_local_map: upload failed"); d_q = (const uint8_t*)p;
        if (!(p = up(q_xy, sizeof(long long) * 2 * nqs))) return bh_set_err(ctx, BH_ERR_HIP, "bh_track_local_map: upload failed"); d_xy = (const long long*)p;
        {   // nq, M and the first stage's verdicts side by side
            char* d = ip; ip += al(sizeof(int) * 3 * B);
            BH_HIP(ctx, hipMemcpyAsync(d, nq, sizeof(int) * B, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d + sizeof(int) * B, M, sizeof(int) * B, hipMemcpyHostToDevice, st));
            d_nq = (const int*)d; d_M = d_nq + B;
            if (stage1_status) { BH_HIP(ctx, hipMemcpyAsync(d + 2 * sizeof(int) * B, stage1_status, sizeof(int) * B, hipMemcpyHostToDevice, st)); d_s1 = d_M + B; }
        }
        {   // R, t, see
            double* d = (double*)ip; ip += al(sizeof(double) * 15 * B);
            BH_HIP(ctx, hipMemcpyAsync(d, R_in, sizeof(double) * 9 * B, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d + 9 * (size_t)B, t_in, sizeof(double) * 3 * B, hipMemcpyHostToDevice, st));
            if (see) BH_HIP(ctx, hipMemcpyAsync(d + 12 * (size_t)B, see, sizeof(double) * 3 * B, hipMemcpyHostToDevice, st));
            d_Ri = d; d_ti = d + 9 * (size_t)B; if (see) d_see = d + 12 * (size_t)B;
        }
        if (!(p = up(mp_pts, sizeof(double) * 3 * nm))) return bh_set_err(ctx, BH_ERR_HIP, "bh_track_local_map: upload failed"); d_mp = (const double*)p;
        if (!(p = up(mp_normals, sizeof(double) * 3 * nm))) return bh_set_err(ctx, BH_ERR_HIP, "bh_track_local_map: upload failed"); d_mn = (const double*)p;
        if (!(p = up(mp_feat, nm * 32))) return bh_set_err(ctx, BH_ERR_HIP, "bh_track_local_map: upload failed"); d_mf = (const uint8_t*)p;
        d_gi = (int*)ip; ip += al(sizeof(int) * nm);
        d_qi = (int*)ip; ip += al(sizeof(int) * no); d_ti2 = (int*)ip; ip += al(sizeof(int) * no); d_ca = (int*)ip; ip += al(sizeof(int) * no); d_di = (int*)ip; ip += al(sizeof(int) * no);
        d_ng = (int*)ip; ip += al(sizeof(int) * B); d_nm = (int*)ip; ip += al(sizeof(int) * B); d_ni = (int*)ip; ip += al(sizeof(int) * B); d_st = (int*)ip; ip += al(sizeof(int) * B);
        d_R = (double*)ip; d_t = d_R + 9 * (size_t)B; ip += al(sizeof(double) * 12 * B);
        d_in = (uint8_t*)ip;
    }
    if (!d_see || d_s1) {                                          // the frame's see_vector follows from the pose the first half has just set
        hipLaunchKernelGGL(k_tlm_prep, dim3((B + 63) / 64), dim3(64), 0, st, d_Ri, d_see ? (double*)nullptr : d_see_w, d_s1, d_M, d_Meff, B);
        if (!d_see) d_see = d_see_w;
        d_M = d_Meff;
    }
    BH_TRY(bh_visibility_gate(ctx, d_mp, d_mn, d_M, B, m_stride, d_Ri, d_ti, d_see, K[0], K[2], K[3], width, height, cos_thr, d_gi, d_ng, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_GATE, k_tlm_gather, dim3((m_stride + 255) / 256, B), dim3(256), st, (const int*)d_gi, (const int*)d_ng, m_stride, d_mf, d_mp, g_desc, g_pts);
    BH_TRY(bh_track_refkf(ctx, d_q, d_nq, (const int64_t*)d_xy, q_stride, g_desc, d_ng, g_pts, m_stride, B, d_max, 1, min_matches, min_inliers, K, iters, huber_delta, chi2_thr,
                          d_qi, d_ti2, d_di, d_nm, d_R, d_t, d_in, d_ni, d_st, out_stride, BH_MEM_DEVICE));
    BH_LAUNCH(ctx, BH_K_GATE, k_tlm_finish, dim3(B), dim3(256), st, (const int*)d_gi, (const int*)d_ng, m_stride, (const int*)d_ti2, (const int*)d_nm, out_stride, d_s1, d_ca, d_st);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(gate_idx, d_gi, sizeof(int) * nm, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_gate, d_ng, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(qidx, d_qi, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(tidx, d_ti2, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(cand, d_ca, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(dist, d_di, sizeof(int) * no, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_match, d_nm, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_inlier, d_ni, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(status, d_st, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(R, d_R, sizeof(double) * 9 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(t, d_t, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(inlier, d_in, no, hipMemcpyDeviceToHost, st));
        return bh_ctx_status(ctx);
    }
    return BH_OK;
}

