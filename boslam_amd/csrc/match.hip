// match.hip — brute-force Hamming matcher with OpenCV's cross-check rule, batched over independent pairs (gfx950).
//
// Replaces cv2.BFMatcher_create(cv2.NORM_HAMMING, crossCheck=True).match(q, t) — reference slam/tracking.py:45,56,121,
// slam/local_mapping.py:21, slam/covisibility_graph.py:34, experiments/pnp_two_way_tracking.py:11,42.
//
// One 512-lane workgroup per (query, train) pair.  The search "for every train descriptor, the closest query, first one on a tie"
// is the O(nq nt) part and it is GEMM-shaped: with the descriptors' bits as 0/1 bytes, hamming(q_i, t_j) = |q_i| + |t_j| - 2 q_i.t_j,
// and the dot products are one v_mfma_i32_32x32x32_i8 per 32 queries x 32 trains x 32 bits (exact in int32).  A wave keeps the
// fragments of FOUR train tiles (128 trains) in registers; the query tiles are unpacked once per workgroup into LDS, already in
// fragment order (a wave's ds_read_b128 of a fragment is 1 KB linear), and every fragment read feeds four MFMAs.  Both operands use
// the same bit -> (lane half, element) map, so the k order inside a fragment cancels out of the dot product.  The epilogue turns an
// accumulator into the packed key ((|q_i| - 2 dot) << 16 | i) with one v_mad_i32_i24 and keeps a signed running minimum per lane
// (|t_j| is the same for the whole column and is added at the end): ties go to the smaller query index, as BFMatcher's strict '<'
// scan does.  The per-query reduction "closest among the trains that chose me, first train wins a tie" is one LDS atomicMin on the
// packed key (distance << 20 | train index).  Integer work, exact.  [round 1-2: xor + v_bcnt on the vector ALU, 24 instructions
// per pair of descriptors at 91 % VALU utilisation: 0.64 ms per 1000 frames]
#include "bh_internal.h"

#define MT       512           // lanes per workgroup
#define MM_QC    384           // queries staged in LDS at a time (12 tiles of 32)
#define MM_TT    4             // train tiles per wave
#define MM_TPR   (MT / 64 * MM_TT * 32)   // trains per round (1024)
#define NQ_MAX   8192          // per-query result slots in LDS
#define NT_MAX   (1 << 20)

typedef int mm_v4 __attribute__((ext_vector_type(4)));
typedef int mm_v16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mt_wave_excl_scan(int v, int lane)
{
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(s, d, 64); if (lane >= d) s += o; }
    return s - v;
}

// 16 bits -> 16 bytes of 0 / 1 (bit k -> byte k): a nibble times 0x204081 puts bit m at 8 m
__device__ __forceinline__ mm_v4 mm_unpack16(uint32_t b)
{
    mm_v4 r;
    r.x = (int)((((b) & 15u) * 0x00204081u) & 0x01010101u);
    r.y = (int)((((b >> 4) & 15u) * 0x00204081u) & 0x01010101u);
    r.z = (int)((((b >> 8) & 15u) * 0x00204081u) & 0x01010101u);
    r.w = (int)((((b >> 12) & 15u) * 0x00204081u) & 0x01010101u);
    return r;
}

__global__ __launch_bounds__(MT) void k_match(const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int q_stride,
                                              const uint8_t* __restrict__ t, const int* __restrict__ nt_arr, int t_stride,
                                              int* __restrict__ qidx, int* __restrict__ tidx, int* __restrict__ dist,
                                              int* __restrict__ n_match, int out_stride, uint32_t* __restrict__ status)
{
    __shared__ mm_v4    s_a[(MM_QC / 32) * 8 * 64];             // [query tile][k step][lane]: 96 KB
    __shared__ int      s_kq[MM_QC];                            // |q_i| << 16 | i
    __shared__ uint32_t s_best[NQ_MAX];
    __shared__ int      s_wcnt[MT / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
    const size_t b = blockIdx.x;
    int nq = nq_arr[b], nt = nt_arr[b];
    if (nq > NQ_MAX || nt > NT_MAX) { if (tid == 0) { atomicOr(status, 4u); n_match[b] = 0; } return; }
    if (nq <= 0 || nt <= 0) { if (tid == 0) n_match[b] = 0; return; }
    const uint8_t* qbase = q + b * (size_t)q_stride * 32;
    const uint4* qb = reinterpret_cast<const uint4*>(qbase);
    const uint4* tb = reinterpret_cast<const uint4*>(t + b * (size_t)t_stride * 32);

    for (int i = tid; i < nq; i += MT) s_best[i] = 0xffffffffu;
    const int nchunks = (nq + MM_QC - 1) / MM_QC;
    const int rounds = (nt + MM_TPR - 1) / MM_TPR;
    for (int r = 0; r < rounds; r++) {
        // my four train tiles: fragments of the 8 k steps (bits 32 s + 16 h ... of train lane & 31) and |t_j|
        mm_v4 bf[MM_TT][8];
        int pt[MM_TT], best[MM_TT];
#pragma unroll
        for (int tt = 0; tt < MM_TT; tt++) {
            const int j = r * MM_TPR + (wv * MM_TT + tt) * 32 + (lane & 31);
            uint4 ta = make_uint4(0, 0, 0, 0), tc = ta;
            if (j < nt) { ta = tb[2 * j]; tc = tb[2 * j + 1]; }
            const uint32_t w[8] = { ta.x, ta.y, ta.z, ta.w, tc.x, tc.y, tc.z, tc.w };
            int pc = 0;
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) { pc += __popc(w[s2]); bf[tt][s2] = mm_unpack16((w[s2] >> (16 * h)) & 0xffffu); }
            pt[tt] = pc; best[tt] = 0x7fffffff;
        }
        for (int c = 0; c < nchunks; c++) {
            const int q0 = c * MM_QC, qn = min(MM_QC, nq - q0), nrows = (qn + 31) & ~31;
            if (nchunks > 1 || r == 0) {
                __syncthreads();
                // (row, half of the descriptor) -> eight fragments.  Consecutive lanes take consecutive ROWS, so the 16-byte stores of one
                // instruction are contiguous (round 4 had a lane per (row, 16-bit slice) with the slice on the fast index: the sixteen lanes
                // of a row stored 512 B apart - one set of four banks - and SQ_LDS_BANK_CONFLICT was three times the LDS-active cycles)
                for (int it = tid; it < nrows * 2; it += MT) {
                    const int hf = it >= nrows ? 1 : 0, i = it - hf * nrows;
                    uint4 u = make_uint4(0, 0, 0, 0);
                    if (i < qn) u = qb[2 * (q0 + i) + hf];
                    const uint32_t w4[4] = { u.x, u.y, u.z, u.w };
                    mm_v4* dst = s_a + ((i >> 5) * 8 + 4 * hf) * 64 + (i & 31);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        dst[j * 64] = mm_unpack16(w4[j] & 0xffffu);
                        dst[j * 64 + 32] = mm_unpack16(w4[j] >> 16);
                    }
                }
                for (int i = tid; i < nrows; i += MT) {
                    int kv = 0x3fffffff;                                    // padding rows never win
                    if (i < qn) {
                        const uint4 u = qb[2 * (q0 + i)], v = qb[2 * (q0 + i) + 1];
                        kv = ((__popc(u.x) + __popc(u.y) + __popc(u.z) + __popc(u.w) + __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w)) << 16) | (q0 + i);
                    }
                    s_kq[i] = kv;
                }
                __syncthreads();
            }
            for (int qt = 0; qt < (nrows >> 5); qt++) {
                mm_v16 acc[MM_TT];
#pragma unroll
                for (int tt = 0; tt < MM_TT; tt++) acc[tt] = mm_v16{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
                for (int s2 = 0; s2 < 8; s2++) {
                    const mm_v4 af = s_a[(qt * 8 + s2) * 64 + lane];
#pragma unroll
                    for (int tt = 0; tt < MM_TT; tt++) acc[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[tt][s2], acc[tt], 0, 0, 0);
                }
                // C layout: column = lane & 31 (the train), row = (reg & 3) + 8 (reg >> 2) + 4 h (the query)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int4 kv = *reinterpret_cast<const int4*>(&s_kq[qt * 32 + 8 * g + 4 * h]);
#pragma unroll
                    for (int tt = 0; tt < MM_TT; tt++) {
                        best[tt] = min(best[tt], __mul24(acc[tt][4 * g], -131072) + kv.x);
                        best[tt] = min(best[tt], __mul24(acc[tt][4 * g + 1], -131072) + kv.y);
                        best[tt] = min(best[tt], __mul24(acc[tt][4 * g + 2], -131072) + kv.z);
                        best[tt] = min(best[tt], __mul24(acc[tt][4 * g + 3], -131072) + kv.w);
                    }
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < MM_TT; tt++) {
            const int j = r * MM_TPR + (wv * MM_TT + tt) * 32 + (lane & 31);
            const int bk = min(best[tt], __shfl_xor(best[tt], 32, 64));
            if (h == 0 && j < nt) atomicMin(&s_best[bk & 0xffff], ((uint32_t)((bk >> 16) + pt[tt]) << 20) | (uint32_t)j);
        }
    }
    __syncthreads();
    // ascending-queryIdx compaction
    const int per = (nq + MT - 1) / MT;
    const int i0 = tid * per;
    int cnt = 0;
    for (int e = 0; e < per; e++) { const int i = i0 + e; if (i < nq && s_best[i] != 0xffffffffu) cnt++; }
    const int wpre = mt_wave_excl_scan(cnt, lane);
    if (lane == 63) s_wcnt[wv] = wpre + cnt;
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < MT / 64; i++) { const int c = s_wcnt[i]; if (i < wv) base += c; total += c; }
    int pos = base + wpre;
    const size_t ob = b * (size_t)out_stride;
    for (int e = 0; e < per; e++) {
        const int i = i0 + e;
        if (i < nq) {
            const uint32_t k = s_best[i];
            if (k != 0xffffffffu) {
                if (pos < out_stride) { qidx[ob + pos] = i; tidx[ob + pos] = (int)(k & 0xfffffu); dist[ob + pos] = (int)(k >> 20); }
                pos++;
            }
        }
    }
    if (tid == 0) { if (total > out_stride) { atomicOr(status, 8u); total = out_stride; } n_match[b] = total; }
}

extern "C" int bh_match_hamming_crosscheck(bh_ctx* ctx, const uint8_t* q, const int* nq, int q_stride,
                                           const uint8_t* t, const int* nt, int t_stride, int B,
                                           int* qidx, int* tidx, int* dist, int* n_match, int out_stride, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!q || !t || !nq || !nt || !qidx || !tidx || !dist || !n_match || B < 0 || q_stride < 0 || t_stride < 0 ||
        out_stride < 1 || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_match_hamming_crosscheck: bad argument");
    if (B == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    BH_TRY(bh_status_ready(ctx));
    const uint8_t* d_q = q; const uint8_t* d_t = t; const int* d_nq = nq; const int* d_nt = nt;
    int *d_qi = qidx, *d_ti = tidx, *d_d = dist, *d_nm = n_match;
    if (mem == BH_MEM_HOST) {
        // host mode: rows actually used per pair may be fewer than the stride; copy the strided blocks whole
        size_t qbytes = 0, tbytes = 0;
        int maxq = 0, maxt = 0;
        for (int b = 0; b < B; b++) {
            if (nq[b] < 0 || nt[b] < 0) return bh_set_err(ctx, BH_ERR_INVALID, "negative descriptor count");
            if (nq[b] > NQ_MAX || nt[b] > NT_MAX) return bh_set_err(ctx, BH_ERR_INVALID, "descriptor count above the supported maximum (nq <= 8192, nt <= 2^20)");
            maxq = nq[b] > maxq ? nq[b] : maxq; maxt = nt[b] > maxt ? nt[b] : maxt;
        }
        if ((B > 1 && (q_stride < maxq || t_stride < maxt))) return bh_set_err(ctx, BH_ERR_INVALID, "stride smaller than a pair's row count");
        qbytes = ((size_t)(B - 1) * q_stride + maxq) * 32; tbytes = ((size_t)(B - 1) * t_stride + maxt) * 32;
        const size_t ob = sizeof(int) * (size_t)B * out_stride;
        BH_TRY(bh_reserve(ctx, &ctx->st_a, qbytes + 32)); BH_TRY(bh_reserve(ctx, &ctx->st_b, tbytes + 32));
        BH_TRY(bh_reserve(ctx, &ctx->st_c, sizeof(int) * 2 * B));
        BH_TRY(bh_reserve(ctx, &ctx->st_d, ob * 3 + sizeof(int) * B));
        if (qbytes) BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, q, qbytes, hipMemcpyHostToDevice, st));
        if (tbytes) BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, t, tbytes, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_c.p, nq, sizeof(int) * B, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync((int*)ctx->st_c.p + B, nt, sizeof(int) * B, hipMemcpyHostToDevice, st));
        d_q = (const uint8_t*)ctx->st_a.p; d_t = (const uint8_t*)ctx->st_b.p;
        d_nq = (const int*)ctx->st_c.p; d_nt = d_nq + B;
        d_qi = (int*)ctx->st_d.p; d_ti = d_qi + (size_t)B * out_stride; d_d = d_ti + (size_t)B * out_stride;
        d_nm = d_d + (size_t)B * out_stride;
    }
    BH_LAUNCH(ctx, BH_K_MATCH, k_match, dim3(B), dim3(MT), st, d_q, d_nq, q_stride, d_t, d_nt, t_stride, d_qi, d_ti, d_d, d_nm,
                       out_stride, (uint32_t*)ctx->status.p);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        const size_t ob = sizeof(int) * (size_t)B * out_stride;
        BH_HIP(ctx, hipMemcpyAsync(qidx, d_qi, ob, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(tidx, d_ti, ob, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(dist, d_d, ob, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(n_match, d_nm, sizeof(int) * B, hipMemcpyDeviceToHost, st));
        return bh_ctx_status(ctx);                                 // synchronises; reads and clears the sticky status word
    }
    return BH_OK;
}

// ------------------------------------------------------------------------------------------------ map-point medoid descriptor
// Replaces the Python double loop of MapPoint.add_observation — slam/nodes.py:142-154: for every
// map point, among its k <= 10 stored observation descriptors pick the one whose column of the (uint8) Hamming distance
// matrix has the smallest median (numpy semantics: mean of the two middle values for even k; first minimum).
// One lane per map point: 45 xor+popcount distances, a 10-element insertion sort per column, integer compare of 2*median.
#define MEDOID_K 10
__device__ __forceinline__ int medoid_index(const uint8_t* __restrict__ desc10, int k)
{
    const uint4* d = reinterpret_cast<const uint4*>(desc10);
    int D[MEDOID_K][MEDOID_K];
#pragma unroll
    for (int a = 0; a < MEDOID_K; a++) {
        D[a][a] = 0;
#pragma unroll
        for (int b = a + 1; b < MEDOID_K; b++) {
            int v = 0;
            if (b < k) {
                const uint4 p0 = d[2 * a], p1 = d[2 * a + 1], q0 = d[2 * b], q1 = d[2 * b + 1];
                v = __popc(p0.x ^ q0.x) + __popc(p0.y ^ q0.y) + __popc(p0.z ^ q0.z) + __popc(p0.w ^ q0.w) +
                    __popc(p1.x ^ q1.x) + __popc(p1.y ^ q1.y) + __popc(p1.z ^ q1.z) + __popc(p1.w ^ q1.w);
                v &= 255;                                   // the reference keeps the matrix in uint8
            }
            D[a][b] = v; D[b][a] = v;
        }
    }
    int best = 0, best_m2 = 1 << 30;
#pragma unroll
    for (int c = 0; c < MEDOID_K; c++) {
        if (c < k) {
            int col[MEDOID_K];
#pragma unroll
            for (int r = 0; r < MEDOID_K; r++) col[r] = r < k ? D[r][c] : 1 << 20;     // padding sorts to the end
#pragma unroll
            for (int a = 0; a < MEDOID_K - 1; a++)                                     // odd-even style fixed network (bubble passes)
#pragma unroll
                for (int b = 0; b < MEDOID_K - 1 - a; b++) { const int lo = min(col[b], col[b + 1]), hi = max(col[b], col[b + 1]); col[b] = lo; col[b + 1] = hi; }
            int m2 = 0;
#pragma unroll
            for (int r = 0; r < MEDOID_K; r++) {
                if ((k & 1) && r == k / 2) m2 = 2 * col[r];
                if (!(k & 1) && (r == k / 2 - 1 || r == k / 2)) m2 += col[r];
            }
            if (m2 < best_m2) { best_m2 = m2; best = c; }
        }
    }
    return best;
}
__global__ __launch_bounds__(256) void k_medoid(const uint8_t* __restrict__ desc, const int* __restrict__ k_arr, int M,
                                                int* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int k = min(k_arr[i], MEDOID_K);
    out[i] = k <= 0 ? -1 : medoid_index(desc + (size_t)i * MEDOID_K * 32, k);
}

// MapPoint.add_observation as a whole - slam/nodes.py:142-156 - for M map points that the new keyframe observes (the loop body of
// CovisibilityGraph.add_kf, slam/covisibility_graph.py:125-136, for the features already associated with a map point):
//   n /= norm(n);  n_obs += 1;  if len(obs) < 10: obs.append(feat), feat = obs[argmin(median(dists, axis=0))];
//   self.n = (n + (n_obs - 1) * self.n) / n_obs
// One lane per map point; arrays are updated in place.
// row i of the map-point arrays takes the observation (descriptor nd, viewing direction (n0, n1, n2), not normalised); obs_kf (may be
// null) records the observing keyframe beside the stored descriptor, as self.obs does
__device__ __forceinline__ void mp_add_obs_row(size_t i, uint8_t* __restrict__ obs, int* __restrict__ obs_kf, int* __restrict__ obs_cnt, int* __restrict__ n_obs,
                                               double* __restrict__ normal, uint8_t* __restrict__ feat, const uint8_t* __restrict__ nd,
                                               double n0, double n1, double n2, int kf_id)
{
    const double nrm = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
    n0 = n0 / nrm; n1 = n1 / nrm; n2 = n2 / nrm;
    const int no = n_obs[i] + 1;
    n_obs[i] = no;
    int k = obs_cnt[i];
    if (k < MEDOID_K) {
        uint8_t* slot = obs + (i * MEDOID_K + k) * 32;
        const uint4* src = reinterpret_cast<const uint4*>(nd);
        reinterpret_cast<uint4*>(slot)[0] = src[0]; reinterpret_cast<uint4*>(slot)[1] = src[1];
        if (obs_kf) obs_kf[i * MEDOID_K + k] = kf_id;
        k++;
        obs_cnt[i] = k;
        const int m = medoid_index(obs + i * MEDOID_K * 32, k);
        const uint4* md = reinterpret_cast<const uint4*>(obs + (i * MEDOID_K + m) * 32);
        reinterpret_cast<uint4*>(feat + i * 32)[0] = md[0]; reinterpret_cast<uint4*>(feat + i * 32)[1] = md[1];
    }
    const double w = (double)(no - 1), dn = (double)no;
    normal[3 * i] = (n0 + w * normal[3 * i]) / dn;
    normal[3 * i + 1] = (n1 + w * normal[3 * i + 1]) / dn;
    normal[3 * i + 2] = (n2 + w * normal[3 * i + 2]) / dn;
}
__global__ __launch_bounds__(256) void k_mp_add_obs(uint8_t* __restrict__ obs, int* __restrict__ obs_cnt, int* __restrict__ n_obs,
                                                    double* __restrict__ normal, uint8_t* __restrict__ feat,
                                                    const uint8_t* __restrict__ new_desc, const double* __restrict__ new_n, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    mp_add_obs_row((size_t)i, obs, nullptr, obs_cnt, n_obs, normal, feat, new_desc + (size_t)i * 32, new_n[3 * (size_t)i], new_n[3 * (size_t)i + 1], new_n[3 * (size_t)i + 2], 0);
}

// ------------------------------------------------------------------------------------------------ add_kf: the per-feature loop as a whole
// CovisibilityGraph.add_kf (slam/covisibility_graph.py:123-136) for ONE keyframe against a device-resident map-point table whose row index
// is the map point's id (MapPoint.cnt, slam/nodes.py:106-110):
//   for i_feat, (feat, d3, id_mp) in enumerate(zip(kf.des, kf.cloud_kp, kf.des2mp)):
//       n = frame.t - d3; norm = |n|; if norm > 0:
//           id_mp == -1 or id_mp not in self.mps -> MapPoint(feat, d3, n / norm, kf.id): id = cnt++, des2mp[i_feat] = id
//           else                                 -> self.mps[id_mp].add_observation(feat, n, kf.id)
// One workgroup: the creations take their ids in feature order (a block-wide scan of the create flags), and two features tied to the SAME
// existing map point (a map point reaches the tracker's candidate list once per keyframe that sees it) are applied one after the other,
// in feature order: occurrence r of an id runs in round r.
#define KFA_T 1024
__global__ __launch_bounds__(KFA_T) void k_kf_add(int kf_id, double t0, double t1, double t2, int N, const uint8_t* __restrict__ des, const double* __restrict__ cloud,
                                                  int* __restrict__ des2mp, int cap, int next_id, uint8_t* __restrict__ alive, double* __restrict__ pt3d,
                                                  double* __restrict__ normal, uint8_t* __restrict__ feat, uint8_t* __restrict__ obs, int* __restrict__ obs_kf,
                                                  int* __restrict__ obs_cnt, int* __restrict__ n_obs, int* __restrict__ first_kf, int* __restrict__ action,
                                                  int* __restrict__ out /* n_created, capacity flag */, int* __restrict__ ids /* scratch int[N] */)
{
    __shared__ int s_w[KFA_T / 64], s_carry, s_maxr, s_ovf;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_carry = 0; s_maxr = 0; s_ovf = 0; }
    __syncthreads();
    // ---- pass 1: classify, create (ids in feature order)
    for (int i0 = 0; i0 < N; i0 += KFA_T) {
        const int i = i0 + tid;
        bool valid = false, create = false;
        double n0 = 0, n1 = 0, n2 = 0, nrm = 0;
        int id = -1;
        if (i < N) {
            n0 = t0 - cloud[3 * (size_t)i]; n1 = t1 - cloud[3 * (size_t)i + 1]; n2 = t2 - cloud[3 * (size_t)i + 2];
            nrm = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
            valid = nrm > 0.0;
            id = des2mp[i];
            // id >= next_id cannot be a map point of the caller's map (ids come from the counter): "not in self.mps", whatever rows earlier
            // chunks of THIS call have just made alive - the verdict must not depend on the chunk size
            create = valid && (id < 0 || id >= next_id || id >= cap || !alive[id]);
        }
        const unsigned long long bm = __ballot(create);
        const int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __popcll(bm);
        __syncthreads();
        int off = s_carry, tot = 0;
        for (int w = 0; w < KFA_T / 64; w++) { const int c = s_w[w]; if (w < wv) off += c; tot += c; }
        if (i < N) {
            int act = 0, keep_id = -1;
            if (create) {
                const int nid = next_id + off + before;
                if (nid < cap) {
                    const size_t r = (size_t)nid;
                    // MapPoint.__init__ normalises the direction it is handed once more: n = (n / norm) / |n / norm|
                    double a0 = n0 / nrm, a1 = n1 / nrm, a2 = n2 / nrm;
                    const double m2 = sqrt((a0 * a0 + a1 * a1) + a2 * a2);
                    normal[3 * r] = a0 / m2; normal[3 * r + 1] = a1 / m2; normal[3 * r + 2] = a2 / m2;
                    pt3d[3 * r] = cloud[3 * (size_t)i]; pt3d[3 * r + 1] = cloud[3 * (size_t)i + 1]; pt3d[3 * r + 2] = cloud[3 * (size_t)i + 2];
                    const uint4* src = reinterpret_cast<const uint4*>(des + (size_t)i * 32);
                    reinterpret_cast<uint4*>(feat + r * 32)[0] = src[0]; reinterpret_cast<uint4*>(feat + r * 32)[1] = src[1];
                    reinterpret_cast<uint4*>(obs + r * MEDOID_K * 32)[0] = src[0]; reinterpret_cast<uint4*>(obs + r * MEDOID_K * 32)[1] = src[1];
                    obs_kf[r * MEDOID_K] = kf_id; obs_cnt[r] = 1; n_obs[r] = 1; first_kf[r] = kf_id; alive[r] = 1;
                    des2mp[i] = nid; act = 1;
                } else s_ovf = 1;                                    // table full: reported, the feature keeps its old association
            } else if (valid) { act = 2; keep_id = id; }
            action[i] = act;
            ids[i] = keep_id;                                        // >= 0: an observation of that map point is due
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    // ---- pass 2: observations; occurrence rank of each feature among the earlier features with the same id
    for (int i0 = 0; i0 < N; i0 += KFA_T) {
        const int i = i0 + tid;
        if (i < N && ids[i] >= 0) {
            const int id = ids[i];
            int r = 0;
            for (int j = 0; j < i; j++) r += ids[j] == id ? 1 : 0;
            action[i] = 2 | (r << 8);
            atomicMax(&s_maxr, r);
        }
    }
    __syncthreads();
    const int maxr = s_maxr;
    for (int r = 0; r <= maxr; r++) {
        for (int i0 = 0; i0 < N; i0 += KFA_T) {
            const int i = i0 + tid;
            if (i < N && ids[i] >= 0 && (action[i] >> 8) == r) {
                const size_t ci = (size_t)i;
                mp_add_obs_row((size_t)ids[i], obs, obs_kf, obs_cnt, n_obs, normal, feat, des + ci * 32,
                               t0 - cloud[3 * ci], t1 - cloud[3 * ci + 1], t2 - cloud[3 * ci + 2], kf_id);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i0 = 0; i0 < N; i0 += KFA_T) { const int i = i0 + tid; if (i < N) action[i] &= 255; }
    if (tid == 0) { out[0] = min(s_carry, max(cap - next_id, 0)); out[1] = s_ovf; }
}

extern "C" int bh_medoid_descriptor(bh_ctx* ctx, const uint8_t* desc, const int* k, int M, int* medoid_idx, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!desc || !k || !medoid_idx || M < 0 || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_medoid_descriptor: bad argument");
    if (M == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint8_t* d_desc = desc; const int* d_k = k; int* d_out = medoid_idx;
    if (mem == BH_MEM_HOST) {
        BH_TRY(bh_reserve(ctx, &ctx->st_a, (size_t)M * MEDOID_K * 32)); BH_TRY(bh_reserve(ctx, &ctx->st_b, sizeof(int) * 2 * (size_t)M));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_a.p, desc, (size_t)M * MEDOID_K * 32, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(ctx->st_b.p, k, sizeof(int) * M, hipMemcpyHostToDevice, st));
        d_desc = (const uint8_t*)ctx->st_a.p; d_k = (const int*)ctx->st_b.p; d_out = (int*)ctx->st_b.p + M;
    }
    BH_LAUNCH(ctx, BH_K_MATCH, k_medoid, dim3((M + 255) / 256), dim3(256), st, d_desc, d_k, M, d_out);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(medoid_idx, d_out, sizeof(int) * M, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

extern "C" int bh_mp_add_observation(bh_ctx* ctx, int M, uint8_t* obs_desc, int* obs_cnt, int* n_obs, double* normal, uint8_t* feat,
                                     const uint8_t* new_desc, const double* new_n, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!obs_desc || !obs_cnt || !n_obs || !normal || !feat || !new_desc || !new_n || M < 0 || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_mp_add_observation: bad argument");
    if (M == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint8_t *d_obs = obs_desc, *d_feat = feat; int *d_cnt = obs_cnt, *d_no = n_obs; double* d_nrm = normal; const uint8_t* d_nd = new_desc; const double* d_nn = new_n;
    const size_t ob = (size_t)M * MEDOID_K * 32, fb = (size_t)M * 32, ib = sizeof(int) * (size_t)M, nb = sizeof(double) * 3 * (size_t)M;
    if (mem == BH_MEM_HOST) {
        for (int i = 0; i < M; i++) if (obs_cnt[i] < 0 || obs_cnt[i] > MEDOID_K) return bh_set_err(ctx, BH_ERR_INVALID, "bh_mp_add_observation: obs_cnt outside [0, 10]");
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        BH_TRY(bh_reserve(ctx, &ctx->st_a, al(ob) + 2 * al(fb) + 2 * al(ib) + 2 * al(nb)));
        char* p = (char*)ctx->st_a.p;
        d_obs = (uint8_t*)p; p += al(ob); d_feat = (uint8_t*)p; p += al(fb); uint8_t* nd = (uint8_t*)p; p += al(fb);
        d_cnt = (int*)p; p += al(ib); d_no = (int*)p; p += al(ib); d_nrm = (double*)p; p += al(nb); double* nn = (double*)p;
        BH_HIP(ctx, hipMemcpyAsync(d_obs, obs_desc, ob, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_feat, feat, fb, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(nd, new_desc, fb, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_cnt, obs_cnt, ib, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(d_no, n_obs, ib, hipMemcpyHostToDevice, st)); BH_HIP(ctx, hipMemcpyAsync(d_nrm, normal, nb, hipMemcpyHostToDevice, st));
        BH_HIP(ctx, hipMemcpyAsync(nn, new_n, nb, hipMemcpyHostToDevice, st));
        d_nd = nd; d_nn = nn;
    }
    BH_LAUNCH(ctx, BH_K_MATCH, k_mp_add_obs, dim3((M + 255) / 256), dim3(256), st, d_obs, d_cnt, d_no, d_nrm, d_feat, d_nd, d_nn, M);
    BH_HIP(ctx, hipGetLastError());
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(obs_desc, d_obs, ob, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(feat, d_feat, fb, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(obs_cnt, d_cnt, ib, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(n_obs, d_no, ib, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(normal, d_nrm, nb, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipStreamSynchronize(st));
    }
    return BH_OK;
}

extern "C" int bh_kf_add_map_points(bh_ctx* ctx, int kf_id, const double frame_t[3], int N, const uint8_t* des, const double* cloud_kp, int* des2mp,
                                    int cap, int next_id, uint8_t* alive, double* pt3d, double* normal, uint8_t* feat, uint8_t* obs_desc, int* obs_kf,
                                    int* obs_cnt, int* n_obs, int* first_kf, int* action, int* n_created, int mem)
{
    if (!ctx) return BH_ERR_INVALID;
    if (!frame_t || !des || !cloud_kp || !des2mp || !alive || !pt3d || !normal || !feat || !obs_desc || !obs_kf || !obs_cnt || !n_obs || !first_kf || !action ||
        !n_created || N < 0 || cap < 1 || next_id < 0 || next_id > cap || (mem != BH_MEM_HOST && mem != BH_MEM_DEVICE))
        return bh_set_err(ctx, BH_ERR_INVALID, "bh_kf_add_map_points: bad argument");
    *n_created = 0;
    if (N == 0) return BH_OK;
    BH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t c = (size_t)cap, n = (size_t)N;
    // per-call scratch: the pending-observation ids and the two result words (+ in host mode everything else)
    const size_t tab_b = al(c) + 2 * al(sizeof(double) * 3 * c) + al(c * 32) + al(c * MEDOID_K * 32) + al(sizeof(int) * MEDOID_K * c) + 3 * al(sizeof(int) * c);
    const size_t in_b = al(n * 32) + al(sizeof(double) * 3 * n) + 2 * al(sizeof(int) * n);
    BH_TRY(bh_reserve(ctx, &ctx->st_b, al(sizeof(int) * n) + 256 + (mem == BH_MEM_HOST ? tab_b + in_b : 0)));
    char* p = (char*)ctx->st_b.p;
    int* ids = (int*)p; p += al(sizeof(int) * n);
    int* d_out = (int*)p; p += 256;
    const uint8_t* d_des = des; const double* d_cl = cloud_kp; int* d_d2m = des2mp; int* d_act = action;
    uint8_t *d_alive = alive, *d_feat = feat, *d_obs = obs_desc; double *d_pt = pt3d, *d_nr = normal; int *d_okf = obs_kf, *d_oc = obs_cnt, *d_no = n_obs, *d_fk = first_kf;
    if (mem == BH_MEM_HOST) {                                      // (test / small-map convenience: the whole table goes up and comes back; the live use keeps it on the device)
        for (int i = 0; i < N; i++) if (des2mp[i] < -1) return bh_set_err(ctx, BH_ERR_INVALID, "bh_kf_add_map_points: des2mp entries are -1 or ids");
        auto up = [&](const void* src, size_t bytes) -> char* { char* d = p; p += al(bytes); return hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) == hipSuccess ? d : nullptr; };
        char* q;
#define KFA_UP(dst, type, src, bytes) if (!(q = up(src, bytes))) return bh_set_err(ctx, BH_ERR_HIP, "bh_kf_add_map_points: upload failed"); dst = (type)q
        KFA_UP(d_des, const uint8_t*, des, n * 32); KFA_UP(d_cl, const double*, cloud_kp, sizeof(double) * 3 * n); KFA_UP(d_d2m, int*, des2mp, sizeof(int) * n);
        d_act = (int*)p; p += al(sizeof(int) * n);
        KFA_UP(d_alive, uint8_t*, alive, c); KFA_UP(d_pt, double*, pt3d, sizeof(double) * 3 * c); KFA_UP(d_nr, double*, normal, sizeof(double) * 3 * c);
        KFA_UP(d_feat, uint8_t*, feat, c * 32); KFA_UP(d_obs, uint8_t*, obs_desc, c * MEDOID_K * 32); KFA_UP(d_okf, int*, obs_kf, sizeof(int) * MEDOID_K * c);
        KFA_UP(d_oc, int*, obs_cnt, sizeof(int) * c); KFA_UP(d_no, int*, n_obs, sizeof(int) * c); KFA_UP(d_fk, int*, first_kf, sizeof(int) * c);
#undef KFA_UP
    }
    BH_LAUNCH(ctx, BH_K_MATCH, k_kf_add, dim3(1), dim3(KFA_T), st, kf_id, frame_t[0], frame_t[1], frame_t[2], N, d_des, d_cl, d_d2m, cap, next_id, d_alive, d_pt, d_nr, d_feat,
              d_obs, d_okf, d_oc, d_no, d_fk, d_act, d_out, ids);
    BH_HIP(ctx, hipGetLastError());
    int ho[2] = { 0, 0 };
    BH_HIP(ctx, hipMemcpyAsync(ho, d_out, sizeof(ho), hipMemcpyDeviceToHost, st));
    if (mem == BH_MEM_HOST) {
        BH_HIP(ctx, hipMemcpyAsync(des2mp, d_d2m, sizeof(int) * n, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(action, d_act, sizeof(int) * n, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(alive, d_alive, c, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(pt3d, d_pt, sizeof(double) * 3 * c, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(normal, d_nr, sizeof(double) * 3 * c, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(feat, d_feat, c * 32, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(obs_desc, d_obs, c * MEDOID_K * 32, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(obs_kf, d_okf, sizeof(int) * MEDOID_K * c, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(obs_cnt, d_oc, sizeof(int) * c, hipMemcpyDeviceToHost, st)); BH_HIP(ctx, hipMemcpyAsync(n_obs, d_no, sizeof(int) * c, hipMemcpyDeviceToHost, st));
        BH_HIP(ctx, hipMemcpyAsync(first_kf, d_fk, sizeof(int) * c, hipMemcpyDeviceToHost, st));
    }
    BH_HIP(ctx, hipStreamSynchronize(st));                      // the count of new map points is the caller's next MapPoint.cnt
    *n_created = ho[0];
    if (ho[1]) return bh_set_err(ctx, BH_ERR_CAPACITY, "bh_kf_add_map_points: the map-point table is full (%d rows)", cap);
    return BH_OK;
}

